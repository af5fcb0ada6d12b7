#!/bin/bash
# same-box A/B of the first-generation stagger (experiment bits 29-30) on the rank-local kernels of 512^3 / 8, interleaved
for rep in 1 2; do
for e in 0 536870912 1073741824 1610612736; do
  python tools/slab_local_bench.py 8 256 40 1 0 1 $e | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ks={k['kernel']:k['avg_ms']*1e3 for k in d['kernels']}
print('exp', d['exp']>>29, ' '.join(f'{n}={v:.1f}' for n,v in ks.items()), 'sum_local_ms', round(d['ms_per_substep_local_incl_copies'],4))
"
done; done
