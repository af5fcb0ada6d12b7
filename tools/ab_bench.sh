#!/bin/bash
# A/B timing of kernel variants inside ONE gpurun (box-to-box variance is ~3 %): experiment masks (MRL_OPT_EXPERIMENT, bench.py --exp) as arguments
for e in "$@"; do
  python bench.py --exp $e --steps 300 --warmup 30 --cpu-steps 0 --mech-grid 0 --profile-steps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('EXP=$e', round(d['ms_per_step'],4), ' '.join('%s=%.1f' % (k['kernel'][3:], k['avg_ms']*1e3) for k in d['kernels']))"
done
