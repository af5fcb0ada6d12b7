#!/bin/bash
# A/B of bench.py flag sets inside one gpurun (same box, interleaved): tools/ab_bench_flags.sh "<flags A>" "<flags B>" [reps]
A=$1; B=$2; R=${3:-3}
for rep in $(seq 1 $R); do
  for f in "$A" "$B"; do
    python bench.py --steps 40 --warmup 10 --cpu-steps 0 --mech-grid 0 --no-variants $f 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('[%s]' % sys.argv[1], 'ms/step %.4f' % j['ms_per_step'], ' '.join('%s %.4f' % (k['kernel'].replace('ch_',''), k['avg_ms']) for k in j['kernels']))
" "$f"
  done
done
