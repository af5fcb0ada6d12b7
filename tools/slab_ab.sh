#!/bin/bash
# same-box A/B of the rank-local slab kernels of 512^3 / 8: experiment masks given as arguments (default: dense planes vs padded)
for rep in 1 2; do for e in ${@:-8388608 0}; do
python tools/slab_local_bench.py 8 256 30 1 0 1 $e 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read())
ks=[k for k in j['kernels'] if k['bytes_per_launch']>0]
print('exp',j['exp'],'sum of kernels %.4f ms'%sum(k['avg_ms']*k['launches']/10 for k in ks),' '.join('%s %.4f'%(k['kernel'].replace('slab_',''),k['avg_ms']) for k in ks))"
done; done
