for rep in 1 2; do for e in 0 268435456; do python tools/slab_local_bench.py 8 256 30 1 0 1 $e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slab-local exp', d['exp'], round(d['ms_per_substep_local_incl_copies'],4), ' '.join('%s=%.1f' % (k['kernel'][5:], k['avg_ms']*1e3) for k in d['kernels']))"; done; done
for rep in 1; do for e in 0; do python bench.py --grid 512 --exp $e --steps 30 --warmup 5 --cpu-steps 0 --mech-grid 0 --no-variants --profile-steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('512^3 exp', $e, round(d['ms_per_step'],4), ' '.join('%s=%.1f' % (k['kernel'][3:], k['avg_ms']*1e3) for k in d['kernels']))"; done; done
