# same-box A/B of two builds (build/ab/*.so) on the kernels that the LDS map / staged twiddles touch
cd /root/repo
cp marlin_amd/lib/libmarlin_hip.so /tmp/_keep.so
for rep in 1 2; do for v in build/ab/a_before.so build/ab/b_after.so; do
  cp $v marlin_amd/lib/libmarlin_hip.so
  echo "== $(basename $v)"
  python tools/slab_local_bench.py 8 256 20 1 0 1 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(' slab-local 512^3/8:', [(k['kernel'], round(k['avg_ms']*1e3,1)) for k in d['kernels'] if 'z' in k['kernel']])"
  for g in 256 512 200 128; do python bench.py --grid $g --steps 20 --warmup 3 --cpu-steps 0 --mech-grid 0 --no-variants 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(' serial', d['config']['grid'][0], round(d['ms_per_step'],4), [(k['kernel'], k['avg_ms']) for k in d['kernels'] if '_z_' in k['kernel']])"; done
  python tools/mech_bench.py 128 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(' mech128', round(d['ms_per_cg_iteration'],4), round(d['small_strain_linear_elastic']['ms_per_cg_iteration'],4), [(k['kernel'], k['avg_ms']) for k in d['kernels'] if '_z_' in k['kernel']])"
done; done
cp /tmp/_keep.so marlin_amd/lib/libmarlin_hip.so
