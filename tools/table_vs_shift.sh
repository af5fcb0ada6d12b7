# evidence for DESIGN 4.2a: (i) table- against shift-addressed kernels on the same shape, (ii) the per-point ratio rank-local / serial at
# about 2 M points per rank for a shift-addressed and a table-addressed partition
cd /root/repo
summ='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ks={k["kernel"]: round(k["avg_ms"]*1e3,1) for k in d["kernels"]}
print(d["global_grid"], "P", d["P"], "exp", d["exp"], "per-substep kernels (us):", ks, "A+B+C+EZ =", round(sum(v for n,v in ks.items() if n in ("slab_A_x_fwd","slab_B_y_fused","slab_C_x_inv","slab_EZ_z_inv_fwd")),1))'
for exp in 0 16777216 0 16777216; do python tools/slab_local_bench.py 8 256 20 1 0 1 $exp 2>/dev/null | python3 -c "$summ"; done
python tools/slab_local_bench.py 4 128 20 1 0 1 0 2>/dev/null | python3 -c "$summ"
python tools/slab_local_bench.py 4 128 20 1 0 1 16777216 2>/dev/null | python3 -c "$summ"
python tools/slab_local_bench.py 4 200 20 1 0 1 0 200,200,200 2>/dev/null | python3 -c "$summ"
python tools/ch_shape_bench.py 256 256 128 30 2>/dev/null | tail -1 | cut -c1-700
python tools/ch_shape_bench.py 200 200 200 30 2>/dev/null | tail -1 | cut -c1-700
