cd /root/repo
tools/profile_gpu.sh r03b > gpurun_out/prof_r03b.log 2>&1
python bench.py > gpurun_out/r03b_bench_256.json 2> gpurun_out/r03b_bench_256.err
python bench.py --grid 512 --steps 20 --warmup 3 --cpu-steps 0 --mech-grid 0 > gpurun_out/r03b_bench_512_1gpu.json 2>/dev/null
python tools/mech_bench.py 256 2 > gpurun_out/r03b_mech256_bench.json 2>/dev/null
python tools/mech_bench.py 128 2 > gpurun_out/r03b_mech128_bench.json 2>/dev/null
tools/profile_slab_local.sh r03b_slab 8 256 20 1 0 1 > gpurun_out/prof_r03b_slab.log 2>&1
marlin_amd/lib/marlin-hip-bench workload=ch gpus=1 slab=1 grid=256 steps=40 warmup=5 2>/dev/null | grep '^{' > gpurun_out/r03b_native_one_rank_slab_256.json
python bench.py --gpus 4 --steps 10 --warmup 3 --profile-steps 4 2>/dev/null | tail -1 > gpurun_out/r03b_bench_4ranks_one_gpu.json
python bench.py --workload mech --gpus 2 --grid 64 --steps 2 2>/dev/null | tail -1 > gpurun_out/r03b_bench_mech_2ranks_one_gpu.json
python tools/slab_local_bench.py 4 200 20 1 0 1 0 200,200,200 > gpurun_out/r03b_slab_local_200_over_4.txt 2>&1
marlin_amd/lib/marlin-hip-bench workload=mech gpus=2 grid=40 steps=2 2>/dev/null | grep '^{' > gpurun_out/r03b_native_mech_2ranks_table_path_40.json
ls -la gpurun_out/ | tail -15
