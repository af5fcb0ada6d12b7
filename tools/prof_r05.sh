#!/bin/bash
# Round-5 evidence in ONE gpurun call, so that bench lines and kernel traces of a configuration come from the same box and the same
# job (a kernel sum above the timed step is not evidence): for 256^3 (headline), 512^3 on one GPU, the rank-local
# kernels of 512^3 / 8 and the 128^3 / 256^3 mechanics solve -- first the plain bench line (JSON), then the rocprofv3 kernel trace +
# FETCH / WRITE counter passes of the same command.  Everything lands under gpurun_out/; the summaries are copied to profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 5 > $O/r05_bench_256_driver_command.json 2> $O/r05_bench_256_driver_command.err
python3 bench.py > $O/r05_bench_256_default.json 2> $O/r05_bench_256_default.err
bash tools/profile_gpu.sh r05_256
python3 bench.py --grid 512 --steps 20 --warmup 3 --cpu-steps 0 --mech-grid 0 --parity-substeps '' > $O/r05_bench_512_1gpu.json 2> $O/r05_bench_512_1gpu.err
bash tools/profile_gpu.sh r05_512 --grid 512 --steps 12 --warmup 3
python3 tools/slab_local_bench.py 8 256 40 1 0 1 0 > $O/r05_slab_local_512_over_8.json 2>/dev/null
bash tools/profile_slab_local.sh r05_slab_local 8 256 40 1 0 1 0
python3 bench.py --workload mech --steps 5 > $O/r05_bench_mech128.json 2>/dev/null
python3 bench.py --workload mech --grid 256 --steps 3 > $O/r05_bench_mech256.json 2>/dev/null
bash tools/profile_mech.sh r05 128 0
python3 bench.py --gpus 2 --device 0 --steps 20 --warmup 5 > $O/r05_bench_2ranks_one_gpu.json 2> $O/r05_bench_2ranks_one_gpu.err
ls -la $O | tail -30
