#!/bin/bash
# The bench lines of tools/prof_r05.sh once more, after its counter files have been installed under profiles/ (bench.py quotes
# profiles/traffic_*.json and says whether they were taken on the kernel sources it runs)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 5 > $O/r05_bench_256_driver_command.json 2> $O/r05_bench_256_driver_command.err
python3 bench.py > $O/r05_bench_256_default.json 2> $O/r05_bench_256_default.err
python3 bench.py --grid 512 --steps 20 --warmup 3 --cpu-steps 0 --mech-grid 0 --parity-substeps '' > $O/r05_bench_512_1gpu.json 2> $O/r05_bench_512_1gpu.err
python3 bench.py --workload mech --steps 5 > $O/r05_bench_mech128.json 2>/dev/null
python3 bench.py --workload mech --grid 256 --steps 3 > $O/r05_bench_mech256.json 2>/dev/null
