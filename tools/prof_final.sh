# final measurement of a round on the GPU box: counters first (so that bench.py finds them taken on the sources it times), then the bench lines
cd /root/repo
TAG=${1:-r03c}
tools/profile_gpu.sh $TAG > gpurun_out/prof_$TAG.log 2>&1
cp gpurun_out/traffic_$TAG.json profiles/traffic_ch256.json
python bench.py > gpurun_out/${TAG}_bench_256.json 2> gpurun_out/${TAG}_bench_256.err
python bench.py --gpus 2 --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_2ranks_one_gpu.json
python bench.py --workload mech --steps 2 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_mech128.json
tail -c 600 gpurun_out/${TAG}_bench_256.json
