#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + the two HBM-traffic PMC passes of bench.py.
# Usage: tools/profile_gpu.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/{trace,fetch,write}
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 30 --warmup 5 --cpu-steps 0 --profile-steps 0 --mech-grid 0 --no-variants $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $ARGS > $OUT/write.log 2>&1
find $OUT -name "*.csv" | head -20
# keep only the small summaries (kernel_stats + a per-kernel aggregation of the counter CSVs)
python3 $R/tools/summarize_prof.py $OUT > $OUT/summary.md 2>&1
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
cat $OUT/summary.md
[ -f $OUT/traffic.json ] && cp $OUT/traffic.json $OUT/../traffic_$TAG.json
