#!/bin/bash
# Run on the GPU box (via gpurun): HBM-traffic PMC passes of the rank-local slab kernels (tools/slab_local_bench.py).
# Usage: tools/profile_slab_local.sh <tag> [slab_local_bench args...]  -> gpurun_out/prof_<tag>/summary.md
set -u
TAG=${1:-slab}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="${*:-8 256 20 2 1 1}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/slab_local_bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/tools/slab_local_bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/tools/slab_local_bench.py $ARGS > $OUT/write.log 2>&1
python3 $R/tools/summarize_prof.py $OUT > $OUT/summary.md 2>&1
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
cat $OUT/summary.md
