#!/bin/bash
# rocprofv3 kernel trace of the mechanics benchmark (config C): gpurun_out/prof_mech_<tag>/
set -u
TAG=${1:-r01}; N=${2:-128}; SLAB=${3:-0}   # SLAB=1: the one-rank slab job (library-owned pipeline)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_mech_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/mech_bench.py $N 2 $SLAB > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/tools/mech_bench.py $N 1 $SLAB > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/tools/mech_bench.py $N 1 $SLAB > $OUT/write.log 2>&1
python3 $R/tools/summarize_prof.py $OUT > $OUT/summary.md 2>&1
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
tail -1 $OUT/trace.log > $OUT/bench.json
cat $OUT/summary.md
