#!/bin/bash
# Timeline of the mechanics benchmark without the event-timed half: where does the stream idle between kernels?
set -u
N=${1:-128}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/mech_gaps_$N
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/mech_bench.py $N ${2:-3} 0 ${3:-0} 0 > $OUT/trace.log 2>&1
python3 $R/tools/gap_census.py $OUT/trace 8 > $OUT/gaps.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete
tail -1 $OUT/trace.log | cut -c1-400
cat $OUT/gaps.txt
