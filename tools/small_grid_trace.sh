#!/bin/bash
# kernel trace of the small-grid substep chain (tools/small_grid_bench.py): which launches a 2-D substep consists of, and the gaps
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/small_grid_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/small_grid_bench.py > $OUT/trace.log 2>&1
python3 $R/tools/gap_census.py $OUT/trace 2 > $OUT/gaps.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete
grep substeps $OUT/trace.log
head -30 $OUT/gaps.txt
