#!/bin/bash
# GPU box: kernel timeline of the slab pipeline on one rank (RCCL self-exchange), to look at overlap and gaps.
# Usage: tools/trace_slab.sh <tag> [bench args]  -> gpurun_out/trace_<tag>/timeline.txt
set -u
TAG=${1:-slab}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --force-slab --steps 12 --warmup 3 --cpu-steps 0 --profile-steps 0 --mech-grid 0 $* > $OUT/trace.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
# last ~2 substeps worth of kernels
tail = rows[-70:]
t0 = tail[0][0]
with open(out + "/timeline.txt", "w") as fh:
    for s, e, n, q, st in tail:
        fh.write(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  q={q} s={st}  {n}\n")
print(open(out + "/timeline.txt").read())
PY
find $OUT -name "*kernel_trace.csv" -delete
