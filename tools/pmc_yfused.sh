#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the fused slab y pass at 512^3 / 8 for experiment masks given as arguments (default: product, dense planes,
# idle lanes loading column 0)  -> gpurun_out/pmc_yfused.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_yfused
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for e in ${@:-0 8388608 33554432}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/e${e}_$c -- python3 $R/tools/slab_local_bench.py 8 256 4 1 0 1 $e > $OUT/e${e}_$c.log 2>&1
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
root = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(root + "/e*/**/*counter_collection.csv", recursive=True):
    m = re.search(r"/e(\d+)_(\w+_SIZE)/", f)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("mrl::p2::", "")[:44]
        if not k.startswith("k_"): continue
        a = agg[(k, m.group(1), row["Counter_Name"])]
        a[0] += 1; a[1] += float(row["Counter_Value"])
for (k, e, c), (n, t) in sorted(agg.items()):
    print("%-46s exp %-9s %-10s %8.1f MB per launch (%d)" % (k, e, c, t / n * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e6, n))
PY
find $OUT -name "*counter_collection.csv" -delete
