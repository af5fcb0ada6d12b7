#!/bin/bash
# LDS bank-conflict share of the mechanics kernels (mech_bench.py 128 1) -> gpurun_out/pmc_lds_mech/summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_lds_mech
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/m -- python3 $R/tools/mech_bench.py ${1:-128} 1 > $OUT/m.log 2>&1
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$OUT/m/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void mrl::p2::", "").replace("void mrl::", "")[:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0)):
    a = v.get("SQ_LDS_IDX_ACTIVE", 0)
    if a <= 0: continue
    n = cnt[(k, "SQ_LDS_IDX_ACTIVE")]
    print("%-72s launches %4d  LDS cycles/launch %12.0f  bank-conflict %5.1f %%" % (k, n, a / n, 100 * v.get("SQ_LDS_BANK_CONFLICT", 0) / a))
PY
find $OUT -name "*counter_collection.csv" -size +2M -delete
cat $OUT/summary.txt
