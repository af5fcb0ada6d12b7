#!/bin/bash
# Run on the GPU box: PMC passes over tools/slab_local_bench.py that compare the forward and the inverse slab x pass
# (write-request sizes and stalls, read-request sizes and stalls, L2 hit rate, address-unit stalls).  -> gpurun_out/pmc_<tag>/summary.txt
# Budget ≈ 3-5 min per pass, four passes: give gpurun --timeout 1500.
set -u
TAG=${1:-slabpass}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="8 256 4 1 0 1"   # few substeps: every launch is serialised and replayed under --pmc (the 10-substep version needed > 5 min per pass)
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/slab_local_bench.py $ARGS > $OUT/p$i.log 2>&1
done
python3 - "$OUT" > $OUT/summary.txt <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0].replace("mrl::p2::", "").replace("void ", "")[:48]
            if not k.startswith("k_"):
                continue
            a = agg[k][row["Counter_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        n, tot = agg[k][c]
        print(f"    {c:44s} {tot / n:16.1f} per launch ({n} launches)")
PY
find $OUT -name "*counter_collection.csv" -delete
cat $OUT/summary.txt
