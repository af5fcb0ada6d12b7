#!/bin/bash
# LDS bank-conflict share of the 512-point and the 256-point kernels: SQ_LDS_BANK_CONFLICT (extra cycles) against SQ_LDS_IDX_ACTIVE (all
# LDS cycles), per kernel.  -> gpurun_out/pmc_lds/summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_lds
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/slab -- python3 $R/tools/slab_local_bench.py 8 256 4 1 0 1 > $OUT/slab.log 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/serial -- python3 $R/bench.py --steps 4 --warmup 2 --cpu-steps 0 --profile-steps 0 --mech-grid 0 --no-variants > $OUT/serial.log 2>&1
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
for tag in ("slab", "serial"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void mrl::p2::", "")[:70]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    print("==", tag)
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0)):
        a = v.get("SQ_LDS_IDX_ACTIVE", 0)
        if a <= 0 or not k.startswith("k_"): continue
        n = cnt[(k, "SQ_LDS_IDX_ACTIVE")]
        print("%-72s launches %3d  LDS cycles/launch %12.0f  bank-conflict %5.1f %%  unaligned-stall %5.1f %%  addr-conflict %5.1f %%" % (
            k, n, a / n, 100 * v.get("SQ_LDS_BANK_CONFLICT", 0) / a, 100 * v.get("SQ_LDS_UNALIGNED_STALL", 0) / a, 100 * v.get("SQ_LDS_ADDR_CONFLICT", 0) / a))
PY
find $OUT -name "*counter_collection.csv" -size +2M -delete
cat $OUT/summary.txt
