#!/bin/bash
# Run on the GPU box: memory-side counters of the 256^3 x-pass access pattern with the dense plane pitch (2064 pieces of 256 B) and with
# one piece of padding (2065), register-staged and LDS-DMA forms (tools/ldsdma_probe.hip mode 7).  -> gpurun_out/pmc_probe/summary.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_probe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pad in 0 16; do
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum"; do   # (a TA_* pass hung for 20 minutes under the persistent kernel: left out)
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $set --output-format csv -d $OUT/pad${pad}_p$i -- $R/marlin_amd/lib/ldsdma_probe 7 $pad > $OUT/pad${pad}_p$i.log 2>&1
  done
done
python3 - "$OUT" > $OUT/summary.txt <<'PY'
import csv, glob, sys, collections, re
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(root + "/pad*/**/*counter_collection.csv", recursive=True):
    pad = re.search(r"pad(\d+)_p", f).group(1)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
            if "k_move" not in k and "k_dma_move" not in k:
                continue
            a = agg[(k, pad)][row["Counter_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
for key in sorted(agg):
    print("%s   plane pitch + %s elements" % key)
    for c in sorted(agg[key]):
        n, tot = agg[key][c]
        print(f"    {c:44s} {tot / n:16.1f} per launch ({n} launches)")
PY
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*.db" -delete
cat $OUT/summary.txt
