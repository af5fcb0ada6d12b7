#!/bin/bash
# Round-4 evidence in ONE gpurun call: kernel traces + FETCH/WRITE counters of the headline bench (256^3), of the 512^3 one-GPU run and
# of the 128^3 mechanics solve, and the SQ / LDS counters of the fused z kernel before (tools/zpass_probe: k_ea = the round-3 kernel)
# and after (k_ea2 = lane-exchange pairing) the change.  Everything lands under gpurun_out/; the summaries are copied to profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash $R/tools/profile_gpu.sh r04_256
bash $R/tools/profile_gpu.sh r04_512 --grid 512 --steps 12 --warmup 3
bash $R/tools/profile_mech.sh r04 128 0
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_zpass_r04
mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/sq -- $R/marlin_amd/lib/zpass_probe > $OUT/sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/lds -- $R/marlin_amd/lib/zpass_probe > $OUT/lds.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $R/marlin_amd/lib/zpass_probe > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $R/marlin_amd/lib/zpass_probe > $OUT/write.log 2>&1
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
        if not k.startswith("k_ea"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
names = sorted({c for v in acc.values() for c in v})
print("per launch (mean over the probe's launches); FETCH_SIZE / WRITE_SIZE in KiB (FETCH x 2 on gfx950 for the byte count)")
print("%-44s " % "kernel" + " ".join("%16s" % n[:16] for n in names))
for k in sorted(acc):
    print("%-44s " % k + " ".join("%16.0f" % (acc[k][n] / max(1, cnt[(k, n)])) for n in names))
PY
find $OUT -name "*counter_collection.csv" -size +1M -delete
cat $OUT/summary.txt
