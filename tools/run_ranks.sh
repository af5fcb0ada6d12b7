#!/bin/bash
# tools/run_ranks.sh <nranks> <case> [key=value ...]: start tests/slab_rank_worker.py once per rank on this box's GPU 0
set -u
N=$1; shift
HERE=$(cd "$(dirname "$0")" && pwd)
JOB=mrlrun_$$
pids=()
for ((r=0; r<N; r++)); do
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python3 "$HERE/../tests/slab_rank_worker.py" $JOB $N $r "$@" 2>/dev/null | grep RESULT &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=1; done
rm -f /dev/shm/$JOB
exit $rc
