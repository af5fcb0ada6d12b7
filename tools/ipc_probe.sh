#!/bin/bash
# tools/ipc_probe.sh <nranks> [same_device] [MiB per peer slot] [MiB of a second buffer] [GiB of ballast]: one process per rank on the GPU(s) of this box; results on stdout
set -u
N=${1:-2}; SAME=${2:-1}; MIB=${3:-32}; MIB2=${4:-0}; EXTRA=${5:-0}
HERE=$(cd "$(dirname "$0")" && pwd)
BIN=$HERE/../marlin_amd/lib/ipc_probe
[ -x "$BIN" ] && [ "$BIN" -nt "$HERE/ipc_probe.hip" ] || /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o "$BIN" "$HERE/ipc_probe.hip" -lrt || exit 1
NAME=/mrl_probe_$$
truncate -s 16384 /dev/shm$NAME
pids=()
for ((r=0; r<N; r++)); do
  MRL_PROBE_RANK=$r MRL_PROBE_SHM=$NAME timeout 60 "$BIN" "$N" "$SAME" "$MIB" "$MIB2" "$EXTRA" &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=1; done
rm -f /dev/shm$NAME
[ $rc = 0 ] && echo "probe PASSED ($N ranks)" || echo "probe FAILED ($N ranks)"
exit $rc
