#!/bin/bash
# A/B several builds of the library over several commands inside one gpurun: tools/ab_cmds.sh <dir with *.so> "<cmd1>" "<cmd2>" ...
D=$1; shift
cp marlin_amd/lib/libmarlin_hip.so /tmp/_keep.so
for rep in 1 2; do
for v in $(ls $D/*.so); do
  cp $v marlin_amd/lib/libmarlin_hip.so
  for c in "$@"; do
    echo "== $(basename $v) :: $c :: $(bash -c "$c" 2>/dev/null | tail -1 | cut -c1-3000)"
  done
done
done
cp /tmp/_keep.so marlin_amd/lib/libmarlin_hip.so
