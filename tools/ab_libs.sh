#!/bin/bash
# A/B several builds of the library inside one gpurun: tools/ab_libs.sh <dir with *.so> <command...>; restores the first one
D=$1; shift
cp marlin_amd/lib/libmarlin_hip.so /tmp/_keep.so
for v in $(ls $D/*.so) $(ls $D/*.so); do
  cp $v marlin_amd/lib/libmarlin_hip.so
  echo "== $(basename $v)"; "$@" 2>/dev/null | tail -1 | cut -c1-6000
done
cp /tmp/_keep.so marlin_amd/lib/libmarlin_hip.so
