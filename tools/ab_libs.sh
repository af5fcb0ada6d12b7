#!/bin/bash
# A/B two builds of the library inside one gpurun: tools/ab_libs.sh <dir with old.so new.so> <command...>
D=$1; shift
for v in old new old new; do
  cp $D/$v.so marlin_amd/lib/libmarlin_hip.so
  echo "== $v"; "$@" 2>/dev/null | tail -1 | cut -c1-900
done
cp $D/new.so marlin_amd/lib/libmarlin_hip.so
