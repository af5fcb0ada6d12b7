#!/bin/bash
# Compile-time register census of the HIP sources (no GPU needed): kernels that spill VGPRs, worst first.
# usage: tools/spill_census.sh [file.hip ...] [-- extra hipcc flags]      (default: the four fused-kernel files)
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/marlin_amd/csrc
FILES=(); EXTRA=()
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; EXTRA=("$@"); break; fi; FILES+=("$1"); shift; done
[ ${#FILES[@]} -eq 0 ] && FILES=(ch_fused.hip slab_fused.hip mech_fused.hip slab_mech_fused.hip)
for f in "${FILES[@]}"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I. -I$R/build/csrc -munsafe-fp-atomics "${EXTRA[@]}" \
    -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/_census.o 2>&1 | python3 -c "
import sys, re
name = None; n = 0; worst = []
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m: name = m.group(1); n += 1
    m = re.search(r'VGPRs Spill: (\d+)', l)
    if m and int(m.group(1)) > 0: worst.append((int(m.group(1)), name))
worst.sort(reverse=True)
print('$f: %d kernels, %d spill' % (n, len(worst)))
for s, k in worst[:12]: print('   %3d  %s' % (s, k[:110]))
"
done
