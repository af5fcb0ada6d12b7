#!/bin/bash
# Two-stage plans (fft_two.h, fft_two_z.h) against the uniform 30- / 20-point plans they replace (experiment bit 1 << 29), same box, same process
# layout: one mrl_ch_substeps call of 20 substeps per time step, best of 5 calls, wall clock around the call + the per-kernel profile.
cd "$(dirname "$0")/.."
for s in "240 240 240" "180 180 180" "160 160 160" "150 150 150" "120 120 120" "240 160 120" "300 300 300" "320 320 320" "400 400 400" "256 256 256" "200 200 200"; do
  python tools/ch_substeps_bench.py $s 20 5 0 2>&1 | tail -1
  python tools/ch_substeps_bench.py $s 20 5 0x20000000 2>&1 | tail -1
done
