#!/usr/bin/env python3
"""The fp32 Cahn-Hilliard substep (mrl_ch_substeps_f32) alone, for kernel traces: usage f32_bench.py [n = 256] [substeps = 200] [warm-up substeps = 50] [timed calls = 1]"""
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marlin_amd.api import Context, ch_params  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dx = 8.0 * math.pi / 200.0
    ctx = Context(3, [n, n, n], [n * dx] * 3)
    p = ch_params()
    g = torch.Generator(device="cuda").manual_seed(0)
    a = (torch.rand(n, n, n, dtype=torch.float32, device="cuda", generator=g) * 0.12 + 0.44)
    b = torch.empty_like(a)
    ring = [ctx.empty_hist_f32(zero=True), ctx.empty_hist_f32(zero=True)]
    warm = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    h, no = ctx.ch_substeps_f32(p, a, b, ring, 1, 0, 2, warm, True, 1e-3)
    ctx.sync()
    runs = []
    for _ in range(int(sys.argv[4]) if len(sys.argv) > 4 else 1):
        t0 = time.perf_counter()
        h, no = ctx.ch_substeps_f32(p, b, a, ring, (h + 1) % 2, 1, 2, k, True, 1e-3)
        ctx.sync()
        runs.append(round((time.perf_counter() - t0) / k * 1e3, 5))
    ms = sorted(runs)[len(runs) // 2]
    print(json.dumps({"n": n, "substeps": k, "warmup": warm, "runs_ms": runs, "ms_per_substep": ms, "G_updates_per_s": n ** 3 / ms * 1e-6}))


if __name__ == "__main__":
    main()
