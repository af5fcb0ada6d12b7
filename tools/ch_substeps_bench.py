#!/usr/bin/env python3
"""Cahn-Hilliard substeps through ONE mrl_ch_substeps call per time step (the product's default) for an arbitrary grid shape:
ch_substeps_bench.py nx ny nz [substeps per call] [calls] [MRL_OPT_EXPERIMENT bits] -> ms per substep by wall clock around the calls and the per-kernel profile"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marlin_amd.api import Context, ch_params  # noqa: E402


def main():
    shape = [int(v) for v in sys.argv[1:4]]
    nsub = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    calls = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    exp = int(sys.argv[6], 0) if len(sys.argv) > 6 else 0
    ctx = Context(3, shape, [float(s) * 0.1256 for s in shape])
    if exp:
        from marlin_amd import api
        ctx.set_option(api.OPT_EXPERIMENT, exp)
    p = ch_params()
    g = torch.Generator(device="cuda").manual_seed(1)
    c = torch.rand(shape, dtype=torch.float64, device="cuda", generator=g) * 0.12 + 0.44
    out = torch.empty_like(c)
    pred = 2
    ring = [ctx.empty_hist() for _ in range(pred)]
    head, n_old = 0, 0
    times = []
    for k in range(calls + 2):
        if k == 2:
            ctx.set_profiling(True)
        if k > 0:
            head, n_old = (head + 1) % pred, min(n_old + 1, pred - 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        head, n_old = ctx.ch_substeps(p, c, out, ring, head, n_old, pred, nsub, k > 0, 1e-3)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3 / nsub)
        c, out = out, c
    ctx.set_profiling(False)
    prof = [k for k in ctx.get_profile() if k["launches"]]
    npts = shape[0] * shape[1] * shape[2]
    best = min(times[2:])
    print(json.dumps({"shape": shape, "experiment": exp, "substeps_per_call": nsub, "ms_per_substep": round(best, 4), "G_updates_per_s": round(npts / best / 1e6, 2),
                      "kernels": {k["kernel"]: [round(k["ms"] / k["launches"] * 1e3, 1), round(k["bytes_per_launch"] / (k["ms"] / k["launches"]) / 1e9, 2)]
                                  for k in prof}}))


if __name__ == "__main__":
    main()
