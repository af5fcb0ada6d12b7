#!/usr/bin/env python3
"""MRL_OPT_CACHE_CHUNK_MB on the serial fused Cahn-Hilliard path: ms per substep and bit-identity against the unchunked schedule.
usage: chunk_sweep.py [nx,ny,nz = 512,512,512] [substeps = 24] [budgets MB = 0,32,64,96,128,160,192 (0 = off)]"""
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import splitmix64_uniform  # noqa: E402
from marlin_amd import api  # noqa: E402


def main():
    shape = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "512,512,512").split(",")]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    budgets = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,32,64,96,128,160,192").split(",")]
    dx = 8.0 * math.pi / 200.0
    ctx = api.Context(3, shape, [n * dx for n in shape])
    p = api.ch_params()
    c0 = torch.from_numpy(splitmix64_uniform(shape[0] * shape[1] * shape[2]).reshape(shape)).cuda()
    ref = None
    out = []
    for rep in range(2):
        for b in budgets:
            ctx.set_option(api.OPT_CACHE_CHUNK_MB, b)
            ring = [ctx.empty_hist(), ctx.empty_hist()]
            for r in ring:
                r.zero_()
            a, c = torch.empty_like(c0), torch.empty_like(c0)
            h, n = ctx.ch_substeps(p, c0, a, ring, 1, 0, 2, 4, True, 1e-3)
            ctx.sync()
            t0 = time.perf_counter()
            ctx.ch_substeps(p, a, c, ring, (h + 1) % 2, 1, 2, k, True, 1e-3)
            ctx.sync()
            ms = (time.perf_counter() - t0) / k * 1e3
            if ref is None:
                ref = c.clone()
            same = bool(torch.equal(c, ref))
            if rep == 1:
                out.append({"budget_mb": b, "ms_per_substep": round(ms, 4), "bit_identical_to_first": same})
            del ring, a, c
    print(json.dumps({"shape": shape, "substeps": k, "runs": out}))


if __name__ == "__main__":
    main()
