#!/usr/bin/env python3
"""Time one Cahn-Hilliard AB2 substep on arbitrary grids (generic any-length path vs power-of-two fast path).
usage: generic_bench.py nx ny [nz]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marlin_amd.api import Context, ch_params  # noqa: E402


def main():
    shape = [int(a) for a in sys.argv[1:]]
    dim = len(shape)
    ctx = Context(dim, shape, [3.0] * dim)
    p = ch_params()
    torch.manual_seed(0)
    c = [(torch.rand(shape, dtype=torch.float64) * 0.12 + 0.44).cuda(), None]
    c[1] = torch.empty_like(c[0])
    Nh = [ctx.empty_hist(), ctx.empty_hist()]
    ctx.ch_substep(p, c[0], c[1], Nh[0], [], 0, 1e-3)
    i = 1
    steps = 50
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.ch_substep(p, c[i], c[1 - i], Nh[i], [Nh[1 - i]], 1, 1e-3)
            i = 1 - i
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    npts = 1
    for s in shape:
        npts *= s
    print(json.dumps({"shape": shape, "ms_per_substep": dt * 1e3, "G_updates_per_s": npts / dt / 1e9}))


if __name__ == "__main__":
    main()
