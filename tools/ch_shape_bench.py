#!/usr/bin/env python3
"""Per-kernel times of the fused Cahn-Hilliard substep for an arbitrary (planned) grid shape: ch_shape_bench.py nx ny nz [steps]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marlin_amd.api import Context, ch_params  # noqa: E402


def main():
    shape = [int(v) for v in sys.argv[1:4]]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
    ctx = Context(3, shape, [float(s) * 0.1256 for s in shape])
    p = ch_params()
    g = torch.Generator(device="cuda").manual_seed(1)
    c = [torch.rand(shape, dtype=torch.float64, device="cuda", generator=g) * 0.12 + 0.44, None]
    c[1] = torch.empty_like(c[0])
    Nh = [ctx.empty_hist(), ctx.empty_hist()]
    for k in range(steps + 5):
        if k == 5:
            torch.cuda.synchronize()
            ctx.set_profiling(True)
        ctx.ch_substep(p, c[k % 2], c[1 - k % 2], Nh[k % 2], [Nh[1 - k % 2]] if k else [], 1 if k else 0, 1e-3)
    torch.cuda.synchronize()
    prof = [k for k in ctx.get_profile() if k["launches"]]
    npts = shape[0] * shape[1] * shape[2]
    tot = sum(k["ms"] / k["launches"] for k in prof)
    print(json.dumps({"shape": shape, "sum_ms": round(tot, 4), "G_updates_per_s": round(npts / tot / 1e6, 2),
                      "kernels": {k["kernel"]: [round(k["ms"] / k["launches"] * 1e3, 1), round(k["bytes_per_launch"] / (k["ms"] / k["launches"]) / 1e9, 2)]
                                  for k in prof}}))


if __name__ == "__main__":
    main()
