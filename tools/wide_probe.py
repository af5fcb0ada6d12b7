#!/usr/bin/env python3
"""A/B of the wide N = 512 plan (experiment bit 1024) on plain transforms: agreement with the 16-point plan and GB/s per pass."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marlin_amd.api import Context  # noqa: E402

for shape in ([512, 64, 512], [64, 512, 512], [512, 512, 512]):
    res = {}
    out = {}
    a = torch.rand(shape, dtype=torch.float64, device="cuda")
    for exp in (0, 1024, 0, 1024):
        ctx = Context(3, shape, [1.0, 1.0, 1.0])
        ctx.set_option(0, exp)
        for _ in range(3):
            s = ctx.fft(a)
        ctx.set_profiling(True)
        for _ in range(10):
            s = ctx.fft(a)
        torch.cuda.synchronize()
        res.setdefault(exp, []).append({k["kernel"]: round(k["bytes_per_launch"] / (k["ms"] / k["launches"]) / 1e6) for k in ctx.get_profile() if k["launches"]})
        back = ctx.ifft(s)
        out[exp] = (s.clone(), (back - a).abs().max().item())
        ctx.close()
    diff = (out[0][0] - out[1024][0]).abs().max().item() / out[0][0].abs().max().item()
    print(json.dumps({"shape": shape, "rel_diff_of_spectra": diff, "roundtrip_err": [out[0][1], out[1024][1]], "GBps": {str(k): v for k, v in res.items()}}))
