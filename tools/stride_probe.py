#!/usr/bin/env python3
"""Which stride a strided FFT pass tolerates: plain r2c transforms of [nx][ny][nz] grids whose x / y passes have different line
strides; prints GB/s per pass (algorithmic bytes / HIP-event time).  usage: stride_probe.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marlin_amd.api import Context  # noqa: E402

for shape in ([512, 64, 512], [64, 512, 512], [512, 512, 64], [256, 256, 256], [256, 64, 256], [512, 8, 512], [512, 512, 512]):
    ctx = Context(3, shape, [1.0, 1.0, 1.0])
    a = torch.rand(shape, dtype=torch.float64, device="cuda")
    for _ in range(3):
        s = ctx.fft(a)
    ctx.set_profiling(True)
    for _ in range(10):
        s = ctx.fft(a)
    torch.cuda.synchronize()
    prof = {k["kernel"]: round(k["bytes_per_launch"] / (k["ms"] / k["launches"]) / 1e6) for k in ctx.get_profile() if k["launches"]}
    nzc = shape[2] // 2 + 1
    print(json.dumps({"shape": shape, "x_stride_KB": shape[1] * nzc * 16 / 1024, "y_stride_KB": nzc * 16 / 1024, "GBps": prof}))
    ctx.close()
