#!/usr/bin/env python3
"""Latency of one Cahn-Hilliard substep on small grids (the reference's regression sizes), inside one mrl_ch_substeps call of 1000
substeps: 128^2 / 200^2 19.7 us, 512^2 30 us, 64^3 32 us -- a chain of dependent launches of a handful of workgroups each."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marlin_amd.api import Context, ch_params
for shape in ([128,128],[200,200],[512,512],[64,64,64]):
    ctx = Context(len(shape), shape, [3.0]*len(shape)); p = ch_params()
    c = [torch.rand(shape, dtype=torch.float64, device='cuda')*0.12+0.44, None]; c[1]=torch.empty_like(c[0])
    Nh=[ctx.empty_hist(), ctx.empty_hist()]
    head,n_old = 1,0
    head,n_old = ctx.ch_substeps(p, c[0], c[1], Nh, head, n_old, 2, 10, True, 1e-3)
    torch.cuda.synchronize(); t0=time.perf_counter()
    head,n_old = ctx.ch_substeps(p, c[1], c[0], Nh, (head+1)%2, 1, 2, 1000, True, 1e-3)
    torch.cuda.synchronize(); t=(time.perf_counter()-t0)/1000
    print(shape, 'substeps call: %.2f us/substep' % (t*1e6))
