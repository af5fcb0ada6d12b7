#!/usr/bin/env python3
"""Wall time of the first and the second call into each part of libmarlin_hip.so in a fresh process (code-object load, plan set-up,
work-array allocation): what a MOOSE run pays once, and what every rank process of the test-suite pays again."""
import math
import os
import sys
import time

t00 = time.perf_counter()
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marlin_amd.api import Context, ch_params  # noqa: E402

t_import = time.perf_counter() - t00


def timed(label, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    print(f"{label:58s} {1e3 * (time.perf_counter() - t0):9.1f} ms", flush=True)
    return r


print(f"{'import torch + marlin_amd':58s} {1e3 * t_import:9.1f} ms")
timed("torch.cuda init (first allocation)", lambda: torch.zeros(4, device="cuda"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shape = [n, n, n]
ctx = timed("Context(3, n^3)", lambda: Context(3, shape, [2 * math.pi] * 3))
a = torch.rand(shape, dtype=torch.float64, device="cuda")
for k in range(2):
    s = timed(f"fft #{k + 1}", lambda: ctx.fft(a))
for k in range(2):
    timed(f"ifft #{k + 1}", lambda: ctx.ifft(s))
p = ch_params()
c = a * 0.12 + 0.44
out = torch.empty_like(c)
N0, N1 = ctx.empty_hist(), ctx.empty_hist()
timed("ch_substep #1 (AB1)", lambda: ctx.ch_substep(p, c, out, N0, [], 0, 1e-3))
timed("ch_substep #2 (AB2)", lambda: ctx.ch_substep(p, out, c, N1, [N0], 1, 1e-3))
timed("ch_substep #3 (AB2)", lambda: ctx.ch_substep(p, c, out, N0, [N1], 1, 1e-3))
K = torch.full(shape, 0.833, dtype=torch.float64, device="cuda")
mu = torch.full(shape, 0.386, dtype=torch.float64, device="cuda")
K[: n // 4, : n // 4, : n // 4] = 8.33
mu[: n // 4, : n // 4, : n // 4] = 3.86
F = torch.eye(3, dtype=torch.float64, device="cuda").expand(shape + [3, 3]).contiguous()
app = torch.zeros(3, 3, dtype=torch.float64, device="cuda")
app[0, 1] = 0.001
for k in range(2):
    timed(f"mech_newton_cg #{k + 1}", lambda: ctx.mech_newton_cg(F, K, mu, app, l_tol=1e-2, nl_rel_tol=2e-2, nl_abs_tol=2e-2))
A = torch.rand(shape + [3, 3], dtype=torch.float64, device="cuda")
for k in range(2):
    timed(f"gamma_apply #{k + 1}", lambda: ctx.gamma_apply(A))
