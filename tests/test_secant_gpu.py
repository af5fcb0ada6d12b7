"""SecantSolver's fused k-space kernels (mrl_secant_begin / mrl_secant_iterate) against the oracle and the reference's
rotating_grain_secant gold file."""
import math

import pytest
import torch

import oracle.marlin_oracle as mo
from tests.conftest import load_golden
from tests.test_oracle_golden import rotating_grain_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("with_L,damping", [(True, 1.0), (False, 1.0), (True, 0.6)])
def test_secant_kernels_match_torch_ops(with_L, damping):
    torch.manual_seed(5)
    from marlin_amd.api import Context
    ctx = Context(2, [24, 20], [3.0, 2.0])
    dom = mo.Domain(2, [24, 20], [3.0, 2.0])
    rs = dom.rshape
    rnd = lambda: torch.randn(rs, dtype=torch.complex128)
    u, N, u_old, u_prev, R_prev = rnd(), rnd(), rnd(), rnd(), rnd()
    R_prev.view(-1)[3] = 0.0
    L = -torch.rand(rs, dtype=torch.float64) * 5.0 if with_L else None
    dt, eps = 0.37, 1e-4
    # begin
    R0w = (N + L * u) * dt if with_L else N * dt
    gw = (u + eps * N) / (1.0 - eps * L) if with_L else u + eps * N
    d = lambda t: None if t is None else t.cuda()
    R0, guess, n0 = ctx.secant_begin(d(u), d(N), d(L), dt, eps)
    assert (R0.cpu() - R0w).abs().max().item() <= 1e-14 * R0w.abs().max().item()
    assert (guess.cpu() - gw).abs().max().item() <= 1e-14 * gw.abs().max().item()
    assert abs(n0 - torch.norm(R0w).item()) <= 1e-13 * n0
    # iterate; make one entry hit the dy == 0 branch
    Rw = (N + L * u) * dt + u_old - u if with_L else N * dt + u_old - u
    R_prev.view(-1)[7] = Rw.view(-1)[7]
    dx, dy = u - u_prev, Rw - R_prev
    duw = torch.where(dy != 0, -Rw * dx / dy, 0.0)
    assert duw.view(-1)[7] == 0
    unw = u + duw if damping == 1.0 else u + duw * damping
    Rp = d(R_prev.clone())
    un, nR, ndu = ctx.secant_iterate(d(u), d(N), d(L), d(u_old), d(u_prev), Rp, dt, damping)
    assert torch.equal(Rp.cpu(), Rw) or (Rp.cpu() - Rw).abs().max().item() <= 1e-15 * Rw.abs().max().item()
    assert ((un.cpu() - unw).abs() / (1.0 + unw.abs())).max().item() <= 1e-13
    assert abs(nR - torch.norm(Rw).item()) <= 1e-13 * nR
    assert abs(ndu - torch.norm(duw).item()) <= 1e-12 * ndu


def test_rotating_grain_secant_gold():
    """the reference's regression case driven from Python over the C ABI (ParsedCompute -> hiprtc kernel, fft, secant kernels)"""
    from marlin_amd.api import Context, ParsedCompute
    g = load_golden("rotating_grain_secant_gold.npz")
    w = 6
    ext = [w * math.pi * 2, w * math.pi * 2 / math.sin(math.pi / 3)]
    ctx = Context(2, [40, 40], ext)
    dom, _, _, variables = rotating_grain_problem(torch.from_numpy(g["psi.0"]))
    L = variables[0][2].cuda().contiguous()
    psi3 = ParsedCompute(ctx, "0.20*psi^2-psi^3", ["psi"])
    psi = torch.from_numpy(g["psi.0"]).cuda()
    ts = mo.IterationAdaptiveDT(1.0, 100, 400, 1.4, 0.9, 500.0)
    its = 0
    for step in range(1, 11):
        sub_dt = ts.next_dt(step, its) / 3
        for _ in range(3):
            ubar, Nbar = ctx.fft(psi), ctx.fft(psi3(psi))
            u_old, u_prev = ubar, ubar
            R_prev, guess, R0n = ctx.secant_begin(ubar, Nbar, L, sub_dt, 1e-4)
            psi = ctx.ifft(guess)
            converged = False
            for its in range(30):
                ubar, Nbar = ctx.fft(psi), ctx.fft(psi3(psi))
                u_new, Rn, _ = ctx.secant_iterate(ubar, Nbar, L, u_old, u_prev, R_prev, sub_dt)
                u_prev = ubar
                psi = ctx.ifft(u_new)
                if Rn < 1e-9 or Rn / R0n < 1e-9:
                    converged = True
                    break
            assert converged
        assert (psi.cpu() - torch.from_numpy(g[f"psi.{step}"])).abs().max().item() <= 1e-10
