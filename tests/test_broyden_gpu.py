"""BroydenSolver's per-k kernels (mrl_broyden_*) against the oracle's restatement of src/tensor_solver/BroydenSolver.C
(the reference holds no regression data for this solver: parity unpinned, restatement vs HIP only)."""
import math

import pytest
import torch

import oracle.marlin_oracle as mo

pytestmark = pytest.mark.gpu


def _problem(n):
    """two coupled reaction-diffusion variables (the Brusselator sources of test/tests/solvers/diagonal.i) on n x n"""
    dom = mo.Domain(2, [n, n], [2.0 * math.pi] * 2)
    state = {"u": (1.0 + 0.1 * torch.sin(dom.axis[0]) * torch.sin(dom.axis[1])).expand(dom.shape).contiguous(),
             "v": (3.0 + 0.1 * torch.cos(dom.axis[0]) * torch.cos(2 * dom.axis[1])).expand(dom.shape).contiguous()}
    Du = mo.reciprocal_laplacian_factor(dom, 1e-2)
    Dv = mo.reciprocal_laplacian_factor(dom, 1e-3)

    def compute(s):
        u, v = s["u"], s["v"]
        s["u_bar"], s["v_bar"] = dom.fft(u), dom.fft(v)
        s["su_bar"] = dom.fft((1.0 - (3.5 + 1.0) * u) + torch.pow(u, 2.0) * v)
        s["sv_bar"] = dom.fft(3.5 * u - torch.pow(u, 2.0) * v)

    return dom, state, compute, [("u", "u_bar", Du, "su_bar"), ("v", "v_bar", Dv, "sv_bar")]


@pytest.mark.parametrize("n", [24, 64])
def test_broyden_matches_oracle(n):
    from marlin_amd.api import Context
    dom, state, compute, variables = _problem(n)
    ref = mo.BroydenSolver(dom, state, compute, variables, substeps=3, max_iterations=30, relative_tolerance=1e-6,
                           absolute_tolerance=1e-10)
    ctx = Context(2, [n, n], [2.0 * math.pi] * 2)
    # the same control flow on the GPU: the compute group of the oracle problem evaluated with the HIP transforms
    g = {"u": state["u"].cuda(), "v": state["v"].cuda()}
    Ls = [variables[0][2].cuda().contiguous(), variables[1][2].cuda().contiguous()]

    def gcompute():
        u, v = g["u"], g["v"]
        g["u_bar"], g["v_bar"] = ctx.fft(u), ctx.fft(v)
        g["su_bar"] = ctx.fft((1.0 - (3.5 + 1.0) * u) + torch.pow(u, 2.0) * v)
        g["sv_bar"] = ctx.fft(3.5 * u - torch.pow(u, 2.0) * v)

    nspec = int(torch.tensor(dom.rshape).prod())
    M = ctx.broyden_init(2, 1.0, nspec)
    sub_dt = 0.05
    trace_ref, trace_gpu, errs = [], [], []
    for _ in range(3):
        ref.substep(sub_dt)
        trace_ref.append((ref.iterations, ref.converged))
        # BroydenSolver::substep on the device
        gcompute()
        u_old = [g["u_bar"], g["v_bar"]]
        R, R0 = ctx.broyden_residual(u_old, [g["su_bar"], g["sv_bar"]], Ls, None, sub_dt)
        Rnorm, its, conv = R0, 0, False
        while its < 30:
            if Rnorm < 1e-10 or Rnorm / R0 < 1e-6:
                conv = True
                break
            S, out = ctx.broyden_predict(M, R, [g["u_bar"], g["v_bar"]], 0.5)
            g["u"], g["v"] = ctx.ifft(out[0]), ctx.ifft(out[1])
            gcompute()
            Rnorm = ctx.broyden_update(M, R, S, [g["u_bar"], g["v_bar"]], [g["su_bar"], g["sv_bar"]], Ls, u_old, sub_dt)
            its += 1
        trace_gpu.append((its, conv))
        # the rank-one updates divide by s^T y down to 1e-12: rounding differences between ATen's and our complex arithmetic
        # are amplified along the iteration, most in the substep that does not converge
        errs.append(max((g["u"].cpu() - state["u"]).abs().max().item(), (g["v"].cpu() - state["v"]).abs().max().item()))
        if len(errs) == 1:   # after the first (converging) substep the inverse-Jacobian approximations still agree
            Mref = ref.M.reshape(-1, 2, 2).permute(1, 2, 0).reshape(4, -1)       # field-major [n*n][n_spec]
            assert (M.cpu() - Mref).abs().max().item() <= 1e-6 * max(1.0, Mref.abs().max().item())
    assert trace_gpu == trace_ref                       # same iteration counts and convergence flags in every substep
    assert errs[0] <= 1e-11 and max(errs) <= 1e-6
    assert any(conv and it > 0 for it, conv in trace_ref)


@pytest.mark.parametrize("nv", [3, 8, 9, 20])
def test_broyden_kernels_for_any_variable_count(nv):
    """one predict + update of nv coupled variables on random data against BroydenSolver.C:124-165 written with torch.matmul
    (the restatement's expressions, oracle/marlin_oracle.py BroydenSolver.substep); nv > 8 takes the kernels' 32-wide instance"""
    from marlin_amd.api import Context
    n = 37
    ctx = Context(1, [2 * (n - 1)], [2.0 * math.pi])
    g = torch.Generator().manual_seed(nv)
    rc = lambda *sh: torch.randn(*sh, dtype=torch.complex128, generator=g)
    u, Nn, u_old, R = rc(n, nv), rc(n, nv), rc(n, nv), rc(n, nv)
    L = -torch.rand(n, nv, dtype=torch.float64, generator=g)
    M = torch.eye(nv, dtype=torch.complex128).expand(n, nv, nv) + 0.1 * rc(n, nv, nv)
    sub_dt = 0.03
    # reference expressions
    sk = -torch.matmul(M, R.unsqueeze(-1))
    skT = sk.squeeze(-1).unsqueeze(-2)
    u_half = u + sk.squeeze(-1) * 0.5
    u2 = u_half * (1.0 + 0.01)            # stands for ifft -> compute group -> fft: any new iterate will do for the update
    Rnew = (Nn + L * u2) * sub_dt + u_old - u2
    yk = (Rnew - R).unsqueeze(-1)
    denom = torch.matmul(skT, yk)
    M_new = M + torch.where(torch.abs(denom) > 1e-12, torch.matmul((sk - torch.matmul(M, yk)), skT) / denom, 0.0)
    # device
    fm = lambda t: [t[:, i].contiguous().cuda() for i in range(nv)]
    Md = M.permute(1, 2, 0).reshape(nv * nv, n).contiguous().cuda()
    Rd = R.t().contiguous().cuda()
    S, out = ctx.broyden_predict(Md, Rd, fm(u), 0.5)
    assert (torch.stack([o.cpu() for o in out], -1) - u_half).abs().max().item() <= 1e-12
    assert (S.cpu().t() - sk.squeeze(-1)).abs().max().item() <= 1e-12
    rn = ctx.broyden_update(Md, Rd, S, fm(u2), fm(Nn), [L[:, i].contiguous().cuda() for i in range(nv)], fm(u_old), sub_dt)
    assert abs(rn - torch.norm(Rnew).item()) <= 1e-11 * rn
    assert (Rd.cpu().t() - Rnew).abs().max().item() <= 1e-12
    got = Md.cpu().reshape(nv, nv, n).permute(2, 0, 1)
    assert (got - M_new).abs().max().item() <= 1e-9 * M_new.abs().max().item()
