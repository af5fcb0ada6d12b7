"""AdamsBashforthMoultonCoupled's per-k dense solve (mrl_kspace_coupled) against the oracle's at::linalg_solve restatement."""
import math

import pytest
import torch

import oracle.marlin_oracle as mo

pytestmark = pytest.mark.gpu


def _problem(nv, shape, seed):
    from marlin_amd.api import Context
    dim = len(shape)
    dom = mo.Domain(dim, list(shape), [2.0 * math.pi] * dim)
    ctx = Context(dim, list(shape), [2.0 * math.pi] * dim)
    g = torch.Generator().manual_seed(seed)
    rs = dom.rshape
    u0 = [torch.randn(rs, dtype=torch.complex128, generator=g) for _ in range(nv)]
    N = [[torch.randn(rs, dtype=torch.complex128, generator=g) for _ in range(1 + (i % 3))] for i in range(nv)]
    coef = [[0.05 * (t + 1) * (-1) ** t for t in range(len(N[i]))] for i in range(nv)]
    # diagonally dominated but strongly coupled operator, with holes (None = zero entry)
    L = [[(-(1.0 + i + j) * 0.3 * torch.rand(rs, dtype=torch.float64, generator=g) - (4.0 if i == j else 0.0))
          if (i == j or (i + 2 * j) % 3 != 0) else None for j in range(nv)] for i in range(nv)]
    return dom, ctx, u0, N, coef, L


def _oracle(dom, u0, N, coef, L, dt, real_rhs, transposed):
    nv = len(u0)
    rhs = []
    for i in range(nv):
        r = u0[i].clone()
        for c, t in zip(coef[i], N[i]):
            r += c * t
        rhs.append(r)
    s = mo.CoupledABM(dom, {}, lambda st: None, [("u%d" % i, "", None, "n%d" % i) for i in range(nv)], L, 1,
                      real_rhs=real_rhs, transposed=transposed)
    return s.solve(rhs, dt)


@pytest.mark.parametrize("nv,shape", [(1, (24,)), (2, (20, 18)), (3, (12, 10, 9)), (4, (16, 15)), (6, (10, 9)), (8, (12, 7))])
@pytest.mark.parametrize("flags", [0, 1, 2, 3])
def test_kspace_coupled_matches_linalg_solve(nv, shape, flags):
    dom, ctx, u0, N, coef, L = _problem(nv, shape, 11 * nv + flags)
    dt = 0.37
    want = _oracle(dom, u0, N, coef, L, dt, real_rhs=not (flags & 2), transposed=not (flags & 1))
    dev = lambda t: None if t is None else t.cuda().contiguous()
    out = [torch.empty(dom.rshape, dtype=torch.complex128, device="cuda") for _ in range(nv)]
    ctx.kspace_coupled(out, [dev(t) for t in u0], [[dev(t) for t in row] for row in N], coef,
                       [[dev(t) for t in row] for row in L], dt, flags)
    for i in range(nv):
        w = want[i] if want[i].is_complex() else want[i].to(torch.complex128)
        scale = max(1.0, w.abs().max().item())
        assert (out[i].cpu() - w).abs().max().item() <= 1e-13 * scale
        if not (flags & 2):
            assert out[i].imag.abs().max().item() == 0.0


def test_kspace_coupled_pivots():
    """an operator whose diagonal vanishes: partial pivoting has to swap rows (A = [[0, 1], [1, 0]] - like)"""
    from marlin_amd.api import Context
    dom = mo.Domain(1, [16], [2.0 * math.pi])
    ctx = Context(1, [16], [2.0 * math.pi])
    rs = dom.rshape
    one = torch.ones(rs, dtype=torch.float64)
    L = [[one.clone(), -3.0 * one], [2.0 * one, one.clone()]]      # dt = 1: A = [[0, -2(T)...]]
    u0 = [torch.randn(rs, dtype=torch.complex128) for _ in range(2)]
    want = _oracle(dom, u0, [[], []], [[], []], L, 1.0, real_rhs=False, transposed=True)
    out = [torch.empty(rs, dtype=torch.complex128, device="cuda") for _ in range(2)]
    ctx.kspace_coupled(out, [t.cuda() for t in u0], [[], []], [[], []], [[t.cuda() for t in r] for r in L], 1.0, 2)
    for i in range(2):
        assert (out[i].cpu() - want[i]).abs().max().item() <= 1e-13


@pytest.mark.parametrize("nv,shape", [(9, (12, 10)), (12, (9, 8, 7)), (16, (40,)), (32, (6, 5))])
@pytest.mark.parametrize("flags", [0, 3])
def test_kspace_coupled_beyond_the_register_kernel(nv, shape, flags):
    """9 ... 32 variables (the reference solves any N, AdamsBashforthMoultonCoupled.C:183): the workspace form against linalg_solve"""
    dom, ctx, u0, N, coef, L = _problem(nv, shape, 5 * nv + flags)
    dt = 0.21
    want = _oracle(dom, u0, N, coef, L, dt, real_rhs=not (flags & 2), transposed=not (flags & 1))
    dev = lambda t: None if t is None else t.cuda().contiguous()
    out = [torch.empty(dom.rshape, dtype=torch.complex128, device="cuda") for _ in range(nv)]
    ctx.kspace_coupled(out, [dev(t) for t in u0], [[dev(t) for t in row] for row in N], coef,
                       [[dev(t) for t in row] for row in L], dt, flags)
    for i in range(nv):
        w = want[i] if want[i].is_complex() else want[i].to(torch.complex128)
        assert (out[i].cpu() - w).abs().max().item() <= 1e-12 * max(1.0, w.abs().max().item())


@pytest.mark.parametrize("nv,shape", [(2, (300, 290)), (5, (33, 31)), (8, (20, 19))])
def test_kspace_coupled_workspace_form_equals_the_register_kernel_bit_for_bit(nv, shape):
    """same elimination, operation for operation: flag 4 (MRL_COUPLED_GENERAL) forces the workspace form at any size; the first shape
    has more k-points than workspace lanes (65536), so lanes are reused"""
    dom, ctx, u0, N, coef, L = _problem(nv, shape, 100 + nv)
    dev = lambda t: None if t is None else t.cuda().contiguous()
    args = ([dev(t) for t in u0], [[dev(t) for t in row] for row in N], coef, [[dev(t) for t in row] for row in L], 0.4)
    a = [torch.empty(dom.rshape, dtype=torch.complex128, device="cuda") for _ in range(nv)]
    b = [torch.empty(dom.rshape, dtype=torch.complex128, device="cuda") for _ in range(nv)]
    ctx.kspace_coupled(a, *args, 2)
    ctx.kspace_coupled(b, *args, 2 | 4)
    for x, y in zip(a, b):
        assert torch.equal(torch.view_as_real(x), torch.view_as_real(y))


def test_kspace_coupled_rejects_bad_arguments():
    from marlin_amd.api import Context, MarlinHipError
    ctx = Context(1, [16], [1.0])
    t = torch.zeros(9, dtype=torch.complex128, device="cuda")
    with pytest.raises(MarlinHipError):
        ctx.kspace_coupled([t] * 33, [t] * 33, [[]] * 33, [[]] * 33, [[None] * 33] * 33, 1.0)
