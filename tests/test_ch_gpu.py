"""Cahn-Hilliard substep on the GPU vs the reference gold file and the oracle, through the C ABI."""
import os

import numpy as np
import pytest
import torch

from oracle import marlin_oracle as mo
from tests.conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu


def _run_hip_ch(ctx, p, c0, nsteps, substeps, dt, pred=1, want_mu=False, carry=False):
    """TensorSolver::computeBuffer loop with the history rules of TensorProblem::advanceState (host logic only).
    carry: spectral carry-over (first substep MRL_CARRY_OUT, then MRL_CARRY_IN on the same array)"""
    c = c0.cuda()
    carried = ctx.empty_hist() if carry else None
    nsub_done = 0
    hist = []
    Nhat = None
    states = []
    mu = torch.empty_like(c) if want_mu else None
    time_step = 0

    def advance():
        nonlocal hist
        if time_step <= 1 or Nhat is None:
            return
        if len(hist) < pred:
            hist.append(None)
        if hist:
            for i in range(len(hist) - 1, 0, -1):
                hist[i] = hist[i - 1]
            hist[0] = Nhat

    for step in range(nsteps):
        time_step += 1
        advance()
        sub_dt = dt / substeps
        for s in range(substeps):
            order = min(len(hist), pred)
            Nnew = ctx.empty_hist()
            cn = torch.empty_like(c)
            ctx.ch_substep(p, c, cn, Nnew, hist[:order], order, sub_dt, mu=mu, cbar=carried,
                           carry=0 if not carry else (1 if nsub_done == 0 else 2))
            nsub_done += 1
            c, Nhat = cn, Nnew
            if s < substeps - 1:
                advance()
        states.append(c.cpu())
    return states, (mu.cpu() if want_mu else None)


@pytest.mark.parametrize("carry", [False, True])
def test_ch_gold_file(carry):
    """test/tests/cahnhilliard/tests:46-57: c.1..c.10 and mu.10 to abs_tol 1e-13 (also with the opt-in spectral carry-over)"""
    from marlin_amd.api import Context, ch_params
    g = load_golden("cahnhilliard_gold.npz")
    ctx = Context(2, [20, 20], [3.0, 3.0])
    c0 = torch.from_numpy(g["c.0"][:20, :20].copy())
    states, mu = _run_hip_ch(ctx, ch_params(), c0, 10, 10, 1e-3, want_mu=True, carry=carry)
    worst = max(np.abs(g[f"c.{k + 1}"][:20, :20] - states[k].numpy()).max() for k in range(10))
    assert worst <= 1e-13, worst
    assert np.abs(g["mu.10"] - mu.numpy()).max() <= 1e-13


def test_ch_gold_file_3d():
    """test/tests/cahnhilliard/tests:13-22 (Domain/dim=3 nx=ny=nz=5): nodal c / elemental mu of map_to_aux_3d.e, abs 1e-13"""
    from marlin_amd.api import Context, ch_params
    g = load_golden("cahnhilliard_3d_gold.npz")
    ctx = Context(3, [5, 5, 5], [3.0, 3.0, 3.0])
    states, mu = _run_hip_ch(ctx, ch_params(), torch.from_numpy(g["c.0"].copy()), 10, 10, 1e-3, want_mu=True)
    worst = max(np.abs(g[f"c.{k + 1}"] - states[k].numpy()).max() for k in range(10))
    assert worst <= 1e-13, worst
    assert np.abs(g["mu.10"] - mu.numpy()).max() <= 1e-13


# one shape per plan family and length (every planned length of fft_pow2.h appears once, with small co-dimensions: the oracle's CPU
# transforms are what this sweep costs), the generic any-length path, 1-D / 2-D
@pytest.mark.parametrize("shape", [(16, 16, 16), (12, 10, 9), (32, 32, 32), (24,), (64, 64, 64), (100, 40, 50),
                                   (200, 32, 40), (128, 128), (200, 100), (64, 400), (96, 192, 48), (384, 96),
                                   (40, 80, 32), (48, 144, 50), (250, 32), (500, 32), (1000, 48), (768, 40, 32),
                                   (2048, 64), (32, 4096), (2048, 32, 32), (256, 32, 32), (32, 64, 512), (1024, 32),
                                   (150, 150), (120, 90), (240, 40, 32), (32, 270, 40), (300, 180), (360, 60), (450, 600),   # planned-unfused path
                                   (160, 64), (64, 320, 32), (640, 160), (32, 40, 1280),                               # ... radix-20 lengths
                                   (240, 120, 32), (150, 180, 32), (160, 160), (120, 240), (180, 150, 40),         # ... two-stage plans on x and y (fft_two.h)
                                   (32, 40, 150), (40, 32, 180), (32, 32, 120), (48, 240, 160),                       # ... and on z (fft_two_z.h)
                                   (400, 40, 32), (32, 400, 40), (40, 32, 400), (300, 320), (320, 32, 300), (32, 300, 320),   # ... 20 points per thread
                                   (192, 32, 192), (192, 192),                                                         # ... fused family, two-stage x / z kernels
                                   (288, 64), (72, 216), (576, 64), (800, 32), (48, 432), (864, 1152)])            # ... further plain plans
def test_ch_vs_oracle(shape):
    from marlin_amd.api import Context, ch_params
    dim = len(shape)
    L = [2.0 + d for d in range(dim)]
    ctx = Context(dim, list(shape), L)
    dom = mo.Domain(dim, list(shape), L)
    torch.manual_seed(11)
    c0 = torch.rand(shape, dtype=torch.float64) * 0.12 + 0.44
    # two time steps of three substeps: AB1, then AB2 with the history rules of the first and of a later time step
    ref = mo.CahnHilliardABM(dom, c0, 0.2, -0.001, mo.mu_double_well, substeps=3)
    for _ in range(2):
        ref.step(3e-3)
    states, _ = _run_hip_ch(ctx, ch_params(), c0, 2, 3, 3e-3)
    assert (states[-1] - ref.c).abs().max().item() <= 1e-13
    # opt-in spectral carry-over (c-hat = ubar of the previous substep): same fields to rounding
    carried, _ = _run_hip_ch(ctx, ch_params(), c0, 2, 3, 3e-3, carry=True)
    assert (carried[-1] - ref.c).abs().max().item() <= 1e-13


@pytest.mark.parametrize("shape", [(240, 60, 32), (120, 150), (160, 180, 40), (150, 32, 32), (180, 120), (400, 32, 40), (300, 320), (320, 40, 32)])
@pytest.mark.parametrize("pred", [3, 5])
def test_two_stage_plans_deep_histories_and_outputs(shape, pred):
    """fft_two.h: the fused x pass of the two-stage plans (120 / 150 / 160 / 180 / 240 points) with AB3 and AB5 histories, the
    optional cbar / mu outputs, against the oracle (1e-13) and against the uniform 30- / 20-point plans it replaces (experiment bit
    1 << 29: the same operations on the pointwise side, so the fields agree to the rounding of the transforms); the profile must show
    the fused x kernel, and the two x kernels of the uniform plans with the experiment bit"""
    from marlin_amd import api
    from marlin_amd.api import Context, ch_params
    dim = len(shape)
    L = [2.0 + d for d in range(dim)]
    dom = mo.Domain(dim, list(shape), L)
    torch.manual_seed(7)
    c0 = torch.rand(shape, dtype=torch.float64) * 0.12 + 0.44
    ref = mo.CahnHilliardABM(dom, c0, 0.2, -0.001, mo.mu_double_well, substeps=3, predictor_order=pred)
    for _ in range(pred + 1):
        ref.step(3e-3)
    got = {}
    for name, exp in (("two_stage", 0), ("uniform", 1 << 29)):
        ctx = Context(dim, list(shape), L)
        ctx.set_option(api.OPT_EXPERIMENT, exp)
        ctx.set_profiling(True)
        states, mu = _run_hip_ch(ctx, ch_params(), c0, pred + 1, 3, 3e-3, pred=pred - 1, want_mu=True)
        slots = {k["kernel"] for k in ctx.get_profile() if k["launches"]}
        if shape[0] == 400:     # a length of the fused family: same profile slot, k_ch_xfused2<400> or k_ch_xfused<400> behind it
            assert "ch_C_x_fused" in slots, slots
        elif name == "two_stage":
            assert "chp_CD_x_fused" in slots and "chp_C_x_mbar" not in slots, slots
        else:
            assert ("chp_C_x_mbar" in slots or "ch_C_x_fused" in slots) and "chp_CD_x_fused" not in slots, slots   # (400: the fused family)
        assert (states[-1] - ref.c).abs().max().item() <= 1e-13
        got[name] = states[-1]
    assert (got["two_stage"] - got["uniform"]).abs().max().item() <= 1e-14


@pytest.mark.parametrize("shape", [(240, 32, 120), (300, 40, 32), (160, 150), (400, 32, 32), (192, 32, 192)])
def test_two_stage_plans_cbar_and_history_outputs(shape):
    """the optional outputs of a substep on the two-stage kernels (k_ch_xfused2: cbar = c-hat of the substep, Nhat = the history
    entry; k_z_fwd2: mu) against the oracle's op sequence, AB1 and AB2"""
    from marlin_amd.api import Context, ch_params
    dim = len(shape)
    L = [4.0, 5.0, 6.0][:dim]
    ctx = Context(dim, list(shape), L)
    dom = mo.Domain(dim, list(shape), L)
    torch.manual_seed(3)
    c0 = torch.rand(shape, dtype=torch.float64) * 0.12 + 0.44
    Mbar = mo.reciprocal_laplacian_factor(dom, 0.2)
    Lbar = mo.reciprocal_laplacian_square_factor(dom, -0.001)
    r1, N1, cb1, mu1 = mo.ch_substep_ops(c0, Mbar, Lbar, [], 2e-3, 0, mo.mu_double_well, dom)
    r2, N2, cb2, mu2 = mo.ch_substep_ops(r1, Mbar, Lbar, [N1], 2e-3, 1, mo.mu_double_well, dom)
    p = ch_params()
    c = c0.cuda()
    Na, Nb = ctx.empty_hist(), ctx.empty_hist()
    cbar, mu = ctx.empty_hist(), torch.empty_like(c)
    c1, c2 = torch.empty_like(c), torch.empty_like(c)
    ctx.ch_substep(p, c, c1, Na, [], 0, 2e-3, cbar=cbar, mu=mu)
    assert (cbar.cpu() - cb1).abs().max().item() <= 1e-13 * cb1.abs().max().item()
    assert (mu.cpu() - mu1).abs().max().item() <= 1e-15
    # Nhat = Mbar mu-hat: the rounding of the transform (~ eps log2(n) max|mu-hat|, white over the spectrum) times the largest |Mbar|
    tolN = lambda m: 4e-16 * np.log2(c0.numel()) * torch.fft.rfftn(m).abs().max().item() * Mbar.abs().max().item()
    assert (Na.cpu() - N1).abs().max().item() <= tolN(mu1)
    assert (c1.cpu() - r1).abs().max().item() <= 1e-14
    # (the second substep starts from the oracle's field: Nhat amplifies the 1e-15 differences of its input by |Mbar| ~ 1e4 at high k)
    ctx.ch_substep(p, r1.cuda(), c2, Nb, [Na], 1, 2e-3, cbar=cbar, mu=mu)
    assert (cbar.cpu() - cb2).abs().max().item() <= 1e-13 * cb2.abs().max().item()
    assert (Nb.cpu() - N2).abs().max().item() <= tolN(mu2)
    assert (mu.cpu() - mu2).abs().max().item() <= 1e-15
    assert (c2.cpu() - r2).abs().max().item() <= 1e-14


def test_random_mixes_of_planned_lengths_against_the_any_length_path():
    """48 random grids (2-D and 3-D) whose extents are drawn from the planned lengths -- uniform, radix-30 / radix-20, two-stage plans
    next to each other on every axis role -- : three substeps (AB1, AB2, AB3) through the planned / fused kernels and through the
    any-length path (experiment bit 2048: generic transforms + separate pointwise kernels, an independent implementation of the same
    operator sequence), fields and mu to 1e-13"""
    import random
    import re
    from marlin_amd import api
    from marlin_amd.api import Context, ch_params
    src = open(os.path.join(ROOT, "marlin_amd", "csrc", "fft_pow2.h")).read()
    lengths = sorted({int(m) for m in re.findall(r"^MRL_PLAN\((\d+),", src, flags=re.M)})
    lengths = [n for n in lengths if n <= 640]
    small = [n for n in lengths if n <= 80]
    rng = random.Random(5)
    bad = []
    for it in range(48):
        dim = 2 if it % 4 == 3 else 3
        big_axis = rng.randrange(dim)
        shape = [rng.choice(lengths) if a == big_axis else rng.choice(small) for a in range(dim)]
        if dim == 3 and (shape[0] * shape[1]) % 2:
            continue
        L = [2.0 + d for d in range(dim)]
        torch.manual_seed(it)
        c0 = (torch.rand(shape, dtype=torch.float64) * 0.12 + 0.44).cuda()
        res = []
        for exp in (0, 2048):
            ctx = Context(dim, shape, L, dense_spectra=exp != 0)
            ctx.set_option(api.OPT_EXPERIMENT, exp)
            ctx.set_profiling(True)
            c, hist = c0.clone(), []
            mu = torch.empty_like(c0)
            for k in range(3):
                Nn, cn = ctx.empty_hist(), torch.empty_like(c)
                ctx.ch_substep(ch_params(), c, cn, Nn, hist[:k], k, 2e-3, mu=mu)
                hist.insert(0, Nn)
                c = cn
            res.append((c.cpu(), mu.cpu()))
            slots = {k["kernel"] for k in ctx.get_profile() if k["launches"]}
            assert ("ch_kspace" in slots) == (exp != 0), (shape, exp, slots)     # the any-length path, and only it, has this kernel
        e = max((a - b).abs().max().item() for a, b in zip(*res))
        if e > 1e-13:
            bad.append((shape, e))
    assert not bad, bad


def test_ch_pfhub_family_and_ab3():
    from marlin_amd.api import Context, ch_params, FE_PFHUB
    shape = (32, 32)
    ctx = Context(2, list(shape), [200.0, 200.0])
    dom = mo.Domain(2, list(shape), [200.0, 200.0])
    x, y = dom.axis[0], dom.axis[1]
    c0 = (0.5 + 0.01 * (torch.cos(0.105 * x) * torch.cos(0.11 * y))).expand(shape).contiguous()
    mu_fn = lambda c: mo.mu_pfhub(c, 5.0, 0.3, 0.7)
    ref = mo.CahnHilliardABM(dom, c0, 5.0, -10.0, mu_fn, substeps=4, predictor_order=3)
    for _ in range(3):
        ref.step(1.0)
    p = ch_params(FE_PFHUB, (5.0, 0.3, 0.7), mobility=5.0, kappa=-10.0)
    states, _ = _run_hip_ch(ctx, p, c0, 3, 4, 1.0, pred=2)
    assert (states[-1] - ref.c).abs().max().item() <= 1e-13
    carried, _ = _run_hip_ch(ctx, p, c0, 3, 4, 1.0, pred=2, carry=True)
    assert (carried[-1] - ref.c).abs().max().item() <= 1e-13


@pytest.mark.parametrize("shape", [(64, 64, 64), (64, 128, 64), (128, 64, 64)])
def test_ch_fused_fast_path_outputs(shape):
    """fused fast path: c, Nhat (history buffer), optional cbar and mu outputs vs the oracle's op sequence"""
    from marlin_amd.api import Context, ch_params
    L = [4.0, 5.0, 6.0]
    ctx = Context(3, list(shape), L)
    dom = mo.Domain(3, list(shape), L)
    torch.manual_seed(3)
    c0 = torch.rand(shape, dtype=torch.float64) * 0.12 + 0.44
    Mbar = mo.reciprocal_laplacian_factor(dom, 0.2)
    Lbar = mo.reciprocal_laplacian_square_factor(dom, -0.001)
    r1, N1, cb1, mu1 = mo.ch_substep_ops(c0, Mbar, Lbar, [], 2e-3, 0, mo.mu_double_well, dom)
    r2, N2, cb2, mu2 = mo.ch_substep_ops(r1, Mbar, Lbar, [N1], 2e-3, 1, mo.mu_double_well, dom)
    r3, N3, _, _ = mo.ch_substep_ops(r2, Mbar, Lbar, [N2, N1], 2e-3, 2, mo.mu_double_well, dom)
    p = ch_params()
    c = c0.cuda()
    Na, Nb, Nc = ctx.empty_hist(), ctx.empty_hist(), ctx.empty_hist()
    cbar, mu = ctx.empty_hist(), torch.empty_like(c)
    c1, c2, c3 = torch.empty_like(c), torch.empty_like(c), torch.empty_like(c)
    ctx.ch_substep(p, c, c1, Na, [], 0, 2e-3, cbar=cbar, mu=mu)
    sc = cb1.abs().max().item()
    assert (cbar.cpu() - cb1).abs().max().item() <= 1e-13 * sc
    assert (mu.cpu() - mu1).abs().max().item() <= 1e-15
    assert (Na.cpu() - N1).abs().max().item() <= 1e-13 * max(1.0, N1.abs().max().item())
    assert (c1.cpu() - r1).abs().max().item() <= 1e-14
    ctx.ch_substep(p, c1, c2, Nb, [Na], 1, 2e-3)
    assert (c2.cpu() - r2).abs().max().item() <= 1e-14
    ctx.ch_substep(p, c2, c3, Nc, [Nb, Na], 2, 2e-3)
    assert (c3.cpu() - r3).abs().max().item() <= 1e-14
    # in-place (c_out aliases c_in) gives the same result
    c2b = c1.clone()
    ctx.ch_substep(p, c2b, c2b, Nc, [Na], 1, 2e-3)
    assert torch.equal(c2b, c2)


@pytest.mark.parametrize("shape,pred", [((64, 64, 64), 2), ((128, 64, 96), 3), ((100, 40, 50), 2), ((12, 10, 9), 2), ((64, 128), 2),
                                        ((160, 48, 240), 3), ((120, 150), 2), ((64, 180, 120), 2), ((240, 32, 160), 2), ((60, 40, 180), 4),
                                        ((400, 32, 400), 3), ((320, 40, 300), 2), ((32, 300, 320), 5),   # two-stage z plans
                                        ((240, 32, 128), 2), ((120, 150, 64), 3), ((60, 90, 100), 2),    # planned x / y, fused-family z
                                        ((192, 48, 192), 3)])                                              # fused family with two-stage x / z kernels
def test_ch_multi_substep_call(shape, pred):
    """mrl_ch_substeps (the substep loop of TensorSolver::computeBuffer in one call; on planned shapes the inverse z pass of a
    substep is fused with the forward z pass of the next one) == the same substeps one call at a time, bit for bit on the fused
    path; history ring, advance rule of the first time step, mu output"""
    from marlin_amd.api import Context, ch_params
    dim = len(shape)
    L = [2.0 + d for d in range(dim)]
    ctx = Context(dim, list(shape), L)
    p = ch_params()
    torch.manual_seed(5)
    c0 = (torch.rand(shape, dtype=torch.float64) * 0.12 + 0.44).cuda()
    sub_dt, substeps = 2e-3, 7
    # reference: one mrl_ch_substep per substep with the host-side history logic (two time steps: the first never advances)
    states, mu_ref = _run_hip_ch(ctx, p, c0.cpu(), 2, substeps, sub_dt * substeps, pred=pred - 1, want_mu=True)
    ring = [ctx.empty_hist() for _ in range(pred)]
    head, n_old = 0, 0
    c, mu = c0.clone(), torch.empty_like(c0)
    for step in range(2):
        advance = step > 0
        if advance:           # TensorProblem::advanceState at the start of the time step
            head, n_old = (head + 1) % pred, min(n_old + 1, pred - 1)
        out = torch.empty_like(c)
        head, n_old = ctx.ch_substeps(p, c, out, ring, head, n_old, pred, substeps, advance, sub_dt, mu=mu)
        c = out
        diff = (c.cpu() - states[step]).abs().max().item()
        assert diff <= 1e-15, diff
    assert (mu.cpu() - mu_ref).abs().max().item() <= 1e-16


@pytest.mark.parametrize("shape", [(64, 64, 64), (12, 10, 9), (64, 128)])
def test_ch_adaptive_dt_restarts_the_order(shape):
    """a changed time step size restarts the Adams-Bashforth order for the first predictor_order - 1 substeps of the step while the
    history keeps advancing (AdamsBashforthMoulton.C:75,88-91; MRL_SUBSTEPS_DT_CHANGED): AB3, 4 substeps per step, dt 1e-3, 1e-3,
    2e-3, 2e-3, 5e-4 -- fused and generic paths against the oracle"""
    from marlin_amd.api import Context, ch_params
    dim = len(shape)
    L = [2.0 + d for d in range(dim)]
    ctx = Context(dim, list(shape), L)
    p = ch_params()
    torch.manual_seed(9)
    c0 = torch.rand(shape, dtype=torch.float64) * 0.12 + 0.44
    dts, substeps, pred = [1e-3, 1e-3, 2e-3, 2e-3, 5e-4], 4, 3
    dom = mo.Domain(dim, list(shape), L)
    ref = mo.CahnHilliardABM(dom, c0, M=0.2, kappa_factor=-0.001, mu_fn=mo.mu_double_well, substeps=substeps, predictor_order=pred)
    ring = [ctx.empty_hist() for _ in range(pred)]
    head, n_old, dt_old = 0, 0, None
    c = c0.cuda()
    for step, dt in enumerate(dts):
        ref.step(dt)
        if step > 0:           # TensorProblem::advanceState at the start of the time step
            head, n_old = (head + 1) % pred, min(n_old + 1, pred - 1)
        out = torch.empty_like(c)
        head, n_old = ctx.ch_substeps(p, c, out, ring, head, n_old, pred, substeps, step > 0, dt / substeps,
                                      dt_changed=dt_old is not None and dt != dt_old)
        dt_old = dt
        c = out
        assert (c.cpu() - ref.c).abs().max().item() <= 1e-13
    assert ref.order_log[8:12] == [0, 0, 2, 2] and ref.order_log[16:] == [0, 0, 2, 2]


def test_config_a_pfhub_1a_128():
    """BASELINE configs[0] as SURVEY 8(d) specifies it: PFHub benchmark 1a (benchmarks/01_spinodal_decomposition/1a_solver.i:45-86)
    on 128^2, L = 200 x 200: f = rho (c - c_alpha)^2 (c_beta - c)^2 with rho 5, c_alpha 0.3, c_beta 0.7, Mbar = -5 k^2,
    Lbar = -10 k^4, the benchmark's three-mode initial condition, AB2, 10 time steps of dt = 1 with 1000 substeps each
    (spectral_solve_substeps = 1000): the HIP path (one library call per time step) against the oracle after each of the first
    two steps (2 000 substeps; the oracle's 10 000 CPU substeps were 70 s of the GPU test tier), and the
    total free energy F = int f + |grad c|^2 (the input's [Postprocess] block, FFTGradientSquare factor 1) falls monotonically.
    Tolerance: the reference's 1e-13 is quoted for 100 substeps (test/tests/cahnhilliard/tests:46-57); the butterflies of this FFT
    and MKL's round differently (1e-16 per transform) and the difference grows linearly with the substep count -- measured 1.8e-14
    after 1 000 and 7.9e-13 after 10 000 substeps (2.9e-15 ... 9.2e-14 with 100 substeps per step) -- so the bound is 1e-13 for
    the first 3 000 substeps and 1e-16 per substep afterwards."""
    from marlin_amd.api import Context, ch_params, FE_PFHUB
    n, Ld = 128, 200.0
    shape = [n, n]
    ctx = Context(2, shape, [Ld, Ld])
    dom = mo.Domain(2, shape, [Ld, Ld])
    x, y = dom.axis[0], dom.axis[1]
    c0 = (0.5 + 0.01 * (torch.cos(0.105 * x) * torch.cos(0.11 * y) + torch.pow(torch.cos(0.13 * x) * torch.cos(0.087 * y), 2)
                        + torch.cos(0.025 * x - 0.15 * y) * torch.cos(0.07 * x - 0.02 * y))).expand(shape).contiguous()
    substeps, pred = 1000, 2
    ref = mo.CahnHilliardABM(dom, c0, 5.0, -10.0, lambda c: mo.mu_pfhub(c, 5.0, 0.3, 0.7), substeps=substeps, predictor_order=pred)
    p = ch_params(FE_PFHUB, (5.0, 0.3, 0.7), mobility=5.0, kappa=-10.0)
    ring = [ctx.empty_hist() for _ in range(pred)]
    head, n_old = 0, 0
    c = c0.cuda()

    def free_energy(field):
        ch = dom.fft(field)
        g2 = sum(dom.ifft(1j * dom.kaxis[d] * ch) ** 2 for d in range(2))
        f = 5.0 * (field - 0.3) ** 2 * (0.7 - field) ** 2 + g2
        return float(f.sum() * (Ld / n) ** 2)

    energies = [free_energy(c0)]
    for step in range(10):
        if step < 2:
            ref.step(1.0)
        if step > 0:
            head, n_old = (head + 1) % pred, min(n_old + 1, pred - 1)
        out = torch.empty_like(c)
        head, n_old = ctx.ch_substeps(p, c, out, ring, head, n_old, pred, substeps, step > 0, 1.0 / substeps)
        c = out
        if step < 2:
            assert (c.cpu() - ref.c).abs().max().item() <= max(1e-13, 1e-16 * substeps * (step + 1))
        energies.append(free_energy(c.cpu()))
    assert all(b < a for a, b in zip(energies, energies[1:])), energies


@pytest.mark.parametrize("family", ["double_well", "pfhub", "parsed"])
@pytest.mark.parametrize("shape,mb", [((64, 64, 64), 1), ((100, 64, 128), 2), ((128, 128, 128), 7)])
def test_cache_chunked_schedule_is_bit_identical(shape, mb, family):
    """MRL_OPT_CACHE_CHUNK_MB (A/B switch, off by default): the plane-wise passes between two x passes run chunk after chunk over x --
    the same kernels on the same data in another order -- so a multi-substep call ends on the same bits, field and chemical
    potential, incl. a last chunk of another size (100 planes in chunks of 30)"""
    from marlin_amd import api
    ctx = api.Context(3, list(shape), [3.0, 2.5, 4.0])
    if family == "double_well":
        p = api.ch_params()
    elif family == "pfhub":
        p = api.ch_params(family=api.FE_PFHUB, coef=(5.0, 0.3, 0.7), mobility=5.0, kappa=-10.0)
    else:   # the user's expression compiled into the z passes (hiprtc): its launcher takes the same chunk offsets
        p = api.ch_params(parsed=api.ParsedCompute(ctx, "0.1*c^2*(c-1)^2+0.01*c^4", inputs=["c"], derivatives=["c"]))
    g = torch.Generator(device="cuda").manual_seed(3)
    c0 = torch.rand(*shape, dtype=torch.float64, device="cuda", generator=g) * 0.12 + 0.44
    res = []
    for budget in (0, mb):
        ctx.set_option(api.OPT_CACHE_CHUNK_MB, budget)
        ring = [ctx.empty_hist(), ctx.empty_hist(), ctx.empty_hist()]
        for r in ring:
            r.zero_()
        out, mu = torch.empty_like(c0), torch.empty_like(c0)
        ctx.ch_substeps(p, c0, out, ring, 2, 0, 3, 7, True, 1e-3, mu=mu)
        ctx.sync()
        res.append((out, mu))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert (res[0][0] - c0).abs().max().item() > 1e-6
    with pytest.raises(api.MarlinHipError):
        ctx.set_option(api.OPT_CACHE_CHUNK_MB, -3)
