"""de Geus mechanics on the GPU vs the reference gold files and the oracle, through the C ABI."""
import math

import numpy as np
import pytest
import torch

from oracle import marlin_oracle as mo
from tests.conftest import load_golden
from tests.test_oracle_golden import MECH_CASES, _mech_setup

pytestmark = pytest.mark.gpu


def _ctx(dim, shape, L):
    from marlin_amd.api import Context
    return Context(dim, list(shape), list(L))


@pytest.mark.parametrize("shape,L", [((8, 6, 10), (1.0, 2.0, 3.0)), ((16, 16, 16), (2 * math.pi,) * 3),
                                     ((9, 7), (1.0, 1.5)), ((32, 32), (2 * math.pi,) * 2), ((5, 7, 9), (1.0, 1.0, 1.0)),
                                     ((64, 64, 64), (1.0, 2.0, 3.0)), ((64, 128, 64), (2.0, 1.0, 1.5))])
def test_gamma_apply(shape, L):
    """G(A) = ifft(Ghat4 : fft(A)) with the stored operator of FFTMechanics.C:74-84 (oracle), 1e-12"""
    dim = len(shape)
    dom = mo.Domain(dim, list(shape), list(L))
    torch.manual_seed(5)
    A = torch.rand(dom.value_shape([dim, dim]), dtype=torch.float64)
    ref = dom.ifft_batched(mo.ddot42(mo.ghat4(dom), dom.fft_batched(A)))
    ctx = _ctx(dim, shape, L)
    got = ctx.gamma_apply(A.cuda()).cpu()
    assert (got - ref).abs().max().item() <= 1e-12
    # G is a projection (G(G(A)) = G(A)) when no axis has a Nyquist bin; with even extents the
    # reference's operator is not idempotent (Nyquist is not zeroed, DomainAction.C:289-291) -- match it
    twice = ctx.gamma_apply(got.cuda()).cpu()
    ref2 = dom.ifft_batched(mo.ddot42(mo.ghat4(dom), dom.fft_batched(ref)))
    assert (twice - ref2).abs().max().item() <= 1e-12
    if all(s % 2 == 1 for s in shape):
        assert (twice - got).abs().max().item() <= 1e-12


@pytest.mark.parametrize("dim,n", [(3, 8), (2, 12)])
def test_stress_and_tangent(dim, n):
    """HyperElasticIsotropic.C:42-52 (P) and K_dF of FFTMechanics.C:107-108 with the stored K4 (oracle)"""
    dom, phase, K, mu = _mech_setup(dim, n)
    ids = mo.MechIdentities(dim)
    torch.manual_seed(7)
    F = torch.eye(dim, dtype=torch.float64) + 0.1 * (torch.rand(dom.value_shape([dim, dim]), dtype=torch.float64) - 0.5)
    dF = torch.rand(dom.value_shape([dim, dim]), dtype=torch.float64) - 0.5
    P_ref, K4 = mo.hyper_elastic_isotropic(dom, ids, F, K, mu)
    KdF_ref = mo.trans2(mo.ddot42(K4, mo.trans2(dF)))
    ctx = _ctx(dim, dom.shape, [2 * math.pi] * dim)
    P = ctx.mech_stress(F.cuda(), K.cuda(), mu.cuda()).cpu()
    KdF = ctx.mech_tangent_apply(F.cuda(), K.cuda(), mu.cuda(), dF.cuda()).cpu()
    assert (P - P_ref).abs().max().item() <= 1e-13
    assert (KdF - KdF_ref).abs().max().item() <= 1e-12


def test_reductions():
    """torch::sum / torch::norm call sites of the CG (MarlinUtils.h:63-109) and DomainAction::average"""
    ctx = _ctx(3, (8, 6, 10), (1.0, 1.0, 1.0))
    torch.manual_seed(2)
    for n in (1, 2, 7, 4320, 100003):
        a = torch.rand(n, dtype=torch.float64) - 0.5
        b = torch.rand(n, dtype=torch.float64) - 0.5
        ad, bd = a.cuda(), b.cuda()
        scale = max(1.0, float(a.abs().sum()))
        assert abs(ctx.dot(ad, bd) - float(torch.sum(a * b))) <= 1e-14 * scale
        assert abs(ctx.norm2(ad) - float(torch.norm(a))) <= 1e-14 * scale
        assert abs(ctx.sum(ad) - float(torch.sum(a))) <= 1e-14 * scale
    f = torch.rand(8, 6, 10, 3, 3, dtype=torch.float64)
    avg = ctx.average(f.cuda())
    assert (avg - f.sum(dim=(0, 1, 2)) / 480.0).abs().max().item() <= 1e-14


@pytest.mark.parametrize("case", ["mech3d", "mech2d"])
def test_mech_gold(case):
    """test/tests/mechanics/tests:2-21 -- F_k.frame to abs_tol 1e-10; iteration counts as the oracle's"""
    p = MECH_CASES[case]
    dim, n = p["dim"], p["n"]
    g = load_golden(p["gold"])
    dom, phase, K, mu = _mech_setup(dim, n)
    ctx = _ctx(dim, dom.shape, [2 * math.pi] * dim)
    oracle = mo.FFTMechanicsOracle(dom, K, mu, l_tol=p["l_tol"], nl_rel_tol=p["nl_rel"], nl_abs_tol=p["nl_abs"],
                                   l_max_its=p["l_max_its"])
    Kd, mud = K.cuda(), mu.cuda()
    F = torch.eye(dim, dtype=torch.float64).expand(dom.value_shape([dim, dim])).contiguous().cuda()
    F_or = F.cpu()
    dt, substeps = p["dt"], p["substeps"]
    t_old = 0.0
    perm = (2, 1, 0) if dim == 3 else (1, 0)
    worst = 0.0
    for step in range(1, 4):
        sub_dt = dt / substeps
        for s in range(substeps):
            t = t_old + s * sub_dt
            # MacroscopicShearTensor.C:31-41 on the host from the device average
            applied = torch.eye(dim, dtype=torch.float64)
            applied[0, 1] = applied[0, 1] + t
            applied = applied - ctx.average(F)
            Fnew, P, stats = ctx.mech_newton_cg(F, Kd, mud, applied.cuda(), l_tol=p["l_tol"],
                                                l_max_its=p["l_max_its"] or 0, nl_rel_tol=p["nl_rel"],
                                                nl_abs_tol=p["nl_abs"])
            if step == 1 and s < 2:   # compare the solver trace with the oracle on the first substeps
                F_or_new, st_or = oracle.compute(F_or, mo.macroscopic_shear(dom, F_or, t))
                assert stats["newton_its"] == st_or.newton_its and stats["cg_its"] == st_or.cg_its
                assert (Fnew.cpu() - F_or_new).abs().max().item() <= 1e-10
                F_or = F_or_new
            F = Fnew
        t_old += dt
        for k in range(dim * dim):
            ref = g[f"F_{k}.{step - 1}"]
            got = F.cpu().reshape(dom.shape + [dim * dim])[..., k].permute(*perm).numpy()
            worst = max(worst, np.abs(ref - got).max())
        # [Postprocess] group of mech3d.i / mech.i: ComputeDisplacements and ComputeVonMisesStress vs the same gold files
        disp = ctx.mech_displacements(F).cpu()
        for k, nm in enumerate(("disp_x", "disp_y", "disp_z")[:dim]):
            worst = max(worst, np.abs(g[f"{nm}.{step - 1}"] - disp[..., k].permute(*perm).numpy()).max())
        if f"sV.{step - 1}" in g:
            sv = ctx.mech_von_mises(P).cpu()
            worst = max(worst, np.abs(g[f"sV.{step - 1}"] - sv.permute(*perm).numpy()).max())
    assert worst <= 1e-10, worst


@pytest.mark.parametrize("shape,L", [((12, 10, 9), (1.0, 2.0, 3.0)), ((15, 8), (2.0, 1.5)), ((64, 64, 64), (1.0, 1.0, 1.0))])
def test_displacements_and_von_mises_vs_oracle(shape, L):
    """random smooth-ish deformation gradients on anisotropic, odd and fast-path grids"""
    torch.manual_seed(3)
    dim = len(shape)
    dom = mo.Domain(dim, list(shape), list(L))
    ctx = _ctx(dim, list(shape), list(L))
    F = torch.eye(dim, dtype=torch.float64) + 0.1 * torch.randn(list(shape) + [dim, dim], dtype=torch.float64)
    want = mo.compute_displacements(dom, F)
    got = ctx.mech_displacements(F.cuda()).cpu()
    assert got.shape == want.shape
    assert (got - want).abs().max().item() <= 1e-13 * max(1.0, want.abs().max().item())
    S = torch.randn(list(shape) + [dim, dim], dtype=torch.float64)
    assert (ctx.mech_von_mises(S.cuda()).cpu() - mo.von_mises_stress(S, dim)).abs().max().item() <= 1e-14


def test_mech_fast_path_vs_oracle():
    """32^3 (power-of-two fast path: field-major vectors, Gamma fused into the x pass) against the oracle's
    FFTMechanics::computeBuffer with the stored Ghat4 / K4 (1296 B per k-point: the reason for the size; 128^3 against the oracle's
    closed form in tests/test_fullsize_gpu.py): same Newton and CG iteration counts, F to 1e-10"""
    dim, n = 3, 32
    dom, phase, K, mu = _mech_setup(dim, n)
    ctx = _ctx(dim, dom.shape, [2 * math.pi] * dim)
    oracle = mo.FFTMechanicsOracle(dom, K, mu, l_tol=1e-2, nl_rel_tol=2e-2, nl_abs_tol=2e-2)
    F = torch.eye(dim, dtype=torch.float64).expand(dom.value_shape([dim, dim])).contiguous()
    Fd, Kd, mud = F.cuda(), K.cuda(), mu.cuda()
    for s in range(2):
        t = 0.001 * (s + 1)
        applied = mo.macroscopic_shear(dom, F, t)
        F_ref, st_ref = oracle.compute(F, applied)
        app_d = torch.eye(dim, dtype=torch.float64)
        app_d[0, 1] += t
        app_d = (app_d - ctx.average(Fd)).cuda()
        Fnew, P, st = ctx.mech_newton_cg(Fd, Kd, mud, app_d, l_tol=1e-2, nl_rel_tol=2e-2, nl_abs_tol=2e-2)
        assert st["newton_its"] == st_ref.newton_its and st["cg_its"] == st_ref.cg_its
        assert (Fnew.cpu() - F_ref).abs().max().item() <= 1e-10
        assert (P.cpu() - oracle.P).abs().max().item() <= 1e-9
        F, Fd = F_ref, Fnew


@pytest.mark.parametrize("shape", [(100, 100, 100), (40, 48, 50), (80, 32, 144)])
def test_gamma_apply_radix10_sizes(shape):
    """register-radix path with the radix 10 / 12 plans against the closed form evaluated with libTorch"""
    L = (1.0, 2.0, 3.0)
    dom = mo.Domain(3, list(shape), list(L))
    torch.manual_seed(6)
    A = torch.rand(dom.value_shape([3, 3]), dtype=torch.float64)
    ref = mo.gamma_closed_form(dom, A)
    got = _ctx(3, shape, L).gamma_apply(A.cuda()).cpu()
    assert (got - ref).abs().max().item() <= 1e-12


def _newton_cg_run(shape, exp, substeps=2):
    """Newton-CG over `substeps` shear increments on a two-phase RVE of the given shape with the library's experiment mask
    `exp` (MRL_OPT_EXPERIMENT; 32 = separate tangent / z / update kernels in the CG iteration)"""
    ctx = _ctx(3, list(shape), [2 * math.pi] * 3)
    ctx.set_option(0, exp)
    nx, ny, nz = shape
    phase = torch.zeros(shape, dtype=torch.float64)
    phase[-(9 * nx // 32):, :9 * ny // 32, -(9 * nz // 32):] = 1.0
    K = ((1.0 - phase) * 0.833 + phase * 8.33).cuda()
    mu = ((1.0 - phase) * 0.386 + phase * 3.86).cuda()
    F = torch.eye(3, dtype=torch.float64).expand(*shape, 3, 3).contiguous().cuda()
    out = []
    for s in range(substeps):
        app = torch.eye(3, dtype=torch.float64)
        app[0, 1] += 0.001 * (s + 1)
        app = (app - ctx.average(F)).cuda()
        F, P, st = ctx.mech_newton_cg(F, K, mu, app, l_tol=1e-2, nl_rel_tol=2e-2, nl_abs_tol=2e-2)
        out.append((F.clone(), P.clone(), st["newton_its"], list(st["cg_its"])))
    return out


@pytest.mark.parametrize("shape", [(32, 32, 32), (40, 32, 64), (32, 48, 256), (128, 128, 128)])
def test_cg_fused_direction_tangent_z_pass(shape):
    """the CG iteration with the direction update, the tangent and the forward z pass of G in one kernel and the solution update
    deferred into it (k_gamma_z_fwd_tangent, z lines of 32 ... 256 points; 128^3 = the streaming variant of BASELINE configs[2])
    against the same solve with separate kernels (experiment option 32): same Newton / CG iteration counts, same fields"""
    fused, plain = _newton_cg_run(shape, 0), _newton_cg_run(shape, 32)
    for (Ff, Pf, nf, cf), (Fp, Pp, np_, cp) in zip(fused, plain):
        assert nf == np_ and cf == cp
        assert (Ff - Fp).abs().max().item() <= 1e-13
        assert (Pf - Pp).abs().max().item() <= 1e-12


@pytest.mark.parametrize("shape,L", [((16, 16, 16), (4 * math.pi,) * 3), ((12, 10, 9), (3.0, 2.0, 2.5)), ((32, 64, 32), (4 * math.pi,) * 3)])
def test_coupled_pf_mech_operators_vs_oracle(shape, L):
    """FFTQuasistaticElasticity + FFTElasticChemicalPotential (coupled_pf_mech.i: lambda = 100, mu = 50, e0 = 0.02) against the
    oracle's statement-by-statement restatement incl. at::linalg_solve (the reference has no gold data for these two: parity
    unpinned), and the physics they encode: the displacement field is the elastic equilibrium of the eigenstrain e0*c"""
    mu, lam, e0 = 50.0, 100.0, 0.02
    dom = mo.Domain(3, list(shape), list(L))
    ctx = _ctx(3, list(shape), list(L))
    torch.manual_seed(11)
    c = torch.rand(shape, dtype=torch.float64) * 0.12 + 0.44
    cbar = dom.fft(c)
    want = mo.quasistatic_elasticity(dom, cbar.clone(), mu, lam, e0)
    cbar_d = ctx.fft(c.cuda())
    got = ctx.qs_elasticity(cbar_d, mu, lam, e0)
    scale = max(w.abs().max().item() for w in want)
    assert scale > 0
    for g, w in zip(got, want):
        assert (g.cpu() - w).abs().max().item() <= 1e-12 * scale
    # zero mean (u-hat(0) = 0) and equilibrium: div sigma(u) = (lambda+mu) grad div u + mu lap u = -b in reciprocal space
    assert all(abs(g.mean().item()) <= 1e-14 * scale for g in got)
    want_mu = mo.elastic_chemical_potential(dom, cbar, want, mu, lam, e0)
    got_mu = ctx.elastic_chemical_potential(cbar_d, got, mu, lam, e0)
    assert (got_mu.cpu() - want_mu).abs().max().item() <= 1e-12 * want_mu.abs().max().item()
    # independent inputs for the second operator (not the first one's output)
    disp = [torch.rand(shape, dtype=torch.float64) - 0.5 for _ in range(3)]
    want_mu = mo.elastic_chemical_potential(dom, cbar, disp, mu, lam, e0)
    got_mu = ctx.elastic_chemical_potential(cbar_d, [d.cuda() for d in disp], mu, lam, e0)
    assert (got_mu.cpu() - want_mu).abs().max().item() <= 1e-12 * want_mu.abs().max().item()
    # error behaviour: 3-D only
    ctx2 = _ctx(2, [8, 8], [1.0, 1.0])
    with pytest.raises(RuntimeError, match="3-D"):
        ctx2.qs_elasticity(ctx2.empty_spec(), mu, lam, e0)


@pytest.mark.parametrize("n,closed", [(16, False), (32, True)])
def test_small_strain_linear_elastic_vs_oracle(n, closed):
    """mrl_mech_small_strain (the wording of BASELINE configs[2]: small-strain linear-elastic RVE, constant tangent) against the oracle's
    restatement (PARITY UNPINNED: the reference has no small-strain solve; it is FFTMechanics.C:96-163's first linear system at F = I):
    same CG iteration count, eps and sigma to 1e-12; 16^3 = generic path with the stored Ghat4, 32^3 = fused field-major path"""
    dom, phase, K, mu = _mech_setup(3, n)
    ctx = _ctx(3, dom.shape, [2 * math.pi] * 3)
    E = torch.tensor([[0.0, 0.01, 0.0], [0.01, 0.0, 0.0], [0.0, 0.0, 0.002]], dtype=torch.float64)
    eps_ref, sig_ref, its_ref = mo.small_strain_linear_elastic(dom, K, mu, E, 1e-6, closed_form=closed)
    eps, sig, st = ctx.mech_small_strain(K.cuda(), mu.cuda(), E.cuda(), l_tol=1e-6)
    assert st["cg_its"] == its_ref
    assert (eps.cpu() - eps_ref).abs().max().item() <= 1e-12
    assert (sig.cpu() - sig_ref).abs().max().item() <= 1e-12
    # the mean strain is the applied one (the CG correction has zero mean); sigma is symmetric
    assert (eps.mean(dim=(0, 1, 2)).cpu() - E).abs().max().item() <= 1e-13
    assert (sig - sig.transpose(-1, -2)).abs().max().item() <= 1e-12


def test_axpy_and_timing_summary():
    """mrl_axpy (y += a x: the vector update of conjugateGradientSolve) and mrl_get_timing (the profile summed over its kernel
    classes) -- the two names of the survey's boundary table that round 4 served only as mrl_axpby / mrl_get_profile"""
    from marlin_amd.api import Context
    ctx = Context(3, [16, 16, 16], [1.0, 1.0, 1.0])
    torch.manual_seed(0)
    x = torch.rand(16, 16, 16, dtype=torch.float64, device="cuda")
    y = torch.rand(16, 16, 16, dtype=torch.float64, device="cuda")
    want = y + 0.375 * x
    ctx.set_profiling(True)
    ctx.axpy(0.375, x, y)
    ctx.axpy(0.0, x, y)
    assert torch.equal(y, want)
    t = ctx.get_timing()
    prof = [k for k in ctx.get_profile() if k["launches"]]
    assert t["kernel_classes"] == len(prof) == 1 and t["launches"] == 2 and t["dominant"] == "axpby"
    assert abs(t["device_ms"] - sum(k["ms"] for k in prof)) <= 1e-12 and t["algorithmic_bytes"] == 2 * 24.0 * x.numel()

