"""The MOOSE-side binding (marlin_plugin/*.C, *.h) EXECUTED: its files are compiled unchanged against the stand-ins of
tests/moose_stub/ (MooseObject / InputParameters / TensorProblem / TensorSolver / SplitOperatorBase / TensorOperator with the
reference's semantics, real libTorch tensors on the HIP device) and driven by `shim-driver` the way the Transient executioner drives
the objects of an input file.  The reference's regression inputs then run through `type = HipAdamsBashforthMoulton`,
`type = HipFFTMechanics`, `type = HipForwardFFT` ... and are diffed against its gold files as its HDF5Diff / CSVDiff testers do."""
import os
import re
import subprocess

import numpy as np
import pytest

from tests.conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu
RUN = os.path.join(ROOT, "marlin_amd", "lib", "shim-driver")


def _run(args, tmp_path):
    assert os.path.exists(RUN), "shim-driver has not been built (python -c 'import __graft_entry__ as g; g.build()')"
    out = subprocess.run([RUN] + args + [f"out={tmp_path}"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr + out.stdout
    return out.stdout


def _orders(log):
    return [int(m) for m in re.findall(r"substep \d+ order (\d+)", log)]


@pytest.mark.parametrize("fuse", ["true", "false"])
def test_cahnhilliard_gold_through_the_shim(fuse, tmp_path):
    """test/tests/cahnhilliard/tests:46-57 (cahnhilliard.i) with [TensorSolver] type = HipAdamsBashforthMoulton: c.1 .. c.10 and
    mu.10 of gold cahnhilliard.h5 to 1e-13, with the whole substep loop in one library call (default) and with the inherited
    TensorSolver::computeBuffer loop over substep().  Default predictor_order = 2: the ten substeps of the first time step are AB1
    (advanceState is a no-op while timeStep() <= 1), every later one AB2; _sub_dt = _dt / substeps and _sub_time ends at the step's
    time (TensorSolver.C:95-96,108)."""
    g = load_golden("cahnhilliard_gold.npz")
    ic = tmp_path / "c0.bin"
    g["c.0"][:20, :20].astype("<f8").tofile(ic)
    log = _run(["case=cahnhilliard", "nx=20", "ny=20", "xmax=3", "ymax=3", f"ic={ic}", "substeps=10", "num_steps=10", "dt=1e-3",
                f"fuse_substeps={fuse}"], tmp_path)
    assert _orders(log) == [0] * 10 + [1] * 90
    steps = re.findall(r"step (\d+) time (\S+) sub_time (\S+) sub_dt (\S+)", log)
    assert len(steps) == 10
    for k, t, st, sdt in steps:
        assert abs(float(sdt) - 1e-4) <= 1e-19 and abs(float(st) - float(t)) <= 1e-15 and abs(float(t) - int(k) * 1e-3) <= 1e-15
    worst = 0.0
    for k in range(1, 11):
        c = np.fromfile(tmp_path / f"c.{k}.bin", dtype="<f8").reshape(20, 20)
        worst = max(worst, np.abs(g[f"c.{k}"][:20, :20] - c).max())
    assert worst <= 1e-13, worst
    mu = np.fromfile(tmp_path / "mu.10.bin", dtype="<f8").reshape(20, 20)
    assert np.abs(g["mu.10"] - mu).max() <= 1e-13


@pytest.mark.parametrize("fuse", ["true", "false"])
def test_cahnhilliard_fft_slab_gold_through_the_shim_on_two_ranks(fuse, tmp_path):
    """test/tests/cahnhilliard/tests:58-70 (cahnhilliard.i, parallel_mode = FFT_SLAB, 2 ranks) through HipAdamsBashforthMoulton: the
    stub's launcher starts two rank processes on GPU 0, its DomainAction partitions as partitionSlabs does (DomainAction.C:510-566),
    and HipDomain takes its several-rank branch -- job name broadcast over the (stub) MPI communicator, mrl_comm_create,
    mrl_ctx_attach_comm, checkLayout against getLocalBounds.  Rank 1's c.1 .. c.10 equal cahnhilliard.rank0001.h5 to 1e-13."""
    import torch
    g = load_golden("cahnhilliard_rank0001_gold.npz")
    torch.manual_seed(0)
    blk = (torch.rand(20, 10, dtype=torch.float64) * (0.56 - 0.44) + 0.44).numpy()
    ic = tmp_path / "c0.bin"
    np.concatenate([blk, blk], axis=1).astype("<f8").tofile(ic)    # both reference ranks draw the same seed-0 block
    log = _run(["nranks=2", "case=cahnhilliard", "nx=20", "ny=20", "xmax=3", "ymax=3", f"ic={ic}", "substeps=10", "num_steps=10",
                "dt=1e-3", f"fuse_substeps={fuse}"], tmp_path)
    assert len(re.findall(r"step (\d+) time", log)) == 10
    worst = 0.0
    for k in range(1, 11):
        c = np.fromfile(tmp_path / f"c.{k}.rank1.bin", dtype="<f8").reshape(20, 10)
        worst = max(worst, np.abs(g[f"c.{k}"] - c).max())
    assert worst <= 1e-13, worst
    # the published spectral buffer is this rank's reciprocal block: 10 of the 20 x planes, all 20 y (c2c on both axes in 2-D FFT_SLAB)
    assert os.path.getsize(tmp_path / "Nhat.10.rank0.bin") == 10 * 20 * 16


@pytest.mark.parametrize("fuse", ["true", "false"])
def test_cahnhilliard_3d_fft_slab_through_the_shim_on_two_ranks(fuse, tmp_path):
    """a 3-D grid on two FFT_SLAB ranks through the shim (the library keeps r2c on z there, the one extent checkLayout lets differ
    from the DomainAction's): both ranks' slabs against the oracle's serial run to 1e-13, changing dt included"""
    import torch
    from oracle import marlin_oracle as mo
    torch.manual_seed(5)
    shape, L = [16, 12, 20], [3.0, 2.0, 4.0]
    c0 = torch.rand(shape, dtype=torch.float64) * 0.12 + 0.44
    ic = tmp_path / "c0.bin"
    c0.numpy().astype("<f8").tofile(ic)
    dts, substeps, pred = [1e-3, 1e-3, 2e-3], 3, 3
    _run(["nranks=2", "case=cahnhilliard", "nx=16", "ny=12", "nz=20", "xmax=3", "ymax=2", "zmax=4", f"ic={ic}", f"substeps={substeps}",
          "dt_sequence=" + ",".join(repr(d) for d in dts), f"predictor_order={pred}", f"fuse_substeps={fuse}"], tmp_path)
    ref = mo.CahnHilliardABM(mo.Domain(3, shape, L), c0, M=0.2, kappa_factor=-0.001, mu_fn=mo.mu_double_well, substeps=substeps,
                             predictor_order=pred)
    for k, dt in enumerate(dts):
        ref.step(dt)
        c = np.concatenate([np.fromfile(tmp_path / f"c.{k + 1}.rank{r}.bin", dtype="<f8").reshape(16, 6, 20) for r in range(2)], axis=1)
        assert np.abs(ref.c.numpy() - c).max() <= 1e-13


@pytest.mark.parametrize("fuse", ["true", "false"])
@pytest.mark.parametrize("pred", [1, 3])
def test_cahnhilliard_adaptive_dt_through_the_shim(fuse, pred, tmp_path):
    """a time step size that changes between steps (AdamsBashforthMoulton.C:75,88-91: the first predictor_order - 1 substeps of such a
    step run at first order), predictor_order = 1 (which the round-4 shim turned into an invalid-argument error) and 3, against the
    oracle: fields 1e-13, the order of every substep equal to the oracle's log"""
    import torch
    from oracle import marlin_oracle as mo
    torch.manual_seed(3)
    shape, L = [20, 20], [3.0, 3.0]
    c0 = torch.rand(shape, dtype=torch.float64) * 0.12 + 0.44
    ic = tmp_path / "c0.bin"
    c0.numpy().astype("<f8").tofile(ic)
    dts, substeps = [1e-3, 1e-3, 2e-3, 2e-3, 5e-4], 4
    log = _run(["case=cahnhilliard", "nx=20", "ny=20", "xmax=3", "ymax=3", f"ic={ic}", f"substeps={substeps}",
                "dt_sequence=" + ",".join(repr(d) for d in dts), f"predictor_order={pred}", f"fuse_substeps={fuse}"], tmp_path)
    ref = mo.CahnHilliardABM(mo.Domain(2, shape, L), c0, M=0.2, kappa_factor=-0.001, mu_fn=mo.mu_double_well, substeps=substeps,
                             predictor_order=pred)
    for k, dt in enumerate(dts):
        ref.step(dt)
        c = np.fromfile(tmp_path / f"c.{k + 1}.bin", dtype="<f8").reshape(20, 20)
        assert np.abs(ref.c.numpy() - c).max() <= 1e-13
    assert _orders(log) == ref.order_log


@pytest.mark.parametrize("fuse", ["true", "false"])
def test_cahnhilliard_3d_private_spectral_layout_through_the_shim(fuse, tmp_path):
    """64^3 is a fast-path shape: the Nhat history lives in the solver-private layout (x planes padded to an odd number of 256-byte
    pieces).  The shim publishes it as a strided view, TensorBuffer::advanceState moves those handles into the history, and the
    substep finds them again -- AB3 over three time steps against the oracle, and the published Mbarmubar values == the oracle's"""
    import torch
    from oracle import marlin_oracle as mo
    torch.manual_seed(5)
    n, L = 64, 3.0
    c0 = torch.rand([n] * 3, dtype=torch.float64) * 0.12 + 0.44
    ic = tmp_path / "c0.bin"
    c0.numpy().astype("<f8").tofile(ic)
    log = _run(["case=cahnhilliard", f"nx={n}", f"ny={n}", f"nz={n}", f"xmax={L}", f"ymax={L}", f"zmax={L}", f"ic={ic}", "substeps=3",
                "num_steps=3", "dt=1e-3", "predictor_order=3", f"fuse_substeps={fuse}"], tmp_path)
    ref = mo.CahnHilliardABM(mo.Domain(3, [n] * 3, [L] * 3), c0, M=0.2, kappa_factor=-0.001, mu_fn=mo.mu_double_well, substeps=3,
                             predictor_order=3)
    for k in range(3):
        ref.step(1e-3)
        c = np.fromfile(tmp_path / f"c.{k + 1}.bin", dtype="<f8").reshape(n, n, n)
        assert np.abs(ref.c.numpy() - c).max() <= 1e-13
        nh = np.fromfile(tmp_path / f"Nhat.{k + 1}.bin", dtype="<f8").reshape(n, n, n // 2 + 1, 2)
        want = torch.view_as_real(ref.Nhat).numpy()
        assert np.abs(want - nh).max() <= 2e-15 * np.abs(want).max() * n ** 3      # (the spectra tolerance of tests/test_fft_gpu.py)
    assert _orders(log) == ref.order_log == [0, 0, 0, 1, 2, 2, 2, 2, 2]


@pytest.mark.parametrize("ss,cs,order", [(10, 0, 1), (10, 0, 2), (10, 0, 3), (20, 0, 4), (10, 1, 1), (10, 2, 1), (10, 2, 2)])
def test_brusselator_gold_through_the_shim(ss, cs, order, tmp_path):
    """test/tests/solvers/tests (diagonal.i, CSVDiff): two variables, the compute group made of HipForwardFFT / HipParsedCompute,
    the linear operators from HipReciprocalLaplacianFactor, [TensorSolver] type = HipAdamsBashforthMoulton without `expression`
    (mrl_kspace_abm per variable, AdamsBashforthMoulton.C:80-101, corrector :117-177) against diagonal_<ss>_<cs>_<order>.csv"""
    g = load_golden("solvers_gold.npz")[f"diagonal_{ss}_{cs}_{order}"]
    _run(["case=brusselator", "nx=150", "ny=150", "xmax=2pi", "ymax=2pi", f"ss={ss}", f"cs={cs}", f"order={order}", "num_steps=25",
          "dt=0.5"], tmp_path)
    got = np.loadtxt(tmp_path / "brusselator.csv", delimiter=",", skiprows=1)
    assert got.shape == g.shape
    assert np.allclose(got[:, 0], g[:, 0])
    err = np.abs(got[1:, 1:] - g[1:, 1:]) / np.maximum(1.0, np.abs(g[1:, 1:]))
    assert err.max() <= 5e-11, err.max()


@pytest.mark.parametrize("ss,cs,order", [(10, 0, 1), (10, 0, 2), (10, 0, 3), (20, 0, 4), (10, 1, 1), (10, 2, 1), (10, 2, 2)])
def test_coupled_gold_through_the_shim(ss, cs, order, tmp_path):
    """test/tests/solvers/tests (coupled.i, CSVDiff): [TensorSolver] type = HipAdamsBashforthMoultonCoupled -- the right-hand sides
    of HipAdamsBashforthMoulton, then one dense 2 x 2 solve per k-point (mrl_kspace_coupled) with the reference's transposed
    assembly and real cast of the right-hand side, which its gold files pin -- against coupled_<ss>_<cs>_<order>.csv"""
    g = load_golden("solvers_gold.npz")[f"coupled_{ss}_{cs}_{order}"]
    _run(["case=coupled", "nx=150", "ny=150", "xmax=2pi", "ymax=2pi", f"ss={ss}", f"cs={cs}", f"order={order}", "num_steps=25",
          "dt=10"], tmp_path)
    got = np.loadtxt(tmp_path / "coupled.csv", delimiter=",", skiprows=1)
    assert got.shape == g.shape
    assert np.allclose(got[:, 0], g[:, 0])
    err = np.abs(got[1:, 1:] - g[1:, 1:]) / np.maximum(1.0, np.abs(g[1:, 1:]))
    assert err.max() <= 5e-11, err.max()


def test_etdrk4_gold_through_the_shim(tmp_path):
    """test/tests/solvers/tests:220-230 (etdrk4_diffusion.i, CSVDiff): [TensorSolver] type = HipETDRK4Solver on the 1-D 64-point
    domain, the exact solution and the squared difference as HipParsedComputes with `t`: mse / rmse of etdrk4_diffusion_rmse.csv"""
    g = load_golden("solvers_gold.npz")["etdrk4_diffusion_rmse"]
    _run(["case=etdrk4", "nx=64", "xmax=2pi", "D=0.05", "k=1.0", "ss=1", "dt=10", "num_steps=10"], tmp_path)
    got = np.loadtxt(tmp_path / "etdrk4.csv", delimiter=",", skiprows=1)
    assert got.shape == g.shape
    assert np.abs(got[1:, 1:] - g[1:, 1:]).max() <= 1e-12


def test_rotating_grain_secant_gold_through_the_shim(tmp_path):
    """test/tests/tensor_compute/tests:90-100 (rotating_grain_secant.i, HDF5Diff abs_tol 1e-10): [TensorSolver] type = HipSecantSolver,
    the Swift-Hohenberg linear operator as a HipParsedCompute on the reciprocal grid, the time stepper following the solver's
    iteration count (IterativeTensorSolverInterface); psi.0 (a MOOSE ParsedFunction) is the IC"""
    import math
    g = load_golden("rotating_grain_secant_gold.npz")
    ic = tmp_path / "psi0.bin"
    g["psi.0"].astype("<f8").tofile(ic)
    ymax = 6 * math.pi * 2 / math.sin(math.pi / 3)
    log = _run(["case=secant", "nx=40", "ny=40", "xmax=12pi", f"ymax={ymax!r}", f"ic={ic}", "substeps=3", "num_steps=10", "dt=1"], tmp_path)
    assert log.count("converged=1") == 10
    worst = 0.0
    for k in range(0, 11):
        psi = np.fromfile(tmp_path / f"psi.{k}.bin", dtype="<f8").reshape(40, 40)
        worst = max(worst, np.abs(g[f"psi.{k}"] - psi).max())
    assert worst <= 1e-10, worst


@pytest.mark.parametrize("n", [24, 64])
def test_broyden_vs_oracle_through_the_shim(n, tmp_path):
    """[TensorSolver] type = HipBroydenSolver on two coupled reaction-diffusion variables (the problem of tests/test_broyden_gpu.py;
    the reference has neither an input nor gold data for BroydenSolver: **parity unpinned**, oracle restatement only): three substeps,
    the oracle's iteration counts and convergence flags in every one, fields 1e-11 after the first (converging) substep and 1e-6
    through the ill-conditioned ones (rank-one updates divided by s.y down to 1e-12)"""
    import math
    import torch
    import oracle.marlin_oracle as mo
    dom = mo.Domain(2, [n, n], [2.0 * math.pi] * 2)
    state = {"u": (1.0 + 0.1 * torch.sin(dom.axis[0]) * torch.sin(dom.axis[1])).expand(dom.shape).contiguous(),
             "v": (3.0 + 0.1 * torch.cos(dom.axis[0]) * torch.cos(2 * dom.axis[1])).expand(dom.shape).contiguous()}
    Du, Dv = mo.reciprocal_laplacian_factor(dom, 1e-2), mo.reciprocal_laplacian_factor(dom, 1e-3)

    def compute(s):
        u, v = s["u"], s["v"]
        s["u_bar"], s["v_bar"] = dom.fft(u), dom.fft(v)
        s["su_bar"] = dom.fft((1.0 - (3.5 + 1.0) * u) + torch.pow(u, 2.0) * v)
        s["sv_bar"] = dom.fft(3.5 * u - torch.pow(u, 2.0) * v)

    ref = mo.BroydenSolver(dom, state, compute, [("u", "u_bar", Du, "su_bar"), ("v", "v_bar", Dv, "sv_bar")], substeps=3, max_iterations=30,
                           relative_tolerance=1e-6, absolute_tolerance=1e-10)
    log = _run(["case=broyden", f"nx={n}", f"ny={n}", "xmax=2pi", "ymax=2pi", "num_steps=3", "dt=0.05"], tmp_path)
    got = [(int(i), bool(int(c))) for i, c in re.findall(r"iterations=(\d+) converged=(\d)", log)]
    want, errs = [], []
    for k in range(3):
        ref.substep(0.05)
        want.append((ref.iterations, ref.converged))
        u = np.fromfile(tmp_path / f"u.{k + 1}.bin", dtype="<f8").reshape(n, n)
        v = np.fromfile(tmp_path / f"v.{k + 1}.bin", dtype="<f8").reshape(n, n)
        errs.append(max(np.abs(u - state["u"].numpy()).max(), np.abs(v - state["v"].numpy()).max()))
    assert got == want, (got, want)
    assert errs[0] <= 1e-11 and max(errs) <= 1e-6, errs


@pytest.mark.parametrize("method", ["SHARP", "HOULI"])
def test_cahnhilliard_explicit_smooth_gold_through_the_shim(method, tmp_path):
    """test/tests/cahnhilliard/tests:121-143 (cahnhilliard_explicit_smooth.i, Exodiff): [TensorSolver] type = HipForwardEulerSolver
    (1000 explicit substeps), the rate de-aliased by a HipDeAliasingTensor inside a reciprocal-grid HipParsedCompute; nodal c and
    elemental mu of sharp.e / houli.e mapped back onto the 50 x 50 grid"""
    g = load_golden(f"cahnhilliard_explicit_{method.lower()}_gold.npz")
    ic = tmp_path / "c0.bin"
    g["c.0"].astype("<f8").tofile(ic)
    _run(["case=explicit", "nx=50", "ny=50", "xmax=3", "ymax=3", f"ic={ic}", f"method={method}", "substeps=50", "num_steps=20", "dt=0.5"],
         tmp_path)
    for k in (1, 2, 5, 10, 20):
        c = np.fromfile(tmp_path / f"c.{k}.bin", dtype="<f8").reshape(50, 50)
        mu = np.fromfile(tmp_path / f"mu.{k}.bin", dtype="<f8").reshape(50, 50)
        assert np.abs(g[f"c.{k}"] - c).max() <= 1e-9
        assert np.abs(g[f"mu.{k}"] - mu).max() <= 1e-10


def test_cahnhilliard_gold_through_the_legacy_time_integrator_shim(tmp_path):
    """cahnhilliard.i in the pre-TensorSolver syntax: explicit compute group (HipParsedCompute with `derivatives = c`, HipForwardFFT x 2,
    Mbar * mubar as a reciprocal-grid HipParsedCompute) + [TensorTimeIntegrators] type = HipFFTSemiImplicit (history_size 1: first
    order while timeStep() <= 1, then (3 N - N_old) / 2) -- the same numbers as AdamsBashforthMoulton orders 1 / 2 to rounding: gold
    cahnhilliard.h5 to 1e-13"""
    g = load_golden("cahnhilliard_gold.npz")
    ic = tmp_path / "c0.bin"
    g["c.0"][:20, :20].astype("<f8").tofile(ic)
    _run(["case=semi_implicit", "nx=20", "ny=20", "xmax=3", "ymax=3", f"ic={ic}", "substeps=10", "num_steps=10", "dt=1e-3"], tmp_path)
    worst = 0.0
    for k in range(1, 11):
        c = np.fromfile(tmp_path / f"c.{k}.bin", dtype="<f8").reshape(20, 20)
        worst = max(worst, np.abs(g[f"c.{k}"][:20, :20] - c).max())
    assert worst <= 1e-13, worst
    mu = np.fromfile(tmp_path / "mu.10.bin", dtype="<f8").reshape(20, 20)
    assert np.abs(g["mu.10"] - mu).max() <= 1e-13


def test_coupled_pf_mech_vs_oracle_through_the_shim(tmp_path):
    """test/tests/tensor_compute/coupled_pf_mech.i (Cahn-Hilliard + HipFFTQuasistaticElasticity + HipFFTElasticChemicalPotential
    + HipInverseFFT through the legacy HipFFTSemiImplicit integrator; lambda = 100, mu = 50, e0 = 0.02) on a 16^3 grid against the oracle's
    restatement of the same input.  The reference has no gold data for this input: **parity unpinned**, see the oracle header."""
    import math
    import torch
    from oracle import marlin_oracle as mo
    n, substeps, steps, dt = 16, 5, 3, 0.05
    torch.manual_seed(5)
    c0 = torch.rand(n, n, n, dtype=torch.float64) * 0.12 + 0.44
    ic = tmp_path / "c0.bin"
    c0.numpy().astype("<f8").tofile(ic)
    _run(["case=coupled_pf_mech", f"nx={n}", f"ny={n}", f"nz={n}", "xmax=4pi", "ymax=4pi", "zmax=4pi", f"ic={ic}", f"substeps={substeps}",
          f"num_steps={steps}", f"dt={dt}"], tmp_path)
    dom = mo.Domain(3, [n] * 3, [4 * math.pi] * 3)
    ref = mo.CoupledPFMech(dom, c0, 0.2, -0.001, mo.mu_double_well, substeps, 50.0, 100.0, 0.02)
    for k in range(1, steps + 1):
        ref.step(dt)
        c = np.fromfile(tmp_path / f"c.{k}.bin", dtype="<f8").reshape(n, n, n)
        assert np.abs(c - ref.c.numpy()).max() <= 1e-13
        mm = np.fromfile(tmp_path / f"mumech.{k}.bin", dtype="<f8").reshape(n, n, n)
        assert np.abs(mm - ref.mumech.numpy()).max() <= 1e-12 * np.abs(ref.mumech.numpy()).max()
        for d, nm in enumerate(("disp_x", "disp_y", "disp_z")):
            u = np.fromfile(tmp_path / f"{nm}.{k}.bin", dtype="<f8").reshape(n, n, n)
            assert np.abs(u - ref.disp[d].numpy()).max() <= 1e-12 * np.abs(ref.disp[d].numpy()).max()


def test_kks_no_flux_bc_gold_through_the_shim(tmp_path):
    """test/tests/kks/tests:13-31 (KKS_no_flux_bc.i; HDF5Diff abs_tol 1e-10, CSVDiff): two coupled variables through
    HipAdamsBashforthMoulton order 3 (10 x 1000 substeps, no linear operator), the Gibbs-energy derivatives as HipParsedComputes,
    HipReciprocalMatDiffusion / HipReciprocalAllenCahn with the smooth-boundary mask.  Frame 0 of the gold file is the IC."""
    g = load_golden("kks_no_flux_bc_gold.npz")
    files = []
    for b in ("c", "eta", "psi"):
        f = tmp_path / f"{b}0.bin"
        g[f"{b}.0"].astype("<f8").tofile(f)
        files.append(f"{b}={f}")
    _run(["case=kks", "nx=20", "ny=20", "xmin=-50", "xmax=50", "ymin=-50", "ymax=50", "num_steps=10", "dt=0.1", "substeps=1000",
          "predictor_order=3"] + files, tmp_path)
    worst = {}
    for b in ("c", "eta", "mu"):
        worst[b] = max(np.abs(g[f"{b}.{k}"] - np.fromfile(tmp_path / f"{b}.{k}.bin", dtype="<f8").reshape(20, 20)).max()
                       for k in range(1, 11))
    assert max(worst.values()) <= 1e-10, worst
    csv = np.loadtxt(tmp_path / "kks.csv", delimiter=",", skiprows=1)
    ref = load_golden("fft_gold.npz")["KKS_no_flux_bc_out"]
    assert csv.shape == ref.shape
    assert np.abs(csv - ref).max() <= 1e-9 * np.abs(ref).max()


def test_mech3d_gold_through_the_shim(tmp_path):
    """test/tests/mechanics/tests:2-11 (mech3d.i) with [mech] type = HipFFTMechanics inside the root group of a HipForwardEulerSolver
    that forwards Fnew -> F: F_k.frame, disp_* (HipComputeDisplacements) and sV (HipComputeVonMisesStress) of gold mech3d.h5 to
    1e-10, two Newton iterations per substep as the reference"""
    g = load_golden("mech3d_gold.npz")
    n = 16
    log = _run(["case=mechanics", "nx=16", "ny=16", "nz=16", "xmax=2pi", "ymax=2pi", "zmax=2pi", "substeps=10", "num_steps=3", "dt=0.01",
                "l_tol=1e-2", "nl_rel_tol=2e-2", "nl_abs_tol=2e-2"], tmp_path)
    its = [int(m) for m in re.findall(r"(\d+) Newton iterations", log)]
    assert len(its) == 30 and set(its) == {2}
    worst = 0.0
    for frame in range(3):
        F = np.fromfile(tmp_path / f"F.{frame}.bin", dtype="<f8").reshape(n, n, n, 9)
        for k in range(9):
            worst = max(worst, np.abs(g[f"F_{k}.{frame}"] - np.transpose(F[..., k], (2, 1, 0))).max())   # XDMF default transpose
        disp = np.fromfile(tmp_path / f"disp.{frame}.bin", dtype="<f8").reshape(n + 1, n + 1, n + 1, 3)
        for k, nm in enumerate(("disp_x", "disp_y", "disp_z")):      # [displacements] type = HipComputeDisplacements
            worst = max(worst, np.abs(g[f"{nm}.{frame}"] - np.transpose(disp[..., k], (2, 1, 0))).max())
        sv = np.fromfile(tmp_path / f"sV.{frame}.bin", dtype="<f8").reshape(n, n, n)   # [vonmises] type = HipComputeVonMisesStress
        worst = max(worst, np.abs(g[f"sV.{frame}"] - np.transpose(sv, (2, 1, 0))).max())
    assert worst <= 1e-10, worst


@pytest.mark.parametrize("case", ["gradient", "gradient_square"])
def test_gradient_cases_through_the_shim_on_four_pencil_ranks(case, tmp_path):
    """test/tests/gradient/tests, the reference's 4-rank FFT_PENCIL spec: the stub's DomainAction partitions as partitionPencils does
    (DomainAction.C:568-698: 2 x 2, r2c on x), HipDomain recognises the pencil partition (a local z extent shorter than the global one),
    sets MRL_FLAG_PENCIL and checks the library's blocks against the DomainAction's; HipParsedCompute / HipForwardFFT / HipFFTGradient /
    HipFFTGradientSquare work on a rank's pencil.  The four partial integrals add up to the serial run's postprocessor value."""
    args = [f"case={case}", "nx=40", "ny=40", "nz=40", "xmax=2pi", "ymax=4pi", "zmax=6pi"]
    _run(args, tmp_path)
    serial = np.loadtxt(tmp_path / f"{case}.csv", delimiter=",", skiprows=1)[-1, 1]
    _run(["nranks=4", "parallel_mode=FFT_PENCIL"] + args, tmp_path)
    parts = [np.loadtxt(tmp_path / f"{case}.rank{r}.csv", delimiter=",", skiprows=1)[-1, 1] for r in range(4)]
    assert abs(sum(parts) - serial) <= 1e-10 * max(1.0, abs(serial)), (parts, serial)
    assert serial < 1e-9     # (spectral derivatives of sin(x) + sin(y) + sin(z) are exact to rounding: gradient*_out.csv)


def test_cahnhilliard_3d_fft_pencil_through_the_shim_on_four_ranks(tmp_path):
    """Cahn-Hilliard substeps on four FFT_PENCIL ranks through HipAdamsBashforthMoulton (the library runs the unfused operator
    sequence over its staged pencil transforms): the four pencils against the oracle's serial run to 1e-13"""
    import torch
    from oracle import marlin_oracle as mo
    torch.manual_seed(6)
    shape, L = [16, 12, 20], [3.0, 2.0, 4.0]
    c0 = torch.rand(shape, dtype=torch.float64) * 0.12 + 0.44
    ic = tmp_path / "c0.bin"
    c0.numpy().astype("<f8").tofile(ic)
    dts, substeps, pred = [1e-3, 1e-3, 2e-3], 3, 2
    _run(["nranks=4", "parallel_mode=FFT_PENCIL", "case=cahnhilliard", "nx=16", "ny=12", "nz=20", "xmax=3", "ymax=2", "zmax=4", f"ic={ic}",
          f"substeps={substeps}", "dt_sequence=" + ",".join(repr(d) for d in dts), f"predictor_order={pred}"], tmp_path)
    ref = mo.CahnHilliardABM(mo.Domain(3, shape, L), c0, M=0.2, kappa_factor=-0.001, mu_fn=mo.mu_double_well, substeps=substeps,
                             predictor_order=pred)
    for k, dt in enumerate(dts):
        ref.step(dt)
        blocks = [np.fromfile(tmp_path / f"c.{k + 1}.rank{r}.bin", dtype="<f8").reshape(16, 6, 10) for r in range(4)]
        c = np.concatenate([np.concatenate([blocks[0], blocks[1]], axis=1), np.concatenate([blocks[2], blocks[3]], axis=1)], axis=2)
        assert np.abs(ref.c.numpy() - c).max() <= 1e-13


def test_mech3d_gold_through_the_shim_on_two_ranks(tmp_path):
    """mech3d.i with parallel_mode = FFT_SLAB on two rank processes through HipFFTMechanics (the Newton-CG solve on the library-owned
    slab pipeline, global CG scalars through its mailbox all-reduce), HipParsedCompute with the coordinate symbols on a rank's block,
    HipComputeVonMisesStress: both ranks' slabs of F_k and sV against gold mech3d.h5 to 1e-10, the serial run's Newton counts"""
    g = load_golden("mech3d_gold.npz")
    n = 16
    log = _run(["nranks=2", "case=mechanics", "nx=16", "ny=16", "nz=16", "xmax=2pi", "ymax=2pi", "zmax=2pi", "substeps=10", "num_steps=3",
                "dt=0.01", "l_tol=1e-2", "nl_rel_tol=2e-2", "nl_abs_tol=2e-2"], tmp_path)
    its = [int(m) for m in re.findall(r"(\d+) Newton iterations", log)]
    assert len(its) >= 30 and set(its) == {2}, its
    worst = 0.0
    for frame in range(3):
        F = np.concatenate([np.fromfile(tmp_path / f"F.{frame}.rank{r}.bin", dtype="<f8").reshape(n, n // 2, n, 9) for r in range(2)], axis=1)
        for k in range(9):
            worst = max(worst, np.abs(g[f"F_{k}.{frame}"] - np.transpose(F[..., k], (2, 1, 0))).max())
        sv = np.concatenate([np.fromfile(tmp_path / f"sV.{frame}.rank{r}.bin", dtype="<f8").reshape(n, n // 2, n) for r in range(2)], axis=1)
        worst = max(worst, np.abs(g[f"sV.{frame}"] - np.transpose(sv, (2, 1, 0))).max())
    assert worst <= 1e-10, worst


@pytest.mark.parametrize("case,bound", [("gradient", None), ("gradient_square", 1e-10)])
def test_gradient_cases_through_the_shim(case, bound, tmp_path):
    """test/tests/gradient/tests (gradient.i, gradient_square.i): HipFFTGradient (X, Y, Z) and HipFFTGradientSquare of
    sin(x)+sin(y)+sin(z) on the 40^3 anisotropic box against the analytic derivatives, every expression a HipParsedCompute; the gold
    value is integrated round-off (7.6e-12 / 6.9e-12) -- ours must be round-off too"""
    g = load_golden("fft_gold.npz")["gradient_out" if case == "gradient" else "gradient_square_out"]
    _run([f"case={case}", "nx=40", "ny=40", "nz=40", "xmax=2pi", "ymax=4pi", "zmax=6pi"], tmp_path)
    got = np.loadtxt(tmp_path / f"{case}.csv", delimiter=",", skiprows=1)
    assert 0.0 <= got[1, 1] <= (bound if bound else 10.0 * g[1, 1])
