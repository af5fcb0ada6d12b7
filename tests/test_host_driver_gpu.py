"""The reference's regression cases run through the C++ host mirror (marlin_amd/host, `marlin-hip-run`,
the stand-in for `marlin-opt -i case.i`) and diffed against the reference's gold files the way its
HDF5Diff tester does (scripts/TestHarness/testers/HDF5Diff.py:14-87: same datasets, max|a-b| <= abs_tol)."""
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu
RUN = os.path.join(ROOT, "marlin_amd", "lib", "marlin-hip-run")


def _run(args, tmp_path):
    assert os.path.exists(RUN), "marlin-hip-run has not been built (python -c 'import __graft_entry__ as g; g.build()')"
    out = subprocess.run([RUN] + args + [f"out={tmp_path}"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr + out.stdout
    return out.stdout


@pytest.mark.parametrize("free_energy", [["A=0.1"], ["expression=0.1*c^2*(c-1)^2"], ["A=0.1", "spectral_carry=1"],
                                         ["integrator=FFTSemiImplicit"], ["A=0.1", "substep_calls=1"]])
def test_cahnhilliard_case(free_energy, tmp_path):
    """test/tests/cahnhilliard/tests:46-57 (cahnhilliard.i): c.1..c.10, mu.10 vs gold, abs_tol 1e-13 -- with the
    built-in double well, with the input file's own ParsedCompute text (expression + derivatives = c), with the spectral
    carry-over, through the legacy FFTSemiImplicit time integrator (FFTSemiImplicit.C:43-62: the same scheme, operator by
    operator), and with one library call per substep instead of one per TensorSolver::computeBuffer (mrl_ch_substeps, default)"""
    g = load_golden("cahnhilliard_gold.npz")
    ic = tmp_path / "c0.bin"
    g["c.0"][:20, :20].astype("<f8").tofile(ic)     # the seed-0 RandomTensor IC is the gold file's frame 0
    _run(["problem=cahnhilliard", "dim=2", "nx=20", "ny=20", "xmax=3", "ymax=3", f"ic={ic}", "substeps=10",
          "num_steps=10", "dt=1e-3", "predictor_order=2", "mobility=0.2", "kappa=-0.001"] + free_energy, tmp_path)
    worst = 0.0
    for k in range(1, 11):
        c = np.fromfile(tmp_path / f"c.{k}.bin", dtype="<f8").reshape(20, 20)
        worst = max(worst, np.abs(g[f"c.{k}"][:20, :20] - c).max())
    assert worst <= 1e-13, worst
    mu = np.fromfile(tmp_path / "mu.10.bin", dtype="<f8").reshape(20, 20)
    assert np.abs(g["mu.10"] - mu).max() <= 1e-13


@pytest.mark.parametrize("case", ["mech3d", "mech2d"])
def test_mechanics_case(case, tmp_path):
    """test/tests/mechanics/tests:2-21 (mech3d.i / mech.i): F_k.frame vs gold, abs_tol 1e-10"""
    if case == "mech3d":
        dim, n, args = 3, 16, ["substeps=10", "dt=0.01", "l_tol=1e-2", "nl_rel_tol=2e-2", "nl_abs_tol=2e-2"]
        gold = "mech3d_gold.npz"
    else:
        dim, n, args = 2, 32, ["substeps=3", "dt=0.02", "l_tol=1e-5", "nl_rel_tol=2e-4", "nl_abs_tol=2e-3", "l_max_its=40"]
        gold = "mech2d_gold.npz"
    g = load_golden(gold)
    size = [f"{k}={n}" for k in ("nx", "ny", "nz")[:dim]] + [f"{k}=2pi" for k in ("xmax", "ymax", "zmax")[:dim]]
    _run(["problem=mechanics", f"dim={dim}", "num_steps=3"] + size + args, tmp_path)
    perm = (2, 1, 0) if dim == 3 else (1, 0)         # XDMF default transpose=true
    worst = 0.0
    for frame in range(3):
        F = np.fromfile(tmp_path / f"F.{frame}.bin", dtype="<f8").reshape([n] * dim + [dim * dim])
        for k in range(dim * dim):
            worst = max(worst, np.abs(g[f"F_{k}.{frame}"] - np.transpose(F[..., k], perm)).max())
        disp = np.fromfile(tmp_path / f"disp.{frame}.bin", dtype="<f8").reshape([n + 1] * dim + [dim])
        for k, nm in enumerate(("disp_x", "disp_y", "disp_z")[:dim]):
            worst = max(worst, np.abs(g[f"{nm}.{frame}"] - np.transpose(disp[..., k], perm)).max())
        if f"sV.{frame}" in g:
            sv = np.fromfile(tmp_path / f"sV.{frame}.bin", dtype="<f8").reshape([n] * dim)
            worst = max(worst, np.abs(g[f"sV.{frame}"] - np.transpose(sv, perm)).max())
    assert worst <= 1e-10, worst


@pytest.mark.parametrize("P,extra", [(2, []), (2, ["transport=2"]), (4, ["spectral_carry=1"]), (2, ["substep_calls=1"])])
def test_cahnhilliard_case_fft_slab(P, extra, tmp_path):
    """test/tests/cahnhilliard/tests:58-70: cahnhilliard.i with parallel_mode = FFT_SLAB on P rank PROCESSES started by the C++
    driver itself (`marlin-hip-run ... parallel_mode=FFT_SLAB nranks=P`, all on GPU 0): the C++ AdamsBashforthMoulton object over
    mrl_ch_substeps with the library-owned exchange; c.1 .. c.10 of cahnhilliard.rank0001.h5 to 1e-13"""
    import torch
    g = load_golden("cahnhilliard_rank0001_gold.npz")
    torch.manual_seed(0)
    blk = (torch.rand(20, 10, dtype=torch.float64) * (0.56 - 0.44) + 0.44).numpy()
    ic = tmp_path / "c0.bin"
    np.concatenate([blk, blk], axis=1).astype("<f8").tofile(ic)    # both reference ranks draw the same seed-0 block
    _run(["problem=cahnhilliard", "dim=2", "nx=20", "ny=20", "xmax=3", "ymax=3", f"ic={ic}", "substeps=10", "num_steps=10", "dt=1e-3",
          "predictor_order=2", "mobility=0.2", "kappa=-0.001", "parallel_mode=FFT_SLAB", f"nranks={P}", "device=0"] + extra, tmp_path)
    nyl = 20 // P
    worst = 0.0
    for k in range(1, 11):
        slabs = [np.fromfile(tmp_path / f"c.{k}.rank{r}.bin", dtype="<f8").reshape(20, nyl) for r in range(P)]
        c = np.concatenate(slabs, axis=1)
        worst = max(worst, np.abs(g[f"c.{k}"] - c[:, 10:]).max())
    assert worst <= 1e-13, worst


@pytest.mark.parametrize("extra", [[], ["substep_calls=1"], ["parallel_mode=FFT_SLAB", "nranks=2", "device=0"]])
def test_cahnhilliard_adaptive_dt(extra, tmp_path):
    """a [TimeStepper] that changes dt between steps: the C++ AdamsBashforthMoulton object restarts the order for the first
    predictor_order - 1 substeps of such a step (AdamsBashforthMoulton.C:75,88-91) -- one library call per computeBuffer, one per
    substep, and on two FFT_SLAB ranks -- against the oracle"""
    import torch
    from oracle import marlin_oracle as mo
    torch.manual_seed(3)
    shape, L = [20, 20], [3.0, 3.0]
    c0 = torch.rand(shape, dtype=torch.float64) * 0.12 + 0.44
    ic = tmp_path / "c0.bin"
    c0.numpy().astype("<f8").tofile(ic)
    dts, substeps, pred = [1e-3, 1e-3, 2e-3, 2e-3, 5e-4], 4, 3
    _run(["problem=cahnhilliard", "dim=2", "nx=20", "ny=20", "xmax=3", "ymax=3", f"ic={ic}", f"substeps={substeps}", f"num_steps={len(dts)}",
          "dt_sequence=" + ",".join(repr(d) for d in dts), f"predictor_order={pred}", "mobility=0.2", "kappa=-0.001"] + extra, tmp_path)
    slab = any(e.startswith("parallel_mode") for e in extra)
    dom = mo.Domain(2, shape, L, slab_c2c=slab)
    ref = mo.CahnHilliardABM(dom, c0, M=0.2, kappa_factor=-0.001, mu_fn=mo.mu_double_well, substeps=substeps, predictor_order=pred)
    for k, dt in enumerate(dts):
        ref.step(dt)
        if slab:
            c = np.concatenate([np.fromfile(tmp_path / f"c.{k + 1}.rank{r}.bin", dtype="<f8").reshape(20, 10) for r in range(2)], axis=1)
        else:
            c = np.fromfile(tmp_path / f"c.{k + 1}.bin", dtype="<f8").reshape(20, 20)
        assert np.abs(ref.c.numpy() - c).max() <= 1e-13


@pytest.mark.parametrize("slab", [False, True])
def test_xdmf_tensor_output_async(slab, tmp_path):
    """[TensorOutputs] XDMFTensorOutput (XDMFTensorOutput.C:278-343, 742-760; TensorOutput.C:66-81) in raw-binary mode through the
    asynchronous path: device-to-host copies on a side stream into pinned staging + a writer thread while the next time step
    computes.  cahnhilliard.i, 10 steps: file <base>[.rankNNNN].c.<frame>.bin holds c (transposed, as the reference writes it for
    Paraview) of time step frame + 1 == gold c.(frame + 1) to 1e-13; the .xmf is well-formed and lists every frame"""
    import xml.etree.ElementTree as ET
    if slab:
        import torch
        g = load_golden("cahnhilliard_rank0001_gold.npz")
        torch.manual_seed(0)
        blk = (torch.rand(20, 10, dtype=torch.float64) * (0.56 - 0.44) + 0.44).numpy()
        c0 = np.concatenate([blk, blk], axis=1)
        extra = ["parallel_mode=FFT_SLAB", "nranks=2", "device=0"]
    else:
        g = load_golden("cahnhilliard_gold.npz")
        c0 = g["c.0"][:20, :20]
        extra = []
    ic = tmp_path / "c0.bin"
    c0.astype("<f8").tofile(ic)
    out = _run(["problem=cahnhilliard", "dim=2", "nx=20", "ny=20", "xmax=3", "ymax=3", f"ic={ic}", "substeps=10", "num_steps=10", "dt=1e-3",
                "predictor_order=2", "mobility=0.2", "kappa=-0.001", "output=xdmf", "file_base=ch"] + extra, tmp_path)
    assert '"frames": 10' in out
    worst = 0.0
    for frame in range(10):
        if slab:
            c = np.fromfile(tmp_path / f"ch.rank0001.c.{frame}.bin", dtype="<f8").reshape(10, 20).T     # stored transposed: [y_local][x]
            worst = max(worst, np.abs(g[f"c.{frame + 1}"] - c).max())
        else:
            c = np.fromfile(tmp_path / f"ch.c.{frame}.bin", dtype="<f8").reshape(20, 20).T
            worst = max(worst, np.abs(g[f"c.{frame + 1}"][:20, :20] - c).max())
    assert worst <= 1e-13, worst
    if not slab:
        mu = np.fromfile(tmp_path / "ch.mu.9.bin", dtype="<f8").reshape(20, 20).T
        assert np.abs(g["mu.10"] - mu).max() <= 1e-13
    root = ET.parse(tmp_path / "ch.xmf").getroot()
    series = root.find("Domain").find("Grid")
    assert series.get("CollectionType") == "Temporal" and len(series.findall("Grid")) == 10
    items = [d.text for d in root.iter("DataItem") if d.get("Format") == "Binary"]
    assert len(items) == 10 * 2 * (2 if slab else 1) and all((tmp_path / t).exists() for t in items)


def _xml_tree(path):
    import xml.etree.ElementTree as ET

    def walk(e):
        return [e.tag, dict(sorted(e.attrib.items())), (e.text or "").strip(), [walk(c) for c in e]]
    return walk(ET.parse(path).getroot())


def _xml_same(a, b, where="/"):
    """XMLDiff: same structure, attribute names and text; numbers (in attributes or text) equal to 1e-12 relative"""
    def same_text(x, y):
        if x == y:
            return True
        xs, ys = x.split(), y.split()
        try:
            return len(xs) == len(ys) and all(abs(float(p) - float(q)) <= 1e-12 * max(1.0, abs(float(q))) for p, q in zip(xs, ys))
        except ValueError:
            return False
    assert a[0] == b[0], (where, a[0], b[0])
    assert sorted(a[1]) == sorted(b[1]), (where, a[1], b[1])
    for k in a[1]:
        assert same_text(a[1][k], b[1][k]), (where, k, a[1][k], b[1][k])
    assert same_text(a[2], b[2]), (where, a[2], b[2])
    assert len(a[3]) == len(b[3]), (where, len(a[3]), len(b[3]))
    for i, (x, y) in enumerate(zip(a[3], b[3])):
        _xml_same(x, y, f"{where}{a[0].split('}')[-1]}[{i}]/")


def test_xdmf_output_xml_and_hdf5_specs(tmp_path):
    """test/tests/cahnhilliard/tests:35-57 (xdmf_output_xml: XMLDiff against gold/cahnhilliard.xmf; xdmf_output_hdf5: HDF5Diff against
    gold/cahnhilliard.h5, abs_tol 1e-13): cahnhilliard.i as the reference runs it -- output on INITIAL and TIMESTEP_END, c as nodal
    data, mu per cell, one HDF5 container -- through the C++ mirror.  The XDMF description must be the gold file's tree (elements,
    attributes, HDF5 paths, times to 1e-12), and all eleven c.k datasets the gold container's (21 x 21, to 1e-13)"""
    import json

    from tests.h5_subset_reader import read_h5
    g = load_golden("cahnhilliard_gold.npz")
    ic = tmp_path / "c0.bin"
    g["c.0"][:20, :20].astype("<f8").tofile(ic)
    out = _run(["problem=cahnhilliard", "dim=2", "nx=20", "ny=20", "xmax=3", "ymax=3", f"ic={ic}", "substeps=10", "num_steps=10", "dt=1e-3",
                "predictor_order=2", "mobility=0.2", "kappa=-0.001", "output=xdmf", "enable_hdf5=true", "output_mode=NODE,CELL",
                "output_initial=true", "file_base=cahnhilliard"], tmp_path)
    assert '"frames": 11' in out
    sets = read_h5(tmp_path / "cahnhilliard.h5")
    assert sorted(sets) == sorted([f"c.{k}" for k in range(11)] + [f"mu.{k}" for k in range(11)])
    for k in range(11):
        assert np.abs(g[f"c.{k}"] - sets[f"c.{k}"].T).max() <= 1e-13, k
    assert np.abs(g["mu.10"] - sets["mu.10"].T).max() <= 1e-13
    with open(os.path.join(ROOT, "tests", "golden", "cahnhilliard_xmf_gold.json")) as f:
        gold_tree = json.load(f)
    _xml_same(_xml_tree(tmp_path / "cahnhilliard.xmf"), gold_tree)


@pytest.mark.parametrize("slab", [False, True])
def test_xdmf_tensor_output_hdf5(slab, tmp_path):
    """the same run with enable_hdf5 = true (XDMFTensorOutput.C:39, 152-160, 323-343, 244-246): one <base>[.rankNNNN].h5 per rank with
    the datasets "c.<frame>" / "mu.<frame>" in its root group, written by the library's own HDF5 container writer (mrl_h5_*; the image
    has no libhdf5) from the asynchronous output thread.  Read back with an independent reader of the file format
    (tests/h5_subset_reader.py) and, where the image has it, with h5dump: dataset names, shapes ([y][x]: transposed, as the reference
    writes for Paraview) and values == gold c.(frame + 1) to 1e-13 -- what the reference's HDF5Diff tester checks.  The serial case
    writes c in NODE mode as cahnhilliard.i does (output_mode, extendTensor: XDMFTensorOutput.C:41-51, 530-557), so its datasets have
    the gold file's own 21 x 21 shape and are compared whole"""
    import shutil
    import subprocess
    import xml.etree.ElementTree as ET

    from tests.h5_subset_reader import read_h5
    if slab:
        import torch
        g = load_golden("cahnhilliard_rank0001_gold.npz")
        torch.manual_seed(0)
        blk = (torch.rand(20, 10, dtype=torch.float64) * (0.56 - 0.44) + 0.44).numpy()
        c0 = np.concatenate([blk, blk], axis=1)
        extra = ["parallel_mode=FFT_SLAB", "nranks=2", "device=0"]
    else:
        g = load_golden("cahnhilliard_gold.npz")
        c0 = g["c.0"][:20, :20]
        extra = ["output_mode=NODE,CELL"]     # cahnhilliard.i writes c as nodal data (21 x 21 with the periodic wrap), mu per cell
    ic = tmp_path / "c0.bin"
    c0.astype("<f8").tofile(ic)
    out = _run(["problem=cahnhilliard", "dim=2", "nx=20", "ny=20", "xmax=3", "ymax=3", f"ic={ic}", "substeps=10", "num_steps=10", "dt=1e-3",
                "predictor_order=2", "mobility=0.2", "kappa=-0.001", "output=xdmf", "enable_hdf5=true", "file_base=ch"] + extra, tmp_path)
    assert '"frames": 10' in out
    name = "ch.rank0001.h5" if slab else "ch.h5"
    sets = read_h5(tmp_path / name)
    assert sorted(sets) == sorted([f"c.{k}" for k in range(10)] + [f"mu.{k}" for k in range(10)])
    h5dump = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else None)
    worst = 0.0
    for frame in range(10):
        c = sets[f"c.{frame}"]
        assert c.shape == ((10, 20) if slab else (21, 21)) and c.dtype == np.dtype("<f8")
        ref = g[f"c.{frame + 1}"]               # serial: the gold dataset as it is, including the wrapped row and column (NODE mode)
        worst = max(worst, np.abs(ref - c.T).max())
        if h5dump:
            raw = tmp_path / f"c.{frame}.raw"
            subprocess.run([h5dump, "-d", f"/c.{frame}", "-b", "LE", "-o", str(raw), str(tmp_path / name)], check=True, capture_output=True)
            assert np.array_equal(np.fromfile(raw, dtype="<f8").reshape(c.shape), c)
    assert worst <= 1e-13, worst
    if not slab:
        assert np.abs(g["mu.10"] - sets["mu.9"].T).max() <= 1e-13
    root = ET.parse(tmp_path / "ch.xmf").getroot()
    items = [d.text for d in root.iter("DataItem") if d.get("Format") == "HDF"]
    assert len(items) == 10 * 2 * (2 if slab else 1)
    if not slab:
        centers = {a.get("Name"): a.get("Center") for a in root.iter("Attribute")}
        assert centers == {"c": "Node", "mu": "Cell"}
    assert all(t.split(":/")[0] in ("ch.h5", "ch.rank0000.h5", "ch.rank0001.h5") and (tmp_path / t.split(":/")[0]).exists() for t in items)
    assert not list(tmp_path.glob("ch*.bin"))


def test_cahnhilliard_fft_slab_exchange_buffers_beyond_4_gib(tmp_path):
    """512 x 1024 x 1024 on TWO rank processes of the C++ driver (all-native processes: the system HIP runtime): rank-local spectral
    arrays of 2.2 GB, a 4.36 GB two-field forward exchange buffer and a 2.18 GB inverse one, mapped between the processes through HIP
    IPC; the fused slab pipeline with the 64-bit y pass.  Two AB substeps against the serial fused path on the same grid through the
    global checksums sum(c) and sum(c^2) (1e-13 relative; the fields themselves are compared at this size in
    tests/test_slab_native_gpu.py::test_native_slab_exchange_buffers_beyond_4_gib on one rank)"""
    import json
    common = ["problem=cahnhilliard", "dim=3", "nx=512", "ny=1024", "nz=1024", "xmax=8pi", "ymax=16pi", "zmax=16pi", "ic=splitmix64",
              "substeps=2", "num_steps=1", "dt=2e-3", "predictor_order=2", "mobility=0.2", "kappa=-0.001", "output=none"]
    par = json.loads(_run(common + ["parallel_mode=FFT_SLAB", "nranks=2", "device=0"], tmp_path).strip().splitlines()[-1])
    ser = json.loads(_run(common, tmp_path).strip().splitlines()[-1])
    assert abs(par["sum_c"] - ser["sum_c"]) <= 1e-13 * abs(ser["sum_c"]), (par, ser)
    assert abs(par["sum_c2"] - ser["sum_c2"]) <= 1e-13 * abs(ser["sum_c2"]), (par, ser)
    assert abs(ser["sum_c"] / (512 * 1024 * 1024) - 0.5) < 1e-3       # (mass of the uniform [0.44, 0.56] initial condition, conserved)


def test_mechanics_case_fft_slab(tmp_path):
    """mech3d.i (test/tests/mechanics/tests:2-21) on 2 rank processes: the C++ FFTMechanics object over mrl_mech_newton_cg on slab
    contexts; F_k.frame and sV of mech3d.h5 to 1e-10"""
    g = load_golden("mech3d_gold.npz")
    n, P = 16, 2
    _run(["problem=mechanics", "dim=3", "num_steps=3", "nx=16", "ny=16", "nz=16", "xmax=2pi", "ymax=2pi", "zmax=2pi", "substeps=10",
          "dt=0.01", "l_tol=1e-2", "nl_rel_tol=2e-2", "nl_abs_tol=2e-2", "parallel_mode=FFT_SLAB", f"nranks={P}", "device=0"], tmp_path)
    worst = 0.0
    for frame in range(3):
        F = np.concatenate([np.fromfile(tmp_path / f"F.{frame}.rank{r}.bin", dtype="<f8").reshape(n, n // P, n, 9) for r in range(P)], axis=1)
        for k in range(9):
            worst = max(worst, np.abs(g[f"F_{k}.{frame}"] - np.transpose(F[..., k], (2, 1, 0))).max())
        if f"sV.{frame}" in g:
            sv = np.concatenate([np.fromfile(tmp_path / f"sV.{frame}.rank{r}.bin", dtype="<f8").reshape(n, n // P, n) for r in range(P)], axis=1)
            worst = max(worst, np.abs(g[f"sV.{frame}"] - np.transpose(sv, (2, 1, 0))).max())
    assert worst <= 1e-10, worst


def test_mechanics_tensor_output_block_of_mech3d(tmp_path):
    """The [TensorOutputs] block of test/tests/mechanics/mech3d.i:95-103 reproduced as written -- buffer = 'disp sV F phase',
    output_mode = 'OVERSIZED_NODAL CELL CELL NODE' (XDMFTensorOutput.C:41-51,287), enable_hdf5 = true, TIMESTEP_END -- through the
    mirror's XDMFTensorOutput and the library's HDF5 writer: EVERY dataset of the gold file mech3d.h5 (disp_x/y/z on 17^3 nodes,
    phase extended periodically to 17^3, sV and F_0..F_8 on 16^3 cells, frames 0..2) by name, shape and value (1e-10)"""
    from tests.h5_subset_reader import read_h5
    g = load_golden("mech3d_gold.npz")
    _run(["problem=mechanics", "dim=3", "num_steps=3", "nx=16", "ny=16", "nz=16", "xmax=2pi", "ymax=2pi", "zmax=2pi", "substeps=10",
          "dt=0.01", "l_tol=1e-2", "nl_rel_tol=2e-2", "nl_abs_tol=2e-2", "output=xdmf", "enable_hdf5=true"], tmp_path)
    got = read_h5(tmp_path / "mech_out.h5")
    want = [k for k in g.files if "." in k]
    assert set(want) <= set(got), sorted(set(want) - set(got))
    worst = 0.0
    for k in want:
        assert got[k].shape == g[k].shape, (k, got[k].shape, g[k].shape)
        worst = max(worst, np.abs(got[k] - g[k]).max())
    assert worst <= 1e-10, worst
    assert got["disp_x.0"].shape == (17, 17, 17) and got["phase.2"].shape == (17, 17, 17) and got["F_8.1"].shape == (16, 16, 16)
    xmf = (tmp_path / "mech_out.xmf").read_text()
    assert 'Name="disp_x" Center="Node"' in xmf and 'Name="sV" Center="Cell"' in xmf and 'Name="phase" Center="Node"' in xmf


def test_error_behaviour(tmp_path):
    """mooseError-style failures: bad dimension, unreadable IC"""
    out = subprocess.run([RUN, "problem=cahnhilliard", "dim=4"], capture_output=True, text=True)
    assert out.returncode == 1 and "Unsupported mesh dimension" in out.stderr
    out = subprocess.run([RUN, "problem=cahnhilliard", "dim=2", "nx=8", "ny=8", "xmax=1", "ymax=1", "ic=/nonexistent"],
                         capture_output=True, text=True)
    assert out.returncode == 1 and "cannot read" in out.stderr


@pytest.mark.parametrize("name,ss,cs,order", [("diagonal_10_0_1", 10, 0, 1), ("diagonal_10_0_2", 10, 0, 2),
                                              ("diagonal_10_0_3", 10, 0, 3), ("diagonal_20_0_4", 20, 0, 4),
                                              ("diagonal_10_1_1", 10, 1, 1), ("diagonal_10_2_1", 10, 2, 1),
                                              ("diagonal_10_2_2", 10, 2, 2)])
def test_solver_cases(name, ss, cs, order, tmp_path):
    """test/tests/solvers/tests (diagonal.i, CSVDiff): 150^2 Brusselator through the native ParsedCompute, ForwardFFT,
    ReciprocalLaplacianFactor and the multi-variable ABM with Adams-Moulton corrector; min / max / integral per step
    against the reference's gold CSV (14 significant digits)"""
    g = load_golden("solvers_gold.npz")[name]
    _run(["problem=brusselator", "dim=2", "nx=150", "ny=150", "xmax=2pi", "ymax=2pi", f"ss={ss}", f"cs={cs}", f"order={order}",
          "num_steps=25", "dt=0.5"], tmp_path)
    got = np.loadtxt(tmp_path / "brusselator.csv", delimiter=",", skiprows=1)
    assert got.shape == g.shape
    assert np.allclose(got[:, 0], g[:, 0])
    err = np.abs(got[1:, 1:] - g[1:, 1:]) / np.maximum(1.0, np.abs(g[1:, 1:]))
    assert err.max() <= 5e-11, err.max()


_COUPLED = [(10, 0, 1), (10, 0, 2), (10, 0, 3), (20, 0, 4), (10, 1, 1), (10, 2, 1), (10, 2, 2)]


@pytest.mark.parametrize("kind", ["coupled", "nl_coupled"])
@pytest.mark.parametrize("ss,cs,order", _COUPLED)
def test_coupled_solver_cases(kind, ss, cs, order, tmp_path):
    """test/tests/solvers/tests (coupled.i: AdamsBashforthMoultonCoupled, per-k dense 2x2 solve incl. the reference's
    real cast of the right-hand side; nl_coupled.i: diagonal ABM with complex reciprocal-space ParsedComputes) vs gold CSV"""
    g = load_golden("solvers_gold.npz")[f"{kind}_{ss}_{cs}_{order}"]
    _run([f"problem={kind}", "dim=2", "nx=150", "ny=150", "xmax=2pi", "ymax=2pi", f"ss={ss}", f"cs={cs}", f"order={order}",
          "num_steps=25", "dt=10"], tmp_path)
    got = np.loadtxt(tmp_path / f"{kind}.csv", delimiter=",", skiprows=1)
    assert got.shape == g.shape
    assert np.allclose(got[:, 0], g[:, 0])
    err = np.abs(got[1:, 1:] - g[1:, 1:]) / np.maximum(1.0, np.abs(g[1:, 1:]))
    assert err.max() <= 5e-11, err.max()


def test_rotating_grain_secant_case(tmp_path):
    """test/tests/tensor_compute/tests:90-100 (rotating_grain_secant.i, HDF5Diff abs_tol 1e-10): SecantSolver +
    SwiftHohenbergLinear + TensorSolveIterationAdaptiveDT through the host mirror; psi.0 (a MOOSE ParsedFunction) is the IC"""
    import math
    g = load_golden("rotating_grain_secant_gold.npz")
    ic = tmp_path / "psi0.bin"
    g["psi.0"].astype("<f8").tofile(ic)
    ymax = 6 * math.pi * 2 / math.sin(math.pi / 3)
    log = _run(["problem=rotating_grain_secant", "dim=2", "nx=40", "ny=40", "xmax=12pi", f"ymax={ymax!r}", f"ic={ic}",
                "substeps=3", "num_steps=10", "dt=1"], tmp_path)
    assert log.count("converged=1") == 10
    worst = 0.0
    for k in range(0, 11):
        psi = np.fromfile(tmp_path / f"psi.{k}.bin", dtype="<f8").reshape(40, 40)
        worst = max(worst, np.abs(g[f"psi.{k}"] - psi).max())
    assert worst <= 1e-10, worst


@pytest.mark.parametrize("scale", [1.0, 0.5])
def test_rotating_grain_secant_with_predictor(scale, tmp_path):
    """the same case with a LinearTensorPredictor on psi (src/tensor_predictor/LinearTensorPredictor.C:19-46, applied by
    SecantSolver.C:100 through IterativeTensorSolverInterface::applyPredictors): host mirror against the oracle.  The reference
    has no test that uses a predictor, so this is parity against the restatement only (unpinned)"""
    import math

    import torch

    from oracle import marlin_oracle as mo
    from tests.test_oracle_golden import rotating_grain_problem
    g = load_golden("rotating_grain_secant_gold.npz")
    ic = tmp_path / "psi0.bin"
    g["psi.0"].astype("<f8").tofile(ic)
    ymax = 6 * math.pi * 2 / math.sin(math.pi / 3)
    log = _run(["problem=rotating_grain_secant", "dim=2", "nx=40", "ny=40", "xmax=12pi", f"ymax={ymax!r}", f"ic={ic}",
                "substeps=3", "num_steps=6", "dt=1", f"predictor_scale={scale}"], tmp_path)
    assert log.count("converged=1") == 6
    dom, state, compute, variables = rotating_grain_problem(torch.from_numpy(g["psi.0"]))
    solver = mo.SecantSolver(dom, state, compute, variables, substeps=3)
    solver.add_predictor("psi", scale)
    ts = mo.IterationAdaptiveDT(1.0, min_iterations=100, max_iterations=400, growth_factor=1.4, cutback_factor=0.9, dtmax=500.0)
    plain = 0.0
    for step in range(1, 7):
        solver.step(ts.next_dt(step, solver.iterations))
        psi = np.fromfile(tmp_path / f"psi.{step}.bin", dtype="<f8").reshape(40, 40)
        assert np.abs(state["psi"].numpy() - psi).max() <= 1e-10
        plain = max(plain, np.abs(g[f"psi.{step}"] - psi).max())
    assert plain <= 1e-7      # the predictor only moves the starting guess: the converged fields agree with the plain run's


def test_rotating_grain_secant_failed_solve_restores_old_solution(tmp_path):
    """SecantSolver.C:152-159,175-184: a solve that runs out of iterations (max_iterations = 2 here) is declared not converged and
    the old solution is restored (`_buffer = ifft(u_old)`); the next substep starts from it.  Host mirror against the oracle's
    restatement of the same control flow (the reference's gold run never rejects a solve)"""
    import math

    import torch

    from oracle import marlin_oracle as mo
    from tests.test_oracle_golden import rotating_grain_problem
    g = load_golden("rotating_grain_secant_gold.npz")
    ic = tmp_path / "psi0.bin"
    g["psi.0"].astype("<f8").tofile(ic)
    ymax = 6 * math.pi * 2 / math.sin(math.pi / 3)
    log = _run(["problem=rotating_grain_secant", "dim=2", "nx=40", "ny=40", "xmax=12pi", f"ymax={ymax!r}", f"ic={ic}",
                "substeps=3", "num_steps=3", "dt=1", "max_iterations=2"], tmp_path)
    assert log.count("converged=0") == 3 and "Solve not converged." in log
    dom, state, compute, variables = rotating_grain_problem(torch.from_numpy(g["psi.0"]))
    solver = mo.SecantSolver(dom, state, compute, variables, substeps=3, max_iterations=2)
    ts = mo.IterationAdaptiveDT(1.0, min_iterations=100, max_iterations=400, growth_factor=1.4, cutback_factor=0.9, dtmax=500.0)
    for step in range(1, 4):
        solver.step(ts.next_dt(step, solver.iterations))
        assert not solver.converged
        psi = np.fromfile(tmp_path / f"psi.{step}.bin", dtype="<f8").reshape(40, 40)
        assert np.abs(state["psi"].numpy() - psi).max() <= 1e-12
        # restored: the field is the initial condition up to one fft / ifft round trip per substep
        assert np.abs(g["psi.0"] - psi).max() <= 1e-13


def test_etdrk4_case(tmp_path):
    """test/tests/solvers/tests (etdrk4_diffusion.i): ETDRK4Solver built from fused parsed kernels vs gold mse / rmse"""
    g = load_golden("solvers_gold.npz")["etdrk4_diffusion_rmse"]
    _run(["problem=etdrk4_diffusion", "dim=1", "nx=64", "xmax=2pi", "D=0.05", "k=1.0", "ss=1", "dt=10", "num_steps=10"], tmp_path)
    got = np.loadtxt(tmp_path / "etdrk4.csv", delimiter=",", skiprows=1)
    assert got.shape == g.shape
    assert np.abs(got[1:, 1:] - g[1:, 1:]).max() <= 1e-12


def test_gradient_case(tmp_path):
    """test/tests/gradient/tests (gradient.i): FFTGradient of sin(x)+sin(y)+sin(z) on a 40^3 anisotropic box vs the
    analytic gradient; the gold value is the integrated round-off (7.6e-12) -- ours must be round-off too"""
    g = load_golden("fft_gold.npz")["gradient_out"]
    _run(["problem=gradient", "dim=3", "nx=40", "ny=40", "nz=40", "xmax=2pi", "ymax=4pi", "zmax=6pi"], tmp_path)
    got = np.loadtxt(tmp_path / "gradient.csv", delimiter=",", skiprows=1)
    assert 0.0 <= got[1, 1] <= 10.0 * g[1, 1]


def test_gradient_case_fft_slab(tmp_path):
    """test/tests/gradient/tests:10-19 (gradient_cpu_slab): the same input with parallel_mode = FFT_SLAB on THREE ranks (40 planes
    over 3 ranks: the uneven 14 / 13 / 13 split of partitionSlabs), here three rank processes on one GPU through the C++ driver --
    FFTGradient over the library's slab transforms (any-size stages, exchanges owned by the library), the integral all-reduced;
    against the serial gold CSV as in the reference"""
    g = load_golden("fft_gold.npz")["gradient_out"]
    _run(["problem=gradient", "dim=3", "nx=40", "ny=40", "nz=40", "xmax=2pi", "ymax=4pi", "zmax=6pi", "parallel_mode=FFT_SLAB", "nranks=3",
          "device=0"], tmp_path)
    got = np.loadtxt(tmp_path / "gradient.csv", delimiter=",", skiprows=1)
    assert 0.0 <= got[1, 1] <= 10.0 * g[1, 1]
    _run(["problem=gradient_square", "dim=3", "nx=40", "ny=40", "nz=40", "xmax=2pi", "ymax=4pi", "zmax=6pi", "parallel_mode=FFT_SLAB",
          "nranks=3", "device=0"], tmp_path)
    got2 = np.loadtxt(tmp_path / "gradient_square.csv", delimiter=",", skiprows=1)
    assert 0.0 <= got2[1, 1] <= 1e-10


def test_gradient_case_fft_pencil(tmp_path):
    """test/tests/gradient/tests:21-29 (gradient_cpu_pencil): the same input with parallel_mode = FFT_PENCIL on FOUR ranks (2 x 2
    pencils of the 40^3 box), here four rank processes on one GPU through the C++ driver -- FFTGradient over the library's pencil
    transforms (r2c along x, two staged exchanges each way, owned by the library), the integral all-reduced; against the serial gold
    CSV as in the reference"""
    g = load_golden("fft_gold.npz")["gradient_out"]
    _run(["problem=gradient", "dim=3", "nx=40", "ny=40", "nz=40", "xmax=2pi", "ymax=4pi", "zmax=6pi", "parallel_mode=FFT_PENCIL", "nranks=4",
          "device=0"], tmp_path)
    got = np.loadtxt(tmp_path / "gradient.csv", delimiter=",", skiprows=1)
    assert 0.0 <= got[1, 1] <= 10.0 * g[1, 1]
    _run(["problem=gradient_square", "dim=3", "nx=40", "ny=40", "nz=40", "xmax=2pi", "ymax=4pi", "zmax=6pi", "parallel_mode=FFT_PENCIL",
          "nranks=4", "device=0"], tmp_path)
    got2 = np.loadtxt(tmp_path / "gradient_square.csv", delimiter=",", skiprows=1)
    assert 0.0 <= got2[1, 1] <= 1e-10


def test_cahnhilliard_case_fft_pencil(tmp_path):
    """cahnhilliard.i's solver block with parallel_mode = FFT_PENCIL (every solver of the reference reaches the decomposition only
    through DomainAction::fft / ifft): 3 steps of 4 AB2 substeps on a 40 x 36 x 30 box on FOUR rank processes (2 x 2 pencils) through the
    C++ driver -- AdamsBashforthMoulton over mrl_ch_substeps on pencil contexts, the operator sequence over the staged transforms --
    against the serial run of the same input through the global checksums sum(c), sum(c^2) (1e-12 relative)"""
    import json
    common = ["problem=cahnhilliard", "dim=3", "nx=40", "ny=36", "nz=30", "xmax=5.0", "ymax=4.5", "zmax=3.75", "ic=splitmix64",
              "substeps=4", "num_steps=3", "dt=4e-3", "predictor_order=2", "mobility=0.2", "kappa=-0.001", "output=none"]
    par = json.loads(_run(common + ["parallel_mode=FFT_PENCIL", "nranks=4", "device=0"], tmp_path).strip().splitlines()[-1])
    ser = json.loads(_run(common, tmp_path).strip().splitlines()[-1])
    assert abs(par["sum_c"] - ser["sum_c"]) <= 1e-12 * abs(ser["sum_c"]), (par, ser)
    assert abs(par["sum_c2"] - ser["sum_c2"]) <= 1e-12 * abs(ser["sum_c2"]), (par, ser)
    assert abs(ser["sum_c2"] / (40 * 36 * 30) - 0.25) < 5e-3 and par["time"] == ser["time"]


def test_gradient_square_case(tmp_path):
    """test/tests/gradient/tests (gradient_square.i): FFTGradientSquare of sin(x)+sin(y)+sin(z) vs cos^2 sums; the gold
    value is integrated round-off (6.9e-12)"""
    g = load_golden("fft_gold.npz")["gradient_square_out"]
    _run(["problem=gradient_square", "dim=3", "nx=40", "ny=40", "nz=40", "xmax=2pi", "ymax=4pi", "zmax=6pi"], tmp_path)
    got = np.loadtxt(tmp_path / "gradient_square.csv", delimiter=",", skiprows=1)
    assert 0.0 <= got[1, 1] <= 10.0 * g[1, 1]


def test_postprocessors_case(tmp_path):
    """test/tests/postprocessors/tests (postprocessors.i): extreme_value.csv, average.csv, integral.csv,
    reciprocal_integral.csv at t = 0 and count.csv over two steps of the ForwardEulerSolver (10 substeps each); u integrates
    du/dt = c exactly, so its integral after t is t * int_c"""
    _run(["problem=postprocessors", "dim=2", "nx=40", "ny=40", "xmax=2", "ymax=3", "num_steps=2", "substeps=10", "dt=1"], tmp_path)
    got = np.loadtxt(tmp_path / "postprocessors.csv", delimiter=",", skiprows=1)
    assert np.allclose(got[0, 1:6], [3.2375, -1.6375, 0.8, 4.8, 4.8], rtol=0, atol=1e-12)   # the five gold CSVs
    assert got[:, 6].tolist() == [0, 10, 20]                                                  # count.csv
    assert np.allclose(got[:, 7], [0.0, 4.8, 9.6], rtol=0, atol=1e-11)


@pytest.mark.parametrize("method", ["SHARP", "HOULI"])
def test_cahnhilliard_explicit_smooth_case(method, tmp_path):
    """test/tests/cahnhilliard/tests:121-143 (cahnhilliard_explicit_smooth.i, Exodiff: rel 5.5e-6): explicit Euler (1000 substeps)
    with the DeAliasingTensor filter; nodal c and elemental mu of sharp.e / houli.e mapped back onto the 50 x 50 grid"""
    g = load_golden(f"cahnhilliard_explicit_{method.lower()}_gold.npz")
    ic = tmp_path / "c0.bin"
    g["c.0"].astype("<f8").tofile(ic)
    _run(["problem=cahnhilliard_explicit", "dim=2", "nx=50", "ny=50", "xmax=3", "ymax=3", f"ic={ic}", f"method={method}",
          "substeps=50", "num_steps=20", "dt=0.5"], tmp_path)
    for k in (1, 2, 5, 10, 20):
        c = np.fromfile(tmp_path / f"c.{k}.bin", dtype="<f8").reshape(50, 50)
        mu = np.fromfile(tmp_path / f"mu.{k}.bin", dtype="<f8").reshape(50, 50)
        assert np.abs(g[f"c.{k}"] - c).max() <= 1e-9
        assert np.abs(g[f"mu.{k}"] - mu).max() <= 1e-10


def test_cahnhilliard_3d_case(tmp_path):
    """test/tests/cahnhilliard/tests:13-22: cahnhilliard.i with Domain/dim=3 nx=ny=nz=5 zmax=3 vs map_to_aux_3d.e"""
    g = load_golden("cahnhilliard_3d_gold.npz")
    ic = tmp_path / "c0.bin"
    g["c.0"].astype("<f8").tofile(ic)
    _run(["problem=cahnhilliard", "dim=3", "nx=5", "ny=5", "nz=5", "xmax=3", "ymax=3", "zmax=3", f"ic={ic}", "substeps=10",
          "num_steps=10", "dt=1e-3", "predictor_order=2", "mobility=0.2", "kappa=-0.001", "A=0.1"], tmp_path)
    for k in range(1, 11):
        c = np.fromfile(tmp_path / f"c.{k}.bin", dtype="<f8").reshape(5, 5, 5)
        mu = np.fromfile(tmp_path / f"mu.{k}.bin", dtype="<f8").reshape(5, 5, 5)
        assert np.abs(g[f"c.{k}"] - c).max() <= 1e-13
        assert np.abs(g[f"mu.{k}"] - mu).max() <= 1e-13


def test_kks_no_flux_bc_case(tmp_path):
    """test/tests/kks/tests:13-31 (KKS_no_flux_bc.i; HDF5Diff abs_tol 1e-10, CSVDiff): two coupled variables, ParsedCompute with
    symbolic derivatives of the KKS Gibbs energy, ReciprocalMatDiffusion / ReciprocalAllenCahn with the smooth-boundary mask,
    AdamsBashforthMoulton order 3, 10 x 1000 substeps.  The initial fields (MOOSE parsed functions) are frame 0 of the gold file."""
    g = load_golden("kks_no_flux_bc_gold.npz")
    files = []
    for b in ("c", "eta", "psi"):
        f = tmp_path / f"{b}0.bin"
        g[f"{b}.0"].astype("<f8").tofile(f)
        files.append(f"{b}={f}")
    _run(["problem=kks", "dim=2", "nx=20", "ny=20", "xmin=-50", "xmax=50", "ymin=-50", "ymax=50", "num_steps=10", "dt=0.1",
          "substeps=1000", "predictor_order=3"] + files, tmp_path)
    worst = {}
    for b in ("c", "eta", "mu"):
        worst[b] = max(np.abs(g[f"{b}.{k}"] - np.fromfile(tmp_path / f"{b}.{k}.bin", dtype="<f8").reshape(20, 20)).max()
                       for k in range(1, 11))
    assert max(worst.values()) <= 1e-10, worst
    csv = np.loadtxt(tmp_path / "kks.csv", delimiter=",", skiprows=1)
    ref = load_golden("fft_gold.npz")["KKS_no_flux_bc_out"]
    assert csv.shape == ref.shape
    assert np.abs(csv - ref).max() <= 1e-9 * np.abs(ref).max()


def test_interface_velocity_case(tmp_path):
    """test/tests/postprocessors/tests (interface_velocity.i, CSVDiff): TensorInterfaceVelocityPostprocessor of c = sin(x + 0.2 t)
    on a 10 x 2 grid -- FFT gradients, the previous state of the buffer, the |grad| > 1e-3 mask, max reduction"""
    ref = load_golden("fft_gold.npz")["interface_velocity_out"]
    _run(["problem=interface_velocity", "dim=2", "nx=10", "ny=2", "xmax=4pi", "ymax=1", "num_steps=10", "dt=0.01"], tmp_path)
    got = np.loadtxt(tmp_path / "interface_velocity.csv", delimiter=",", skiprows=1)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1e-12


def test_smooth_rectangle_case(tmp_path):
    """test/tests/tensor_compute/tests:101-111 (smooth_rectangle.i, HDF5Diff abs_tol 1e-10): sharp / COS / TANH box profiles on
    100 x 100 (default transpose of the XDMF output)"""
    g = load_golden("smooth_rectangle_gold.npz")
    _run(["problem=smooth_rectangle", "dim=2", "nx=100", "ny=100", "xmax=20", "ymax=20"], tmp_path)
    for b in ("rectangle_sharp", "rectangle_cos", "rectangle_tanh"):
        got = np.fromfile(tmp_path / f"{b}.0.bin", dtype="<f8").reshape(100, 100)
        assert np.abs(g[f"{b}.0"] - got.T).max() <= 1e-13, b



def test_coupled_pf_mech_case(tmp_path):
    """test/tests/tensor_compute/coupled_pf_mech.i (Cahn-Hilliard + FFTQuasistaticElasticity + FFTElasticChemicalPotential through
    the legacy FFTSemiImplicit integrator; lambda = 100, mu = 50, e0 = 0.02) on a 16^3 grid through the host mirror against the
    oracle's restatement of the same input.  The reference has no gold data for this input: parity unpinned, see the oracle header."""
    import math

    import torch

    from oracle import marlin_oracle as mo
    n, substeps, steps, dt = 16, 5, 3, 0.05
    torch.manual_seed(5)
    c0 = torch.rand(n, n, n, dtype=torch.float64) * 0.12 + 0.44
    ic = tmp_path / "c0.bin"
    c0.numpy().astype("<f8").tofile(ic)
    _run(["problem=coupled_pf_mech", "dim=3", f"nx={n}", f"ny={n}", f"nz={n}", "xmax=4pi", "ymax=4pi", "zmax=4pi", f"ic={ic}",
          f"substeps={substeps}", f"num_steps={steps}", f"dt={dt}"], tmp_path)
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
    assert np.abs(ref.c.numpy() - c0.numpy()).max() > 1e-4      # the fields did evolve


def test_cahnhilliard_precision_float32_whole_run(tmp_path):
    """precision=float32 (the reference's per-run precision switch, MarlinUtils.C:39-44): the same time loop straight over
    mrl_ch_substeps_f32 -- 3 steps of 20 substeps on 64^3 from the splitmix64 initial condition -- ends at the float64 run's checksums to
    float32 accuracy (sum c: the scheme conserves mass; sum c^2 measures the field itself)"""
    import json
    base = ["problem=cahnhilliard", "dim=3", "nx=64", "ny=64", "nz=64", "xmax=8.04", "ymax=8.04", "zmax=8.04", "ic=splitmix64", "substeps=20",
            "num_steps=3", "dt=0.02", "predictor_order=2", "mobility=0.2", "kappa=-0.001", "output=none"]
    res = {}
    for prec in ("float64", "float32"):
        out = subprocess.run([RUN] + base + [f"precision={prec}", f"out={tmp_path}"], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        res[prec] = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    a, b = res["float64"], res["float32"]
    assert b["precision"] == "float32"
    assert abs(b["sum_c"] - a["sum_c"]) <= 2e-6 * abs(a["sum_c"])
    assert abs(b["sum_c2"] - a["sum_c2"]) <= 2e-6 * abs(a["sum_c2"])
    # a length outside the fp32 instantiations is refused with the reason
    bad = subprocess.run([RUN] + [x if not x.startswith("nx=") else "nx=96" for x in base] + ["precision=float32", f"out={tmp_path}"],
                         capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "float32" in (bad.stderr + bad.stdout)
