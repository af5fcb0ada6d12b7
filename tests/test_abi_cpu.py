"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol include/marlin_hip.h
declares, and its host-only entry points (partition, reciprocal axes, error paths) follow the reference."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest
import torch

from oracle import marlin_oracle as mo
from tests.conftest import ROOT


def test_library_exports_every_declared_symbol():
    from marlin_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "marlin_hip.h")).read()
    declared = set(re.findall(r"\b(mrl_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in marlin_hip.h but not exported"
    assert set(_lib.SIGNATURES) == declared
    assert lib.mrl_abi_version() == 3


def test_partition_matches_reference_helper():
    from marlin_amd.api import partition
    for total, w in [(20, [1, 1]), (21, [1, 1, 1]), (22, [1, 1, 1]), (512, [1] * 8), (10, [3, 1]), (257, [1] * 8), (7, [5, 1, 1])]:
        assert partition(total, len(w), w) == mo.partition_helper(total, w)


@pytest.mark.parametrize("n,dx,rfft", [(20, 0.15, False), (20, 0.15, True), (9, 1.0 / 9, False), (9, 1.0 / 9, True),
                                       (256, 8 * math.pi / 200, True), (1, 1.0, False)])
def test_reciprocal_axis_bit_exact(n, dx, rfft):
    """k = 2*pi*fftfreq / rfftfreq with the reference's rounding sequence (DomainAction.C:268-293)"""
    from marlin_amd.api import reciprocal_axis
    f = (torch.fft.rfftfreq if rfft else torch.fft.fftfreq)(n, dx, dtype=torch.float64) * 2.0 * math.pi
    assert np.array_equal(np.array(reciprocal_axis(n, dx, rfft)), f.numpy())


def test_error_paths_without_gpu():
    from marlin_amd import _lib
    from marlin_amd._lib import MrlDomain
    lib = _lib.load()
    h = C.c_void_p()
    d = MrlDomain()
    d.dim = 4
    assert lib.mrl_ctx_create(C.byref(h), C.byref(d)) == -1
    assert b"Unsupported mesh dimension" in lib.mrl_last_error(None)
    d.dim = 2
    d.n[0], d.n[1] = 8, 8
    d.max[0], d.max[1] = 1.0, 0.0
    d.nranks, d.rank = 1, 0
    assert lib.mrl_ctx_create(C.byref(h), C.byref(d)) == -1
    assert b"Max coordinate" in lib.mrl_last_error(None)
    out = (C.c_int64 * 4)()
    assert lib.mrl_partition(2, 4, None, out) == -1
    if not torch.cuda.is_available():
        d.max[1] = 1.0
        d.device = -1
        rc = lib.mrl_ctx_create(C.byref(h), C.byref(d))
        assert rc == -3 and b"no HIP device" in lib.mrl_last_error(None)   # fails loudly: no CPU fallback


def test_bench_launch_chain_refuses_without_touching_a_gpu():
    """`bench.py --gpus N` on a box where N rank processes do not fit the visible cards (here: no GPU at all, or a one-GPU box with N = 8)
    says so and leaves with rc 2 before any rank is started; the N = 1 path is not affected (it needs a GPU and fails loudly)"""
    import subprocess
    import sys
    if torch.cuda.device_count() > 1:
        pytest.skip("a multi-GPU box would run the job")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 2 and "rank processes per card" in r.stderr, (r.returncode, r.stderr[-400:])
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_launchers_refuse_to_exec_rank_processes_under_a_profiler_preload():
    """a profiler preload (rocprofv3) initialises the GPU before main(); the fork + exec of rank processes from such a
    process takes the machine down on this pool.  bench.py --gpus N, marlin-hip-bench gpus=N and marlin-hip-run nranks=N all refuse
    (rc != 0, a message that says what to profile instead) BEFORE anything is started -- checked with the variables a preload sets,
    no profiler involved."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ROCP_TOOL_LIBRARIES"] = "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 2 and "profiler preload" in r.stderr, (r.returncode, r.stderr[-400:])
    for exe, argv in (("marlin-hip-bench", ["gpus=2", "steps=1", "warmup=0"]),
                      ("marlin-hip-run", ["problem=cahnhilliard", "dim=2", "nx=20", "ny=20", "parallel_mode=FFT_SLAB", "nranks=2"])):
        path = os.path.join(ROOT, "marlin_amd", "lib", exe)
        if not os.path.exists(path):
            pytest.skip(f"{exe} has not been built")
        r = subprocess.run([path] + argv, capture_output=True, text=True, timeout=60, env=env, cwd=ROOT)
        assert r.returncode != 0 and "profiler preload" in r.stderr, (exe, r.returncode, r.stderr[-400:])


def test_moose_shim_sources_only_call_what_the_abi_declares():
    """marlin_plugin/ (the MOOSE-side classes a Marlin maintainer compiles; INTEGRATION.md quotes them): every mrl_* function, MRL_*
    constant and mrl_ch_params / mrl_mech_params member they use is declared in include/marlin_hip.h (a static check beside
    the compiled one: tests/moose_stub builds these files, tests/test_moose_shim_gpu.py runs them)"""
    hdr = open(os.path.join(ROOT, "include", "marlin_hip.h")).read()
    plug = os.path.join(ROOT, "marlin_plugin")
    files = [os.path.join(d, f) for d, _, fs in os.walk(plug) for f in fs if f.endswith((".h", ".C"))]
    assert len(files) >= 5
    for path in files:
        src = open(path).read()
        code = re.sub(r"//[^\n]*", "", src)
        for fn in set(re.findall(r"\b(mrl_[a-z0-9_]+)\s*\(", code)):
            assert re.search(r"\b%s\s*\(" % fn, hdr), (os.path.basename(path), fn)
        for const in set(re.findall(r"\b(MRL_[A-Z0-9_]+)\b", code)):
            assert re.search(r"\b%s\b" % const, hdr), (os.path.basename(path), const)
        for member in set(re.findall(r"\b_p\.([a-z_]+)\b", code)) | set(re.findall(r"\b_prm\.([a-z_]+)\b", code)):
            assert re.search(r"\b%s\b[^;]*;" % member, hdr), (os.path.basename(path), member)
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for f in ("HipDomain.h", "HipAdamsBashforthMoulton.C", "HipFFTMechanics.C", "HipSpectralComputes.C", "HipSpectralOperators.C",
              "marlin_plugin.mk"):
        assert f in integ, f


def test_moose_shim_compiles_against_the_stub_and_registers_its_classes():
    """tests/moose_stub: the files of marlin_plugin/ compile unchanged against the MOOSE stand-ins + libTorch and link with
    libmarlin_hip.so (build() makes marlin_amd/lib/shim-driver); registerMooseObject has registered every class INTEGRATION.md names.
    Without a GPU the driver refuses to run a case (no CPU path behind the shim either)."""
    import subprocess
    exe = os.path.join(ROOT, "marlin_amd", "lib", "shim-driver")
    assert os.path.exists(exe), "shim-driver has not been built (python -c 'import __graft_entry__ as g; g.build()')"
    r = subprocess.run([exe, "case=types"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    types = set(r.stdout.split())
    want = {"HipAdamsBashforthMoulton", "HipAdamsBashforthMoultonCoupled", "HipFFTMechanics", "HipForwardFFT", "HipInverseFFT",
            "HipParsedCompute", "HipReciprocalLaplacianFactor", "HipReciprocalLaplacianSquareFactor", "HipFFTGradient",
            "HipFFTGradientSquare", "HipComputeDisplacements", "HipComputeVonMisesStress", "HipETDRK4Solver", "HipSecantSolver",
            "HipBroydenSolver", "HipForwardEulerSolver", "HipDeAliasingTensor", "HipSwiftHohenbergLinear", "HipFFTSemiImplicit",
            "HipReciprocalMatDiffusion", "HipReciprocalAllenCahn", "HipFFTQuasistaticElasticity", "HipFFTElasticChemicalPotential"}
    assert want <= types, want - types
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for t in want:
        assert t in integ, t
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, "case=cahnhilliard"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 1 and "needs a GPU" in r.stderr


def test_lds_conflict_model_reproduces_the_measured_shares():
    """tools/lds_conflict_model.py (the bank model behind LineMapParams, profiles/HISTORY.md 3.2): the old line map of the 512-point z kernels costs
    40 % conflict cycles (measured: 41-47 %), the adopted xor swizzle none; the 256-point plan was and stays conflict-free"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ldsmodel", os.path.join(ROOT, "tools", "lds_conflict_model.py"))
    src = open(spec.origin).read().split("PLANS = ")[0]      # definitions only (the rest of the file prints a report)
    ns = {}
    exec(compile(src, spec.origin, "exec"), ns)

    def total(N, P, rad, T, at):
        rows = ns["model"](N, P, rad, T, at)
        return sum(w + r for _, _, w, _, r, _ in rows), sum(wi + ri for _, _, _, wi, _, ri in rows)
    old512, ideal512 = total(512, 16, [8, 8, 8], 8, lambda p, l: l * 544 + p + (p >> 4))
    new512, _ = total(512, 16, [8, 8, 8], 8, lambda p, l: l * 512 + (p ^ ((p >> 3) & 7)))
    assert (old512, new512, ideal512) == (2560, 1536, 1536)
    old256, ideal256 = total(256, 16, [16, 16], 8, lambda p, l: l * 272 + p + (p >> 4))
    assert old256 == ideal256 == 384
    # the parameters in fft_pow2.h are the ones the model was run with
    hdr = open(os.path.join(ROOT, "marlin_amd", "csrc", "fft_pow2.h")).read()
    assert re.search(r"MRL_LINEMAP\(512, 0, 3, 7, 0\)", hdr) and re.search(r"MRL_LINEMAP\(200, 4, 3, 1, 0\)", hdr)


def test_python_constants_match_the_header():
    """the ctypes plumbing (marlin_amd/api.py) names options, flags and transports by value: every one of them against the #define /
    enum of include/marlin_hip.h, so that a renumbering in the header cannot silently change what a test or bench.py asks for"""
    import re
    from marlin_amd import api
    text = open(os.path.join(ROOT, "include", "marlin_hip.h")).read()
    values = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"#define\s+(MRL_[A-Z0-9_]+)\s+(-?(?:0x[0-9a-fA-F]+|\d+))\b", text)}
    values.update({m.group(1): int(m.group(2)) for m in re.finditer(r"\b(MRL_[A-Z0-9_]+)\s*=\s*(-?\d+)", text)})
    pairs = {"OPT_EXPERIMENT": "MRL_OPT_EXPERIMENT", "OPT_SLAB_NSUB": "MRL_OPT_SLAB_NSUB", "OPT_SLAB_CARRY": "MRL_OPT_SLAB_CARRY",
             "OPT_VERIFY_EXCHANGE": "MRL_OPT_VERIFY_EXCHANGE", "OPT_VERIFY_MISMATCHES": "MRL_OPT_VERIFY_MISMATCHES",
             "OPT_CACHE_CHUNK_MB": "MRL_OPT_CACHE_CHUNK_MB", "TRANSPORT_PEER_STORE": "MRL_TRANSPORT_PEER_STORE",
             "TRANSPORT_PEER_COPY": "MRL_TRANSPORT_PEER_COPY", "TRANSPORT_RCCL": "MRL_TRANSPORT_RCCL"}
    for py, c in pairs.items():
        assert c in values, c
        assert getattr(api, py) == values[c], (py, getattr(api, py), values[c])
    assert api.Context.COUPLED_L_AS_WRITTEN == values["MRL_COUPLED_L_AS_WRITTEN"]
    assert api.Context.COUPLED_COMPLEX_RHS == values["MRL_COUPLED_COMPLEX_RHS"]
    assert values["MRL_COUPLED_GENERAL"] == 4 and values["MRL_FLAG_PENCIL"] == 8
