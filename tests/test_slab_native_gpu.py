"""The library-owned multi-GPU path (include/marlin_hip.h: mrl_comm_*, mrl_ctx_attach_comm) with REAL rank processes: every
test starts P child processes (tests/slab_rank_worker.py), all on GPU 0 of this box, each with its own slab context and
communicator.  They exchange through HIP IPC peer mappings and device-side flags exactly as P ranks on P GPUs do (RCCL cannot
be exercised this way: it refuses several ranks on one device).  Parity: the serial oracle on the global field and the
reference's 2-rank gold file (test/tests/cahnhilliard/gold/cahnhilliard.rank0001.h5, abs 1e-13)."""
import itertools
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "slab_rank_worker.py")
_counter = itertools.count()
EXTRA_KV = [x for x in os.environ.get("MRL_TEST_EXTRA_KV", "").split() if x]   # debugging aid, e.g. exp=1048576 (host-side trace)


def run_job(P, case, *kv, timeout=240):
    """start P rank processes, return their RESULT records in rank order; any failure shows every rank's stderr"""
    job = f"mrltest_{os.getpid()}_{next(_counter)}"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if os.environ.get("MRL_TEST_HIP_LOG"):   # debugging aid: the HIP runtime's own call log in every rank's stderr
        env["AMD_LOG_LEVEL"] = os.environ["MRL_TEST_HIP_LOG"]
    procs = [subprocess.Popen([sys.executable, WORKER, job, str(P), str(r), case, *kv], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, env=env, cwd=ROOT) for r in range(P)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    except subprocess.TimeoutExpired:
        for p in procs:
            if p.poll() is None:
                p.kill()
        tails = []
        for r, p in enumerate(procs):   # what every rank had printed when the job was stopped
            so, se = p.communicate()
            tails.append(f"--- rank {r} (rc {p.returncode}) ---\n{(se or '')[-1500:]}\n{(so or '')[-300:]}")
        raise AssertionError(f"job {case} {kv} timed out after {timeout} s\n" + "\n".join(tails))
    finally:
        shm = f"/dev/shm/{job}"
        if os.path.exists(shm):
            os.unlink(shm)
    res = []
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        line = [ln for ln in so.splitlines() if ln.startswith("RESULT ")]
        assert p.returncode == 0 and line, f"rank {r} failed (rc {p.returncode}):\n{se[-3000:]}\n{so[-1000:]}"
        res.append(json.loads(line[-1][7:]))
    return res


@pytest.mark.parametrize("P,transport", [(2, 1), (2, 2), (4, 1)])
def test_native_slab_ch_gold_rank1(P, transport):
    """test/tests/cahnhilliard/tests:58-70 (cahnhilliard.i, parallel_mode = FFT_SLAB): c.1 .. c.10 of rank 1's gold file, by P
    processes through mrl_ch_substeps with peer stores (1) and copy-engine pushes (2)"""
    res = run_job(P, "chgold", f"transport={transport}")
    assert all(r["transport"] == transport for r in res)
    gold = [r["max_gold_err"] for r in res if "max_gold_err" in r]
    assert len(gold) == P // 2 and max(gold) <= 1e-13, res
    assert max(r["max_err"] for r in res) <= 1e-13, res


@pytest.mark.parametrize("P,shape,transport,nsub,carry", [
    (2, "64,64,64", 1, 1, 0), (2, "64,64,64", 2, 2, 0), (4, "64,128,64", 1, 1, 0), (4, "64,64,64", 1, 3, 1), (4, "128,64,64", 2, 1, 1),
    (2, "8,6,10", 1, 1, 0), (3, "9,7,5", 2, 2, 0), (3, "9,7,5", 1, 1, 1), (2, "16,12", 1, 1, 0)])
def test_native_slab_ch_vs_oracle(P, shape, transport, nsub, carry):
    """planned shapes (fused kernels scattering into the peers' buffers) and odd / uneven / 2-D shapes (generic stages + pushes):
    two time steps of three substeps (AB1 then AB2) against the serial oracle"""
    res = run_job(P, "ch", f"shape={shape}", f"transport={transport}", f"nsub={nsub}", f"carry={carry}")
    assert max(r["max_err"] for r in res) <= 1e-13, res
    if P > 1:
        assert all(r["stats"]["exchanges"] > 0 and r["stats"]["bytes_sent"] > 0 for r in res)


@pytest.mark.parametrize("P,shape,transport,nsub,carry,exp", [
    (3, "64,64,64", 1, 1, 0, 0),        # test/tests/tensor_compute/parallel_roundtrip_3d.i's grid on 3 ranks: 22 / 21 / 21 planes
    (3, "64,64,64", 2, 2, 1, 0),        # ... copy-engine pushes, two kz sub-blocks, spectral carry-over
    (2, "200,200,200", 1, 1, 0, 0),     # examples/cahn_hilliard/cahnhilliard2.i:7-13 (200^3) on 2 and 4 ranks: ny / P = 100, 50
    (4, "200,200,200", 1, 2, 0, 0),
    (2, "64,128,64", 1, 1, 0, 1 << 24)])   # a shift-addressable shape through the table-addressed kernels
def test_native_slab_table_addressed_pipeline(P, shape, transport, nsub, carry, exp):
    """partitions that are not equal powers of two take the FUSED slab pipeline (table-addressed chunks:
    k_pass_sub_t, k_ch_yfused_t scattering into the peers' buffers), not the generic stages: the profile slots of the fused passes
    are present on every rank, fields vs the serial oracle to 1e-13"""
    res = run_job(P, "ch", f"shape={shape}", f"transport={transport}", f"nsub={nsub}", f"carry={carry}", f"exp={exp}", "verify=1", timeout=600)
    assert max(r["max_err"] for r in res) <= 1e-13, res
    for r in res:
        # MRL_OPT_VERIFY_EXCHANGE was on: the re-read kernel ran behind every arrival wait and found no word that a system-scope load
        # sees differently from a plain one
        assert "slab_exchange_verify" in r["kernels"] and r["verify_mismatches"] == 0, r
        assert {"slab_A_x_fwd", "slab_B_y_fused", "slab_C_x_inv"} <= set(r["kernels"]), r["kernels"]
        assert not {"slab_x_fwd", "slab_y_fwd", "slab_pack"} & set(r["kernels"]), r["kernels"]
        assert r["stats"]["exchanges"] > 0 and r["stats"]["bytes_sent"] > 0


@pytest.mark.parametrize("P,shape,transport", [(2, "16,12,10", 1), (3, "9,7,5", 2), (4, "64,64,64", 1), (2, "16,12", 2),
                                               (3, "64,64,64", 1)])   # (test/tests/tensor_compute/parallel_roundtrip_3d.i: 64^3 on 3 ranks, 22 / 21 / 21 planes)
def test_native_slab_fft(P, shape, transport):
    """DomainAction::fft / ifft in FFT_SLAB mode through mrl_fft_r2c / mrl_fft_c2r with the library-owned exchange (repeated
    forward transforms: the acknowledgement flags), and a global reduction"""
    res = run_job(P, "fft", f"shape={shape}", f"transport={transport}")
    assert max(r["max_err"] for r in res) <= 1e-13, res


def test_native_single_rank_all_transports():
    """one rank: the same pipeline with self-exchanges, including the RCCL transport (grouped send/recv to itself)"""
    for transport in (1, 2, 3):
        res = run_job(1, "ch", "shape=64,64,64", f"transport={transport}")
        assert res[0]["transport"] == transport and res[0]["max_err"] <= 1e-13, res


@pytest.mark.parametrize("P,transport", [(2, 1), (2, 2)])
def test_native_slab_mechanics_gold(P, transport):
    """test/tests/mechanics/tests:2-21 (mech3d.i, 16^3, 3 steps x 10 substeps) on P rank processes through mrl_mech_newton_cg:
    F_k of mech3d.h5 to 1e-10, and the serial oracle's Newton / CG iteration counts on every rank (generic stages: 16 is unplanned)"""
    res = run_job(P, "mech", "gold=1", f"transport={transport}", timeout=600)
    assert all(r["traces_ok"] for r in res), res
    assert max(r["max_gold_err"] for r in res) <= 1e-10, res
    assert max(r["max_err"] for r in res) <= 1e-10, res


@pytest.mark.parametrize("P,shape,transport", [(2, "32,32,32", 1), (4, "32,64,32", 1), (4, "64,32,32", 2)])
def test_native_slab_mechanics_fused_vs_oracle(P, shape, transport):
    """planned shapes: the fused field-major row pipeline (peer stores from the x / y pass kernels, CG direction + tangent + forward
    z pass in one kernel, CG scalars all-reduced on the device) against the ORACLE's serial Newton-CG: same iteration counts,
    F to 1e-10"""
    res = run_job(P, "mech", f"shape={shape}", f"transport={transport}", timeout=600)
    assert all(r["traces_ok"] for r in res), res
    assert max(r["max_err"] for r in res) <= 1e-10, res


@pytest.mark.parametrize("P,shape,transport,exp", [
    (3, "64,32,32", 1, 0),       # 22 / 21 / 21 x planes, 11 / 11 / 10 y rows: all rows per launch, peer stores
    (3, "32,64,32", 2, 0),       # ... copy-engine pushes: the row pipeline (one exchange per tensor row)
    (2, "40,40,40", 1, 0),       # planned non-power-of-two extents, ny / P = 20
    (2, "32,32,32", 1, 1 << 24)])  # a shift-addressable shape through the table-addressed kernels
def test_native_slab_mechanics_table_addressed(P, shape, transport, exp):
    """VERDICT r02 item 4 (mechanics): partitions that are not equal powers of two run the Newton-CG solve on field-major vectors through
    the FUSED Gamma pipeline with table-addressed chunks (k_pass_sub_mft, k_gamma_yfused_t), not the generic value-major stages: the
    fused passes' profile slots are present and the generic ones absent on every rank; the oracle's iteration counts, F to 1e-10"""
    res = run_job(P, "mech", f"shape={shape}", f"transport={transport}", f"exp={exp}", timeout=600)
    assert all(r["traces_ok"] for r in res), res
    assert max(r["max_err"] for r in res) <= 1e-10, res
    for r in res:
        assert {"slab_gamma_x_fwd", "slab_gamma_y_fused", "slab_gamma_x_inv"} <= set(r["kernels"]), r["kernels"]
        assert not {"slab_x_fwd", "slab_y_fwd", "slab_pack", "gamma_project_fm", "mech_gamma_project"} & set(r["kernels"]), r["kernels"]


def test_native_slab_mechanics_config_e_at_size_vs_the_serial_hip_solver():
    """BASELINE configs[4] at its size: the de Geus RVE on 256^3 (inclusion of bench.py --workload mech) over 4 rank processes -- 256 x
    64 x 256 per rank, the all-rows-per-launch Gamma pipeline over one nine-field exchange (peer stores), CG scalars all-reduced on the
    device -- against the serial HIP solver on the whole grid (oracle-checked at 128^3 in tests/test_fullsize_gpu.py; the oracle needs
    minutes per solve here): identical Newton / CG iteration counts on every rank, F to 1e-10"""
    res = run_job(4, "mech", "shape=256,256,256", "transport=1", "ref=hip", timeout=900)
    assert all(r["traces_ok"] for r in res), res
    assert max(r["max_err"] for r in res) <= 1e-10, res
    assert sum(res[0]["cg_its_last"]) >= 20, res


@pytest.mark.parametrize("P,carry", [(4, 0), (2, 1)])
def test_native_bench_configuration_equals_serial(P, carry):
    """the configuration `bench.py --gpus P` runs per rank with the native driver (grid_for(P, 256): 512 x 512 x 256 on 4 ranks, the
    bench's initial condition, one mrl_ch_substeps call) on P rank processes against the serial fused path on the same global grid:
    4 substeps, every rank's slab to 1e-13, total mass conserved"""
    res = run_job(P, "chbench", f"carry={carry}", timeout=600)
    assert max(r["max_err"] for r in res) <= 1e-13, res
    assert max(r["mass_err"] for r in res) <= 1e-12, res


def test_native_slab_exchange_buffers_beyond_4_gib():
    """rank-local arrays of 2.2 GB: the two-field forward exchange buffer (4.36 GB) is beyond 32-bit byte offsets, which sent such
    grids (1024^3 on 2 or 4 GPUs) to the any-size stages.  The y pass's 64-bit variant keeps them on the fused pipeline: a one-rank
    slab job (communicator, flags, exchange tables, peer-store kernels) against the serial fused path on the same grid (itself
    oracle-checked at the sizes the oracle reaches), 2 substeps, to 1e-13, mass conserved.  (Two rank processes with buffers of this size: covered through the C++ driver,
    tests/test_host_driver_gpu.py::test_cahnhilliard_fft_slab_exchange_buffers_beyond_4_gib -- inside a PyTorch process the HIP runtime
    bundled with the wheel does not return from hipIpcOpenMemHandle for multi-GB buffers, profiles/HISTORY.md 4.1.)"""
    res = run_job(1, "chbench", "shape=512,512,1024", "steps=2", "ic=rand", *EXTRA_KV, timeout=300)
    assert max(r["max_err"] for r in res) <= 1e-13, res
    assert max(r["mass_err"] for r in res) <= 1e-12, res


def test_native_lost_peer_times_out_cleanly():
    """a rank that leaves the job: the others' next solver call returns MRL_ERR_COMM (-6) once the bounded device-side wait gives up
    (4 s here), and later waits return at once -- no hung GPU, no hung host"""
    res = run_job(3, "lost_peer", "timeout=4", timeout=180)
    stayed = [r for r in res if not r["left"]]
    assert len(stayed) == 2
    for r in stayed:
        assert r["code"] == -6 and "did not arrive" in r["message"], r
        assert 3.0 <= r["seconds"] <= 30.0, r


def test_bench_drivers_agree_on_two_ranks():
    """`bench.py --gpus 2` with the native driver (library-owned exchange, transports tuned in the warm-up) and with the
    torch.distributed driver (marlin_amd/slab.py; gloo here, because two ranks share this box's GPU) evolve the same field: the
    global checksum sum(c^2) of their JSON lines agrees to 1e-13 relative, and the native line shows that bytes crossed ranks"""
    import socket

    def free_port():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        return port

    def run(extra):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "64", "--steps", "6", "--warmup",
               "2", "--profile-steps", "2", "--no-variants"] + extra
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        assert out.returncode == 0, out.stderr[-3000:]
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert line, out.stdout[-1000:]
        return json.loads(line[-1])

    native = run([])
    python = run(["--driver", "python", "--backend", "gloo"])
    a, b = native["field_checksum"]["sum_c_squared"], python["field_checksum"]["sum_c_squared"]
    assert abs(a - b) <= 1e-13 * abs(a), (a, b)
    assert native["config"]["driver"].startswith("native") and python["config"]["driver"] == "python"
    assert native["exchange"]["bytes_sent_to_peers_per_step_rank0"] > 0
    assert native["exchange"]["transport"]["selected"] in ("peer_store", "peer_copy")
    assert native["n_gpus"] == 2 and native["config"]["grid"] == [64, 128, 64]


@pytest.mark.parametrize("P,shape,transport", [(4, "16,12,10", 1), (4, "40,40,40", 2), (4, "64,64,64", 1), (4, "9,8,7", 1), (4, "7,9,11", 2)])
def test_native_pencil_transforms_vs_oracle(P, shape, transport):
    """parallel_mode = FFT_PENCIL (the last SURVEY 8(f) item; DomainAction.C:568-742, 1021-1047, 1105-1404) on P = py x pz rank
    PROCESSES sharing this box's GPU: mrl_fft_r2c / mrl_fft_c2r with the four staged exchanges owned by the library against the oracle's
    restatement of the reference's stages and the serial transform of the global array (2e-15 x n relative), the round trip (1e-14,
    64^3 included), an inverse transform of a non-Hermitian spectrum (irfft semantics), block shapes / begins / reciprocal axes of
    partitionPencils bit for bit, global reductions, MRL_ERR_UNSUPPORTED from the mechanics entry points, and four Cahn-Hilliard
    substeps (mrl_ch_substeps x 3 + mrl_ch_substep: the operator sequence over the staged transforms) against the oracle's serial
    solution of the global field, 1e-13.  (At most four rank
    processes: this process holds the GPU too and a box admits six; the 2 x 3 and 2 x 4 process grids are covered by the oracle-level
    CPU tests, tests/test_pencil_cpu.py.)"""
    res = run_job(P, "pencil", f"shape={shape}", f"transport={transport}")
    n = max(int(x) for x in shape.split(","))
    assert all(r["layout_ok"] and r["axes_ok"] and r["refused"] for r in res), res
    assert max(r["max_err"] for r in res) <= 2e-15 * n, res
    assert max(r["ch_err"] for r in res) <= 1e-13, res
    assert all(r["stats"]["exchanges"] > 0 and r["stats"]["bytes_sent"] > 0 for r in res)


def test_rccl_bring_up_is_exercised_up_to_comm_init_on_one_gpu():
    """RCCL with N > 1 ranks cannot run on a one-GPU box (it refuses two ranks on one device), but everything up
    to ncclCommInitRank can: 2 and 4 rank processes load the library, hold the SAME unique id after the bootstrap broadcast (hash in
    mrl_comm_describe), see the shared device in the placement census, and report RCCL as *unavailable* (MRL_ERR_UNSUPPORTED, status
    text) -- not as a failed ncclCommInitRank; the host collectives keep working afterwards.  src/actions/DomainAction.C:163-199
    (one rank <-> one device) is the placement rule RCCL insists on."""
    for P in (2, 4):
        res = run_job(P, "rccl_preflight")
        ids = {r["describe"]["rccl_unique_id_hash"] for r in res}
        assert len(ids) == 1 and ids != {""}, res
        for r in res:
            d = r["describe"]
            assert r["rc"] == -2 and not r["switched"] and r["switch_code"] == -2, r
            assert d["rccl_loaded"] and d["rccl_status"].startswith("unavailable:") and "share device" in d["rccl_status"], d
            assert d["distinct_devices"] == 1 and len(d["devices_per_rank"]) == P and len(set(d["devices_per_rank"])) == 1, d
            assert d["rccl_comm_nranks"] == -1
            assert r["allreduce"] == [P * (P + 1) / 2.0]


