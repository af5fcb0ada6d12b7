"""bench.py --gpus N starts its own rank processes: the launcher never touches the GPU, the ranks are native
C++ processes (marlin_amd/lib/marlin-hip-bench) that pick their device from the host-local rank as the reference does
(src/actions/DomainAction.C:163-199).  On a one-GPU box both ranks share device 0 over HIP IPC: a functional check of the whole
chain (launch, communicator, tuning phase, timed region, JSON), not a rate."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _last_json(out):
    for ln in reversed(out.strip().splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    raise AssertionError("no JSON line in:\n" + out[-2000:])


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def test_bench_two_ranks_without_a_launcher():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--grid", "64", "--steps", "6", "--warmup", "2", "--profile-steps", "2"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 2 and j["steps"] == 6 and j["warmup"] == 2 and j["value"] > 0
    assert j["config"]["grid"] == [64, 128, 64] and j["dtype"] == "f64"
    ex = j["exchange"]
    assert ex["ranks"] == 2 and len(ex["devices_per_rank"]) == 2
    # (two exchanges per substep and kz sub-block: the two-field forward one and the inverse one; the tuning picks 1, 2 or 4 sub-blocks)
    assert ex["transport"]["selected"] in ("peer_store", "peer_copy", "rccl") and ex["transport"]["nsub"] in (1, 2, 4)
    # exactly two exchanges per substep and kz sub-block (reference data flow: the two-field forward one and the inverse one) over the
    # timed regions: a window that dropped or double-counted an exchange would not give the integer
    assert ex["bytes_sent_to_peers_per_step_rank0"] > 0 and abs(ex["exchanges_per_step"] - 2 * ex["transport"]["nsub"]) <= 1e-9
    # every tuned candidate that ran agrees on the field checksum, and the consumers' system-scope re-reads found nothing stale
    ran = [c for c in ex["transport"]["tuned"] if "ms_per_step" in c]
    assert len(ran) >= 2 and all(c["checksum_agrees"] for c in ran)
    assert all(c.get("receive_buffer_reread_mismatches", 0) == 0 for c in ran)
    assert "exposed_wait_ms_per_step" in ex and "local_kernels_only" in j["variants"]
    assert j["launcher"]["ranks_run_as"].startswith("native C++ rank processes")
    assert ex["runtime"]["hip_runtime_version"] > 0 and "libamdhip64" in ex["runtime"]["hip_library"]
    # on one device the in-kernel-flag candidates are not tried, RCCL is reported as unavailable on this placement
    # (two ranks on one device) -- not as a failed ncclCommInitRank --, the tuning phase is bounded and says how long it took, and the
    # line carries the median protocol
    assert ex["distinct_devices"] == 1 and len(set(ex["physical_devices_per_rank"])) == 1
    assert not [c for c in ex["transport"]["tuned"] if c.get("flags") == "in-kernel flags"]
    un = [c for c in ex["transport"]["tuned"] if c.get("transport") == "rccl" and "unavailable" in c]
    assert un and "unavailable:" in un[0]["unavailable"] and "share device" in un[0]["unavailable"], ex["transport"]["tuned"]
    assert ex["runtime"]["rccl_status"].startswith("unavailable:") and ex["runtime"]["rccl_unique_id_hash"]
    assert 0.0 < ex["transport"]["tuning_s"] <= ex["transport"]["tune_budget_s"] + 30.0
    assert len(j["repeats_ms"]) == 7 and "MEDIAN" in j["timing_protocol"]
    # both denominators of the parallel efficiency measured inside the job, the rank-local cost and the exposed
    # waits at the top level (two ranks sharing ONE card: the numbers are a functional record, not a scaling result)
    pe = j["parallel_efficiency"]
    assert pe["weak"]["one_gpu"]["grid"] == [64, 64, 64] and pe["strong"]["one_gpu"]["grid"] == [64, 128, 64]
    for side in ("weak", "strong"):
        one = pe[side]["one_gpu"]
        assert one["ms_per_step"] > 0 and abs(pe[side]["efficiency"] - j["value"] / (2 * one["value"])) <= 1e-9 * pe[side]["efficiency"]
    assert j["local_kernels_only_ms"] > 0 and j["exposed_wait_ms_per_step"] >= 0
    assert "rccl_comm_nranks" in ex["runtime"]


def test_bench_two_ranks_under_torch_distributed_run():
    """the driver's launch form: rank 0 of the launcher job starts the native ranks, the other launcher rank waits for its verdict"""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29611", "bench.py", "--gpus", "2", "--grid", "64", "--steps", "4", "--warmup", "1", "--no-variants",
                        "--transport", "peer_store", "--profile-steps", "2"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 2 and j["exchange"]["transport"]["selected"] == "peer_store"
    assert "rank 0 of torch.distributed.run" in j["launcher"]["started_by"]


def test_bench_mech_two_ranks_without_a_launcher():
    r = subprocess.run([sys.executable, "bench.py", "--workload", "mech", "--gpus", "2", "--grid", "32", "--steps", "2"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 2 and j["config"]["grid"] == [32, 64, 32] and j["steps"] > 0
    assert j["exchange"]["ranks"] == 2 and j["exchange"]["bytes_sent_to_peers_per_step_rank0"] > 0
    ran = [c for c in j["exchange"]["transport"]["tuned"] if "ms_per_cg_iteration" in c]
    assert ran and max(c["norm_F"] for c in ran) - min(c["norm_F"] for c in ran) <= 1e-9 * ran[0]["norm_F"]
    ex = j["exchange"]   # the same placement / link / wait keys as the Cahn-Hilliard line
    assert len(ex["devices_per_rank"]) == 2 and ex["distinct_devices"] == 1
    assert ex["link_GBps_out_rank0"] > 0 and ex["exposed_wait_ms_per_step"] >= 0.0


def test_native_launcher_does_not_wait_for_the_survivors_of_a_lost_rank():
    """a rank that dies after the bootstrap leaves its peers in a collective (communicator time-out: 120 s): the launcher reaps in any
    order, gives the others a grace period, then ends exactly the processes it started and returns non-zero -- bench.py's fallback
    chain starts seconds later, not minutes"""
    import time
    exe = os.path.join(ROOT, "marlin_amd", "lib", "marlin-hip-bench")
    t0 = time.perf_counter()
    r = subprocess.run([exe, "workload=ch", "gpus=2", "device=0", "grid=64", "steps=2", "warmup=1", "test_die_rank=1", "launch_grace_s=3"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 1, (r.returncode, r.stderr[-2000:])
    assert time.perf_counter() - t0 < 60.0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_refuses_more_than_four_ranks_per_card():
    """--gpus 8 on a one-GPU box would put eight rank processes on one card: refused before anything is started (rc 2, no GPU call)"""
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs a one-GPU box")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "2"], cwd=ROOT, env=_env(), capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 2 and "rank processes per card" in r.stderr, (r.returncode, r.stderr[-500:])
