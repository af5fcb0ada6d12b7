"""The host half of the library-owned multi-GPU transport (marlin_amd/csrc/comm.hip) on CPU: P real processes attach to one
POSIX shared-memory bootstrap segment and run barriers, all-gathers (the channel that carries HIP IPC handles and the RCCL unique
id) and sum / min / max all-reduces, each checked inside the library against its closed form.  No GPU is touched."""
import multiprocessing as mp
import os

import pytest


def _rank(name, nranks, rank, rounds, q):
    from marlin_amd import _lib
    lib = _lib.load()
    q.put((rank, lib.mrl_comm_bootstrap_selftest(name.encode(), nranks, rank, rounds)))


@pytest.mark.parametrize("nranks", [2, 3, 8])
def test_bootstrap_collectives(nranks):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"mrl_selftest_{os.getpid()}_{nranks}"
    procs = [ctx.Process(target=_rank, args=(name, nranks, r, 50, q)) for r in range(nranks)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(nranks))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert res == [(r, 0) for r in range(nranks)]
    assert not os.path.exists(f"/dev/shm/{name}")      # rank 0 unlinked the segment


def test_bootstrap_rejects_bad_arguments():
    from marlin_amd import _lib
    lib = _lib.load()
    assert lib.mrl_comm_bootstrap_selftest(b"x", 0, 0, 1) == -1
    assert lib.mrl_comm_bootstrap_selftest(b"x", 2, 2, 1) == -1
    assert lib.mrl_comm_bootstrap_selftest(b"single", 1, 0, 3) == 0
