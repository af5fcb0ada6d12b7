"""world_size-2/3 gloo runs (CPU) of marlin_amd.slab's host logic: exchange, buffer layouts of the
mrl_slab_* contract, history ring and the first-step-is-AB1 rule, against the serial oracle and the
reference's 2-rank gold file (test/tests/cahnhilliard/tests:58-70)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import marlin_oracle as mo
from tests.conftest import load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from marlin_amd.api import ch_params
        from marlin_amd.slab import SlabCahnHilliard
        from tests.slab_oracle_stages import OracleSlabStages

        dim, shape, L, c0 = case["dim"], case["shape"], case["L"], case["c0"]
        st = OracleSlabStages(dim, shape, L, world, rank)
        s = SlabCahnHilliard(dim, shape, L, ch_params(), world, rank, stages=st, nsub=case["nsub"],
                             carry=case.get("carry", False), predictor_order=case.get("pred", 2))
        yb, nyl = st.real_begin[1], st.real_shape[1]
        s.set_local(c0[:, yb:yb + nyl].contiguous())
        out = []
        for k in range(case["nsteps"]):
            s.step(case["dts"][k] if "dts" in case else case["dt"], case["substeps"])
            out.append(s.current().clone().numpy())
        q.put((rank, yb, nyl, out))
    finally:
        dist.destroy_process_group()


def _run(world, case):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res)


@pytest.mark.parametrize("carry", [False, True])
def test_slab_gold_rank1_two_ranks(carry):
    """2-rank FFT_SLAB Cahn-Hilliard: rank 1 must reproduce gold/cahnhilliard.rank0001.h5 to 1e-13 -- with the reference's
    data flow and with the spectral carry-over (c-hat = ubar of the previous substep, one field on the forward exchange)"""
    g = load_golden("cahnhilliard_rank0001_gold.npz")
    torch.manual_seed(0)
    blk = torch.rand(20, 10, dtype=torch.float64) * (0.56 - 0.44) + 0.44
    c0 = torch.cat([blk, blk], dim=1)           # every rank draws the same seed-0 block (RandomTensor.C:41-54)
    case = dict(dim=2, shape=[20, 20], L=[3.0, 3.0], c0=c0, nsteps=10, dt=1e-3, substeps=10, nsub=1, carry=carry)
    res = _run(2, case)
    rank, yb, nyl, states = res[1]
    assert (yb, nyl) == (10, 10)
    worst = max(np.abs(g[f"c.{k + 1}"] - states[k]).max() for k in range(10))
    assert worst <= 1e-13, worst


@pytest.mark.parametrize("world,shape,nsub,carry", [(2, [8, 6, 10], 3, False), (3, [9, 7, 5], 1, False), (2, [8, 6, 10], 2, True)])
def test_slab_3d_matches_serial_oracle(world, shape, nsub, carry):
    """3-D r2c slab run (uneven partitions, odd sizes) == serial oracle restricted to the rank's y-slab"""
    torch.manual_seed(4)
    c0 = torch.rand(shape, dtype=torch.float64) * 0.12 + 0.44
    L = [3.0, 2.0, 2.5]
    case = dict(dim=3, shape=shape, L=L, c0=c0, nsteps=2, dt=1e-3, substeps=3, nsub=nsub, carry=carry)
    res = _run(world, case)
    dom = mo.Domain(3, shape, L)
    ref = mo.CahnHilliardABM(dom, c0, M=0.2, kappa_factor=-0.001, mu_fn=mo.mu_double_well, substeps=3)
    for k in range(2):
        ref.step(1e-3)
        for rank, yb, nyl, states in res:
            assert np.abs(ref.c[:, yb:yb + nyl].numpy() - states[k]).max() <= 1e-13
    assert ref.order_log == [0, 0, 0, 1, 1, 1]


def test_slab_adaptive_dt_restarts_the_order():
    """a time step size that changes between steps restarts the Adams-Bashforth order for the first predictor_order - 1 substeps
    of the step (AdamsBashforthMoulton.C:75,88-91) while the history keeps advancing: AB3, 4 substeps, dt 1e-3, 1e-3, 2e-3, 2e-3,
    5e-4 on two gloo ranks == the serial oracle"""
    torch.manual_seed(8)
    shape, L = [8, 6, 10], [3.0, 2.0, 2.5]
    c0 = torch.rand(shape, dtype=torch.float64) * 0.12 + 0.44
    dts = [1e-3, 1e-3, 2e-3, 2e-3, 5e-4]
    case = dict(dim=3, shape=shape, L=L, c0=c0, nsteps=len(dts), dts=dts, substeps=4, nsub=2, pred=3)
    res = _run(2, case)
    dom = mo.Domain(3, shape, L)
    ref = mo.CahnHilliardABM(dom, c0, M=0.2, kappa_factor=-0.001, mu_fn=mo.mu_double_well, substeps=4, predictor_order=3)
    for k, dt in enumerate(dts):
        ref.step(dt)
        for rank, yb, nyl, states in res:
            assert np.abs(ref.c[:, yb:yb + nyl].numpy() - states[k]).max() <= 1e-13
    # step 1: no history (AB1); step 2: history grows 1, 2, 2, 2; steps 3 and 5: dt changed -> two first-order substeps, then AB3
    assert ref.order_log == [0, 0, 0, 0, 1, 2, 2, 2, 0, 0, 2, 2, 2, 2, 2, 2, 0, 0, 2, 2]
