"""de Geus mechanics on a slab decomposition (BASELINE configs[4]) on ONE GPU: the P ranks are P threads, each with its
own context; the transposes and the all-reduce of the CG scalars go through an in-process loop-back communicator, so the
real SlabMechanics driver runs unchanged.  Parity: the reference's gold files (the serial result, abs_tol 1e-10)."""
import math
import threading

import numpy as np
import pytest
import torch

from tests.conftest import load_golden
from tests.test_oracle_golden import MECH_CASES, _mech_setup

pytestmark = pytest.mark.gpu


class ThreadComm:
    def __init__(self, P):
        self.P = P
        self.bar = threading.Barrier(P)
        self.slots = [None] * P

    def for_rank(self, r):
        return _RankComm(self, r)


class _RankComm:
    def __init__(self, hub, r):
        self.hub, self.r = hub, r

    def exchange(self, send_counts, recv_counts):
        return _RankExchange(self.hub, self.r, send_counts, recv_counts)

    def allreduce(self, values):
        h = self.hub
        h.slots[self.r] = list(values)
        h.bar.wait()
        out = [sum(h.slots[p][i] for p in range(h.P)) for i in range(len(values))]
        h.bar.wait()
        return out


class _RankExchange:
    def __init__(self, hub, r, sc, rc):
        self.hub, self.r = hub, r
        self.so = np.concatenate([[0], np.cumsum([2 * c for c in sc])])
        self.ro = np.concatenate([[0], np.cumsum([2 * c for c in rc])])

    def run(self, send, recv, async_op=False):
        h = self.hub
        h.slots[self.r] = (send, self.so)
        h.bar.wait()
        for p in range(h.P):
            s, so = h.slots[p]
            recv[self.ro[p]:self.ro[p + 1]].copy_(s[so[self.r]:so[self.r + 1]])
        h.bar.wait()
        return None


def _rank_main(r, P, hub, case, K, mu, out, errors):
    try:
        from marlin_amd.slab import SlabMechanics
        p = MECH_CASES[case]
        dim, n = p["dim"], p["n"]
        L = [2.0 * math.pi] * dim
        from marlin_amd.slab import HipSlabStages
        st = HipSlabStages(dim, [n] * dim, L, P, r)
        yb, nyl = st.real_begin[1], st.real_shape[1]
        Kl, mul = K[:, yb:yb + nyl].contiguous().cuda(), mu[:, yb:yb + nyl].contiguous().cuda()
        m = SlabMechanics(dim, [n] * dim, L, P, r, Kl, mul, comm=hub.for_rank(r), l_tol=p["l_tol"], l_max_its=p["l_max_its"] or 0,
                          nl_rel_tol=p["nl_rel"], nl_abs_tol=p["nl_abs"], stages=st)
        F = torch.eye(dim, dtype=torch.float64).expand(list(st.real_shape) + [dim, dim]).contiguous().cuda().reshape(-1)
        frames, traces = [], []
        t_old = 0.0
        for step in range(3):
            sub_dt = p["dt"] / p["substeps"]
            for s in range(p["substeps"]):
                t = t_old + s * sub_dt
                avg = m.average(F)                                      # MacroscopicShearTensor.C:31-41
                applied = torch.eye(dim, dtype=torch.float64)
                applied[0, 1] = applied[0, 1] + t
                applied = (applied - torch.tensor(avg, dtype=torch.float64).reshape(dim, dim)).cuda()
                F, P_, stats = m.newton_cg(F, applied)
                traces.append((stats["newton_its"], tuple(stats["cg_its"])))
            t_old += p["dt"]
            frames.append(F.cpu().reshape(list(st.real_shape) + [dim * dim]))
        out[r] = (yb, nyl, frames, traces)
    except Exception as e:      # surface the failure in the main thread and release the others
        errors.append(e)
        hub.bar.abort()


@pytest.mark.parametrize("case,P", [("mech3d", 2), ("mech3d", 4)])
def test_slab_mechanics_gold(case, P):
    p = MECH_CASES[case]
    dim, n = p["dim"], p["n"]
    g = load_golden(p["gold"])
    dom, phase, K, mu = _mech_setup(dim, n)
    hub = ThreadComm(P)
    out, errors = [None] * P, []
    threads = [threading.Thread(target=_rank_main, args=(r, P, hub, case, K, mu, out, errors)) for r in range(P)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    assert all(o is not None for o in out)
    assert len({o[3][0] for o in out}) == 1                      # every rank saw the same Newton / CG trace
    perm = (2, 1, 0) if dim == 3 else (1, 0)
    worst = 0.0
    for frame in range(3):
        F = torch.cat([o[2][frame] for o in out], dim=1)         # y-slabs back together
        for k in range(dim * dim):
            worst = max(worst, np.abs(g[f"F_{k}.{frame}"] - F[..., k].permute(*perm).numpy()).max())
    assert worst <= 1e-10, worst


def test_slab_mechanics_2d_full_spectrum():
    """2-D slab contexts transform c2c on both axes like the reference's FFT_SLAB mode (all axes fftfreq,
    DomainAction.C:279-281): the doubly-Nyquist mode of the projection then carries the opposite sign of the serial r2c
    run, so the 2-D slab solve is compared with the oracle's FFTMechanics on the FFT_SLAB axes, not with the serial gold."""
    from oracle import marlin_oracle as mo
    case, P = "mech2d", 2
    p = MECH_CASES[case]
    dim, n = p["dim"], p["n"]
    dom, phase, K, mu = _mech_setup(dim, n)
    hub = ThreadComm(P)
    out, errors = [None] * P, []
    threads = [threading.Thread(target=_rank_main, args=(r, P, hub, case, K, mu, out, errors)) for r in range(P)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    dom_c2c = mo.Domain(dim, [n] * dim, [2.0 * math.pi] * dim, slab_c2c=True)
    oracle = mo.FFTMechanicsOracle(dom_c2c, K, mu, l_tol=p["l_tol"], nl_rel_tol=p["nl_rel"], nl_abs_tol=p["nl_abs"],
                                   l_max_its=p["l_max_its"])
    F = torch.eye(dim, dtype=torch.float64).expand(dom.value_shape([dim, dim])).contiguous()
    t_old = 0.0
    for step in range(3):
        sub_dt = p["dt"] / p["substeps"]
        for s in range(p["substeps"]):
            F, _ = oracle.compute(F, mo.macroscopic_shear(dom_c2c, F, t_old + s * sub_dt))
        t_old += p["dt"]
        got = torch.cat([o[2][step] for o in out], dim=1)
        assert (got - F.reshape(got.shape)).abs().max().item() <= 1e-10
