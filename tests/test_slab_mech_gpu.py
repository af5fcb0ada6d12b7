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


def _run_threads(P, target, *args):
    hub = ThreadComm(P)
    out, errors = [None] * P, []

    def main(r):
        try:
            out[r] = target(r, P, hub.for_rank(r), *args)
        except Exception as e:      # surface the failure in the main thread and release the others
            errors.append(e)
            hub.bar.abort()

    threads = [threading.Thread(target=main, args=(r,)) for r in range(P)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    assert all(o is not None for o in out)
    return out


_FAST_SHAPE, _FAST_L = [64, 64, 64], [2.0 * math.pi, 3.0, 4.0]


def _gamma_rank(r, P, comm, A, fast):
    from marlin_amd.slab import HipSlabStages, SlabMechanics
    st = HipSlabStages(3, _FAST_SHAPE, _FAST_L, P, r)
    yb, nyl = st.real_begin[1], st.real_shape[1]
    one = torch.ones(st.real_shape, dtype=torch.float64, device="cuda")
    m = SlabMechanics(3, _FAST_SHAPE, _FAST_L, P, r, one, one, comm=comm, stages=st, fast=fast)
    assert m.fast == fast
    loc = A[:, yb:yb + nyl].contiguous().cuda().reshape(-1)
    out = torch.empty_like(loc)
    if fast:
        fm, ofm = torch.empty_like(loc), torch.empty_like(loc)
        m._relayout(True, loc, fm)
        m.gamma(fm, ofm, -2.0)
        m._relayout(False, ofm, out)
    else:
        m.gamma(loc, out, -2.0)
    return yb, nyl, out.cpu().reshape(list(st.real_shape) + [3, 3])


@pytest.mark.parametrize("P", [2, 4, 8, 32])
def test_slab_gamma_fused_rows(P):
    """the fused row pipeline (z+x passes -> exchange -> y pass with the projection in place -> exchange -> x+z passes) ==
    the oracle's closed-form Gamma operator on the global field == the per-component generic stages.  P = 32: two y planes per
    rank, fewer than the four threads of a line -- the table-addressed y pass (k_gamma_yfused_t) behind the same staged entry points"""
    from oracle import marlin_oracle as mo
    torch.manual_seed(21)
    A = torch.rand(_FAST_SHAPE + [3, 3], dtype=torch.float64) - 0.5
    dom = mo.Domain(3, _FAST_SHAPE, _FAST_L)
    want = -2.0 * mo.gamma_closed_form(dom, A)
    fast = _run_threads(P, _gamma_rank, A, True)
    got = torch.cat([o[2] for o in fast], dim=1)
    assert (got - want).abs().max().item() <= 1e-13
    if P == 2:
        slow = _run_threads(P, _gamma_rank, A, False)
        assert (torch.cat([o[2] for o in slow], dim=1) - got).abs().max().item() <= 1e-13


def _newton_rank(r, P, comm, K, mu, applied, tangent_fusion=True):
    from marlin_amd.slab import HipSlabStages, SlabMechanics
    st = HipSlabStages(3, _FAST_SHAPE, _FAST_L, P, r)
    yb, nyl = st.real_begin[1], st.real_shape[1]
    m = SlabMechanics(3, _FAST_SHAPE, _FAST_L, P, r, K[:, yb:yb + nyl].contiguous().cuda(), mu[:, yb:yb + nyl].contiguous().cuda(),
                      comm=comm, l_tol=1e-2, nl_rel_tol=2e-2, nl_abs_tol=2e-2, stages=st, tangent_fusion=tangent_fusion)
    assert m.fast
    assert bool(m.lib.mrl_slab_gamma_tangent_fusable(m.ctx.h))
    F = torch.eye(3, dtype=torch.float64).expand(list(st.real_shape) + [3, 3]).contiguous().cuda().reshape(-1)
    Fn, Pn, stats = m.newton_cg(F, applied.cuda())
    return yb, nyl, Fn.cpu().reshape(list(st.real_shape) + [3, 3]), Pn.cpu().reshape(list(st.real_shape) + [3, 3]), stats


@pytest.mark.parametrize("tangent_fusion", [True, False])
def test_slab_mechanics_fused_vs_serial(tangent_fusion):
    """one Newton-CG solve of the de Geus RVE at 64^3 on 4 loop-back ranks (fused field-major slab path; with the CG direction
    update + tangent fused into the forward z pass and the solution update deferred, mrl_slab_gamma_tangent_z_fwd, and with the
    separate kernels) == the serial fused solver of the same library: same Newton / CG iteration counts, F and P to 1e-10"""
    from marlin_amd.api import Context
    torch.manual_seed(2)
    n = 64
    phase = torch.zeros(_FAST_SHAPE, dtype=torch.float64)
    s = 9 * n // 32
    phase[-s:, :s, -s:] = 1.0                                # test/src/tensor_computes/PhaseMechanicsTest.C:36-45
    K = 0.833 + phase * (8.33 - 0.833)
    mu = 0.386 + phase * (3.86 - 0.386)
    applied = torch.zeros(3, 3, dtype=torch.float64)
    applied[0, 1] = 0.01
    out = _run_threads(4, _newton_rank, K, mu, applied, tangent_fusion)
    F = torch.cat([o[2] for o in out], dim=1)
    P_ = torch.cat([o[3] for o in out], dim=1)
    ctx = Context(3, _FAST_SHAPE, _FAST_L)
    F0 = torch.eye(3, dtype=torch.float64).expand(_FAST_SHAPE + [3, 3]).contiguous().cuda()
    Fs, Ps, st = ctx.mech_newton_cg(F0, K.cuda(), mu.cuda(), applied.cuda(), l_tol=1e-2, l_max_its=0, nl_rel_tol=2e-2, nl_abs_tol=2e-2)
    assert len({(o[4]["newton_its"], tuple(o[4]["cg_its"])) for o in out}) == 1
    assert out[0][4]["newton_its"] == st["newton_its"] and list(out[0][4]["cg_its"]) == list(st["cg_its"])
    assert (F - Fs.cpu()).abs().max().item() <= 1e-10
    assert (P_ - Ps.cpu()).abs().max().item() <= 1e-10
