"""HIP slab stages on ONE GPU: the P ranks of a slab decomposition are P contexts in this process and
the global transpose is a loop-back copy between their send/recv buffers (the RCCL exchange itself is
covered over gloo in test_slab_gloo.py).  Parity against the serial oracle on the global field and the
reference's 2-rank gold file."""
import math

import numpy as np
import pytest
import torch

from oracle import marlin_oracle as mo
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


class Loopback:
    """all-to-all between P in-process ranks: recv[q] chunk p = send[p] chunk q"""

    def __init__(self):
        self.members = []   # (send_counts, recv_counts)

    def factory(self, send_counts, recv_counts):
        x = _LoopX(self, len(self.members), send_counts, recv_counts)
        self.members.append(x)
        return x


class _LoopX:
    def __init__(self, hub, idx, sc, rc):
        self.hub, self.idx = hub, idx
        self.sc, self.rc = [2 * c for c in sc], [2 * c for c in rc]

    def run(self, send, recv, async_op=False):
        raise RuntimeError("loop-back ranks are driven phase by phase")


def _a2a(xs, sends, recvs):
    P = len(xs)
    soff = [np.concatenate([[0], np.cumsum(x.sc)]) for x in xs]
    roff = [np.concatenate([[0], np.cumsum(x.rc)]) for x in xs]
    for q in range(P):
        for p in range(P):
            assert xs[p].sc[q] == xs[q].rc[p]
            recvs[q][roff[q][p]:roff[q][p + 1]].copy_(sends[p][soff[p][q]:soff[p][q + 1]])


def _make(dim, shape, L, P, **kw):
    from marlin_amd.api import ch_params
    from marlin_amd.slab import SlabCahnHilliard
    hub = Loopback()
    return [SlabCahnHilliard(dim, shape, L, ch_params(), P, r, exchange_factory=hub.factory, **kw) for r in range(P)]


def _substep_all(solvers):
    S = range(solvers[0].nsub)
    for s in solvers:
        s.phase_z()
    for k in S:
        for s in solvers:
            s.phase_a(k)
        ex = [s.fwd_exchange(k) for s in solvers]     # (exchange, send view, recv view): one field only with the carry-over
        _a2a([e[0] for e in ex], [e[1] for e in ex], [e[2] for e in ex])
    for k in S:
        for s in solvers:
            s.phase_b(k)
        _a2a([s.x_inv[k] for s in solvers], [s.send_i[k] for s in solvers], [s.recv_i[k] for s in solvers])
    for k in S:
        for s in solvers:
            s.phase_c(k)
    for s in solvers:
        s.phase_e()


def _run_all(solvers, count, advance=True):
    """SlabCahnHilliard.run() for all loop-back ranks in lock step: the fused z passes between the substeps"""
    for k in range(count):
        for s in solvers:
            if k == 0:
                s.phase_z()
            else:
                s.phase_ez(advance)
        S = range(solvers[0].nsub)
        for j in S:
            for s in solvers:
                s.phase_a(j)
            ex = [s.fwd_exchange(j) for s in solvers]
            _a2a([e[0] for e in ex], [e[1] for e in ex], [e[2] for e in ex])
        for j in S:
            for s in solvers:
                s.phase_b(j)
            _a2a([s.x_inv[j] for s in solvers], [s.send_i[j] for s in solvers], [s.recv_i[j] for s in solvers])
        for j in S:
            for s in solvers:
                s.phase_c(j)
    for s in solvers:
        s.phase_e()


def _step_all(solvers, dt, substeps):
    for s in solvers:
        s.time_step += 1
        if s.time_step > 1:
            s.advance_state()
        s.sub_dt = dt / substeps
    for k in range(substeps):
        _substep_all(solvers)
        for s in solvers:
            if k < substeps - 1 and s.time_step > 1:
                s.advance_state()


def _gather(solvers):
    return torch.cat([s.current().cpu() for s in solvers], dim=1)


@pytest.mark.parametrize("shape,P,spectrum", [((8, 6, 10), 2, 0), ((9, 7, 5), 3, 0), ((16, 12), 2, 1), ((20, 20), 4, 1),
                                               ((8, 6, 4), 2, 1), ((64, 64, 64), 4, 0)])
def test_slab_fft_roundtrip_and_spectrum(shape, P, spectrum):
    """mrl_slab_fwd_* == rows [xb:xe] of the serial transform; inverse stages bring the field back"""
    from marlin_amd.slab import HipSlabStages
    dim = len(shape)
    L = [1.0 + 0.5 * d for d in range(dim)]
    torch.manual_seed(11)
    a = torch.rand(shape, dtype=torch.float64)
    full = torch.fft.fftn(a) if spectrum == 1 else torch.fft.rfftn(a)
    sts = [HipSlabStages(dim, list(shape), L, P, r, spectrum=spectrum) for r in range(P)]
    cnt_f = [st.counts(True) for st in sts]
    cnt_b = [st.counts(False) for st in sts]

    class X:
        def __init__(self, sc, rc):
            self.sc, self.rc = [2 * c for c in sc], [2 * c for c in rc]
    sends, recvs, specs = [], [], []
    for st, (sc, rc) in zip(sts, cnt_f):
        yb, nyl = st.real_begin[1], st.real_shape[1]
        loc = a[:, yb:yb + nyl].contiguous().cuda()
        send = st.empty(2 * sum(sc))
        st.fwd_local(loc, send)
        sends.append(send)
        recvs.append(st.empty(2 * sum(rc)))
    _a2a([X(*c) for c in cnt_f], sends, recvs)
    for st, recv in zip(sts, recvs):
        nspec = int(np.prod(st.recip_shape))
        spec = st.empty(2 * nspec)
        st.fwd_finish(recv, spec)
        xb, nxl = st.recip_begin[0], st.recip_shape[0]
        got = torch.view_as_complex(spec.cpu().reshape(-1, 2)).reshape(st.recip_shape)
        assert (got - full[xb:xb + nxl]).abs().max().item() <= 2e-15 * full.abs().max().item() * max(shape)
        specs.append(spec)
    sends, recvs = [], []
    for st, spec, (sc, rc) in zip(sts, specs, cnt_b):
        send = st.empty(2 * sum(sc))
        st.inv_local(spec, send)
        sends.append(send)
        recvs.append(st.empty(2 * sum(rc)))
    _a2a([X(*c) for c in cnt_b], sends, recvs)
    for st, recv in zip(sts, recvs):
        yb, nyl = st.real_begin[1], st.real_shape[1]
        out = st.empty(int(np.prod(st.real_shape)))
        st.inv_finish(recv, out)
        assert (out.cpu().reshape(st.real_shape) - a[:, yb:yb + nyl]).abs().max().item() <= 1e-13


@pytest.mark.parametrize("carry", [False, True])
def test_slab_ch_gold_rank1(carry):
    """test/tests/cahnhilliard/tests:58-70: rank 1 of the 2-rank FFT_SLAB run, c.1..c.10 to 1e-13 (also with the spectral
    carry-over: c-hat = ubar of the previous substep)"""
    g = load_golden("cahnhilliard_rank0001_gold.npz")
    torch.manual_seed(0)
    blk = torch.rand(20, 10, dtype=torch.float64) * (0.56 - 0.44) + 0.44
    solvers = _make(2, [20, 20], [3.0, 3.0], 2, nsub=1, carry=carry)
    for s in solvers:
        s.set_local(blk.cuda())
    worst = 0.0
    for k in range(10):
        _step_all(solvers, 1e-3, 10)
        worst = max(worst, np.abs(g[f"c.{k + 1}"] - solvers[1].current().cpu().numpy()).max())
    assert worst <= 1e-13, worst


@pytest.mark.parametrize("shape,P,nsub", [((8, 6, 10), 2, 1), ((9, 7, 5), 3, 2), ((32, 32, 32), 4, 3), ((64, 64, 64), 2, 1),
                                          ((64, 64, 64), 2, 4), ((64, 128, 64), 8, 5), ((128, 32, 32), 4, 2),
                                          ((32, 64, 32), 32, 1)])    # 2 rows per chunk: the y pass without uniform chunk offsets
def test_slab_ch_matches_serial_oracle(shape, P, nsub):
    torch.manual_seed(4)
    c0 = torch.rand(shape, dtype=torch.float64) * 0.12 + 0.44
    L = [3.0, 2.0, 2.5]
    solvers = _make(3, list(shape), L, P, nsub=nsub)
    for s in solvers:
        yb, nyl = s.st.real_begin[1], s.st.real_shape[1]
        s.set_local(c0[:, yb:yb + nyl].contiguous().cuda())
    dom = mo.Domain(3, list(shape), L)
    ref = mo.CahnHilliardABM(dom, c0, M=0.2, kappa_factor=-0.001, mu_fn=mo.mu_double_well, substeps=3)
    for k in range(2):
        ref.step(1e-3)
        _step_all(solvers, 1e-3, 3)
        assert (_gather(solvers) - ref.c).abs().max().item() <= 1e-13
    assert [s.last_order for s in solvers] == [1] * P


@pytest.mark.parametrize("shape,P,nsub,exp", [((64, 64, 64), 3, 2, 0),       # the reference's 3-rank 64^3 test: 22 / 21 / 21 planes
                                              ((100, 200, 50), 4, 1, 0),     # radix-10 family, ny / P = 50, nx / P = 25
                                              ((200, 100, 40), 2, 3, 0),
                                              ((96, 48, 64), 5, 1, 0),       # uneven on both axes: 20/19/19/19/19 and 10/10/10/9/9
                                              ((64, 128, 64), 4, 2, 1 << 24),  # a shift-addressable shape through the table kernels
                                              ((512, 48, 32), 3, 2, 0)])     # 512-point x lines on 171 / 171 / 170 planes: the wide plan, table-addressed
def test_slab_table_addressed_pipeline(shape, P, nsub, exp):
    """partitions that are not equal powers of two (the reference's 200^3 example grid on 2 / 4 ranks, its 3-rank
    64^3 test, device_weights) run the FUSED slab pipeline with table-addressed chunks (k_pass_sub_t, k_ch_yfused_t) instead of the
    generic stages: fields vs the serial oracle to 1e-13 with and without the spectral carry-over, and the profile shows the fused
    y pass"""
    torch.manual_seed(4)
    c0 = torch.rand(shape, dtype=torch.float64) * 0.12 + 0.44
    L = [3.0, 2.0, 2.5]
    plain = _make(3, list(shape), L, P, nsub=nsub, exp=exp)
    carry = _make(3, list(shape), L, P, nsub=nsub, carry=True, exp=exp)
    assert int(plain[0].ctx.lib.mrl_slab_ch_spec_pitch(plain[0].ctx.h)) % 8 == 0     # the planned pipeline's padded rows
    for s in plain + carry:
        yb, nyl = s.st.real_begin[1], s.st.real_shape[1]
        s.set_local(c0[:, yb:yb + nyl].contiguous().cuda())
    plain[0].ctx.set_profiling(True)
    dom = mo.Domain(3, list(shape), L)
    ref = mo.CahnHilliardABM(dom, c0, M=0.2, kappa_factor=-0.001, mu_fn=mo.mu_double_well, substeps=3)
    for k in range(2):
        ref.step(1e-3)
        _step_all(plain, 1e-3, 3)
        _step_all(carry, 1e-3, 3)
        assert (_gather(plain) - ref.c).abs().max().item() <= 1e-13
        assert (_gather(carry) - ref.c).abs().max().item() <= 1e-13
    names = {k["kernel"] for k in plain[0].ctx.get_profile() if k["launches"]}
    assert {"slab_A_x_fwd", "slab_B_y_fused", "slab_C_x_inv"} <= names, names


@pytest.mark.parametrize("shape,P,nsub", [((8, 6, 10), 2, 2), ((9, 7, 5), 3, 1), ((64, 64, 64), 2, 4), ((64, 128, 64), 8, 3),
                                          ((128, 32, 32), 4, 2), ((32, 64, 32), 32, 2),
                                          ((40, 48, 240), 2, 1), ((240, 40, 48), 2, 2), ((48, 240, 40), 3, 1), ((160, 120, 150), 2, 1)])  # 2 x 3 x 5 lengths
def test_slab_ch_carry_over(shape, P, nsub):
    """spectral carry-over (MRL_CARRY_OUT on the first substep, MRL_CARRY_IN afterwards) vs the reference's data flow on
    the same kernels and vs the serial oracle: generic path (odd / uneven) and fused fast path, AB1 -> AB2 history"""
    torch.manual_seed(4)
    c0 = torch.rand(shape, dtype=torch.float64) * 0.12 + 0.44
    L = [3.0, 2.0, 2.5]
    plain = _make(3, list(shape), L, P, nsub=nsub)
    carry = _make(3, list(shape), L, P, nsub=nsub, carry=True)
    for s in plain + carry:
        yb, nyl = s.st.real_begin[1], s.st.real_shape[1]
        s.set_local(c0[:, yb:yb + nyl].contiguous().cuda())
    dom = mo.Domain(3, list(shape), L)
    ref = mo.CahnHilliardABM(dom, c0, M=0.2, kappa_factor=-0.001, mu_fn=mo.mu_double_well, substeps=4)
    for k in range(3):
        ref.step(1e-3)
        _step_all(plain, 1e-3, 4)
        _step_all(carry, 1e-3, 4)
        assert (_gather(carry) - _gather(plain)).abs().max().item() <= 1e-13
        assert (_gather(carry) - ref.c).abs().max().item() <= 1e-13
    assert [s.mode for s in carry] == [2] * P and [s.last_order for s in carry] == [1] * P
    # the N-hat history agrees with the plain run's up to the rounding of the transforms: the two flows pack different
    # pairs of real sequences into one complex z transform (c + i*mu vs two lines of mu), so mu-hat inherits an error of a few
    # eps * |c-hat|_max = eps * sum(c), which N-hat = -k^2 M mu-hat scales by up to k_max^2 * M
    k2max = sum((math.pi * n / l) ** 2 for n, l in zip(shape, L))
    tol = 64 * 2.2e-16 * float(c0.sum()) * k2max * 0.2
    for a, b in zip(carry, plain):
        assert (a.spec(a.cur) - b.spec(b.cur)).abs().max().item() <= tol
    # an external change of c invalidates the carried spectrum
    for s in carry:
        s.set_local(s.current() * 0.5 + 0.25)
    for s in plain:
        s.set_local(s.current() * 0.5 + 0.25)
    _step_all(plain, 1e-3, 2)
    _step_all(carry, 1e-3, 2)
    assert (_gather(carry) - _gather(plain)).abs().max().item() <= 1e-13


@pytest.mark.parametrize("shape,P,nsub,carry", [((64, 64, 64), 2, 2, False), ((64, 64, 64), 2, 2, True), ((64, 128, 64), 4, 3, True),
                                                ((12, 10, 8), 2, 1, True), ((9, 7, 5), 3, 1, False),
                                                # ny / P = 2 rows per chunk < the 4 threads of a 64-point line: the y pass without
                                                # the wave-uniform chunk offsets (k_ch_yfused<.., ALIGNED = false>)
                                                ((32, 64, 32), 32, 1, True), ((32, 64, 32), 32, 2, False)])
def test_slab_run_fused_z_passes(shape, P, nsub, carry):
    """run(count): the inverse z pass of a substep fused with the forward z pass of the next one (planned shapes; the generic
    path runs the two passes back to back) == the same substeps one at a time, incl. the first-time-step rule (no history
    rotation) and the carry-over"""
    torch.manual_seed(4)
    c0 = torch.rand(shape, dtype=torch.float64) * 0.12 + 0.44
    L = [3.0, 2.0, 2.5]
    a = _make(3, list(shape), L, P, nsub=nsub, carry=carry)
    b = _make(3, list(shape), L, P, nsub=nsub, carry=carry)
    for s in a + b:
        yb, nyl = s.st.real_begin[1], s.st.real_shape[1]
        s.set_local(c0[:, yb:yb + nyl].contiguous().cuda())
    # time step 1: 4 substeps without history rotation; time step 2: 5 substeps with it
    for s in a:
        s.time_step, s.sub_dt = 1, 1e-3
    for k in range(4):
        _substep_all(a)
    _run_all(b, 4, advance=False)
    assert (_gather(a) - _gather(b)).abs().max().item() <= 1e-15
    for s in a + b:
        s.advance_state()
    for k in range(5):
        _substep_all(a)
        if k < 4:
            for s in a:
                s.advance_state()
    _run_all(b, 5, advance=True)
    assert (_gather(a) - _gather(b)).abs().max().item() <= 1e-15
    assert [s.last_order for s in b] == [1] * P
    for x, y in zip(a, b):
        assert (x.spec(x.cur) - y.spec(y.cur)).abs().max().item() <= 1e-12 * max(1.0, x.cur.abs().max().item())


def test_rccl_exchange_single_rank():
    """SlabExchange over the nccl (= RCCL) backend: a world of one rank exercises the all_to_all_single
    call path (split sizes, float64 payload, async work handle) on the GPU box's single device."""
    import os
    import torch.distributed as dist
    from marlin_amd.slab import SlabExchange
    from tests.test_slab_gloo import _free_port
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        x = SlabExchange([1000], [1000])
        assert x.mode == "a2a"
        send = torch.rand(2000, dtype=torch.float64, device="cuda")
        recv = torch.zeros_like(send)
        w = x.run(send, recv, async_op=True)
        w.wait()
        torch.cuda.synchronize()
        assert torch.equal(send, recv)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("carry,nsub", [(False, 4), (True, 2), (True, 3)])
def test_slab_pipeline_over_rccl_single_rank(carry, nsub):
    """the production driver end to end on the GPU box's single device: SlabCahnHilliard.substep() with its asynchronous
    RCCL all_to_all_single exchanges (a one-rank communicator, MRL_FLAG_SLAB context), the local passes on a high-priority
    stream as bench.py runs them -- against the serial fused substep of the same library"""
    import os
    import torch.distributed as dist
    from marlin_amd.api import Context, ch_params
    from marlin_amd.slab import SlabCahnHilliard
    from tests.test_slab_gloo import _free_port
    shape, L = [64, 64, 64], [3.0, 2.0, 2.5]
    torch.manual_seed(8)
    c0 = (torch.rand(shape, dtype=torch.float64) * 0.12 + 0.44).cuda()
    p = ch_params()
    ctx = Context(3, shape, L)
    want, Nh = [c0.clone(), torch.empty_like(c0)], [ctx.empty_hist(), ctx.empty_hist()]
    for k in range(6):
        ctx.ch_substep(p, want[k % 2], want[1 - k % 2], Nh[k % 2], [Nh[1 - k % 2]] if k else [], 1 if k else 0, 1e-3)
    want = want[0].clone()
    torch.cuda.synchronize()        # the reference ran on the default stream; the pipeline below runs on another one
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    prev = torch.cuda.current_stream()
    try:
        torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
        s = SlabCahnHilliard(3, shape, L, p, 1, 0, nsub=nsub, carry=carry)
        assert s.x_fwd[0].mode == "a2a"
        s.set_local(c0)
        for _ in range(6):
            s.substep()
        torch.cuda.synchronize()
        assert s.last_order == 1 and s.mode == (2 if carry else 0)
        assert (s.current() - want).abs().max().item() <= 1e-13
    finally:
        torch.cuda.set_stream(prev)
        dist.destroy_process_group()


def test_spec_pitch_and_stage_argument_checks():
    """mrl_slab_ch_spec_pitch: padded to 128-byte rows on the planned pipeline, the natural nz/2+1 otherwise; the fused mechanics
    stage refuses contexts it was not built for (error code + message, no launch)"""
    from marlin_amd.api import Context
    from marlin_amd.slab import HipSlabStages
    fast = HipSlabStages(3, [64, 64, 64], [1.0, 1.0, 1.0], 2, 0)
    assert fast.recip_shape[-1] == 33 and fast.spec_pitch == 40
    generic = HipSlabStages(3, [12, 10, 9], [1.0, 1.0, 1.0], 2, 1)
    assert generic.spec_pitch == generic.recip_shape[-1] == 5
    serial = Context(3, [64, 64, 64], [1.0, 1.0, 1.0])
    assert int(serial.lib.mrl_slab_ch_spec_pitch(serial.h)) == 33
    assert not serial.lib.mrl_slab_gamma_tangent_fusable(serial.h)
    assert fast.ctx.lib.mrl_slab_gamma_tangent_fusable(fast.ctx.h) and not generic.ctx.lib.mrl_slab_gamma_tangent_fusable(generic.ctx.h)
    z = torch.zeros(16, dtype=torch.float64, device="cuda")
    rc = serial.lib.mrl_slab_gamma_tangent_z_fwd(serial.h, z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 0.0,
                                                 None, 0.0)
    assert rc != 0 and b"slab" in serial.lib.mrl_last_error(serial.h).lower()
    # row stage without an input field and without spectra left by the fused stage
    snd = torch.zeros(8, dtype=torch.float64, device="cuda")
    rc = fast.ctx.lib.mrl_slab_gamma_row_fwd(fast.ctx.h, 0, None, snd.data_ptr())
    assert rc != 0 and b"mrl_slab_gamma_tangent_z_fwd" in fast.ctx.lib.mrl_last_error(fast.ctx.h)
