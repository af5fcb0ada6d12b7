"""Slab-decomposed (multi-GPU) Cahn-Hilliard driver: one process per GPU, the global transpose of
DomainAction::fftSlab / ifftSlab (src/actions/DomainAction.C:869-1019) done as an RCCL all-to-all.

Host logic only.  The local stages are the `mrl_slab_*` entry points of libmarlin_hip.so (HIP kernels);
this module owns the exchange (torch.distributed: `nccl` = RCCL over xGMI on the GPUs, `gloo` in the CPU
tests), the send/receive buffers and the solver's history rotation (TensorSolver.C:93-109,
TensorBuffer.h:62-79).  A substep is pipelined over `nsub` sub-blocks of the kz axis (after the z pass
every kz plane is an independent 2-D problem), so the links and the GPU work at the same time:

    Z    mu = f'(c); z pass of c and mu
    A_s  forward x pass of both fields for kz in K_s                      -> send_f[s]
         all-to-all s (asynchronous; both fields in one message per peer)
    B_s  forward y pass, Mbar*mubar, ABM update, inverse y pass            -> send_i[s]
         all-to-all s (asynchronous)
    C_s  inverse x pass
    E    inverse z pass                                                    -> c
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


def _prefix(counts: Sequence[int]) -> List[int]:
    out, s = [], 0
    for c in counts:
        out.append(s)
        s += c
    return out


CARRY_NONE, CARRY_OUT, CARRY_IN = 0, 1, 2


class SlabExchange:
    """The global transpose: chunk p of `send` goes to rank p, chunk p of `recv` came from rank p.

    Counts are in complex elements (mrl_slab_counts); buffers are flat float64 tensors holding
    interleaved (re, im).  `mode="a2a"` uses all_to_all_single (RCCL), `mode="p2p"` batched
    isend/irecv (works on gloo, which has no all-to-all)."""

    def __init__(self, send_counts, recv_counts, group=None, mode: Optional[str] = None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        assert len(send_counts) == self.world and len(recv_counts) == self.world
        self.send_split = [2 * int(c) for c in send_counts]
        self.recv_split = [2 * int(c) for c in recv_counts]
        self.send_off = _prefix(self.send_split)
        self.recv_off = _prefix(self.recv_split)
        if mode is None:
            mode = "a2a" if dist.get_backend(group) == "nccl" else "p2p"
        self.mode = mode

    @property
    def send_len(self):
        return sum(self.send_split)

    @property
    def recv_len(self):
        return sum(self.recv_split)

    def run(self, send: torch.Tensor, recv: torch.Tensor, async_op: bool = False):
        assert send.numel() == self.send_len and recv.numel() == self.recv_len
        if self.mode == "a2a":
            return dist.all_to_all_single(recv, send, self.recv_split, self.send_split, group=self.group,
                                          async_op=async_op)
        if send.is_cuda and dist.get_backend(self.group) == "gloo":
            # smoke-test path only (several ranks sharing one GPU, where RCCL refuses to run): stage through the host
            hs, hr = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
            self.run(hs, hr)
            recv.copy_(hr)
            return _WorkList([]) if async_op else None
        # own chunk: local copy; the others: one isend + one irecv per peer
        r = self.rank
        recv[self.recv_off[r]:self.recv_off[r] + self.recv_split[r]].copy_(
            send[self.send_off[r]:self.send_off[r] + self.send_split[r]])
        ops = []
        for p in range(self.world):
            if p == r:
                continue
            ops.append(dist.P2POp(dist.isend, send[self.send_off[p]:self.send_off[p] + self.send_split[p]], p,
                                  group=self.group))
            ops.append(dist.P2POp(dist.irecv, recv[self.recv_off[p]:self.recv_off[p] + self.recv_split[p]], p,
                                  group=self.group))
        works = dist.batch_isend_irecv(ops) if ops else []
        if async_op:
            return _WorkList(works)
        for w in works:
            w.wait()
        return None


class _WorkList:
    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()


class HipSlabStages:
    """The local stages of one rank, bound to libmarlin_hip.so (no CPU fallback)."""

    def __init__(self, dim, shape, L, nranks, rank, spectrum=None, weights=None, device=None):
        from . import api

        if spectrum is None:
            spectrum = api.SPECTRUM_HALF if dim == 3 else api.SPECTRUM_FULL
        self.ctx = api.Context(dim, shape, L, nranks=nranks, rank=rank, weights=weights, spectrum=spectrum, slab=True,
                               device=device)
        self.lib = self.ctx.lib
        self.device = self.ctx.device
        self.real_shape = self.ctx.real_shape
        self.real_begin = self.ctx.real_begin
        self.recip_shape = self.ctx.recip_shape
        self.recip_begin = self.ctx.recip_begin
        self.spec_pitch = int(self.lib.mrl_slab_ch_spec_pitch(self.ctx.h))

    def counts(self, forward: bool):
        n = self.ctx.nranks
        sc, rc = (C.c_int64 * n)(), (C.c_int64 * n)()
        self.ctx._check(self.lib.mrl_slab_counts(self.ctx.h, 1 if forward else 0, sc, rc, None, None))
        return list(sc), list(rc)

    def _p(self, t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    def fwd_local(self, real_in, send):
        self.ctx._check(self.lib.mrl_slab_fwd_local(self.ctx.h, self._p(real_in), self._p(send)))

    def fwd_finish(self, recv, spec_out):
        self.ctx._check(self.lib.mrl_slab_fwd_finish(self.ctx.h, self._p(recv), self._p(spec_out)))

    def inv_local(self, spec_in, send):
        self.ctx._check(self.lib.mrl_slab_inv_local(self.ctx.h, self._p(spec_in), self._p(send)))

    def inv_finish(self, recv, real_out):
        self.ctx._check(self.lib.mrl_slab_inv_finish(self.ctx.h, self._p(recv), self._p(real_out)))

    # carry: CARRY_NONE / CARRY_OUT / CARRY_IN (include/marlin_hip.h: spectral carry-over)
    def ch_counts(self, sub, nsub, forward: bool, carry: int = 0):
        n = self.ctx.nranks
        sc, rc = (C.c_int64 * n)(), (C.c_int64 * n)()
        self.ctx._check(self.lib.mrl_slab_ch_counts(self.ctx.h, sub, nsub, 1 if forward else 0, carry, sc, rc))
        return list(sc), list(rc)

    def ch_z_fwd(self, p, c_in, mu=None, carry: int = 0):
        self.ctx._check(self.lib.mrl_slab_ch_z_fwd(self.ctx.h, C.byref(p), self._p(c_in), self._p(mu), carry))

    def ch_x_fwd(self, sub, nsub, send, carry: int = 0):
        self.ctx._check(self.lib.mrl_slab_ch_x_fwd(self.ctx.h, sub, nsub, self._p(send), carry))

    def ch_kspace(self, p, sub, nsub, recv, send, Nnew, Nold, order, sub_dt, cbar=None, carry: int = 0):
        arr = (C.c_void_p * max(1, len(Nold)))(*[t.data_ptr() for t in Nold])
        self.ctx._check(self.lib.mrl_slab_ch_kspace(self.ctx.h, C.byref(p), sub, nsub, self._p(recv), self._p(send),
                                                    self._p(Nnew), arr, order, sub_dt, self._p(cbar), carry))

    def ch_x_inv(self, sub, nsub, recv):
        self.ctx._check(self.lib.mrl_slab_ch_x_inv(self.ctx.h, sub, nsub, self._p(recv)))

    def ch_z_inv(self, c_out):
        self.ctx._check(self.lib.mrl_slab_ch_z_inv(self.ctx.h, self._p(c_out)))

    def ch_z_inv_fwd(self, p, mu=None, carry: int = 0):
        self.ctx._check(self.lib.mrl_slab_ch_z_inv_fwd(self.ctx.h, C.byref(p), self._p(mu), carry))

    def empty(self, n):
        return torch.zeros(n, dtype=torch.float64, device=self.device)


class SlabCahnHilliard:
    """AdamsBashforthMoulton::substep (+ its compute group) on a slab-decomposed grid.

    `stages` is the rank-local compute (HipSlabStages by default; the CPU tests inject an oracle-backed
    object with the same methods to exercise this host logic over gloo).  `exchange_factory(send_counts,
    recv_counts)` builds the transposes (SlabExchange by default; the single-GPU loop-back harness of
    the GPU tests injects its own and drives the phases of all ranks in lock step).  `nsub` = number of
    kz sub-blocks the substep is pipelined over.

    `carry=True` switches the spectral carry-over on (include/marlin_hip.h): the first substep after the field was set
    runs the reference's data flow and keeps ubar; every later substep uses it as c-hat, so that only mu travels forward
    (2 slab transposes per substep instead of 3).  Anything that changes c from outside must go through set_local /
    set_initial / invalidate_carry()."""

    def __init__(self, dim, shape, L, params, nranks, rank, predictor_order: int = 2, sub_dt: float = 1e-3,
                 stages=None, exchange_factory: Optional[Callable] = None, nsub: int = 2, carry: bool = False, exp: int = 0):
        self.p = params
        self.carry = carry
        self._carry_valid = False
        self.mode = CARRY_NONE
        self.nranks, self.rank = nranks, rank
        self.st = stages if stages is not None else HipSlabStages(dim, shape, L, nranks, rank)
        self.ctx = getattr(self.st, "ctx", None)
        if exp and self.ctx is not None:           # experiment switches that change the exchange layouts must precede the counts below
            self.ctx.set_option(0, exp)
        self.pred = predictor_order - 1            # AdamsBashforthMoulton.C:48
        self.sub_dt = sub_dt
        nzc = self.st.recip_shape[2] if dim == 3 else 1
        self.nsub = max(1, min(nsub, nzc))
        mk = exchange_factory if exchange_factory is not None else (lambda s, r: SlabExchange(s, r))
        e = self.st.empty
        self.x_fwd, self.x_inv, self.x_fwd1, self._len1 = [], [], [], []
        self.send_f, self.recv_f, self.send_i, self.recv_i = [], [], [], []
        for s in range(self.nsub):
            fs, fr = self.st.ch_counts(s, self.nsub, True)
            bs, br = self.st.ch_counts(s, self.nsub, False)
            self.x_fwd.append(mk(fs, fr))
            self.x_inv.append(mk(bs, br))
            if carry:   # the one-field forward exchange (mu only); its buffers are prefixes of the two-field ones
                f1s, f1r = self.st.ch_counts(s, self.nsub, True, CARRY_IN)
                self.x_fwd1.append(mk(f1s, f1r))
                self._len1.append((2 * sum(f1s), 2 * sum(f1r)))
            self.send_f.append(e(2 * sum(fs)))
            self.recv_f.append(e(2 * sum(fr)))
            self.send_i.append(e(2 * sum(bs)))
            self.recv_i.append(e(2 * sum(br)))
        nreal = 1
        for s in self.st.real_shape:
            nreal *= s
        nspec = 1
        for s in self.st.recip_shape:
            nspec *= s
        self.c = e(nreal)
        self.c_new = e(nreal)
        self._pitch = int(self.st.spec_pitch)              # padded kz pitch of the rank-local spectral arrays
        nspec = nspec // self.st.recip_shape[-1] * self._pitch
        self.Nhat = [e(2 * nspec) for _ in range(self.pred + 2)]   # ring: current + history + one free
        self.cbar = e(2 * nspec) if carry else None                # carried spectrum: ubar of the last substep
        self.hist: List[torch.Tensor] = []                         # N-hat_old[0..] (handles into the ring)
        self.cur: Optional[torch.Tensor] = None                    # N-hat of the last substep
        self.time_step = 0
        self.dt_old: Optional[float] = None                        # the previous time step's dt (TensorSolver.C:48)
        self.dt_changed = False
        self._substep_index = 0

    # ---- state ---------------------------------------------------------------------------------
    def set_initial(self, gen: Callable[[int, int], "object"]):
        """gen(count, offset) -> values of the global row-major field [nx][ny][nz] at flat indices
        offset .. offset+count (numpy array or tensor); every rank fills its own y-slab."""
        shp, beg = self.st.real_shape, self.st.real_begin
        dim = len(shp)
        host = torch.empty(shp, dtype=torch.float64)
        ny_glob = self._global_ny
        if dim == 2:
            nx, nyl = shp
            for ix in range(nx):
                host[ix] = torch.as_tensor(gen(nyl, ix * ny_glob + beg[1]), dtype=torch.float64)
        else:
            nx, nyl, nz = shp
            for ix in range(nx):
                host[ix] = torch.as_tensor(gen(nyl * nz, (ix * ny_glob + beg[1]) * nz), dtype=torch.float64).reshape(nyl, nz)
        self.c.copy_(host.reshape(-1))
        self._carry_valid = False

    def reset(self, gen: Callable[[int, int], "object"]):
        """back to the initial condition with an empty history (the state right after construction + set_initial)"""
        self.set_initial(gen)
        self.hist = []
        self.cur = None
        self.time_step = 0
        self.dt_old = None
        self.dt_changed = False
        self._substep_index = 0

    def set_local(self, c_local: torch.Tensor):
        self.c.copy_(c_local.reshape(-1))
        self._carry_valid = False

    def spec(self, t: torch.Tensor) -> torch.Tensor:
        """complex view [x_me][ny]..[nz/2+1] of one of the rank-local spectral arrays (Nhat ring, cbar) without its row padding"""
        shp = list(self.st.recip_shape)
        v = torch.view_as_complex(t.view(*shp[:-1], self._pitch, 2))
        return v[..., :shp[-1]]

    def invalidate_carry(self):
        """c was modified by something else than this solver: the next substep recomputes c-hat from it"""
        self._carry_valid = False

    @property
    def _global_ny(self):
        return self.st.recip_shape[1]

    def current(self) -> torch.Tensor:
        return self.c.reshape(self.st.real_shape)

    # ---- history: TensorBuffer<T>::advanceState (TensorBuffer.h:62-79) as a ring of device buffers
    def advance_state(self):
        if self.cur is None or self.pred == 0:
            return
        if len(self.hist) < self.pred:
            self.hist.append(None)
        for i in range(len(self.hist) - 1, 0, -1):
            self.hist[i] = self.hist[i - 1]
        self.hist[0] = self.cur

    def _free_Nhat(self):
        busy = {id(t) for t in self.hist if t is not None}
        if self.cur is not None:
            busy.add(id(self.cur))
        for t in self.Nhat:
            if id(t) not in busy:
                return t
        raise RuntimeError("history ring exhausted")

    # ---- the phases of one substep -------------------------------------------------------------------
    def phase_z(self):
        self.mode = CARRY_NONE if not self.carry else (CARRY_IN if self._carry_valid else CARRY_OUT)
        self.st.ch_z_fwd(self.p, self.c, carry=self.mode)
        self._order = self._ab_order()
        self._new = self._free_Nhat()

    def _ab_order(self):
        """AdamsBashforthMoulton.C:75,88-91: the order the history allows, restarted at first order when dt changed"""
        return min(0 if (self._substep_index < self.pred and self.dt_changed) else len(self.hist), self.pred)

    def phase_a(self, s):
        self.st.ch_x_fwd(s, self.nsub, self.send_f[s], carry=self.mode)

    def fwd_exchange(self, s):
        """(exchange object, send view, recv view) of sub-block s for the current carry mode"""
        if self.mode == CARRY_IN:
            ns, nr = self._len1[s]
            return self.x_fwd1[s], self.send_f[s][:ns], self.recv_f[s][:nr]
        return self.x_fwd[s], self.send_f[s], self.recv_f[s]

    def phase_b(self, s):
        self.st.ch_kspace(self.p, s, self.nsub, self.recv_f[s], self.send_i[s], self._new, self.hist[:self._order],
                          self._order, self.sub_dt, cbar=self.cbar, carry=self.mode)

    def phase_c(self, s):
        self.st.ch_x_inv(s, self.nsub, self.recv_i[s])

    def _finish(self):
        self.cur = self._new
        self.last_order = self._order
        self._carry_valid = self.carry

    def phase_e(self):
        self.st.ch_z_inv(self.c_new)
        self.c, self.c_new = self.c_new, self.c
        self._finish()

    def phase_ez(self, advance: bool = True):
        """between two substeps of one run(): the inverse z pass of the finished substep fused with the forward z pass of the
        next one (the intermediate real field is not materialised; current() is valid again after the closing phase_e)"""
        self._finish()
        if advance:
            self.advance_state()
        self.mode = CARRY_NONE if not self.carry else CARRY_IN
        self.st.ch_z_inv_fwd(self.p, carry=self.mode)
        self._substep_index += 1
        self._order = self._ab_order()
        self._new = self._free_Nhat()

    def substep(self, advance: bool = True):
        """One substep including all exchanges; `advance` rotates the history afterwards (what
        TensorSolver::computeBuffer does between substeps).  Exchanges are asynchronous: a work handle's
        wait() only orders the compute stream behind that exchange, the host never blocks."""
        self.phase_z()
        self._exchange_phases()
        self.phase_e()
        if advance:
            self.advance_state()

    def _exchange_phases(self):
        S = range(self.nsub)
        wf, wi = [], []
        for s in S:
            self.phase_a(s)
            x, snd, rcv = self.fwd_exchange(s)
            wf.append(x.run(snd, rcv, async_op=True))
        for s in S:
            if wf[s] is not None:
                wf[s].wait()
            self.phase_b(s)
            wi.append(self.x_inv[s].run(self.send_i[s], self.recv_i[s], async_op=True))
        for s in S:
            if wi[s] is not None:
                wi[s].wait()
            self.phase_c(s)

    def run(self, count: int, advance: bool = True, advance_after: bool = False):
        """`count` substeps as one unit (the substep loop of TensorSolver::computeBuffer): between two substeps the two z passes
        are one kernel.  `advance`: rotate the history between substeps (False while timeStep() <= 1); `advance_after`: also
        after the last one (what a following run() / substep() of the same time step needs)."""
        self._substep_index = 0
        for k in range(count):
            if k == 0:
                self.phase_z()
            else:
                self.phase_ez(advance)
            self._exchange_phases()
        self.phase_e()
        if advance_after:
            self.advance_state()

    def step(self, dt: float, substeps: int):
        """One MOOSE time step: advanceState is a no-op while timeStep() <= 1 (TensorProblem.C:451-472),
        so every substep of the first step is AB1."""
        self.time_step += 1
        self.dt_changed = self.dt_old is not None and dt != self.dt_old
        self.dt_old = dt
        if self.time_step > 1:
            self.advance_state()
        self.sub_dt = dt / substeps
        self.run(substeps, advance=self.time_step > 1)


class TorchComm:
    """Exchanges and scalar all-reduces over torch.distributed (RCCL on GPUs, gloo on CPU)."""

    def __init__(self, device=None, group=None):
        self.group = group
        self.device = device

    def exchange(self, send_counts, recv_counts):
        return SlabExchange(send_counts, recv_counts, group=self.group)

    def allreduce(self, values: Sequence[float]) -> List[float]:
        t = torch.tensor(list(values), dtype=torch.float64, device=self.device)
        dist.all_reduce(t, group=self.group)
        return t.tolist()


class SlabMechanics:
    """FFTMechanics::computeBuffer (src/tensor_computes/FFTMechanics.C:96-163) with HyperElasticIsotropic and
    MooseTensor::conjugateGradientSolve (include/utils/MarlinUtils.h:55-131) on a slab-decomposed grid.

    Every field is the rank's y-slab of the reference's value-major tensor, flat [nx*nyl*nz][D*D].  The driver below
    is host control flow only: each arithmetic step is a libmarlin_hip.so call on the local slab (pointwise stress /
    tangent kernels, slab FFT stages around the field-major Gamma projection, vector updates, local dot products);
    `comm` supplies the transposes and the all-reduce of the 1-3 CG scalars per iteration that the reference lacks
    (its norms are serial-only, DomainAction.C:1564-1567)."""

    def __init__(self, dim, shape, L, nranks, rank, K_local, mu_local, comm=None, l_tol=1e-2, l_max_its=0, nl_rel_tol=1e-5,
                 nl_abs_tol=1e-8, nl_max_its=100, stages=None, fast: Optional[bool] = None, tangent_fusion: bool = True):
        """fast=None: use the fused field-major row pipeline (mrl_slab_gamma_row_*) when the context supports it;
        tangent_fusion=False keeps the tangent and the forward z pass of the CG iteration as separate kernels (A/B, tests)"""
        self.no_tangent_fusion = not tangent_fusion
        self.st = stages if stages is not None else HipSlabStages(dim, shape, L, nranks, rank)
        self.ctx = self.st.ctx
        self.lib = self.ctx.lib
        self.dim, self.dd = dim, dim * dim
        self.comm = comm if comm is not None else TorchComm(device=self.st.device)
        self.npts = 1
        for s in self.st.real_shape:
            self.npts *= s
        self.nspec = 1
        for s in self.st.recip_shape:
            self.nspec *= s
        self.nglobal = 1
        for s in shape[:dim]:
            self.nglobal *= s
        fs, fr = self.st.counts(True)
        bs, br = self.st.counts(False)
        self.x_fwd = self.comm.exchange(fs, fr)
        self.x_inv = self.comm.exchange(bs, br)
        e = self.st.empty
        self.K, self.mu = K_local.reshape(-1).contiguous(), mu_local.reshape(-1).contiguous()
        self.send_f, self.recv_f = e(2 * sum(fs)), e(2 * sum(fr))
        self.send_i, self.recv_i = e(2 * sum(bs)), e(2 * sum(br))
        self.fm = e(self.npts * self.dd)                 # field-major real work array
        self.spec = e(2 * self.nspec * self.dd)          # field-major spectra
        self.l_tol, self.l_max_its = l_tol, (l_max_its or self.nglobal)
        self.nl_rel_tol, self.nl_abs_tol, self.nl_max_its = nl_rel_tol, nl_abs_tol, nl_max_its
        can = dim == 3 and self.npts % 2 == 0 and bool(self.lib.mrl_slab_fast_path(self.ctx.h))
        if fast and not can:
            raise ValueError("the fused slab mechanics path needs a 3-D context with planned extents and equal power-of-two partitions")
        self.fast = can if fast is None else bool(fast)
        if self.fast:
            # one message per tensor row and peer; the inverse exchange lands in the (by then free) send buffer of the row
            n = self.ctx.nranks
            sc, rc = (C.c_int64 * n)(), (C.c_int64 * n)()
            self._chk(self.lib.mrl_slab_gamma_counts(self.ctx.h, 1, sc, rc))
            gs, gr = list(sc), list(rc)
            self.xg_fwd = self.comm.exchange(gs, gr)
            self.xg_inv = self.comm.exchange(gr, gs)
            self.g_send = [e(2 * sum(gs)) for _ in range(3)]
            self.g_recv = [e(2 * sum(gr)) for _ in range(3)]

    # ---- thin wrappers over the C ABI (local work only)
    def _chk(self, rc):
        self.ctx._check(rc)

    def _p(self, t):
        return C.c_void_p(t.data_ptr())

    def _axpby(self, a, x, b, y, out):
        self._chk(self.lib.mrl_axpby(self.ctx.h, a, self._p(x), b, self._p(y), self._p(out), out.numel()))

    def _dot(self, a, b):
        return self.comm.allreduce([self.ctx.dot(a, b)])[0]

    def _norm(self, a):
        return self.comm.allreduce([self.ctx.dot(a, a)])[0] ** 0.5

    def average(self, F):
        """DomainAction::average of a value-major field: local sums / global count, summed over ranks"""
        out = (C.c_double * self.dd)()
        self._chk(self.lib.mrl_average(self.ctx.h, self._p(F), self.dd, out))
        return self.comm.allreduce(list(out))

    # (fast path: every vector of the solve is field-major [D*D][npts]; otherwise value-major as in the reference)
    def stress(self, F, out):
        fn = self.lib.mrl_mech_stress_fm if self.fast else self.lib.mrl_mech_stress
        self._chk(fn(self.ctx.h, self._p(F), self._p(self.K), self._p(self.mu), self._p(out)))

    def tangent(self, Flin, dF, out):
        fn = self.lib.mrl_mech_tangent_apply_fm if self.fast else self.lib.mrl_mech_tangent_apply
        self._chk(fn(self.ctx.h, self._p(Flin), self._p(self.K), self._p(self.mu), self._p(dF), self._p(out)))

    def _relayout(self, to_fm: bool, src, dst):
        self._chk(self.lib.mrl_relayout(self.ctx.h, 1 if to_fm else 0, self._p(src), self._p(dst), self.npts, self.dd))

    def gamma_fast(self, A, out, scale=1.0, dotv=None):
        """out = scale * G(A) on field-major fields: three tensor rows, each z+x passes -> all-to-all -> fused y pass with the
        projection -> all-to-all -> inverse x+z passes; row r+1 is transformed while row r is on the wire.
        dotv: also return the all-reduced sum(out * dotv), accumulated by the last z pass (the p.Ap of the CG).
        A = None: the z spectra of the nine fields were left in the context by mrl_slab_gamma_tangent_z_fwd (x passes only)"""
        h = self.ctx.h
        wf, wi = [], []
        for r in range(3):
            self._chk(self.lib.mrl_slab_gamma_row_fwd(h, r, self._p(A) if A is not None else None, self._p(self.g_send[r])))
            wf.append(self.xg_fwd.run(self.g_send[r], self.g_recv[r], async_op=True))
        for r in range(3):
            if wf[r] is not None:
                wf[r].wait()
            self._chk(self.lib.mrl_slab_gamma_row_mid(h, self._p(self.g_recv[r]), scale))
            wi.append(self.xg_inv.run(self.g_recv[r], self.g_send[r], async_op=True))
        for r in range(3):
            if wi[r] is not None:
                wi[r].wait()
            self._chk(self.lib.mrl_slab_gamma_row_inv(h, r, self._p(self.g_send[r]), self._p(out),
                                                      self._p(dotv) if dotv is not None else None))
        if dotv is not None:
            loc = C.c_double()
            self._chk(self.lib.mrl_slab_gamma_dot(h, C.byref(loc)))
            return self.comm.allreduce([loc.value])[0]
        return None

    def gamma(self, A, out, scale=1.0):
        """out = scale * G(A): per component slab transform, field-major projection, inverse"""
        if self.fast:
            return self.gamma_fast(A, out, scale)
        self._chk(self.lib.mrl_relayout(self.ctx.h, 1, self._p(A), self._p(self.fm), self.npts, self.dd))
        for c in range(self.dd):
            self.st.fwd_local(self.fm[c * self.npts:(c + 1) * self.npts], self.send_f)
            self.x_fwd.run(self.send_f, self.recv_f)
            self.st.fwd_finish(self.recv_f, self.spec[2 * c * self.nspec:2 * (c + 1) * self.nspec])
        self._chk(self.lib.mrl_slab_gamma_project(self.ctx.h, self._p(self.spec), scale))
        for c in range(self.dd):
            self.st.inv_local(self.spec[2 * c * self.nspec:2 * (c + 1) * self.nspec], self.send_i)
            self.x_inv.run(self.send_i, self.recv_i)
            self.st.inv_finish(self.recv_i, self.fm[c * self.npts:(c + 1) * self.npts])
        self._chk(self.lib.mrl_relayout(self.ctx.h, 0, self._p(self.fm), self._p(out), self.npts, self.dd))

    def _cg_fast(self, Flin, b, x):
        """the same iteration with the vector work fused into three kernels per iteration: p <- r + beta p inside the tangent
        kernel, p.Ap inside the last z pass of the Gamma pipeline, x / r updates and r.r in one pass"""
        e = self.st.empty
        n = b.numel()
        b_norm = self._norm(b)
        if b_norm == 0.0:
            return 0
        tmp, Ap, r, p = e(n), e(n), e(n), e(n)
        self.tangent(Flin, x, tmp)
        self.gamma(tmp, Ap)
        self._axpby(1.0, b, -1.0, Ap, r)            # r = b - A x
        p.copy_(r)
        rz_old = self._dot(r, r)
        beta = 0.0
        rr = C.c_double()
        fuse = bool(self.lib.mrl_slab_gamma_tangent_fusable(self.ctx.h)) and not self.no_tangent_fusion
        alpha_prev, pending = 0.0, False
        for k in range(self.l_max_its):
            if fuse:
                # [x += alpha_prev p ;] p = r + beta p ; z spectra of K_dF(p) stay in the context: K_dF(p) is never written
                self._chk(self.lib.mrl_slab_gamma_tangent_z_fwd(self.ctx.h, self._p(Flin), self._p(self.K), self._p(self.mu),
                                                               self._p(p), self._p(r), beta,
                                                               self._p(x) if pending else None, alpha_prev))
                pAp = self.gamma_fast(None, Ap, 1.0, dotv=p)
            else:
                self._chk(self.lib.mrl_mech_tangent_dir_fm(self.ctx.h, self._p(Flin), self._p(self.K), self._p(self.mu), self._p(p),
                                                           self._p(r), beta, self._p(tmp)))       # p = r + beta p ; tmp = K_dF(p)
                pAp = self.gamma_fast(tmp, Ap, 1.0, dotv=p)
            alpha = rz_old / pAp
            if fuse:   # the x update waits for the next direction kernel (or the axpby after the loop)
                self._chk(self.lib.mrl_cg_update_r(self.ctx.h, alpha, self._p(r), self._p(Ap), n, C.byref(rr)))
                alpha_prev, pending = alpha, True
            else:
                self._chk(self.lib.mrl_cg_update(self.ctx.h, alpha, self._p(x), self._p(r), self._p(p), self._p(Ap), n, C.byref(rr)))
            rz_new = self.comm.allreduce([rr.value])[0]
            if rz_new ** 0.5 <= self.l_tol * b_norm:
                if pending:
                    self._axpby(alpha_prev, p, 1.0, x, x)
                return k + 1
            beta = rz_new / rz_old
            rz_old = rz_new
        if pending:
            self._axpby(alpha_prev, p, 1.0, x, x)
        return self.l_max_its

    # ---- MooseTensor::conjugateGradientSolve with A = G o K_dF
    def _cg(self, Flin, b, x):
        if self.fast:
            return self._cg_fast(Flin, b, x)
        e = self.st.empty
        n = b.numel()
        b_norm = self._norm(b)
        if b_norm == 0.0:
            return 0
        tmp, Ap, r, p = e(n), e(n), e(n), e(n)
        self.tangent(Flin, x, tmp)
        self.gamma(tmp, Ap)
        self._axpby(1.0, b, -1.0, Ap, r)            # r = b - A x
        p.copy_(r)
        rz_old = self._dot(r, r)
        for k in range(self.l_max_its):
            self.tangent(Flin, p, tmp)
            self.gamma(tmp, Ap)
            alpha = rz_old / self._dot(p, Ap)
            self._axpby(1.0, x, alpha, p, x)
            self._axpby(1.0, r, -alpha, Ap, r)
            rz_new = self._dot(r, r)
            if rz_new ** 0.5 <= self.l_tol * b_norm:
                return k + 1
            self._axpby(1.0, r, rz_new / rz_old, p, p)
            rz_old = rz_new
        return self.l_max_its

    def newton_cg(self, F, applied: Optional[torch.Tensor]):
        """-> (Fnew, P, stats); F, Fnew, P: local value-major slabs (flat); applied: [D*D] device tensor or None"""
        e = self.st.empty
        n = self.npts * self.dd
        F = F.reshape(-1)
        if self.fast:                               # the solve runs on field-major vectors; convert at this boundary only
            Ffm = e(n)
            self._relayout(True, F, Ffm)
            F = Ffm
        u, P, b, x, tmp = F.clone(), e(n), e(n), torch.zeros(n, dtype=torch.float64, device=F.device), e(n)
        stats = {"newton_its": 0, "cg_its": []}
        if applied is not None:
            if self.fast:
                app = applied.reshape(self.dd, 1).expand(self.dd, self.npts).contiguous().reshape(-1)
            else:
                app = applied.reshape(1, self.dd).expand(self.npts, self.dd).contiguous().reshape(-1)
            self.tangent(F, app, tmp)               # K4 stays linearised at F for the first solve
            self.gamma(tmp, b, -1.0)
            self._axpby(1.0, u, 1.0, app, u)
        else:
            b.zero_()
        Fn = self._norm(u)
        lin = F
        iiter = 0
        while True:
            stats["cg_its"].append(self._cg(lin, b, x))
            self._axpby(1.0, u, 1.0, x, u)
            lin = u
            self.stress(u, P)
            self.gamma(P, b, -1.0)
            anorm = self._norm(x)
            rnorm = anorm / Fn
            stats["newton_its"] = iiter + 1
            if (rnorm < self.nl_rel_tol or anorm < self.nl_abs_tol) and iiter > 0:
                break
            iiter += 1
            if iiter > self.nl_max_its:
                raise RuntimeError("nl_max_its: Exceeded the maximum number of nonlinear iterations without converging.")
        if self.fast:                               # back to the reference's value-major layout
            self._relayout(False, u, tmp)
            self._relayout(False, P, b)
            u, P = tmp, b
        return u, P, stats
