"""Slab-decomposed (multi-GPU) Cahn-Hilliard driver: one process per GPU, the global transpose of
DomainAction::fftSlab / ifftSlab (src/actions/DomainAction.C:869-1019) done as an RCCL all-to-all.

Host logic only.  The local stages are the `mrl_slab_*` entry points of libmarlin_hip.so (HIP kernels);
this module owns the exchange (torch.distributed: `nccl` = RCCL over xGMI on the GPUs, `gloo` in the CPU
tests), the send/receive buffers and the solver's history rotation (TensorSolver.C:93-109,
TensorBuffer.h:62-79).  A substep is three phases separated by the two exchanges:

    A  mu = f'(c); z and x passes of fft(c), fft(mu)           -> send (2 fields)
       exchange (forward, per field; the c field travels while the mu field's x pass runs)
    B  y pass of both fields, Mbar*mubar, ABM update, inverse y pass    -> send
       exchange (inverse)
    C  inverse x and z passes                                            -> c
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
        self.ctx = api.Context(dim, shape, L, nranks=nranks, rank=rank, weights=weights, spectrum=spectrum,
                               device=device)
        self.lib = self.ctx.lib
        self.device = self.ctx.device
        self.real_shape = self.ctx.real_shape
        self.real_begin = self.ctx.real_begin
        self.recip_shape = self.ctx.recip_shape
        self.recip_begin = self.ctx.recip_begin

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

    def ch_fwd_local(self, p, c_in, send2, part=-1, mu=None):
        self.ctx._check(self.lib.mrl_slab_ch_fwd_local(self.ctx.h, C.byref(p), self._p(c_in), self._p(send2),
                                                       self._p(mu), part))

    def ch_kspace(self, p, recv2, send, Nnew, Nold, order, sub_dt, cbar=None):
        arr = (C.c_void_p * max(1, len(Nold)))(*[t.data_ptr() for t in Nold])
        self.ctx._check(self.lib.mrl_slab_ch_kspace(self.ctx.h, C.byref(p), self._p(recv2), self._p(send),
                                                    self._p(Nnew), arr, order, sub_dt, self._p(cbar)))

    def empty(self, n):
        return torch.empty(n, dtype=torch.float64, device=self.device)


class SlabCahnHilliard:
    """AdamsBashforthMoulton::substep (+ its compute group) on a slab-decomposed grid.

    `stages` is the rank-local compute (HipSlabStages by default; the CPU tests inject an oracle-backed
    object with the same methods to exercise this host logic over gloo).  `exchange_factory(send_counts,
    recv_counts)` builds the transposes (SlabExchange by default; the single-GPU loop-back harness of
    the GPU tests injects its own and drives the phases of all ranks in lock step)."""

    def __init__(self, dim, shape, L, params, nranks, rank, predictor_order: int = 2, sub_dt: float = 1e-3,
                 stages=None, exchange_factory: Optional[Callable] = None, overlap: bool = True):
        self.p = params
        self.nranks, self.rank = nranks, rank
        self.st = stages if stages is not None else HipSlabStages(dim, shape, L, nranks, rank)
        self.ctx = getattr(self.st, "ctx", None)
        self.pred = predictor_order - 1            # AdamsBashforthMoulton.C:48
        self.sub_dt = sub_dt
        self.overlap = overlap
        mk = exchange_factory if exchange_factory is not None else (lambda s, r: SlabExchange(s, r))
        fs, fr = self.st.counts(True)
        bs, br = self.st.counts(False)
        self.x_fwd = mk(fs, fr)
        self.x_inv = mk(bs, br)
        self.n_fs, self.n_fr = 2 * sum(fs), 2 * sum(fr)   # doubles per field
        self.n_bs, self.n_br = 2 * sum(bs), 2 * sum(br)
        e = self.st.empty
        self.send2 = e(2 * self.n_fs)
        self.recv2 = e(2 * self.n_fr)
        self.send = e(self.n_bs)
        self.recv = e(self.n_br)
        nreal = 1
        for s in self.st.real_shape:
            nreal *= s
        nspec = 1
        for s in self.st.recip_shape:
            nspec *= s
        self.c = e(nreal)
        self.c_new = e(nreal)
        self.Nhat = [e(2 * nspec) for _ in range(self.pred + 2)]   # ring: current + history + one free
        self.hist: List[torch.Tensor] = []                         # N̂_old[0..] (handles into the ring)
        self.cur: Optional[torch.Tensor] = None                    # N̂ of the last substep
        self.time_step = 0
        self._pending = []

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

    def set_local(self, c_local: torch.Tensor):
        self.c.copy_(c_local.reshape(-1))

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

    # ---- the three phases ------------------------------------------------------------------------
    def phase_a(self):
        if self.overlap:
            self.st.ch_fwd_local(self.p, self.c, self.send2, part=0)
        else:
            self.st.ch_fwd_local(self.p, self.c, self.send2, part=-1)

    def phase_a2(self):
        if self.overlap:
            self.st.ch_fwd_local(self.p, self.c, self.send2, part=1)

    def phase_b(self):
        order = min(len(self.hist), self.pred)       # AdamsBashforthMoulton.C:90-91 (constant dt)
        new = self._free_Nhat()
        self.st.ch_kspace(self.p, self.recv2, self.send, new, self.hist[:order], order, self.sub_dt)
        self.cur = new
        self.last_order = order

    def phase_c(self):
        self.st.inv_finish(self.recv, self.c_new)
        self.c, self.c_new = self.c_new, self.c

    def substep(self, advance: bool = True):
        """One substep including both exchanges; `advance` rotates the history afterwards (what
        TensorSolver::computeBuffer does between substeps)."""
        nf, nr = self.n_fs, self.n_fr
        self.phase_a()
        w0 = self.x_fwd.run(self.send2[:nf], self.recv2[:nr], async_op=True)
        self.phase_a2()
        w1 = self.x_fwd.run(self.send2[nf:], self.recv2[nr:], async_op=True)
        for w in (w0, w1):
            if w is not None:
                w.wait()
        self.phase_b()
        self.x_inv.run(self.send, self.recv)
        self.phase_c()
        if advance:
            self.advance_state()

    def step(self, dt: float, substeps: int):
        """One MOOSE time step: advanceState is a no-op while timeStep() <= 1 (TensorProblem.C:451-472),
        so every substep of the first step is AB1."""
        self.time_step += 1
        if self.time_step > 1:
            self.advance_state()
        self.sub_dt = dt / substeps
        for s in range(substeps):
            self.substep(advance=False)
            if s < substeps - 1 and self.time_step > 1:
                self.advance_state()
