"""Thin Python plumbing above the C ABI: device memory comes from torch, everything else is
`libmarlin_hip.so`.  No arithmetic happens here."""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import MrlChParams, MrlDomain, MrlMechParams, MrlMechStats

SPECTRUM_HALF, SPECTRUM_FULL = 0, 1
FE_DOUBLE_WELL, FE_PFHUB, FE_PARSED = 0, 1, 2


class MarlinHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[mrl {code}] {msg}")
        self.code = code
        self.message = msg


def _ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError("device tensor required (the HIP path has no CPU fallback)")
    if not t.is_contiguous():
        raise ValueError("contiguous tensor required")
    return C.c_void_p(t.data_ptr())


def reciprocal_axis(n: int, dx: float, rfft: bool) -> List[float]:
    lib = _lib.load()
    cnt = n // 2 + 1 if rfft else n
    out = (C.c_double * cnt)()
    rc = lib.mrl_reciprocal_axis(n, dx, 1 if rfft else 0, out)
    if rc != 0:
        raise MarlinHipError(rc, lib.mrl_last_error(None).decode())
    return list(out)


def pencil_factors(nranks: int, n: Sequence[int]):
    """DomainAction::partitionPencils' process grid (py, pz) for `nranks` ranks on an nx x ny x nz grid (host only); raises
    MarlinHipError with the reference's message when no factorisation fits"""
    lib = _lib.load()
    nn = (C.c_int64 * 3)(*[int(v) for v in n])
    py, pz = C.c_int32(), C.c_int32()
    rc = lib.mrl_pencil_factors(nranks, nn, C.byref(py), C.byref(pz))
    if rc != 0:
        raise MarlinHipError(rc, lib.mrl_last_error(None).decode())
    return py.value, pz.value


def pencil_layout(nranks: int, rank: int, n: Sequence[int]) -> dict:
    """host only: blocks and message sizes (complex elements per peer) of the staged exchanges of an FFT_PENCIL job (mrl_pencil_layout)"""
    lib = _lib.load()
    nn = (C.c_int64 * 3)(*[int(v) for v in n])
    rn, rb, kn, kb = ((C.c_int64 * 3)() for _ in range(4))
    cnt = [(C.c_int64 * nranks)() for _ in range(4)]
    rc = lib.mrl_pencil_layout(nranks, rank, nn, rn, rb, kn, kb, *cnt)
    if rc != 0:
        raise MarlinHipError(rc, lib.mrl_last_error(None).decode())
    return {"real_shape": list(rn), "real_begin": list(rb), "recip_shape": list(kn), "recip_begin": list(kb),
            "stage1_send": list(cnt[0]), "stage1_recv": list(cnt[1]), "stage2_send": list(cnt[2]), "stage2_recv": list(cnt[3])}


def partition(total: int, nranks: int, weights: Optional[Sequence[int]] = None) -> List[int]:
    lib = _lib.load()
    out = (C.c_int64 * nranks)()
    w = (C.c_int64 * nranks)(*weights) if weights is not None else None
    rc = lib.mrl_partition(total, nranks, w, out)
    if rc != 0:
        raise MarlinHipError(rc, lib.mrl_last_error(None).decode())
    return list(out)


def ch_params(family=FE_DOUBLE_WELL, coef=(0.1,), mobility=0.2, kappa=-0.001, parsed=None) -> MrlChParams:
    """parsed: a ParsedCompute of the free energy differentiated w.r.t. its single input (family is then FE_PARSED)"""
    p = MrlChParams()
    if parsed is not None:
        family = FE_PARSED
        p.parsed = parsed.h
        p._keepalive = parsed
    p.family = family
    for i, v in enumerate(coef):
        p.coef[i] = v
    p.mobility = mobility
    p.kappa = kappa
    return p


TRANSPORT_AUTO, TRANSPORT_PEER_STORE, TRANSPORT_PEER_COPY, TRANSPORT_RCCL = 0, 1, 2, 3
TRANSPORT_NAMES = {1: "peer_store", 2: "peer_copy", 3: "rccl"}
OPT_EXPERIMENT, OPT_SLAB_NSUB, OPT_SLAB_CARRY, OPT_VERIFY_EXCHANGE, OPT_VERIFY_MISMATCHES, OPT_CACHE_CHUNK_MB = 0, 1, 2, 3, 4, 5


class Comm:
    """One mrl_comm: the library-owned multi-GPU transport of a job of `nranks` processes on one node (one process per GPU).
    `name` identifies the job (all ranks pass the same string; it names the POSIX shared-memory bootstrap segment)."""

    def __init__(self, name: str, nranks: int, rank: int, device: Optional[int] = None, transport: int = TRANSPORT_AUTO,
                 timeout: Optional[float] = None):
        self.lib = _lib.load()
        if device is None:
            device = torch.cuda.current_device()
        h = C.c_void_p()
        rc = self.lib.mrl_comm_create(C.byref(h), name.encode(), nranks, rank, device, transport)
        if rc != 0:
            raise MarlinHipError(rc, self.lib.mrl_comm_last_error(None).decode())
        self.h = h
        self.nranks, self.rank = nranks, rank
        if timeout is not None:
            self._check(self.lib.mrl_comm_set_timeout(h, float(timeout)))

    def _check(self, rc):
        if rc != 0:
            raise MarlinHipError(rc, self.lib.mrl_comm_last_error(self.h).decode())

    @property
    def transport(self) -> int:
        return self.lib.mrl_comm_transport(self.h)

    def set_transport(self, transport: int):
        self._check(self.lib.mrl_comm_set_transport(self.h, transport))

    def set_timeout(self, seconds: float):
        self._check(self.lib.mrl_comm_set_timeout(self.h, float(seconds)))

    def reset_error(self):
        self._check(self.lib.mrl_comm_reset_error(self.h))

    def barrier(self):
        self._check(self.lib.mrl_comm_barrier(self.h))

    def allreduce(self, values: Sequence[float], op: int = 0) -> List[float]:
        n = len(values)
        arr = (C.c_double * max(1, n))(*values)
        self._check(self.lib.mrl_comm_allreduce(self.h, arr, n, op))
        return list(arr)[:n]

    def stats(self):
        n, b = C.c_int64(), C.c_double()
        self._check(self.lib.mrl_comm_stats(self.h, C.byref(n), C.byref(b)))
        return {"exchanges": n.value, "bytes_sent": b.value}

    def rccl_preflight(self) -> int:
        """COLLECTIVE: the RCCL bring-up stage by stage (library, unique-id broadcast, placement, ncclCommInitRank); returns the
        status code (0 ready, -2 unavailable on this placement, -6 failed) instead of raising: describe() has the details"""
        return int(self.lib.mrl_comm_rccl_preflight(self.h))

    def describe(self) -> dict:
        """what the communicator runs on (HIP runtime / library actually mapped, RCCL library and the rank count it reports, ...)"""
        import json
        buf = C.create_string_buffer(4096)
        self._check(self.lib.mrl_comm_describe(self.h, buf, len(buf)))
        return json.loads(buf.value.decode())

    def close(self):
        """collective.  Contexts still attached lose their exchange pipelines here (mrl_comm_destroy tears them down and detaches
        them); they stay valid for rank-local work and for Context.close()."""
        if getattr(self, "h", None):
            self.lib.mrl_comm_destroy(self.h)
            self.h = None


class Context:
    """One mrl_ctx (= one DomainAction on one rank / GPU)."""

    def __init__(self, dim: int, n: Sequence[int], mx: Sequence[float], mn: Sequence[float] = (0.0, 0.0, 0.0),
                 nranks: int = 1, rank: int = 0, weights: Optional[Sequence[int]] = None,
                 spectrum: int = SPECTRUM_HALF, device: Optional[int] = None, use_torch_stream: bool = True,
                 slab: bool = False, dense_spectra: bool = False, pencil: bool = False):
        self.lib = _lib.load()
        d = MrlDomain()
        d.dim = dim
        for i in range(3):
            d.n[i] = int(n[i]) if i < dim else 1
            d.min[i] = float(mn[i]) if i < len(mn) else 0.0
            d.max[i] = float(mx[i]) if i < dim else 1.0
        if device is None:
            device = torch.cuda.current_device()
        d.device = device
        d.nranks, d.rank = nranks, rank
        self._weights = (C.c_int64 * nranks)(*weights) if weights is not None else None
        d.weights = self._weights
        d.spectrum = spectrum
        d.stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream) if use_torch_stream else None
        # MRL_FLAG_OWN_STREAM | _SLAB | _DENSE_SPECTRA | _PENCIL
        d.flags = (0 if use_torch_stream else 1) | (2 if slab else 0) | (4 if dense_spectra else 0) | (8 if pencil else 0)
        h = C.c_void_p()
        rc = self.lib.mrl_ctx_create(C.byref(h), C.byref(d))
        if rc != 0:
            raise MarlinHipError(rc, self.lib.mrl_last_error(None).decode())
        self.h = h
        self.dim = dim
        self.device = torch.device("cuda", device)
        rn, rb, kn, kb = ((C.c_int64 * 3)() for _ in range(4))
        self._check(self.lib.mrl_local_shape(h, rn, rb, kn, kb))
        self.real_shape = [rn[i] for i in range(dim)]
        self.real_begin = [rb[i] for i in range(dim)]
        self.recip_shape = [kn[i] for i in range(dim)]
        self.recip_begin = [kb[i] for i in range(dim)]
        self.nranks, self.rank = nranks, rank
        self.pencil_grid = None
        if pencil:   # (py, pz) of DomainAction::partitionPencils
            py, pz = C.c_int32(), C.c_int32()
            self._check(self.lib.mrl_pencil_grid(h, C.byref(py), C.byref(pz)))
            self.pencil_grid = (py.value, pz.value)

    def close(self):
        if getattr(self, "h", None):
            self.lib.mrl_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise MarlinHipError(rc, self.lib.mrl_last_error(self.h).decode())

    def attach_comm(self, comm: Optional["Comm"]):
        """slab contexts: hand the exchanges to the library (mrl_ch_substeps, mrl_fft_*, mrl_mech_newton_cg, reductions become global)"""
        self._check(self.lib.mrl_ctx_attach_comm(self.h, comm.h if comm is not None else None))
        self._comm = comm   # (keeps the communicator object alive as long as this context refers to it)
        self._comm = comm

    def set_option(self, option: int, value: int):
        self._check(self.lib.mrl_ctx_set_option(self.h, option, int(value)))

    def get_option(self, option: int) -> int:
        return int(self.lib.mrl_ctx_get_option(self.h, option))

    @property
    def spec_pitch(self) -> int:
        """last-axis pitch (complex elements) of the rank-local spectral arrays of the slab Cahn-Hilliard pipeline"""
        return int(self.lib.mrl_slab_ch_spec_pitch(self.h))

    # ---- helpers
    def empty_real(self, *value_dims, batch_first: Optional[int] = None):
        shape = ([batch_first] if batch_first else []) + list(self.real_shape) + list(value_dims)
        return torch.empty(shape, dtype=torch.float64, device=self.device)

    def empty_spec(self, *value_dims, batch_first: Optional[int] = None):
        shape = ([batch_first] if batch_first else []) + list(self.recip_shape) + list(value_dims)
        return torch.empty(shape, dtype=torch.complex128, device=self.device)

    @property
    def spec_elems(self) -> int:
        """complex elements of one solver-private spectral array (Nhat history, cbar): mrl_ch_spec_elems"""
        return int(self.lib.mrl_ch_spec_elems(self.h))

    def spec_layout(self):
        """(plane_pitch, row_pitch) of the solver-private spectral arrays: element (ix, iy, kz) at ix*plane + iy*row + kz"""
        pl, rw = C.c_int64(), C.c_int64()
        self._check(self.lib.mrl_ch_spec_layout(self.h, C.byref(pl), C.byref(rw)))
        return pl.value, rw.value

    def empty_hist(self, zero: bool = False) -> torch.Tensor:
        """One array of the Cahn-Hilliard solver (an Nhat history entry, cbar / the carried spectrum) in its private layout
        (mrl_ch_spec_elems / mrl_ch_spec_layout: x planes padded on fused fast-path shapes), returned as a strided VIEW of the
        reciprocal shape -- its data pointer is what the library takes, `.cpu()` / `.contiguous()` give the dense values."""
        n = self.spec_elems
        buf = (torch.zeros if zero else torch.empty)(n, dtype=torch.complex128, device=self.device)
        return self.hist_view(buf)

    def hist_view(self, buf: torch.Tensor) -> torch.Tensor:
        plane, row = self.spec_layout()
        shp = list(self.recip_shape)
        if self.dim == 3:
            return torch.as_strided(buf, shp, (plane, row, 1))
        if self.dim == 2:   # serial 2-D contexts are [1][nx][ny']: one plane
            return torch.as_strided(buf, shp, (row, 1))
        return torch.as_strided(buf, shp, (1,))

    def reciprocal_axis(self, axis: int) -> torch.Tensor:
        n = self.recip_shape[axis]
        out = (C.c_double * n)()
        self._check(self.lib.mrl_ctx_reciprocal_axis(self.h, axis, out, n))
        return torch.tensor(list(out), dtype=torch.float64)

    def sync(self):
        self._check(self.lib.mrl_sync(self.h))

    # ---- FFT service (DomainAction::fft / ifft)
    def fft(self, t: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """t: [grid...] or value-major [grid..., *values]"""
        vals = list(t.shape[self.dim:])
        batch = 1
        for v in vals:
            batch *= v
        assert list(t.shape[:self.dim]) == self.real_shape, (t.shape, self.real_shape)
        if out is None:
            out = self.empty_spec(*vals)
        self._check(self.lib.mrl_fft_r2c(self.h, _ptr(t), _ptr(out), batch, 1 if batch > 1 else 0))
        return out

    def ifft(self, t: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        vals = list(t.shape[self.dim:])
        batch = 1
        for v in vals:
            batch *= v
        assert list(t.shape[:self.dim]) == self.recip_shape, (t.shape, self.recip_shape)
        if out is None:
            out = self.empty_real(*vals)
        self._check(self.lib.mrl_fft_c2r(self.h, _ptr(t), _ptr(out), batch, 1 if batch > 1 else 0))
        return out

    def fft_fields(self, t: torch.Tensor) -> torch.Tensor:
        """field-major batch: t [B, grid...] -> [B, recip...]"""
        B = t.shape[0]
        out = self.empty_spec(batch_first=B)
        self._check(self.lib.mrl_fft_r2c(self.h, _ptr(t), _ptr(out), B, 0))
        return out

    def ifft_fields(self, t: torch.Tensor) -> torch.Tensor:
        B = t.shape[0]
        out = self.empty_real(batch_first=B)
        self._check(self.lib.mrl_fft_c2r(self.h, _ptr(t), _ptr(out), B, 0))
        return out

    # ---- Cahn-Hilliard
    def ch_mu(self, p: MrlChParams, c: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            out = torch.empty_like(c)
        self._check(self.lib.mrl_ch_mu(self.h, C.byref(p), _ptr(c), _ptr(out), c.numel()))
        return out

    def ch_substep(self, p: MrlChParams, c_in, c_out, Nhat_new, Nhat_old: Sequence[torch.Tensor], order: int,
                   sub_dt: float, cbar=None, mu=None, carry: int = 0):
        """carry: 0 = the reference's data flow; 1 = same, ubar kept in `cbar`; 2 = `cbar` holds c-hat (the previous ubar) on
        entry and receives the new ubar (spectral carry-over, include/marlin_hip.h)"""
        arr = (C.c_void_p * max(1, len(Nhat_old)))(*[self._hist_ptr(t).value for t in Nhat_old])
        self._check(self.lib.mrl_ch_substep(self.h, C.byref(p), _ptr(c_in), _ptr(c_out), self._hist_ptr(Nhat_new), arr, order,
                                            sub_dt, self._hist_ptr(cbar), _ptr(mu), carry))

    def _hist_ptr(self, t):
        """device pointer of a solver-private spectral array (empty_hist); refuses arrays that are too small for the layout --
        the kernels would write past their end"""
        if t is None:
            return None
        if not t.is_cuda:
            raise ValueError("device tensor required (the HIP path has no CPU fallback)")
        have = t.untyped_storage().nbytes() - t.storage_offset() * t.element_size()
        if have < 16 * self.spec_elems:
            raise ValueError(f"history / cbar arrays need mrl_ch_spec_elems = {self.spec_elems} complex values (Context.empty_hist), "
                             f"this one holds {have // 16}")
        return C.c_void_p(t.data_ptr())

    def ch_substeps(self, p: MrlChParams, c_in, c_out, ring: Sequence[torch.Tensor], head: int, n_old: int, predictor_order: int,
                    count: int, advance: bool, sub_dt: float, mu=None, dt_changed: bool = False):
        """`count` substeps in one call (TensorSolver::computeBuffer's loop) -> (head, n_old) after the call; the newest Nhat is
        in ring[(head + 1) % len(ring)]"""
        arr = (C.c_void_p * len(ring))(*[self._hist_ptr(t).value for t in ring])
        h, n = C.c_int32(head), C.c_int32(n_old)
        self._check(self.lib.mrl_ch_substeps(self.h, C.byref(p), _ptr(c_in), _ptr(c_out), arr, len(ring), C.byref(h), C.byref(n),
                                             predictor_order, count, (1 if advance else 0) | (2 if dt_changed else 0), sub_dt, _ptr(mu)))
        return h.value, n.value

    # ---- fp32 form of the fused Cahn-Hilliard solver (mrl_ch_substeps_f32; serial 3-D contexts of a few planned extents)
    @property
    def spec_elems_f32(self) -> int:
        return int(self.lib.mrl_ch_spec_elems_f32(self.h))

    def empty_hist_f32(self, zero: bool = False) -> torch.Tensor:
        """a history array of the fp32 solver (complex64, solver-private layout) as a strided view of the reciprocal shape"""
        n = self.spec_elems_f32
        if n == 0:
            raise MarlinHipError(-2, "mrl_ch_substeps_f32 does not cover this context (mrl_ch_spec_elems_f32 = 0)")
        pl, rw = C.c_int64(), C.c_int64()
        self._check(self.lib.mrl_ch_spec_layout_f32(self.h, C.byref(pl), C.byref(rw)))
        buf = (torch.zeros if zero else torch.empty)(n, dtype=torch.complex64, device=self.device)
        return torch.as_strided(buf, list(self.recip_shape), (pl.value, rw.value, 1))

    def ch_substeps_f32(self, p: MrlChParams, c_in, c_out, ring: Sequence[torch.Tensor], head: int, n_old: int, predictor_order: int,
                        count: int, advance: bool, sub_dt: float, dt_changed: bool = False):
        assert c_in.dtype == torch.float32 and c_out.dtype == torch.float32 and all(t.dtype == torch.complex64 for t in ring)
        need = 8 * self.spec_elems_f32
        for t in ring:
            if t.untyped_storage().nbytes() - t.storage_offset() * t.element_size() < need:
                raise ValueError("history arrays of the fp32 solver need mrl_ch_spec_elems_f32 complex64 values (Context.empty_hist_f32)")
        arr = (C.c_void_p * len(ring))(*[t.data_ptr() for t in ring])
        h, n = C.c_int32(head), C.c_int32(n_old)
        self._check(self.lib.mrl_ch_substeps_f32(self.h, C.byref(p), _ptr(c_in), _ptr(c_out), arr, len(ring), C.byref(h), C.byref(n),
                                                 predictor_order, count, (1 if advance else 0) | (2 if dt_changed else 0), sub_dt))
        return h.value, n.value

    def kspace_abm(self, out, ubar0, N: Sequence[torch.Tensor], coef: Sequence[float], L, dt: float):
        n = len(N)
        arr = (C.c_void_p * max(1, n))(*[t.data_ptr() for t in N])
        cf = (C.c_double * max(1, n))(*coef)
        self._check(self.lib.mrl_kspace_abm(self.h, _ptr(out), _ptr(ubar0), arr, cf, n, _ptr(L), dt, ubar0.numel()))

    COUPLED_L_AS_WRITTEN, COUPLED_COMPLEX_RHS = 1, 2

    def kspace_coupled(self, out: Sequence[torch.Tensor], ubar0: Sequence[torch.Tensor], N: Sequence[Sequence[torch.Tensor]],
                       coef: Sequence[Sequence[float]], L: Sequence[Sequence[Optional[torch.Tensor]]], dt: float, flags: int = 0):
        """AdamsBashforthMoultonCoupled's per-k dense solve (see mrl_kspace_coupled); L[i][j] = real array or None"""
        nv = len(out)
        flatN = [t for row in N for t in row]
        flatc = [float(c) for row in coef for c in row]
        outs = (C.c_void_p * nv)(*[t.data_ptr() for t in out])
        u0 = (C.c_void_p * nv)(*[t.data_ptr() for t in ubar0])
        arr = (C.c_void_p * max(1, len(flatN)))(*[t.data_ptr() for t in flatN])
        cf = (C.c_double * max(1, len(flatc)))(*flatc)
        nt = (C.c_int32 * nv)(*[len(row) for row in N])
        Lp = (C.c_void_p * (nv * nv))(*[(L[i][j].data_ptr() if L[i][j] is not None else None) for i in range(nv) for j in range(nv)])
        self._check(self.lib.mrl_kspace_coupled(self.h, nv, outs, u0, arr, cf, nt, Lp, dt, flags, ubar0[0].numel()))

    def secant_begin(self, u, N, L, sub_dt: float, dt_epsilon: float):
        """-> (R0, guess, |R0|)  (SecantSolver.C:79-101)"""
        R0, guess = torch.empty_like(u), torch.empty_like(u)
        ss = (C.c_double * 1)()
        self._check(self.lib.mrl_secant_begin(self.h, _ptr(u), _ptr(N), _ptr(L), sub_dt, dt_epsilon, _ptr(R0), _ptr(guess), ss,
                                              u.numel()))
        return R0, guess, math.sqrt(ss[0])

    def secant_iterate(self, u, N, L, u_old, u_prev, R_prev, sub_dt: float, damping: float = 1.0):
        """-> (u_new, |R|, |du|); R_prev is replaced by the new residual in place  (SecantSolver.C:121-140)"""
        u_new = torch.empty_like(u)
        ss = (C.c_double * 2)()
        self._check(self.lib.mrl_secant_iterate(self.h, _ptr(u), _ptr(N), _ptr(L), _ptr(u_old), _ptr(u_prev), _ptr(R_prev), sub_dt,
                                                damping, _ptr(u_new), ss, u.numel()))
        return u_new, math.sqrt(ss[0]), math.sqrt(ss[1])

    def histogram(self, a: torch.Tensor, edges: Sequence[float]):
        """TensorHistogram: counts per bin [edge_i, edge_i+1), last bin closed"""
        nb = len(edges) - 1
        e = (C.c_double * (nb + 1))(*[float(v) for v in edges])
        out = (C.c_int64 * nb)()
        self._check(self.lib.mrl_histogram(self.h, _ptr(a), a.numel(), e, nb, out))
        return list(out)

    # ---- BroydenSolver building blocks (field-major state arrays owned by the caller)
    @staticmethod
    def _pp(ts):
        return (C.c_void_p * max(1, len(ts)))(*[(t.data_ptr() if t is not None else None) for t in ts])

    def broyden_init(self, nvar: int, factor: float, nspec: int) -> torch.Tensor:
        M = torch.empty(nvar * nvar, nspec, dtype=torch.complex128, device=self.device)
        self._check(self.lib.mrl_broyden_init(self.h, nvar, factor, _ptr(M), nspec))
        return M

    def broyden_residual(self, u, N, L, u_old, sub_dt: float):
        nv, n = len(u), u[0].numel()
        R = torch.empty(nv, n, dtype=torch.complex128, device=self.device)
        ss = (C.c_double * 1)()
        self._check(self.lib.mrl_broyden_residual(self.h, nv, self._pp(u), self._pp(N), self._pp(L),
                                                  self._pp(u_old) if u_old is not None else None, sub_dt, _ptr(R), ss, n))
        return R, math.sqrt(ss[0])

    def broyden_predict(self, M, R, u, step: float = 0.5):
        nv, n = len(u), u[0].numel()
        S = torch.empty(nv, n, dtype=torch.complex128, device=self.device)
        out = [torch.empty_like(t) for t in u]
        self._check(self.lib.mrl_broyden_predict(self.h, nv, _ptr(M), _ptr(R), self._pp(u), step, _ptr(S), self._pp(out), n))
        return S, out

    def broyden_update(self, M, R, S, u, N, L, u_old, sub_dt: float) -> float:
        nv, n = len(u), u[0].numel()
        ss = (C.c_double * 1)()
        self._check(self.lib.mrl_broyden_update(self.h, nv, _ptr(M), _ptr(R), _ptr(S), self._pp(u), self._pp(N), self._pp(L),
                                                self._pp(u_old), sub_dt, ss, n))
        return math.sqrt(ss[0])

    # ---- de Geus mechanics (value-major [grid..., D, D] fields)
    def gamma_apply(self, A: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            out = torch.empty_like(A)
        self._check(self.lib.mrl_gamma_apply(self.h, _ptr(A), _ptr(out)))
        return out

    def mech_stress(self, F, K, mu, out=None):
        if out is None:
            out = torch.empty_like(F)
        self._check(self.lib.mrl_mech_stress(self.h, _ptr(F), _ptr(K), _ptr(mu), _ptr(out)))
        return out

    def mech_displacements(self, F: torch.Tensor) -> torch.Tensor:
        """ComputeDisplacements: [grid..., D, D] -> node values [(n+1)..., D]"""
        out = torch.empty([n + 1 for n in F.shape[:self.dim]] + [self.dim], dtype=torch.float64, device=F.device)
        self._check(self.lib.mrl_mech_displacements(self.h, _ptr(F), _ptr(out)))
        return out

    def mech_von_mises(self, stress: torch.Tensor) -> torch.Tensor:
        out = torch.empty(stress.shape[:self.dim], dtype=torch.float64, device=stress.device)
        self._check(self.lib.mrl_mech_von_mises(self.h, _ptr(stress), _ptr(out)))
        return out

    def qs_elasticity(self, cbar: torch.Tensor, mu: float, lam: float, e0: float):
        """FFTQuasistaticElasticity: the three displacement fields of the homogeneous elastic equilibrium with eigenstrain e0*c"""
        disp = [torch.empty(self.real_shape, dtype=torch.float64, device=cbar.device) for _ in range(3)]
        arr = (C.c_void_p * 3)(*[d.data_ptr() for d in disp])
        self._check(self.lib.mrl_qs_elasticity(self.h, _ptr(cbar), mu, lam, e0, arr))
        return disp

    def elastic_chemical_potential(self, cbar: torch.Tensor, disp, mu: float, lam: float, e0: float) -> torch.Tensor:
        """FFTElasticChemicalPotential: reciprocal-space elastic contribution to the chemical potential"""
        out = self.empty_spec()
        arr = (C.c_void_p * 3)(*[d.data_ptr() for d in disp])
        self._check(self.lib.mrl_elastic_chemical_potential(self.h, _ptr(cbar), arr, mu, lam, e0, _ptr(out)))
        return out

    def mech_tangent_apply(self, F, K, mu, dF, out=None):
        if out is None:
            out = torch.empty_like(F)
        self._check(self.lib.mrl_mech_tangent_apply(self.h, _ptr(F), _ptr(K), _ptr(mu), _ptr(dF), _ptr(out)))
        return out

    def mech_newton_cg(self, F, K, mu, applied, l_tol=1e-2, l_max_its=0, nl_rel_tol=1e-5, nl_abs_tol=1e-8,
                       nl_max_its=100, Fnew=None, P=None):
        """FFTMechanics::computeBuffer; returns (Fnew, P, stats dict)."""
        prm = MrlMechParams()
        prm.l_tol, prm.l_max_its = l_tol, l_max_its
        prm.nl_rel_tol, prm.nl_abs_tol, prm.nl_max_its = nl_rel_tol, nl_abs_tol, nl_max_its
        if Fnew is None:
            Fnew = torch.empty_like(F)
        if P is None:
            P = torch.empty_like(F)
        st = MrlMechStats()
        self._check(self.lib.mrl_mech_newton_cg(self.h, C.byref(prm), _ptr(F), _ptr(K), _ptr(mu), _ptr(applied),
                                                _ptr(Fnew), _ptr(P), C.byref(st)))
        stats = {"newton_its": st.newton_its, "cg_its": [st.cg_its[i] for i in range(min(st.newton_its, 64))],
                 "cg_its_total": st.cg_its_total, "anorm": st.last_anorm, "rnorm": st.last_rnorm, "Fn": st.Fn}
        return Fnew, P, stats

    def mech_small_strain(self, K, mu, E, l_tol=1e-2, l_max_its=0):
        """small-strain linear-elastic RVE (mrl_mech_small_strain): returns (eps, sigma, stats dict); E = [dim, dim] device tensor"""
        prm = MrlMechParams()
        prm.l_tol, prm.l_max_its = l_tol, l_max_its
        prm.nl_rel_tol, prm.nl_abs_tol, prm.nl_max_its = 0.0, 0.0, 1
        shape = list(K.shape) + [self.dim, self.dim]
        eps = torch.empty(shape, dtype=torch.float64, device=K.device)
        sigma = torch.empty_like(eps)
        st = MrlMechStats()
        self._check(self.lib.mrl_mech_small_strain(self.h, C.byref(prm), _ptr(K), _ptr(mu), _ptr(E), _ptr(eps), _ptr(sigma), C.byref(st)))
        return eps, sigma, {"cg_its": st.cg_its[0], "anorm": st.last_anorm, "Fn": st.Fn}

    # ---- reductions (synchronous)
    def _scalar(self, fn, *args):
        out = C.c_double()
        self._check(fn(self.h, *args, C.byref(out)))
        return float(out.value)

    def dot(self, a, b):
        return self._scalar(self.lib.mrl_dot, _ptr(a), _ptr(b), a.numel())

    def norm2(self, a):
        return self._scalar(self.lib.mrl_norm2, _ptr(a), a.numel())

    def sum(self, a):
        return self._scalar(self.lib.mrl_sum, _ptr(a), a.numel())

    def average(self, a: torch.Tensor) -> torch.Tensor:
        """DomainAction::average of a value-major field -> host tensor of the value shape"""
        vals = list(a.shape[self.dim:])
        ncomp = 1
        for v in vals:
            ncomp *= v
        out = (C.c_double * ncomp)()
        self._check(self.lib.mrl_average(self.h, _ptr(a), ncomp, out))
        return torch.tensor(list(out), dtype=torch.float64).reshape(vals)

    # ---- timing
    def timer_start(self):
        self._check(self.lib.mrl_timer_start(self.h))

    def timer_stop(self) -> float:
        ms = C.c_float()
        self._check(self.lib.mrl_timer_stop(self.h, C.byref(ms)))
        return float(ms.value)

    def set_profiling(self, on: bool):
        self._check(self.lib.mrl_set_profiling(self.h, 1 if on else 0))

    def get_timing(self) -> dict:
        """mrl_get_timing: the profile summed over its kernel classes"""
        from ._lib import MrlTiming
        t = MrlTiming()
        self._check(self.lib.mrl_get_timing(self.h, C.byref(t)))
        return {"kernel_classes": t.kernel_classes, "launches": t.launches, "device_ms": t.device_ms,
                "algorithmic_bytes": t.algorithmic_bytes, "dominant": t.dominant.decode() if t.dominant else None,
                "dominant_ms": t.dominant_ms}

    def axpy(self, a: float, x: torch.Tensor, y: torch.Tensor):
        """y += a x (mrl_axpy)"""
        self._check(self.lib.mrl_axpy(self.h, float(a), x.data_ptr(), y.data_ptr(), x.numel()))

    def get_profile(self):
        res = []
        slot = 0
        while True:
            name = C.c_char_p()
            ms = C.c_double()
            cnt = C.c_int64()
            nbytes = C.c_double()
            rc = self.lib.mrl_get_profile(self.h, slot, C.byref(name), C.byref(ms), C.byref(cnt), C.byref(nbytes))
            if rc != 0:
                break
            res.append({"kernel": name.value.decode(), "ms": ms.value, "launches": cnt.value,
                        "bytes_per_launch": nbytes.value})
            slot += 1
        return res


class ParsedCompute:
    """ParsedCompute (src/tensor_computes/ParsedCompute.C:50-265): expression -> derivatives -> fused HIP kernel.
    ctx=None parses / differentiates / simplifies only (no GPU): `.tree` is the simplified expression."""

    def __init__(self, ctx: Optional[Context], expression: str, inputs: Sequence[str] = (), complex_inputs: Sequence[str] = (),
                 constants: Optional[dict] = None, derivatives: Sequence[str] = (), extra_symbols: bool = False,
                 reciprocal: bool = False):
        self.lib = _lib.load()
        self.ctx = ctx
        constants = constants or {}
        names = [n.encode() for n in inputs]
        cn = [n.encode() for n in constants]
        dn = [n.encode() for n in derivatives]
        arr = lambda items: (C.c_char_p * max(1, len(items)))(*items)
        flags = (C.c_int * max(1, len(inputs)))(*[1 if n in complex_inputs else 0 for n in inputs])
        cv = (C.c_double * max(1, len(constants)))(*[float(v) for v in constants.values()])
        h = C.c_void_p()
        rc = self.lib.mrl_parsed_create(ctx.h if ctx else None, C.byref(h), expression.encode(), len(names), arr(names), flags,
                                        len(cn), arr(cn), cv, len(dn), arr(dn), 1 if extra_symbols else 0,
                                        1 if reciprocal else 0)
        if rc != 0:
            raise MarlinHipError(rc, self.lib.mrl_last_error(ctx.h if ctx else None).decode())
        self.h = h
        self.inputs = list(inputs)
        self.reciprocal = reciprocal
        self.is_complex = bool(self.lib.mrl_parsed_is_complex(h))
        self.tree = self.lib.mrl_parsed_string(h).decode()
        self.source = self.lib.mrl_parsed_source(h).decode()

    def __call__(self, *tensors: torch.Tensor, out: Optional[torch.Tensor] = None, count: Optional[int] = None,
                 time: float = 0.0) -> torch.Tensor:
        assert self.ctx is not None, "created without a context"
        assert len(tensors) == len(self.inputs)
        grid = None
        if count is None:
            if tensors:
                count = tensors[0].numel()
            else:           # no inputs: the context's local (reciprocal) grid
                grid = self.ctx.recip_shape if self.reciprocal else self.ctx.real_shape
                count = int(math.prod(grid))
        if out is None:
            shape = tensors[0].shape if tensors else (tuple(grid) if grid else (count,))
            out = torch.empty(shape, dtype=torch.complex128 if self.is_complex else torch.float64, device=self.ctx.device)
        arr = (C.c_void_p * max(1, len(tensors)))(*[t.data_ptr() for t in tensors])
        self.ctx._check(self.lib.mrl_parsed_eval(self.h, arr, _ptr(out), count, time))
        return out

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.mrl_parsed_destroy(self.h)
                self.h = None
        except Exception:
            pass


class H5File:
    """HDF5 container of XDMFTensorOutput (mrl_h5_*: host code, no GPU needed): datasets "<name>.<frame>" in the root group."""

    _DTYPES = {"float64": 0, "float32": 1, "int32": 2, "int64": 3}

    def __init__(self, path: str):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        rc = self._lib.mrl_h5_create(str(path).encode(), C.byref(self._h))
        if rc != 0:
            raise MarlinHipError(rc, f"cannot create {path}")

    def _check(self, rc):
        if rc != 0:
            raise MarlinHipError(rc, self._lib.mrl_h5_last_error(self._h).decode())

    def write(self, name: str, array):
        import numpy as np
        a = np.ascontiguousarray(array)
        if a.dtype.name not in self._DTYPES:
            raise ValueError(f"unsupported dtype {a.dtype}")
        dims = (C.c_int64 * a.ndim)(*a.shape)
        self._check(self._lib.mrl_h5_write(self._h, name.encode(), self._DTYPES[a.dtype.name], a.ndim, dims, a.ctypes.data_as(C.c_void_p)))

    def flush(self):
        self._check(self._lib.mrl_h5_flush(self._h))

    def close(self):
        if self._h:
            rc = self._lib.mrl_h5_close(self._h)
            self._h = C.c_void_p()
            if rc != 0:
                raise MarlinHipError(rc, "closing the HDF5 file failed")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
