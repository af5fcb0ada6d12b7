"""Oracle-backed rank-local slab stages (TEST INFRASTRUCTURE): the same buffer layouts as the
mrl_slab_* entry points (include/marlin_hip.h), computed with libTorch CPU ops.  Lets the CPU tests
drive marlin_amd.slab's host logic (exchange, history ring, phase order) over gloo without a GPU."""
import torch

from oracle import marlin_oracle as mo


class OracleSlabStages:
    def __init__(self, dim, shape, L, nranks, rank, half=None):
        self.dim, self.nranks, self.rank = dim, nranks, rank
        self.half = (dim == 3) if half is None else half
        self.n = list(shape) + [1] * (3 - dim)                     # [nx][ny][nz], nz = 1 in 2-D
        self.dom = mo.Domain(dim, list(shape), list(L), slab_c2c=not self.half)
        self.px = mo.partition_helper(self.n[0], [1] * nranks)      # reciprocal split (x)
        self.py = mo.partition_helper(self.n[1], [1] * nranks)      # real-space split (y)
        self.xb = [sum(self.px[:r]) for r in range(nranks)]
        self.yb = [sum(self.py[:r]) for r in range(nranks)]
        self.nzc = self.n[2] // 2 + 1 if self.half else self.n[2]
        self.nxl, self.nyl = self.px[rank], self.py[rank]
        self.real_shape = [self.n[0], self.nyl] + ([self.n[2]] if dim == 3 else [])
        self.real_begin = [0, self.yb[rank]] + ([0] if dim == 3 else [])
        self.recip_shape = [self.nxl, self.n[1]] + ([self.nzc] if dim == 3 else [])
        self.spec_pitch = self.recip_shape[-1]
        self.recip_begin = [self.xb[rank], 0] + ([0] if dim == 3 else [])
        self.device = torch.device("cpu")
        k = [a.reshape(-1) for a in self.dom.kaxis]
        kx = k[0][self.xb[rank]:self.xb[rank] + self.nxl].reshape(-1, 1, 1)
        ky = k[1].reshape(1, -1, 1)
        kz = (k[2] if dim == 3 else torch.zeros(1, dtype=torch.float64)).reshape(1, 1, -1)
        self.k2 = kx * kx + ky * ky + kz * kz
        self._mu = None

    def empty(self, n):
        return torch.zeros(n, dtype=torch.float64)

    def counts(self, forward):
        to_p = [self.px[p] * self.nyl * self.nzc for p in range(self.nranks)]
        from_p = [self.nxl * self.py[p] * self.nzc for p in range(self.nranks)]
        return (to_p, from_p) if forward else (from_p, to_p)

    # complex views of flat float64 buffers
    @staticmethod
    def _c(buf):
        return torch.view_as_complex(buf.reshape(-1, 2))

    def _fwd_local(self, real):
        r = real.reshape(self.n[0], self.nyl, self.n[2])
        z = torch.fft.rfft(r, dim=2) if self.half else torch.fft.fft(r.to(torch.complex128), dim=2)
        return torch.fft.fft(z, dim=0).contiguous()

    def _unpack(self, recv_c):
        out = torch.empty(self.nxl, self.n[1], self.nzc, dtype=torch.complex128)
        off = 0
        for p in range(self.nranks):
            cnt = self.nxl * self.py[p] * self.nzc
            out[:, self.yb[p]:self.yb[p] + self.py[p], :] = recv_c[off:off + cnt].reshape(self.nxl, self.py[p], self.nzc)
            off += cnt
        return out

    def _pack(self, dense, send_c):
        off = 0
        for p in range(self.nranks):
            cnt = self.nxl * self.py[p] * self.nzc
            send_c[off:off + cnt] = dense[:, self.yb[p]:self.yb[p] + self.py[p], :].reshape(-1)
            off += cnt

    def fwd_local(self, real_in, send):
        self._c(send)[:] = self._fwd_local(real_in).reshape(-1)

    def fwd_finish(self, recv, spec_out):
        self._c(spec_out)[:] = torch.fft.fft(self._unpack(self._c(recv)), dim=1).reshape(-1)

    def inv_local(self, spec_in, send):
        d = torch.fft.ifft(self._c(spec_in).reshape(self.nxl, self.n[1], self.nzc), dim=1)
        self._pack(d, self._c(send))

    def inv_finish(self, recv, real_out):
        d = torch.fft.ifft(self._c(recv).reshape(self.n[0], self.nyl, self.nzc), dim=0)
        if self.half:
            r = torch.fft.irfft(d, n=self.n[2], dim=2)
        else:
            r = torch.real(torch.fft.ifft(d, dim=2))
        real_out[:] = r.reshape(-1)

    # ---- Cahn-Hilliard substep pipelined over kz sub-blocks (layouts: include/marlin_hip.h) ------------
    def _sub(self, sub, nsub):
        ks = mo.partition_helper(self.nzc, [1] * nsub)
        return sum(ks[:sub]), ks[sub]

    # carry: 0 = the reference's data flow, 1 = same + ubar kept in cbar, 2 = c-hat taken from cbar, only mu travels forward
    def ch_counts(self, sub, nsub, forward, carry=0):
        k0, ksub = self._sub(sub, nsub)
        to_p = [self.px[p] * self.nyl * ksub for p in range(self.nranks)]
        from_p = [self.nxl * self.py[p] * ksub for p in range(self.nranks)]
        nf = 1 if carry == 2 else 2
        return ([nf * c for c in to_p], [nf * c for c in from_p]) if forward else (from_p, to_p)

    def ch_z_fwd(self, p, c_in, mu=None, carry=0):
        r = c_in.reshape(self.n[0], self.nyl, self.n[2])
        m = mo.mu_double_well(r, p.coef[0])
        zf = (lambda t: torch.fft.rfft(t, dim=2)) if self.half else (lambda t: torch.fft.fft(t.to(torch.complex128), dim=2))
        self._w = [zf(m)] if carry == 2 else [zf(r), zf(m)]

    def ch_x_fwd(self, sub, nsub, send, carry=0):
        k0, ksub = self._sub(sub, nsub)
        sc = self._c(send)
        off = 0
        xs = [torch.fft.fft(w[:, :, k0:k0 + ksub], dim=0) for w in self._w]
        for p in range(self.nranks):
            cnt = self.px[p] * self.nyl * ksub
            for f in range(len(xs)):
                sc[off:off + cnt] = xs[f][self.xb[p]:self.xb[p] + self.px[p]].reshape(-1)
                off += cnt

    def ch_kspace(self, p, sub, nsub, recv, send, Nnew, Nold, order, sub_dt, cbar=None, carry=0):
        k0, ksub = self._sub(sub, nsub)
        rc = self._c(recv)
        nf = 1 if carry == 2 else 2
        dense = [torch.empty(self.nxl, self.n[1], ksub, dtype=torch.complex128) for _ in range(nf)]
        off = 0
        for q in range(self.nranks):
            cnt = self.nxl * self.py[q] * ksub
            for f in range(nf):
                dense[f][:, self.yb[q]:self.yb[q] + self.py[q], :] = rc[off:off + cnt].reshape(self.nxl, self.py[q], ksub)
                off += cnt
        full = (self.nxl, self.n[1], self.nzc)
        if carry == 2:
            cb = self._c(cbar).reshape(full)[:, :, k0:k0 + ksub].clone()
        else:
            cb = torch.fft.fft(dense[0], dim=1)
        mb = torch.fft.fft(dense[-1], dim=1)
        k2 = self.k2[:, :, k0:k0 + ksub]
        Mbar = -k2 * p.mobility
        Lbar = k2 * k2 * p.kappa
        Nhat = Mbar * mb
        shape = (self.nxl, self.n[1], self.nzc)
        self._c(Nnew).reshape(shape)[:, :, k0:k0 + ksub] = Nhat
        ubar = cb + (sub_dt * mo.AB_BETA[order][0]) * Nhat
        for i in range(order):
            ubar += (sub_dt * mo.AB_BETA[order][i + 1]) * self._c(Nold[i]).reshape(shape)[:, :, k0:k0 + ksub]
        ubar /= (1.0 - sub_dt * Lbar)
        if carry:
            self._c(cbar).reshape(full)[:, :, k0:k0 + ksub] = ubar
        u = torch.fft.ifft(ubar, dim=1)
        sc = self._c(send)
        off = 0
        for q in range(self.nranks):
            cnt = self.nxl * self.py[q] * ksub
            sc[off:off + cnt] = u[:, self.yb[q]:self.yb[q] + self.py[q], :].reshape(-1)
            off += cnt

    def ch_x_inv(self, sub, nsub, recv):
        k0, ksub = self._sub(sub, nsub)
        if sub == 0:
            self._winv = torch.empty(self.n[0], self.nyl, self.nzc, dtype=torch.complex128)
        rc = self._c(recv)
        d = torch.empty(self.n[0], self.nyl, ksub, dtype=torch.complex128)
        off = 0
        for p in range(self.nranks):
            cnt = self.px[p] * self.nyl * ksub
            d[self.xb[p]:self.xb[p] + self.px[p]] = rc[off:off + cnt].reshape(self.px[p], self.nyl, ksub)
            off += cnt
        self._winv[:, :, k0:k0 + ksub] = torch.fft.ifft(d, dim=0)

    def ch_z_inv_fwd(self, p, mu=None, carry=0):
        r = torch.empty(self.n[0] * self.nyl * self.n[2], dtype=torch.float64)
        self.ch_z_inv(r)
        self.ch_z_fwd(p, r, mu, carry)

    def ch_z_inv(self, c_out):
        if self.half:
            r = torch.fft.irfft(self._winv, n=self.n[2], dim=2)
        else:
            r = torch.real(torch.fft.ifft(self._winv, dim=2))
        c_out[:] = r.reshape(-1)
