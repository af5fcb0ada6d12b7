"""GPU diagnostic: growth of the difference between the spectral carry-over and the reference's data flow."""
import sys

import torch

sys.path.insert(0, ".")
from tests.test_slab_gpu import _make, _step_all, _gather  # noqa: E402

for shape, P, nsub in [((9, 7, 5), 3, 1), ((9, 7, 5), 1, 1), ((8, 7, 6), 1, 1), ((8, 6, 5), 1, 1)]:
    torch.manual_seed(4)
    L = [3.0, 2.0, 2.5]
    c0 = torch.rand(shape, dtype=torch.float64) * 0.12 + 0.44
    plain = _make(3, list(shape), L, P, nsub=nsub)
    carry = _make(3, list(shape), L, P, nsub=nsub, carry=True)
    for s in plain + carry:
        yb, nyl = s.st.real_begin[1], s.st.real_shape[1]
        s.set_local(c0[:, yb:yb + nyl].contiguous().cuda())
    for k in range(6):
        _step_all(plain, 1e-3, 5)
        _step_all(carry, 1e-3, 5)
        d = _gather(carry) - _gather(plain)
        dN = max((a.cur - b.cur).abs().max().item() for a, b in zip(carry, plain))
        # carried spectrum vs the transform of the real field (rank 0 rows of the global rfftn)
        full = torch.fft.rfftn(_gather(carry))
        s0 = carry[0]
        nxl = s0.st.recip_shape[0]
        cb = torch.view_as_complex(s0.cbar.cpu().reshape(-1, 2)).reshape(s0.st.recip_shape)
        dcb = (cb - full[:nxl]).abs()
        print(shape, k, "dc", d.abs().max().item(), "mean", d.mean().item(), "dN", dN, "Nmax", max(b.cur.abs().max().item() for b in plain),
              "cbar-fft(c)", dcb.max().item(), "at", tuple(int(i) for i in (dcb == dcb.max()).nonzero()[0]))
