#!/usr/bin/env python3
"""Time the LOCAL work of one rank of the slab-decomposed de Geus mechanics solve (BASELINE configs[4]: 256^3 over 8 GPUs)
on a single GPU: the exchanges are device copies of the rank's own send buffers (same byte counts), the all-reduced CG
scalars are the local ones times P.  Fixed number of CG iterations (l_tol = 0).
usage: slab_mech_local_bench.py [P] [n] [its] [fast 0|1]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marlin_amd.slab import HipSlabStages, SlabMechanics  # noqa: E402


class _Copy:
    def __init__(self, sc, rc):
        assert sum(sc) == sum(rc)

    def run(self, send, recv, async_op=False):
        recv.copy_(send)
        return None


class _Comm:
    def __init__(self, P):
        self.P = P

    def exchange(self, sc, rc):
        return _Copy(sc, rc)

    def allreduce(self, values):
        return [v * self.P for v in values]


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    its = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    fast = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
    fusion = bool(int(sys.argv[5])) if len(sys.argv) > 5 else True
    shape, L = [n] * 3, [6.283185307179586] * 3
    st = HipSlabStages(3, shape, L, P, 0)
    K = torch.full(st.real_shape, 0.833, dtype=torch.float64, device="cuda")
    mu = torch.full(st.real_shape, 0.386, dtype=torch.float64, device="cuda")
    m = SlabMechanics(3, shape, L, P, 0, K, mu, comm=_Comm(P), l_tol=0.0, l_max_its=its, stages=st, fast=fast, tangent_fusion=fusion)
    nvec = m.npts * 9
    g = torch.Generator(device="cuda").manual_seed(1)
    F = torch.eye(3, dtype=torch.float64, device="cuda").reshape(9, 1).expand(9, m.npts).contiguous().reshape(-1) if fast else \
        torch.eye(3, dtype=torch.float64, device="cuda").reshape(1, 9).expand(m.npts, 9).contiguous().reshape(-1)
    b = torch.rand(nvec, dtype=torch.float64, device="cuda", generator=g) - 0.5
    x = torch.zeros(nvec, dtype=torch.float64, device="cuda")
    m._cg(F, b, x)                       # warm-up
    torch.cuda.synchronize()
    x.zero_()
    t0 = time.perf_counter()
    done = m._cg(F, b, x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / done
    st.ctx.set_profiling(True)
    x.zero_()
    m._cg(F, b, x)
    torch.cuda.synchronize()
    prof = [k for k in st.ctx.get_profile() if k["launches"]]
    for k in prof:
        k["avg_ms"] = k["ms"] / k["launches"]
        k["per_iteration_ms"] = k["ms"] / (its + 1)
    print(json.dumps({"P": P, "grid": shape, "local_real": st.real_shape, "fast": fast, "tangent_fusion": fusion, "cg_iterations": done,
                      "ms_per_cg_iteration_local_incl_copies_and_syncs": round(ms, 4),
                      "kernels": [{"kernel": k["kernel"], "launches": k["launches"], "avg_ms": round(k["avg_ms"], 4),
                                   "per_iteration_ms": round(k["per_iteration_ms"], 4)} for k in prof]}))


if __name__ == "__main__":
    main()
