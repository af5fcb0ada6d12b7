#!/usr/bin/env python3
"""de Geus RVE benchmark (BASELINE.json configs[2]): 3-D n^3, cubic inclusion
phase[-s:, :s, -s:] = 1 with s = 9n/32 (test/src/tensor_computes/PhaseMechanicsTest.C:36-45), K = 0.833/8.33,
mu = 0.386/3.86 (examples/degeus_mechanics/mech.i:23-38), shear ramp, l_tol = 1e-2, nl tolerances 2e-2.
Reports time per CG iteration and the per-kernel device times.   usage: mech_bench.py [n] [substeps] [slab 0|1] [experiment mask] [profile 0|1]
slab = 1: the same solve as a ONE-rank slab job through the library's row pipeline (communicator, flags, exchange tables)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marlin_amd.api import Comm, Context  # noqa: E402


def run(n=128, substeps=2, profile=True, slab=False, exp=0):
    L = 2.0 * torch.pi
    if slab:
        comm = Comm(f"mrl_mechbench_{os.getpid()}", 1, 0, device=0)
        ctx = Context(3, [n, n, n], [L, L, L], nranks=1, rank=0, slab=True)
        ctx.attach_comm(comm)
    else:
        ctx = Context(3, [n, n, n], [L, L, L])
    if exp:
        ctx.set_option(0, exp)
    s = 9 * n // 32
    phase = torch.zeros(n, n, n, dtype=torch.float64)
    phase[-s:, :s, -s:] = 1.0
    K = ((1.0 - phase) * 0.833 + phase * 8.33).cuda()
    mu = ((1.0 - phase) * 0.386 + phase * 3.86).cuda()
    F = torch.eye(3, dtype=torch.float64).expand(n, n, n, 3, 3).contiguous().cuda()
    sub_dt = 0.01 / 10
    res = []
    for it in range((2 if profile else 1) * substeps + 1):
        if it == substeps + 1:      # second half: per-kernel event timing (adds launch overhead; not part of the headline time)
            ctx.set_profiling(True)
        t = it * sub_dt
        applied = torch.eye(3, dtype=torch.float64)
        applied[0, 1] += t
        applied = (applied - ctx.average(F)).cuda()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        Fnew, P, st = ctx.mech_newton_cg(F, K, mu, applied, l_tol=1e-2, nl_rel_tol=2e-2, nl_abs_tol=2e-2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        F = Fnew
        if 1 <= it <= substeps:
            res.append((dt, st["cg_its_total"], st["newton_its"]))
    prof = [k for k in ctx.get_profile() if k["launches"]]
    ctx.set_profiling(False)
    # the small-strain linear-elastic form of the same RVE (mrl_mech_small_strain: constant tangent, one CG solve), serial contexts
    small = None
    if not slab:
        E = torch.zeros(3, 3, dtype=torch.float64)
        E[0, 1] = E[1, 0] = 0.005
        E = E.cuda()
        ctx.mech_small_strain(K, mu, E, l_tol=1e-6)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, _, sst = ctx.mech_small_strain(K, mu, E, l_tol=1e-6)
        torch.cuda.synchronize()
        small = {"l_tol": 1e-6, "cg_iterations": sst["cg_its"], "ms_per_cg_iteration": (time.perf_counter() - t0) / max(sst["cg_its"], 1) * 1e3}
    tot_its = sum(r[1] for r in res)
    tot_t = sum(r[0] for r in res)
    npts = n ** 3
    # 2*9*B_fft(n) + 232 + 504 bytes per point per CG iteration
    h = 8.0 * (1.0 + 2.0 / n)
    bpi = 2 * 9 * (8.0 + 5.0 * h) + 232 + 504
    out = {"n": n, "substeps": substeps, "slab_one_rank": slab, "exp": exp, "newton_its": [r[2] for r in res], "cg_its": [r[1] for r in res],
           "ms_per_cg_iteration": tot_t / max(tot_its, 1) * 1e3,
           "algorithmic_bytes_per_point_per_cg_iteration": bpi,
           "achieved_GBps": bpi * npts * tot_its / tot_t / 1e9, "small_strain_linear_elastic": small,
           "kernels": sorted([{"kernel": k["kernel"], "total_ms": round(k["ms"], 3), "launches": k["launches"],
                               "avg_ms": round(k["ms"] / k["launches"], 4),
                               "GBps": round(k["bytes_per_launch"] / (k["ms"] / k["launches"]) / 1e6, 1)} for k in prof],
                             key=lambda k: -k["total_ms"])}
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    substeps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    slab = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
    exp = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    profile = bool(int(sys.argv[5])) if len(sys.argv) > 5 else True   # 0: no event-timed second half (for timeline traces)
    print(json.dumps(run(n, substeps, profile=profile, slab=slab, exp=exp)))


if __name__ == "__main__":
    main()
