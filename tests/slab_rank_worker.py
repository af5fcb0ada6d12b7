"""One rank of a multi-process slab job on the GPU(s) of this box; started once per rank by tests/test_slab_native_gpu.py (and
tools/), never imported by the product.  All ranks may share ONE GPU: the library's transport (HIP IPC peer mappings + flags)
works between processes on the same device, which RCCL refuses.

usage: slab_rank_worker.py <job> <nranks> <rank> <case> [key=value ...]   -> one JSON line on stdout
cases
  ch       3-D / 2-D Cahn-Hilliard through mrl_ch_substeps on a slab context with the library-owned exchange, compared with
           the serial oracle on the global field (shape=.., steps=.., substeps=.., transport=1|2|3, nsub=.., carry=0|1)
  chgold   test/tests/cahnhilliard/cahnhilliard.i on nranks ranks (2-D 20^2, full spectrum) against gold c.1 .. c.10 of
           cahnhilliard.h5 (this rank's slab) -- with nranks = 2, rank 1 is the reference's cahnhilliard.rank0001.h5 run
  fft      mrl_fft_r2c / mrl_fft_c2r on a slab context against torch.fft on the global array
"""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse(argv):
    kv = {}
    for a in argv:
        k, v = a.split("=", 1)
        kv[k] = v
    return kv


def ring_arrays(ctx, n):
    shp = list(ctx.recip_shape)
    pitch = ctx.spec_pitch if ctx.dim == 3 else shp[-1]
    count = 1
    for s in shp[:-1]:
        count *= s
    return [torch.zeros(2 * count * pitch, dtype=torch.float64, device=ctx.device) for _ in range(n)]


def run_ch(job, P, r, kv, gold=False):
    from marlin_amd import api
    from oracle import marlin_oracle as mo

    pred = int(kv.get("pred", 2))
    if gold:
        # test/tests/cahnhilliard/tests:58-70: 2-rank FFT_SLAB run of cahnhilliard.i; both ranks draw the same seed-0 block, so the
        # global field is cat([blk, blk]) and the gold file holds the slab global[:, 10:20] of rank 1
        from tests.conftest import load_golden
        g = load_golden("cahnhilliard_rank0001_gold.npz")
        shape, L, steps, substeps = [20, 20], [3.0, 3.0], 10, 10
        torch.manual_seed(0)
        blk = torch.rand(20, 10, dtype=torch.float64) * (0.56 - 0.44) + 0.44
        c0 = torch.cat([blk, blk], dim=1)
    else:
        shape = [int(x) for x in kv.get("shape", "64,64,64").split(",")]
        L = [3.0 + 0.25 * i for i in range(len(shape))]
        steps, substeps = int(kv.get("steps", 2)), int(kv.get("substeps", 3))
        torch.manual_seed(5)
        c0 = 0.44 + 0.12 * torch.rand(shape, dtype=torch.float64)
    dt = 1e-3
    dim = len(shape)
    spectrum = api.SPECTRUM_HALF if dim == 3 else api.SPECTRUM_FULL
    comm = api.Comm(job, P, r, device=0, transport=int(kv.get("transport", 0)), timeout=40.0)
    ctx = api.Context(dim, shape, L, nranks=P, rank=r, spectrum=spectrum, slab=True, device=0)
    ctx.attach_comm(comm)
    ctx.set_option(api.OPT_SLAB_NSUB, int(kv.get("nsub", 1)))
    ctx.set_option(api.OPT_SLAB_CARRY, int(kv.get("carry", 0)))
    p = api.ch_params()
    yb, nyl = ctx.real_begin[1], ctx.real_shape[1]
    c = c0[:, yb:yb + nyl].contiguous().cuda()
    ring = ring_arrays(ctx, pred)
    head, n_old = 0, 0
    dom = mo.Domain(dim, shape, L, slab_c2c=(dim == 2))
    ref = mo.CahnHilliardABM(dom, c0, M=0.2, kappa_factor=-0.001, mu_fn=mo.mu_double_well, substeps=substeps, predictor_order=pred)
    errs, gold_errs = [], []
    for step in range(1, steps + 1):
        ref.step(dt)
        # TensorProblem: advanceState (a no-op while timeStep() <= 1), then the solver's substep loop = ONE library call
        if step > 1:
            head = (head + 1) % len(ring)
            n_old = min(n_old + 1, pred - 1)
        out = torch.empty_like(c)
        head, n_old = ctx.ch_substeps(p, c, out, ring, head, n_old, pred, substeps, step > 1, dt / substeps)
        ctx.sync()
        c = out
        errs.append((c.cpu() - ref.c[:, yb:yb + nyl]).abs().max().item())
        if gold and yb >= 10:
            gk = torch.from_numpy(g[f"c.{step}"].copy())
            gold_errs.append((c.cpu() - gk[:, yb - 10:yb - 10 + nyl]).abs().max().item())
    st, tr = comm.stats(), comm.transport
    ctx.close()
    comm.close()
    out = {"max_err": max(errs), "transport": tr, "stats": st}
    if gold_errs:
        out["max_gold_err"] = max(gold_errs)
    return out


def run_fft(job, P, r, kv):
    from marlin_amd import api
    shape = [int(x) for x in kv.get("shape", "16,12,10").split(",")]
    dim = len(shape)
    spectrum = int(kv.get("spectrum", 0 if dim == 3 else 1))
    comm = api.Comm(job, P, r, device=0, transport=int(kv.get("transport", 0)), timeout=40.0)
    ctx = api.Context(dim, shape, [1.0 + 0.5 * d for d in range(dim)], nranks=P, rank=r, spectrum=spectrum, slab=True, device=0)
    ctx.attach_comm(comm)
    torch.manual_seed(11)
    a = torch.rand(shape, dtype=torch.float64)
    full = torch.fft.fftn(a) if spectrum == 1 else torch.fft.rfftn(a)
    yb, nyl = ctx.real_begin[1], ctx.real_shape[1]
    xb, nxl = ctx.recip_begin[0], ctx.recip_shape[0]
    loc = a[:, yb:yb + nyl].contiguous().cuda()
    errs = []
    for rep in range(3):   # repeated forward transforms reuse the receive buffer: the acknowledgement flags are exercised
        spec = ctx.fft(loc)
        ctx.sync()
        errs.append((spec.cpu() - full[xb:xb + nxl]).abs().max().item() / full.abs().max().item())
    back = ctx.ifft(spec)
    ctx.sync()
    errs.append((back.cpu() - a[:, yb:yb + nyl]).abs().max().item())
    tot = ctx.sum(loc)
    errs.append(abs(tot - a.sum().item()) / a.sum().item())
    ctx.close()
    comm.close()
    return {"max_err": max(errs), "errs": errs}


def main():
    job, P, r, case = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    kv = parse(sys.argv[5:])
    torch.cuda.set_device(0)
    if case == "ch":
        out = run_ch(job, P, r, kv)
    elif case == "chgold":
        out = run_ch(job, P, r, kv, gold=True)
    elif case == "fft":
        out = run_fft(job, P, r, kv)
    else:
        raise SystemExit(f"unknown case {case}")
    out.update({"rank": r, "case": case})
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
