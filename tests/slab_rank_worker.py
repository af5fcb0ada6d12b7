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
    if "exp" in kv:
        ctx.set_option(api.OPT_EXPERIMENT, int(kv["exp"]))
    ctx.set_profiling(True)
    if kv.get("verify", "0") == "1":   # consumers re-read their receive buffers with system-scope loads (MRL_OPT_VERIFY_EXCHANGE)
        ctx.set_option(api.OPT_VERIFY_EXCHANGE, 1)
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
    kernels = sorted(k["kernel"] for k in ctx.get_profile() if k["launches"])
    mismatches = ctx.get_option(api.OPT_VERIFY_MISMATCHES)
    ctx.close()
    comm.close()
    out = {"max_err": max(errs), "transport": tr, "stats": st, "kernels": kernels, "verify_mismatches": mismatches}
    if gold_errs:
        out["max_gold_err"] = max(gold_errs)
    return out


def run_chbench(job, P, r, kv):
    """exactly what `bench.py --gpus P` runs per rank with the native driver -- grid_for(P, n) with the bench's initial condition,
    mrl_ch_substeps on a slab context -- against the serial fused path (mrl_ch_substeps on a one-GPU context of the same GLOBAL grid)
    in this process: `steps` substeps in one call, this rank's slab to 1e-13"""
    from bench import grid_for, splitmix64_uniform
    from marlin_amd import api
    import time
    t_start = time.time()

    def mark(what):
        print(f"[rank {r} +{time.time() - t_start:6.1f}s] {what}", file=sys.stderr, flush=True)

    n = int(kv.get("n", 256))
    steps = int(kv.get("steps", 4))
    shape = [int(x) for x in kv["shape"].split(",")] if "shape" in kv else grid_for(P, n)
    dx = 8.0 * math.pi / 200.0
    L = [s * dx for s in shape]
    npts = shape[0] * shape[1] * shape[2]
    if kv.get("ic", "splitmix") == "rand":   # big grids: the same seeded device generator in every rank process (same device, same stream of numbers)
        torch.manual_seed(1234)
        c0 = torch.rand(shape, dtype=torch.float64, device="cuda") * 0.12 + 0.44
    else:
        c0 = torch.from_numpy(splitmix64_uniform(npts).reshape(shape))
    mass0 = float(c0.sum(dtype=torch.float64).item())
    p = api.ch_params()
    noref = kv.get("noref", "0") == "1"   # debugging aid: no serial reference (nothing but the slab job touches the device)
    want = torch.empty(shape, dtype=torch.float64, device="cuda")
    if not noref:
        serial = api.Context(3, shape, L, device=0)
        ring = [serial.empty_hist(), serial.empty_hist()]
        serial.ch_substeps(p, c0.cuda(), want, ring, 1, 0, 2, steps, True, 1e-3)
        serial.sync()
        mark("serial reference done")
        del ring
        serial.close()
    comm = api.Comm(job, P, r, device=0, transport=int(kv.get("transport", 0)), timeout=60.0)
    mark("communicator up")
    ctx = api.Context(3, shape, L, nranks=P, rank=r, slab=True, device=0)
    ctx.attach_comm(comm)
    ctx.set_option(api.OPT_SLAB_NSUB, int(kv.get("nsub", 1)))
    ctx.set_option(api.OPT_SLAB_CARRY, int(kv.get("carry", 0)))
    if "exp" in kv:
        ctx.set_option(api.OPT_EXPERIMENT, int(kv["exp"]))
    yb, nyl = ctx.real_begin[1], ctx.real_shape[1]
    want = want[:, yb:yb + nyl].contiguous()
    c = c0[:, yb:yb + nyl].contiguous().cuda()
    del c0
    torch.cuda.empty_cache()
    out = torch.empty_like(c)
    mark("slab solve enqueue")
    ctx.ch_substeps(p, c, out, ring_arrays(ctx, 2), 1, 0, 2, steps, True, 1e-3)
    mark("slab solve enqueued")
    ctx.sync()
    mark("slab solve done")
    err = 0.0 if noref else (out - want).abs().max().item()
    mass = comm.allreduce([float(out.sum(dtype=torch.float64).item())])[0]
    ctx.close()
    comm.close()
    return {"max_err": err, "mass_err": abs(mass - mass0) / npts, "grid": shape}


def run_lost_peer(job, P, r, kv):
    """a peer that stops taking part must end in MRL_ERR_COMM on the others within the time-out, not in a hung GPU: every rank runs one
    solver call (the exchange pipeline is built collectively), then the last rank leaves while the others start a second call"""
    from marlin_amd import api
    shape = [64, 64, 64]
    comm = api.Comm(job, P, r, device=0, timeout=float(kv.get("timeout", 4.0)))
    ctx = api.Context(3, shape, [3.0, 3.0, 3.0], nranks=P, rank=r, slab=True, device=0)
    ctx.attach_comm(comm)
    yb, nyl = ctx.real_begin[1], ctx.real_shape[1]
    torch.manual_seed(1)
    c = (0.44 + 0.12 * torch.rand(shape, dtype=torch.float64))[:, yb:yb + nyl].contiguous().cuda()
    out = torch.empty_like(c)
    ring = ring_arrays(ctx, 2)
    ctx.ch_substeps(api.ch_params(), c, out, ring, 1, 0, 2, 2, True, 1e-3)
    ctx.sync()
    if r == P - 1:
        return {"left": True}       # (its buffers stay mapped in the peers; the process simply ends)
    import time
    t0 = time.perf_counter()
    code, msg = 0, ""
    try:
        ctx.ch_substeps(api.ch_params(), out, c, ring, 0, 1, 2, 2, True, 1e-3)
        ctx.sync()
    except api.MarlinHipError as e:
        code, msg = e.code, e.message
    return {"left": False, "code": code, "message": msg, "seconds": time.perf_counter() - t0}


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


def run_rccl_preflight(job, P, r, kv):
    """everything of the RCCL bring-up that P rank processes on ONE GPU can exercise (mrl_comm_rccl_preflight): the library loads, rank
    0's unique id reaches every rank through the bootstrap segment, the placement check sees the shared device and reports RCCL as
    unavailable (MRL_ERR_UNSUPPORTED) instead of calling ncclCommInitRank into its 'invalid usage' failure; the communicator stays usable"""
    from marlin_amd import api
    comm = api.Comm(job, P, r, device=0, transport=0, timeout=40.0)
    rc = comm.rccl_preflight()
    d = comm.describe()
    # the transport switch reports the same verdict, and the communicator still works afterwards
    switched = True
    try:
        comm.set_transport(api.TRANSPORT_RCCL)
    except api.MarlinHipError as e:
        switched = False
        switch_code = e.code
    v = [float(r + 1)]
    tot = comm.allreduce(v, 0)
    comm.close()
    return {"rc": rc, "describe": d, "switched": switched, "switch_code": None if switched else switch_code, "allreduce": tot}


def run_pencil(job, P, r, kv):
    """parallel_mode = FFT_PENCIL (DomainAction.C:568-742, 1021-1047, 1105-1404): mrl_fft_r2c / mrl_fft_c2r on a pencil context with the
    library-owned staged exchanges, against the oracle's restatement of the reference's stages (oracle.PencilDomain) and the serial
    torch transform of the global array; local shapes, begins and reciprocal axes against partitionPencils"""
    import math
    from marlin_amd import api
    from oracle import marlin_oracle as mo
    shape = [int(x) for x in kv.get("shape", "16,12,10").split(",")]
    L = [2.0 * math.pi, 4.0 * math.pi, 6.0 * math.pi]
    comm = api.Comm(job, P, r, device=0, transport=int(kv.get("transport", 0)), timeout=40.0)
    ctx = api.Context(3, shape, L, nranks=P, rank=r, pencil=True, device=0)
    ctx.attach_comm(comm)
    dom = mo.PencilDomain(shape, L, P)
    rs, ks = dom.real_slices(r), dom.recip_slices(r)
    layout_ok = (ctx.pencil_grid == (dom.Py, dom.Pz)
                 and ctx.real_shape == [s.stop - s.start for s in rs] and ctx.real_begin == [s.start for s in rs]
                 and ctx.recip_shape == [s.stop - s.start for s in ks] and ctx.recip_begin == [s.start for s in ks])
    axes_ok = all(bool((ctx.reciprocal_axis(d).cpu() == dom.kaxis[d][ks[d]]).all()) for d in range(3))
    torch.manual_seed(11)
    g = torch.rand(shape, dtype=torch.float64)
    blocks = dom.split(g)
    ref = dom.fft(blocks)[r]
    serial = torch.fft.fftn(torch.fft.rfft(g, dim=0), dim=(1, 2))[ks]
    loc = blocks[r].cuda()
    errs = []
    scale = serial.abs().max().item()
    for rep in range(3):   # repeated transforms reuse the receive buffers: the acknowledgement flags are exercised
        spec = ctx.fft(loc)
        ctx.sync()
        errs.append((spec.cpu() - ref).abs().max().item() / scale)
        errs.append((spec.cpu() - serial).abs().max().item() / scale)
    back = ctx.ifft(spec)
    ctx.sync()
    errs.append((back.cpu() - blocks[r]).abs().max().item())
    # an inverse transform of a spectrum that is NOT the image of a real field in its self-conjugate x bins (irfft ignores their
    # imaginary parts): against the oracle's irfft-based stages
    torch.manual_seed(12 + r)
    junk = torch.randn(ctx.recip_shape, dtype=torch.complex128)
    junk_all = []
    for q in range(P):
        torch.manual_seed(12 + q)
        junk_all.append(torch.randn([s.stop - s.start for s in dom.recip_slices(q)], dtype=torch.complex128))
    back2 = ctx.ifft(junk.cuda())
    ctx.sync()
    ref2 = dom.ifft(junk_all)[r]
    errs.append((back2.cpu() - ref2).abs().max().item() / max(ref2.abs().max().item(), 1e-300))
    tot = ctx.sum(loc)
    errs.append(abs(tot - g.sum().item()) / g.sum().item())
    # the mechanics entry points refuse pencil contexts (the reference's mechanics norms are serial-only, DomainAction.C:1564-1567)
    refused = False
    try:
        ctx.gamma_apply(torch.zeros(ctx.real_shape + [3, 3], dtype=torch.float64, device="cuda"))
    except api.MarlinHipError as e:
        refused = e.code == -2
    # Cahn-Hilliard substeps on the pencil context (the operator sequence of AdamsBashforthMoulton.C:88-101 over the staged transforms)
    # against the oracle's serial solution of the global field: AB1 + AB2 + AB2 through mrl_ch_substeps, then one mrl_ch_substep
    dser = mo.Domain(3, shape, L)
    c0 = 0.44 + 0.12 * g
    Mbar = mo.reciprocal_laplacian_factor(dser, 0.2)
    Lbar = mo.reciprocal_laplacian_square_factor(dser, -0.001)
    want, hist = c0, []
    for k in range(4):
        want, Nn, _, _ = mo.ch_substep_ops(want, Mbar, Lbar, hist[:1], 1e-3, min(k, 1), mo.mu_double_well, dser)
        hist = [Nn]
        if k == 2:
            want3 = want
    p = api.ch_params()
    ring = [torch.zeros(2 * spec.numel(), dtype=torch.float64, device="cuda") for _ in range(2)]
    cl = dom.split(c0)[r].cuda()
    out3 = torch.empty_like(cl)
    head, n_old = ctx.ch_substeps(p, cl, out3, ring, 1, 0, 2, 3, True, 1e-3)
    ctx.sync()
    ch_err = (out3.cpu() - dom.split(want3)[r]).abs().max().item()
    head = (head + 1) % 2                       # advance_state: the newest N-hat becomes old[0]
    out4 = torch.empty_like(cl)
    ctx.ch_substep(p, out3, out4, ring[(head + 1) % 2], [ring[head]], 1, 1e-3)
    ctx.sync()
    ch_err = max(ch_err, (out4.cpu() - dom.split(want)[r]).abs().max().item())
    st = comm.stats()
    ctx.close()
    comm.close()
    return {"max_err": max(errs), "errs": errs, "layout_ok": layout_ok, "axes_ok": axes_ok, "refused": refused, "stats": st,
            "grid": [dom.Py, dom.Pz], "ch_err": ch_err}


def run_mech(job, P, r, kv):
    """FFTMechanics::computeBuffer through mrl_mech_newton_cg on a slab context (library-owned exchanges, device-side all-reduce of the
    CG scalars).  case=gold: test/tests/mechanics/mech3d.i against mech3d.h5 (abs 1e-10); otherwise an n^3 two-phase RVE (planned
    shapes: the fused field-major row pipeline) against the oracle's serial solve with the same Newton / CG iteration counts."""
    import numpy as np
    from marlin_amd import api
    from oracle import marlin_oracle as mo
    from tests.conftest import load_golden
    from tests.test_oracle_golden import MECH_CASES, _mech_setup
    gold = kv.get("gold", "0") == "1"
    if gold:
        pz = MECH_CASES["mech3d"]
        dim, n = pz["dim"], pz["n"]
        shape = [n] * dim
        dom, phase, K, mu = _mech_setup(dim, n)
        g = load_golden(pz["gold"])
        l_tol, nl_rel, nl_abs, dt, substeps, nsteps = pz["l_tol"], pz["nl_rel"], pz["nl_abs"], pz["dt"], pz["substeps"], 3
    else:
        shape = [int(x) for x in kv.get("shape", "32,32,32").split(",")]
        dim = 3
        dom = mo.Domain(3, shape, [2 * math.pi] * 3)
        nx, ny, nz = shape
        phase = torch.zeros(shape, dtype=torch.float64)
        phase[-(9 * nx // 32):, :9 * ny // 32, -(9 * nz // 32):] = 1.0       # test/src/tensor_computes/PhaseMechanicsTest.C:36-45
        K = (1.0 - phase) * 0.833 + phase * 8.33
        mu = (1.0 - phase) * 0.386 + phase * 3.86
        l_tol, nl_rel, nl_abs, dt, substeps, nsteps = 1e-2, 2e-2, 2e-2, 0.01, int(kv.get("substeps", 2)), 1
    comm = api.Comm(job, P, r, device=0, transport=int(kv.get("transport", 0)), timeout=60.0)
    ctx = api.Context(dim, shape, [2 * math.pi] * dim, nranks=P, rank=r, slab=True, device=0)
    ctx.attach_comm(comm)
    ctx.set_profiling(True)
    if kv.get("exp"):
        ctx.set_option(api.OPT_EXPERIMENT, int(kv["exp"]))
    yb, nyl = ctx.real_begin[1], ctx.real_shape[1]
    Kl, mul = K[:, yb:yb + nyl].contiguous().cuda(), mu[:, yb:yb + nyl].contiguous().cuda()
    F = torch.eye(dim, dtype=torch.float64).expand(list(ctx.real_shape) + [dim, dim]).contiguous().cuda()
    # ref=hip: the SERIAL HIP solver on the whole grid as the reference (sizes the oracle cannot reach in test time; the serial solver
    # itself is oracle-checked up to 128^3, tests/test_fullsize_gpu.py)
    hip_ref = kv.get("ref", "oracle") == "hip"
    if hip_ref:
        sctx = api.Context(dim, shape, [2 * math.pi] * dim)
        Ks, mus = K.cuda(), mu.cuda()
        Fref = torch.eye(dim, dtype=torch.float64).expand(list(shape) + [dim, dim]).contiguous().cuda()
    else:
        ref = mo.FFTMechanicsOracle(dom, K, mu, l_tol=l_tol, nl_rel_tol=nl_rel, nl_abs_tol=nl_abs)
        Fref = torch.eye(dim, dtype=torch.float64).expand(dom.value_shape([dim, dim])).contiguous()
    perm = (2, 1, 0)
    errs, gold_errs, traces_ok = [], [], True
    t_old = 0.0
    import time
    t_lib = t_ref = 0.0
    nsteps = int(kv.get("steps", nsteps))
    for step in range(nsteps):
        sub_dt = dt / substeps
        for sidx in range(substeps):
            t = t_old + sidx * sub_dt
            applied = torch.eye(dim, dtype=torch.float64)
            applied[0, 1] = applied[0, 1] + t
            t0 = time.perf_counter()
            applied = (applied - ctx.average(F)).cuda()                    # MacroscopicShearTensor.C:31-41 (global average)
            F, Pk, st = ctx.mech_newton_cg(F, Kl, mul, applied, l_tol=l_tol, nl_rel_tol=nl_rel, nl_abs_tol=nl_abs)
            t1 = time.perf_counter()
            if hip_ref:
                app_s = torch.eye(dim, dtype=torch.float64)
                app_s[0, 1] = app_s[0, 1] + t
                app_s = (app_s - sctx.average(Fref)).cuda()
                Fref, _, sst = sctx.mech_newton_cg(Fref, Ks, mus, app_s, l_tol=l_tol, nl_rel_tol=nl_rel, nl_abs_tol=nl_abs)
                ok = st["newton_its"] == sst["newton_its"] and list(st["cg_its"]) == list(sst["cg_its"])
            else:
                Fref, rst = ref.compute(Fref, mo.macroscopic_shear(dom, Fref, t))
                ok = st["newton_its"] == rst.newton_its and list(st["cg_its"]) == list(rst.cg_its)
            t_lib += t1 - t0
            t_ref += time.perf_counter() - t1
            traces_ok = traces_ok and ok
        t_old += dt
        errs.append((F - Fref[:, yb:yb + nyl].to(F.device)).abs().max().item())
        if gold:
            Fl = F.cpu().reshape(list(ctx.real_shape) + [dim * dim])
            for k in range(dim * dim):
                gk = torch.from_numpy(np.ascontiguousarray(g[f"F_{k}.{step}"])).permute(*perm)   # XDMF default transpose
                gold_errs.append((Fl[..., k] - gk[:, yb:yb + nyl]).abs().max().item())
    tr = comm.transport
    kernels = sorted(k["kernel"] for k in ctx.get_profile() if k["launches"])
    ctx.close()
    comm.close()
    if hip_ref:
        sctx.close()
    out = {"max_err": max(errs), "traces_ok": bool(traces_ok), "transport": tr, "seconds_library": round(t_lib, 2), "kernels": kernels,
           "seconds_oracle": round(t_ref, 2), "cg_its_last": list(st["cg_its"])}
    if gold_errs:
        out["max_gold_err"] = max(gold_errs)
    return out


def main():
    job, P, r, case = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    kv = parse(sys.argv[5:])
    torch.set_num_threads(max(1, 12 // P))   # the oracle side: the ranks share the box's 16-core CPU allotment
    torch.cuda.set_device(0)
    if case == "ch":
        out = run_ch(job, P, r, kv)
    elif case == "chgold":
        out = run_ch(job, P, r, kv, gold=True)
    elif case == "rccl_preflight":
        out = run_rccl_preflight(job, P, r, kv)
    elif case == "pencil":
        out = run_pencil(job, P, r, kv)
    elif case == "fft":
        out = run_fft(job, P, r, kv)
    elif case == "chbench":
        out = run_chbench(job, P, r, kv)
    elif case == "lost_peer":
        out = run_lost_peer(job, P, r, kv)
    elif case == "mech":
        out = run_mech(job, P, r, kv)
    else:
        raise SystemExit(f"unknown case {case}")
    out.update({"rank": r, "case": case})
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
