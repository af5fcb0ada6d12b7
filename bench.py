#!/usr/bin/env python3
"""Headline benchmark: grid-point-updates/s of the Cahn-Hilliard semi-implicit spectral substep
(BASELINE.json: 3-D 256^3 fp64, AB2) on N MI355X GPUs + fraction of the HBM roofline.

A "step" is one solver substep (AdamsBashforthMoulton::substep + its compute group) of the whole
grid.  N=1: 256^3 on one GPU.  N>1: slab decomposition with an RCCL all-to-all per transform;
per-GPU work is held at 256^3 points (weak scaling; N=8 is the 512^3 configuration of north_star).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

# the host driver of this pool only supports dmabuf IPC (RCCL / cross-process device memory); harmless elsewhere
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)


def splitmix64_uniform(count, seed=0, lo=0.44, hi=0.56, offset=0):
    """Counter-based IC: element i = lo + (hi-lo) * (splitmix64(seed + offset + i) >> 11) * 2^-53."""
    idx = np.arange(offset, offset + count, dtype=np.uint64) + np.uint64(seed)
    with np.errstate(over="ignore"):
        z = idx * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return lo + (hi - lo) * u


def algorithmic_bytes_per_update(n_last, n_old):
    """3*B_fft(n) + n_old*8*(1+2/n), B_fft = 8 + 5*8*(1+2/n)."""
    h = 8.0 * (1.0 + 2.0 / n_last)
    return 3.0 * (8.0 + 5.0 * h) + n_old * h


def grid_for(ngpus, base):
    """weak scaling: per-GPU work fixed at base^3 points; axes doubled in the order y, x, z."""
    g = [base, base, base]
    order = [1, 0, 2]
    k = 0
    m = ngpus
    while m > 1:
        g[order[k % 3]] *= 2
        m //= 2
        k += 1
    return g


# bench profile slot -> kernel name prefix in the rocprofv3 traces
KERNEL_OF_SLOT = {"ch_A_z_fwd": "k_z_fwd<", "ch_B_y_fwd": "k_pass<", "ch_C_x_fused": "k_ch_xfused<", "ch_D_y_inv": "k_pass<",
                  "ch_E_z_inv": "k_z_inv<"}


def measured_traffic(slot, n, order_tag):
    """HBM bytes per launch of the kernel behind `slot`, from the committed rocprofv3 PMC passes of this same
    command (profiles/traffic_ch<n>.json, written by tools/profile_gpu.sh: FETCH_SIZE x2 (gfx950) + WRITE_SIZE)."""
    path = os.path.join(ROOT, "profiles", f"traffic_ch{n}.json")
    if not os.path.exists(path) or slot not in KERNEL_OF_SLOT:
        return None, None
    with open(path) as f:
        t = json.load(f)
    pref = KERNEL_OF_SLOT[slot]
    cands = [k for k in t if k.startswith(pref) and (order_tag is None or order_tag in k)]
    if slot == "ch_B_y_fwd":
        cands = [k for k in cands if "false, 2" in k]
    if slot == "ch_D_y_inv":
        cands = [k for k in cands if "true, 1" in k]
    if len(cands) != 1:
        return None, None
    v = t[cands[0]]
    if v.get("fetch_bytes") is None or v.get("write_bytes") is None:
        return None, None
    return v["fetch_bytes"] + v["write_bytes"], os.path.relpath(path, ROOT)


def cpu_baseline(shape, dx, sample_steps):
    """The oracle (libTorch CPU ops in the reference's order) timed on this box's host cores."""
    from oracle import marlin_oracle as mo

    L = [s * dx for s in shape]
    dom = mo.Domain(3, list(shape), L)
    Mbar = mo.reciprocal_laplacian_factor(dom, 0.2)
    Lbar = mo.reciprocal_laplacian_square_factor(dom, -0.001)
    n = int(np.prod(shape))
    c = torch.from_numpy(splitmix64_uniform(n).reshape(shape))
    # the host cores this process may actually use (the GPU box hands out a CPU share, not the whole host)
    threads = max(1, min(len(os.sched_getaffinity(0)), 16))

    def sample(nthreads, budget_s, max_steps):
        torch.set_num_threads(nthreads)
        cc, N0, _, _ = mo.ch_substep_ops(c, Mbar, Lbar, [], 1e-3, 0, mo.mu_double_well, dom)  # warm-up + history
        t0 = time.perf_counter()
        done = 0
        for _ in range(max_steps):
            cc, N1, _, _ = mo.ch_substep_ops(cc, Mbar, Lbar, [N0], 1e-3, 1, mo.mu_double_well, dom)
            N0 = N1
            done += 1
            if time.perf_counter() - t0 > budget_s:   # bounded sample
                break
        return done, time.perf_counter() - t0

    steps_mt, dt_mt = sample(threads, 16.0, sample_steps)
    # the reference's default is ONE libTorch thread (it only raises the count for --n-threads, TensorProblem.C:77-82)
    steps_1t, dt_1t = sample(1, 6.0, max(1, sample_steps // 8))
    return {
        "value": n * steps_mt / dt_mt,
        "unit": "grid-point-updates/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{steps_mt} AB2 substeps of the same {shape[0]}x{shape[1]}x{shape[2]} fp64 grid "
                  f"(libTorch CPU ops in the reference's order, {threads} threads, {dt_mt:.1f} s)",
        "single_thread": {"value": n * steps_1t / dt_1t, "cores": 1,
                          "sample": f"{steps_1t} substeps, 1 thread (the reference's default), {dt_1t:.1f} s"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--grid", dest="n", type=int, default=256, help="base grid edge (per-GPU work = n^3 points)")
    ap.add_argument("--cpu-steps", type=int, default=24, help="substeps of the CPU baseline sample (0 = skip)")
    ap.add_argument("--profile-steps", type=int, default=10)
    ap.add_argument("--mech-grid", type=int, default=128, help="edge of the de Geus RVE side benchmark (config C); 0 = skip")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for smoke runs)")
    ap.add_argument("--nsub", type=int, default=2,
                    help="kz sub-blocks the slab substep is pipelined over (N > 1).  2: the rank-local kernels of 512^3 / 8 take 0.56 ms "
                         "per substep against 0.60 ms for 4 (smaller launches fill the chip worse), and every extra collective costs "
                         "~22 us of launch gap on the exchange stream, which is the critical path (DESIGN.md section 4)")
    ap.add_argument("--compute-stream", default="high", choices=["high", "default"],
                    help="N > 1: run the local passes on a high-priority stream so that they do not share a hardware queue "
                         "with RCCL's stream (streams of equal priority are multiplexed onto a few queues and then serialise)")
    ap.add_argument("--substeps-per-call", type=int, default=0,
                    help="substeps per solver call (N = 1: one library call, (mrl_ch_substeps = the substep loop of TensorSolver::computeBuffer; the "
                         "reference's cahnhilliard2.i runs 1000 substeps per solver call).  0 = all timed steps in one call; 1 = one "
                         "mrl_ch_substep per step.  Within a call the inverse z pass of a substep is fused with the forward z pass of the "
                         "next one (bit-identical fields; the intermediate real field is not written)")
    ap.add_argument("--carry", default="auto", choices=["auto", "on", "off"],
                    help="spectral carry-over (c-hat of a substep = ubar of the previous one, so only mu is transformed "
                         "forward: 2 slab transposes per substep instead of the reference's 3; results agree to rounding, see "
                         "include/marlin_hip.h).  auto = on for N > 1 (exchange-bound), off for N = 1 (the reference's data flow). "
                         "The metric is scored with the reference's 153 B/update either way")
    ap.add_argument("--force-slab", action="store_true",
                    help="run the slab pipeline (incl. the RCCL all-to-all calls) even with one rank: a single-GPU check of the N>1 code path")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    dev = local_rank % torch.cuda.device_count()   # (several ranks on one GPU only in the single-GPU smoke run)
    torch.cuda.set_device(dev)

    from marlin_amd.api import Context, ch_params

    dx = 8.0 * np.pi / 200.0   # examples/cahn_hilliard/cahnhilliard2.i:8-13
    shape = grid_for(world, args.n)
    L = [s * dx for s in shape]
    p = ch_params()             # f = 0.1 c^2 (c-1)^2, M = 0.2, kappa factor -0.001 (cahnhilliard2.i:61-91)
    sub_dt = 1e-3
    npts = int(np.prod(shape))

    slab = world > 1 or args.force_slab
    if slab:
        import torch.distributed as dist
        from marlin_amd.slab import SlabCahnHilliard

        if "RANK" not in os.environ:       # --force-slab without a launcher
            os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533", "RANK": "0", "WORLD_SIZE": "1"})

        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(args.backend)
        if args.compute_stream == "high":
            torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
        solver = SlabCahnHilliard(3, shape, L, p, world, rank, nsub=args.nsub, carry=args.carry != "off")
        step = solver.substep

        def steps(count):   # `count` substeps per solver call: the z passes between two substeps are one kernel
            solver.run(count, advance=True, advance_after=True)

        barrier = dist.barrier
        solver.set_initial(lambda count, offset: splitmix64_uniform(count, offset=offset))
    else:
        ctx = Context(3, shape, L)
        c = [torch.from_numpy(splitmix64_uniform(npts).reshape(shape)).cuda(), None]
        c[1] = torch.empty_like(c[0])
        Nh = [ctx.empty_spec(), ctx.empty_spec()]
        state = {"i": 0, "have_old": False}
        carried = ctx.empty_spec() if args.carry == "on" else None

        # Nh is the history ring of the AB2 scheme (two arrays): ring["head"] = slot of Nhat_old[0], the substep writes the other
        # slot; TensorBuffer::advanceState between substeps = the written slot becomes the head
        ring = {"head": 1, "n_old": 0}

        def step():
            i = state["i"]
            if state["have_old"]:
                ring["head"], ring["n_old"] = (ring["head"] + 1) % 2, 1
            order, new = ring["n_old"], (ring["head"] + 1) % 2
            mode = 0 if carried is None else (2 if state["have_old"] else 1)
            ctx.ch_substep(p, c[i], c[1 - i], Nh[new], [Nh[ring["head"]]] if order else [], order, sub_dt, cbar=carried, carry=mode)
            state["i"] = 1 - i
            state["have_old"] = True

        def steps(count):   # the same substeps, `count` per library call
            i = state["i"]
            if state["have_old"]:
                ring["head"], ring["n_old"] = (ring["head"] + 1) % 2, 1
            ring["head"], ring["n_old"] = ctx.ch_substeps(p, c[i], c[1 - i], Nh, ring["head"], ring["n_old"], 2, count, True, sub_dt)
            state["i"] = 1 - i
            state["have_old"] = True

        def barrier():
            pass

    def total_mass():
        m = (solver.current() if slab else c[state["i"]]).sum(dtype=torch.float64).reshape(1)
        if slab:
            m = m if args.backend == "nccl" else m.cpu()
            dist.all_reduce(m)
        return float(m.item())

    mass0 = total_mass()
    per_call = 1 if (not slab and args.carry == "on") else (args.substeps_per_call if args.substeps_per_call > 0 else args.steps)

    def run(nsteps):
        if per_call == 1:
            for _ in range(nsteps):
                step()
            return
        done = 0
        while done < nsteps:
            n = min(per_call, nsteps - done)
            steps(n)
            done += n

    run(args.warmup)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    single_ms = None
    if per_call != 1:
        # for comparison: the same substeps with one call each (mrl_ch_substep: every substep writes and re-reads c)
        n1 = min(args.steps, 50)
        step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n1):
            step()
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - t1) / n1 * 1e3
    if slab:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: the field must still be a bounded concentration field, and the scheme conserves mass exactly (the k = 0 mode has
    # Mbar = Lbar = 0), on every rank count -- a wrong exchange or a missed stream dependency shows up here
    cur = solver.current() if slab else c[state["i"]]
    assert torch.isfinite(cur).all() and 0.0 < float(cur.min()) and float(cur.max()) < 1.0
    mass1 = total_mass()
    assert abs(mass1 - mass0) <= 1e-11 * abs(mass0), (mass0, mass1)

    # per-kernel device time with HIP events on the launch stream (event pair per launch)
    prof_ctx = solver.ctx if slab else ctx
    prof_ctx.set_profiling(True)
    run(args.profile_steps)
    torch.cuda.synchronize()
    kernels = prof_ctx.get_profile()
    prof_ctx.set_profiling(False)

    if rank == 0:
        value = npts * args.steps / elapsed
        bpu = algorithmic_bytes_per_update(shape[2], 1)
        kernels = [k for k in kernels if k["launches"] > 0]
        for k in kernels:
            k["avg_ms"] = k["ms"] / k["launches"]
            k["gbps"] = k["bytes_per_launch"] / (k["avg_ms"] * 1e-3) / 1e9 if k["avg_ms"] > 0 else 0.0
        dom_k = max(kernels, key=lambda k: k["ms"]) if kernels else None
        roofline = None
        if dom_k:
            roofline = {
                "bound": "hbm",
                "kernel": dom_k["kernel"],
                "achieved": round(dom_k["gbps"], 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(dom_k["gbps"] / HBM_PEAK_GBPS, 4),
                "traffic": None,
                "traffic_source": None,
                "avg_launch_ms": round(dom_k["avg_ms"], 5),
                "algorithmic_bytes_per_launch": dom_k["bytes_per_launch"],
            }
        if roofline and not slab:
            tr, src = measured_traffic(dom_k["kernel"], args.n, f"<{args.n}, 1" if dom_k["kernel"] == "ch_C_x_fused" else None)
            roofline["traffic"], roofline["traffic_source"] = tr, src
        out = {
            "metric": "grid-point-updates/sec, 3-D Cahn-Hilliard semi-implicit spectral substep (AB2, fp64)",
            "value": value,
            "unit": "grid-point-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (splitmix64 uniform [0.44,0.56] initial concentration)",
            "config": {
                "workload": f"3D Cahn-Hilliard {shape[0]}x{shape[1]}x{shape[2]} fp64 semi-implicit spectral step, AB2, "
                            f"f=0.1c^2(c-1)^2, M=0.2, kappa=-0.001, sub_dt=1e-3",
                "grid": shape,
                "decomposition": "none" if not slab else f"slab x{world} ({'RCCL' if args.backend == 'nccl' else args.backend} all-to-all, {args.nsub} kz sub-blocks in flight)",
                "spectral_carry_over": bool(args.carry == "on" or (slab and args.carry == "auto")),
                "substeps_per_library_call": per_call,
                "ms_per_step_with_one_call_per_substep": single_ms,
            },
            "substep_algorithmic_bytes_per_update": bpu,
            "substep_achieved_GBps": value * bpu / 1e9,
            "substep_frac_of_hbm_peak": value * bpu / 1e9 / HBM_PEAK_GBPS,
            "roofline": roofline,
            "kernels": [{"kernel": k["kernel"], "avg_ms": round(k["avg_ms"], 5), "launches_per_step":
                         k["launches"] / args.profile_steps, "algorithmic_GBps": round(k["gbps"], 1)} for k in kernels],
        }
        if not slab and args.mech_grid > 0:
            # side measurement (BASELINE configs[2]): de Geus RVE Newton-CG, time per CG iteration, SURVEY 8(d) byte model
            from tools.mech_bench import run as mech_run
            del c, Nh
            torch.cuda.empty_cache()
            m = mech_run(args.mech_grid, 2, profile=False)
            out["mechanics"] = {"workload": f"de Geus finite-strain RVE {args.mech_grid}^3, Newton-CG (l_tol 1e-2)",
                                "ms_per_cg_iteration": m["ms_per_cg_iteration"], "cg_iterations_per_substep": m["cg_its"],
                                "algorithmic_bytes_per_point_per_cg_iteration": m["algorithmic_bytes_per_point_per_cg_iteration"],
                                "achieved_GBps": m["achieved_GBps"], "frac_of_hbm_peak": m["achieved_GBps"] / HBM_PEAK_GBPS}
        if not slab and args.cpu_steps > 0:
            out["cpu_baseline"] = cpu_baseline(shape, dx, args.cpu_steps)
    if slab:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner to the C stdout buffer; drain it first so that the JSON line is the last line
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
