#!/usr/bin/env python3
"""Copy what tools/prof_r05.sh left under gpurun_out/ into profiles/ (the tracked evidence): counter traffic files (with the SHA-256 of
the kernel sources they were taken on), rocprofv3 summaries, the bench lines of the same job, and the kernel-sum-against-timed-step record."""
import json
import os
import re
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")


def last_json(path):
    lines = [l for l in open(path).read().strip().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def avg_us(summary, pattern):
    for line in open(summary):
        if re.search(pattern, line) and line.startswith("|"):
            return float(line.split("|")[4])
    return float("nan")


def main():
    for src, dst in [("traffic_r05_256.json", "traffic_ch256.json"), ("traffic_r05_512.json", "traffic_ch512.json"),
                     ("prof_mech_r05/traffic.json", "traffic_mech128.json"),
                     ("prof_r05_256/summary.md", "r05_rocprofv3_bench_256.md"), ("prof_r05_512/summary.md", "r05_rocprofv3_bench_512_1gpu.md"),
                     ("prof_r05_slab_local/summary.md", "r05_slab_local_512_over_8.md"), ("prof_mech_r05/summary.md", "r05_rocprofv3_mech128.md")]:
        shutil.copyfile(os.path.join(G, src), os.path.join(P, dst))
    for name in ["r05_bench_256_driver_command", "r05_bench_256_default", "r05_bench_512_1gpu", "r05_bench_mech128", "r05_bench_mech256",
                 "r05_bench_2ranks_one_gpu", "r05_slab_local_512_over_8"]:
        with open(os.path.join(P, name + ".json"), "w") as f:
            f.write(json.dumps(last_json(os.path.join(G, name + ".json"))) + "\n")
    out = ["Round 5: bench lines and rocprofv3 kernel traces of ONE gpurun job on ONE box (tools/prof_r05.sh), so that the kernel sums can be",
           "read against the timed steps (VERDICT r04 weak 4).  us per launch = rocprofv3 --kernel-trace --stats average.", ""]
    for tag, n, label in (("256", 256, "256^3 (headline)"), ("512", 512, "512^3 on one GPU")):
        s = os.path.join(G, f"prof_r05_{tag}", "summary.md")
        k = [avg_us(s, rf"k_ch_xfused<{n}, 1,"), avg_us(s, rf"k_pass<{n}, false, 2>"), avg_us(s, rf"k_z_inv_fwd<{n},"), avg_us(s, rf"k_pass<{n}, true, 1>")]
        prof_line = last_json(os.path.join(G, f"prof_r05_{tag}", "trace.log")) if os.path.exists(os.path.join(G, f"prof_r05_{tag}", "trace.log")) else None
        plain = last_json(os.path.join(G, "r05_bench_256_default.json" if n == 256 else "r05_bench_512_1gpu.json"))
        out.append(f"{label}: steady-state substep = k_ch_xfused {k[0]:.1f} + k_pass {k[1]:.1f} + k_z_inv_fwd {k[2]:.1f} + k_pass {k[3]:.1f} = {sum(k):.1f} us")
        prof_ms = f"{prof_line['ms_per_step'] * 1e3:.1f} us" if prof_line else "n/a"
        out.append(f"    timed step of the SAME (profiled) run: {prof_ms};  timed step of the plain run in the same job: {plain['ms_per_step'] * 1e3:.1f} us")
        out.append("")
    s = os.path.join(G, "prof_r05_slab_local", "summary.md")
    k = [avg_us(s, r"k_pass_sub_w<Wide512, false"), avg_us(s, r"k_ch_yfused<512"), avg_us(s, r"k_pass_sub_w<Wide512, true"), avg_us(s, r"k_z_inv_fwd<512")]
    out.append(f"rank-local kernels of 512^3 / 8 (tools/slab_local_bench.py 8 256 40 1 0 1 0): x forward {k[0]:.1f} + y fused {k[1]:.1f} + "
               f"x inverse {k[2]:.1f} + fused z {k[3]:.1f} = {sum(k):.1f} us")
    open(os.path.join(P, "r05_bench_vs_trace_same_job.txt"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
