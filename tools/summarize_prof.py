#!/usr/bin/env python3
"""Summarise a tools/profile_gpu.sh output directory: per-kernel time (rocprofv3 --stats) and HBM
traffic per launch from the FETCH_SIZE / WRITE_SIZE passes (gfx950: FETCH_SIZE counts half of a wide
coalesced streaming read -> doubled, MI355X_MICROARCH.md 'HBM')."""
import csv
import glob
import os
import sys
from collections import defaultdict


def kernel_sources_sha256():
    """hash of every kernel source of the library (marlin_amd/csrc/*.hip, *.h): bench.py recomputes it and reports whether the
    committed counter file still describes the kernels it is timing"""
    import hashlib
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "marlin_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(here)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(here, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()


def short(name):
    name = name.split("(")[0]
    return name.replace("mrl::p2::", "").replace("mrl::", "").replace("void ", "")[:60]


def main(root):
    import json
    traffic = {}
    print(f"# rocprofv3 summary ({os.path.basename(root)})\n")
    stats = glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        print("## kernel time (rocprofv3 --kernel-trace --stats)\n")
        print("| kernel | calls | total ms | avg us | % |")
        print("|---|---|---|---|---|")
        with open(stats[0]) as f:
            for row in csv.DictReader(f):
                print(f"| {short(row['Name'])} | {row['Calls']} | {float(row['TotalDurationNs'])/1e6:.3f} | "
                      f"{float(row['AverageNs'])/1e3:.2f} | {float(row['Percentage']):.1f} |")
    for label, sub, scale in (("FETCH_SIZE", "fetch", 2.0), ("WRITE_SIZE", "write", 1.0)):
        files = glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        agg = defaultdict(lambda: [0, 0.0])
        with open(files[0]) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != label:
                    continue
                k = short(row["Kernel_Name"])
                agg[k][0] += 1
                agg[k][1] += float(row["Counter_Value"])
        print(f"\n## {label} per launch (KiB counter x1024{' x2 (gfx950 streaming-read correction)' if scale == 2.0 else ''})\n")
        print("| kernel | launches | MB per launch |")
        print("|---|---|---|")
        for k, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f"| {k} | {n} | {tot / n * 1024.0 * scale / 1e6:.1f} |")
            traffic.setdefault(k, {})[label] = tot / n * 1024.0 * scale
    if traffic:
        # per-launch HBM bytes (FETCH_SIZE x2 + WRITE_SIZE) per kernel: read back by bench.py as roofline.traffic
        out = {k: {"fetch_bytes": v.get("FETCH_SIZE"), "write_bytes": v.get("WRITE_SIZE")} for k, v in traffic.items()
               if k.startswith("k_")}
        out["_kernel_sources_sha256"] = kernel_sources_sha256()   # ties the counters to the kernels they were measured on
        with open(os.path.join(root, "traffic.json"), "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1])
