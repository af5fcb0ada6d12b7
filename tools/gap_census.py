#!/usr/bin/env python3
"""Where a stream idles: reads a rocprofv3 kernel trace (…_kernel_trace.csv), orders the dispatches by start time and lists the
gaps between the end of one kernel and the start of the next, grouped by (previous kernel -> next kernel).
usage: gap_census.py <dir or csv> [min gap us = 8]"""
import collections
import csv
import glob
import os
import sys


def short(name):
    name = name.replace("void ", "").replace("mrl::", "")
    cut = name.find("(")
    return (name if cut < 0 else name[:cut])[:48]


def main():
    src = sys.argv[1]
    min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
    files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    if not rows:
        sys.exit("no dispatches found")
    busy = sum(e - s for s, e, _ in rows)
    span = rows[-1][1] - rows[0][0]
    gaps = collections.defaultdict(lambda: [0, 0.0])
    small = 0.0
    for (s0, e0, k0), (s1, e1, k1) in zip(rows, rows[1:]):
        g = (s1 - e0) * 1e-3
        if g >= min_gap:
            gaps[(k0, k1)][0] += 1
            gaps[(k0, k1)][1] += g
        elif g > 0:
            small += g
    print(f"{len(rows)} dispatches, span {span * 1e-6:.2f} ms, kernels busy {busy * 1e-6:.2f} ms ({busy / span:.1%}); "
          f"gaps below {min_gap:g} us: {small * 1e-3:.2f} ms in all")
    per = collections.defaultdict(lambda: [0, 0.0])
    for s0, e0, k0 in rows:
        per[k0][0] += 1
        per[k0][1] += (e0 - s0) * 1e-3
    print(f"{'kernel':60s} {'count':>6s} {'total ms':>9s} {'avg us':>8s}")
    for k0, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:30]:
        print(f"{k0:60s} {c:6d} {t * 1e-3:9.3f} {t / c:8.1f}")
    print()
    print(f"{'previous kernel -> next kernel':100s} {'count':>6s} {'total ms':>9s} {'avg us':>8s}")
    for (k0, k1), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{(k0 + ' -> ' + k1):100s} {c:6d} {t * 1e-3:9.3f} {t / c:8.1f}")


if __name__ == "__main__":
    main()
