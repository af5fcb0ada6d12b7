#!/usr/bin/env python3
"""Bank-conflict model of the Stockham exchanges of the z kernels (position-fastest LDS maps) with the per-instruction banking of
MI355X_MICROARCH.md section LDS: ds_write_b128 = 8 groups of 8 contiguous lanes on 32 banks (8 slots of 16 B), ds_read_b128 = 4 groups of
16 lanes {0-3,12-15,20-27}, {4-11,16-19,28-31}, (+32) on 64 banks (16 slots).  Prints LDS-array cycles per exchange for candidate maps."""
import sys

RGROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
RGROUPS = RGROUPS + [[x + 32 for x in g] for g in RGROUPS]
WGROUPS = [list(range(8 * g, 8 * g + 8)) for g in range(8)]


def cycles(slots, groups, nslots):
    tot = 0
    for g in groups:
        per = {}
        for lane in g:
            s = slots[lane]
            per.setdefault(s % nslots, set()).add(s)
        tot += max(len(v) for v in per.values())
    return tot


def model(N, P, radices, T, at):
    TPL = N // P
    nt = T * TPL
    waves = (nt + 63) // 64
    res = []
    NS = 1
    for R in radices[:-1]:
        S = P // R
        w_cyc = r_cyc = w_ideal = r_ideal = 0
        for wv in range(waves):
            lanes = [wv * 64 + k for k in range(64) if wv * 64 + k < nt]
            for i in range(S):
                for t in range(R):
                    slots = {}
                    for k, tid in enumerate(lanes):
                        q, l = tid % TPL, tid // TPL
                        b = q + i * TPL
                        p0 = (b // NS) * NS * R + (b % NS)
                        slots[k] = at(p0 + t * NS, l)
                    for k in range(len(lanes), 64):
                        slots[k] = slots[0]
                    w_cyc += cycles(slots, WGROUPS, 8)
                    w_ideal += 8
            for m in range(P):
                slots = {}
                for k, tid in enumerate(lanes):
                    q, l = tid % TPL, tid // TPL
                    slots[k] = at(q + m * TPL, l)
                for k in range(len(lanes), 64):
                    slots[k] = slots[0]
                r_cyc += cycles(slots, RGROUPS, 16)
                r_ideal += 4
        res.append((R, NS, w_cyc, w_ideal, r_cyc, r_ideal))
        NS *= R
    return res


def report(name, N, P, radices, T, at):
    tot = ideal = 0
    parts = []
    for R, NS, w, wi, r, ri in model(N, P, radices, T, at):
        parts.append(f"R{R}/Ns{NS}: write {w}/{wi} read {r}/{ri}")
        tot += w + r
        ideal += wi + ri
    print(f"{name:44s} N={N:4d} cycles {tot:6d} ideal {ideal:6d}  conflict share {100 * (tot - ideal) / tot:5.1f} %   " + "; ".join(parts))


PLANS = {512: (16, [8, 8, 8], 8), 256: (16, [16, 16], 8), 128: (16, [16, 8], 32), 200: (10, [10, 10, 2], 12), 64: (16, [8, 8], 64)}
for N, (P, rad, T) in PLANS.items():
    LP = N + N // 16
    report("MapLine: l*LP + p + (p>>4)", N, P, rad, T, lambda p, l, LP=LP: l * LP + p + (p >> 4))
    LP8 = N + N // 8
    report("pad per 8: l*LP8 + p + (p>>3)", N, P, rad, T, lambda p, l, LP8=LP8: l * LP8 + p + (p >> 3))
    LP2 = N + N // 16 + N // 128
    report("pad per 16 and per 128", N, P, rad, T, lambda p, l, LP2=LP2: l * LP2 + p + (p >> 4) + (p >> 7))
    for odd in (1, 3, 5):
        LPo = N + odd
        report(f"no pad in line, line pitch N+{odd}", N, P, rad, T, lambda p, l, LPo=LPo: l * LPo + p)
    LPx = N + 1
    report("xor swizzle p ^ ((p>>3)&7), pitch N+1", N, P, rad, T, lambda p, l, LPx=LPx: l * LPx + (p ^ ((p >> 3) & 7)))
    report("xor swizzle p ^ ((p>>4)&15), pitch N+1", N, P, rad, T, lambda p, l, LPx=LPx: l * LPx + (p ^ ((p >> 4) & 15)))
    print()


def twiddle_cycles(N, P, radices, T, wmap=lambda i: i):
    """ds_read_b128 of W[t*step] in the stages with Ns > 1 (lanes of a 16-lane group hold different q -> different twiddles)"""
    TPL = N // P
    nt = T * TPL
    waves = (nt + 63) // 64
    cyc = ideal = 0
    NS = 1
    for si, R in enumerate(radices):
        if NS > 1:
            S = P // R
            for wv in range(waves):
                lanes = [wv * 64 + k for k in range(64) if wv * 64 + k < nt]
                for i in range(S):
                    for t in range(1, R):
                        slots = {}
                        for k, tid in enumerate(lanes):
                            q = tid % TPL
                            b = q + i * TPL
                            step = (b % NS) * (N // (NS * R))
                            slots[k] = wmap(t * step)
                        for k in range(len(lanes), 64):
                            slots[k] = slots[0]
                        cyc += cycles(slots, RGROUPS, 16)
                        ideal += 4
        NS *= R
    return cyc, ideal


print("twiddle reads (W[t*step], plain table):")
for N, (P, rad, T) in PLANS.items():
    c, i = twiddle_cycles(N, P, rad, T)
    ex = sum(w + r for _, _, w, _, r, _ in model(N, P, rad, T, lambda p, l, LP=N + N // 16: l * LP + p + (p >> 4)))
    print(f"  N={N:4d}: {c} cycles, ideal {i}  (exchange cycles with MapLine: {ex})")

# search: at(p, l) = l*(N + c) + (p ^ ((p >> s) & m)) + (p >> a if a else 0)
print("best maps per N (exchange only):")
for N, (P, rad, T) in PLANS.items():
    best = []
    for c in (0, 1, 2, 3, 4, 5, 8, 9, 16, 17):
        for s in (0, 1, 2, 3, 4, 5, 6):
            for m in (0, 1, 3, 7, 15):
                for a in (0, 3, 4, 5):
                    if m == 0 and s:
                        continue
                    extra = (N >> a) if a else 0
                    LP = N + c + extra
                    def at(p, l, LP=LP, s=s, m=m, a=a):
                        return l * LP + (p ^ ((p >> s) & m)) + ((p >> a) if a else 0)
                    # (the swizzle must stay a bijection within the line: p ^ f(high bits) is, as long as the xor touches lower bits only)
                    if m and (m >> s) != 0 and s < 4 and (m.bit_length() > s):
                        continue
                    tot = sum(w + r for _, _, w, _, r, _ in model(N, P, rad, T, at))
                    best.append((tot, LP * T * 16, c, s, m, a))
    best.sort()
    ideal = sum(wi + ri for _, _, _, wi, _, ri in model(N, P, rad, T, lambda p, l: l * (N + 1) + p))
    print(f"  N={N}: ideal {ideal}; best:", best[:4])
