"""The two-stage plans with per-stage ownership (marlin_amd/csrc/fft_two.h, fft_two_z.h) without a GPU: a numpy model of the index
scheme the kernels implement (pattern A -> one LDS exchange -> pattern B, the radix-15 butterfly, the staged twiddle table of the z
kernels), and the consistency of the plan tables and dispatch lists spread over four source files."""
import os
import re

import numpy as np

from tests.conftest import ROOT

CSRC = os.path.join(ROOT, "marlin_amd", "csrc")


def _read(name):
    return open(os.path.join(CSRC, name)).read()


def _plans(text, macro):
    return {int(m[0]): tuple(int(v) for v in m[1:]) for m in re.findall(rf"^{macro}\((\d+), (\d+), (\d+), (\d+)\)", text, flags=re.M)}


def _two_stage(x, RA, RB, staged):
    """fft2 / fft2z: threads q < RB hold x[q + RB t] (t < RA) and do one radix-RA butterfly; outputs go to position RA q + t of the
    exchange; threads q < RA read q + RA t' (t' < RB), multiply by w_N^(t' q) and do one radix-RB butterfly: X[q + RA t']"""
    N = RA * RB
    ex = np.zeros(N, complex)
    for q in range(RB):
        ex[RA * q + np.arange(RA)] = np.fft.fft(x[q + RB * np.arange(RA)])
    tw = np.exp(-2j * np.pi * np.arange(N) / N)
    # the z kernels stage the twiddles [t - 1][q]: entry (t - 1) RA + q = tw[t q]  (tw2_issue)
    table = np.array([tw[(s // RA + 1) * (s % RA)] for s in range((RB - 1) * RA)])
    out = np.zeros(N, complex)
    for q in range(RA):
        b = ex[q + RA * np.arange(RB)].copy()
        for t in range(1, RB):
            b[t] *= table[(t - 1) * RA + q] if staged else tw[t * q]
        out[q + RA * np.arange(RB)] = np.fft.fft(b)
    return out


def test_index_scheme_of_every_two_stage_plan():
    rng = np.random.default_rng(0)
    strided = _plans(_read("fft_two.h"), "MRL_PLAN2")
    zplans = _plans(_read("fft_two_z.h"), "MRL_ZPLAN2")
    assert set(strided) == {120, 150, 160, 180, 240, 300, 320, 400, 192} and set(zplans) == {120, 150, 160, 180, 240, 300, 320, 192}
    for plans, staged in ((strided, False), (zplans, True)):
        for n, (r0, r1, _) in plans.items():
            assert r0 * r1 == n and max(r0, r1) <= 20
            x = rng.random(n) + 1j * rng.random(n)
            want = np.fft.fft(x)
            # forward: A -> B with radices (r0, r1); the inverse passes run the same routine with the radices swapped: B -> A
            assert np.abs(_two_stage(x, r0, r1, staged) - want).max() <= 1e-12 * np.abs(want).max()
            assert np.abs(_two_stage(x, r1, r0, staged) - want).max() <= 1e-12 * np.abs(want).max()
            # the inverse through the swap trick: swap re / im, forward transform, swap back = N * ifft
            sw = lambda z: z.imag + 1j * z.real
            assert np.abs(sw(_two_stage(sw(want), r1, r0, staged)) / n - x).max() <= 1e-13


def test_radix_15_butterfly_decomposition_and_constants():
    """bfly<15> (fft_pow2.h): n = 5 n1 + n2, k = k1 + 3 k2 -- radix 3 over n1, twiddle W15^(n2 k1), radix 5 over n2 -- with the
    constants as they are written in the source"""
    src = _read("fft_pow2.h")
    body = src[src.index("__device__ __forceinline__ void bfly<15>"):]
    body = body[:body.index("kcplx r[15];")]
    consts = {}
    for slot, c, s, j in re.findall(r"a\[(\d+)\] = cmul\(a\[\d+\], mkc\((-?[\d.]+), (-?[\d.]+)\)\);\s*// W15\^(\d+)", body):
        consts[int(slot)] = (complex(float(c), float(s)), int(j))
    assert sorted(consts) == [6, 7, 8, 9, 11, 12, 13, 14]
    for slot, (w, j) in consts.items():
        n2, k1 = slot % 5, slot // 5
        assert j == n2 * k1 and abs(w - np.exp(-2j * np.pi * j / 15)) <= 4e-16     # (numpy rounds the argument; the literals come from long double)
    rng = np.random.default_rng(1)
    a = rng.random(15) + 1j * rng.random(15)
    want = np.fft.fft(a)
    w = a.copy()
    for n2 in range(5):
        w[[n2, n2 + 5, n2 + 10]] = np.fft.fft(w[[n2, n2 + 5, n2 + 10]])
    for slot, (c, _) in consts.items():
        w[slot] *= c
    got = np.zeros(15, complex)
    for k1 in range(3):
        got[k1 + 3 * np.arange(5)] = np.fft.fft(w[5 * k1:5 * k1 + 5])
    assert np.abs(got - want).max() <= 1e-14


def test_plan_tables_and_dispatch_lists_agree():
    two, twoz, planned, expr = _read("fft_two.h"), _read("fft_two_z.h"), _read("ch_planned.hip"), _read("expr.hip")
    strided = _plans(two, "MRL_PLAN2")
    zplans = _plans(twoz, "MRL_ZPLAN2")
    listed = lambda text, fn: {int(v) for v in re.findall(r"n == (\d+)", text[text.index(fn):].split("}")[0])}
    lens = listed(two, "constexpr bool two_stage_len")
    # passes and z kernels: the same lengths; 400 (x pass) and 192 (x pass, fused z pass) are fused-family lengths that ch_fused.hip
    # serves with single two-stage kernels
    assert lens == set(zplans) - {192} == listed(twoz, "constexpr bool two_stage_z_len")
    assert set(strided) == lens | {400, 192}
    assert "two_stage_len(n) || n == 400" in two
    sw = planned[planned.index("#define MRL_SWITCH_N2(n, CALL)"):planned.index("#define MRL_SWITCH_N2X")]
    assert {int(v) for v in re.findall(r"case (\d+):", sw)} == lens                         # run-time length -> template (ch_planned.hip)
    swx = planned[planned.index("#define MRL_SWITCH_N2X"):planned.index("// experiment bit 1 << 29")]
    assert {int(v) for v in re.findall(r"case (\d+):", swx)} == {400}
    assert {int(v) for v in re.findall(r"MRL_Z2\((\d+)\)", expr)} - {0} == lens              # run-time compiled z kernels (expr.hip)
    for n, (r0, r1, t) in strided.items():                                                 # tile shapes of the strided passes
        tpl = max(r0, r1)
        assert t * tpl <= 256 and 16 * (n + n * t) + 8 * n <= 80 * 1024                    # threads; LDS for two workgroups per CU
    for n, (r0, r1, lpb) in zplans.items():
        tpl, lp = max(r0, r1), n + n // 16 + 1
        assert lpb * tpl <= 256 and (r1 - 1) * r0 <= n and (r0 - 1) * r1 <= n              # staged twiddle tables fit N entries each
        assert 16 * (2 * n + lpb * lp) <= 80 * 1024
