// Power-of-two Stockham FFT building blocks for gfx950 (fp64, no MFMA: the path is HBM-bound).
//
// Every thread owns P complex values of one line (P = 16 for the power-of-two sizes, 10 for the 2^a 5^b sizes):
// positions q + m*TPL, m = 0..P-1, where q = thread index within the line (TPL = N/P threads per line).  A
// transform is 2 to 4 radix stages (16/8/4/10/5/2 point butterflies in registers); between stages the line is
// exchanged through LDS in natural (Stockham autosort) order, after which each thread again owns q + m*TPL.  The
// first stage therefore loads straight from HBM and the last stage stores straight to HBM with the same index
// pattern; a workgroup holds T lines (4096 points = 64 KiB of exchange space for the power-of-two sizes).
// Twiddles exp(-2 pi i k/N) are staged once per workgroup into LDS (exact table values, no
// recurrences).  Inverse transforms use the swap trick: ifft(x) = swap(fft(swap(x))).
#pragma once
#include "mrl_internal.h"

// Scalar type of the kernels built from these headers: double (the product path) unless a translation unit defines MRL_KREAL = float
// and MRL_P2NS before including them (ch_fused_f32.hip: the fp32 instantiation of the fused Cahn-Hilliard path, its own namespace).
#ifndef MRL_KREAL
#define MRL_KREAL double
#endif
#ifndef MRL_P2NS
#define MRL_P2NS p2
#endif

namespace mrl {
namespace MRL_P2NS {

typedef MRL_KREAL kreal;
template <class T>
struct Complex2;
template <>
struct Complex2<double> {
  typedef double2 type;
};
template <>
struct Complex2<float> {
  typedef float2 type;
};
typedef Complex2<kreal>::type kcplx;  // interleaved (re, im)
__device__ __forceinline__ kcplx mkc(kreal x, kreal y) {
  kcplx r;
  r.x = x;
  r.y = y;
  return r;
}


__device__ __forceinline__ kcplx cadd(kcplx a, kcplx b) { return mkc(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ kcplx csub(kcplx a, kcplx b) { return mkc(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ kcplx cmul(kcplx a, kcplx b) {
  return mkc(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ kcplx mul_mi(kcplx a) { return mkc(a.y, -a.x); }  // * (-i)
__device__ __forceinline__ kcplx cswap(kcplx a) { return mkc(a.y, a.x); }

// Plan<N>: P = points per thread, up to four radix stages r0..r3 (1 = unused; every radix divides P), T = lines per
// workgroup.  TPL = N/P threads own one line, a workgroup is NT = T*TPL threads (256 for the power-of-two sizes,
// 240-255 for the 2^a 5^b sizes) and exchanges T*N points through LDS.
template <int N>
struct Plan;
#define MRL_PLAN(N_, P_, R0, R1, R2, R3, T_)                                              \
  template <>                                                                             \
  struct Plan<N_> {                                                                       \
    static constexpr int P = P_, r0 = R0, r1 = R1, r2 = R2, r3 = R3, T = T_;              \
    static constexpr int ns = 1 + (R1 > 1) + (R2 > 1) + (R3 > 1);                         \
    static constexpr int TPL = N_ / P_, NT = T_ * (N_ / P_);                              \
    static_assert(R0 * R1 * R2 * R3 == N_ && N_ % P_ == 0 && NT <= 256, "bad plan");      \
  };
MRL_PLAN(32, 16, 16, 2, 1, 1, 128)
// (fp32 instantiation, ch_fused_f32.hip: the same plans.  Round 4 tried the two ways to 256-byte pieces per line element for 8-byte
// complex values -- 32 points per thread: the fused kernels spill 50-110 VGPRs; twice the lines per tile with 512-thread workgroups: no
// spills, but one workgroup per CU in lockstep: 0.205-0.209 ms per 256^3 substep against 0.193-0.198 ms with these plans (first-region
// figures; steady state 0.174-0.176 ms).  A register budget of three waves per SIMD for the float x-fused kernel (176 -> 168 VGPRs, 11-25
// spilled): 0.186 against 0.174-0.176 ms in interleaved runs.  Kept as is.)
MRL_PLAN(64, 16, 8, 8, 1, 1, 64)
MRL_PLAN(128, 16, 16, 8, 1, 1, 32)
MRL_PLAN(256, 16, 16, 16, 1, 1, 16)
MRL_PLAN(512, 16, 8, 8, 8, 1, 8)
MRL_PLAN(1024, 16, 16, 8, 8, 1, 4)
// long lines of 1-D / 2-D problems: the strided passes only gather 32 / 16 bytes per row here (their LDS tile holds T whole lines)
MRL_PLAN(2048, 16, 16, 16, 8, 1, 2)
MRL_PLAN(4096, 16, 16, 16, 16, 1, 1)
// sizes 2^a 5^b (the reference's own examples run 100^3 and 200^3 grids): radix 10 / 5 / 2, 10 points per thread
MRL_PLAN(40, 10, 10, 2, 2, 1, 64)
MRL_PLAN(50, 10, 10, 5, 1, 1, 51)
MRL_PLAN(80, 10, 10, 2, 2, 2, 32)
MRL_PLAN(100, 10, 10, 10, 1, 1, 25)
MRL_PLAN(200, 10, 10, 10, 2, 1, 12)   // (16 lines = 256-byte pieces, 320-thread workgroups: 200^3 substep 0.196 against 0.172-0.174 ms, round 4)
MRL_PLAN(250, 10, 10, 5, 5, 1, 10)
MRL_PLAN(400, 10, 10, 10, 2, 2, 6)
MRL_PLAN(500, 10, 10, 10, 5, 1, 5)
MRL_PLAN(1000, 10, 10, 10, 10, 1, 2)
// sizes 2^a 3^b: radix 12 / 4 / 2, 12 points per thread
MRL_PLAN(48, 12, 12, 4, 1, 1, 64)
MRL_PLAN(96, 12, 12, 4, 2, 1, 32)
MRL_PLAN(144, 12, 12, 12, 1, 1, 21)
MRL_PLAN(192, 12, 12, 4, 4, 1, 16)
MRL_PLAN(384, 12, 12, 4, 4, 2, 8)
MRL_PLAN(768, 12, 12, 4, 4, 4, 4)
// sizes 2^a 3^b 5^c with all three primes (120, 150, 240, 270 ... of the reference's own inputs, test/tests/solvers/diagonal.i): radix
// 30 first, then 2 / 3 / 5 / 10; 30 points per thread.  120 complex registers per array: plain transforms only (z passes, strided
// passes); the fused Cahn-Hilliard kernels, which hold three to four arrays, are not instantiated for them (ch_planned.hip)
MRL_PLAN(60, 30, 30, 2, 1, 1, 64)
MRL_PLAN(90, 30, 30, 3, 1, 1, 42)
MRL_PLAN(120, 30, 30, 2, 2, 1, 32)
MRL_PLAN(150, 30, 30, 5, 1, 1, 25)
MRL_PLAN(180, 30, 30, 3, 2, 1, 21)
MRL_PLAN(240, 30, 30, 2, 2, 2, 16)
MRL_PLAN(270, 30, 30, 3, 3, 1, 14)
MRL_PLAN(300, 30, 30, 10, 1, 1, 12)
MRL_PLAN(360, 30, 30, 3, 2, 2, 10)
MRL_PLAN(450, 30, 30, 5, 3, 1, 8)
MRL_PLAN(600, 30, 30, 10, 2, 1, 6)
// sizes 2^a 5 with a >= 5 (160, 320, 640, 1280: 2^a 5 needs too many radix-2 stages behind a radix-10 one): radix 20 first, then 4 / 2;
// 20 points per thread = 80 registers per array: like the radix-30 lengths, plain transforms only (ch_planned.hip)
// (128-thread workgroups: the LDS tile of T lines stays at 40 KB as for the other plans)
MRL_PLAN(160, 20, 20, 4, 2, 1, 16)
MRL_PLAN(320, 20, 20, 4, 4, 1, 8)
MRL_PLAN(640, 20, 20, 4, 4, 2, 4)
MRL_PLAN(1280, 20, 20, 4, 4, 4, 2)
MRL_PLAN(800, 20, 20, 10, 4, 1, 3)
// 2^a 3^b with b >= 2 (72 ... 1152): the 12-point plans of the fused family, here for the plain kernels only (instantiating the fused
// kernels for every further length is what costs build time and binary size)
MRL_PLAN(72, 12, 12, 3, 2, 1, 32)
MRL_PLAN(216, 12, 12, 3, 3, 2, 12)
MRL_PLAN(288, 12, 12, 12, 2, 1, 8)
MRL_PLAN(432, 12, 12, 12, 3, 1, 6)
MRL_PLAN(576, 12, 12, 12, 4, 1, 4)
MRL_PLAN(864, 12, 12, 12, 3, 2, 3)
MRL_PLAN(1152, 12, 12, 12, 4, 2, 2)

// Lines per workgroup of the z kernels (k_z_fwd / k_z_inv / k_z_inv_fwd): their lines are contiguous in memory, so the tile
// width T of the strided passes (T adjacent lines = one coalesced segment) buys them nothing, while smaller workgroups mean more
// of them per CU in different phases (load / transform / store).  Measured at 256^3 with 8 instead of 16 lines: z inverse+forward
// 80 -> 75 us, substep 0.329 -> 0.324 ms in a same-box A/B; the strided passes lose 8-15 % with the same change, hence a
// separate constant.  Halving or quartering the lines at 128 / 200 / 384 / 512 points changed nothing measurable.
template <int N>
struct ZPlan {
  static constexpr int T = Plan<N>::T, NT = Plan<N>::NT;
};
template <>
struct ZPlan<256> {
  static constexpr int T = 8, NT = 8 * Plan<256>::TPL;
};
// ... and of the fused inverse + forward z kernel (k_z_inv_fwd): three transforms between one burst of loads and the stores, so more,
// smaller workgroups in different phases pay at 512 points (tools/zpass_probe.hip, same box: 8 -> 4 lines 84.7 -> 82.1 us on the
// slab-local 512^3 / 8 arrays; 256 points: 72.2 vs 71.2 us in the probe, and 0.3052 / 0.3051 vs 0.3039 / 0.3053 ms per substep in two
// interleaved bench.py runs of each build on one box: within the noise, unchanged)
template <int N>
struct ZPlanEA {
  static constexpr int T = ZPlan<N>::T, NT = ZPlan<N>::NT;
};
template <>
struct ZPlanEA<512> {
  static constexpr int T = 4, NT = 4 * Plan<512>::TPL;
};
#undef MRL_PLAN

__device__ __forceinline__ void bfly4(kcplx &a0, kcplx &a1, kcplx &a2, kcplx &a3) {
  const kcplx t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = mul_mi(csub(a1, a3));
  a0 = cadd(t0, t2);
  a1 = cadd(t1, t3);
  a2 = csub(t0, t2);
  a3 = csub(t1, t3);
}

// natural-order DFT of R register values a[0..R-1]
template <int R>
__device__ __forceinline__ void bfly(kcplx (&a)[R]);

template <>
__device__ __forceinline__ void bfly<4>(kcplx (&a)[4]) {
  bfly4(a[0], a[1], a[2], a[3]);
}

template <>
__device__ __forceinline__ void bfly<8>(kcplx (&a)[8]) {
  const kreal h = 0.70710678118654752440;
  kcplx e0 = a[0], e1 = a[2], e2 = a[4], e3 = a[6];
  kcplx o0 = a[1], o1 = a[3], o2 = a[5], o3 = a[7];
  bfly4(e0, e1, e2, e3);
  bfly4(o0, o1, o2, o3);
  o1 = mkc(h * (o1.x + o1.y), h * (o1.y - o1.x));    // * W8^1 = (h, -h)
  o2 = mul_mi(o2);                                            // * W8^2 = -i
  o3 = mkc(h * (o3.y - o3.x), -h * (o3.x + o3.y));   // * W8^3 = (-h, -h)
  a[0] = cadd(e0, o0);
  a[4] = csub(e0, o0);
  a[1] = cadd(e1, o1);
  a[5] = csub(e1, o1);
  a[2] = cadd(e2, o2);
  a[6] = csub(e2, o2);
  a[3] = cadd(e3, o3);
  a[7] = csub(e3, o3);
}

template <>
__device__ __forceinline__ void bfly<16>(kcplx (&a)[16]) {
  const kreal c1 = 0.92387953251128675613, s1 = 0.38268343236508977173, h = 0.70710678118654752440;
  // inner DFT-4 over n1 for each n2: (a[n2], a[n2+4], a[n2+8], a[n2+12]) -> A[n2][k1] at a[n2+4*k1]
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) bfly4(a[n2], a[n2 + 4], a[n2 + 8], a[n2 + 12]);
  // twiddles W16^(n2*k1)
  a[5] = cmul(a[5], mkc(c1, -s1));                      // 1*1
  a[9] = mkc(h * (a[9].x + a[9].y), h * (a[9].y - a[9].x));  // 1*2 -> W16^2
  a[13] = cmul(a[13], mkc(s1, -c1));                    // 1*3
  a[6] = mkc(h * (a[6].x + a[6].y), h * (a[6].y - a[6].x));  // 2*1 -> W16^2
  a[10] = mul_mi(a[10]);                                         // 2*2 -> W16^4
  a[14] = mkc(h * (a[14].y - a[14].x), -h * (a[14].x + a[14].y));  // 2*3 -> W16^6
  a[7] = cmul(a[7], mkc(s1, -c1));                      // 3*1 -> W16^3
  a[11] = mkc(h * (a[11].y - a[11].x), -h * (a[11].x + a[11].y));  // 3*2 -> W16^6
  a[15] = cmul(a[15], mkc(-c1, s1));                    // 3*3 -> W16^9
  // outer DFT-4 over n2 for each k1: X[k1 + 4*k2] lands at a[4*k1 + k2]
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) bfly4(a[4 * k1], a[4 * k1 + 1], a[4 * k1 + 2], a[4 * k1 + 3]);
  // transpose to natural order
  kcplx t;
#define MRL_SWAP(i, j) \
  t = a[i];            \
  a[i] = a[j];         \
  a[j] = t;
  MRL_SWAP(1, 4) MRL_SWAP(2, 8) MRL_SWAP(3, 12) MRL_SWAP(6, 9) MRL_SWAP(7, 13) MRL_SWAP(11, 14)
#undef MRL_SWAP
}

template <>
__device__ __forceinline__ void bfly<3>(kcplx (&a)[3]) {
  const kreal s = 0.86602540378443864676;  // sin(pi/3)
  const kcplx t = cadd(a[1], a[2]);
  const kreal hf = 0.5;
  const kcplx m = mkc(a[0].x - hf * t.x, a[0].y - hf * t.y);
  const kcplx n = mkc(s * (a[1].x - a[2].x), s * (a[1].y - a[2].y));
  a[0] = cadd(a[0], t);
  a[1] = mkc(m.x + n.y, m.y - n.x);  // m - i n
  a[2] = mkc(m.x - n.y, m.y + n.x);  // m + i n
}

// radix 12 = 4 x 3 (Cooley-Tukey): n = 3 n1 + n2, k = k1 + 4 k2
template <>
__device__ __forceinline__ void bfly<12>(kcplx (&a)[12]) {
  const kreal h = 0.5, s = 0.86602540378443864676;
  // radix 4 over n1 for each n2: (a[n2], a[n2+3], a[n2+6], a[n2+9]) -> A[n2][k1] left in the same slots (k1 = slot/3)
#pragma unroll
  for (int n2 = 0; n2 < 3; ++n2) bfly4(a[n2], a[n2 + 3], a[n2 + 6], a[n2 + 9]);
  // twiddles W12^(n2*k1): slot n2 + 3*k1
  a[4] = cmul(a[4], mkc(s, -h));     // 1*1 -> W12^1
  a[7] = cmul(a[7], mkc(h, -s));     // 1*2 -> W12^2
  a[10] = mul_mi(a[10]);                      // 1*3 -> W12^3 = -i
  a[5] = cmul(a[5], mkc(h, -s));     // 2*1 -> W12^2
  a[8] = cmul(a[8], mkc(-h, -s));    // 2*2 -> W12^4
  a[11] = mkc(-a[11].x, -a[11].y);   // 2*3 -> W12^6 = -1
  // radix 3 over n2 for each k1: X[k1 + 4 k2]
  kcplx r[12];
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    kcplx b[3] = {a[3 * k1], a[3 * k1 + 1], a[3 * k1 + 2]};
    bfly<3>(b);
#pragma unroll
    for (int k2 = 0; k2 < 3; ++k2) r[k1 + 4 * k2] = b[k2];
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) a[i] = r[i];
}

template <>
__device__ __forceinline__ void bfly<2>(kcplx (&a)[2]) {
  const kcplx t = a[0];
  a[0] = cadd(t, a[1]);
  a[1] = csub(t, a[1]);
}

// radix 5: X1,4 = m1 -+ i n1, X2,3 = m2 -+ i n2 (forward sign)
__device__ __forceinline__ void bfly5(kcplx &a0, kcplx &a1, kcplx &a2, kcplx &a3, kcplx &a4) {
  const kreal c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;  // cos(2 pi/5), cos(4 pi/5)
  const kreal s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;   // sin(2 pi/5), sin(4 pi/5)
  const kcplx t1 = cadd(a1, a4), t2 = cadd(a2, a3), t3 = csub(a1, a4), t4 = csub(a2, a3);
  const kcplx m1 = mkc(a0.x + c1 * t1.x + c2 * t2.x, a0.y + c1 * t1.y + c2 * t2.y);
  const kcplx m2 = mkc(a0.x + c2 * t1.x + c1 * t2.x, a0.y + c2 * t1.y + c1 * t2.y);
  const kcplx n1 = mkc(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y);
  const kcplx n2 = mkc(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y);
  a0 = mkc(a0.x + t1.x + t2.x, a0.y + t1.y + t2.y);
  a1 = mkc(m1.x + n1.y, m1.y - n1.x);  // m1 - i n1
  a4 = mkc(m1.x - n1.y, m1.y + n1.x);  // m1 + i n1
  a2 = mkc(m2.x + n2.y, m2.y - n2.x);
  a3 = mkc(m2.x - n2.y, m2.y + n2.x);
}

template <>
__device__ __forceinline__ void bfly<5>(kcplx (&a)[5]) {
  bfly5(a[0], a[1], a[2], a[3], a[4]);
}

// radix 10 = 2 x 5 (Cooley-Tukey): n = 5 n1 + n2, k = k1 + 2 k2
template <>
__device__ __forceinline__ void bfly<10>(kcplx (&a)[10]) {
  // W10^j = exp(-2 pi i j / 10), j = 1..4
  const kreal c1 = 0.80901699437494742410, s1 = 0.58778525229247312917;  // cos, sin(pi/5)
  const kreal c2 = 0.30901699437494742410, s2 = 0.95105651629515357212;  // cos, sin(2 pi/5)
  kcplx e[5], o[5];
#pragma unroll
  for (int n2 = 0; n2 < 5; ++n2) {
    e[n2] = cadd(a[n2], a[n2 + 5]);  // k1 = 0
    o[n2] = csub(a[n2], a[n2 + 5]);  // k1 = 1
  }
  o[1] = cmul(o[1], mkc(c1, -s1));
  o[2] = cmul(o[2], mkc(c2, -s2));
  o[3] = cmul(o[3], mkc(-c2, -s2));
  o[4] = cmul(o[4], mkc(-c1, -s1));
  bfly5(e[0], e[1], e[2], e[3], e[4]);  // X[2 k2]
  bfly5(o[0], o[1], o[2], o[3], o[4]);  // X[1 + 2 k2]
#pragma unroll
  for (int k2 = 0; k2 < 5; ++k2) {
    a[2 * k2] = e[k2];
    a[2 * k2 + 1] = o[k2];
  }
}

// radix 15 = 3 x 5 (Cooley-Tukey): n = 5 n1 + n2, k = k1 + 3 k2 -- the two-stage plans of fft_two.h / fft_two_z.h
template <>
__device__ __forceinline__ void bfly<15>(kcplx (&a)[15]) {
  // radix 3 over n1 for each n2: (a[n2], a[n2+5], a[n2+10]) -> A[n2][k1] left in slot n2 + 5 k1
#pragma unroll
  for (int n2 = 0; n2 < 5; ++n2) {
    kcplx b[3] = {a[n2], a[n2 + 5], a[n2 + 10]};
    bfly<3>(b);
    a[n2] = b[0];
    a[n2 + 5] = b[1];
    a[n2 + 10] = b[2];
  }
  // twiddles W15^(n2 k1), k1 = 1, 2
  a[6] = cmul(a[6], mkc(0.913545457642600895493, -0.406736643075800207754));    // W15^1
  a[7] = cmul(a[7], mkc(0.669130606358858213826, -0.743144825477394235010));    // W15^2
  a[8] = cmul(a[8], mkc(0.309016994374947424076, -0.951056516295153572111));    // W15^3
  a[9] = cmul(a[9], mkc(-0.104528463267653471389, -0.994521895368273336916));   // W15^4
  a[11] = cmul(a[11], mkc(0.669130606358858213826, -0.743144825477394235010));  // W15^2
  a[12] = cmul(a[12], mkc(-0.104528463267653471389, -0.994521895368273336916)); // W15^4
  a[13] = cmul(a[13], mkc(-0.809016994374947424104, -0.587785252292473129135)); // W15^6
  a[14] = cmul(a[14], mkc(-0.978147600733805637930, 0.207911690817759337087));  // W15^8
  // radix 5 over n2 for each k1: X[k1 + 3 k2]
  kcplx r[15];
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1) {
    bfly5(a[5 * k1], a[5 * k1 + 1], a[5 * k1 + 2], a[5 * k1 + 3], a[5 * k1 + 4]);
#pragma unroll
    for (int k2 = 0; k2 < 5; ++k2) r[k1 + 3 * k2] = a[5 * k1 + k2];
  }
#pragma unroll
  for (int i = 0; i < 15; ++i) a[i] = r[i];
}

// radix 20 = 4 x 5 (Cooley-Tukey): n = 5 n1 + n2, k = k1 + 4 k2 -- first stage of the lengths 2^a 5 with a >= 5 (160, 320, 640, 1280)
template <>
__device__ __forceinline__ void bfly<20>(kcplx (&a)[20]) {
  // radix 4 over n1 for each n2: (a[n2], a[n2+5], a[n2+10], a[n2+15]) -> A[n2][k1] left in slot 5 k1 + n2
#pragma unroll
  for (int n2 = 0; n2 < 5; ++n2) bfly4(a[n2], a[n2 + 5], a[n2 + 10], a[n2 + 15]);
  // twiddles W20^(n2 k1)
  a[6] = cmul(a[6], mkc(0.951056516295153531182, -0.309016994374947395752));  // W20^1
  a[7] = cmul(a[7], mkc(0.809016994374947451263, -0.587785252292473137103));  // W20^2
  a[8] = cmul(a[8], mkc(0.587785252292473137103, -0.809016994374947451263));  // W20^3
  a[9] = cmul(a[9], mkc(0.309016994374947451263, -0.951056516295153531182));  // W20^4
  a[11] = cmul(a[11], mkc(0.809016994374947451263, -0.587785252292473137103));  // W20^2
  a[12] = cmul(a[12], mkc(0.309016994374947451263, -0.951056516295153531182));  // W20^4
  a[13] = cmul(a[13], mkc(-0.309016994374947340241, -0.951056516295153642204));  // W20^6
  a[14] = cmul(a[14], mkc(-0.809016994374947340241, -0.587785252292473248126));  // W20^8
  a[16] = cmul(a[16], mkc(0.587785252292473137103, -0.809016994374947451263));  // W20^3
  a[17] = cmul(a[17], mkc(-0.309016994374947340241, -0.951056516295153642204));  // W20^6
  a[18] = cmul(a[18], mkc(-0.951056516295153531182, -0.309016994374947506774));  // W20^9
  a[19] = cmul(a[19], mkc(-0.809016994374947562285, 0.587785252292473026081));  // W20^12
  // radix 5 over n2 for each k1: X[k1 + 4 k2]
  kcplx r[20];
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    bfly5(a[5 * k1], a[5 * k1 + 1], a[5 * k1 + 2], a[5 * k1 + 3], a[5 * k1 + 4]);
#pragma unroll
    for (int k2 = 0; k2 < 5; ++k2) r[k1 + 4 * k2] = a[5 * k1 + k2];
  }
#pragma unroll
  for (int i = 0; i < 20; ++i) a[i] = r[i];
}

// radix 30 = 3 x 10 (Cooley-Tukey): n = 10 n1 + n2, k = k1 + 3 k2
template <>
__device__ __forceinline__ void bfly<30>(kcplx (&a)[30]) {
  // radix 3 over n1 for each n2: (a[n2], a[n2+10], a[n2+20]) -> A[n2][k1] left in the same slots (k1 = slot / 10)
#pragma unroll
  for (int n2 = 0; n2 < 10; ++n2) {
    kcplx b[3] = {a[n2], a[n2 + 10], a[n2 + 20]};
    bfly<3>(b);
    a[n2] = b[0];
    a[n2 + 10] = b[1];
    a[n2 + 20] = b[2];
  }
  // twiddles W30^(n2 * k1), k1 = 1, 2
  a[11] = cmul(a[11], mkc(0.978147600733805637929, -0.207911690817759337102));  // W30^1
  a[12] = cmul(a[12], mkc(0.913545457642600895502, -0.406736643075800207754));  // W30^2
  a[13] = cmul(a[13], mkc(0.809016994374947424102, -0.587785252292473129169));  // W30^3
  a[14] = cmul(a[14], mkc(0.669130606358858213826, -0.743144825477394235015));  // W30^4
  a[15] = cmul(a[15], mkc(0.5, -0.866025403784438646764));  // W30^5
  a[16] = cmul(a[16], mkc(0.309016994374947424102, -0.951056516295153572116));  // W30^6
  a[17] = cmul(a[17], mkc(0.1045284632676534714, -0.994521895368273336923));  // W30^7
  a[18] = cmul(a[18], mkc(-0.1045284632676534714, -0.994521895368273336923));  // W30^8
  a[19] = cmul(a[19], mkc(-0.309016994374947424102, -0.951056516295153572116));  // W30^9
  a[21] = cmul(a[21], mkc(0.913545457642600895502, -0.406736643075800207754));  // W30^2
  a[22] = cmul(a[22], mkc(0.669130606358858213826, -0.743144825477394235015));  // W30^4
  a[23] = cmul(a[23], mkc(0.309016994374947424102, -0.951056516295153572116));  // W30^6
  a[24] = cmul(a[24], mkc(-0.1045284632676534714, -0.994521895368273336923));  // W30^8
  a[25] = cmul(a[25], mkc(-0.5, -0.866025403784438646764));  // W30^10
  a[26] = cmul(a[26], mkc(-0.809016994374947424102, -0.587785252292473129169));  // W30^12
  a[27] = cmul(a[27], mkc(-0.978147600733805637929, -0.207911690817759337102));  // W30^14
  a[28] = cmul(a[28], mkc(-0.978147600733805637929, 0.207911690817759337102));  // W30^16
  a[29] = cmul(a[29], mkc(-0.809016994374947424102, 0.587785252292473129169));  // W30^18
  // radix 10 over n2 for each k1: X[k1 + 3 k2]
  kcplx r[30];
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1) {
    kcplx b[10];
#pragma unroll
    for (int n2 = 0; n2 < 10; ++n2) b[n2] = a[10 * k1 + n2];
    bfly<10>(b);
#pragma unroll
    for (int k2 = 0; k2 < 10; ++k2) r[k1 + 3 * k2] = b[k2];
  }
#pragma unroll
  for (int i = 0; i < 30; ++i) a[i] = r[i];
}

// LDS index maps: p = position within the line, l = line within the workgroup
template <int N>
struct MapStrided {  // lines fastest (lanes of a wave vary l): conflict-free without padding
  static constexpr int T = Plan<N>::T;
  __device__ __forceinline__ static int at(int p, int l) { return p * T + l; }
  static constexpr int size = N * T;
  static constexpr bool staged_tw = false;  // all lanes of a 16-lane group read the same twiddle (broadcast): the natural table
};
// Position-fastest map of the z kernels (lanes of a wave vary q): index = l * LP + (p ^ ((p >> XS) & XM)) + (PA ? p >> PA : 0),
// LP = N + C + (PA ? N >> PA : 0).  The parameters per length come from tools/lds_conflict_model.py, which applies the
// per-instruction banking of the LDS (ds_write_b128: 8 lanes on 32 banks, ds_read_b128: 4 groups of 16 lanes on 64 banks) to the
// Stockham exchanges of each plan; the default (one pad element per 16 positions) is conflict-free for the 16 x 16 / 16 x 8 / 16 x 2
// plans only.  Measured before (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE): 41-47 % of the LDS cycles of the 512-point z kernels.
// XM < 2^XS and the lengths are multiples of XM + 1: the swizzle permutes positions inside aligned blocks of the line.
template <int N>
struct LineMapParams {
  static constexpr int C = 0, XS = 0, XM = 0, PA = 4;
};
#define MRL_LINEMAP(N_, C_, XS_, XM_, PA_)                      \
  template <>                                                   \
  struct LineMapParams<N_> {                                    \
    static constexpr int C = C_, XS = XS_, XM = XM_, PA = PA_;  \
  };
//           N    C XS XM PA     LDS cycles of the exchanges of one tile: before -> now (ideal)
MRL_LINEMAP(40, 4, 2, 1, 0)      // 2016 -> 1024 (960)
MRL_LINEMAP(48, 4, 2, 3, 0)      // 1344 -> 576 (576)
MRL_LINEMAP(50, 3, 0, 0, 0)      // 1169 -> 760 (480)
MRL_LINEMAP(60, 2, 1, 1, 0)      // 1440 -> 720 (720)
MRL_LINEMAP(64, 0, 2, 3, 4)      // 1280 -> 768 (768)
MRL_LINEMAP(72, 5, 4, 1, 3)      // 1792 -> 1424 (864)
MRL_LINEMAP(80, 8, 3, 1, 0)      // 3424 -> 1696 (1440)
MRL_LINEMAP(90, 1, 0, 0, 5)      // 1938 -> 1053 (720)
MRL_LINEMAP(96, 8, 3, 3, 0)      // 2496 -> 1280 (1152)
MRL_LINEMAP(100, 2, 3, 1, 0)     // 1175 -> 808 (480)
MRL_LINEMAP(120, 4, 2, 1, 0)     // 4320 -> 1472 (1440)
MRL_LINEMAP(144, 4, 3, 3, 0)     // 1212 -> 812 (576)
MRL_LINEMAP(150, 1, 0, 0, 5)     // 3072 -> 1123 (720)
MRL_LINEMAP(160, 8, 3, 3, 0)     // 2176 -> 1024 (960)
MRL_LINEMAP(180, 2, 1, 1, 0)     // 4734 -> 1860 (1440)
MRL_LINEMAP(192, 0, 3, 3, 0)     // 2112 -> 1280 (1152)
MRL_LINEMAP(200, 4, 3, 1, 0)     // 1896 -> 1428 (960)
MRL_LINEMAP(216, 2, 3, 1, 0)     // 3114 -> 2448 (1728)
MRL_LINEMAP(240, 8, 3, 1, 0)     // 6256 -> 2288 (2160)
MRL_LINEMAP(250, 8, 3, 1, 0)     // 2125 -> 1434 (960)
MRL_LINEMAP(270, 3, 0, 0, 5)     // 4563 -> 2243 (1440)
MRL_LINEMAP(288, 8, 3, 3, 0)     // 1632 -> 960 (864)
MRL_LINEMAP(300, 5, 0, 0, 5)     // 2952 -> 1178 (720)
MRL_LINEMAP(320, 0, 3, 3, 0)     // 1856 -> 1024 (960)
MRL_LINEMAP(360, 4, 3, 1, 0)     // 5340 -> 2570 (2160)
MRL_LINEMAP(384, 0, 3, 3, 0)     // 2880 -> 1856 (1728)
MRL_LINEMAP(400, 8, 3, 1, 0)     // 2685 -> 1662 (1440)
MRL_LINEMAP(432, 4, 3, 3, 0)     // 2124 -> 1272 (1152)
MRL_LINEMAP(450, 8, 0, 0, 5)     // 4522 -> 2330 (1440)
MRL_LINEMAP(500, 2, 3, 1, 0)     // 2015 -> 1328 (960)
MRL_LINEMAP(512, 0, 3, 7, 0)     // 2560 -> 1536 (1536)
MRL_LINEMAP(576, 0, 3, 3, 0)     // 1584 -> 960 (864)
MRL_LINEMAP(600, 4, 3, 1, 0)     // 4321 -> 1956 (1440)
MRL_LINEMAP(640, 0, 3, 3, 0)     // 2496 -> 1504 (1440)
MRL_LINEMAP(768, 0, 3, 3, 0)     // 2880 -> 1856 (1728)
MRL_LINEMAP(800, 8, 3, 3, 0)     // 1896 -> 1100 (960)
MRL_LINEMAP(864, 8, 3, 3, 0)     // 2736 -> 1872 (1728)
MRL_LINEMAP(1000, 2, 3, 1, 0)    // 1774 -> 1136 (960)
MRL_LINEMAP(1024, 0, 4, 7, 0)    // 2048 -> 1536 (1536)
MRL_LINEMAP(1152, 0, 3, 3, 0)    // 2160 -> 1392 (1296)
MRL_LINEMAP(1280, 0, 3, 3, 0)    // 2496 -> 1504 (1440)
MRL_LINEMAP(2048, 0, 4, 7, 0)    // 2048 -> 1536 (1536)
MRL_LINEMAP(4096, 0, 4, 7, 0)    // 2048 -> 1536 (1536)
#undef MRL_LINEMAP

template <int N>
struct MapLine {
  using Pm = LineMapParams<N>;
  static_assert(Pm::XM < (1 << Pm::XS) || Pm::XM == 0, "the swizzle may only touch bits below its source bits");
  static_assert(N % (Pm::XM + 1) == 0, "the swizzle permutes inside aligned blocks: the length must be a multiple of the block");
  static constexpr int LP = N + Pm::C + (Pm::PA ? (N >> Pm::PA) : 0);
  // twiddles of a stage laid out [t - 1][k] (k = the lane-dependent index): see stage()
  static constexpr bool staged_tw = true;
  __device__ __forceinline__ static int at(int p, int l) {
    return l * LP + (Pm::XM ? (p ^ ((p >> Pm::XS) & Pm::XM)) : p) + (Pm::PA ? (p >> Pm::PA) : 0);
  }
  static constexpr int size = Plan<N>::T * LP;    // for Plan<N>::T lines
  static constexpr int zsize = ZPlan<N>::T * LP;  // for the z kernels' ZPlan<N>::T lines
};

// (q + off) % NS and (q + off) / NS for q < TPL and an offset that is a compile-time constant after unrolling
template <int NS, int TPL>
__device__ __forceinline__ int mod_ns(int q, int off) {
  const int qn = TPL <= NS ? q : q % NS;
  const int s = qn + off % NS;
  return s >= NS ? s - NS : s;
}
template <int NS, int TPL>
__device__ __forceinline__ int div_ns(int q, int off) {
  const int qn = TPL <= NS ? q : q % NS, qq = TPL <= NS ? 0 : q / NS;
  return off / NS + qq + (qn + off % NS >= NS ? 1 : 0);
}

// radix-R stage on the P register values (v[i + S*t] = element t of butterfly i, S = P/R butterflies per thread)
// ST (staged twiddle table, the z kernels): W holds, stage after stage, the factors [t - 1][k] = w_N^(t k N / (Ns R)) with k fastest
// -- the lanes of a wave differ in k, so a plain W[t k N / (Ns R)] is a strided LDS read (8-way conflicts in the 512-point plan:
// 1664 LDS cycles per tile where 448 suffice); the table of stage Ns starts at Ns - r0 (the stages before it hold that many entries).
template <int N, int R, int NS, bool ST = false>
__device__ __forceinline__ void stage(kcplx (&v)[Plan<N>::P], int q, const kcplx *W) {
  constexpr int P = Plan<N>::P, S = P / R, TPL = N / P;
  static_assert(P % R == 0, "radix must divide the points per thread");
#pragma unroll
  for (int i = 0; i < S; ++i) {
    kcplx a[R];
#pragma unroll
    for (int t = 0; t < R; ++t) a[t] = v[i + S * t];
    if (NS > 1) {
      // k = (q + i TPL) % NS without a division: i TPL splits into a compile-time multiple of NS and a compile-time remainder.
      // (The plain `b % NS` form is also MISCOMPILED by hipcc 7.2 for N = 240, NS = 30, TPL = 8: the 16-bit multiply-shift it
      // emits for the modulo ends with the butterflies i = 8 and i = 12 reading the imaginary part of the staged twiddle of
      // another butterfly -- every 240-point z transform was wrong until round 5 compared one with the oracle.)
      const int k = mod_ns<NS, TPL>(q, i * TPL);
      if (ST) {
        const kcplx *Ws = W + (NS - Plan<N>::r0) + k;
#pragma unroll
        for (int t = 1; t < R; ++t) a[t] = cmul(a[t], Ws[(t - 1) * NS]);
      } else {
        const int step = k * (N / (NS * R));
#pragma unroll
        for (int t = 1; t < R; ++t) a[t] = cmul(a[t], W[t * step]);
      }
    }
    bfly<R>(a);
#pragma unroll
    for (int t = 0; t < R; ++t) v[i + S * t] = a[t];
  }
}

// write the outputs of a radix-R stage (Ns = NS) in Stockham order, then re-own q + m*TPL
template <int N, int R, int NS, class Map>
__device__ __forceinline__ void exchange(kcplx (&v)[Plan<N>::P], int q, int l, kcplx *X) {
  constexpr int P = Plan<N>::P, S = P / R, TPL = N / P;
  __syncthreads();  // previous readers of X are done
#pragma unroll
  for (int i = 0; i < S; ++i) {
    const int p0 = div_ns<NS, TPL>(q, i * TPL) * NS * R + mod_ns<NS, TPL>(q, i * TPL);
#pragma unroll
    for (int t = 0; t < R; ++t) X[Map::at(p0 + t * NS, l)] = v[i + S * t];
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < P; ++m) v[m] = X[Map::at(q + m * TPL, l)];
}

// full forward transform of the line owned by (q, l); v in: x[q + m*TPL], out: X[q + m*TPL]
template <int N, class Map>
__device__ __forceinline__ void fft_line(kcplx (&v)[Plan<N>::P], int q, int l, kcplx *X, const kcplx *W) {
  using Pl = Plan<N>;
  constexpr bool ST = Map::staged_tw;
  stage<N, Pl::r0, 1, ST>(v, q, W);
  exchange<N, Pl::r0, 1, Map>(v, q, l, X);
  stage<N, Pl::r1, Pl::r0, ST>(v, q, W);
  if constexpr (Pl::ns >= 3) {
    exchange<N, Pl::r1, Pl::r0, Map>(v, q, l, X);
    stage<N, Pl::r2, Pl::r0 * Pl::r1, ST>(v, q, W);
  }
  if constexpr (Pl::ns >= 4) {
    exchange<N, Pl::r2, Pl::r0 * Pl::r1, Map>(v, q, l, X);
    stage<N, Pl::r3, Pl::r0 * Pl::r1 * Pl::r2, ST>(v, q, W);
  }
}

// bijective XCD-aware remap: hardware deals block b to XCD b % 8; give each XCD a contiguous
// range of logical tiles so neighbouring tiles (which share partial 128-B lines) share an L2.
// (Round 5 A/B against the identity map -- consecutive tiles round-robin over the XCDs, the shape that wins in a plain mover:
// serial y passes 92 -> 122 us at 256^3 and 860 -> 1465 us at 512^3, 256^3 substep 0.305 -> 0.368 ms; the slab-local kernels,
// whose pieces are whole lines, do not care (+-1 %): profiles/r05_ab_xcd_remap_vs_identity.txt.)
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nb) {
  const unsigned q8 = nb >> 3, r8 = nb & 7, xcd = b & 7, idx = b >> 3;
  return (xcd < r8) ? xcd * (q8 + 1) + idx : r8 * (q8 + 1) + (xcd - r8) * q8 + idx;
}

// the same ranges walked in descending order: a consumer kernel that starts where its producer ended finds the
// producer's last ~256 MB still in the Infinity Cache
__device__ __forceinline__ unsigned xcd_remap_rev(unsigned b, unsigned nb) {
  const unsigned q8 = nb >> 3, r8 = nb & 7, xcd = b & 7, idx = b >> 3;
  const unsigned len = q8 + (xcd < r8 ? 1u : 0u);
  const unsigned start = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  return start + (len - 1 - idx);
}

// Twiddle staging split in two so that a kernel can issue the table loads BEFORE its operand loads (vmcnt
// retires in order: the small L2-resident table loads must not queue behind a full HBM round trip) and
// write them to LDS AFTER the operand loads are in flight.
template <int N, int NTH = Plan<N>::NT>
struct TwRegs {
  static constexpr int NT = NTH, CNT = (N + NT - 1) / NT;
  kcplx v[CNT];
};
template <int N, int NTH>
__device__ __forceinline__ void tw_issue(TwRegs<N, NTH> &r, const kcplx *__restrict__ tw) {
#pragma unroll
  for (int j = 0; j < TwRegs<N, NTH>::CNT; ++j) {
    const int idx = threadIdx.x + j * NTH;
    r.v[j] = idx < N ? tw[idx] : mkc(0.0, 0.0);
  }
}
// the staged table of the z kernels (stage(): ST): entry s of stage Ns is w_N^(t k N / (Ns R)), t = s / Ns + 1, k = s % Ns, gathered
// from the natural table (L1 / L2 resident); N - r0 entries in all
template <int N>
__device__ __forceinline__ int tw_staged_src(int s) {
  using Pl = Plan<N>;
  constexpr int n1 = Pl::r0, c1 = (Pl::r1 - 1) * n1;        // first twiddled stage: Ns = r0, radix r1
  constexpr int n2 = n1 * Pl::r1, c2 = (Pl::r2 - 1) * n2;   // second
  constexpr int n3 = n2 * Pl::r2;                            // third
  if (s < c1) return (s / n1 + 1) * (s % n1) * (N / (n1 * Pl::r1));
  s -= c1;
  if (c2 > 0 && s < c2) return (s / n2 + 1) * (s % n2) * (N / (n2 * Pl::r2));
  s -= c2;
  return (s / n3 + 1) * (s % n3) * (N / (n3 * Pl::r3));
}
template <int N, int NTH>
__device__ __forceinline__ void tw_issue_staged(TwRegs<N, NTH> &r, const kcplx *__restrict__ tw) {
#pragma unroll
  for (int j = 0; j < TwRegs<N, NTH>::CNT; ++j) {
    const int idx = threadIdx.x + j * NTH;
    r.v[j] = idx < N - Plan<N>::r0 ? tw[tw_staged_src<N>(idx)] : mkc(0.0, 0.0);
  }
}
template <int N, int NTH>
__device__ __forceinline__ void tw_commit(const TwRegs<N, NTH> &r, kcplx *W) {
#pragma unroll
  for (int j = 0; j < TwRegs<N, NTH>::CNT; ++j) {
    const int idx = threadIdx.x + j * NTH;
    if (idx < N) W[idx] = r.v[j];
  }
}

}  // namespace MRL_P2NS
}  // namespace mrl
