// The contiguous-axis (z) kernels of the two-stage plans (fft_two.h): real lines <-> half spectra for 120 / 150 / 160 / 180 / 240 points
// with at most 16 points per thread, one LDS exchange per transform and the twiddles of the second stage staged [t][q] so that the
// lanes of a line read consecutive table entries.  Same roles as k_z_fwd / k_z_inv / k_z_inv_fwd of fft_pow2_kernels.h (two real
// lines per complex transform, DomainAction::fft / ifft along the last axis, DomainAction.C:268-296) and the same arithmetic around
// the transforms (mu_eval, the k <-> N - k separation), so the fields agree with the uniform 30- / 20-point plans to the rounding of
// the butterflies.
//   pattern A (real side)      thread q < R1 holds positions q + R1 t, t < R0
//   pattern B (spectral side)  thread q < R0 holds bins q + R0 t', t' < R1
// 240 = 16 x 15 and 160 = 16 x 10 put the 16 on the spectral side (every lane of a 16-lane group owns bins); the first-stage
// writes of a radix-16 stage are 256 bytes apart and take one pad element per 16 positions.
#pragma once
#include "fft_pow2_launch.h"

namespace mrl {
namespace MRL_P2NS {

// the lengths with a two-stage plan (fft_two.h has the strided passes for the same list)
constexpr bool two_stage_z_len(long long n) { return n == 120 || n == 150 || n == 160 || n == 180 || n == 240 || n == 300 || n == 320; }

template <int N>
struct ZPlan2;
#define MRL_ZPLAN2(N_, R0_, R1_, LPB_)                                                             \
  template <>                                                                                      \
  struct ZPlan2<N_> {                                                                              \
    static constexpr int R0 = R0_, R1 = R1_, PM = (R0_ > R1_ ? R0_ : R1_), TPL = PM, LPB = LPB_;  \
    static constexpr int NT = LPB_ * TPL, LP = N_ + N_ / 16 + 1;                                   \
    static constexpr int NTABF = (R1_ - 1) * R0_, NTABI = (R0_ - 1) * R1_;                         \
    static_assert(R0_ * R1_ == N_ && NT <= 256 && NTABF <= N_ && NTABI <= N_, "bad plan");        \
  };
// (lines per workgroup: 4 / 8 / 16 measured level on 150 / 160 / 180-point lines, round 5 -- those grids are short launches of 1-2 k
// workgroups, bound by their first and last generation rather than by the tile shape)
MRL_ZPLAN2(120, 10, 12, 16)
MRL_ZPLAN2(150, 10, 15, 8)
MRL_ZPLAN2(160, 16, 10, 8)
MRL_ZPLAN2(180, 12, 15, 8)
MRL_ZPLAN2(240, 16, 15, 8)
MRL_ZPLAN2(192, 16, 12, 8)   // (the fused inverse + forward z pass of the fused family's 192-point grids, ch_fused.hip)
// 20 points per thread on the real side, 20 threads per line (fft_two.h); 12 lines per workgroup = 240 threads (8 lines = 2.5 waves:
// fused z pass 177 -> 156 us at 300^3; 320 with the 20 on the spectral side instead: 208 -> 175 us at 320^3); 16 lines at 240 points: level
MRL_ZPLAN2(300, 20, 15, 12)
// (200 = 20 x 10 was measured for the fused inverse + forward z pass of the fused family: 55.7-56.3 us against 44.4-45.2 us of
// k_z_inv_fwd<200> at 200^3 -- ten of twenty lanes carry the whole real-side work; not instantiated)
MRL_ZPLAN2(320, 20, 16, 12)
#undef MRL_ZPLAN2

template <int N>
constexpr size_t lds_two_z(int ntab) {
  return sizeof(kcplx) * ((size_t)ntab * N + (size_t)ZPlan2<N>::LPB * ZPlan2<N>::LP);
}

// staged twiddles of the second stage of a transform with first radix RA: entry (t - 1) RA + q = w_N^(t q), t = 1 .. RB - 1, q < RA
template <int N, int RA, int RB, int NT>
struct Tw2Regs {
  static constexpr int CNT = ((RB - 1) * RA + NT - 1) / NT;
  kcplx v[CNT];
};
template <int N, int RA, int RB, int NT>
__device__ __forceinline__ void tw2_issue(Tw2Regs<N, RA, RB, NT> &r, const kcplx *__restrict__ tw) {
#pragma unroll
  for (int j = 0; j < Tw2Regs<N, RA, RB, NT>::CNT; ++j) {
    const int s = threadIdx.x + j * NT;
    r.v[j] = s < (RB - 1) * RA ? tw[(s / RA + 1) * (s % RA)] : mkc(0.0, 0.0);
  }
}
template <int N, int RA, int RB, int NT>
__device__ __forceinline__ void tw2_commit(const Tw2Regs<N, RA, RB, NT> &r, kcplx *W) {
#pragma unroll
  for (int j = 0; j < Tw2Regs<N, RA, RB, NT>::CNT; ++j) {
    const int s = threadIdx.x + j * NT;
    if (s < (RB - 1) * RA) W[s] = r.v[j];
  }
}

// forward DFT of a line held position-fastest in LDS: in v[t] = x[q + RB t] (threads q < RB), out v[t'] = X[q + RA t'] (threads q < RA)
template <int N, int RA, int RB>
__device__ __forceinline__ void fft2z(kcplx (&v)[ZPlan2<N>::PM], int q, int l, kcplx *X, const kcplx *Wt) {
  constexpr int TPL = ZPlan2<N>::TPL, LP = ZPlan2<N>::LP;
  constexpr bool PAD = RA == 16;
  kcplx *Xl = X + l * LP;
  {
    kcplx a[RA];
#pragma unroll
    for (int t = 0; t < RA; ++t) a[t] = v[t];
    bfly<RA>(a);
    __syncthreads();  // previous readers of X are done
    if (RB == TPL || q < RB) {
#pragma unroll
      for (int t = 0; t < RA; ++t) {
        const int p = RA * q + t;
        Xl[PAD ? p + (p >> 4) : p] = a[t];
      }
    }
  }
  __syncthreads();
  {
    const int qc = (RA == TPL || q < RA) ? q : RA - 1;
    kcplx b[RB];
#pragma unroll
    for (int t = 0; t < RB; ++t) {
      const int p = qc + RA * t;
      b[t] = Xl[PAD ? p + (p >> 4) : p];
    }
#pragma unroll
    for (int t = 1; t < RB; ++t) b[t] = cmul(b[t], Wt[(t - 1) * RA + qc]);
    bfly<RB>(b);
#pragma unroll
    for (int t = 0; t < RB; ++t) v[t] = b[t];
  }
}

// v (pattern B) = transform of the packed line a + i b: half spectra of a (o0) and b (o1), bins 0 .. N/2.  The k <-> N - k pairing:
// with 16 threads on the spectral side (160, 240) bin N - k of the line sits in lane (16 - q) % 16 of the same 16-lane group, register
// R1 - 1 - t' (lane 0: its own register (R1 - t') % R1) and comes over with ds_bpermute (store_half_spectra of fft_pow2_kernels.h);
// the other lengths go through a natural-order copy of the line in LDS.  (Measured level with the LDS form at 160 / 240 points: these
// kernels are not bound by their LDS traffic.)  Every thread of the workgroup must call it.
template <int N>
__device__ __forceinline__ void store_half_spectra2(const kcplx (&v)[ZPlan2<N>::PM], int q, int l, kcplx *X, bool valid, kcplx *o0, kcplx *o1) {
  using Pl = ZPlan2<N>;
  constexpr int R0 = Pl::R0, R1 = Pl::R1;
  if constexpr (R0 == 16 && Pl::TPL == 16) {
    const int partner = (int)(threadIdx.x & 63u) - q + ((16 - q) & 15);
#pragma unroll
    for (int t = 0; t <= (N / 2) / R0; ++t) {   // bins beyond N/2 are somebody's mirror image only
      const int k = q + R0 * t;
      const kcplx s = lane_get(v[R1 - 1 - t], partner);
      const kcplx own = v[(R1 - t) % R1];
      const kcplx xk = v[t];
      const kcplx xn = mkc(q == 0 ? own.x : s.x, q == 0 ? own.y : s.y);
      if (valid && k <= N / 2) {
        o0[k] = mkc(kreal(0.5) * (xk.x + xn.x), kreal(0.5) * (xk.y - xn.y));
        o1[k] = mkc(kreal(0.5) * (xk.y + xn.y), kreal(-0.5) * (xk.x - xn.x));
      }
    }
  } else {
    kcplx *Xl = X + l * Pl::LP;
    const bool own = R0 == Pl::TPL || q < R0;
    __syncthreads();
    if (own) {
#pragma unroll
      for (int t = 0; t < R1; ++t) Xl[q + R0 * t] = v[t];
    }
    __syncthreads();
    if (!valid || !own) return;
#pragma unroll
    for (int t = 0; t < R1; ++t) {
      const int k = q + R0 * t;
      if (k <= N / 2) {
        const kcplx xk = v[t];
        const kcplx xn = Xl[k == 0 ? 0 : N - k];
        o0[k] = mkc(kreal(0.5) * (xk.x + xn.x), kreal(0.5) * (xk.y - xn.y));
        o1[k] = mkc(kreal(0.5) * (xk.y + xn.y), kreal(-0.5) * (xk.x - xn.x));
      }
    }
  }
}

// half spectra A, B of two real lines -> v (pattern B) = swap(X), X[p] = A[p] + i B[p] for p <= N/2 and conj(A[N-p]) + i conj(B[N-p])
// beyond (load_half_spectra of fft_pow2_kernels.h, its LDS-free form)
template <int N>
__device__ __forceinline__ void load_half_spectra2(kcplx (&v)[ZPlan2<N>::PM], int q, const kcplx *A, const kcplx *B) {
  using Pl = ZPlan2<N>;
  constexpr int R0 = Pl::R0, R1 = Pl::R1;
  const int qc = (R0 == Pl::TPL || q < R0) ? q : R0 - 1;
  kcplx av[R1], bv[R1];
#pragma unroll
  for (int t = 0; t < R1; ++t) {
    const int p = qc + R0 * t;
    const int k = (p <= N / 2) ? p : N - p;
    av[t] = A[k];
    bv[t] = B[k];
  }
#pragma unroll
  for (int t = 0; t < R1; ++t) {
    const int p = qc + R0 * t;
    const bool lo = p <= N / 2;
    const int k = lo ? p : N - p;
    kcplx a = av[t], b = bv[t];
    if (k == 0 || k == N / 2) {  // c2r ignores the imaginary part of the self-conjugate bins
      a.y = 0.0;
      b.y = 0.0;
    }
    const kcplx x = lo ? mkc(a.x - b.y, a.y + b.x) : mkc(a.x + b.y, b.x - a.y);
    v[t] = cswap(x);
  }
}

// ---- z forward.  MODE 0 (PAIR): rows 2L, 2L+1 of `in` -> rows 2L, 2L+1 of out0.  MODE 1 (CH): row L of `in` (= c) -> row L of out0
//      (c-hat_z) and out1 (mu-hat_z), mu = f'(c) optionally written to mu_out.  nlines = number of complex transforms.
template <int N, int MODE, int FAM>
__global__ void __launch_bounds__(ZPlan2<N>::NT, 2) k_z_fwd2(const kreal *__restrict__ in, kcplx *__restrict__ out0, kcplx *__restrict__ out1,
                                                              kreal *__restrict__ mu_out, ChDev chp, long long nlines,
                                                              const kcplx *__restrict__ tw, ZLay zl) {
  using Pl = ZPlan2<N>;
  constexpr int R0 = Pl::R0, R1 = Pl::R1, TPL = Pl::TPL, LPB = Pl::LPB, NZC = N / 2 + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  kcplx *W = reinterpret_cast<kcplx *>(smem);
  kcplx *X = W + N;
  const int q = threadIdx.x % TPL, l = threadIdx.x / TPL;
  const long long L = (long long)xcd_remap(blockIdx.x, gridDim.x) * LPB + l;
  const bool valid = L < nlines;
  Tw2Regs<N, R0, R1, Pl::NT> twr;
  tw2_issue(twr, tw);
  const long long Lc = valid ? L : 0;  // out-of-range lines transform line 0 again and store nothing
  const long long r0 = (MODE == 1) ? Lc : 2 * Lc;
  const bool ownA = R1 == TPL || q < R1;
  const int qa = ownA ? q : R1 - 1;
  kcplx v[Pl::PM];
  {
    const kreal *p0 = in + r0 * N + qa;
    kreal a[R0], b[R0];
#pragma unroll
    for (int t = 0; t < R0; ++t) a[t] = p0[t * R1];
    if (MODE != 1) {
#pragma unroll
      for (int t = 0; t < R0; ++t) b[t] = p0[N + t * R1];
    }
    tw2_commit(twr, W);
    if (MODE == 1) {
#pragma unroll
      for (int t = 0; t < R0; ++t) b[t] = mu_eval<FAM>(chp, a[t]);
    }
#pragma unroll
    for (int t = 0; t < R0; ++t) v[t] = mkc(a[t], b[t]);
    if (MODE == 1 && mu_out && valid && ownA) {
      kreal *pm = mu_out + r0 * N + q;
#pragma unroll
      for (int t = 0; t < R0; ++t) pm[t * R1] = v[t].y;
    }
  }
  fft2z<N, R0, R1>(v, q, l, X, W);
  kcplx *o0 = (MODE != 1) ? out0 + zrow(2 * Lc, NZC, zl) : out0 + zrow(Lc, NZC, zl);
  kcplx *o1 = (MODE != 1) ? out0 + zrow(2 * Lc + 1, NZC, zl) : out1 + zrow(Lc, NZC, zl);
  store_half_spectra2<N>(v, q, l, X, valid, o0, o1);
}

// ---- z inverse (PAIR): rows 2L, 2L+1 of the half spectrum `in` -> real rows 2L, 2L+1 of out, * scale
template <int N>
__global__ void __launch_bounds__(ZPlan2<N>::NT, 2) k_z_inv2(const kcplx *__restrict__ in, kreal *__restrict__ out, kreal scale,
                                                              long long nlines, const kcplx *__restrict__ tw, ZLay zl) {
  using Pl = ZPlan2<N>;
  constexpr int R0 = Pl::R0, R1 = Pl::R1, TPL = Pl::TPL, LPB = Pl::LPB, NZC = N / 2 + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  kcplx *W = reinterpret_cast<kcplx *>(smem);
  kcplx *X = W + N;
  const int q = threadIdx.x % TPL, l = threadIdx.x / TPL;
  const long long L = (long long)xcd_remap(blockIdx.x, gridDim.x) * LPB + l;
  const bool valid = L < nlines;
  Tw2Regs<N, R1, R0, Pl::NT> twr;
  tw2_issue(twr, tw);
  const long long Lc = valid ? L : 0;
  kcplx v[Pl::PM];
  load_half_spectra2<N>(v, q, in + zrow(2 * Lc, NZC, zl), in + zrow(2 * Lc + 1, NZC, zl));
  tw2_commit(twr, W);
  fft2z<N, R1, R0>(v, q, l, X, W);
  if (valid && (R1 == TPL || q < R1)) {
    kreal *o0 = out + (2 * L) * N + q;
#pragma unroll
    for (int t = 0; t < R0; ++t) {
      // swap back: real part (row 2L) = v.y, imaginary part (row 2L+1) = v.x
      o0[t * R1] = v[t].y * scale;
      o0[N + t * R1] = v[t].x * scale;
    }
  }
}

// a product that is never contracted into a following add: the scaled real values below feed the first butterfly of the forward
// transform, and whether hipcc fuses `v * scale + w` there may differ between two compilations of the kernel (the run-time compiled
// instance with a generated chemical potential must agree bit for bit with the built-in families, and the fused kernel with the
// two separate ones, which store the product)
__device__ __forceinline__ kreal mul_exact(kreal a, kreal b) {
#pragma clang fp contract(off)
  return a * b;
}

// ---- z inverse of substep n fused with the z forward (CH mode) of substep n + 1 (k_z_inv_fwd of fft_pow2_kernels.h): rows 2L, 2L+1 of
//      `in` -> the two real lines c = irfft(.) * scale stay in registers -> mu = f'(c) -> each line packed as c + i mu and transformed
//      forward -> rows 2L, 2L+1 of out0 (c-hat_z) and out1 (mu-hat_z).  in == out0 is allowed.  Same arithmetic, in the same order, as
//      k_z_inv2 followed by k_z_fwd2<CH>: bit-identical fields.
template <int N, int FAM>
__global__ void __launch_bounds__(ZPlan2<N>::NT, 2) k_z_inv_fwd2(const kcplx *__restrict__ in, kcplx *__restrict__ out0, kcplx *__restrict__ out1,
                                                                  kreal *__restrict__ mu_out, ChDev chp, kreal scale, long long nlines,
                                                                  const kcplx *__restrict__ tw, ZLay zl) {
  using Pl = ZPlan2<N>;
  constexpr int R0 = Pl::R0, R1 = Pl::R1, TPL = Pl::TPL, LPB = Pl::LPB, NZC = N / 2 + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  kcplx *Wi = reinterpret_cast<kcplx *>(smem);   // staged twiddles of the inverse direction (first radix R1)
  kcplx *Wf = Wi + N;                            // ... of the forward direction (first radix R0)
  kcplx *X = Wf + N;
  const int q = threadIdx.x % TPL, l = threadIdx.x / TPL;
  const long long L = (long long)xcd_remap(blockIdx.x, gridDim.x) * LPB + l;
  const bool valid = L < nlines;
  Tw2Regs<N, R1, R0, Pl::NT> twi;
  Tw2Regs<N, R0, R1, Pl::NT> twf;
  tw2_issue(twi, tw);
  tw2_issue(twf, tw);
  const long long Lc = valid ? L : 0;  // out-of-range lines transform line pair 0 again and store nothing
  kcplx v[Pl::PM];
  load_half_spectra2<N>(v, q, in + zrow(2 * Lc, NZC, zl), in + zrow(2 * Lc + 1, NZC, zl));
  tw2_commit(twi, Wi);
  tw2_commit(twf, Wf);
  fft2z<N, R1, R0>(v, q, l, X, Wi);
  const bool ownA = R1 == TPL || q < R1;
  kreal cb[R0];  // second line (row 2L+1), kept while the first one is transformed
#pragma unroll
  for (int t = 0; t < R0; ++t) {
    const kreal ca = mul_exact(v[t].y, scale);
    cb[t] = mul_exact(v[t].x, scale);
    v[t] = mkc(ca, mu_eval<FAM>(chp, ca));
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half == 1) {
#pragma unroll
      for (int t = 0; t < R0; ++t) v[t] = mkc(cb[t], mu_eval<FAM>(chp, cb[t]));
    }
    if (mu_out && valid && ownA) {
      kreal *pm = mu_out + (2 * L + half) * N + q;
#pragma unroll
      for (int t = 0; t < R0; ++t) pm[t * R1] = v[t].y;
    }
    fft2z<N, R0, R1>(v, q, l, X, Wf);
    store_half_spectra2<N>(v, q, l, X, valid, out0 + zrow(2 * Lc + half, NZC, zl), out1 + zrow(2 * Lc + half, NZC, zl));
  }
}

#ifndef __HIPCC_RTC__   // (the kernels above are also compiled at run time with a generated chemical potential: expr.hip)
template <int N, int MODE, int FAM>
inline int launch_z_fwd2(mrl_ctx *ctx, const kreal *in, kcplx *o0, kcplx *o1, kreal *mu, const ChDev &chp, long long nlines,
                         ZLay zl = ZLay{0u, 0u}) {
  static std::atomic<bool> attr{false};
  constexpr size_t lds = lds_two_z<N>(1);
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY((set_lds_attr(ctx, k_z_fwd2<N, MODE, FAM>, lds)));
    attr.store(true, std::memory_order_release);
  }
  constexpr int LPB = ZPlan2<N>::LPB;
  const long long nb = (nlines + LPB - 1) / LPB;
  hipLaunchKernelGGL((k_z_fwd2<N, MODE, FAM>), dim3((unsigned)nb), dim3(ZPlan2<N>::NT), lds, ctx->stream, in, o0, o1, mu, chp, nlines, tw_table(ctx, 2), zl);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}
template <int N>
inline int launch_z_inv2(mrl_ctx *ctx, const kcplx *in, kreal *out, kreal scale, long long nlines, ZLay zl = ZLay{0u, 0u}) {
  static std::atomic<bool> attr{false};
  constexpr size_t lds = lds_two_z<N>(1);
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY((set_lds_attr(ctx, k_z_inv2<N>, lds)));
    attr.store(true, std::memory_order_release);
  }
  constexpr int LPB = ZPlan2<N>::LPB;
  const long long nb = (nlines + LPB - 1) / LPB;
  hipLaunchKernelGGL((k_z_inv2<N>), dim3((unsigned)nb), dim3(ZPlan2<N>::NT), lds, ctx->stream, in, out, scale, nlines, tw_table(ctx, 2), zl);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}
// nlines = number of line PAIRS
template <int N, int FAM>
inline int launch_z_inv_fwd2(mrl_ctx *ctx, const kcplx *in, kcplx *o0, kcplx *o1, kreal *mu, const ChDev &chp, kreal scale, long long nlines,
                             ZLay zl = ZLay{0u, 0u}) {
  static std::atomic<bool> attr{false};
  constexpr size_t lds = lds_two_z<N>(2);
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY((set_lds_attr(ctx, k_z_inv_fwd2<N, FAM>, lds)));
    attr.store(true, std::memory_order_release);
  }
  constexpr int LPB = ZPlan2<N>::LPB;
  const long long nb = (nlines + LPB - 1) / LPB;
  hipLaunchKernelGGL((k_z_inv_fwd2<N, FAM>), dim3((unsigned)nb), dim3(ZPlan2<N>::NT), lds, ctx->stream, in, o0, o1, mu, chp, scale, nlines,
                     tw_table(ctx, 2), zl);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

#endif  // __HIPCC_RTC__

}  // namespace MRL_P2NS
}  // namespace mrl
