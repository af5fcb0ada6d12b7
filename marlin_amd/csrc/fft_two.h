// Two-stage plans with per-stage ownership for the strided passes of the lengths 2^a 3^b 5^c that the uniform plans of fft_pow2.h
// can only serve with 30 (or 20) points per thread: 120 = 10 x 12, 150 = 10 x 15, 160 = 10 x 16, 180 = 12 x 15, 240 = 15 x 16, and with
// 20 points per thread 300 = 20 x 15, 320 = 20 x 16, 400 = 20 x 20 (the fused x pass only).
//
// A uniform plan (Plan<N>) gives every thread the same P points q + m TPL through all stages, so a length whose factors do not share
// a common P ends at P = 30: 120 vector registers per array, one field per launch, no room for the fused Cahn-Hilliard x pass, and
// three LDS exchanges (240 = 30 x 2 x 2 x 2).  Here a line of N = R0 R1 points is owned by max(R0, R1) threads that change what they
// hold between the two stages (one LDS exchange):
//   pattern A   thread q < R1 holds x[q + R1 t], t < R0           -> stage 1: one radix-R0 butterfly per thread
//   pattern B   thread q < R0 holds X[q + R0 t'], t' < R1         <- stage 2: twiddle w_N^(t' q), one radix-R1 butterfly per thread
// (Stockham: stage 1 writes its outputs at R0 q + t, stage 2 reads q + R0 t' and leaves X in natural order.)  The same routine with
// the radices swapped maps pattern B to pattern A -- the inverse passes (conjugation by swapping re / im, as everywhere here) -- so a
// forward + inverse pair like the fused x pass loads and stores in A and works on the spectrum in B.  At most 16 points per thread:
// 64 registers per array, the budget of the 256-point kernels.  Threads beyond a pattern's count load clamped duplicates and store
// nothing.  Measured on 240^3, same box (profiles/r05_ab_two_stage_plans_vs_uniform.txt): forward y of both fields 101.8 -> 71.6 us
// (4.4 -> 6.2 TB/s), x update 51.8 + 102.5 (two kernels) -> 92.9 us (4.3 -> 6.0 TB/s), inverse y 52.5 -> 40.1 us.
#pragma once
#include <atomic>
#include "ch_fused_body.h"
#include "fft_pow2_launch.h"

namespace mrl {
namespace MRL_P2NS {

template <int N>
struct Plan2 {
  static constexpr bool ok = false;
};
#define MRL_PLAN2(N_, R0_, R1_, T_)                                                    \
  template <>                                                                          \
  struct Plan2<N_> {                                                                   \
    static constexpr bool ok = true;                                                   \
    static constexpr int R0 = R0_, R1 = R1_, PM = (R0_ > R1_ ? R0_ : R1_), TPL = PM;   \
    static constexpr int T = T_, NT = T * TPL;   /* T adjacent lines per tile */       \
    static_assert(R0_ * R1_ == N_ && NT <= 256, "bad two-stage plan");                 \
  };
// (R1 >= R0 for the 16-point plans: pattern A, which carries three of the five streams of the fused x pass, is the one with every
// thread busy; the 20-point plans put the 20 on the A side -- 80 registers while one array is loaded and transformed, 60-64 per array
// in the update, where three arrays are live: no spills, against 28-270 spilled registers the other way round)
// 16 lines per tile = 256-byte pieces
MRL_PLAN2(120, 10, 12, 16)
MRL_PLAN2(150, 10, 15, 16)
MRL_PLAN2(160, 10, 16, 16)
MRL_PLAN2(180, 12, 15, 16)
MRL_PLAN2(240, 15, 16, 16)
// 20 points per thread (80 registers per array): both fields of a y pass still fit one launch; the fused x pass keeps TWO arrays and
// re-reads the Nhat it has just stored (k_ch_xfused2<REREAD>).  300 = 20 x 15, 320 = 20 x 16 replace 30- / 20-point uniform plans,
// 400 = 20 x 20 the four-stage 10-point plan of the fused family (10 x 10 x 2 x 2, 40 threads per line, tiles of 6 lines = 96-byte
// pieces: its fused x pass ran at 3.4 TB/s).  Tiles of 12 lines (192-byte pieces) where the LDS tile allows it, 8 at 400 points.
MRL_PLAN2(300, 20, 15, 12)
MRL_PLAN2(320, 20, 16, 12)
MRL_PLAN2(400, 20, 20, 8)
MRL_PLAN2(192, 12, 16, 16)   // (fused x pass of the fused family's 192-point grids: 52.5 -> 50.0 us at 192^3)
// (200 = 20 x 10 was measured too: its fused x pass 67.4-68.3 us against 62.5-63.6 us of the uniform 10-point plan at 200^3 -- ten of
// twenty lanes idle on the pattern-A side, which carries three of the five streams; not instantiated)
#undef MRL_PLAN2

// lengths whose strided passes and z kernels run two-stage plans ...
constexpr bool two_stage_len(long long n) { return n == 120 || n == 150 || n == 160 || n == 180 || n == 240 || n == 300 || n == 320; }
// ... and whose fused x pass does: 400 = 20 x 20 keeps the uniform 10-point plan of the fused family for its y and z passes (measured
// faster there: 402 / 354 us against 476 / 381 us at 400^3) and takes only this kernel (735 -> 547 us)
constexpr bool two_stage_x_len(long long n) { return two_stage_len(n) || n == 400; }   // (192: ch_fused.hip only)

// what a thread holds in a pattern: CNT points q + STRIDE t, and whether thread q holds anything
template <int N, bool PATTERN_B>
struct Own2 {
  static constexpr int CNT = PATTERN_B ? Plan2<N>::R1 : Plan2<N>::R0;
  static constexpr int STRIDE = PATTERN_B ? Plan2<N>::R0 : Plan2<N>::R1;   // = number of threads that hold something
  static constexpr bool all = STRIDE == Plan2<N>::TPL;
  __device__ __forceinline__ static bool active(int q) { return all || q < STRIDE; }
  __device__ __forceinline__ static int clamp(int q) { return all ? q : (q < STRIDE ? q : STRIDE - 1); }
};

// forward DFT of a line: in v[t] = x[q + RB t] (t < RA, threads q < RB), out v[t'] = X[q + RA t'] (t' < RB, threads q < RA)
template <int N, int RA, int RB>
__device__ __forceinline__ void fft2(kcplx (&v)[Plan2<N>::PM], int q, int l, kcplx *X, const kcplx *W) {
  static_assert(RA * RB == N, "radices");
  constexpr int T = Plan2<N>::T, TPL = Plan2<N>::TPL;
  {
    kcplx a[RA];
#pragma unroll
    for (int t = 0; t < RA; ++t) a[t] = v[t];
    bfly<RA>(a);
    __syncthreads();  // previous readers of X are done
    if (RB == TPL || q < RB) {
#pragma unroll
      for (int t = 0; t < RA; ++t) X[(RA * q + t) * T + l] = a[t];
    }
  }
  __syncthreads();
  {
    const int qc = (RA == TPL || q < RA) ? q : RA - 1;  // (threads without a butterfly read valid elements and are never stored)
    kcplx b[RB];
#pragma unroll
    for (int t = 0; t < RB; ++t) b[t] = X[(qc + RA * t) * T + l];
#pragma unroll
    for (int t = 1; t < RB; ++t) b[t] = cmul(b[t], W[t * qc]);
    bfly<RB>(b);
#pragma unroll
    for (int t = 0; t < RB; ++t) v[t] = b[t];
  }
}
template <int N>
__device__ __forceinline__ void fft2_a_to_b(kcplx (&v)[Plan2<N>::PM], int q, int l, kcplx *X, const kcplx *W) {
  fft2<N, Plan2<N>::R0, Plan2<N>::R1>(v, q, l, X, W);
}
template <int N>
__device__ __forceinline__ void fft2_b_to_a(kcplx (&v)[Plan2<N>::PM], int q, int l, kcplx *X, const kcplx *W) {
  fft2<N, Plan2<N>::R1, Plan2<N>::R0>(v, q, l, X, W);
}

template <int N>
constexpr size_t lds_two() {
  return sizeof(kcplx) * (N + N * Plan2<N>::T);
}

// ---------------------------------------------------------------------------------------------
// strided c2c pass (PassArgs as k_pass): forward loads pattern A and stores pattern B, inverse the other way round
template <int N, bool INV, int NF>
__global__ void __launch_bounds__(Plan2<N>::NT, 2) k_pass2(PassArgs a, const kcplx *__restrict__ tw) {
  using Pl = Plan2<N>;
  using In = Own2<N, INV>;
  using Out = Own2<N, !INV>;
  constexpr int T = Pl::T, NT = Pl::NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  kcplx *W = reinterpret_cast<kcplx *>(smem);
  kcplx *X = W + N;
  if (a.stop && *a.stop) return;
  const int l = threadIdx.x % T, q = threadIdx.x / T;
  const unsigned logical = a.reverse ? xcd_remap_rev(blockIdx.x, gridDim.x) : xcd_remap(blockIdx.x, gridDim.x);
  const long long o = logical / a.tiles_per_outer;
  const long long i = (long long)(logical % a.tiles_per_outer) * T + l;
  const bool valid = i < a.inner;
  TwRegs<N, NT> twr;
  tw_issue<N, NT>(twr, tw);
  const long long ic = valid ? i : 0;
  kcplx v[NF][Pl::PM];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const kcplx *p = a.in[f] + o * a.so_in + ic + (long long)In::clamp(q) * a.sn_in;
#pragma unroll
    for (int m = 0; m < In::CNT; ++m) v[f][m] = p[(long long)m * In::STRIDE * a.sn_in];
  }
  tw_commit<N, NT>(twr, W);
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    if (INV) {
#pragma unroll
      for (int m = 0; m < In::CNT; ++m) v[f][m] = cswap(v[f][m]);
      fft2_b_to_a<N>(v[f], q, l, X, W);
    } else {
      fft2_a_to_b<N>(v[f], q, l, X, W);
    }
    if (valid && Out::active(q)) {
      kcplx *p = a.out[f] + o * a.so_out + i + (long long)q * a.sn_out;
#pragma unroll
      for (int m = 0; m < Out::CNT; ++m) p[(long long)m * Out::STRIDE * a.sn_out] = INV ? cswap(v[f][m]) : v[f][m];
    }
  }
}

template <int N, bool INV, int NF>
inline int launch_pass2(mrl_ctx *ctx, PassArgs a, const kcplx *tw) {
  static std::atomic<bool> attr{false};  // (two host threads may both set the attribute: harmless, and no torn flag)
  constexpr size_t lds = lds_two<N>();
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY(set_lds_attr(ctx, k_pass2<N, INV, NF>, lds));
    attr.store(true, std::memory_order_release);
  }
  constexpr int T = Plan2<N>::T;
  a.tiles_per_outer = (int)((a.inner + T - 1) / T);
  const long long nb = a.outer * a.tiles_per_outer;
  hipLaunchKernelGGL((k_pass2<N, INV, NF>), dim3((unsigned)nb), dim3(Plan2<N>::NT), lds, ctx->stream, a, tw);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

// ---------------------------------------------------------------------------------------------
// The fused Cahn-Hilliard x pass on a two-stage plan: what k_ch_xfused (ch_xfused.h) does for the uniform plans, on dense arrays
// [nx][inner]: forward x of mu-hat and c-hat (pattern A -> B), Nhat = Mbar mu-hat (stored: the history), ABM predictor with ORDER
// old Nhat, 1 / (1 - dt Lbar), inverse x (B -> A).  AdamsBashforthMoulton.C:94-101, ReciprocalLaplacianFactor.C:28-31,
// ReciprocalLaplacianSquareFactor.C:28-32; k^2 = (kx^2 + ky^2) + kz^2 as DomainAction.C:1503-1509; the same expressions in the same
// association as ch_fused_body.h.  Byte offsets are 32-bit (the launcher checks that an array is < 4 GiB).
struct X2Args {
  const kcplx *chat, *muhat;
  kcplx *ubar;          // out (may alias chat)
  kcplx *Nnew;          // out
  kcplx *cbar;          // optional out: c-hat of this substep
  const kcplx *Nold[4];
  kreal coef[5];        // sub_dt * beta[order][i]
  kreal M, kappa, dt;
  long long inner;      // ny * nzc = valid elements of an x plane
  long long plane;      // elements between two x planes (>= inner: the solver-private layout of the fused path pads it)
  int nzc;
  const kreal *kx, *ky, *kz;
};

// an opaque copy of a lane offset: the offsets of a later phase are recomputed from it instead of being kept (and spilled) since the
// first phase that used them
__device__ __forceinline__ unsigned fresh(unsigned x) {
  asm volatile("" : "+v"(x));
  return x;
}

// ... and one that cannot be formed before `after` exists
__device__ __forceinline__ unsigned fresh_after(unsigned x, kreal after) {
  asm volatile("" : "+v"(x) : "v"(after));
  return x;
}

// points per part of the history loop: the largest divisor of cnt with order * h <= 8
constexpr int part_len(int cnt, int order) {
  int best = 1;
  for (int h = 1; h <= cnt; ++h)
    if (cnt % h == 0 && order * h <= 8) best = h;
  return order == 0 ? cnt : best;
}

// REREAD (20 points per thread, and AB3 - AB5 everywhere): three arrays (of 80 registers) and deep histories do not fit, so Nhat is not
// kept: it is stored (cached, not streamed) and read again with the history in the update -- each thread re-reads exactly what it
// wrote, a few hundred cycles later, out of L2.  No kernel of the family spills with it (without: 35-270 VGPRs at AB3 - AB5).
template <int N, int ORDER, bool NT_HIST, bool REREAD = (Plan2<N>::PM > 16 || ORDER >= 2)>
__global__ void __launch_bounds__(Plan2<N>::NT, 2) k_ch_xfused2(X2Args a, const kcplx *__restrict__ tw) {
#pragma clang fp contract(off)
  using Pl = Plan2<N>;
  using A = Own2<N, false>;
  using B = Own2<N, true>;
  constexpr int T = Pl::T, NT = Pl::NT, PM = Pl::PM, CNT = (N + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  kcplx *W = reinterpret_cast<kcplx *>(smem);
  kcplx *X = W + N;
  kreal *KX = reinterpret_cast<kreal *>(X + N * T);
  const int l = threadIdx.x % T, q = threadIdx.x / T;
  const long long i = (long long)xcd_remap(blockIdx.x, gridDim.x) * T + l;
  const bool valid = i < a.inner;
  const long long ic = valid ? i : 0;
  kcplx twv[CNT];
  kreal kxv[CNT];
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    const int idx = threadIdx.x + j * NT;
    twv[j] = idx < N ? tw[idx] : mkc(0.0, 0.0);
    kxv[j] = idx < N ? a.kx[idx] : 0.0;
  }
  const kreal ky = a.ky[ic / a.nzc], kz = a.kz[ic % a.nzc];
  // byte offsets of the thread's elements: pattern A (q + R1 t), pattern B (q + R0 t')
  const unsigned plane = (unsigned)a.plane * (unsigned)sizeof(kcplx);
  const unsigned offA0 = ((unsigned)ic + (unsigned)A::clamp(q) * (unsigned)a.plane) * (unsigned)sizeof(kcplx), stepA = A::STRIDE * plane;
  const unsigned offB0 = ((unsigned)ic + (unsigned)B::clamp(q) * (unsigned)a.plane) * (unsigned)sizeof(kcplx), stepB = B::STRIDE * plane;
  kcplx v[PM], cp[PM];
#pragma unroll
  for (int m = 0; m < A::CNT; ++m) v[m] = ldc(a.muhat, offA0 + m * stepA);
  if (!REREAD) {
#pragma unroll
    for (int m = 0; m < A::CNT; ++m) cp[m] = ldc(a.chat, offA0 + m * stepA);
  }
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    const int idx = threadIdx.x + j * NT;
    if (idx < N) {
      W[idx] = twv[j];
      KX[idx] = kxv[j];
    }
  }
  const kreal ky2 = ky * ky, kz2 = kz * kz;
  const bool storeB = valid && B::active(q);
  const int qb = B::clamp(q);

  // ---- 1. mu-hat: forward x (its exchange publishes W and KX to the workgroup)
  fft2_a_to_b<N>(v, q, l, X, W);

  // ---- 2. Nhat = Mbar * mu-hat, Mbar = -k^2 M
  kcplx Nv[REREAD ? 1 : PM];
#pragma unroll
  for (int m = 0; m < B::CNT; ++m) {
    const kreal kl = KX[qb + m * B::STRIDE];
    const kreal Mbar = -((kl * kl + ky2) + kz2) * a.M;
    const kcplx nh = mkc(Mbar * v[m].x, Mbar * v[m].y);
    if (REREAD)
      v[m] = nh;
    else
      Nv[m] = nh;
  }
  if (storeB) {
#pragma unroll
    for (int m = 0; m < B::CNT; ++m) {
      if (REREAD)
        stc(a.Nnew, offB0 + m * stepB, v[m]);
      else if (NT_HIST)
        stc_nt(a.Nnew, offB0 + m * stepB, Nv[m]);
      else
        stc(a.Nnew, offB0 + m * stepB, Nv[m]);
    }
  }
  if (REREAD) {  // the second field, now that the first one's registers are free
#pragma unroll
    for (int m = 0; m < A::CNT; ++m) cp[m] = ldc(a.chat, offA0 + m * stepA);
  }
  // ---- first-order history: half of it is requested before the c-hat transform and is in flight during it
  constexpr int PRE = B::CNT / 2;
  const unsigned offB1 = fresh(offB0);
  kcplx o1[(ORDER == 1 && !REREAD) ? PM : 1];
  if (ORDER == 1 && !REREAD) {
#pragma unroll
    for (int m = 0; m < PRE; ++m) o1[ORDER == 1 ? m : 0] = NT_HIST ? ldc_nt(a.Nold[0], offB1 + m * stepB) : ldc(a.Nold[0], offB1 + m * stepB);
  }

  // ---- 3. c-hat: forward x
  fft2_a_to_b<N>(cp, q, l, X, W);
  const unsigned offB2 = fresh(offB0);
  if (a.cbar && storeB) {
#pragma unroll
    for (int m = 0; m < B::CNT; ++m) stc_nt(a.cbar, offB2 + m * stepB, cp[m]);
  }

  // ---- 4. ubar = (cbar + (dt b0) N + sum (dt b_i) Nold_i) / (1 - dt Lbar), the reference's association.  The first term is formed
  //      at once, so that N's registers are free before the history arrives
  if (!REREAD) {
#pragma unroll
    for (int m = 0; m < B::CNT; ++m) {
      cp[m].x = cp[m].x + a.coef[0] * Nv[m].x;
      cp[m].y = cp[m].y + a.coef[0] * Nv[m].y;
    }
  }
  if (ORDER == 1 && !REREAD) {
#pragma unroll
    for (int m = PRE; m < B::CNT; ++m) o1[ORDER == 1 ? m : 0] = NT_HIST ? ldc_nt(a.Nold[0], offB2 + m * stepB) : ldc(a.Nold[0], offB2 + m * stepB);
#pragma unroll
    for (int m = 0; m < B::CNT; ++m) {
      kcplx u = cp[m];
      u.x += a.coef[1] * o1[ORDER == 1 ? m : 0].x;
      u.y += a.coef[1] * o1[ORDER == 1 ? m : 0].y;
      const kreal kl = KX[qb + m * B::STRIDE];
      const kreal k2 = (kl * kl + ky2) + kz2;
      const kreal Lb = k2 * k2 * a.kappa;
      const kreal scl = kreal(1.0) / (kreal(1.0) - a.dt * Lb);
      v[m] = mkc(u.y * scl, u.x * scl);  // swapped for the inverse transform
    }
  } else {
    // deeper histories (and the re-read Nhat) a part of the line at a time: at most 8 old values in flight per thread
    constexpr int NH = ORDER + (REREAD ? 1 : 0);
    constexpr int H = part_len(B::CNT, NH);
    constexpr int NPART = B::CNT / H;
    static_assert(NPART * H == B::CNT, "the history parts must cover the line");
#pragma unroll
    for (int part = 0; part < NPART; ++part) {
      kcplx o[ORDER > 0 ? ORDER : 1][H];
      kcplx nn[REREAD ? H : 1];
      // (a copy per part that depends on the last result of the part before: its loads cannot be issued, and their landing registers
      // held, before that part has been consumed -- hoisted to the top they spill 40-170 registers)
      const unsigned ob = (part == 0 ? fresh(offB0) : fresh_after(offB0, v[part * H - 1].x)) + (part * H) * stepB;
#pragma unroll
      for (int h = 0; h < ORDER; ++h) {
#pragma unroll
        for (int j = 0; j < H; ++j) o[h][j] = NT_HIST ? ldc_nt(a.Nold[h], ob + j * stepB) : ldc(a.Nold[h], ob + j * stepB);
      }
      if (REREAD) {
#pragma unroll
        for (int j = 0; j < H; ++j) nn[j] = ldc(a.Nnew, ob + j * stepB);
      }
#pragma unroll
      for (int j = 0; j < H; ++j) {
        const int m = part * H + j;
        kcplx u = cp[m];
        if (REREAD) {
          u.x = u.x + a.coef[0] * nn[j].x;
          u.y = u.y + a.coef[0] * nn[j].y;
        }
#pragma unroll
        for (int h = 0; h < ORDER; ++h) {
          u.x += a.coef[h + 1] * o[h][j].x;
          u.y += a.coef[h + 1] * o[h][j].y;
        }
        const kreal kl = KX[qb + m * B::STRIDE];
        const kreal k2 = (kl * kl + ky2) + kz2;
        const kreal Lb = k2 * k2 * a.kappa;
        const kreal scl = kreal(1.0) / (kreal(1.0) - a.dt * Lb);
        v[m] = mkc(u.y * scl, u.x * scl);
      }
    }
  }

  // ---- 5. inverse x (unnormalised; 1/N applied by the final z pass)
  fft2_b_to_a<N>(v, q, l, X, W);
  if (valid && A::active(q)) {
    const unsigned offA1 = fresh(offA0);
#pragma unroll
    for (int m = 0; m < A::CNT; ++m) stc(a.ubar, offA1 + m * stepA, cswap(v[m]));
  }
}

template <int N, int ORDER>
inline int launch_xfused2(mrl_ctx *ctx, const X2Args &a, const kcplx *tw) {
  constexpr size_t lds = lds_two<N>() + sizeof(kreal) * N;
  constexpr int T = Plan2<N>::T;
  const long long nb = (a.inner + T - 1) / T;
  // old / new Nhat are not touched again within the substep: streamed past the Infinity Cache when the arrays are large against it
  // (the choice of ch_xfused.h: 128^3 loses 8 % with it, 256^3 gains 10 %)
  const bool nt = (double)sizeof(kcplx) * (double)N * (double)a.plane >= 96.0e6;
  if (nt) {
    static std::atomic<bool> attr{false};
    if (!attr.load(std::memory_order_acquire)) {
      MRL_TRY((set_lds_attr(ctx, k_ch_xfused2<N, ORDER, true>, lds)));
      attr.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_ch_xfused2<N, ORDER, true>), dim3((unsigned)nb), dim3(Plan2<N>::NT), lds, ctx->stream, a, tw);
  } else {
    static std::atomic<bool> attr{false};
    if (!attr.load(std::memory_order_acquire)) {
      MRL_TRY((set_lds_attr(ctx, k_ch_xfused2<N, ORDER, false>, lds)));
      attr.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_ch_xfused2<N, ORDER, false>), dim3((unsigned)nb), dim3(Plan2<N>::NT), lds, ctx->stream, a, tw);
  }
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

}  // namespace MRL_P2NS
}  // namespace mrl
