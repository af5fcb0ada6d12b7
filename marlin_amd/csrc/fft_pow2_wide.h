// "Wide" plans for long power-of-two lines of the PLAIN strided passes: 32 points per thread, two radix stages (32 x 16 for 512
// points) instead of three, 16 lines per workgroup (256-byte segments instead of 128-byte ones), and the one Stockham exchange done
// for the real and the imaginary parts separately so that the LDS tile stays at 64 KB (two workgroups per CU).  One array is 128
// vector registers: plain one-field passes only (the fused kernels hold three to four arrays and keep the 16-point plans).
#pragma once
#include <atomic>
#include "fft_pow2_launch.h"

namespace mrl {
namespace p2 {

// radix 32 = 2 x 16 (decimation in frequency): X[2 k2] = DFT16(a[n] + a[n+16]), X[1 + 2 k2] = DFT16((a[n] - a[n+16]) W32^n)
template <>
__device__ __forceinline__ void bfly<32>(cplx (&a)[32]) {
  cplx e[16], o[16];
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    e[n] = cadd(a[n], a[n + 16]);
    o[n] = csub(a[n], a[n + 16]);
  }
  o[1] = cmul(o[1], make_double2(0.980785280403230449126, -0.195090322016128267848));  // W32^1
  o[2] = cmul(o[2], make_double2(0.923879532511286756128, -0.382683432365089771728));  // W32^2
  o[3] = cmul(o[3], make_double2(0.831469612302545237079, -0.555570233019602224743));  // W32^3
  o[4] = cmul(o[4], make_double2(0.707106781186547524401, -0.707106781186547524401));  // W32^4
  o[5] = cmul(o[5], make_double2(0.555570233019602224743, -0.831469612302545237079));  // W32^5
  o[6] = cmul(o[6], make_double2(0.382683432365089771728, -0.923879532511286756128));  // W32^6
  o[7] = cmul(o[7], make_double2(0.195090322016128267848, -0.980785280403230449126));  // W32^7
  o[8] = cmul(o[8], make_double2(8.47842766036889964396e-32, -1.0));  // W32^8
  o[9] = cmul(o[9], make_double2(-0.195090322016128267848, -0.980785280403230449126));  // W32^9
  o[10] = cmul(o[10], make_double2(-0.382683432365089771728, -0.923879532511286756128));  // W32^10
  o[11] = cmul(o[11], make_double2(-0.555570233019602224743, -0.831469612302545237079));  // W32^11
  o[12] = cmul(o[12], make_double2(-0.707106781186547524401, -0.707106781186547524401));  // W32^12
  o[13] = cmul(o[13], make_double2(-0.831469612302545237079, -0.555570233019602224743));  // W32^13
  o[14] = cmul(o[14], make_double2(-0.923879532511286756128, -0.382683432365089771728));  // W32^14
  o[15] = cmul(o[15], make_double2(-0.980785280403230449126, -0.195090322016128267848));  // W32^15
  bfly<16>(e);
  bfly<16>(o);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    a[2 * k] = e[k];
    a[2 * k + 1] = o[k];
  }
}

struct Wide512 {
  static constexpr int N = 512, P = 32, r0 = 32, r1 = 16, T = 16, TPL = 16, NT = 256;
};

template <class PL, int R, int NS>
__device__ __forceinline__ void stage_w(cplx (&v)[PL::P], int q, const cplx *W) {
  constexpr int P = PL::P, S = P / R, TPL = PL::TPL, N = PL::N;
#pragma unroll
  for (int i = 0; i < S; ++i) {
    cplx a[R];
#pragma unroll
    for (int t = 0; t < R; ++t) a[t] = v[i + S * t];
    if (NS > 1) {
      const int b = q + i * TPL;
      const int step = (b % NS) * (N / (NS * R));
#pragma unroll
      for (int t = 1; t < R; ++t) a[t] = cmul(a[t], W[t * step]);
    }
    bfly<R>(a);
#pragma unroll
    for (int t = 0; t < R; ++t) v[i + S * t] = a[t];
  }
}

// Stockham exchange after a radix-R stage (Ns = NS), real parts first, then imaginary parts, through a tile of N x T doubles
template <class PL, int R, int NS>
__device__ __forceinline__ void exchange_split(cplx (&v)[PL::P], int q, int l, double *X) {
  constexpr int P = PL::P, S = P / R, TPL = PL::TPL, T = PL::T;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();  // previous readers of X are done
#pragma unroll
    for (int i = 0; i < S; ++i) {
      const int b = q + i * TPL;
      const int p0 = (b / NS) * NS * R + (b % NS);
#pragma unroll
      for (int t = 0; t < R; ++t) X[(p0 + t * NS) * T + l] = half ? v[i + S * t].y : v[i + S * t].x;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < P; ++m) {
      const double w = X[(q + m * TPL) * T + l];
      if (half)
        v[m].y = w;
      else
        v[m].x = w;
    }
  }
}

template <class PL>
__device__ __forceinline__ void fft_line_w(cplx (&v)[PL::P], int q, int l, double *X, const cplx *W) {
  stage_w<PL, PL::r0, 1>(v, q, W);
  exchange_split<PL, PL::r0, 1>(v, q, l, X);
  stage_w<PL, PL::r1, PL::r0>(v, q, W);
}

template <class PL>
constexpr size_t lds_wide() {
  return sizeof(cplx) * PL::N + sizeof(double) * PL::N * PL::T;
}

// the strided c2c pass of k_pass (one field), wide plan
template <class PL, bool INV>
__global__ void __launch_bounds__(PL::NT, 2) k_pass_w(PassArgs a, const cplx *__restrict__ tw) {
  constexpr int P = PL::P, TPL = PL::TPL, T = PL::T, N = PL::N, NT = PL::NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx *W = reinterpret_cast<cplx *>(smem);
  double *X = reinterpret_cast<double *>(W + N);
  if (a.stop && *a.stop) return;
  const int l = threadIdx.x % T, q = threadIdx.x / T;
  const unsigned logical = a.reverse ? xcd_remap_rev(blockIdx.x, gridDim.x) : xcd_remap(blockIdx.x, gridDim.x);
  const long long o = logical / a.tiles_per_outer;
  const long long i = (long long)(logical % a.tiles_per_outer) * T + l;
  const bool valid = i < a.inner;
  TwRegs<N, NT> twr;
  tw_issue<N, NT>(twr, tw);
  const long long ic = valid ? i : 0;
  cplx v[P];
  const cplx *p = a.in[0] + o * a.so_in + ic + (long long)q * a.sn_in;
#pragma unroll
  for (int m = 0; m < P; ++m) v[m] = p[(long long)m * TPL * a.sn_in];
  tw_commit<N, NT>(twr, W);
  if (INV) {
#pragma unroll
    for (int m = 0; m < P; ++m) v[m] = cswap(v[m]);
  }
  fft_line_w<PL>(v, q, l, X, W);
  if (valid) {
    cplx *d = a.out[0] + o * a.so_out + i + (long long)q * a.sn_out;
#pragma unroll
    for (int m = 0; m < P; ++m) d[(long long)m * TPL * a.sn_out] = INV ? cswap(v[m]) : v[m];
  }
}

// ... and of k_pass_sub (kz sub-block of the slab pipeline, scatter through the destination table), one field per launch.
// UTAB: the chunk length 2^sh_out is a multiple of TPL, so the chunk index of element n = q + m TPL depends on m only and the table
// entry is a scalar load instead of one vector load per element in front of every store
template <class PL, bool INV, bool UTAB = false>
__global__ void __launch_bounds__(PL::NT, 2) k_pass_sub_w(SubPassArgs a, const cplx *__restrict__ tw) {
  constexpr int P = PL::P, TPL = PL::TPL, T = PL::T, N = PL::N, NT = PL::NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx *W = reinterpret_cast<cplx *>(smem);
  double *X = reinterpret_cast<double *>(W + N);
  const int l = threadIdx.x % T, q = threadIdx.x / T;
  unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
  // fields back to back in ONE launch (a.nb workgroups each): twice the workgroups between two launch boundaries
  const unsigned f = logical >= a.nb ? 1u : 0u;
  logical -= f * a.nb;
  const cplx *__restrict__ src = a.in[f];
  const unsigned fo = a.fo_out + f * a.fs_out;
  const unsigned i = logical * T + l;
  const bool valid = i < (unsigned)(a.rows * a.tcols);
  const unsigned ic = valid ? i : 0u;
  const unsigned row = ic / (unsigned)a.tcols, col = ic - row * (unsigned)a.tcols;
  const unsigned bi = row * a.pitch_in + min(col, (unsigned)a.cols - 1u), bo = row * a.pitch_out + col;
  const unsigned mi = (a.sh_in < 31) ? ((1u << a.sh_in) - 1u) : 0xffffffffu;
  const unsigned mo = (a.sh_out < 31) ? ((1u << a.sh_out) - 1u) : 0xffffffffu;
  TwRegs<N, NT> twr;
  tw_issue<N, NT>(twr, tw);
  cplx v[P];
#pragma unroll
  for (int m = 0; m < P; ++m) {
    const unsigned n = q + m * TPL;
    v[m] = src[bi + (a.sh_in < 31 ? (n >> a.sh_in) * a.cs_in : 0u) + (n & mi) * a.sn_in];
  }
  tw_commit<N, NT>(twr, W);
  if (INV) {
#pragma unroll
    for (int m = 0; m < P; ++m) v[m] = cswap(v[m]);
  }
  fft_line_w<PL>(v, q, l, X, W);
  if (valid) {
#pragma unroll
    for (int m = 0; m < P; ++m) {
      const unsigned n = q + m * TPL;
      if (INV) {
        a.out[0][bo + (n & mo) * a.sn_out] = cswap(v[m]);
      } else {
        cplx *base = UTAB ? a.otab[(unsigned)(m * TPL) >> a.sh_out] : a.otab[n >> a.sh_out];
        base[fo + bo + (n & mo) * a.sn_out] = v[m];
      }
    }
  }
  if (!INV) signal_tail(a.sig);
}

template <class PL, bool INV>
inline int launch_pass_w(mrl_ctx *ctx, PassArgs a, const cplx *tw) {
  static std::atomic<bool> attr{false};  // (two host threads may both set the attribute: harmless, and no torn flag)
  constexpr size_t lds = lds_wide<PL>();
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY((set_lds_attr(ctx, k_pass_w<PL, INV>, lds)));
    attr.store(true, std::memory_order_release);
  }
  a.tiles_per_outer = (int)((a.inner + PL::T - 1) / PL::T);
  const long long nb = a.outer * a.tiles_per_outer;
  hipLaunchKernelGGL((k_pass_w<PL, INV>), dim3((unsigned)nb), dim3(PL::NT), lds, ctx->stream, a, tw);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

template <class PL, bool INV, bool UTAB>
inline int launch_pass_sub_w_v(mrl_ctx *ctx, const SubPassArgs &a, const cplx *tw, long long nb) {  // nb = all workgroups
  static std::atomic<bool> attr{false};  // (two host threads may both set the attribute: harmless, and no torn flag)
  constexpr size_t lds = lds_wide<PL>();
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY((set_lds_attr(ctx, k_pass_sub_w<PL, INV, UTAB>, lds)));
    attr.store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL((k_pass_sub_w<PL, INV, UTAB>), dim3((unsigned)nb), dim3(PL::NT), lds, ctx->stream, a, tw);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

// nf = 1 or 2 fields (a.in[f] -> chunk offset fo_out + f * fs_out), INV: one field
template <class PL, bool INV>
inline int launch_pass_sub_w(mrl_ctx *ctx, SubPassArgs a, const cplx *tw, int nf = 1) {
  if (a.tcols == 0) a.tcols = a.cols;
  a.nb = (unsigned)(((long long)a.rows * a.tcols + PL::T - 1) / PL::T);
  const long long nb = (long long)nf * a.nb;
  if (a.sig.expected == 0) a.sig.expected = (unsigned)nb;
  if constexpr (!INV) {
    if (a.sh_out < 31 && (1 << a.sh_out) % PL::TPL == 0) return launch_pass_sub_w_v<PL, INV, true>(ctx, a, tw, nb);
  }
  return launch_pass_sub_w_v<PL, INV, false>(ctx, a, tw, nb);
}

}  // namespace p2
}  // namespace mrl
