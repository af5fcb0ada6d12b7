// Table-addressed slab kernels: the chunked exchange layouts of partitions that are NOT equal powers of two (the reference's 200^3 example
// grid on 2 or 4 ranks, its 3-rank 64^3 test, device_weights).  Shared by the Cahn-Hilliard pipeline (slab_fused.hip) and the Gamma
// operator (slab_mech_fused.hip).  fp64 only.
#pragma once
#include <atomic>

#include "fft_pow2_launch.h"
#include "fft_pow2_wide.h"

namespace mrl {
namespace p2 {

// Table-addressed forms of the two slab kernels that see the chunked exchange layouts, for partitions that are not equal powers of
// two (slab_fast_table).  Same transforms, same pointwise arithmetic in the same order as the shift-addressed kernels (ch_fused_body):
// only the address of an element differs -- the chunk index and the offset inside the chunk are looked up (tables: L1 / L2 resident,
// <= 5 x 4 bytes per line element) instead of being split off the index by shifts.
struct SubPassTabs {
  const unsigned *xch;    // [nx] rank whose chunk holds x plane n
  const unsigned *xoff;   // [nx] element offset of plane n inside a field block of that chunk: (n - first plane) * plane pitch
  const unsigned *fsz;    // [P]  elements between two fields of the chunk for rank p (host side; the kernels use xfs)
  const unsigned *xin;    // [nx] inverse: element offset of plane n, field 0, in the received buffer (chunk offset + plane offset)
  const unsigned *xfs;    // [nx] inverse: elements between two fields of the chunk that holds plane n
};

template <int N, bool INV, int NF>
__global__ void __launch_bounds__(Plan<N>::NT, 2) k_pass_sub_t(SubPassArgs a, SubPassTabs t, const cplx *__restrict__ tw) {
  constexpr int P = Plan<N>::P, TPL = Plan<N>::TPL, T = Plan<N>::T;
  using Map = MapStrided<N>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx *W = reinterpret_cast<cplx *>(smem);
  cplx *X = W + N;
  const int l = threadIdx.x % T, q = threadIdx.x / T;
  const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned i = logical * T + l;
  const bool valid = i < (unsigned)(a.rows * a.tcols);
  const unsigned ic = valid ? i : 0u;
  const unsigned row = ic / (unsigned)a.tcols, col = ic - row * (unsigned)a.tcols;
  const unsigned bi = row * a.pitch_in + min(col, (unsigned)a.cols - 1u), bo = row * a.pitch_out + col;
  TwRegs<N> twr;
  tw_issue<N>(twr, tw);
  cplx v[NF][P];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
#pragma unroll
    for (int m = 0; m < P; ++m) {
      const unsigned n = q + m * TPL;
      if (INV)
        v[f][m] = a.in[f][bi + t.xin[n]];
      else
        v[f][m] = a.in[f][bi + n * a.sn_in];
    }
  }
  tw_commit<N>(twr, W);
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    if (INV) {
#pragma unroll
      for (int m = 0; m < P; ++m) v[f][m] = cswap(v[f][m]);
    }
    fft_line<N, Map>(v[f], q, l, X, W);
    if (valid) {
#pragma unroll
      for (int m = 0; m < P; ++m) {
        const unsigned n = q + m * TPL;
        if (INV) {
          a.out[f][bo + n * a.sn_out] = cswap(v[f][m]);
        } else {
          a.otab[t.xch[n]][(unsigned)f * t.xfs[n] + bo + t.xoff[n]] = v[f][m];
        }
      }
    }
  }
  if (!INV) signal_tail(a.sig);
}

template <int N, bool INV, int NF>
inline int launch_pass_sub_t(mrl_ctx *ctx, SubPassArgs a, const SubPassTabs &t, const cplx *tw) {
  static std::atomic<bool> attr{false};  // (two host threads may both set the attribute: harmless, and no torn flag)
  constexpr size_t lds = lds_strided<N>();
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY((set_lds_attr(ctx, k_pass_sub_t<N, INV, NF>, lds)));
    attr.store(true, std::memory_order_release);
  }
  constexpr int T = Plan<N>::T;
  if (a.tcols == 0) a.tcols = a.cols;
  const long long nb = ((long long)a.rows * a.tcols + T - 1) / T;
  if (a.sig.expected == 0) a.sig.expected = (unsigned)nb;
  hipLaunchKernelGGL((k_pass_sub_t<N, INV, NF>), dim3((unsigned)nb), dim3(Plan<N>::NT), lds, ctx->stream, a, t, tw);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

// k_pass_sub_t over SEVERAL fields in one launch (the nine fields of the slab Gamma operator; compare k_pass_sub_mf): gridDim = nf * a.nb,
// the field is the slow block index.  Field f is a.in[0] / a.out[0] + f * fdense on the dense (rank-local) side and lies f * fsz[p]
// elements into the chunk of rank p on the exchange-layout side (cofi: built for nf fields per chunk).
template <int N, bool INV>
__global__ void __launch_bounds__(Plan<N>::NT, 2) k_pass_sub_mft(SubPassArgs a, SubPassTabs t, const cplx *__restrict__ tw) {
  constexpr int P = Plan<N>::P, TPL = Plan<N>::TPL, T = Plan<N>::T;
  using Map = MapStrided<N>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx *W = reinterpret_cast<cplx *>(smem);
  cplx *X = W + N;
  const int l = threadIdx.x % T, q = threadIdx.x / T;
  unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned f = logical / a.nb;
  logical -= f * a.nb;
  const unsigned i = logical * T + l;
  const bool valid = i < (unsigned)(a.rows * a.tcols);
  const unsigned ic = valid ? i : 0u;
  const unsigned row = ic / (unsigned)a.tcols, col = ic - row * (unsigned)a.tcols;
  const unsigned bi = row * a.pitch_in + min(col, (unsigned)a.cols - 1u), bo = row * a.pitch_out + col;
  TwRegs<N> twr;
  tw_issue<N>(twr, tw);
  cplx v[P];
#pragma unroll
  for (int m = 0; m < P; ++m) {
    const unsigned n = q + m * TPL;
    if (INV) {
      v[m] = a.in[0][bi + t.xin[n] + f * t.xfs[n]];
    } else {
      v[m] = a.in[0][(size_t)f * a.fdense + bi + n * a.sn_in];
    }
  }
  tw_commit<N>(twr, W);
  if (INV) {
#pragma unroll
    for (int m = 0; m < P; ++m) v[m] = cswap(v[m]);
  }
  fft_line<N, Map>(v, q, l, X, W);
  if (valid) {
#pragma unroll
    for (int m = 0; m < P; ++m) {
      const unsigned n = q + m * TPL;
      if (INV) {
        a.out[0][(size_t)f * a.fdense + bo + n * a.sn_out] = cswap(v[m]);
      } else {
        a.otab[t.xch[n]][f * t.xfs[n] + bo + t.xoff[n]] = v[m];
      }
    }
  }
  if (!INV) signal_tail(a.sig);
}

template <int N, bool INV>
inline int launch_pass_sub_mft(mrl_ctx *ctx, SubPassArgs a, const SubPassTabs &t, const cplx *tw, int nf) {
  static std::atomic<bool> attr{false};  // (two host threads may both set the attribute: harmless, and no torn flag)
  constexpr size_t lds = lds_strided<N>();
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY((set_lds_attr(ctx, k_pass_sub_mft<N, INV>, lds)));
    attr.store(true, std::memory_order_release);
  }
  constexpr int T = Plan<N>::T;
  if (a.tcols == 0) a.tcols = a.cols;
  a.nb = (unsigned)(((long long)a.rows * a.tcols + T - 1) / T);
  const long long nb = (long long)nf * a.nb;
  if (a.sig.expected == 0) a.sig.expected = (unsigned)nb;
  hipLaunchKernelGGL((k_pass_sub_mft<N, INV>), dim3((unsigned)nb), dim3(Plan<N>::NT), lds, ctx->stream, a, t, tw);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

// The wide 512-point plan (fft_pow2_wide.h: 32 points per thread, two stages, 256-byte segments) with table-addressed chunks: nf = 1 or 2
// fields back to back in one launch (forward), one field (inverse).  Same transform as k_pass_sub_w, same addressing as k_pass_sub_t.
template <class PL, bool INV>
__global__ void __launch_bounds__(PL::NT, 2) k_pass_sub_wt(SubPassArgs a, SubPassTabs t, const cplx *__restrict__ tw) {
  constexpr int P = PL::P, TPL = PL::TPL, T = PL::T, N = PL::N, NT = PL::NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx *W = reinterpret_cast<cplx *>(smem);
  double *X = reinterpret_cast<double *>(W + N);
  const int l = threadIdx.x % T, q = threadIdx.x / T;
  unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned f = logical >= a.nb ? 1u : 0u;
  logical -= f * a.nb;
  const cplx *__restrict__ src = a.in[f];
  const unsigned i = logical * T + l;
  const bool valid = i < (unsigned)(a.rows * a.tcols);
  const unsigned ic = valid ? i : 0u;
  const unsigned row = ic / (unsigned)a.tcols, col = ic - row * (unsigned)a.tcols;
  const unsigned bi = row * a.pitch_in + min(col, (unsigned)a.cols - 1u), bo = row * a.pitch_out + col;
  TwRegs<N, NT> twr;
  tw_issue<N, NT>(twr, tw);
  cplx v[P];
#pragma unroll
  for (int m = 0; m < P; ++m) {
    const unsigned n = q + m * TPL;
    v[m] = INV ? src[bi + t.xin[n]] : src[bi + n * a.sn_in];
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
      if (INV)
        a.out[0][bo + n * a.sn_out] = cswap(v[m]);
      else
        a.otab[t.xch[n]][f * t.xfs[n] + bo + t.xoff[n]] = v[m];
    }
  }
  if (!INV) signal_tail(a.sig);
}

template <class PL, bool INV>
inline int launch_pass_sub_wt(mrl_ctx *ctx, SubPassArgs a, const SubPassTabs &t, const cplx *tw, int nf = 1) {
  static std::atomic<bool> attr{false};  // (two host threads may both set the attribute: harmless, and no torn flag)
  constexpr size_t lds = lds_wide<PL>();
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY((set_lds_attr(ctx, k_pass_sub_wt<PL, INV>, lds)));
    attr.store(true, std::memory_order_release);
  }
  if (a.tcols == 0) a.tcols = a.cols;
  a.nb = (unsigned)(((long long)a.rows * a.tcols + PL::T - 1) / PL::T);
  const long long nb = (long long)nf * a.nb;
  if (a.sig.expected == 0) a.sig.expected = (unsigned)nb;
  hipLaunchKernelGGL((k_pass_sub_wt<PL, INV>), dim3((unsigned)nb), dim3(PL::NT), lds, ctx->stream, a, t, tw);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

struct YTabs {
  const unsigned *ych;  // [ny] rank whose chunk holds y row j
  const unsigned *yD;   // [ny] (j - first row of that chunk) * kp
  const unsigned *yB;   // [ny] x-plane pitch of that chunk
  const unsigned *yC;   // [ny] elements between the two fields of that chunk (forward, two fields)
  const unsigned *yA;   // [ny] element offset of row j at x plane 0, field 0, in the received forward buffer
};


}  // namespace p2
}  // namespace mrl
