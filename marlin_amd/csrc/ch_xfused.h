// The fused Cahn-Hilliard x pass (forward x on mu-hat and c-hat, Nhat = Mbar mu-hat, ABM predictor, 1/(1 - dt Lbar), inverse x) and its
// launchers: shared by ch_fused.hip (fp64, the product path) and ch_fused_f32.hip (the fp32 instantiation: MRL_KREAL = float).
#pragma once
#include <atomic>
#include "ch_fused_body.h"
#include "fft_pow2_launch.h"

namespace mrl {

namespace MRL_P2NS {

struct FusedArgs {
  FusedCommon c;      // chat/muhat/ubar, Nnew/cbar/Nold: all in the solver-private layout [x][plane], rows [y][kz] inside a plane
  long long inner;    // ny*nzc: valid elements of a plane
  long long plane;    // elements between two x planes (>= inner; mrl_ctx::spec_plane)
  int nzc;
  const kreal *kx, *ky, *kz;
};

// NT: stream the arrays that are not re-read within the substep past the Infinity Cache (ch_fused_body.h); chosen per launch
template <int N, int ORDER, int PRE, bool SPEC_C, bool NT, bool BIG = false>
__global__ void __launch_bounds__(Plan<N>::NT, 2) k_ch_xfused(FusedArgs a, const kcplx *__restrict__ tw) {
  constexpr int TPL = Plan<N>::TPL, T = Plan<N>::T;
  using Map = MapStrided<N>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  kcplx *W = reinterpret_cast<kcplx *>(smem);
  kcplx *X = W + N;
  kreal *KX = reinterpret_cast<kreal *>(X + Map::size);
  const int l = threadIdx.x % T, q = threadIdx.x / T;
  const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
  const long long i = (long long)logical * T + l;
  const bool valid = i < a.inner;
  const long long iv = valid ? i : 0;
  // byte offset of line element m: (i + (q + m*TPL)*plane) * 16, the same for every array of the pass
  kcplx *const ubar = a.c.ubar;
  if constexpr (BIG) {  // arrays >= 4 GiB: m * (line stride) is wave-uniform and 64-bit, the rest fits 32 bits (checked by the launcher)
    const unsigned off0 = (unsigned)((iv + (long long)q * a.plane) * sizeof(kcplx));
    const unsigned long long step = (unsigned long long)(TPL * a.plane) * (unsigned long long)sizeof(kcplx);
    auto off = [=](int m) { return BigOff{(unsigned long long)m * step, off0}; };
    auto stu = [=](int m, kcplx val) { stc(ubar, BigOff{(unsigned long long)m * step, off0}, val); };
    ch_fused_body<N, ORDER, true, PRE, SPEC_C, NT, NT, NT>(a.c, tw, a.kx, a.ky + iv / a.nzc, a.kz + iv % a.nzc, valid, q, l, off, OffSame{}, off, stu, W, X, KX);
  } else {
    const unsigned off0 = (unsigned)(iv + (long long)q * a.plane) * (unsigned)sizeof(kcplx), step = (unsigned)(TPL * a.plane) * (unsigned)sizeof(kcplx);
    auto off = [=](int m) { return off0 + (unsigned)m * step; };
    auto stu = [=](int m, kcplx val) { stc(ubar, off0 + (unsigned)m * step, val); };
    ch_fused_body<N, ORDER, true, PRE, SPEC_C, NT, NT, NT>(a.c, tw, a.kx, a.ky + iv / a.nzc, a.kz + iv % a.nzc, valid, q, l, off, OffSame{}, off, stu, W, X, KX);
  }
}

// lengths for which the 64-bit-offset variant is instantiated (x axes long enough for a >= 4 GiB half-spectrum array)
template <int N>
constexpr bool big_capable() {
  return N == 512 || N == 768 || N == 1000 || N == 1024;
}

// The non-temporal variant pays off when the arrays are large against the 256 MB Infinity Cache (a resident old Nhat is a hit
// that streaming gives away: 128^3 +8 %, 200^3 +0.5 %, 256^3 -10 %, 384^3 -3.5 %, 512^3 -2 % in A/B runs) and when a thread's
// accesses cover whole 128-B lines (T lines x 16 B; 400^3 with its 96-B pieces: +7 %): instantiated for the long x axes only.
template <int N>
constexpr bool nt_capable() {
  return (N == 256 || N == 384 || N == 512) && (Plan<N>::T * sizeof(kcplx)) % 128 == 0;
}

template <int N, int ORDER, bool SPEC_C, bool NT, int PRE, bool BIG = false>
static int launch_xfused_v(mrl_ctx *ctx, const FusedArgs &a, const kcplx *tw) {
  static std::atomic<bool> attr{false};  // (two host threads may both set the attribute: harmless, and no torn flag)
  constexpr size_t lds = lds_strided<N>() + sizeof(kreal) * N;
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY((set_lds_attr(ctx, k_ch_xfused<N, ORDER, PRE, SPEC_C, NT, BIG>, lds)));
    attr.store(true, std::memory_order_release);
  }
  constexpr int T = Plan<N>::T;
  const long long nb = (a.inner + T - 1) / T;
  hipLaunchKernelGGL((k_ch_xfused<N, ORDER, PRE, SPEC_C, NT, BIG>), dim3((unsigned)nb), dim3(Plan<N>::NT), lds, ctx->stream, a, tw);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

// prefetch depth of the first-order history: half a line for the two-stage plans; the three-stage plans (512 points and more) hold
// more twiddle / index state across the transforms and spill 7-14 VGPRs with that (tools/spill_census.sh).  Round 4, 512^3 on one
// GPU, same box, interleaved: depth 8 instead of 4 changes nothing (fused x pass 1141 / 1141 vs 1144 / 1157 us) -- the pass runs at
// the rate of its memory pattern (tools/xfused_probe.hip: with the three transforms removed it takes the same 1.06-1.10 ms)
#ifndef MRL_XFUSED_PRE3_DIV
#define MRL_XFUSED_PRE3_DIV 4
#endif
template <int N, int ORDER, bool SPEC_C, int PRE = (Plan<N>::ns >= 3 ? Plan<N>::P / MRL_XFUSED_PRE3_DIV : Plan<N>::P / 2)>
static int launch_xfused(mrl_ctx *ctx, const FusedArgs &a, const kcplx *tw) {
  if ((double)sizeof(kcplx) * (double)N * (double)a.plane >= 4294967296.0) {   // one array is 4 GiB or more: 64-bit uniform part of the offsets
    if constexpr (big_capable<N>() && !SPEC_C) {
      return launch_xfused_v<N, ORDER, SPEC_C, nt_capable<N>(), PRE, true>(ctx, a, tw);
    } else {
      return MRL_ERR_UNSUPPORTED;
    }
  }
  if constexpr (nt_capable<N>()) {
    const kreal array_bytes = (double)sizeof(kcplx) * (double)N * (double)a.plane;
    if (array_bytes >= 96.0e6) return launch_xfused_v<N, ORDER, SPEC_C, true, PRE>(ctx, a, tw);
  }
  return launch_xfused_v<N, ORDER, SPEC_C, false, PRE>(ctx, a, tw);
}

}  // namespace MRL_P2NS

}  // namespace mrl
