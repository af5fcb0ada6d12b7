// Host-side launchers of the power-of-two kernels, shared by ch_fused.hip, slab_fused.hip and mech_fused.hip.
#pragma once
#include <atomic>
#include "fft_pow2_kernels.h"

namespace mrl {
namespace MRL_P2NS {

// twiddle table exp(-2 pi i k / n) of an axis in the scalar type of this translation unit (the fp32 copy is built on first use)
inline const kcplx *tw_table(mrl_ctx *ctx, int axis) {
  if constexpr (sizeof(kreal) == 8) {
    return reinterpret_cast<const kcplx *>(ctx->ax[axis].d_tw);
  } else {
    if (!ctx->ax[axis].d_tw32 && axis_tw32(ctx, axis) != MRL_OK) return nullptr;
    return reinterpret_cast<const kcplx *>(ctx->ax[axis].d_tw32);
  }
}

template <class K>
inline int set_lds_attr(mrl_ctx *ctx, K kernel, size_t lds) {
  if (lds > 64 * 1024) {
    MRL_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
  }
  return MRL_OK;
}

// nlines = number of complex transforms (MODE 0: pairs of real lines; MODE 1: one CH line each)
template <int N, int MODE, int FAM>
inline int launch_z_fwd(mrl_ctx *ctx, const kreal *in, kcplx *o0, kcplx *o1, kreal *mu, const ChDev &chp,
                        long long nlines, ZLay zl = ZLay{0u, 0u}) {
  static std::atomic<bool> attr{false};  // (two host threads may both set the attribute: harmless, and no torn flag)
  constexpr size_t lds = lds_line<N>();
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY(set_lds_attr(ctx, k_z_fwd<N, MODE, FAM>, lds));
    attr.store(true, std::memory_order_release);
  }
  constexpr int LPB = ZPlan<N>::T;
  const long long nb = (nlines + LPB - 1) / LPB;
  hipLaunchKernelGGL((k_z_fwd<N, MODE, FAM>), dim3((unsigned)nb), dim3(ZPlan<N>::NT), lds, ctx->stream, in, o0, o1, mu, chp, nlines,
                     tw_table(ctx, 2), zl);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

template <int N>
inline int launch_z_inv(mrl_ctx *ctx, const kcplx *in, kreal *out, kreal scale, long long nlines, ZLay zl = ZLay{0u, 0u}) {
  static std::atomic<bool> attr{false};  // (two host threads may both set the attribute: harmless, and no torn flag)
  constexpr size_t lds = lds_line<N>();
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY(set_lds_attr(ctx, k_z_inv<N, false>, lds));
    attr.store(true, std::memory_order_release);
  }
  constexpr int LPB = ZPlan<N>::T;
  const long long nb = (nlines + LPB - 1) / LPB;
  hipLaunchKernelGGL((k_z_inv<N, false>), dim3((unsigned)nb), dim3(ZPlan<N>::NT), lds, ctx->stream, in, out, scale, nlines,
                     tw_table(ctx, 2), nullptr, nullptr, zl);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

// nlines = number of line PAIRS (= complex inverse transforms)
template <int N, int FAM, bool MU_ONLY = false>
inline int launch_z_inv_fwd(mrl_ctx *ctx, const kcplx *in, kcplx *o0, kcplx *o1, kreal *mu, const ChDev &chp, kreal scale,
                            long long nlines, ZLay zl = ZLay{0u, 0u}) {
  static std::atomic<bool> attr{false};  // (two host threads may both set the attribute: harmless, and no torn flag)
  constexpr size_t lds = lds_line_ea<N>();
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY((set_lds_attr(ctx, k_z_inv_fwd<N, FAM, MU_ONLY>, lds)));
    attr.store(true, std::memory_order_release);
  }
  constexpr int LPB = ZPlanEA<N>::T;
  const long long nb = (nlines + LPB - 1) / LPB;
  hipLaunchKernelGGL((k_z_inv_fwd<N, FAM, MU_ONLY>), dim3((unsigned)nb), dim3(ZPlanEA<N>::NT), lds, ctx->stream, in, o0, o1, mu, chp, scale,
                     nlines, tw_table(ctx, 2), zl);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

// z inverse that also leaves sum(out * dotv) as one partial per workgroup in `partial`; *nblocks = their number
template <int N>
inline int launch_z_inv_dot(mrl_ctx *ctx, const kcplx *in, kreal *out, kreal scale, long long nlines, const kreal *dotv,
                            kreal *partial, int *nblocks, ZLay zl = ZLay{0u, 0u}, const int *stop = nullptr) {
  static std::atomic<bool> attr{false};  // (two host threads may both set the attribute: harmless, and no torn flag)
  constexpr size_t lds = lds_line<N>();
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY(set_lds_attr(ctx, k_z_inv<N, true>, lds));
    attr.store(true, std::memory_order_release);
  }
  constexpr int LPB = ZPlan<N>::T;
  const long long nb = (nlines + LPB - 1) / LPB;
  hipLaunchKernelGGL((k_z_inv<N, true>), dim3((unsigned)nb), dim3(ZPlan<N>::NT), lds, ctx->stream, in, out, scale, nlines,
                     tw_table(ctx, 2), dotv, partial, zl, stop);
  MRL_HIP(ctx, hipGetLastError());
  *nblocks = (int)nb;
  return MRL_OK;
}

template <int N, bool INV, int NF>
inline int launch_pass_t(mrl_ctx *ctx, PassArgs a, const kcplx *tw) {
  static std::atomic<bool> attr{false};  // (two host threads may both set the attribute: harmless, and no torn flag)
  constexpr size_t lds = lds_strided<N>();
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY(set_lds_attr(ctx, k_pass<N, INV, NF>, lds));
    attr.store(true, std::memory_order_release);
  }
  constexpr int T = Plan<N>::T;
  a.tiles_per_outer = (int)((a.inner + T - 1) / T);
  const long long nb = a.outer * a.tiles_per_outer;
  hipLaunchKernelGGL((k_pass<N, INV, NF>), dim3((unsigned)nb), dim3(Plan<N>::NT), lds, ctx->stream, a, tw);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

template <int N, bool INV, int NF>
inline int launch_pass_sub(mrl_ctx *ctx, SubPassArgs a, const kcplx *tw) {
  static std::atomic<bool> attr{false};  // (two host threads may both set the attribute: harmless, and no torn flag)
  constexpr size_t lds = lds_strided<N>();
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY(set_lds_attr(ctx, k_pass_sub<N, INV, NF>, lds));
    attr.store(true, std::memory_order_release);
  }
  constexpr int T = Plan<N>::T;
  if (a.tcols == 0) a.tcols = a.cols;
  const long long nb = ((long long)a.rows * a.tcols + T - 1) / T;
  if (a.sig.expected == 0) a.sig.expected = (unsigned)nb;  // (a caller that spreads one exchange over several launches sets the total)
  hipLaunchKernelGGL((k_pass_sub<N, INV, NF>), dim3((unsigned)nb), dim3(Plan<N>::NT), lds, ctx->stream, a, tw);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

template <int N, bool INV>
inline int launch_pass_sub_mf(mrl_ctx *ctx, SubPassArgs a, const kcplx *tw, int nf) {
  static std::atomic<bool> attr{false};  // (two host threads may both set the attribute: harmless, and no torn flag)
  constexpr size_t lds = lds_strided<N>();
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY(set_lds_attr(ctx, k_pass_sub_mf<N, INV>, lds));
    attr.store(true, std::memory_order_release);
  }
  constexpr int T = Plan<N>::T;
  if (a.tcols == 0) a.tcols = a.cols;
  a.nb = (unsigned)(((long long)a.rows * a.tcols + T - 1) / T);
  const long long nb = (long long)nf * a.nb;
  if (a.sig.expected == 0) a.sig.expected = (unsigned)nb;
  hipLaunchKernelGGL((k_pass_sub_mf<N, INV>), dim3((unsigned)nb), dim3(Plan<N>::NT), lds, ctx->stream, a, tw);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

}  // namespace MRL_P2NS

// every length with a Plan<N> that the fast paths are instantiated for
#define MRL_SWITCH_N(n, CALL)  \
  switch (n) {                 \
    case 64: { constexpr int NN = 64; CALL; } break;   \
    case 128: { constexpr int NN = 128; CALL; } break; \
    case 256: { constexpr int NN = 256; CALL; } break; \
    case 512: { constexpr int NN = 512; CALL; } break; \
    case 1024: { constexpr int NN = 1024; CALL; } break; \
    case 2048: { constexpr int NN = 2048; CALL; } break; \
    case 4096: { constexpr int NN = 4096; CALL; } break; \
    case 100: { constexpr int NN = 100; CALL; } break; \
    case 200: { constexpr int NN = 200; CALL; } break; \
    case 400: { constexpr int NN = 400; CALL; } break; \
    case 96: { constexpr int NN = 96; CALL; } break;   \
    case 192: { constexpr int NN = 192; CALL; } break; \
    case 384: { constexpr int NN = 384; CALL; } break; \
    case 32: { constexpr int NN = 32; CALL; } break;   \
    case 40: { constexpr int NN = 40; CALL; } break;   \
    case 50: { constexpr int NN = 50; CALL; } break;   \
    case 80: { constexpr int NN = 80; CALL; } break;   \
    case 250: { constexpr int NN = 250; CALL; } break; \
    case 500: { constexpr int NN = 500; CALL; } break; \
    case 1000: { constexpr int NN = 1000; CALL; } break; \
    case 48: { constexpr int NN = 48; CALL; } break;   \
    case 144: { constexpr int NN = 144; CALL; } break; \
    case 768: { constexpr int NN = 768; CALL; } break; \
    default: return MRL_ERR_UNSUPPORTED;               \
  }

// workgroups of one k_pass_sub launch over rows x cols lines
template <int N>
inline unsigned pass_sub_blocks(long long rows, long long cols) {
  constexpr int T = MRL_P2NS::Plan<N>::T;
  return (unsigned)((rows * cols + T - 1) / T);
}

// ... plus the lengths whose plans exist for the PLAIN transforms only (30 points per thread: see fft_pow2.h)
#define MRL_SWITCH_N_PLAIN(n, CALL)                      \
  switch (n) {                                           \
    case 60: { constexpr int NN = 60; CALL; } break;     \
    case 90: { constexpr int NN = 90; CALL; } break;     \
    case 120: { constexpr int NN = 120; CALL; } break;   \
    case 150: { constexpr int NN = 150; CALL; } break;   \
    case 180: { constexpr int NN = 180; CALL; } break;   \
    case 240: { constexpr int NN = 240; CALL; } break;   \
    case 270: { constexpr int NN = 270; CALL; } break;   \
    case 300: { constexpr int NN = 300; CALL; } break;   \
    case 360: { constexpr int NN = 360; CALL; } break;   \
    case 450: { constexpr int NN = 450; CALL; } break;   \
    case 600: { constexpr int NN = 600; CALL; } break;   \
    case 160: { constexpr int NN = 160; CALL; } break;   \
    case 320: { constexpr int NN = 320; CALL; } break;   \
    case 640: { constexpr int NN = 640; CALL; } break;   \
    case 1280: { constexpr int NN = 1280; CALL; } break; \
    case 800: { constexpr int NN = 800; CALL; } break;   \
    case 72: { constexpr int NN = 72; CALL; } break;   \
    case 216: { constexpr int NN = 216; CALL; } break;   \
    case 288: { constexpr int NN = 288; CALL; } break;   \
    case 432: { constexpr int NN = 432; CALL; } break;   \
    case 576: { constexpr int NN = 576; CALL; } break;   \
    case 864: { constexpr int NN = 864; CALL; } break;   \
    case 1152: { constexpr int NN = 1152; CALL; } break;   \
    default: MRL_SWITCH_N(n, CALL)                       \
  }

inline bool plain30_ok(long long n) {
  switch (n) {
    case 60: case 90: case 120: case 150: case 180: case 240: case 270: case 300: case 360: case 450: case 600: return true;
    case 160: case 320: case 640: case 1280: return true;   // (radix-20 plans: plain kernels only, as the radix-30 ones)
    case 800: case 72: case 216: case 288: case 432: case 576: case 864: case 1152: return true;   // (further 2^a 5^2 / 2^a 3^b lengths)
    default: return false;
  }
}

inline bool pow2_ok(long long n) {
  switch (n) {
    case 32: case 64: case 128: case 256: case 512: case 1024: case 2048: case 4096:  // 2^a
    case 40: case 50: case 80: case 100: case 200: case 250: case 400: case 500: case 1000:  // 2^a 5^b
    case 48: case 96: case 144: case 192: case 384: case 768:             // 2^a 3^b
      return true;
    default:
      return false;
  }
}
inline bool is_pow2(long long n) { return n > 0 && (n & (n - 1)) == 0; }
// every length with a plan for the plain transforms
inline bool plain_ok(long long n) { return pow2_ok(n) || plain30_ok(n); }

// Adams-Bashforth coefficients (src/tensor_solver/AdamsBashforthMoulton.C:67-73, incl. the AB5 190/720 entry)
static const double kBetaAB[5][5] = {
    {1.0, 0.0, 0.0, 0.0, 0.0},
    {3.0 / 2.0, -1.0 / 2.0, 0.0, 0.0, 0.0},
    {23.0 / 12.0, -16.0 / 12.0, 5.0 / 12.0, 0.0, 0.0},
    {55.0 / 24.0, -59.0 / 24.0, 37.0 / 24.0, -9.0 / 24.0, 0.0},
    {190.0 / 720.0, -2774.0 / 720.0, 2616.0 / 720.0, -1274.0 / 720.0, 251.0 / 720.0},
};

}  // namespace mrl
