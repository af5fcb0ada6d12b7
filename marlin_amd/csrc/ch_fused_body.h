// Body of the fused Cahn-Hilliard k-space pass, shared by the serial kernel (lines along x, ch_fused.hip)
// and the slab kernel (lines along y on the exchange layout, slab_fused.hip):
//   forward transform of mu-hat and c-hat along the line axis, Nhat = Mbar*mu-hat (stored: it is the
//   history), ABM predictor with ORDER old Nhat, 1/(1 - dt*Lbar), inverse transform of ubar.
//   (AdamsBashforthMoulton.C:94-101, ReciprocalLaplacianFactor.C:28-31, ReciprocalLaplacianSquareFactor.C:28-32)
//
// Memory-latency structure: every HBM load of the workgroup is issued as early as its registers allow --
// the small twiddle / k-axis loads first (vmcnt retires in order), then the 16 mu-hat and the 16 c-hat
// values of the thread, and the old Nhat values right after the Nhat stores -- so the three transforms run
// while the next operands are in flight instead of paying one full HBM latency per phase.
#pragma once
#include <type_traits>

#include "fft_pow2.h"


namespace mrl {
namespace MRL_P2NS {

struct FusedCommon {
  const kcplx *chat;   // c-hat, work layout
  const kcplx *muhat;  // mu-hat, work layout
  kcplx *ubar;         // out, work layout (may alias chat)
  kcplx *Nnew;         // out, dense reference layout
  kcplx *cbar;         // optional out, dense: c-hat of this substep (the reference's cbar buffer)
  kcplx *carry;        // spectral carry-over (dense): SPEC_C -> in: c-hat of this substep, out: ubar = c-hat of the next one;
                      // otherwise optional out (ubar), which bootstraps the carry-over
  const kcplx *Nold[4];
  kreal coef[5];     // sub_dt * beta[order][i]
  kreal M, kappa, dt;
};

// LINE_IS_X: the line axis is x (k^2 = (kl^2 + ka^2) + kb^2 with ka = ky, kb = kz); otherwise the line
// axis is y (k^2 = (ka^2 + kl^2) + kb^2 with ka = kx, kb = kz) -- the reference's association kx*kx + ky*ky + kz*kz.
// 32-bit BYTE offsets from a wave-uniform base pointer: the loads / stores take the "SGPR base + 32-bit VGPR
// offset" form, which halves the address registers of the 5 x 16 accesses (arrays of the fast path are < 4 GiB).
__device__ __forceinline__ kcplx ldc(const kcplx *base, unsigned boff) {
  return *reinterpret_cast<const kcplx *>(reinterpret_cast<const char *>(base) + boff);
}
__device__ __forceinline__ void stc(kcplx *base, unsigned boff, kcplx v) {
  *reinterpret_cast<kcplx *>(reinterpret_cast<char *>(base) + boff) = v;
}
// streaming (non-temporal) forms for the arrays that are not touched again within the substep (old / new Nhat): they
// should not displace the work arrays, which the next pass re-reads, from the 256 MB Infinity Cache
typedef kreal nt_v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ kcplx ldc_nt(const kcplx *base, unsigned boff) {
  const nt_v2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_v2 *>(reinterpret_cast<const char *>(base) + boff));
  return mkc(v.x, v.y);
}
__device__ __forceinline__ void stc_nt(kcplx *base, unsigned boff, kcplx v) {
  nt_v2 w;
  w.x = v.x;
  w.y = v.y;
  __builtin_nontemporal_store(w, reinterpret_cast<nt_v2 *>(reinterpret_cast<char *>(base) + boff));
}

// Arrays of 4 GiB and more (1024^3 on one 288 GB GPU): an offset is a WAVE-UNIFORM 64-bit part (m * line stride: folded into the
// scalar base address) plus a 32-bit per-lane part, so the accesses keep the "SGPR base + 32-bit VGPR offset" form
struct BigOff {
  unsigned long long uni;
  unsigned lane;
};
__device__ __forceinline__ kcplx ldc(const kcplx *base, BigOff o) {
  return *reinterpret_cast<const kcplx *>(reinterpret_cast<const char *>(base) + o.uni + o.lane);
}
__device__ __forceinline__ void stc(kcplx *base, BigOff o, kcplx v) {
  *reinterpret_cast<kcplx *>(reinterpret_cast<char *>(base) + o.uni + o.lane) = v;
}
__device__ __forceinline__ kcplx ldc_nt(const kcplx *base, BigOff o) {
  const nt_v2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_v2 *>(reinterpret_cast<const char *>(base) + o.uni + o.lane));
  return mkc(v.x, v.y);
}
__device__ __forceinline__ void stc_nt(kcplx *base, BigOff o, kcplx v) {
  nt_v2 w;
  w.x = v.x;
  w.y = v.y;
  __builtin_nontemporal_store(w, reinterpret_cast<nt_v2 *>(reinterpret_cast<char *>(base) + o.uni + o.lane));
}

// OffW / OffD: callables m -> byte offset of the thread's m-th line element in the work layout / the dense
// reference layout (computed from a few live values instead of 2 x 16 held registers).
// StU: callable (m, value) that stores the thread's m-th element of the ubar output (work layout on one GPU; on the slab path
// the inverse exchange layout, scattered through a per-destination pointer table).
// SPEC_C: c-hat is not transformed from the work layout but read, already in reciprocal space, from a.carry (dense), which
// receives ubar in place: irfftn followed by rfftn is the identity up to rounding, so the next substep's c-hat IS this ubar.
// The old / new Nhat arrays (and the optional cbar output) are accessed non-temporally: they are not touched again within the
// substep and must not displace the work arrays from the Infinity Cache (measured at 256^3: fused x pass 155 -> 128 us, the
// following y pass 54.5 -> 49 us).  NT_W: the same for the mu-hat loads, which are dead after this pass (another 2.5 % of the substep).
// NT_CARRY: the same for the carried spectrum (read and rewritten once per substep).
// NT_HIST: ... for the old / new Nhat arrays and the cbar output.
// OffM: offset callable of the mu-hat loads where they differ from the c-hat ones (the table-driven slab pass: the second field of a
// received chunk lies a chunk-dependent distance behind the first); the other callers pass OffSame{} and the body uses `offw` (a
// second copy of the same callable cost the 512-point slab kernel 33 spilled VGPRs: 80 MB of scratch traffic per launch).
struct OffSame {};
template <int N, int ORDER, bool LINE_IS_X, int PRE, bool SPEC_C, bool NT_W, bool NT_CARRY, bool NT_HIST, class OffW,
          class OffM, class OffD, class StU>
__device__ __forceinline__ void ch_fused_body(const FusedCommon &a, const kcplx *__restrict__ tw,
                                              const kreal *__restrict__ kline, const kreal *__restrict__ ka_ptr,
                                              const kreal *__restrict__ kb_ptr, bool valid, int q, int l,
                                              OffW offw, OffM offm, OffD offd, StU stu, kcplx *W, kcplx *X,
                                              kreal *KL) {
#pragma clang fp contract(off)
  constexpr int P = Plan<N>::P, TPL = Plan<N>::TPL, NT = Plan<N>::NT, CNT = (N + NT - 1) / NT;
  static_assert(PRE <= P, "prefetch depth");
  using Map = MapStrided<N>;

  // ---- issue every early load: small ones first
  kcplx twv[CNT];
  kreal klv[CNT];
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    const int idx = threadIdx.x + j * NT;
    twv[j] = idx < N ? tw[idx] : mkc(0.0, 0.0);
    klv[j] = idx < N ? kline[idx] : 0.0;
  }
  // Loads are unconditional (the caller clamps the offsets of out-of-range lanes to a valid element; their
  // results are never stored): a branch around them would make hipcc's vmcnt bookkeeping fall back to vmcnt(0).
  const kreal ka = *ka_ptr, kb = *kb_ptr;
  kcplx v[P], cp[P];
#pragma unroll
  for (int m = 0; m < P; ++m) {
    if constexpr (std::is_same<OffM, OffSame>::value)
      v[m] = NT_W ? ldc_nt(a.muhat, offw(m)) : ldc(a.muhat, offw(m));
    else
      v[m] = NT_W ? ldc_nt(a.muhat, offm(m)) : ldc(a.muhat, offm(m));
  }
#pragma unroll
  for (int m = 0; m < P; ++m) cp[m] = SPEC_C ? (NT_CARRY ? ldc_nt(a.carry, offd(m)) : ldc(a.carry, offd(m))) : ldc(a.chat, offw(m));
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    const int idx = threadIdx.x + j * NT;
    if (idx < N) {
      W[idx] = twv[j];
      KL[idx] = klv[j];
    }
  }
  const kreal ka2 = ka * ka, kb2 = kb * kb;

  // ---- 1. mu-hat: forward transform (its first exchange publishes W and KL to the workgroup)
  fft_line<N, Map>(v, q, l, X, W);

  // ---- 2. Nhat = Mbar * mu-hat, Mbar = -k^2 * M
  kcplx Nv[P];
#pragma unroll
  for (int m = 0; m < P; ++m) {
    const kreal kl = KL[q + m * TPL];
    const kreal k2 = LINE_IS_X ? (kl * kl + ka2) + kb2 : (ka2 + kl * kl) + kb2;
    const kreal Mbar = -k2 * a.M;
    Nv[m] = mkc(Mbar * v[m].x, Mbar * v[m].y);
  }
  if (valid) {
#pragma unroll
    for (int m = 0; m < P; ++m) {
      if (NT_HIST)
        stc_nt(a.Nnew, offd(m), Nv[m]);
      else
        stc(a.Nnew, offd(m), Nv[m]);
    }
  }
  if (SPEC_C) {
    // the carried c-hat needs no transform, so the first term of the update, cbar + (dt b0) N, is formed here and N's 64 registers
    // are free again before the history arrives (the same expression as in step 4, only evaluated earlier)
#pragma unroll
    for (int m = 0; m < P; ++m) {
      cp[m].x = cp[m].x + a.coef[0] * Nv[m].x;
      cp[m].y = cp[m].y + a.coef[0] * Nv[m].y;
    }
  }

  // ---- first-order history: PRE of the 16 old Nhat values are requested before the c-hat transform and are in flight
  //      during it, the rest right after it (PRE is tuned per kernel against the 256-VGPR / two-waves-per-SIMD limit)
  kcplx o1[ORDER == 1 ? P : 1];
  if (ORDER == 1) {
#pragma unroll
    for (int m = 0; m < PRE; ++m) o1[ORDER == 1 ? m : 0] = NT_HIST ? ldc_nt(a.Nold[0], offd(m)) : ldc(a.Nold[0], offd(m));
  }

  // ---- 3. c-hat: forward transform (unless it is carried over in reciprocal space)
  if (!SPEC_C) fft_line<N, Map>(cp, q, l, X, W);
  if (a.cbar && valid) {
#pragma unroll
    for (int m = 0; m < P; ++m) stc_nt(a.cbar, offd(m), cp[m]);
  }

  // ---- 4. ubar = (cbar + (dt b0) N + sum (dt b_i) Nold_i) / (1 - dt*Lbar), the reference's association
  if (ORDER == 1) {
#pragma unroll
    for (int m = PRE; m < P; ++m) o1[ORDER == 1 ? m : 0] = NT_HIST ? ldc_nt(a.Nold[0], offd(m)) : ldc(a.Nold[0], offd(m));
#pragma unroll
    for (int m = 0; m < P; ++m) {
      kcplx u = cp[m];
      if (!SPEC_C) {
        u.x = u.x + a.coef[0] * Nv[m].x;
        u.y = u.y + a.coef[0] * Nv[m].y;
      }
      u.x += a.coef[1] * o1[ORDER == 1 ? m : 0].x;
      u.y += a.coef[1] * o1[ORDER == 1 ? m : 0].y;
      const kreal kl = KL[q + m * TPL];
      const kreal k2 = LINE_IS_X ? (kl * kl + ka2) + kb2 : (ka2 + kl * kl) + kb2;
      const kreal Lb = k2 * k2 * a.kappa;
      const kreal scl = kreal(1.0) / (kreal(1.0) - a.dt * Lb);
      v[m] = mkc(u.y * scl, u.x * scl);  // swapped for the inverse transform
    }
  } else {
    //    (deeper histories: half of the points at a time; a run-time trip count here would make hipcc wait vmcnt(0) per element)
    //    (third and fourth order histories: a quarter at a time, 4 x P/4 old values in flight instead of 4 x P/2 = 128 VGPRs)
    constexpr int H = (ORDER >= 3 && P % 4 == 0) ? P / 4 : P / 2;
#pragma unroll
    for (int half = 0; half < P / H; ++half) {
      kcplx o[ORDER > 0 ? ORDER : 1][H];
#pragma unroll
      for (int h = 0; h < ORDER; ++h) {
#pragma unroll
        for (int j = 0; j < H; ++j) o[h][j] = NT_HIST ? ldc_nt(a.Nold[h], offd(half * H + j)) : ldc(a.Nold[h], offd(half * H + j));
      }
#pragma unroll
      for (int j = 0; j < H; ++j) {
        const int m = half * H + j;
        kcplx u = cp[m];
        if (!SPEC_C) {
          u.x = u.x + a.coef[0] * Nv[m].x;
          u.y = u.y + a.coef[0] * Nv[m].y;
        }
#pragma unroll
        for (int h = 0; h < ORDER; ++h) {
          u.x += a.coef[h + 1] * o[h][j].x;
          u.y += a.coef[h + 1] * o[h][j].y;
        }
        const kreal kl = KL[q + m * TPL];
        const kreal k2 = LINE_IS_X ? (kl * kl + ka2) + kb2 : (ka2 + kl * kl) + kb2;
        const kreal Lb = k2 * k2 * a.kappa;
        const kreal scl = kreal(1.0) / (kreal(1.0) - a.dt * Lb);
        v[m] = mkc(u.y * scl, u.x * scl);
      }
    }
  }

  if ((SPEC_C || a.carry) && valid) {
#pragma unroll
    for (int m = 0; m < P; ++m) {
      if (NT_CARRY)
        stc_nt(a.carry, offd(m), cswap(v[m]));
      else
        stc(a.carry, offd(m), cswap(v[m]));
    }
  }

  // ---- 5. inverse transform (unnormalised; 1/N applied by the final z pass)
  fft_line<N, Map>(v, q, l, X, W);
  if (valid) {
#pragma unroll
    for (int m = 0; m < P; ++m) stu(m, cswap(v[m]));
  }
}

}  // namespace MRL_P2NS
}  // namespace mrl
