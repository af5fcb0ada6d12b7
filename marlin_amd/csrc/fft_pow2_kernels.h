// Fast-path kernels (3-D, power-of-two extents) built from fft_pow2.h.
//   k_z_fwd      real lines -> half spectra along z; two real sequences per complex transform
//                (CH mode: c and mu=f'(c) of the same line; PAIR mode: two adjacent lines)
//   k_pass       c2c transform along a strided axis (x or y), forward or inverse, NF fields
//   k_z_inv      half spectra -> real lines (two lines per complex transform), scaled
// The fused Cahn-Hilliard x-pass lives in ch_fused.hip.
#pragma once
#include "comm_dev.h"
#include "fft_pow2.h"

namespace mrl {
namespace MRL_P2NS {

struct ChDev {
  int family;
  kreal c0, c1, c2;
  kreal k[8];  // named constants of a parsed free energy (MRL_FE_PARSED)
};

// Solver-private spectral layout of the fused serial path: rows of NZC complex values, `lpp` rows per x plane, and every plane `pad`
// elements longer than lpp * NZC.  The pad makes the stride between consecutive x planes an ODD number of 256-byte pieces: the x
// passes gather 256-byte pieces one plane apart, and with the natural pitch of the power-of-two grids (256^3: 2064 pieces) they fall
// on 8 of the memory channels only (tools/ldsdma_probe.hip: the same bytes move in 67 us with the dense pitch and in 56 us with one
// piece of padding per plane).  pad = 0: dense.
struct ZLay {
  unsigned lpp, pad;
};
__device__ __forceinline__ long long zrow(long long row, int nzc, ZLay z) {
  return row * nzc + (z.pad ? (long long)((unsigned)row / z.lpp) * z.pad : 0ll);
}

// MRL_FE_PARSED: the chemical potential generated from the user's expression (expr.hip); it only exists in the
// run-time compiled (hiprtc) instance of k_z_fwd, where its definition is appended to these headers
__device__ kreal mrl_user_mu(kreal c, const kreal *k);

// FAM is a compile-time constant: a run-time family test inside the unrolled load loop makes hipcc
// branch around every element and wait vmcnt(0) per load (16 dependent HBM round trips).
template <int FAM>
__device__ __forceinline__ kreal mu_eval(const ChDev &p, kreal c) {
#pragma clang fp contract(off)
  if constexpr (FAM == MRL_FE_DOUBLE_WELL) {
    const kreal one = 1.0, two = 2.0;
    const kreal cm1 = c - one;
    return (p.c0 * (two * c)) * (cm1 * cm1) + (p.c0 * (c * c)) * (two * cm1);
  } else if constexpr (FAM == MRL_FE_PFHUB) {
    const kreal a = c - p.c1;
    const kreal b = p.c2 - c;
    const kreal two = 2.0, mone = -1.0;
    return (p.c0 * (two * a)) * (b * b) + (p.c0 * (a * a)) * ((two * b) * mone);
  } else {
    return mrl_user_mu(c, p.k);
  }
}



// ---------------------------------------------------------------------------------------------
// Lane exchange for the k <-> N - k pairing of the z kernels (round 4).  A line is owned by TPL consecutive lanes (register m of
// lane q = position q + m TPL), so position N - k of the line lives in lane (TPL - q) % TPL of the SAME wave, register P - 1 - m (lane
// 0: its own register P - m).  ds_bpermute moves it there without touching LDS memory: the forward kernels lose their natural-order
// copy (one of two LDS round trips per transform at 256 points, one of three at 512), the inverse kernels load every spectral element
// once instead of twice (the mirrored half came through L1 again and cost 64 VGPRs of landing space).  Same operands, same
// operations: bit-identical results.  tools/zpass_probe.hip, same box: k_z_inv_fwd<256> 77.5 -> 72.2 us, <512> 88 -> 85 us.
template <int N>
constexpr bool lane_pair_ok() {
  constexpr int TPL = Plan<N>::TPL;
  // power of two <= 64: a line never straddles two waves.  (Round 4 also laid the 100- / 200-point lines out wave by wave -- 6 / 3 whole
  // lines per wave and 4 spare lanes shadowing the first ones -- to give them the lane exchange: bit-identical, and 200^3 0.167-0.169 ms
  // with and without it, 100^3 0.0393 against 0.0402 ms; not kept.)
  return TPL <= 64 && (TPL & (TPL - 1)) == 0 && Plan<N>::P % 2 == 0;
}
__device__ __forceinline__ double lane_get(double x, int src) {
  union {
    double d;
    int i[2];
  } u;
  u.d = x;
  u.i[0] = __builtin_amdgcn_ds_bpermute(src << 2, u.i[0]);
  u.i[1] = __builtin_amdgcn_ds_bpermute(src << 2, u.i[1]);
  return u.d;
}
__device__ __forceinline__ float lane_get(float x, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, x)));
}
__device__ __forceinline__ kcplx lane_get(kcplx x, int src) { return mkc(lane_get(x.x, src), lane_get(x.y, src)); }
// lane (of the wave) that holds positions N - k of the line of lane-in-line q
template <int N>
__device__ __forceinline__ int lane_partner(int q) {
  constexpr int TPL = Plan<N>::TPL;
  return (int)(threadIdx.x & 63u) - q + ((TPL - q) & (TPL - 1));
}

// v = transform of the packed line x = a + i b (registers q + m TPL): store the half spectra of a (o0) and b (o1), k = 0 .. N/2.
// Every lane of the line must call it (the exchange is wave-wide); `X`, `l` are only used by the LDS fallback (other lengths).
template <int N, class Map>
__device__ __forceinline__ void store_half_spectra(const kcplx (&v)[Plan<N>::P], int q, int l, kcplx *X, bool valid, kcplx *o0, kcplx *o1) {
  constexpr int P = Plan<N>::P, TPL = Plan<N>::TPL;
  if constexpr (lane_pair_ok<N>()) {
    const int partner = lane_partner<N>(q);
#pragma unroll
    for (int m = 0; m < P / 2; ++m) {
      const int k = q + m * TPL;
      const kcplx s = lane_get(v[P - 1 - m], partner);
      const kcplx own = v[(P - m) % P];
      const kcplx xk = v[m];
      const kcplx xn = mkc(q == 0 ? own.x : s.x, q == 0 ? own.y : s.y);
      if (valid) {
        o0[k] = mkc(kreal(0.5) * (xk.x + xn.x), kreal(0.5) * (xk.y - xn.y));
        o1[k] = mkc(kreal(0.5) * (xk.y + xn.y), kreal(-0.5) * (xk.x - xn.x));
      }
    }
    if (q == 0 && valid) {  // the Nyquist bin pairs with itself
      const kcplx xk = v[P / 2];
      o0[N / 2] = mkc(kreal(0.5) * (xk.x + xk.x), kreal(0.5) * (xk.y - xk.y));
      o1[N / 2] = mkc(kreal(0.5) * (xk.y + xk.y), kreal(-0.5) * (xk.x - xk.x));
    }
  } else {
    // natural-order copy in LDS for the k <-> N-k pairing
    __syncthreads();
#pragma unroll
    for (int m = 0; m < P; ++m) X[Map::at(q + m * TPL, l)] = v[m];
    __syncthreads();
    if (!valid) return;
#pragma unroll
    for (int m = 0; m <= P / 2; ++m) {
      const int k = q + m * TPL;
      if (k > N / 2) break;  // (only q = 0 owns the Nyquist bin)
      const kcplx xk = v[m];
      const kcplx xn = X[Map::at(k == 0 ? 0 : N - k, l)];
      o0[k] = mkc(kreal(0.5) * (xk.x + xn.x), kreal(0.5) * (xk.y - xn.y));
      o1[k] = mkc(kreal(0.5) * (xk.y + xn.y), kreal(-0.5) * (xk.x - xn.x));
    }
  }
}

// The inverse of it on the load side: half spectra A, B of two real lines -> v = swap(X), X[p] = A[p] + i B[p] for p <= N/2 and
// conj(A[N-p]) + i conj(B[N-p]) beyond (the swap turns the forward transform that follows into the inverse one); the twiddle table
// is committed to LDS while the loads are in flight.
template <int N, int NTH>
__device__ __forceinline__ void load_half_spectra(kcplx (&v)[Plan<N>::P], int q, const kcplx *A, const kcplx *B, const TwRegs<N, NTH> &twr,
                                                  kcplx *W) {
  constexpr int P = Plan<N>::P, TPL = Plan<N>::TPL;
  if constexpr (lane_pair_ok<N>()) {
    const int partner = lane_partner<N>(q);
    kcplx av[P / 2], bv[P / 2];
#pragma unroll
    for (int m = 0; m < P / 2; ++m) {
      av[m] = A[q + m * TPL];
      bv[m] = B[q + m * TPL];
    }
    kcplx aN = A[q == 0 ? N / 2 : q], bN = B[q == 0 ? N / 2 : q];  // Nyquist bin: lane 0 (the others re-read an element they hold)
    tw_commit<N>(twr, W);
    kcplx yh[P / 2];
#pragma unroll
    for (int m = 0; m < P / 2; ++m) {
      kcplx a = av[m], b = bv[m];
      // for position N - k (k = q + m TPL, never 0 or N/2 in the lanes that use it): conj(A[k]) + i conj(B[k])
      yh[m] = cswap(mkc(a.x + b.y, b.x - a.y));
      if (m == 0) {  // k = 0 (lane 0): c2r ignores the imaginary part of the self-conjugate bins
        a.y = q == 0 ? kreal(0.0) : a.y;
        b.y = q == 0 ? kreal(0.0) : b.y;
      }
      v[m] = cswap(mkc(a.x - b.y, a.y + b.x));
    }
    aN.y = 0.0;
    bN.y = 0.0;
    const kcplx vN = cswap(mkc(aN.x - bN.y, aN.y + bN.x));
#pragma unroll
    for (int m = P / 2; m < P; ++m) {
      // position p = q + m TPL > N/2 is held by the partner lane as yh[P - 1 - m]; lane 0: p = m TPL, its own yh[P - m] (m = P/2: Nyquist)
      const kcplx s = lane_get(yh[P - 1 - m], partner);
      const kcplx own = (m == P / 2) ? vN : yh[(P - m) % (P / 2)];
      v[m] = mkc(q == 0 ? own.x : s.x, q == 0 ? own.y : s.y);
    }
  } else {
    kcplx av[P], bv[P];
#pragma unroll
    for (int m = 0; m < P; ++m) {
      const int p = q + m * TPL;
      const int k = (p <= N / 2) ? p : N - p;
      av[m] = A[k];
      bv[m] = B[k];
    }
    tw_commit<N>(twr, W);
#pragma unroll
    for (int m = 0; m < P; ++m) {
      const int p = q + m * TPL;
      const bool lo = p <= N / 2;
      const int k = lo ? p : N - p;
      kcplx a = av[m], b = bv[m];
      if (k == 0 || k == N / 2) {  // c2r ignores the imaginary part of the self-conjugate bins
        a.y = 0.0;
        b.y = 0.0;
      }
      // X[p] = A + iB (p <= N/2), conj(A[k]) + i conj(B[k]) otherwise ; then swap for the inverse
      const kcplx x = lo ? mkc(a.x - b.y, a.y + b.x) : mkc(a.x + b.y, b.x - a.y);
      v[m] = cswap(x);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// z forward.  MODE 0 (PAIR): rows 2L, 2L+1 of `in` -> rows 2L, 2L+1 of out0.
//             MODE 1 (CH)  : row L of `in` (=c) -> row L of out0 (c-hat_z) and out1 (mu-hat_z);
//                            optionally writes mu to mu_out.
//             MODE 2 (MU)  : rows 2L, 2L+1 of `in` (=c) -> rows 2L, 2L+1 of out0 = transform of mu = f'(c) only
//                            (the spectral carry-over pipeline, where c-hat is not recomputed); optional mu_out.
// nlines = number of complex transforms.
template <int N, int MODE, int FAM>
__global__ void __launch_bounds__(ZPlan<N>::NT, 2) k_z_fwd(const kreal *__restrict__ in, kcplx *__restrict__ out0,
                                               kcplx *__restrict__ out1, kreal *__restrict__ mu_out, ChDev chp,
                                               long long nlines, const kcplx *__restrict__ tw, ZLay zl) {
  constexpr int P = Plan<N>::P, TPL = Plan<N>::TPL, LPB = ZPlan<N>::T, NZC = N / 2 + 1;
  using Map = MapLine<N>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  kcplx *W = reinterpret_cast<kcplx *>(smem);
  kcplx *X = W + N;
  const int q = threadIdx.x % TPL, l = threadIdx.x / TPL;
  const long long L = (long long)xcd_remap(blockIdx.x, gridDim.x) * LPB + l;
  const bool valid = L < nlines;
  TwRegs<N, ZPlan<N>::NT> twr;
  tw_issue_staged<N>(twr, tw);

  kcplx v[P];
  const long long Lc = valid ? L : 0;  // out-of-range lanes transform line 0 again and store nothing
  const long long r0 = (MODE == 1) ? Lc : 2 * Lc;
  {
    const kreal *p0 = in + r0 * N + q;
    kreal a[P], b[P];
#pragma unroll
    for (int m = 0; m < P; ++m) a[m] = p0[m * TPL];
    if (MODE != 1) {
#pragma unroll
      for (int m = 0; m < P; ++m) b[m] = p0[N + m * TPL];
    }
    tw_commit<N>(twr, W);
    if (MODE == 1) {
#pragma unroll
      for (int m = 0; m < P; ++m) b[m] = mu_eval<FAM>(chp, a[m]);
    }
    if (MODE == 2) {
#pragma unroll
      for (int m = 0; m < P; ++m) {
        a[m] = mu_eval<FAM>(chp, a[m]);
        b[m] = mu_eval<FAM>(chp, b[m]);
      }
    }
#pragma unroll
    for (int m = 0; m < P; ++m) v[m] = mkc(a[m], b[m]);
    if (MODE == 1 && mu_out && valid) {
      kreal *pm = mu_out + r0 * N + q;
#pragma unroll
      for (int m = 0; m < P; ++m) pm[m * TPL] = v[m].y;
    }
    if (MODE == 2 && mu_out && valid) {
      kreal *pm = mu_out + r0 * N + q;
#pragma unroll
      for (int m = 0; m < P; ++m) {
        pm[m * TPL] = v[m].x;
        pm[N + m * TPL] = v[m].y;
      }
    }
  }
  fft_line<N, Map>(v, q, l, X, W);
  kcplx *o0 = (MODE != 1) ? out0 + zrow(2 * Lc, NZC, zl) : out0 + zrow(Lc, NZC, zl);
  kcplx *o1 = (MODE != 1) ? out0 + zrow(2 * Lc + 1, NZC, zl) : out1 + zrow(Lc, NZC, zl);
  store_half_spectra<N, Map>(v, q, l, X, valid, o0, o1);
}

// ---------------------------------------------------------------------------------------------
// z inverse (PAIR): rows 2L, 2L+1 of the half spectrum `in` -> real rows 2L, 2L+1 of out, * scale.
// DOT: additionally accumulates sum(out * dotv) over the rows written (one partial per workgroup, deterministic): the
// p.Ap of the conjugate-gradient iteration, taken while Ap is still in registers instead of re-reading it from HBM.
template <int N, bool DOT = false>
__global__ void __launch_bounds__(ZPlan<N>::NT, 2) k_z_inv(const kcplx *__restrict__ in, kreal *__restrict__ out, kreal scale,
                                               long long nlines, const kcplx *__restrict__ tw,
                                               const kreal *__restrict__ dotv = nullptr, kreal *__restrict__ partial = nullptr,
                                               ZLay zl = ZLay{0u, 0u}, const int *__restrict__ stop = nullptr) {
  constexpr int P = Plan<N>::P, TPL = Plan<N>::TPL, LPB = ZPlan<N>::T, NZC = N / 2 + 1;
  if (stop && *stop) return;  // see PassArgs::stop
  using Map = MapLine<N>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  kcplx *W = reinterpret_cast<kcplx *>(smem);
  kcplx *X = W + N;
  const int q = threadIdx.x % TPL, l = threadIdx.x / TPL;
  const long long L = (long long)xcd_remap(blockIdx.x, gridDim.x) * LPB + l;
  const bool valid = L < nlines;
  TwRegs<N, ZPlan<N>::NT> twr;
  tw_issue_staged<N>(twr, tw);
  kcplx v[P];
  load_half_spectra<N>(v, q, in + zrow(2 * (valid ? L : 0), NZC, zl), in + zrow(2 * (valid ? L : 0) + 1, NZC, zl), twr, W);
  fft_line<N, Map>(v, q, l, X, W);
  kreal acc = 0.0;
  if (valid) {
    kreal *o0 = out + (2 * L) * N + q;
    kreal pa[DOT ? P : 1], pb[DOT ? P : 1];
    if (DOT) {
      const kreal *d0 = dotv + (2 * L) * N + q;
#pragma unroll
      for (int m = 0; m < P; ++m) {
        pa[DOT ? m : 0] = d0[m * TPL];
        pb[DOT ? m : 0] = d0[N + m * TPL];
      }
    }
#pragma unroll
    for (int m = 0; m < P; ++m) {
      // swap back: real part (row 2L) = v.y, imaginary part (row 2L+1) = v.x
      const kreal ra = v[m].y * scale, rb = v[m].x * scale;
      o0[m * TPL] = ra;
      o0[N + m * TPL] = rb;
      if (DOT) acc += ra * pa[DOT ? m : 0] + rb * pb[DOT ? m : 0];
    }
  }
  if (DOT) {
    // workgroup sum through LDS (the exchange tile is free again); NT need not be a multiple of 64
    constexpr int NT = ZPlan<N>::NT;
    kreal *S = reinterpret_cast<kreal *>(X);
    __syncthreads();
    S[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < 64) {
      kreal s = 0.0;
      for (int i = threadIdx.x; i < NT; i += 64) s += S[i];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
      if (threadIdx.x == 0) partial[blockIdx.x] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// z inverse of substep n fused with the z forward (CH mode) of substep n + 1: rows 2L, 2L+1 of the half spectrum `in`
// (ubar after the inverse x and y passes) -> the two real lines c = irfft(.) * scale stay in registers -> mu = f'(c) ->
// each line packed as c + i*mu and transformed forward -> rows 2L, 2L+1 of out0 (c-hat_z) and out1 (mu-hat_z).
// Between two substeps of one TensorSolver::computeBuffer call the real field c is neither written nor read again (the
// reference rebinds the buffer every substep and only the last one is visible outside the solver): 268 MB of the 673 MB that
// the two separate passes move at 256^3.  in == out0 is allowed (a workgroup reads its rows before it writes them).
// Same arithmetic, in the same order, as k_z_inv followed by k_z_fwd<CH>: bit-identical fields.
// MU_ONLY (the spectral carry-over of the slab pipeline, where c-hat is not recomputed): the two lines of mu are packed into
// ONE forward transform -> rows 2L, 2L+1 of out0 = mu-hat_z; out1 unused.
template <int N, int FAM, bool MU_ONLY = false>
__global__ void __launch_bounds__(ZPlanEA<N>::NT, 2) k_z_inv_fwd(const kcplx *__restrict__ in, kcplx *__restrict__ out0,
                                                                  kcplx *__restrict__ out1, kreal *__restrict__ mu_out, ChDev chp,
                                                                  kreal scale, long long nlines, const kcplx *__restrict__ tw, ZLay zl) {
  constexpr int P = Plan<N>::P, TPL = Plan<N>::TPL, LPB = ZPlanEA<N>::T, NZC = N / 2 + 1;
  using Map = MapLine<N>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  kcplx *W = reinterpret_cast<kcplx *>(smem);
  kcplx *X = W + N;
  const int q = threadIdx.x % TPL, l = threadIdx.x / TPL;
  const long long L = (long long)xcd_remap(blockIdx.x, gridDim.x) * LPB + l;
  const bool valid = L < nlines;
  TwRegs<N, ZPlanEA<N>::NT> twr;
  tw_issue_staged<N>(twr, tw);
  const long long Lc = valid ? L : 0;  // out-of-range lanes transform line pair 0 again and store nothing
  kcplx v[P];
  load_half_spectra<N>(v, q, in + zrow(2 * Lc, NZC, zl), in + zrow(2 * Lc + 1, NZC, zl), twr, W);
  fft_line<N, Map>(v, q, l, X, W);
  if constexpr (MU_ONLY) {
#pragma unroll
    for (int m = 0; m < P; ++m) v[m] = mkc(mu_eval<FAM>(chp, v[m].y * scale), mu_eval<FAM>(chp, v[m].x * scale));
    if (mu_out && valid) {
      kreal *pm = mu_out + (2 * L) * N + q;
#pragma unroll
      for (int m = 0; m < P; ++m) {
        pm[m * TPL] = v[m].x;
        pm[N + m * TPL] = v[m].y;
      }
    }
    fft_line<N, Map>(v, q, l, X, W);
    store_half_spectra<N, Map>(v, q, l, X, valid, out0 + zrow(2 * Lc, NZC, zl), out0 + zrow(2 * Lc + 1, NZC, zl));
    return;
  }
  kreal cb[P];  // second line (row 2L+1), kept while the first one is transformed
  {
#pragma unroll
    for (int m = 0; m < P; ++m) {
      const kreal ca = v[m].y * scale;
      cb[m] = v[m].x * scale;
      v[m] = mkc(ca, mu_eval<FAM>(chp, ca));
    }
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half == 1) {
#pragma unroll
      for (int m = 0; m < P; ++m) v[m] = mkc(cb[m], mu_eval<FAM>(chp, cb[m]));
    }
    if (mu_out && valid) {
      kreal *pm = mu_out + (2 * L + half) * N + q;
#pragma unroll
      for (int m = 0; m < P; ++m) pm[m * TPL] = v[m].y;
    }
    fft_line<N, Map>(v, q, l, X, W);
    store_half_spectra<N, Map>(v, q, l, X, valid, out0 + zrow(2 * Lc + half, NZC, zl), out1 + zrow(2 * Lc + half, NZC, zl));
  }
}

// ---------------------------------------------------------------------------------------------
// strided c2c pass.  Arrays are [outer][N][inner] complex (inner contiguous); a workgroup owns
// T = 4096/N consecutive `inner` positions of one `outer` slice for all N points of the axis.
struct PassArgs {
  const kcplx *in[2];
  kcplx *out[2];
  long long inner;           // contiguous extent
  long long outer;           // number of outer slices
  long long so_in, so_out;   // outer strides (elements)
  long long sn_in, sn_out;   // stride between successive points of the axis
  int tiles_per_outer;
  kreal scale;
  int reverse;  // traverse tiles in descending order: start where the producer kernel ended (Infinity Cache reuse)
  const int *stop;  // optional device word: non-zero = this launch has nothing to do (a conjugate-gradient solve that converged while
                    // the next iteration was already enqueued, mech.hip); nullptr everywhere else
};

template <int N, bool INV, int NF>
__global__ void __launch_bounds__(Plan<N>::NT, 2) k_pass(PassArgs a, const kcplx *__restrict__ tw) {
  constexpr int P = Plan<N>::P, TPL = Plan<N>::TPL, T = Plan<N>::T;
  using Map = MapStrided<N>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  kcplx *W = reinterpret_cast<kcplx *>(smem);
  kcplx *X = W + N;
  if (a.stop && *a.stop) return;  // (wave-uniform scalar load; nullptr on the Cahn-Hilliard path)
  const int l = threadIdx.x % T, q = threadIdx.x / T;
  const unsigned logical = a.reverse ? xcd_remap_rev(blockIdx.x, gridDim.x) : xcd_remap(blockIdx.x, gridDim.x);
  const long long o = logical / a.tiles_per_outer;
  const long long i = (long long)(logical % a.tiles_per_outer) * T + l;
  const bool valid = i < a.inner;
  TwRegs<N> twr;
  tw_issue<N>(twr, tw);
  // all fields' operands are requested up front: field 1 is in flight while field 0 is transformed
  // (unconditional loads from a clamped position: a branch here degrades hipcc's vmcnt counting to vmcnt(0))
  const long long ic = valid ? i : 0;
  kcplx v[NF][P];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const kcplx *p = a.in[f] + o * a.so_in + ic + (long long)q * a.sn_in;
#pragma unroll
    for (int m = 0; m < P; ++m) v[f][m] = p[(long long)m * TPL * a.sn_in];
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
      kcplx *p = a.out[f] + o * a.so_out + i + (long long)q * a.sn_out;
#pragma unroll
      for (int m = 0; m < P; ++m) p[(long long)m * TPL * a.sn_out] = INV ? cswap(v[f][m]) : v[f][m];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// strided c2c pass on a kz SUB-BLOCK of the slab pipeline.  The contiguous index is 2-D (row, col) with
// separate row pitches for input and output, and the axis index n may be chunked by destination / source
// rank:  element(n, row, col) = base + (n >> sh)*cs + (n & ((1<<sh)-1))*sn + row*pitch + col.
// (sh = 31: plain stride.)  Offsets are 32-bit element counts (fast-path arrays are < 2^31 elements).
// Forward passes (INV = false) write the exchange layout: chunk c = n >> sh_out belongs to destination rank c and
// starts at otab[c] -- an address in rank c's receive buffer (peer stores over xGMI, transport PEER_STORE) or in the local
// send buffer; field f of a chunk lies fs_out elements further.  After its last store every workgroup counts itself and the
// last one raises the arrival flags (comm_dev.h) when `sig` asks for it.
struct SubPassArgs {
  const kcplx *in[2];
  kcplx *out[2];              // INV only (dense output)
  kcplx *const *otab;         // !INV: destination of chunk c (device table, one entry per rank)
  int rows, cols;
  int tcols;                 // tiles run over rows x tcols (0 = cols).  !INV with tcols = pitch_out: a tile is T consecutive elements
                             // of the padded OUTPUT rows, i.e. every store of a wave is whole 128-byte lines (columns >= cols repeat
                             // the last line and land in the padding); the loads straddle lines instead, which costs nothing
  unsigned pitch_in, pitch_out;
  unsigned sn_in, sn_out;
  int sh_in, sh_out;
  unsigned cs_in;
  unsigned fs_out, fo_out;   // !INV: element offset of field f within a chunk = fo_out + f * fs_out
  int nt_out;                // !INV: non-temporal stores (experiment)
  unsigned nb;               // k_pass_sub_w / k_pass_sub_mf: workgroups per field (set by the launcher)
  unsigned fdense;           // k_pass_sub_mf: elements between two fields of the dense (rank-local) array
  SignalArgs sig;
};

template <int N, bool INV, int NF>
__global__ void __launch_bounds__(Plan<N>::NT, 2) k_pass_sub(SubPassArgs a, const kcplx *__restrict__ tw) {
  constexpr int P = Plan<N>::P, TPL = Plan<N>::TPL, T = Plan<N>::T;
  using Map = MapStrided<N>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  kcplx *W = reinterpret_cast<kcplx *>(smem);
  kcplx *X = W + N;
  const int l = threadIdx.x % T, q = threadIdx.x / T;
  const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned i = logical * T + l;
  const bool valid = i < (unsigned)(a.rows * a.tcols);
  const unsigned ic = valid ? i : 0u;
  const unsigned row = ic / (unsigned)a.tcols, col = ic - row * (unsigned)a.tcols;
  const unsigned bi = row * a.pitch_in + min(col, (unsigned)a.cols - 1u), bo = row * a.pitch_out + col;
  const unsigned mi = (a.sh_in < 31) ? ((1u << a.sh_in) - 1u) : 0xffffffffu;
  const unsigned mo = (a.sh_out < 31) ? ((1u << a.sh_out) - 1u) : 0xffffffffu;
  TwRegs<N> twr;
  tw_issue<N>(twr, tw);
  kcplx v[NF][P];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
#pragma unroll
    for (int m = 0; m < P; ++m) {
      const unsigned n = q + m * TPL;
      v[f][m] = a.in[f][bi + (a.sh_in < 31 ? (n >> a.sh_in) * a.cs_in : 0u) + (n & mi) * a.sn_in];
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
          a.out[f][bo + (n & mo) * a.sn_out] = cswap(v[f][m]);
        } else {
          kcplx *base = a.otab[n >> a.sh_out];
          kcplx *dst = base + (a.fo_out + (unsigned)f * a.fs_out + bo + (n & mo) * a.sn_out);
          if (a.nt_out)
            st_nt(dst, v[f][m]);
          else
            *dst = v[f][m];
        }
      }
    }
  }
  if (!INV) signal_tail(a.sig);
}

// k_pass_sub over SEVERAL fields in one launch (the nine fields of the slab Gamma operator): gridDim = nf * a.nb, the field is the
// slow block index.  Field f is a.in[0] + f * fdense (dense side) and chunk offset f * fs_out (exchange-layout side); the chunk of
// one peer holds all nf fields (cs_in = nf * fs_out on the inverse side).
template <int N, bool INV>
__global__ void __launch_bounds__(Plan<N>::NT, 2) k_pass_sub_mf(SubPassArgs a, const kcplx *__restrict__ tw) {
  constexpr int P = Plan<N>::P, TPL = Plan<N>::TPL, T = Plan<N>::T;
  using Map = MapStrided<N>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  kcplx *W = reinterpret_cast<kcplx *>(smem);
  kcplx *X = W + N;
  const int l = threadIdx.x % T, q = threadIdx.x / T;
  unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned f = logical / a.nb;
  logical -= f * a.nb;
  const unsigned i = logical * T + l;
  const bool valid = i < (unsigned)(a.rows * a.tcols);
  const unsigned ic = valid ? i : 0u;
  const unsigned row = ic / (unsigned)a.tcols, col = ic - row * (unsigned)a.tcols;
  const unsigned bi = row * a.pitch_in + min(col, (unsigned)a.cols - 1u), bo = row * a.pitch_out + col;
  const unsigned mi = (a.sh_in < 31) ? ((1u << a.sh_in) - 1u) : 0xffffffffu;
  const unsigned mo = (a.sh_out < 31) ? ((1u << a.sh_out) - 1u) : 0xffffffffu;
  const kcplx *__restrict__ src = a.in[0] + (size_t)f * (INV ? a.fs_out : a.fdense);
  TwRegs<N> twr;
  tw_issue<N>(twr, tw);
  kcplx v[P];
#pragma unroll
  for (int m = 0; m < P; ++m) {
    const unsigned n = q + m * TPL;
    v[m] = src[bi + (a.sh_in < 31 ? (n >> a.sh_in) * a.cs_in : 0u) + (n & mi) * a.sn_in];
  }
  tw_commit<N>(twr, W);
  if (INV) {
#pragma unroll
    for (int m = 0; m < P; ++m) v[m] = cswap(v[m]);
  }
  fft_line<N, Map>(v, q, l, X, W);
  if (valid) {
    kcplx *dense = INV ? a.out[0] + (size_t)f * a.fdense : nullptr;
#pragma unroll
    for (int m = 0; m < P; ++m) {
      const unsigned n = q + m * TPL;
      if (INV) {
        dense[bo + (n & mo) * a.sn_out] = cswap(v[m]);
      } else {
        kcplx *base = a.otab[n >> a.sh_out];
        base[f * a.fs_out + bo + (n & mo) * a.sn_out] = v[m];
      }
    }
  }
  if (!INV) signal_tail(a.sig);
}

template <int N>
constexpr size_t lds_line() {  // the z kernels: ZPlan<N>::T lines
  return sizeof(kcplx) * (N + MapLine<N>::zsize);
}
template <int N>
constexpr size_t lds_line_ea() {  // k_z_inv_fwd: ZPlanEA<N>::T lines
  return sizeof(kcplx) * (N + ZPlanEA<N>::T * MapLine<N>::LP);
}
template <int N>
constexpr size_t lds_line_full() {  // MapLine tiles of Plan<N>::T lines (k_gamma_z_fwd_tangent)
  return sizeof(kcplx) * (N + MapLine<N>::size);
}
template <int N>
constexpr size_t lds_strided() {
  return sizeof(kcplx) * (N + MapStrided<N>::size);
}

}  // namespace MRL_P2NS
}  // namespace mrl
