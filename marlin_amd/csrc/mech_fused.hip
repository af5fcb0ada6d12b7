// Fast path of the Gamma-operator application G(A) = ifft( Ghat4 : fft(A) ) (FFTMechanics.C:74-84,105-106) for
// 3-D power-of-two grids, on FIELD-MAJOR data: the 9 components are 9 contiguous scalar fields [c][nx][ny][nz].
// (The Newton-CG driver keeps its vectors field-major internally, so every kernel streams coalesced; the
// value-major layout of the reference only exists at the C ABI, mech.hip converts on entry/exit.)
//   z forward   k_z_fwd<PAIR> over all 9 fields' lines           r + h      per field
//   y forward   k_pass over [9*nx] slices                         2h
//   x + Gamma   k_gamma_xfused: per tensor row i: forward x of (A_i0, A_i1, A_i2), s = (sum_k A_ik q_k)/|q|^2,
//               out_ij = s q_j, inverse x -- Ghat4 (1296 B per k-point in the reference) is never formed      2h
//   y inverse   k_pass                                            2h
//   z inverse   k_z_inv, x scale/N                                h + r
// = 9 * (2r + 8h) bytes per application (r = 8 B, h = 8(1+2/n) B per grid point): 724.5 B/pt at n = 128.
#include "fft_pow2_launch.h"
#include <atomic>

// Register budget of k_gamma_xfused (16 points per thread: s, v0, v1 = 192 VGPRs of the 256 that two workgroups per CU allow):
// the three output transforms must NOT be unrolled into each other (the scheduler then overlaps them and spills 50-90 VGPRs: 128^3
// 78 -> 62 us without the unrolling), and only half of the third component is prefetched behind the second transform (-> 58 us).
// Compile-time switches so that variants can be built side by side (make EXTRA=-D...) and compared on one GPU box.
#ifndef MRL_GAMMA_PRE
#define MRL_GAMMA_PRE 8
#endif
#ifndef MRL_GAMMA_PRE_BIG  // lines of 256 points and more (12-44 spilled VGPRs with 8)
#define MRL_GAMMA_PRE_BIG 4
#endif
#ifndef MRL_GAMMA_JUNROLL
#define MRL_GAMMA_JUNROLL 1
#endif
#include "mech_math.h"

namespace mrl {

namespace p2 {

struct GammaArgs {
  cplx *spec;        // [9][nx][plane], transformed along z and y; projected in place
  long long field;   // elements per field = nx * plane
  long long inner;   // ny * nzc: valid elements of an x plane
  long long plane;   // elements between two x planes (padded to an odd number of 256-byte pieces: fft_pow2_kernels.h ZLay)
  int nzc;
  const double *kx, *ky, *kz;
  double scale;
  const int *stop;   // optional: non-zero = nothing to do (PassArgs::stop)
};

template <int N>
__global__ void __launch_bounds__(Plan<N>::NT, 2) k_gamma_xfused(GammaArgs a, const cplx *__restrict__ tw) {
#pragma clang fp contract(off)
  constexpr int P = Plan<N>::P, TPL = Plan<N>::TPL, T = Plan<N>::T;
  if (a.stop && *a.stop) return;
  constexpr int GPRE0 = N >= 256 ? MRL_GAMMA_PRE_BIG : MRL_GAMMA_PRE, GPRE = GPRE0 < P ? GPRE0 : P;
  using Map = MapStrided<N>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx *W = reinterpret_cast<cplx *>(smem);
  cplx *X = W + N;
  double *KX = reinterpret_cast<double *>(X + Map::size);
  const int l = threadIdx.x % T, q = threadIdx.x / T;
  const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
  const int row = blockIdx.y;
  const long long i = (long long)logical * T + l;
  const bool valid = i < a.inner;
  const long long iv = valid ? i : 0;

  constexpr int NT = Plan<N>::NT, CNT = (N + NT - 1) / NT;
  cplx twv[CNT];
  double kxv[CNT];
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    const int idx = threadIdx.x + j * NT;
    twv[j] = idx < N ? tw[idx] : make_double2(0.0, 0.0);
    kxv[j] = idx < N ? a.kx[idx] : 0.0;
  }
  const double ky = a.ky[iv / a.nzc], kz = a.kz[iv % a.nzc];
  // wave-uniform field bases + one 32-bit byte offset per thread (fields are < 4 GiB, checked by mech_fast_ok): the loads and
  // stores take the "SGPR base + VGPR offset" form instead of 3 x 16 64-bit addresses held in registers
  char *b0 = reinterpret_cast<char *>(a.spec + (long long)(row * 3 + 0) * a.field);
  char *b1 = b0 + a.field * 16, *b2 = b1 + a.field * 16;
  const unsigned boff0 = (unsigned)((iv + (long long)q * a.plane) * 16), stepB = (unsigned)((long long)TPL * a.plane * 16);
  auto ldf = [=](const char *b, int m) { return *reinterpret_cast<const cplx *>(b + (boff0 + (unsigned)m * stepB)); };
  cplx v0[P], v1[P];
#pragma unroll
  for (int m = 0; m < P; ++m) v0[m] = ldf(b0, m);
#pragma unroll
  for (int m = 0; m < P; ++m) v1[m] = ldf(b1, m);
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    const int idx = threadIdx.x + j * NT;
    if (idx < N) {
      W[idx] = twv[j];
      KX[idx] = kxv[j];
    }
  }

  // s = sum_k A_ik q_k  (q = (kx along the line, ky, kz))
  cplx s[P];
  fft_line<N, Map>(v0, q, l, X, W);
#pragma unroll
  for (int m = 0; m < P; ++m) {
    const double kx = KX[q + m * TPL];
    s[m] = make_double2(v0[m].x * kx, v0[m].y * kx);
  }
#pragma unroll
  for (int m = 0; m < GPRE; ++m) v0[m] = ldf(b2, m);  // third component: GPRE values in flight during the second transform
  fft_line<N, Map>(v1, q, l, X, W);
#pragma unroll
  for (int m = GPRE; m < P; ++m) v0[m] = ldf(b2, m);
#pragma unroll
  for (int m = 0; m < P; ++m) {
    s[m].x += v1[m].x * ky;
    s[m].y += v1[m].y * ky;
  }
  fft_line<N, Map>(v0, q, l, X, W);
  const double kyz2 = ky * ky + kz * kz;
#pragma unroll
  for (int m = 0; m < P; ++m) {
    const double kx = KX[q + m * TPL];
    const double Q = kx * kx + kyz2;
    const double inv = (Q == 0.0) ? 0.0 : a.scale / Q;
    s[m].x = (s[m].x + v0[m].x * kz) * inv;
    s[m].y = (s[m].y + v0[m].y * kz) * inv;
  }

  // out_ij = s q_j, inverse x (unnormalised; 1/N applied by the z pass); swap trick for the inverse
#pragma unroll MRL_GAMMA_JUNROLL
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int m = 0; m < P; ++m) {
      const double qj = (j == 0) ? KX[q + m * TPL] : (j == 1 ? ky : kz);
      v0[m] = make_double2(s[m].y * qj, s[m].x * qj);
    }
    fft_line<N, Map>(v0, q, l, X, W);
    if (valid) {
      char *o = (j == 0) ? b0 : (j == 1 ? b1 : b2);
#pragma unroll
      for (int m = 0; m < P; ++m) *reinterpret_cast<cplx *>(o + (boff0 + (unsigned)m * stepB)) = cswap(v0[m]);
    }
  }
}

template <int N>
static int launch_gamma_xfused(mrl_ctx *ctx, const GammaArgs &a) {
  static std::atomic<bool> attr{false};  // (two host threads may both set the attribute: harmless, and no torn flag)
  constexpr size_t lds = lds_strided<N>() + sizeof(double) * N;
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY(set_lds_attr(ctx, k_gamma_xfused<N>, lds));
    attr.store(true, std::memory_order_release);
  }
  constexpr int T = Plan<N>::T;
  const long long nb = (a.inner + T - 1) / T;
  hipLaunchKernelGGL((k_gamma_xfused<N>), dim3((unsigned)nb, 3), dim3(Plan<N>::NT), lds, ctx->stream, a, ctx->ax[0].d_tw);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}


// Conjugate-gradient direction update + tangent application fused into the forward z pass of the Gamma operator:
//   p <- r + (S[i_num]/S[i_den]) p ;  spec = fft_z( K4 : p )        (MarlinUtils.h:107-117 + FFTMechanics.C:107-108 + the z pass of :105)
// The tangent couples the 9 components of a grid point, a z transform the points of one line, so a workgroup takes a tile of
// 512 grid points = R = 512/N complete z lines (the same lines of all 9 fields): phase 1 evaluates the tangent at two points
// per thread, 16 B per lane and component stream exactly as k_mech_tangent_fm2 does, and parks the 9 results in LDS as the
// inputs of 9*R/2 complex transforms (two real lines each); phase 2 runs them.  K4:p (9 fields) is never written to HBM nor read
// back: 2 x 72 of the 450 B per grid point that the two separate kernels move.
// XUPD: the solution update of the PREVIOUS iteration, x += (S[i_arz]/S[i_apAp]) p_old, rides along (p_old is in registers here
// anyway; k_cg_update<.., SKIP_X> then leaves x and p alone): one read of the 9-field direction less per iteration.
template <int N, bool NTV, bool XUPD>
__global__ void __launch_bounds__(Plan<N>::NT, 2) k_gamma_z_fwd_tangent(const double *__restrict__ F, const double *__restrict__ K,
                                                                        const double *__restrict__ mu, double *__restrict__ pdir,
                                                                        const double *__restrict__ r, const double *__restrict__ S,
                                                                        int i_num, int i_den, cplx *__restrict__ spec,
                                                                        long long npts, long long rows_total,
                                                                        const cplx *__restrict__ tw, double *__restrict__ xsol,
                                                                        int i_arz, int i_apAp, ZLay zl, const int *__restrict__ stop) {
#pragma clang fp contract(off)
  constexpr int P = Plan<N>::P, TPL = Plan<N>::TPL, NZC = N / 2 + 1, R = 512 / N, NL = 9 * R / 2;
  if (stop && *stop) return;  // the solve converged while this iteration was already enqueued: p and x stay as they are
  static_assert(Plan<N>::NT == 256 && R >= 2 && R % 2 == 0 && NL <= Plan<N>::T, "tile shape");
  using Map = MapLine<N>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx *W = reinterpret_cast<cplx *>(smem);
  cplx *X = W + N;
  double *Xd = reinterpret_cast<double *>(X);
  const int t = threadIdx.x;
  const long long tile = xcd_remap(blockIdx.x, gridDim.x);
  TwRegs<N> twr;
  tw_issue_staged<N>(twr, tw);

  // ---- phase 1: two grid points per thread
  const double beta = S[i_num] / S[i_den];
  const double alpha = XUPD ? S[i_arz] / S[i_apAp] : 0.0;
  const long long h = tile * 256 + t;
  {
    Mat<3> f[2], d[2];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const double2 *fp = reinterpret_cast<const double2 *>(F + (long long)c * npts) + h;
      const double2 fv = NTV ? ld_nt(fp) : *fp;
      f[0].a[c / 3][c % 3] = fv.x;
      f[1].a[c / 3][c % 3] = fv.y;
      double2 *pp = reinterpret_cast<double2 *>(pdir + (long long)c * npts) + h;
      const double2 pv = *pp;
      if (XUPD) {
        double2 *xp = reinterpret_cast<double2 *>(xsol + (long long)c * npts) + h;
        const double2 xv = NTV ? ld_nt(xp) : *xp;
        const double2 xn = make_double2(xv.x + alpha * pv.x, xv.y + alpha * pv.y);
        if (NTV)
          st_nt(xp, xn);
        else
          *xp = xn;
      }
      const double2 rv = reinterpret_cast<const double2 *>(r + (long long)c * npts)[h];
      const double2 dv = make_double2(rv.x + beta * pv.x, rv.y + beta * pv.y);
      if (NTV)
        st_nt(pp, dv);
      else
        *pp = dv;
      d[0].a[c / 3][c % 3] = dv.x;
      d[1].a[c / 3][c % 3] = dv.y;
    }
    const double2 *Kp = reinterpret_cast<const double2 *>(K) + h, *mp = reinterpret_cast<const double2 *>(mu) + h;
    const double2 Kv = NTV ? ld_nt(Kp) : *Kp, mv = NTV ? ld_nt(mp) : *mp;
    tw_commit<N>(twr, W);
    const Mat<3> o0 = svk_tangent<3>(f[0], d[0], Kv.x, mv.x);
    const Mat<3> o1 = svk_tangent<3>(f[1], d[1], Kv.y, mv.y);
    // points 2t, 2t+1 of the tile: line (2t)/N, positions z, z+1; lines 2j, 2j+1 are the real / imaginary part of transform j
    const int row = (2 * t) / N, z = (2 * t) % N, pair = row >> 1, odd = row & 1;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const int line = c * (R / 2) + pair;
      Xd[2 * Map::at(z, line) + odd] = o0.a[c / 3][c % 3];
      Xd[2 * Map::at(z + 1, line) + odd] = o1.a[c / 3][c % 3];
    }
  }
  __syncthreads();

  // ---- phase 2: the 9*R/2 transforms (threads beyond them transform line 0 again and store nothing)
  const int q = t % TPL, l = t / TPL;
  const bool valid = l < NL;
  const int lr = valid ? l : 0;
  cplx v[P];
#pragma unroll
  for (int m = 0; m < P; ++m) v[m] = X[Map::at(q + m * TPL, lr)];
  __syncthreads();
  fft_line<N, Map>(v, q, l, X, W);
  const int c = lr / (R / 2), pr = lr % (R / 2);
  const long long row0 = (long long)c * rows_total + tile * R + 2 * pr;  // (rows of all nine fields are numbered through: plane = row / ny)
  // the k <-> N - k pairing by lane exchange where the plan allows it (fft_pow2_kernels.h: store_half_spectra), through LDS otherwise
  store_half_spectra<N, Map>(v, q, l, X, valid, spec + zrow(row0, NZC, zl), spec + zrow(row0 + 1, NZC, zl));
}

template <int N>
static int launch_gamma_z_fwd_tangent(mrl_ctx *ctx, const double *F, const double *K, const double *mu, double *p, const double *r,
                                      const double *S, int i_num, int i_den, cplx *spec, long long npts, long long rows, bool nt,
                                      double *x, int i_arz, int i_apAp, ZLay zl, const int *stop = nullptr) {
  static std::atomic<bool> attr{false};  // (two host threads may both set the attribute: harmless, and no torn flag)
  constexpr size_t lds = lds_line_full<N>();
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY(set_lds_attr(ctx, k_gamma_z_fwd_tangent<N, true, true>, lds));
    MRL_TRY(set_lds_attr(ctx, k_gamma_z_fwd_tangent<N, false, true>, lds));
    MRL_TRY(set_lds_attr(ctx, k_gamma_z_fwd_tangent<N, true, false>, lds));
    MRL_TRY(set_lds_attr(ctx, k_gamma_z_fwd_tangent<N, false, false>, lds));
    attr.store(true, std::memory_order_release);
  }
  const unsigned nb = (unsigned)(npts / 512);
#define MRL_GZT(NTV_, XU_)                                                                                                        \
  hipLaunchKernelGGL((k_gamma_z_fwd_tangent<N, NTV_, XU_>), dim3(nb), dim3(256), lds, ctx->stream, F, K, mu, p, r, S, i_num, i_den, \
                     spec, npts, rows, ctx->ax[2].d_tw, x, i_arz, i_apAp, zl, stop)
  if (nt) {
    if (x) MRL_GZT(true, true); else MRL_GZT(true, false);
  } else {
    if (x) MRL_GZT(false, true); else MRL_GZT(false, false);
  }
#undef MRL_GZT
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

}  // namespace p2

// x-plane pitch of the spectral scratch [9][nx][plane] of the Gamma operator: ny * nzc padded to an odd number of 256-byte pieces (the
// fused x pass gathers 256-byte pieces one plane apart; the natural pitch of a power-of-two grid -- 128^3: 520 pieces -- puts them on
// 16 of the memory channels).  Internal scratch: no interface sees it.
static long long mech_plane(const mrl_ctx *ctx) {
  long long plane = (ctx->n[1] * ctx->nrec[2] + 15) / 16 * 16;
  if ((plane / 16) % 2 == 0) plane += 16;
  return (ctx->exp & (1 << 22)) ? ctx->n[1] * ctx->nrec[2] : plane;   // experiment bit 1 << 22: dense planes (A/B)
}

bool mech_fast_ok(const mrl_ctx *ctx) {
  // (one spectral field < 4 GiB: k_gamma_xfused addresses it with 32-bit byte offsets)
  return ctx->dim == 3 && !ctx->slab && !ctx->pencil && ctx->spectrum == MRL_SPECTRUM_HALF && pow2_ok(ctx->n[0]) &&
         pow2_ok(ctx->n[1]) && pow2_ok(ctx->n[2]) && 16.0 * (double)(ctx->n[0] * (ctx->n[1] * ctx->nrec[2] + 32)) < 4294967296.0;
}

// out = scale * G(A), A and out field-major real [9][nx][ny][nz] (out may alias A).  dotv != nullptr: the last pass
// also accumulates sum(out * dotv) into the device scalar d_dot (deterministic two-stage sum)
int reduce_finalize_from(mrl_ctx *ctx, const double *partial, int nb, double *d_scalar);
int reduce_finalize_from_guarded(mrl_ctx *ctx, const double *partial, int nb, double *d_scalar, const int *stop);
// the passes after the forward z pass: y forward, x + projection, y inverse, z inverse (+ optional dot product)
// stop (optional): device word checked by every launch (a CG iteration enqueued ahead of the host's convergence test, mech.hip)
static int gamma_fast_rest(mrl_ctx *ctx, cplx *spec, double *out, double scale, const double *dotv, double *d_dot,
                           const int *stop = nullptr) {
  const long long nx = ctx->n[0], ny = ctx->n[1], nz = ctx->n[2], nzc = ctx->nrec[2];
  const long long nreal = nx * ny * nz, nspec = nx * ny * nzc, plane = mech_plane(ctx);
  const p2::ZLay zl{(unsigned)ny, (unsigned)(plane - ny * nzc)};
  const double r = 8.0 * nreal * 9, h = 16.0 * nspec * 9;
  p2::PassArgs pa{};
  pa.in[0] = spec;
  pa.out[0] = spec;
  pa.scale = 1.0;
  pa.inner = nzc;
  pa.outer = 9 * nx;
  pa.so_in = pa.so_out = plane;
  pa.sn_in = pa.sn_out = nzc;
  pa.stop = stop;
  {
    ProfScope ps(ctx, "gamma_y_fwd", 2.0 * h);
    pa.reverse = 1;
    MRL_SWITCH_N(ny, MRL_TRY((p2::launch_pass_t<NN, false, 1>(ctx, pa, ctx->ax[1].d_tw))));
  }
  {
    ProfScope ps(ctx, "gamma_x_fused", 2.0 * h);
    p2::GammaArgs g{};
    g.spec = spec;
    g.field = nx * plane;
    g.inner = ny * nzc;
    g.plane = plane;
    g.nzc = (int)nzc;
    g.kx = ctx->d_k[0];
    g.ky = ctx->d_k[1];
    g.kz = ctx->d_k[2];
    g.scale = scale;
    g.stop = stop;
    MRL_SWITCH_N(nx, MRL_TRY((p2::launch_gamma_xfused<NN>(ctx, g))));
  }
  {
    ProfScope ps(ctx, "gamma_y_inv", 2.0 * h);
    pa.reverse = 0;
    MRL_SWITCH_N(ny, MRL_TRY((p2::launch_pass_t<NN, true, 1>(ctx, pa, ctx->ax[1].d_tw))));
  }
  const double norm = 1.0 / ((double)nx * (double)ny * (double)nz);
  if (!dotv) {
    ProfScope ps(ctx, "gamma_z_inv", r + h);
    MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_inv<NN>(ctx, spec, out, norm, 9 * nx * ny / 2, zl))));
    return MRL_OK;
  }
  ProfScope ps(ctx, "gamma_z_inv_dot", 2.0 * r + h);
  const long long max_blocks = 9 * nx * ny / 2;  // >= the number of workgroups for every plan
  MRL_TRY(ensure_work(ctx, 3, sizeof(double) * (size_t)max_blocks));
  int nb = 0;
  MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_inv_dot<NN>(ctx, spec, out, norm, 9 * nx * ny / 2, dotv, ctx->d_work[3], &nb, zl, stop))));
  return reduce_finalize_from_guarded(ctx, ctx->d_work[3], nb, d_dot, stop);
}

int gamma_fast(mrl_ctx *ctx, const double *A, double *out, double scale, const double *dotv, double *d_dot) {
  const long long nx = ctx->n[0], ny = ctx->n[1], nz = ctx->n[2], nzc = ctx->nrec[2];
  const long long nreal = nx * ny * nz, nspec = nx * ny * nzc, plane = mech_plane(ctx);
  const p2::ZLay zl{(unsigned)ny, (unsigned)(plane - ny * nzc)};
  MRL_TRY(ensure_work(ctx, 4, sizeof(cplx) * nx * plane * 9));
  cplx *spec = reinterpret_cast<cplx *>(ctx->d_work[4]);
  const double r = 8.0 * nreal * 9, h = 16.0 * nspec * 9;
  {
    ProfScope ps(ctx, "gamma_z_fwd", r + h);
    p2::ChDev none{};
    MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_fwd<NN, 0, 0>(ctx, A, spec, nullptr, nullptr, none, 9 * nx * ny / 2, zl))));
  }
  return gamma_fast_rest(ctx, spec, out, scale, dotv, d_dot);
}

// shared with the slab pipeline (slab_mech_fused.hip): the same kernel on the rank's [nx][ny_local][nz] block
int gamma_z_fwd_tangent_launch(mrl_ctx *ctx, const double *F, const double *K, const double *mu, double *p, const double *r,
                               const double *S, int i_num, int i_den, cplx *spec, long long npts, long long rows, int nz, bool nt,
                               double *x, int i_arz, int i_apAp) {
  const p2::ZLay zl{0u, 0u};  // (the slab work arrays are dense)
  switch (nz) {
    case 32: return p2::launch_gamma_z_fwd_tangent<32>(ctx, F, K, mu, p, r, S, i_num, i_den, spec, npts, rows, nt, x, i_arz, i_apAp, zl);
    case 64: return p2::launch_gamma_z_fwd_tangent<64>(ctx, F, K, mu, p, r, S, i_num, i_den, spec, npts, rows, nt, x, i_arz, i_apAp, zl);
    case 128: return p2::launch_gamma_z_fwd_tangent<128>(ctx, F, K, mu, p, r, S, i_num, i_den, spec, npts, rows, nt, x, i_arz, i_apAp, zl);
    case 256: return p2::launch_gamma_z_fwd_tangent<256>(ctx, F, K, mu, p, r, S, i_num, i_den, spec, npts, rows, nt, x, i_arz, i_apAp, zl);
    default: return set_error(ctx, MRL_ERR_UNSUPPORTED, "fused tangent + z pass: unplanned z length %d", nz);
  }
}

// the fused CG direction update + tangent + forward z pass exists for z lines of 32 ... 256 points in whole 512-point tiles
bool gamma_tangent_fusable(const mrl_ctx *ctx) {
  if (!mech_fast_ok(ctx)) return false;
  const long long nz = ctx->n[2], rows = ctx->n[0] * ctx->n[1];
  if (!(nz == 32 || nz == 64 || nz == 128 || nz == 256)) return false;
  return rows % (512 / nz) == 0;
}

// [x += (S[i_arz]/S[i_apAp]) p, the previous iteration's solution update, if x != nullptr ;]
// p <- r + (S[i_num]/S[i_den]) p ; out = G(K4 : p) ; *d_dot = p . out     (one CG iteration's operator application)
int gamma_fast_tangent_dir(mrl_ctx *ctx, const double *F, const double *K, const double *mu, double *p, const double *r,
                           const double *S, int i_num, int i_den, double *out, double *d_dot, bool nt, double *x, int i_arz,
                           int i_apAp, const int *stop) {
  const long long nx = ctx->n[0], ny = ctx->n[1], nz = ctx->n[2], nzc = ctx->nrec[2];
  const long long npts = nx * ny * nz, nspec = nx * ny * nzc, plane = mech_plane(ctx);
  const p2::ZLay zl{(unsigned)ny, (unsigned)(plane - ny * nzc)};
  MRL_TRY(ensure_work(ctx, 4, sizeof(cplx) * nx * plane * 9));
  cplx *spec = reinterpret_cast<cplx *>(ctx->d_work[4]);
  {
    ProfScope ps(ctx, "gamma_z_fwd_tangent_dir", 8.0 * npts * ((x ? 6 : 4) * 9 + 2) + 16.0 * nspec * 9);
    switch (nz) {
      case 32: MRL_TRY((p2::launch_gamma_z_fwd_tangent<32>(ctx, F, K, mu, p, r, S, i_num, i_den, spec, npts, nx * ny, nt, x, i_arz, i_apAp, zl, stop))); break;
      case 64: MRL_TRY((p2::launch_gamma_z_fwd_tangent<64>(ctx, F, K, mu, p, r, S, i_num, i_den, spec, npts, nx * ny, nt, x, i_arz, i_apAp, zl, stop))); break;
      case 128: MRL_TRY((p2::launch_gamma_z_fwd_tangent<128>(ctx, F, K, mu, p, r, S, i_num, i_den, spec, npts, nx * ny, nt, x, i_arz, i_apAp, zl, stop))); break;
      case 256: MRL_TRY((p2::launch_gamma_z_fwd_tangent<256>(ctx, F, K, mu, p, r, S, i_num, i_den, spec, npts, nx * ny, nt, x, i_arz, i_apAp, zl, stop))); break;
      default: return set_error(ctx, MRL_ERR_UNSUPPORTED, "gamma_fast_tangent_dir: unplanned z length");
    }
  }
  return gamma_fast_rest(ctx, spec, out, 1.0, p, d_dot, stop);
}

}  // namespace mrl
