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

namespace mrl {

namespace p2 {

struct GammaArgs {
  cplx *spec;        // [9][nx][inner], transformed along z and y; projected in place
  long long field;   // elements per field = nx * inner
  long long inner;   // ny * nzc
  int nzc;
  const double *kx, *ky, *kz;
  double scale;
};

template <int N>
__global__ void __launch_bounds__(Plan<N>::NT, 2) k_gamma_xfused(GammaArgs a, const cplx *__restrict__ tw) {
#pragma clang fp contract(off)
  constexpr int P = Plan<N>::P, TPL = Plan<N>::TPL, T = Plan<N>::T;
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
  cplx *f0 = a.spec + (long long)(row * 3 + 0) * a.field + iv + (long long)q * a.inner;
  cplx *f1 = f0 + a.field, *f2 = f1 + a.field;
  const long long step = (long long)TPL * a.inner;
  cplx v0[P], v1[P];
#pragma unroll
  for (int m = 0; m < P; ++m) v0[m] = f0[m * step];
#pragma unroll
  for (int m = 0; m < P; ++m) v1[m] = f1[m * step];
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
  for (int m = 0; m < P; ++m) v0[m] = f2[m * step];  // third component: in flight during the second transform
  fft_line<N, Map>(v1, q, l, X, W);
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
#pragma unroll
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int m = 0; m < P; ++m) {
      const double qj = (j == 0) ? KX[q + m * TPL] : (j == 1 ? ky : kz);
      v0[m] = make_double2(s[m].y * qj, s[m].x * qj);
    }
    fft_line<N, Map>(v0, q, l, X, W);
    if (valid) {
      cplx *o = (j == 0) ? f0 : (j == 1 ? f1 : f2);
#pragma unroll
      for (int m = 0; m < P; ++m) o[m * step] = cswap(v0[m]);
    }
  }
}

template <int N>
static int launch_gamma_xfused(mrl_ctx *ctx, const GammaArgs &a) {
  static bool attr = false;
  constexpr size_t lds = lds_strided<N>() + sizeof(double) * N;
  if (!attr) {
    MRL_TRY(set_lds_attr(ctx, k_gamma_xfused<N>, lds));
    attr = true;
  }
  constexpr int T = Plan<N>::T;
  const long long nb = (a.inner + T - 1) / T;
  hipLaunchKernelGGL((k_gamma_xfused<N>), dim3((unsigned)nb, 3), dim3(Plan<N>::NT), lds, ctx->stream, a, ctx->ax[0].d_tw);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

}  // namespace p2

bool mech_fast_ok(const mrl_ctx *ctx) {
  return ctx->dim == 3 && !ctx->slab && ctx->spectrum == MRL_SPECTRUM_HALF && pow2_ok(ctx->n[0]) &&
         pow2_ok(ctx->n[1]) && pow2_ok(ctx->n[2]);
}

// out = scale * G(A), A and out field-major real [9][nx][ny][nz] (out may alias A).  dotv != nullptr: the last pass
// also accumulates sum(out * dotv) into the device scalar d_dot (deterministic two-stage sum)
int reduce_finalize_from(mrl_ctx *ctx, const double *partial, int nb, double *d_scalar);
int gamma_fast(mrl_ctx *ctx, const double *A, double *out, double scale, const double *dotv, double *d_dot) {
  const long long nx = ctx->n[0], ny = ctx->n[1], nz = ctx->n[2], nzc = ctx->nrec[2];
  const long long nreal = nx * ny * nz, nspec = nx * ny * nzc;
  MRL_TRY(ensure_work(ctx, 4, sizeof(cplx) * nspec * 9));
  cplx *spec = reinterpret_cast<cplx *>(ctx->d_work[4]);
  const double r = 8.0 * nreal * 9, h = 16.0 * nspec * 9;
  {
    ProfScope ps(ctx, "gamma_z_fwd", r + h);
    p2::ChDev none{};
    MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_fwd<NN, 0, 0>(ctx, A, spec, nullptr, nullptr, none, 9 * nx * ny / 2))));
  }
  p2::PassArgs pa{};
  pa.in[0] = spec;
  pa.out[0] = spec;
  pa.scale = 1.0;
  pa.inner = nzc;
  pa.outer = 9 * nx;
  pa.so_in = pa.so_out = ny * nzc;
  pa.sn_in = pa.sn_out = nzc;
  {
    ProfScope ps(ctx, "gamma_y_fwd", 2.0 * h);
    pa.reverse = 1;
    MRL_SWITCH_N(ny, MRL_TRY((p2::launch_pass_t<NN, false, 1>(ctx, pa, ctx->ax[1].d_tw))));
  }
  {
    ProfScope ps(ctx, "gamma_x_fused", 2.0 * h);
    p2::GammaArgs g{};
    g.spec = spec;
    g.field = nspec;
    g.inner = ny * nzc;
    g.nzc = (int)nzc;
    g.kx = ctx->d_k[0];
    g.ky = ctx->d_k[1];
    g.kz = ctx->d_k[2];
    g.scale = scale;
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
    MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_inv<NN>(ctx, spec, out, norm, 9 * nx * ny / 2))));
    return MRL_OK;
  }
  ProfScope ps(ctx, "gamma_z_inv_dot", 2.0 * r + h);
  const long long max_blocks = 9 * nx * ny / 2;  // >= the number of workgroups for every plan
  MRL_TRY(ensure_work(ctx, 3, sizeof(double) * (size_t)max_blocks));
  int nb = 0;
  MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_inv_dot<NN>(ctx, spec, out, norm, 9 * nx * ny / 2, dotv, ctx->d_work[3], &nb))));
  return reduce_finalize_from(ctx, ctx->d_work[3], nb, d_dot);
}

}  // namespace mrl
