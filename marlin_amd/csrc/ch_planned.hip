// Planned path for grids whose extents have register-radix plans for the plain transforms but are not all in the fused family of
// ch_fused.hip: every axis length in {60, 90, 120, 150, 180, 240, 270, 300, 360, 450, 600} (radix 30 first), {160, 320, 640, 800,
// 1280} (radix 20 first) or {72, 216, 288, 432, 576, 864, 1152}, mixed freely with the lengths of the fused path.  The reference's
// own solver tests run 150^2 (test/tests/solvers/diagonal.i:5-95); such sizes used to take the any-length path (fft_generic.hip).
//   plain transforms        z pass (two real lines per complex transform) -> y pass -> x pass, each reading and writing its array once
//   Cahn-Hilliard substep   forward z with mu = f'(c) in the loader (c and mu in one complex transform) -> forward y of both fields ->
//                           x passes with the k-space update (Nhat = Mbar mu-hat, ABM predictor, 1/(1 - dt Lbar);
//                           AdamsBashforthMoulton.C:94-99) -> inverse y -> inverse z
//   120 / 150 / 160 / 180 / 240 / 300 / 320 points   two-stage plans (fft_two.h, fft_two_z.h: at most 16 / 20 points per thread): both fields per y
//                           launch, ONE fused x kernel, and -- when the z extent is one of them -- ch_substeps_planned, the substep
//                           loop with the inverse z pass of a substep fused into the forward z pass of the next: 14 h of traffic
//                           per AB2 substep (h = one half-spectrum array), as on the fused path
//   the other lengths       uniform 30- / 20- / 12-point plans: one field per y launch, the x update in two kernels (k_x_mbar,
//                           k_x_update), separate z passes: 15 h
// Same operations in the same order on the pointwise side everywhere: results equal the fused and the generic path to the rounding of
// the transforms.
#include "fft_pow2_launch.h"
#include "fft_two.h"
#include "fft_two_z.h"
#include <atomic>

namespace mrl {

// the same launchers for the lengths of the fused path (ch_fused.hip)
int pass_launch_std(mrl_ctx *ctx, long long n, bool inv, int nf, const p2::PassArgs &a, const cplx *tw);
int z_fwd_launch_std(mrl_ctx *ctx, long long n, int mode, int fam, const double *in, cplx *o0, cplx *o1, double *mu, const p2::ChDev &chp,
                     long long nlines);
int z_inv_launch_std(mrl_ctx *ctx, long long n, const cplx *in, double *out, double scale, long long nlines);
int z_inv_fwd_launch_std(mrl_ctx *ctx, long long n, int fam, const cplx *in, cplx *o0, cplx *o1, double *mu, const p2::ChDev &chp, double scale,
                         long long nlines);
int ch_kspace_launch(mrl_ctx *ctx, const ChP &cp, const double *cbar, const double *mubar, double *Nhat, double *ubar,
                     const double *const *Nold, int order, double sub_dt);

#define MRL_SWITCH_N30(n, CALL)                          \
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
    default: return MRL_ERR_UNSUPPORTED;                 \
  }

#define MRL_SWITCH_N2(n, CALL)                          \
  switch (n) {                                          \
    case 120: { constexpr int NN = 120; CALL; } break;  \
    case 150: { constexpr int NN = 150; CALL; } break;  \
    case 160: { constexpr int NN = 160; CALL; } break;  \
    case 180: { constexpr int NN = 180; CALL; } break;  \
    case 240: { constexpr int NN = 240; CALL; } break;  \
    case 300: { constexpr int NN = 300; CALL; } break;  \
    case 320: { constexpr int NN = 320; CALL; } break;  \
    default: return MRL_ERR_UNSUPPORTED;                \
  }

#define MRL_SWITCH_N2X(n, CALL)                         \
  switch (n) {                                          \
    case 400: { constexpr int NN = 400; CALL; } break;  \
    default: MRL_SWITCH_N2(n, CALL)                     \
  }

// experiment bit 1 << 29: the uniform 30- / 20-point plans where the two-stage plans of fft_two.h would run (A/B, tests)
static bool two_stage(const mrl_ctx *ctx, long long n) { return p2::two_stage_len(n) && !(ctx->exp & (1 << 29)); }
static bool two_stage_x(const mrl_ctx *ctx, long long n) { return p2::two_stage_x_len(n) && !(ctx->exp & (1 << 29)); }

static int pass_launch(mrl_ctx *ctx, long long n, bool inv, int nf, const p2::PassArgs &a, const cplx *tw) {
  if (!two_stage(ctx, n) && !plain30_ok(n)) return pass_launch_std(ctx, n, inv, nf, a, tw);
  if (two_stage(ctx, n)) {  // 16 (20) points per thread at most: both fields of the forward passes in one launch
    if (inv) {
      if (nf == 2) {
        MRL_SWITCH_N2(n, MRL_TRY((p2::launch_pass2<NN, true, 2>(ctx, a, tw))));
      } else {
        MRL_SWITCH_N2(n, MRL_TRY((p2::launch_pass2<NN, true, 1>(ctx, a, tw))));
      }
    } else {
      if (nf == 2) {
        MRL_SWITCH_N2(n, MRL_TRY((p2::launch_pass2<NN, false, 2>(ctx, a, tw))));
      } else {
        MRL_SWITCH_N2(n, MRL_TRY((p2::launch_pass2<NN, false, 1>(ctx, a, tw))));
      }
    }
    return MRL_OK;
  }
  // 30 points per thread: one field per launch (two fields would need 240 of the 256 vector registers for the data alone)
  for (int f = 0; f < nf; ++f) {
    p2::PassArgs b = a;
    b.in[0] = a.in[f];
    b.out[0] = a.out[f];
    if (inv) {
      MRL_SWITCH_N30(n, MRL_TRY((p2::launch_pass_t<NN, true, 1>(ctx, b, tw))));
    } else {
      MRL_SWITCH_N30(n, MRL_TRY((p2::launch_pass_t<NN, false, 1>(ctx, b, tw))));
    }
  }
  return MRL_OK;
}

static int z_fwd_launch(mrl_ctx *ctx, long long n, int mode, int fam, const double *in, cplx *o0, cplx *o1, double *mu, const p2::ChDev &chp,
                        long long nlines) {
  if (!two_stage(ctx, n) && !plain30_ok(n)) return z_fwd_launch_std(ctx, n, mode, fam, in, o0, o1, mu, chp, nlines);
  if (two_stage(ctx, n)) {
    if (mode == 0) {
      MRL_SWITCH_N2(n, MRL_TRY((p2::launch_z_fwd2<NN, 0, 0>(ctx, in, o0, o1, mu, chp, nlines))));
    } else if (fam == MRL_FE_DOUBLE_WELL) {
      MRL_SWITCH_N2(n, MRL_TRY((p2::launch_z_fwd2<NN, 1, MRL_FE_DOUBLE_WELL>(ctx, in, o0, o1, mu, chp, nlines))));
    } else {
      MRL_SWITCH_N2(n, MRL_TRY((p2::launch_z_fwd2<NN, 1, MRL_FE_PFHUB>(ctx, in, o0, o1, mu, chp, nlines))));
    }
    return MRL_OK;
  }
  if (mode == 0) {
    MRL_SWITCH_N30(n, MRL_TRY((p2::launch_z_fwd<NN, 0, 0>(ctx, in, o0, o1, mu, chp, nlines))));
  } else if (fam == MRL_FE_DOUBLE_WELL) {
    MRL_SWITCH_N30(n, MRL_TRY((p2::launch_z_fwd<NN, 1, MRL_FE_DOUBLE_WELL>(ctx, in, o0, o1, mu, chp, nlines))));
  } else {
    MRL_SWITCH_N30(n, MRL_TRY((p2::launch_z_fwd<NN, 1, MRL_FE_PFHUB>(ctx, in, o0, o1, mu, chp, nlines))));
  }
  return MRL_OK;
}

static int z_inv_launch(mrl_ctx *ctx, long long n, const cplx *in, double *out, double scale, long long nlines) {
  if (!two_stage(ctx, n) && !plain30_ok(n)) return z_inv_launch_std(ctx, n, in, out, scale, nlines);
  if (two_stage(ctx, n)) {
    MRL_SWITCH_N2(n, MRL_TRY((p2::launch_z_inv2<NN>(ctx, in, out, scale, nlines))));
    return MRL_OK;
  }
  MRL_SWITCH_N30(n, MRL_TRY((p2::launch_z_inv<NN>(ctx, in, out, scale, nlines))));
  return MRL_OK;
}


namespace p2 {

// ---- the k-space update folded into the x passes (what k_ch_xfused does in one kernel on the fused path, here in two so that one
//      30-point array per thread suffices) -------------------------------------------------------------------------------------
struct XKArgs {
  const cplx *in;      // [nx][inner] work array, line along x (stride inner)
  cplx *out;           // k_x_mbar: Nhat_new (dense, same layout) ; k_x_update: ubar, x-inverted (may alias in)
  const cplx *Nnew;    // k_x_update
  const cplx *Nold[4];
  cplx *cbar;          // k_x_update: optional c-hat output
  double coef[5];      // sub_dt * beta[order][i]
  double M, kappa, dt;
  long long inner;
  int nzc;
  const double *kx, *ky, *kz;
};

// Nhat = Mbar * fft_x(mu-hat), Mbar = -k^2 M  (ReciprocalLaplacianFactor.C:28-31; k^2 = (kx^2 + ky^2) + kz^2 as DomainAction.C:1503-1509)
template <int N>
__global__ void __launch_bounds__(Plan<N>::NT, 2) k_x_mbar(XKArgs a, const cplx *__restrict__ tw) {
#pragma clang fp contract(off)
  constexpr int P = Plan<N>::P, TPL = Plan<N>::TPL, T = Plan<N>::T, NT = Plan<N>::NT, CNT = (N + NT - 1) / NT;
  using Map = MapStrided<N>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx *W = reinterpret_cast<cplx *>(smem);
  cplx *X = W + N;
  double *KX = reinterpret_cast<double *>(X + Map::size);
  const int l = threadIdx.x % T, q = threadIdx.x / T;
  const long long i = (long long)xcd_remap(blockIdx.x, gridDim.x) * T + l;
  const bool valid = i < a.inner;
  const long long ic = valid ? i : 0;
  cplx twv[CNT];
  double kxv[CNT];
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    const int idx = threadIdx.x + j * NT;
    twv[j] = idx < N ? tw[idx] : make_double2(0.0, 0.0);
    kxv[j] = idx < N ? a.kx[idx] : 0.0;
  }
  const double ky = a.ky[ic / a.nzc], kz = a.kz[ic % a.nzc];
  cplx v[P];
  const cplx *src = a.in + ic + (long long)q * a.inner;
#pragma unroll
  for (int m = 0; m < P; ++m) v[m] = src[(long long)m * TPL * a.inner];
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    const int idx = threadIdx.x + j * NT;
    if (idx < N) {
      W[idx] = twv[j];
      KX[idx] = kxv[j];
    }
  }
  fft_line<N, Map>(v, q, l, X, W);
  if (!valid) return;
  const double ky2 = ky * ky, kz2 = kz * kz;
  cplx *dst = a.out + i + (long long)q * a.inner;
#pragma unroll
  for (int m = 0; m < P; ++m) {
    const double kl = KX[q + m * TPL];
    const double Mbar = -((kl * kl + ky2) + kz2) * a.M;
    dst[(long long)m * TPL * a.inner] = make_double2(Mbar * v[m].x, Mbar * v[m].y);
  }
}

// c-hat = fft_x(.) ; ubar = (c-hat + (dt b0) Nhat + sum (dt b_i) Nhat_old_i) / (1 - dt Lbar) ; inverse x transform
// (AdamsBashforthMoulton.C:94-101; Lbar = k^2 k^2 kappa, ReciprocalLaplacianSquareFactor.C:28-32; the divide as libTorch evaluates it)
template <int N, int ORDER>
__global__ void __launch_bounds__(Plan<N>::NT, 2) k_x_update(XKArgs a, const cplx *__restrict__ tw) {
#pragma clang fp contract(off)
  constexpr int P = Plan<N>::P, TPL = Plan<N>::TPL, T = Plan<N>::T, NT = Plan<N>::NT, CNT = (N + NT - 1) / NT;
  using Map = MapStrided<N>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx *W = reinterpret_cast<cplx *>(smem);
  cplx *X = W + N;
  double *KX = reinterpret_cast<double *>(X + Map::size);
  const int l = threadIdx.x % T, q = threadIdx.x / T;
  const long long i = (long long)xcd_remap(blockIdx.x, gridDim.x) * T + l;
  const bool valid = i < a.inner;
  const long long ic = valid ? i : 0;
  cplx twv[CNT];
  double kxv[CNT];
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    const int idx = threadIdx.x + j * NT;
    twv[j] = idx < N ? tw[idx] : make_double2(0.0, 0.0);
    kxv[j] = idx < N ? a.kx[idx] : 0.0;
  }
  const double ky = a.ky[ic / a.nzc], kz = a.kz[ic % a.nzc];
  cplx v[P];
  const long long e0 = ic + (long long)q * a.inner, step = (long long)TPL * a.inner;
#pragma unroll
  for (int m = 0; m < P; ++m) v[m] = a.in[e0 + m * step];
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    const int idx = threadIdx.x + j * NT;
    if (idx < N) {
      W[idx] = twv[j];
      KX[idx] = kxv[j];
    }
  }
  fft_line<N, Map>(v, q, l, X, W);
  if (a.cbar && valid) {
#pragma unroll
    for (int m = 0; m < P; ++m) a.cbar[e0 + m * step] = v[m];
  }
  const double ky2 = ky * ky, kz2 = kz * kz;
  // the history streams through in thirds of the line: 10 x (1 + ORDER) values in flight instead of 30 x (1 + ORDER)
  // (quarters for the 20-point plans)
  constexpr int NPART = P % 3 == 0 ? 3 : 4, H = P / NPART;
  static_assert(NPART * H == P, "the history parts must cover the line");
#pragma unroll
  for (int part = 0; part < NPART; ++part) {
    cplx nn[H], no[ORDER > 0 ? ORDER : 1][H];
#pragma unroll
    for (int j = 0; j < H; ++j) nn[j] = a.Nnew[e0 + (part * H + j) * step];
#pragma unroll
    for (int hh = 0; hh < ORDER; ++hh) {
#pragma unroll
      for (int j = 0; j < H; ++j) no[hh][j] = a.Nold[hh][e0 + (part * H + j) * step];
    }
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const int m = part * H + j;
      cplx u = v[m];
      u.x = u.x + a.coef[0] * nn[j].x;
      u.y = u.y + a.coef[0] * nn[j].y;
#pragma unroll
      for (int hh = 0; hh < ORDER; ++hh) {
        u.x += a.coef[hh + 1] * no[hh][j].x;
        u.y += a.coef[hh + 1] * no[hh][j].y;
      }
      const double kl = KX[q + m * TPL];
      const double k2 = (kl * kl + ky2) + kz2;
      const double scl = 1.0 / (1.0 - a.dt * (k2 * k2 * a.kappa));
      v[m] = make_double2(u.y * scl, u.x * scl);  // swapped for the inverse transform
    }
  }
  fft_line<N, Map>(v, q, l, X, W);
  if (!valid) return;
#pragma unroll
  for (int m = 0; m < P; ++m) a.out[e0 + m * step] = cswap(v[m]);
}

template <int N>
static int launch_x_mbar(mrl_ctx *ctx, const XKArgs &a, const cplx *tw) {
  static std::atomic<bool> attr{false};  // (two host threads may both set the attribute: harmless, and no torn flag)
  constexpr size_t lds = lds_strided<N>() + sizeof(double) * N;
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY(set_lds_attr(ctx, k_x_mbar<N>, lds));
    attr.store(true, std::memory_order_release);
  }
  const long long nb = (a.inner + Plan<N>::T - 1) / Plan<N>::T;
  hipLaunchKernelGGL((k_x_mbar<N>), dim3((unsigned)nb), dim3(Plan<N>::NT), lds, ctx->stream, a, tw);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

template <int N, int ORDER>
static int launch_x_update(mrl_ctx *ctx, const XKArgs &a, const cplx *tw) {
  static std::atomic<bool> attr{false};  // (two host threads may both set the attribute: harmless, and no torn flag)
  constexpr size_t lds = lds_strided<N>() + sizeof(double) * N;
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY((set_lds_attr(ctx, k_x_update<N, ORDER>, lds)));
    attr.store(true, std::memory_order_release);
  }
  const long long nb = (a.inner + Plan<N>::T - 1) / Plan<N>::T;
  hipLaunchKernelGGL((k_x_update<N, ORDER>), dim3((unsigned)nb), dim3(Plan<N>::NT), lds, ctx->stream, a, tw);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

}  // namespace p2

struct PGeo {
  long long nx, ny, nz, nzc;
  const cplx *tw_x, *tw_y;
};
static PGeo pgeo(const mrl_ctx *ctx) {
  PGeo g;
  const int ax = ctx->dim == 3 ? 0 : 1;  // 2-D grids run as [nx][1][ny'] (serial contexts right-align the user axes)
  g.nx = ctx->n[ax];
  g.ny = ctx->dim == 3 ? ctx->n[1] : 1;
  g.nz = ctx->n[2];
  g.nzc = ctx->nrec[2];
  g.tw_x = ctx->ax[ax].d_tw;
  g.tw_y = ctx->ax[1].d_tw;
  return g;
}

// serial half-spectrum contexts whose extents all have plans for the plain transforms and at least one of which needs this path
bool planned_unfused_ok(const mrl_ctx *ctx) {
  if (ctx->slab || ctx->pencil || ctx->spectrum != MRL_SPECTRUM_HALF || (ctx->dim != 2 && ctx->dim != 3)) return false;
  if (ctx->exp & 2048) return false;  // experiment: planned shapes through the any-length path (A/B and parity, as fast_path_ok)
  const PGeo g = pgeo(ctx);
  if ((g.nx * g.ny) % 2) return false;  // the z passes carry two lines per complex transform
  const bool all = plain_ok(g.nx) && (g.ny == 1 || plain_ok(g.ny)) && plain_ok(g.nz);
  const bool any30 = plain30_ok(g.nx) || plain30_ok(g.ny) || plain30_ok(g.nz);
  return all && any30;
}

static int pass_axis(mrl_ctx *ctx, const PGeo &g, int axis, bool inv, int nf, cplx *a0, cplx *a1) {
  if (axis == 1 && g.ny == 1) return MRL_OK;
  p2::PassArgs a{};
  a.in[0] = a0;
  a.in[1] = a1;
  a.out[0] = a0;
  a.out[1] = a1;
  a.scale = 1.0;
  if (axis == 1) {
    a.inner = g.nzc;
    a.outer = g.nx;
    a.so_in = a.so_out = g.ny * g.nzc;
    a.sn_in = a.sn_out = g.nzc;
  } else {
    a.inner = g.ny * g.nzc;
    a.outer = 1;
    a.sn_in = a.sn_out = g.ny * g.nzc;
  }
  return pass_launch(ctx, axis == 1 ? g.ny : g.nx, inv, nf, a, axis == 1 ? g.tw_y : g.tw_x);
}

int fft_forward_planned(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch) {
  const PGeo g = pgeo(ctx);
  const long long nreal = g.nx * g.ny * g.nz, nspec = g.nx * g.ny * g.nzc;
  p2::ChDev none{};
  for (long long b = 0; b < batch; ++b) {
    cplx *out = reinterpret_cast<cplx *>(d_out) + b * nspec;
    {
      ProfScope ps(ctx, "z_fwd_pair", 8.0 * nreal + 16.0 * nspec);
      MRL_TRY(z_fwd_launch(ctx, g.nz, 0, 0, d_in + b * nreal, out, nullptr, nullptr, none, g.nx * g.ny / 2));
    }
    {
      ProfScope ps(ctx, "pass_y", 32.0 * nspec);
      MRL_TRY(pass_axis(ctx, g, 1, false, 1, out, nullptr));
    }
    ProfScope ps(ctx, "pass_x", 32.0 * nspec);
    MRL_TRY(pass_axis(ctx, g, 0, false, 1, out, nullptr));
  }
  return MRL_OK;
}

int fft_inverse_planned(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch) {
  const PGeo g = pgeo(ctx);
  const long long nreal = g.nx * g.ny * g.nz, nspec = g.nx * g.ny * g.nzc;
  const double scale = 1.0 / ((double)g.nx * (double)g.ny * (double)g.nz);
  MRL_TRY(ensure_work(ctx, 0, sizeof(cplx) * nspec));
  cplx *w = reinterpret_cast<cplx *>(ctx->d_work[0]);
  for (long long b = 0; b < batch; ++b) {
    MRL_HIP(ctx, hipMemcpyAsync(w, reinterpret_cast<const cplx *>(d_in) + b * nspec, sizeof(cplx) * nspec, hipMemcpyDeviceToDevice, ctx->stream));
    {
      ProfScope ps(ctx, "pass_x", 32.0 * nspec);
      MRL_TRY(pass_axis(ctx, g, 0, true, 1, w, nullptr));
    }
    {
      ProfScope ps(ctx, "pass_y", 32.0 * nspec);
      MRL_TRY(pass_axis(ctx, g, 1, true, 1, w, nullptr));
    }
    ProfScope ps(ctx, "z_inv_pair", 8.0 * nreal + 16.0 * nspec);
    MRL_TRY(z_inv_launch(ctx, g.nz, w, d_out + b * nreal, scale, g.nx * g.ny / 2));
  }
  return MRL_OK;
}

// forward y of both fields -> x passes with the k-space update (AdamsBashforthMoulton.C:94-101) -> inverse y: what lies between the
// forward z pass of a substep and the inverse z pass that ends it; w_c holds ubar (x- and y-inverted) afterwards
static int planned_kspace_passes(mrl_ctx *ctx, const PGeo &g, const ChP &cp, cplx *w_c, cplx *w_mu, double *Nhat_new,
                                 const double *const *Nhat_old, int order, double sub_dt, double *cbar) {
  const long long nspec = g.nx * g.ny * g.nzc;
  const double h = 16.0 * nspec;
  {
    ProfScope ps(ctx, "chp_B_y_fwd", 4.0 * h);
    MRL_TRY(pass_axis(ctx, g, 1, false, 2, w_c, w_mu));
  }
  if (two_stage_x(ctx, g.nx) && 16.0 * (double)nspec < 4294967296.0) {
    // one kernel: forward x of both fields, the k-space update, inverse x (5 h at AB2, as k_ch_xfused on the fused path)
    const int ax = ctx->dim == 3 ? 0 : 1;
    p2::X2Args a{};
    a.chat = w_c;
    a.muhat = w_mu;
    a.ubar = w_c;
    a.Nnew = reinterpret_cast<cplx *>(Nhat_new);
    a.cbar = reinterpret_cast<cplx *>(cbar);
    for (int i = 0; i < order; ++i) a.Nold[i] = reinterpret_cast<const cplx *>(Nhat_old[i]);
    for (int i = 0; i <= order; ++i) a.coef[i] = sub_dt * kBetaAB[order][i];
    a.M = cp.M;
    a.kappa = cp.kappa;
    a.dt = sub_dt;
    a.inner = g.ny * g.nzc;
    a.plane = a.inner;
    a.nzc = (int)g.nzc;
    a.kx = ctx->d_k[ax];
    a.ky = ctx->dim == 3 ? ctx->d_k[1] : ctx->d_k[0];  // 2-D: the unused axis {0}
    a.kz = ctx->d_k[2];
    ProfScope ps(ctx, "chp_CD_x_fused", (4.0 + order + (cbar ? 1.0 : 0.0)) * h);
    switch (order) {
      case 0: MRL_SWITCH_N2X(g.nx, MRL_TRY((p2::launch_xfused2<NN, 0>(ctx, a, g.tw_x)))); break;
      case 1: MRL_SWITCH_N2X(g.nx, MRL_TRY((p2::launch_xfused2<NN, 1>(ctx, a, g.tw_x)))); break;
      case 2: MRL_SWITCH_N2X(g.nx, MRL_TRY((p2::launch_xfused2<NN, 2>(ctx, a, g.tw_x)))); break;
      case 3: MRL_SWITCH_N2X(g.nx, MRL_TRY((p2::launch_xfused2<NN, 3>(ctx, a, g.tw_x)))); break;
      default: MRL_SWITCH_N2X(g.nx, MRL_TRY((p2::launch_xfused2<NN, 4>(ctx, a, g.tw_x)))); break;
    }
  } else if (plain30_ok(g.nx)) {
    // the k-space update rides on the x passes: mu-hat -> Nhat in one pass, c-hat -> ubar -> inverse x in a second one (6 h
    // instead of 11 h for the three x passes + the k-space kernel)
    const int ax = ctx->dim == 3 ? 0 : 1;
    p2::XKArgs a{};
    a.inner = g.ny * g.nzc;
    a.nzc = (int)g.nzc;
    a.kx = ctx->d_k[ax];
    a.ky = ctx->dim == 3 ? ctx->d_k[1] : ctx->d_k[0];  // 2-D: the unused axis {0}
    a.kz = ctx->d_k[2];
    a.M = cp.M;
    a.kappa = cp.kappa;
    a.dt = sub_dt;
    for (int i = 0; i <= order; ++i) a.coef[i] = sub_dt * kBetaAB[order][i];
    {
      ProfScope ps(ctx, "chp_C_x_mbar", 2.0 * h);
      a.in = w_mu;
      a.out = reinterpret_cast<cplx *>(Nhat_new);
      MRL_SWITCH_N30(g.nx, MRL_TRY((p2::launch_x_mbar<NN>(ctx, a, g.tw_x))));
    }
    ProfScope ps(ctx, "chp_D_x_update", (3.0 + order + (cbar ? 1.0 : 0.0)) * h);
    a.in = w_c;
    a.out = w_c;
    a.Nnew = reinterpret_cast<const cplx *>(Nhat_new);
    for (int i = 0; i < order; ++i) a.Nold[i] = reinterpret_cast<const cplx *>(Nhat_old[i]);
    a.cbar = reinterpret_cast<cplx *>(cbar);
    switch (order) {
      case 0: MRL_SWITCH_N30(g.nx, MRL_TRY((p2::launch_x_update<NN, 0>(ctx, a, g.tw_x)))); break;
      case 1: MRL_SWITCH_N30(g.nx, MRL_TRY((p2::launch_x_update<NN, 1>(ctx, a, g.tw_x)))); break;
      case 2: MRL_SWITCH_N30(g.nx, MRL_TRY((p2::launch_x_update<NN, 2>(ctx, a, g.tw_x)))); break;
      case 3: MRL_SWITCH_N30(g.nx, MRL_TRY((p2::launch_x_update<NN, 3>(ctx, a, g.tw_x)))); break;
      default: MRL_SWITCH_N30(g.nx, MRL_TRY((p2::launch_x_update<NN, 4>(ctx, a, g.tw_x)))); break;
    }
  } else {
    {
      ProfScope ps(ctx, "chp_C_x_fwd", 4.0 * h);
      MRL_TRY(pass_axis(ctx, g, 0, false, 2, w_c, w_mu));
    }
    if (cbar) MRL_HIP(ctx, hipMemcpyAsync(cbar, w_c, sizeof(cplx) * nspec, hipMemcpyDeviceToDevice, ctx->stream));
    {
      ProfScope ps(ctx, "chp_K_kspace", (4.0 + order) * h);
      MRL_TRY(ch_kspace_launch(ctx, cp, reinterpret_cast<const double *>(w_c), reinterpret_cast<const double *>(w_mu), Nhat_new,
                               reinterpret_cast<double *>(w_c), Nhat_old, order, sub_dt));  // elementwise: ubar overwrites c-hat
    }
    ProfScope ps(ctx, "chp_D_x_inv", 2.0 * h);
    MRL_TRY(pass_axis(ctx, g, 0, true, 1, w_c, nullptr));
  }
  {
    ProfScope ps(ctx, "chp_E_y_inv", 2.0 * h);
    MRL_TRY(pass_axis(ctx, g, 1, true, 1, w_c, nullptr));
  }
  return MRL_OK;
}

// one AdamsBashforthMoulton::substep with its compute group (AdamsBashforthMoulton.C:60-101); a parsed free energy is compiled into
// the forward z pass at run time as on the fused path (expr.hip: parsed_z_fwd_launch); no spectral carry-over
int ch_substep_planned(mrl_ctx *ctx, const ChP &cp, const double *c_in, double *c_out, double *Nhat_new, const double *const *Nhat_old,
                       int order, double sub_dt, double *cbar, double *mu, int carry) {
  if (!planned_unfused_ok(ctx) || carry != MRL_CARRY_NONE) return MRL_ERR_UNSUPPORTED;
  const PGeo g = pgeo(ctx);
  const long long nreal = g.nx * g.ny * g.nz, nspec = g.nx * g.ny * g.nzc;
  MRL_TRY(ensure_work(ctx, 1, sizeof(cplx) * nspec));
  MRL_TRY(ensure_work(ctx, 2, sizeof(cplx) * nspec));
  cplx *w_c = reinterpret_cast<cplx *>(ctx->d_work[1]);
  cplx *w_mu = reinterpret_cast<cplx *>(ctx->d_work[2]);
  p2::ChDev chp{cp.family, cp.c0, cp.c1, cp.c2, {}};
  const double h = 16.0 * nspec;
  {
    ProfScope ps(ctx, "chp_A_z_fwd", 8.0 * nreal + 2.0 * h + (mu ? 8.0 * nreal : 0.0));
    if (cp.family == MRL_FE_PARSED) {
      MRL_TRY(parsed_z_fwd_launch(ctx, cp.parsed, (int)g.nz, 1, c_in, w_c, w_mu, mu, g.nx * g.ny));
    } else {
      MRL_TRY(z_fwd_launch(ctx, g.nz, 1, cp.family, c_in, w_c, w_mu, mu, chp, g.nx * g.ny));
    }
  }
  MRL_TRY(planned_kspace_passes(ctx, g, cp, w_c, w_mu, Nhat_new, Nhat_old, order, sub_dt, cbar));
  ProfScope ps(ctx, "chp_F_z_inv", h + 8.0 * nreal);
  const double scale = 1.0 / ((double)g.nx * (double)g.ny * (double)g.nz);
  return z_inv_launch(ctx, g.nz, w_c, c_out, scale, g.nx * g.ny / 2);
}


// The substep loop of TensorSolver::computeBuffer (TensorSolver.C:95-108) in one call for grids whose z extent has a two-stage plan
// (fft_two_z.h) and a built-in free energy: between two substeps the inverse z pass and the next forward z pass are one kernel and
// the real field never reaches HBM (ch_substeps_fused of ch_fused.hip, for the planned path): 14 h of traffic per AB2 substep.
int ch_substeps_planned(mrl_ctx *ctx, const ChP &cp, const double *c_in, double *c_out, double *const *ring, int ring_size, int *head,
                        int *n_old, int pred, int count, int advance, double sub_dt, double *mu, bool dt_changed) {
  if (!planned_unfused_ok(ctx)) return MRL_ERR_UNSUPPORTED;
  const PGeo g = pgeo(ctx);
  // a fused inverse + forward z kernel exists for the two-stage lengths and for the lengths of the fused family (a grid like
  // 240 x 240 x 256 is on this path because of its x and y extents); the uniform 30- / 20-point z plans have none
  const bool z_two = two_stage(ctx, g.nz), z_std = !z_two && pow2_ok(g.nz);
  if (!z_two && !z_std) return MRL_ERR_UNSUPPORTED;
  const long long nreal = g.nx * g.ny * g.nz, nspec = g.nx * g.ny * g.nzc;
  MRL_TRY(ensure_work(ctx, 1, sizeof(cplx) * nspec));
  MRL_TRY(ensure_work(ctx, 2, sizeof(cplx) * nspec));
  cplx *w_c = reinterpret_cast<cplx *>(ctx->d_work[1]);
  cplx *w_mu = reinterpret_cast<cplx *>(ctx->d_work[2]);
  p2::ChDev chp{cp.family, cp.c0, cp.c1, cp.c2, {}};
  const double h = 16.0 * nspec;
  const double scale = 1.0 / ((double)g.nx * (double)g.ny * (double)g.nz);
  for (int k = 0; k < count; ++k) {
    double *mu_k = (k == count - 1) ? mu : nullptr;   // the buffer `mu` holds f'(c) of the last substep's input field
    if (k == 0) {
      ProfScope ps(ctx, "chp_A_z_fwd", 8.0 * nreal + 2.0 * h + (mu_k ? 8.0 * nreal : 0.0));
      if (cp.family == MRL_FE_PARSED) {
        MRL_TRY(parsed_z_fwd_launch(ctx, cp.parsed, (int)g.nz, 1, c_in, w_c, w_mu, mu_k, g.nx * g.ny));
      } else {
        MRL_TRY(z_fwd_launch(ctx, g.nz, 1, cp.family, c_in, w_c, w_mu, mu_k, chp, g.nx * g.ny));
      }
    } else {
      ProfScope ps(ctx, "chp_FA_z_inv_fwd", 3.0 * h + (mu_k ? 8.0 * nreal : 0.0));
      if (cp.family == MRL_FE_PARSED) {
        MRL_TRY(parsed_z_inv_fwd_launch(ctx, cp.parsed, (int)g.nz, w_c, w_c, w_mu, mu_k, scale, g.nx * g.ny / 2, false));
      } else if (z_std) {
        MRL_TRY(z_inv_fwd_launch_std(ctx, g.nz, cp.family, w_c, w_c, w_mu, mu_k, chp, scale, g.nx * g.ny / 2));
      } else if (cp.family == MRL_FE_DOUBLE_WELL) {
        MRL_SWITCH_N2(g.nz, MRL_TRY((p2::launch_z_inv_fwd2<NN, MRL_FE_DOUBLE_WELL>(ctx, w_c, w_c, w_mu, mu_k, chp, scale, g.nx * g.ny / 2))));
      } else {
        MRL_SWITCH_N2(g.nz, MRL_TRY((p2::launch_z_inv_fwd2<NN, MRL_FE_PFHUB>(ctx, w_c, w_c, w_mu, mu_k, chp, scale, g.nx * g.ny / 2))));
      }
    }
    const int order = (dt_changed && k < pred) ? 0 : (*n_old < pred ? *n_old : pred);   // AdamsBashforthMoulton.C:90-91
    const int slot_new = (*head + 1) % ring_size;
    const double *old[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < order; ++i) old[i] = ring[((*head - i) % ring_size + ring_size) % ring_size];
    MRL_TRY(planned_kspace_passes(ctx, g, cp, w_c, w_mu, ring[slot_new], old, order, sub_dt, nullptr));
    if (advance && k < count - 1) {   // TensorSolver.C:105-106
      *head = slot_new;
      if (*n_old < pred) *n_old += 1;
    }
  }
  ProfScope ps(ctx, "chp_F_z_inv", h + 8.0 * nreal);
  return z_inv_launch(ctx, g.nz, w_c, c_out, scale, g.nx * g.ny / 2);
}

}  // namespace mrl
