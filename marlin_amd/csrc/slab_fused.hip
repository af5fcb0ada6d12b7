// Fast path of the slab-decomposed Cahn-Hilliard substep (3-D, power-of-two extents, r2c on z, equal
// partitions).  Same kernels as the serial fast path (fft_pow2_kernels.h); the only kernel that sees
// the exchange layout is the fused y pass:
//   A  k_z_fwd<CH>      c -> (c-hat_z, mu-hat_z) on the real slab [nx][nyl][nzc], written into the send buffer
//      k_pass<x>        forward x, in place: the result is already ordered by destination rank
//   -- exchange (per field) --
//   B  k_ch_yfused      gathers lines along y from the received chunks [p][nxl][nyl][nzc], forward y on both
//                       fields, Nhat = Mbar*mu-hat (dense, reference layout), ABM predictor, 1/(1-dt*Lbar),
//                       inverse y, scattered back into the same chunk layout for the inverse exchange
//   -- exchange --
//   C  k_pass<x>        inverse x on the dense [nx][nyl][nzc] array that arrived
//      k_z_inv          c2r along z, 1/N
// (AdamsBashforthMoulton.C:60-101 with DomainAction::fftSlab/ifftSlab, DomainAction.C:869-1019.)
#include "ch_fused_body.h"
#include "fft_pow2_launch.h"

namespace mrl {

namespace p2 {

struct YFusedArgs {
  FusedCommon c;      // chat/muhat/ubar in the exchange layout [p][nxl][nyl][nzc]; Nnew/cbar/Nold dense [nxl][ny][nzc]
  int nxl, nzc;
  int nyl_shift;      // log2(ny / P)
  long long chunk;    // nxl * nyl * nzc: elements of one chunk
  int tiles_per_x;
  const double *kx, *ky, *kz;  // local reciprocal axes
};

template <int N, int ORDER>
__global__ void __launch_bounds__(256, 2) k_ch_yfused(YFusedArgs a, const cplx *__restrict__ tw) {
  constexpr int TPL = N / 16, T = 4096 / N;
  using Map = MapStrided<N>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx *W = reinterpret_cast<cplx *>(smem);
  cplx *X = W + N;
  double *KY = reinterpret_cast<double *>(X + Map::size);
  const int l = threadIdx.x % T, q = threadIdx.x / T;
  const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
  const int ix = logical / a.tiles_per_x;
  const int kz0 = (logical % a.tiles_per_x) * T + l;
  const bool valid = kz0 < a.nzc;
  const int kzi = valid ? kz0 : 0;
  // element (ix, j, kz): exchange layout  p*chunk + (ix*nyl + (j - p*nyl))*nzc + kz,  p = j >> nyl_shift
  //                      dense layout     (ix*N + j)*nzc + kz
  const unsigned nzcB = (unsigned)a.nzc * 16u, chunkB = (unsigned)a.chunk * 16u, kzB = (unsigned)kzi * 16u;
  const int sh = a.nyl_shift, msk = (1 << sh) - 1;
  auto offc = [=](int m) {
    const int j = q + m * TPL;
    return (unsigned)(j >> sh) * chunkB + (unsigned)((ix << sh) + (j & msk)) * nzcB + kzB;
  };
  const unsigned d0 = (unsigned)(((long long)ix * N + q) * a.nzc + kzi) * 16u, dstep = (unsigned)(TPL * a.nzc) * 16u;
  auto offd = [=](int m) { return d0 + (unsigned)m * dstep; };
  ch_fused_body<N, ORDER, false>(a.c, tw, a.ky, a.kx + ix, a.kz + kzi, valid, q, l, offc, offd, W, X, KY);
}

template <int N, int ORDER>
static int launch_yfused(mrl_ctx *ctx, YFusedArgs a) {
  static bool attr = false;
  constexpr size_t lds = lds_strided<N>() + sizeof(double) * N;
  if (!attr) {
    MRL_TRY(set_lds_attr(ctx, k_ch_yfused<N, ORDER>, lds));
    attr = true;
  }
  constexpr int T = 4096 / N;
  a.tiles_per_x = (a.nzc + T - 1) / T;
  const long long nb = (long long)a.nxl * a.tiles_per_x;
  hipLaunchKernelGGL((k_ch_yfused<N, ORDER>), dim3((unsigned)nb), dim3(256), lds, ctx->stream, a, ctx->ax[1].d_tw);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

}  // namespace p2

int slab_fast_ok(const mrl_ctx *ctx) {
  if (!(ctx->dim == 3 && ctx->nranks > 1 && ctx->spectrum == MRL_SPECTRUM_HALF && pow2_ok(ctx->n[0]) &&
        pow2_ok(ctx->n[1]) && pow2_ok(ctx->n[2])))
    return 0;
  // equal power-of-two partitions: chunk addressing by shifts in k_ch_yfused
  const long long nyl = ctx->n[1] / ctx->nranks, nxl = ctx->n[0] / ctx->nranks;
  if (nyl * ctx->nranks != ctx->n[1] || nxl * ctx->nranks != ctx->n[0] || (nyl & (nyl - 1))) return 0;
  for (int p = 0; p < ctx->nranks; ++p)
    if (ctx->part_real[p] != nyl || ctx->part_recip[p] != nxl) return 0;
  return 1;
}

static int xpass(mrl_ctx *ctx, bool inv, const cplx *in, cplx *out) {
  const long long nx = ctx->n[0], nyl = ctx->nloc[1], nzc = ctx->nrec[2];
  p2::PassArgs a{};
  a.in[0] = in;
  a.out[0] = out;
  a.scale = 1.0;
  a.inner = nyl * nzc;
  a.outer = 1;
  a.sn_in = a.sn_out = nyl * nzc;
  const cplx *tw = ctx->ax[0].d_tw;
  if (inv) {
    MRL_SWITCH_N(nx, MRL_TRY((p2::launch_pass_t<NN, true, 1>(ctx, a, tw))));
  } else {
    MRL_SWITCH_N(nx, MRL_TRY((p2::launch_pass_t<NN, false, 1>(ctx, a, tw))));
  }
  return MRL_OK;
}

int slab_ch_fwd_local_fast(mrl_ctx *ctx, const ChP &cp, const double *c_in, double *send, double *mu, int part) {
  const long long nx = ctx->n[0], nyl = ctx->nloc[1], nz = ctx->n[2], nzc = ctx->nrec[2];
  const long long nfield = nx * nyl * nzc;
  cplx *s_c = reinterpret_cast<cplx *>(send), *s_mu = s_c + nfield;
  const double h = 16.0 * nfield;
  if (part != 1) {
    {
      ProfScope ps(ctx, "slab_A_z_fwd", 8.0 * nx * nyl * nz + 2.0 * h + (mu ? 8.0 * nx * nyl * nz : 0.0));
      p2::ChDev chp{cp.family, cp.c0, cp.c1, cp.c2};
      if (cp.family == MRL_FE_DOUBLE_WELL) {
        MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_fwd<NN, 1, MRL_FE_DOUBLE_WELL>(ctx, c_in, s_c, s_mu, mu, chp, nx * nyl))));
      } else {
        MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_fwd<NN, 1, MRL_FE_PFHUB>(ctx, c_in, s_c, s_mu, mu, chp, nx * nyl))));
      }
    }
    ProfScope ps(ctx, "slab_A_x_fwd", 2.0 * h);
    MRL_TRY(xpass(ctx, false, s_c, s_c));
  }
  if (part != 0) {
    ProfScope ps(ctx, "slab_A_x_fwd", 2.0 * h);
    MRL_TRY(xpass(ctx, false, s_mu, s_mu));
  }
  return MRL_OK;
}

int slab_ch_kspace_fast(mrl_ctx *ctx, const ChP &cp, const double *recv, double *send, double *Nhat_new,
                        const double *const *Nhat_old, int order, double sub_dt, double *cbar) {
  const long long nxl = ctx->nrec[0], ny = ctx->n[1], nzc = ctx->nrec[2];
  const long long nspec = nxl * ny * nzc;
  const long long nyl = ny / ctx->nranks;
  p2::YFusedArgs a{};
  a.c.chat = reinterpret_cast<const cplx *>(recv);
  a.c.muhat = a.c.chat + nspec;
  a.c.ubar = reinterpret_cast<cplx *>(send);
  a.c.Nnew = reinterpret_cast<cplx *>(Nhat_new);
  a.c.cbar = reinterpret_cast<cplx *>(cbar);
  for (int i = 0; i < order; ++i) a.c.Nold[i] = reinterpret_cast<const cplx *>(Nhat_old[i]);
  for (int i = 0; i <= order; ++i) a.c.coef[i] = sub_dt * kBetaAB[order][i];
  a.nxl = (int)nxl;
  a.nzc = (int)nzc;
  a.nyl_shift = 0;
  while ((1LL << a.nyl_shift) < nyl) ++a.nyl_shift;
  a.chunk = nxl * nyl * nzc;
  a.kx = ctx->d_k[0];
  a.ky = ctx->d_k[1];
  a.kz = ctx->d_k[2];
  a.c.M = cp.M;
  a.c.kappa = cp.kappa;
  a.c.dt = sub_dt;
  ProfScope ps(ctx, "slab_B_y_fused", (4.0 + order + (cbar ? 1.0 : 0.0)) * 16.0 * nspec);
  switch (order) {
    case 0: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused<NN, 0>(ctx, a)))); break;
    case 1: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused<NN, 1>(ctx, a)))); break;
    case 2: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused<NN, 2>(ctx, a)))); break;
    case 3: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused<NN, 3>(ctx, a)))); break;
    default: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused<NN, 4>(ctx, a)))); break;
  }
  return MRL_OK;
}

int slab_inv_finish_fast(mrl_ctx *ctx, const double *recv, double *real_out) {
  const long long nx = ctx->n[0], nyl = ctx->nloc[1], nz = ctx->n[2], nzc = ctx->nrec[2];
  const long long nfield = nx * nyl * nzc;
  MRL_TRY(ensure_work(ctx, 11, sizeof(cplx) * nfield));
  cplx *w = reinterpret_cast<cplx *>(ctx->d_work[11]);
  {
    ProfScope ps(ctx, "slab_C_x_inv", 32.0 * nfield);
    MRL_TRY(xpass(ctx, true, reinterpret_cast<const cplx *>(recv), w));
  }
  ProfScope ps(ctx, "slab_C_z_inv", 16.0 * nfield + 8.0 * nx * nyl * nz);
  const double scale = 1.0 / ((double)nx * (double)ctx->n[1] * (double)nz);
  MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_inv<NN>(ctx, w, real_out, scale, nx * nyl / 2))));
  return MRL_OK;
}

}  // namespace mrl
