// fp32 instantiation of the fused serial Cahn-Hilliard path: mrl_ch_substeps_f32.
//
// The reference selects its floating-point precision per run (src/utils/MarlinUtils.C:39-44 -> floatTensorOptions, set up in
// src/actions/DomainAction.C:81,201), and the only GPU numbers it publishes are fp32 (Apple MPS, doc/content/installation.md:36-43).
// This translation unit compiles the SAME kernel templates as ch_fused.hip -- fft_pow2.h, fft_pow2_kernels.h, ch_fused_body.h,
// ch_xfused.h: same butterflies, same pointwise expressions in the same association -- with MRL_KREAL = float in their own namespace
// (p2f), so that the fp64 product path and its symbols are untouched.  Half the bytes per update on a bandwidth-bound path.
// Scope: serial 3-D contexts, extents in {64, 100, 128, 200, 256, 400, 512}, the built-in free-energy families, predictor order <= 3,
// arrays below 4 GiB; everything else returns MRL_ERR_UNSUPPORTED (the fp64 entry points cover it).  Never the headline: bench.py
// reports it as variants.fp32.
#define MRL_KREAL float
#define MRL_P2NS p2f
#include "ch_xfused.h"

namespace mrl {

int ch_check_params(mrl_ctx *ctx, const mrl_ch_params *p, ChP &cp);

#define MRL_SWITCH_F32(n, CALL)                          \
  switch (n) {                                           \
    case 64: { constexpr int NN = 64; CALL; } break;     \
    case 128: { constexpr int NN = 128; CALL; } break;   \
    case 256: { constexpr int NN = 256; CALL; } break;   \
    case 512: { constexpr int NN = 512; CALL; } break;   \
    case 100: { constexpr int NN = 100; CALL; } break;   \
    case 200: { constexpr int NN = 200; CALL; } break;   \
    case 400: { constexpr int NN = 400; CALL; } break;   \
    default: return MRL_ERR_UNSUPPORTED;                 \
  }

static bool f32_len_ok(long long n) { return n == 64 || n == 128 || n == 256 || n == 512 || n == 100 || n == 200 || n == 400; }

// x-plane pitch (complex64 elements) of the solver's spectral arrays: an odd number of 256-byte pieces (32 elements), DESIGN.md 3.1
static long long f32_plane(const mrl_ctx *ctx) {
  long long plane = (ctx->n[1] * ctx->nrec[2] + 31) / 32 * 32;
  if ((plane / 32) % 2 == 0) plane += 32;
  return plane;
}

static bool f32_ok(const mrl_ctx *ctx) {
  return !ctx->slab && !ctx->pencil && ctx->dim == 3 && ctx->spectrum == MRL_SPECTRUM_HALF && f32_len_ok(ctx->n[0]) && f32_len_ok(ctx->n[1]) &&
         f32_len_ok(ctx->n[2]) && 8.0 * (double)ctx->n[0] * (double)f32_plane(ctx) < 4294967296.0;
}

static int f32_pass_y(mrl_ctx *ctx, bool inv, int nf, p2f::kcplx *a0, p2f::kcplx *a1, long long plane, bool reverse) {
  const long long nx = ctx->n[0], ny = ctx->n[1], nzc = ctx->nrec[2];
  p2f::PassArgs a{};
  a.in[0] = a0;
  a.in[1] = a1;
  a.out[0] = a0;
  a.out[1] = a1;
  a.scale = 1.0f;
  a.reverse = reverse ? 1 : 0;
  a.inner = nzc;
  a.outer = nx;
  a.so_in = a.so_out = plane;
  a.sn_in = a.sn_out = nzc;
  const p2f::kcplx *tw = p2f::tw_table(ctx, 1);
  if (!tw) return MRL_ERR_HIP;
  if (nf == 2) {
    MRL_SWITCH_F32(ny, MRL_TRY((p2f::launch_pass_t<NN, false, 2>(ctx, a, tw))));
  } else if (inv) {
    MRL_SWITCH_F32(ny, MRL_TRY((p2f::launch_pass_t<NN, true, 1>(ctx, a, tw))));
  } else {
    MRL_SWITCH_F32(ny, MRL_TRY((p2f::launch_pass_t<NN, false, 1>(ctx, a, tw))));
  }
  return MRL_OK;
}

// the substep loop of TensorSolver::computeBuffer (TensorSolver.C:93-109) in fp32: ch_substeps_fused (ch_fused.hip) with float arrays
static int ch_substeps_fused_f32(mrl_ctx *ctx, const ChP &cp, const float *c_in, float *c_out, float *const *ring, int ring_size, int *head,
                                 int *n_old, int pred, int count, int advance, double sub_dt, bool dt_changed) {
  const long long nx = ctx->n[0], ny = ctx->n[1], nz = ctx->n[2], nzc = ctx->nrec[2], plane = f32_plane(ctx);
  const long long nreal = nx * ny * nz, nspec = nx * ny * nzc;
  const p2f::ZLay zl{(unsigned)ny, (unsigned)(plane - ny * nzc)};
  MRL_TRY(ensure_work(ctx, 21, sizeof(p2f::kcplx) * (size_t)(nx * plane)));
  MRL_TRY(ensure_work(ctx, 22, sizeof(p2f::kcplx) * (size_t)(nx * plane)));
  p2f::kcplx *w_c = reinterpret_cast<p2f::kcplx *>(ctx->d_work[21]);
  p2f::kcplx *w_mu = reinterpret_cast<p2f::kcplx *>(ctx->d_work[22]);
  for (int a = 0; a < 3; ++a) {
    MRL_TRY(axis_tw32(ctx, a));
    MRL_TRY(axis_k32(ctx, a));
  }
  const p2f::ChDev chp{cp.family, (float)cp.c0, (float)cp.c1, (float)cp.c2, {}};
  const double h = 8.0 * nspec;  // bytes of one complex64 half-spectrum array
  const float scale = (float)(1.0 / ((double)nx * (double)ny * (double)nz));
  for (int k = 0; k < count; ++k) {
    if (k == 0) {
      ProfScope ps(ctx, "ch32_A_z_fwd", 4.0 * nreal + 2.0 * h);
      if (cp.family == MRL_FE_DOUBLE_WELL) {
        MRL_SWITCH_F32(nz, MRL_TRY((p2f::launch_z_fwd<NN, 1, MRL_FE_DOUBLE_WELL>(ctx, c_in, w_c, w_mu, nullptr, chp, nx * ny, zl))));
      } else {
        MRL_SWITCH_F32(nz, MRL_TRY((p2f::launch_z_fwd<NN, 1, MRL_FE_PFHUB>(ctx, c_in, w_c, w_mu, nullptr, chp, nx * ny, zl))));
      }
    } else {
      ProfScope ps(ctx, "ch32_EA_z_inv_fwd", 3.0 * h);
      if (cp.family == MRL_FE_DOUBLE_WELL) {
        MRL_SWITCH_F32(nz, MRL_TRY((p2f::launch_z_inv_fwd<NN, MRL_FE_DOUBLE_WELL>(ctx, w_c, w_c, w_mu, nullptr, chp, scale, nx * ny / 2, zl))));
      } else {
        MRL_SWITCH_F32(nz, MRL_TRY((p2f::launch_z_inv_fwd<NN, MRL_FE_PFHUB>(ctx, w_c, w_c, w_mu, nullptr, chp, scale, nx * ny / 2, zl))));
      }
    }
    {
      ProfScope ps(ctx, "ch32_B_y_fwd", 4.0 * h);
      MRL_TRY(f32_pass_y(ctx, false, 2, w_c, w_mu, plane, true));
    }
    const int order = (dt_changed && k < pred) ? 0 : (*n_old < pred ? *n_old : pred);   // AdamsBashforthMoulton.C:90-91
    const int slot_new = (*head + 1) % ring_size;
    {
      ProfScope ps(ctx, "ch32_C_x_fused", (4.0 + order) * h);
      p2f::FusedArgs a{};
      a.c.chat = w_c;
      a.c.muhat = w_mu;
      a.c.ubar = w_c;
      a.c.Nnew = reinterpret_cast<p2f::kcplx *>(ring[slot_new]);
      for (int i = 0; i < order; ++i) a.c.Nold[i] = reinterpret_cast<const p2f::kcplx *>(ring[((*head - i) % ring_size + ring_size) % ring_size]);
      for (int i = 0; i <= order; ++i) a.c.coef[i] = (float)(sub_dt * kBetaAB[order][i]);
      a.inner = ny * nzc;
      a.plane = plane;
      a.nzc = (int)nzc;
      a.kx = ctx->ax[0].d_k32;
      a.ky = ctx->ax[1].d_k32;
      a.kz = ctx->ax[2].d_k32;
      a.c.M = (float)cp.M;
      a.c.kappa = (float)cp.kappa;
      a.c.dt = (float)sub_dt;
      const p2f::kcplx *tw = p2f::tw_table(ctx, 0);
      switch (order) {
        case 0: MRL_SWITCH_F32(nx, MRL_TRY((p2f::launch_xfused<NN, 0, false>(ctx, a, tw)))); break;
        case 1: MRL_SWITCH_F32(nx, MRL_TRY((p2f::launch_xfused<NN, 1, false>(ctx, a, tw)))); break;
        default: MRL_SWITCH_F32(nx, MRL_TRY((p2f::launch_xfused<NN, 2, false>(ctx, a, tw)))); break;
      }
    }
    {
      ProfScope ps(ctx, "ch32_D_y_inv", 2.0 * h);
      MRL_TRY(f32_pass_y(ctx, true, 1, w_c, nullptr, plane, false));
    }
    if (advance && k < count - 1) {   // TensorSolver.C:105-106
      *head = slot_new;
      if (*n_old < pred) *n_old += 1;
    }
  }
  ProfScope ps(ctx, "ch32_E_z_inv", h + 4.0 * nreal);
  MRL_SWITCH_F32(nz, MRL_TRY((p2f::launch_z_inv<NN>(ctx, w_c, c_out, scale, nx * ny / 2, zl))));
  return MRL_OK;
}

}  // namespace mrl

using namespace mrl;

extern "C" {

int64_t mrl_ch_spec_elems_f32(const mrl_ctx *ctx) { return ctx && f32_ok(ctx) ? ctx->n[0] * f32_plane(ctx) : 0; }

int mrl_ch_spec_layout_f32(const mrl_ctx *ctx, int64_t *plane_pitch, int64_t *row_pitch) {
  if (!ctx || !f32_ok(ctx)) return MRL_ERR_UNSUPPORTED;
  if (plane_pitch) *plane_pitch = f32_plane(ctx);
  if (row_pitch) *row_pitch = ctx->nrec[2];
  return MRL_OK;
}

int mrl_ch_substeps_f32(mrl_ctx *ctx, const mrl_ch_params *p, const float *d_c_in, float *d_c_out, float *const *d_Nhat_ring, int ring_size,
                        int *head, int *n_old, int predictor_order, int count, int advance, double sub_dt) {
  if (!ctx) return MRL_ERR_INVALID;
  ChP cp;
  MRL_TRY(ch_check_params(ctx, p, cp));
  const int pred = predictor_order - 1;
  const bool dt_changed = (advance & MRL_SUBSTEPS_DT_CHANGED) != 0;
  advance &= MRL_SUBSTEPS_ADVANCE;
  if (!d_c_in || !d_c_out || !d_Nhat_ring || !head || !n_old || count < 1) return set_error(ctx, MRL_ERR_INVALID, "mrl_ch_substeps_f32: bad argument");
  if (predictor_order < 1 || predictor_order > 3)
    return set_error(ctx, MRL_ERR_UNSUPPORTED, "mrl_ch_substeps_f32: predictor orders 1 ... 3 (the fp64 entry point has all five)");
  if (ring_size < pred + 1 || *head < 0 || *head >= ring_size || *n_old < 0 || *n_old > pred)
    return set_error(ctx, MRL_ERR_INVALID, "mrl_ch_substeps_f32: history ring of %d arrays needed (head %d, n_old %d)", pred + 1, *head, *n_old);
  for (int i = 0; i < ring_size; ++i)
    if (!d_Nhat_ring[i]) return set_error(ctx, MRL_ERR_INVALID, "history ring entry %d missing", i);
  if (cp.family == MRL_FE_PARSED) return set_error(ctx, MRL_ERR_UNSUPPORTED, "mrl_ch_substeps_f32: built-in free-energy families only");
  if (!f32_ok(ctx))
    return set_error(ctx, MRL_ERR_UNSUPPORTED, "mrl_ch_substeps_f32: serial 3-D contexts with extents in {64, 100, 128, 200, 256, 400, 512} only");
  const int rc = ch_substeps_fused_f32(ctx, cp, d_c_in, d_c_out, d_Nhat_ring, ring_size, head, n_old, pred, count, advance, sub_dt, dt_changed);
  if (rc == MRL_ERR_UNSUPPORTED) return set_error(ctx, rc, "mrl_ch_substeps_f32: shape not instantiated");
  return rc;
}

}  // extern "C"
