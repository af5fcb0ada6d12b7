// Slab-decomposed transforms (parallel_mode = FFT_SLAB), split at the global transpose:
//   DomainAction::partitionSlabs (src/actions/DomainAction.C:510-566),
//   fftSlab (:869-938), ifftSlab (:940-1019).
// Real space is split along y  ([nx][ny/P][nz] per rank), reciprocal space along x ([nx/P][ny][nzc]).
// Forward : z and x passes on the real slab -> exchange of [nx/P][ny/P][nzc] chunks -> y pass.
// Inverse : y pass -> exchange -> x and z passes.
// The x-pass output [nx][nyl][nzc] is already ordered by destination rank (x is the slowest index), and
// what arrives for the inverse is again a dense [nx][nyl][nzc] array, so only the y pass sees the chunked
// layout: [p][nxl][nyl_p][nzc].  The exchange itself belongs to the caller (RCCL all-to-all in
// marlin_amd/slab.py; MPI in the MOOSE shim) -- this file never communicates.
//
// Deviation from the reference, results identical: with MRL_SPECTRUM_HALF the z axis stays r2c in slab
// mode (the reference switches to a full c2c transform, DomainAction.C:279-281), which halves the
// exchanged volume.  MRL_SPECTRUM_FULL reproduces the reference's layout exactly.
#include "mrl_internal.h"

namespace mrl {

// generic passes (fft_plan.hip)
int pass_z_forward(mrl_ctx *ctx, const double *d_in, double *d_out, long long A0, long long A1, long long batch, int layout);
int pass_strided(mrl_ctx *ctx, int a, int sign, const double *d_in, double *d_out, long long A0, long long A1,
                 long long nzc, long long batch, int layout);
int pass_z_inverse(mrl_ctx *ctx, const double *d_in, double *d_out, long long A0, long long A1, long long batch,
                   int layout, double scale);
// Cahn-Hilliard pointwise pieces (ch.hip)
int ch_check_params(mrl_ctx *ctx, const mrl_ch_params *p, ChP &cp);
int ch_mu_launch(mrl_ctx *ctx, const ChP &cp, const double *c, double *mu, long long count);
int ch_kspace_launch(mrl_ctx *ctx, const ChP &cp, const double *cbar, const double *mubar, double *Nhat, double *ubar,
                     const double *const *Nold, int order, double sub_dt);

struct ChunkTab {
  int nranks;
  long long off[64];   // complex-element offset of chunk p in the chunked buffer
  long long ny_p[64];  // y extent of chunk p
  long long yb_p[64];  // first global y of chunk p
};

// dense [nxl][ny][nzc]  <->  chunked [p][nxl][ny_p][nzc]   (TO_CHUNKS: dense -> chunked)
template <bool TO_CHUNKS>
__global__ void __launch_bounds__(256) k_slab_repack(ChunkTab t, const double2 *__restrict__ in, double2 *__restrict__ out,
                                                      long long nxl, long long ny, long long nzc) {
  const long long total = nxl * ny * nzc;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long kz = e % nzc, r = e / nzc, j = r % ny, ix = r / ny;
    int p = 0;
    while (p + 1 < t.nranks && j >= t.yb_p[p + 1]) ++p;
    const long long c = t.off[p] + (ix * t.ny_p[p] + (j - t.yb_p[p])) * nzc + kz;
    if (TO_CHUNKS)
      out[c] = in[e];
    else
      out[e] = in[c];
  }
}

static void chunk_table(const mrl_ctx *ctx, ChunkTab &t) {
  t.nranks = ctx->nranks;
  const long long nxl = ctx->nrec[0], nzc = ctx->nrec[2];
  long long off = 0, yb = 0;
  for (int p = 0; p < ctx->nranks; ++p) {
    t.off[p] = off;
    t.ny_p[p] = ctx->part_real[p];
    t.yb_p[p] = yb;
    off += nxl * ctx->part_real[p] * nzc;
    yb += ctx->part_real[p];
  }
}

static int check_slab(mrl_ctx *ctx, const char *what) {
  if (ctx->nranks < 2) return set_error(ctx, MRL_ERR_INVALID, "%s: not a slab context (nranks = 1)", what);
  if (ctx->nranks > 64) return set_error(ctx, MRL_ERR_UNSUPPORTED, "%s: at most 64 ranks", what);
  return MRL_OK;
}

static int repack(mrl_ctx *ctx, bool to_chunks, const double *in, double *out) {
  ChunkTab t;
  chunk_table(ctx, t);
  const long long nxl = ctx->nrec[0], ny = ctx->n[1], nzc = ctx->nrec[2];
  const long long total = nxl * ny * nzc;
  long long nb = (total + 255) / 256;
  if (nb > 16384) nb = 16384;
  ProfScope ps(ctx, to_chunks ? "slab_pack" : "slab_unpack", 32.0 * (double)total);
  if (to_chunks)
    hipLaunchKernelGGL(k_slab_repack<true>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, t,
                       reinterpret_cast<const double2 *>(in), reinterpret_cast<double2 *>(out), nxl, ny, nzc);
  else
    hipLaunchKernelGGL(k_slab_repack<false>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, t,
                       reinterpret_cast<const double2 *>(in), reinterpret_cast<double2 *>(out), nxl, ny, nzc);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

// ---- the four generic stages ---------------------------------------------------------------------
int slab_fwd_local(mrl_ctx *ctx, const double *real_in, double *send) {
  const long long nx = ctx->n[0], nyl = ctx->nloc[1], nzc = ctx->nrec[2];
  {
    ProfScope ps(ctx, "slab_z_fwd", 8.0 * nx * nyl * ctx->n[2] + 16.0 * nx * nyl * nzc);
    MRL_TRY(pass_z_forward(ctx, real_in, send, nx, nyl, 1, 0));
  }
  ProfScope ps(ctx, "slab_x_fwd", 32.0 * nx * nyl * nzc);
  return pass_strided(ctx, 0, -1, send, send, nx, nyl, nzc, 1, 0);
}

int slab_fwd_finish(mrl_ctx *ctx, const double *recv, double *spec_out) {
  const long long nxl = ctx->nrec[0], ny = ctx->n[1], nzc = ctx->nrec[2];
  MRL_TRY(repack(ctx, false, recv, spec_out));
  ProfScope ps(ctx, "slab_y_fwd", 32.0 * nxl * ny * nzc);
  return pass_strided(ctx, 1, -1, spec_out, spec_out, nxl, ny, nzc, 1, 0);
}

int slab_inv_local(mrl_ctx *ctx, const double *spec_in, double *send) {
  const long long nxl = ctx->nrec[0], ny = ctx->n[1], nzc = ctx->nrec[2];
  MRL_TRY(ensure_work(ctx, 11, sizeof(cplx) * nxl * ny * nzc));
  {
    ProfScope ps(ctx, "slab_y_inv", 32.0 * nxl * ny * nzc);
    MRL_TRY(pass_strided(ctx, 1, +1, spec_in, ctx->d_work[11], nxl, ny, nzc, 1, 0));
  }
  return repack(ctx, true, ctx->d_work[11], send);
}

int slab_inv_finish(mrl_ctx *ctx, const double *recv, double *real_out) {
  const long long nx = ctx->n[0], nyl = ctx->nloc[1], nzc = ctx->nrec[2];
  MRL_TRY(ensure_work(ctx, 11, sizeof(cplx) * nx * nyl * nzc));
  {
    ProfScope ps(ctx, "slab_x_inv", 32.0 * nx * nyl * nzc);
    MRL_TRY(pass_strided(ctx, 0, +1, recv, ctx->d_work[11], nx, nyl, nzc, 1, 0));
  }
  const double scale = 1.0 / ((double)ctx->n[0] * (double)ctx->n[1] * (double)ctx->n[2]);
  ProfScope ps(ctx, "slab_z_inv", 8.0 * nx * nyl * ctx->n[2] + 16.0 * nx * nyl * nzc);
  return pass_z_inverse(ctx, ctx->d_work[11], real_out, nx, nyl, 1, 0, scale);
}

// fast path (slab_fused.hip): returns MRL_ERR_UNSUPPORTED when the shape has no fast kernels
int slab_fast_ok(const mrl_ctx *ctx);
int slab_ch_fwd_local_fast(mrl_ctx *ctx, const ChP &cp, const double *c_in, double *send, double *mu, int part);
int slab_ch_kspace_fast(mrl_ctx *ctx, const ChP &cp, const double *recv, double *send, double *Nhat_new,
                        const double *const *Nhat_old, int order, double sub_dt, double *cbar);
int slab_inv_finish_fast(mrl_ctx *ctx, const double *recv, double *real_out);

}  // namespace mrl

using namespace mrl;

extern "C" {

int mrl_slab_counts(const mrl_ctx *ctx, int forward, int64_t *h_send_counts, int64_t *h_recv_counts,
                    int64_t *h_send_offsets, int64_t *h_recv_offsets) {
  if (!ctx) return MRL_ERR_INVALID;
  if (ctx->nranks < 2) return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_counts: not a slab context");
  const long long nzc = ctx->nrec[2];
  long long so = 0, ro = 0;
  for (int p = 0; p < ctx->nranks; ++p) {
    // a chunk always pairs one rank's x range with the other rank's y range
    const long long to_p = forward ? ctx->part_recip[p] * ctx->nloc[1] * nzc : ctx->nrec[0] * ctx->part_real[p] * nzc;
    const long long from_p = forward ? ctx->nrec[0] * ctx->part_real[p] * nzc : ctx->part_recip[p] * ctx->nloc[1] * nzc;
    if (h_send_counts) h_send_counts[p] = to_p;
    if (h_recv_counts) h_recv_counts[p] = from_p;
    if (h_send_offsets) h_send_offsets[p] = so;
    if (h_recv_offsets) h_recv_offsets[p] = ro;
    so += to_p;
    ro += from_p;
  }
  return MRL_OK;
}

int mrl_slab_fwd_local(mrl_ctx *ctx, const double *d_real_in, double *d_send) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_TRY(check_slab(ctx, "mrl_slab_fwd_local"));
  if (!d_real_in || !d_send) return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_fwd_local: null buffer");
  return slab_fwd_local(ctx, d_real_in, d_send);
}

int mrl_slab_fwd_finish(mrl_ctx *ctx, const double *d_recv, double *d_spec_out) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_TRY(check_slab(ctx, "mrl_slab_fwd_finish"));
  if (!d_recv || !d_spec_out) return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_fwd_finish: null buffer");
  return slab_fwd_finish(ctx, d_recv, d_spec_out);
}

int mrl_slab_inv_local(mrl_ctx *ctx, const double *d_spec_in, double *d_send) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_TRY(check_slab(ctx, "mrl_slab_inv_local"));
  if (!d_spec_in || !d_send) return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_inv_local: null buffer");
  return slab_inv_local(ctx, d_spec_in, d_send);
}

int mrl_slab_inv_finish(mrl_ctx *ctx, const double *d_recv, double *d_real_out) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_TRY(check_slab(ctx, "mrl_slab_inv_finish"));
  if (!d_recv || !d_real_out) return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_inv_finish: null buffer");
  if (slab_fast_ok(ctx)) return slab_inv_finish_fast(ctx, d_recv, d_real_out);
  return slab_inv_finish(ctx, d_recv, d_real_out);
}

int mrl_slab_ch_fwd_local(mrl_ctx *ctx, const mrl_ch_params *p, const double *d_c_in, double *d_send, double *d_mu,
                          int part) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_TRY(check_slab(ctx, "mrl_slab_ch_fwd_local"));
  ChP cp;
  MRL_TRY(ch_check_params(ctx, p, cp));
  if (!d_c_in || !d_send) return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_ch_fwd_local: null buffer");
  if (part < -1 || part > 1) return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_ch_fwd_local: part must be -1, 0 or 1");
  if (slab_fast_ok(ctx)) return slab_ch_fwd_local_fast(ctx, cp, d_c_in, d_send, d_mu, part);
  const long long nreal = real_count_local(ctx);
  const long long nchunk = ctx->n[0] * ctx->nloc[1] * ctx->nrec[2];  // complex elements of one field's send buffer
  double *mu = d_mu;
  if (!mu) {
    MRL_TRY(ensure_work(ctx, 12, sizeof(double) * (nreal + 2)));
    mu = ctx->d_work[12];
  }
  if (part != 1) {
    MRL_TRY(ch_mu_launch(ctx, cp, d_c_in, mu, nreal));
    MRL_TRY(slab_fwd_local(ctx, d_c_in, d_send));
  }
  if (part != 0) MRL_TRY(slab_fwd_local(ctx, mu, d_send + 2 * nchunk));
  return MRL_OK;
}

int mrl_slab_ch_kspace(mrl_ctx *ctx, const mrl_ch_params *p, const double *d_recv, double *d_send, double *d_Nhat_new,
                       const double *const *d_Nhat_old, int order, double sub_dt, double *d_cbar) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_TRY(check_slab(ctx, "mrl_slab_ch_kspace"));
  ChP cp;
  MRL_TRY(ch_check_params(ctx, p, cp));
  if (!d_recv || !d_send || !d_Nhat_new) return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_ch_kspace: null buffer");
  if (order < 0 || order > 4) return set_error(ctx, MRL_ERR_INVALID, "predictor order %d out of range", order + 1);
  for (int i = 0; i < order; ++i)
    if (!d_Nhat_old || !d_Nhat_old[i]) return set_error(ctx, MRL_ERR_INVALID, "history entry %d missing", i);
  if (slab_fast_ok(ctx))
    return slab_ch_kspace_fast(ctx, cp, d_recv, d_send, d_Nhat_new, d_Nhat_old, order, sub_dt, d_cbar);
  const long long nspec = spec_count_local(ctx);
  MRL_TRY(ensure_work(ctx, 13, sizeof(cplx) * nspec));
  MRL_TRY(ensure_work(ctx, 14, sizeof(cplx) * nspec));
  MRL_TRY(ensure_work(ctx, 15, sizeof(cplx) * nspec));
  double *cbar = d_cbar ? d_cbar : ctx->d_work[13];
  double *mubar = ctx->d_work[14];
  MRL_TRY(slab_fwd_finish(ctx, d_recv, cbar));
  MRL_TRY(slab_fwd_finish(ctx, d_recv + 2 * nspec, mubar));
  double *ubar = ctx->d_work[15];
  {
    ProfScope ps(ctx, "ch_kspace", 16.0 * (double)nspec * (4 + order));
    MRL_TRY(ch_kspace_launch(ctx, cp, cbar, mubar, d_Nhat_new, ubar, d_Nhat_old, order, sub_dt));
  }
  return slab_inv_local(ctx, ubar, d_send);
}

}  // extern "C"
