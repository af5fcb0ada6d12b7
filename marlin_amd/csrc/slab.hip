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
#include "comm_dev.h"
#include "slab_stages.h"

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
  if (!ctx->slab) return set_error(ctx, MRL_ERR_INVALID, "%s: not a slab context (nranks = 1 without MRL_FLAG_SLAB)", what);
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

// ---- Cahn-Hilliard substep pipelined over kz sub-blocks -------------------------------------------------
// generic passes on a sub-range (fft_plan.hip), k-space update on a kz sub-range (ch.hip)
int pass_lines(mrl_ctx *ctx, int axis, int sign, const double *in, double *out, long long outer, long long inner,
               long long so, long long si, long long sn, int lines_fastest = 1);
int ch_kspace_sub_launch(mrl_ctx *ctx, const ChP &cp, const double *cbar, const double *mubar, double *Nhat, double *ubar,
                         const double *const *Nold, int order, double sub_dt, long long k0, long long ksub);

// kz sub-block s of nsub: DomainAction's partition helper with equal weights
int slab_sub_range(mrl_ctx *ctx, int sub, int nsub, long long *k0, long long *ksub) {
  const long long nzc = ctx->nrec[2];
  if (nsub < 1 || nsub > nzc || sub < 0 || sub >= nsub)
    return set_error(ctx, MRL_ERR_INVALID, "kz sub-block %d of %d out of range (nzc = %lld)", sub, nsub, nzc);
  long long total = nzc, remaining = nsub, b = 0;
  for (int i = 0; i <= sub; ++i) {
    long long n = total / remaining;
    // (measured: keeping the widths off powers of two -- row pitch 16*ksub bytes of the exchange layout -- changes nothing)
    if (i == nsub - 1) n = total;
    if (i == sub) {
      *k0 = b;
      *ksub = n;
    }
    b += n;
    total -= n;
    remaining -= 1;
  }
  return MRL_OK;
}

// copy a [n0][n1][n2] block (n2 contiguous) between two strided complex arrays
__global__ void __launch_bounds__(256) k_copy3(const double2 *__restrict__ src, double2 *__restrict__ dst, long long n0,
                                               long long n1, long long n2, long long ss0, long long ss1, long long ds0,
                                               long long ds1) {
  const long long total = n0 * n1 * n2;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long c = e % n2, r = e / n2, b = r % n1, a = r / n1;
    dst[a * ds0 + b * ds1 + c] = src[a * ss0 + b * ss1 + c];
  }
}

static int copy3(mrl_ctx *ctx, const double *src, double *dst, long long n0, long long n1, long long n2, long long ss0,
                 long long ss1, long long ds0, long long ds1) {
  const long long total = n0 * n1 * n2;
  if (total == 0) return MRL_OK;
  long long nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(k_copy3, dim3((unsigned)nb), dim3(256), 0, ctx->stream, reinterpret_cast<const double2 *>(src),
                     reinterpret_cast<double2 *>(dst), n0, n1, n2, ss0, ss1, ds0, ds1);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

// generic work arrays: 13, 14 = c-hat / mu-hat on the real slab [nx][nyl][nzc]; 12 = mu; 15 = dense reciprocal
// c-hat [nxl][ny][nzc]; 11 = dense reciprocal mu-hat / ubar, later the inverse x output [nx][nyl][nzc]
static int gen_work(mrl_ctx *ctx) {
  const size_t a = sizeof(cplx) * (size_t)(ctx->n[0] * ctx->nloc[1] * ctx->nrec[2]);
  const size_t b = sizeof(cplx) * (size_t)(ctx->nrec[0] * ctx->n[1] * ctx->nrec[2]);
  MRL_TRY(ensure_work(ctx, 13, a));
  MRL_TRY(ensure_work(ctx, 14, a));
  MRL_TRY(ensure_work(ctx, 15, b));
  MRL_TRY(ensure_work(ctx, 11, a > b ? a : b));
  MRL_TRY(ensure_work(ctx, 10, a > b ? a : b));
  return MRL_OK;
}

int gen_z_fwd(mrl_ctx *ctx, const ChP &cp, const double *c_in, double *d_mu, int carry) {
  const long long nx = ctx->n[0], nyl = ctx->nloc[1];
  const long long nreal = real_count_local(ctx);
  MRL_TRY(gen_work(ctx));
  double *mu = d_mu;
  if (!mu) {
    MRL_TRY(ensure_work(ctx, 12, sizeof(double) * (nreal + 2)));
    mu = ctx->d_work[12];
  }
  MRL_TRY(ch_mu_launch(ctx, cp, c_in, mu, nreal));
  ProfScope ps(ctx, "slab_z_fwd", (carry == MRL_CARRY_IN ? 1.0 : 2.0) * (8.0 * nreal + 16.0 * nx * nyl * ctx->nrec[2]));
  if (carry != MRL_CARRY_IN) MRL_TRY(pass_z_forward(ctx, c_in, ctx->d_work[13], nx, nyl, 1, 0));
  return pass_z_forward(ctx, mu, ctx->d_work[14], nx, nyl, 1, 0);
}

int gen_x_fwd(mrl_ctx *ctx, long long k0, long long ksub, double *send, int carry) {
  const long long nx = ctx->n[0], nyl = ctx->nloc[1], nzc = ctx->nrec[2];
  long long off = 0, xb = 0;
  const int f0 = carry == MRL_CARRY_IN ? 1 : 0, nf = 2 - f0;  // carry-over: mu-hat only
  for (int f = f0; f < 2; ++f) {
    double *w = ctx->d_work[13 + f] + 2 * k0;
    ProfScope ps(ctx, "slab_x_fwd", 32.0 * nx * nyl * ksub);
    MRL_TRY(pass_lines(ctx, 0, -1, w, w, nyl, ksub, nzc, 1, nyl * nzc));
  }
  ProfScope ps(ctx, "slab_pack", nf * 32.0 * nx * nyl * ksub);
  for (int p = 0; p < ctx->nranks; ++p) {
    const long long nxp = ctx->part_recip[p], chunk = nxp * nyl * ksub;
    for (int f = f0; f < 2; ++f)
      MRL_TRY(copy3(ctx, ctx->d_work[13 + f] + 2 * (xb * nyl * nzc + k0), send + 2 * (off + (f - f0) * chunk), nxp, nyl, ksub,
                    nyl * nzc, nzc, nyl * ksub, ksub));
    off += nf * chunk;
    xb += nxp;
  }
  return MRL_OK;
}

int gen_kspace(mrl_ctx *ctx, const ChP &cp, long long k0, long long ksub, const double *recv, double *send,
                      double *Nhat_new, const double *const *Nhat_old, int order, double sub_dt, double *d_cbar, int carry) {
  const long long nxl = ctx->nrec[0], ny = ctx->n[1], nzc = ctx->nrec[2];
  MRL_TRY(gen_work(ctx));
  const bool spec = carry == MRL_CARRY_IN;
  // carry-over: d_cbar holds c-hat (already in reciprocal space) on entry and receives ubar; otherwise it is the optional
  // c-hat output (MRL_CARRY_NONE) or receives ubar only (MRL_CARRY_OUT)
  double *cbar = (d_cbar && carry != MRL_CARRY_OUT) ? d_cbar : ctx->d_work[15];
  double *mubar = ctx->d_work[11], *ubar = ctx->d_work[10];
  const long long nf = spec ? 1 : 2;
  {
    ProfScope ps(ctx, "slab_unpack", nf * 32.0 * nxl * ny * ksub);
    long long off = 0, yb = 0;
    for (int p = 0; p < ctx->nranks; ++p) {
      const long long nyp = ctx->part_real[p], chunk = nxl * nyp * ksub;
      if (!spec)
        MRL_TRY(copy3(ctx, recv + 2 * off, cbar + 2 * (yb * nzc + k0), nxl, nyp, ksub, nyp * ksub, ksub, ny * nzc, nzc));
      MRL_TRY(copy3(ctx, recv + 2 * (off + (nf - 1) * chunk), mubar + 2 * (yb * nzc + k0), nxl, nyp, ksub, nyp * ksub, ksub, ny * nzc, nzc));
      off += nf * chunk;
      yb += nyp;
    }
  }
  {
    ProfScope ps(ctx, "slab_y_fwd", nf * 32.0 * nxl * ny * ksub);
    if (!spec) MRL_TRY(pass_lines(ctx, 1, -1, cbar + 2 * k0, cbar + 2 * k0, nxl, ksub, ny * nzc, 1, nzc));
    MRL_TRY(pass_lines(ctx, 1, -1, mubar + 2 * k0, mubar + 2 * k0, nxl, ksub, ny * nzc, 1, nzc));
  }
  {
    ProfScope ps(ctx, "ch_kspace", 16.0 * (double)(nxl * ny * ksub) * (4 + order));
    MRL_TRY(ch_kspace_sub_launch(ctx, cp, cbar, mubar, Nhat_new, ubar, Nhat_old, order, sub_dt, k0, ksub));
  }
  if (carry != MRL_CARRY_NONE)  // ubar of this kz sub-block = c-hat of the next substep
    MRL_TRY(copy3(ctx, ubar + 2 * k0, d_cbar + 2 * k0, nxl, ny, ksub, ny * nzc, nzc, ny * nzc, nzc));
  {
    ProfScope ps(ctx, "slab_y_inv", 32.0 * nxl * ny * ksub);
    MRL_TRY(pass_lines(ctx, 1, +1, ubar + 2 * k0, ubar + 2 * k0, nxl, ksub, ny * nzc, 1, nzc));
  }
  ProfScope ps(ctx, "slab_pack", 32.0 * nxl * ny * ksub);
  long long off = 0, yb = 0;
  for (int p = 0; p < ctx->nranks; ++p) {
    const long long nyp = ctx->part_real[p];
    MRL_TRY(copy3(ctx, ubar + 2 * (yb * nzc + k0), send + 2 * off, nxl, nyp, ksub, ny * nzc, nzc, nyp * ksub, ksub));
    off += nxl * nyp * ksub;
    yb += nyp;
  }
  return MRL_OK;
}

int gen_x_inv(mrl_ctx *ctx, long long k0, long long ksub, const double *recv) {
  const long long nx = ctx->n[0], nyl = ctx->nloc[1], nzc = ctx->nrec[2];
  MRL_TRY(gen_work(ctx));
  double *w = ctx->d_work[13];  // the forward work array of field c is free again
  {
    ProfScope ps(ctx, "slab_unpack", 32.0 * nx * nyl * ksub);
    long long off = 0, xb = 0;
    for (int p = 0; p < ctx->nranks; ++p) {
      const long long nxp = ctx->part_recip[p];
      MRL_TRY(copy3(ctx, recv + 2 * off, w + 2 * (xb * nyl * nzc + k0), nxp, nyl, ksub, nyl * ksub, ksub, nyl * nzc, nzc));
      off += nxp * nyl * ksub;
      xb += nxp;
    }
  }
  ProfScope ps(ctx, "slab_x_inv", 32.0 * nx * nyl * ksub);
  return pass_lines(ctx, 0, +1, w + 2 * k0, w + 2 * k0, nyl, ksub, nzc, 1, nyl * nzc);
}

int gen_z_inv(mrl_ctx *ctx, double *real_out) {
  const long long nx = ctx->n[0], nyl = ctx->nloc[1];
  const double scale = 1.0 / ((double)ctx->n[0] * (double)ctx->n[1] * (double)ctx->n[2]);
  ProfScope ps(ctx, "slab_z_inv", 8.0 * nx * nyl * ctx->n[2] + 16.0 * nx * nyl * ctx->nrec[2]);
  return pass_z_inverse(ctx, ctx->d_work[13], real_out, nx, nyl, 1, 0, scale);
}

}  // namespace mrl

using namespace mrl;

extern "C" {

int mrl_slab_counts(const mrl_ctx *ctx, int forward, int64_t *h_send_counts, int64_t *h_recv_counts,
                    int64_t *h_send_offsets, int64_t *h_recv_offsets) {
  if (!ctx) return MRL_ERR_INVALID;
  if (!ctx->slab) return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_counts: not a slab context");
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
  return slab_inv_finish(ctx, d_recv, d_real_out);
}

int mrl_slab_ch_counts(const mrl_ctx *ctx, int sub, int nsub, int forward, int carry, int64_t *h_send_counts,
                       int64_t *h_recv_counts) {
  if (!ctx) return MRL_ERR_INVALID;
  if (!ctx->slab) return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_ch_counts: not a slab context");
  long long k0, ksub;
  MRL_TRY(slab_sub_range(const_cast<mrl_ctx *>(ctx), sub, nsub, &k0, &ksub));
  const long long nf = (forward && carry != MRL_CARRY_IN) ? 2 : 1;  // the forward messages carry both fields (mu-hat only with the carry-over)
  const long long kp = slab_kpitch(ctx, ksub);  // planned shapes: rows of the exchange layouts are padded to 128-byte lines
  // (planned pipeline: x planes of a chunk lie slab_xplane_of(rows of the chunk) elements apart -- an odd number of 256-byte pieces)
  const bool fast = slab_fast_ok(ctx) != 0;
  for (int p = 0; p < ctx->nranks; ++p) {
    const long long x_p_y_me = fast ? ctx->part_recip[p] * slab_xplane_of(ctx, ctx->nloc[1], kp) : ctx->part_recip[p] * ctx->nloc[1] * kp;
    const long long x_me_y_p = fast ? ctx->nrec[0] * slab_xplane_of(ctx, ctx->part_real[p], kp) : ctx->nrec[0] * ctx->part_real[p] * kp;
    if (h_send_counts) h_send_counts[p] = nf * (forward ? x_p_y_me : x_me_y_p);
    if (h_recv_counts) h_recv_counts[p] = nf * (forward ? x_me_y_p : x_p_y_me);
  }
  return MRL_OK;
}

// pointer table of the staged entry points: chunk p of the caller's contiguous send buffer starts where mrl_slab_ch_counts puts it
static int staged_tab(mrl_ctx *ctx, int slot, double *d_send, int sub, int nsub, int forward, int carry, cplx *const **tab) {
  int64_t cnt[64];
  size_t off[64];
  if (ctx->nranks > 64) return set_error(ctx, MRL_ERR_UNSUPPORTED, "pointer tables hold at most 64 ranks");
  MRL_TRY(mrl_slab_ch_counts(ctx, sub, nsub, forward, carry, cnt, nullptr));
  size_t at = 0;
  for (int p = 0; p < ctx->nranks; ++p) {
    off[p] = at;
    at += sizeof(cplx) * (size_t)cnt[p];
  }
  return local_tab_offsets(ctx, slot, d_send, off, tab);
}

static int check_carry(mrl_ctx *ctx, const char *what, int carry) {
  if (carry != MRL_CARRY_NONE && carry != MRL_CARRY_OUT && carry != MRL_CARRY_IN)
    return set_error(ctx, MRL_ERR_INVALID, "%s: carry must be MRL_CARRY_NONE, _OUT or _IN", what);
  return MRL_OK;
}

int mrl_slab_ch_z_fwd(mrl_ctx *ctx, const mrl_ch_params *p, const double *d_c_in, double *d_mu, int carry) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_TRY(check_slab(ctx, "mrl_slab_ch_z_fwd"));
  ChP cp;
  MRL_TRY(ch_check_params(ctx, p, cp));
  if (!d_c_in) return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_ch_z_fwd: null buffer");
  MRL_TRY(check_carry(ctx, "mrl_slab_ch_z_fwd", carry));
  if (slab_fast_ok(ctx)) return slab_ch_z_fwd_fast(ctx, cp, d_c_in, d_mu, carry);
  return gen_z_fwd(ctx, cp, d_c_in, d_mu, carry);
}

int mrl_slab_ch_x_fwd(mrl_ctx *ctx, int sub, int nsub, double *d_send, int carry) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_TRY(check_slab(ctx, "mrl_slab_ch_x_fwd"));
  if (!d_send) return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_ch_x_fwd: null buffer");
  long long k0, ksub;
  MRL_TRY(slab_sub_range(ctx, sub, nsub, &k0, &ksub));
  MRL_TRY(check_carry(ctx, "mrl_slab_ch_x_fwd", carry));
  if (slab_fast_ok(ctx)) {
    // the pass scatters through a destination table: here every chunk goes to the caller's contiguous send buffer
    cplx *const *tab;
    MRL_TRY(staged_tab(ctx, 0, d_send, sub, nsub, 1, carry, &tab));
    return slab_ch_x_fwd_fast(ctx, (int)k0, (int)ksub, tab, SignalArgs{}, carry);
  }
  return gen_x_fwd(ctx, k0, ksub, d_send, carry);
}

int64_t mrl_slab_ch_spec_pitch(const mrl_ctx *ctx) {
  if (!ctx) return 0;
  const long long nzc = ctx->nrec[ctx->dim - 1];
  return (ctx->slab && slab_fast_ok(ctx)) ? ((nzc + 7) & ~7LL) : nzc;
}

int64_t mrl_slab_ch_k_pitch(const mrl_ctx *ctx, int sub, int nsub) {
  if (!ctx || !ctx->slab) return 0;
  long long k0, ksub;
  if (slab_sub_range(const_cast<mrl_ctx *>(ctx), sub, nsub, &k0, &ksub) != MRL_OK) return 0;
  return slab_kpitch(ctx, ksub);
}

int mrl_slab_ch_kspace(mrl_ctx *ctx, const mrl_ch_params *p, int sub, int nsub, const double *d_recv, double *d_send,
                       double *d_Nhat_new, const double *const *d_Nhat_old, int order, double sub_dt, double *d_cbar, int carry) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_TRY(check_slab(ctx, "mrl_slab_ch_kspace"));
  MRL_TRY(check_carry(ctx, "mrl_slab_ch_kspace", carry));
  if (carry != MRL_CARRY_NONE && !d_cbar)
    return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_ch_kspace: the carry-over needs the d_cbar array");
  ChP cp;
  MRL_TRY(ch_check_params(ctx, p, cp));
  if (!d_recv || !d_send || !d_Nhat_new) return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_ch_kspace: null buffer");
  if (order < 0 || order > 4) return set_error(ctx, MRL_ERR_INVALID, "predictor order %d out of range", order + 1);
  for (int i = 0; i < order; ++i)
    if (!d_Nhat_old || !d_Nhat_old[i]) return set_error(ctx, MRL_ERR_INVALID, "history entry %d missing", i);
  long long k0, ksub;
  MRL_TRY(slab_sub_range(ctx, sub, nsub, &k0, &ksub));
  if (slab_fast_ok(ctx)) {
    cplx *const *tab;
    MRL_TRY(staged_tab(ctx, 1, d_send, sub, nsub, 0, MRL_CARRY_NONE, &tab));
    return slab_ch_kspace_fast(ctx, cp, (int)k0, (int)ksub, d_recv, tab, SignalArgs{}, d_Nhat_new, d_Nhat_old, order, sub_dt, d_cbar, carry);
  }
  return gen_kspace(ctx, cp, k0, ksub, d_recv, d_send, d_Nhat_new, d_Nhat_old, order, sub_dt, d_cbar, carry);
}

int mrl_slab_ch_x_inv(mrl_ctx *ctx, int sub, int nsub, const double *d_recv) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_TRY(check_slab(ctx, "mrl_slab_ch_x_inv"));
  if (!d_recv) return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_ch_x_inv: null buffer");
  long long k0, ksub;
  MRL_TRY(slab_sub_range(ctx, sub, nsub, &k0, &ksub));
  if (slab_fast_ok(ctx)) return slab_ch_x_inv_fast(ctx, (int)k0, (int)ksub, d_recv);
  return gen_x_inv(ctx, k0, ksub, d_recv);
}

int mrl_slab_ch_z_inv(mrl_ctx *ctx, double *d_c_out) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_TRY(check_slab(ctx, "mrl_slab_ch_z_inv"));
  if (!d_c_out) return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_ch_z_inv: null buffer");
  if (slab_fast_ok(ctx)) return slab_ch_z_inv_fast(ctx, d_c_out);
  return gen_z_inv(ctx, d_c_out);
}

int mrl_slab_ch_z_inv_fwd(mrl_ctx *ctx, const mrl_ch_params *p, double *d_mu, int carry) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_TRY(check_slab(ctx, "mrl_slab_ch_z_inv_fwd"));
  MRL_TRY(check_carry(ctx, "mrl_slab_ch_z_inv_fwd", carry));
  ChP cp;
  MRL_TRY(ch_check_params(ctx, p, cp));
  if (slab_fast_ok(ctx) && (ctx->n[0] * ctx->nloc[1]) % 2 == 0) return slab_ch_z_inv_fwd_fast(ctx, cp, d_mu, carry);
  // generic shapes: the two passes one after the other through a scratch real field
  const long long nreal = real_count_local(ctx);
  MRL_TRY(ensure_work(ctx, 9, sizeof(double) * (size_t)(nreal + 2)));
  MRL_TRY(gen_z_inv(ctx, ctx->d_work[9]));
  return gen_z_fwd(ctx, cp, ctx->d_work[9], d_mu, carry);
}

}  // extern "C"
