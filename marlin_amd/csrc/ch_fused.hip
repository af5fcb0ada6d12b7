// Fast path: 3-D power-of-two grids on one GPU.
//   forward r2c / inverse c2r built from the pow2 kernels, and the fused Cahn-Hilliard substep:
//     A  k_z_fwd<CH>   c -> (c-hat_z, mu-hat_z)          mu = f'(c) evaluated in the loader
//     B  k_pass<y>     both fields, forward along y
//     C  k_ch_xfused   forward x on both fields, Nhat = Mbar*mu-hat (stored: history), ABM predictor,
//                      1/(1 - dt*Lbar), inverse x                      (AdamsBashforthMoulton.C:94-101)
//     D  k_pass<y>     inverse along y
//     E  k_z_inv       c2r along z, 1/N
//   Forward order z,y,x and inverse order x,y,z: the fused pass runs along x because its tiles are
//   contiguous in (y,kz) jointly, so the user-visible Nhat arrays are accessed in aligned 256-B pieces.
#include "ch_xfused.h"
#include "fft_pow2_wide.h"
#include "fft_two.h"
#include "fft_two_z.h"

namespace mrl {


// 3-D grids [nx][ny][nz], and 2-D grids [nx][nz'] run as [nx][1][nz'] (serial contexts hold them as internal axes
// (1, nx, ny_user): the user's y is the contiguous r2c axis; the absent middle axis contributes k = 0 exactly)
bool fast_path_ok(const mrl_ctx *ctx) {
  if (ctx->slab || ctx->pencil || ctx->spectrum != MRL_SPECTRUM_HALF) return false;
  if (ctx->exp & 2048) return false;  // experiment: planned shapes through the any-length path (A/B and parity at sizes the oracle cannot reach)
  // the fused kernels address one spectral array with 32-bit byte offsets from its base (ch_fused_body.h); arrays of 4 GiB and
  // more need the 64-bit variant (x lengths 512 / 768 / 1000 / 1024) and a per-lane part (one line stride) below 4 GiB
  if (16.0 * (double)ctx->nrec[0] * (double)ctx->nrec[1] * (double)ctx->nrec[2] >= 4294967296.0) {
    const long long nx = ctx->n[ctx->dim == 3 ? 0 : 1];
    if (!(nx == 512 || nx == 768 || nx == 1000 || nx == 1024)) return false;
    const double P = nx == 1000 ? 10.0 : (nx == 768 ? 12.0 : 16.0);
    if (16.0 * ((double)nx / P) * (double)(ctx->dim == 3 ? ctx->n[1] : 1) * (double)ctx->nrec[2] >= 4294967296.0) return false;
  }
  if (ctx->dim == 3) return pow2_ok(ctx->n[0]) && pow2_ok(ctx->n[1]) && pow2_ok(ctx->n[2]);
  if (ctx->dim == 2) return pow2_ok(ctx->n[1]) && pow2_ok(ctx->n[2]);
  return false;
}

struct Geo {
  long long nx, ny, nz, nzc;
  long long plane;   // elements between two x planes of the solver-private spectral arrays (= ny * nzc when dense)
  p2::ZLay zl;       // the same for the z kernels (rows per plane, extra elements per plane)
  const cplx *tw_x, *tw_y;
  const double *kx, *ky, *kz;
};
static Geo geo_of(const mrl_ctx *ctx) {
  Geo g;
  const int ax = ctx->dim == 3 ? 0 : 1;  // internal axis that carries x
  g.nx = ctx->n[ax];
  g.ny = ctx->dim == 3 ? ctx->n[1] : 1;
  g.nz = ctx->n[2];
  g.nzc = ctx->nrec[2];
  g.plane = ctx->spec_plane ? ctx->spec_plane : g.ny * g.nzc;
  g.zl = p2::ZLay{(unsigned)g.ny, (unsigned)(g.plane - g.ny * g.nzc)};
  g.tw_x = ctx->ax[ax].d_tw;
  g.tw_y = ctx->ax[1].d_tw;
  g.kx = ctx->d_k[ax];
  g.ky = ctx->dim == 3 ? ctx->d_k[1] : ctx->d_k[0];  // 2-D: the unused axis {0}
  g.kz = ctx->d_k[2];
  return g;
}

// 400-point x axes: the fused x pass of the two-stage plan 20 x 20 (fft_two.h: k_ch_xfused2 with the re-read Nhat) instead of
// k_ch_xfused<400> (10 x 10 x 2 x 2, 40 threads per line, tiles of 6 lines = 96-byte pieces: 3.5 TB/s; 735 -> 547 us at 400^3);
// the reference's data flow only (no spectral carry-over); experiment bit 1 << 29 keeps the uniform kernel (A/B).  192-point x axes
// likewise (12 x 16 instead of 12 x 4 x 4: 52.5 -> 50.0 us at 192^3)
static bool x400_two_stage(const mrl_ctx *ctx, long long nx, int carry, double array_bytes) {
  return (nx == 400 || nx == 192) && carry == MRL_CARRY_NONE && !(ctx->exp & (1 << 29)) && array_bytes < 4294967296.0;
}
static int launch_x400(mrl_ctx *ctx, const p2::FusedArgs &f, int order, const cplx *tw, bool nx_is_192) {
  p2::X2Args a{};
  a.chat = f.c.chat;
  a.muhat = f.c.muhat;
  a.ubar = f.c.ubar;
  a.Nnew = f.c.Nnew;
  a.cbar = f.c.cbar;
  for (int i = 0; i < 4; ++i) a.Nold[i] = f.c.Nold[i];
  for (int i = 0; i < 5; ++i) a.coef[i] = f.c.coef[i];
  a.M = f.c.M;
  a.kappa = f.c.kappa;
  a.dt = f.c.dt;
  a.inner = f.inner;
  a.plane = f.plane;
  a.nzc = f.nzc;
  a.kx = f.kx;
  a.ky = f.ky;
  a.kz = f.kz;
  const bool n192 = nx_is_192;
  switch (order) {
    case 0: return n192 ? p2::launch_xfused2<192, 0>(ctx, a, tw) : p2::launch_xfused2<400, 0>(ctx, a, tw);
    case 1: return n192 ? p2::launch_xfused2<192, 1>(ctx, a, tw) : p2::launch_xfused2<400, 1>(ctx, a, tw);
    case 2: return n192 ? p2::launch_xfused2<192, 2>(ctx, a, tw) : p2::launch_xfused2<400, 2>(ctx, a, tw);
    case 3: return n192 ? p2::launch_xfused2<192, 3>(ctx, a, tw) : p2::launch_xfused2<400, 3>(ctx, a, tw);
    default: return n192 ? p2::launch_xfused2<192, 4>(ctx, a, tw) : p2::launch_xfused2<400, 4>(ctx, a, tw);
  }
}

// strided pass along internal axis `a` (0 = x, 1 = y) of NF complex [nx][ny][nzc] arrays
// plane: elements between two x planes of the arrays (0 = dense: the plain transforms on caller arrays)
static int pass_axis(mrl_ctx *ctx, int axis, bool inv, int nf, const cplx *in0, const cplx *in1, cplx *out0,
                     cplx *out1, bool reverse = false, long long x0 = 0, long long x1 = -1, long long plane = 0) {
  const Geo g = geo_of(ctx);
  if (axis == 1 && g.ny == 1) return MRL_OK;  // 2-D: there is no middle axis
  const long long ny = g.ny, nzc = g.nzc;
  if (plane == 0) plane = ny * nzc;
  const long long nx = (x1 < 0 ? g.nx : x1) - x0;  // y pass only: restrict to the x planes [x0, x1)
  if (x0 > 0) {
    in0 += x0 * plane;
    out0 += x0 * plane;
    if (in1) in1 += x0 * plane;
    if (out1) out1 += x0 * plane;
  }
  p2::PassArgs a{};
  a.in[0] = in0;
  a.in[1] = in1;
  a.out[0] = out0;
  a.out[1] = out1;
  a.scale = 1.0;
  a.reverse = reverse ? 1 : 0;
  if (axis == 1) {
    a.inner = nzc;
    a.outer = nx;
    a.so_in = a.so_out = plane;
    a.sn_in = a.sn_out = nzc;
  } else {
    a.inner = ny * nzc;
    a.outer = 1;
    a.so_in = a.so_out = 0;
    a.sn_in = a.sn_out = plane;
  }
  const cplx *tw = axis == 1 ? g.tw_y : g.tw_x;
  const long long n = axis == 1 ? g.ny : g.nx;
  // experiment bit 1024: the wide plan (fft_pow2_wide.h) for 512-point lines.  Measured on plain transforms: +18 % on a 135 MB
  // array (Infinity-Cache resident), -3 ... -4 % on 1 GB arrays -- not the default on one GPU; the slab x passes use it (-12 %)
  if (n == 512 && (ctx->exp & 1024)) {
    for (int f = 0; f < nf; ++f) {
      p2::PassArgs b = a;
      b.in[0] = a.in[f];
      b.out[0] = a.out[f];
      if (inv) {
        MRL_TRY((p2::launch_pass_w<p2::Wide512, true>(ctx, b, tw)));
      } else {
        MRL_TRY((p2::launch_pass_w<p2::Wide512, false>(ctx, b, tw)));
      }
    }
    return MRL_OK;
  }
  if (nf == 2) {
    if (inv) {
      MRL_SWITCH_N(n, MRL_TRY((p2::launch_pass_t<NN, true, 2>(ctx, a, tw))));
    } else {
      MRL_SWITCH_N(n, MRL_TRY((p2::launch_pass_t<NN, false, 2>(ctx, a, tw))));
    }
  } else {
    if (inv) {
      MRL_SWITCH_N(n, MRL_TRY((p2::launch_pass_t<NN, true, 1>(ctx, a, tw))));
    } else {
      MRL_SWITCH_N(n, MRL_TRY((p2::launch_pass_t<NN, false, 1>(ctx, a, tw))));
    }
  }
  return MRL_OK;
}

// launchers by run-time length for the planned-unfused path (ch_planned.hip), which mixes these lengths with the radix-30 ones
int pass_launch_std(mrl_ctx *ctx, long long n, bool inv, int nf, const p2::PassArgs &a, const cplx *tw) {
  if (nf == 2) {
    if (inv) {
      MRL_SWITCH_N(n, MRL_TRY((p2::launch_pass_t<NN, true, 2>(ctx, a, tw))));
    } else {
      MRL_SWITCH_N(n, MRL_TRY((p2::launch_pass_t<NN, false, 2>(ctx, a, tw))));
    }
  } else {
    if (inv) {
      MRL_SWITCH_N(n, MRL_TRY((p2::launch_pass_t<NN, true, 1>(ctx, a, tw))));
    } else {
      MRL_SWITCH_N(n, MRL_TRY((p2::launch_pass_t<NN, false, 1>(ctx, a, tw))));
    }
  }
  return MRL_OK;
}
int z_fwd_launch_std(mrl_ctx *ctx, long long n, int mode, int fam, const double *in, cplx *o0, cplx *o1, double *mu, const p2::ChDev &chp,
                     long long nlines) {
  if (mode == 0) {
    MRL_SWITCH_N(n, MRL_TRY((p2::launch_z_fwd<NN, 0, 0>(ctx, in, o0, o1, mu, chp, nlines))));
  } else if (fam == MRL_FE_DOUBLE_WELL) {
    MRL_SWITCH_N(n, MRL_TRY((p2::launch_z_fwd<NN, 1, MRL_FE_DOUBLE_WELL>(ctx, in, o0, o1, mu, chp, nlines))));
  } else {
    MRL_SWITCH_N(n, MRL_TRY((p2::launch_z_fwd<NN, 1, MRL_FE_PFHUB>(ctx, in, o0, o1, mu, chp, nlines))));
  }
  return MRL_OK;
}
// fused inverse + forward z pass (built-in families) on dense rows; nlines = line pairs
int z_inv_fwd_launch_std(mrl_ctx *ctx, long long n, int fam, const cplx *in, cplx *o0, cplx *o1, double *mu, const p2::ChDev &chp, double scale,
                         long long nlines) {
  if (fam == MRL_FE_DOUBLE_WELL) {
    MRL_SWITCH_N(n, MRL_TRY((p2::launch_z_inv_fwd<NN, MRL_FE_DOUBLE_WELL>(ctx, in, o0, o1, mu, chp, scale, nlines))));
  } else {
    MRL_SWITCH_N(n, MRL_TRY((p2::launch_z_inv_fwd<NN, MRL_FE_PFHUB>(ctx, in, o0, o1, mu, chp, scale, nlines))));
  }
  return MRL_OK;
}
int z_inv_launch_std(mrl_ctx *ctx, long long n, const cplx *in, double *out, double scale, long long nlines) {
  MRL_SWITCH_N(n, MRL_TRY((p2::launch_z_inv<NN>(ctx, in, out, scale, nlines))));
  return MRL_OK;
}

// plain transforms (field-major batch), used by mrl_fft_r2c / mrl_fft_c2r on fast-path shapes
int fft_forward_fast(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch) {
  const Geo g = geo_of(ctx);
  const long long nx = g.nx, ny = g.ny, nz = g.nz, nzc = g.nzc;
  const long long nreal = nx * ny * nz, nspec = nx * ny * nzc;
  p2::ChDev none{};
  for (long long b = 0; b < batch; ++b) {
    const double *in = d_in + b * nreal;
    cplx *out = reinterpret_cast<cplx *>(d_out) + b * nspec;
    {
      ProfScope ps(ctx, "z_fwd_pair", 8.0 * nreal + 16.0 * nspec);
      MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_fwd<NN, 0, 0>(ctx, in, out, nullptr, nullptr, none, nx * ny / 2))));
    }
    {
      ProfScope ps(ctx, "pass_y", 32.0 * nspec);
      MRL_TRY(pass_axis(ctx, 1, false, 1, out, nullptr, out, nullptr));
    }
    {
      ProfScope ps(ctx, "pass_x", 32.0 * nspec);
      MRL_TRY(pass_axis(ctx, 0, false, 1, out, nullptr, out, nullptr));
    }
  }
  return MRL_OK;
}

int fft_inverse_fast(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch) {
  const Geo g = geo_of(ctx);
  const long long nx = g.nx, ny = g.ny, nz = g.nz, nzc = g.nzc;
  const long long nreal = nx * ny * nz, nspec = nx * ny * nzc;
  const double scale = 1.0 / ((double)nx * (double)ny * (double)nz);
  MRL_TRY(ensure_work(ctx, 0, sizeof(cplx) * nspec));
  cplx *w = reinterpret_cast<cplx *>(ctx->d_work[0]);
  for (long long b = 0; b < batch; ++b) {
    const cplx *in = reinterpret_cast<const cplx *>(d_in) + b * nspec;
    double *out = d_out + b * nreal;
    {
      ProfScope ps(ctx, "pass_x", 32.0 * nspec);
      MRL_TRY(pass_axis(ctx, 0, true, 1, in, nullptr, w, nullptr));
    }
    {
      ProfScope ps(ctx, "pass_y", 32.0 * nspec);
      MRL_TRY(pass_axis(ctx, 1, true, 1, w, nullptr, w, nullptr));
    }
    {
      ProfScope ps(ctx, "z_inv_pair", 8.0 * nreal + 16.0 * nspec);
      MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_inv<NN>(ctx, w, out, scale, nx * ny / 2))));
    }
  }
  return MRL_OK;
}

int ch_substep_fused(mrl_ctx *ctx, const ChP &cp, const double *c_in, double *c_out, double *Nhat_new,
                     const double *const *Nhat_old, int order, double sub_dt, double *cbar, double *mu, int carry) {
  if (!fast_path_ok(ctx)) return MRL_ERR_UNSUPPORTED;
  const bool spec = carry == MRL_CARRY_IN;  // spectral carry-over: c-hat is `cbar` (= ubar of the previous substep)
  if (spec && 16.0 * (double)ctx->nrec[0] * (double)ctx->nrec[1] * (double)ctx->nrec[2] >= 4294967296.0)
    return MRL_ERR_UNSUPPORTED;  // (the 64-bit-offset variant of the fused pass exists for the reference's data flow only)
  const Geo g = geo_of(ctx);
  const long long nx = g.nx, ny = g.ny, nz = g.nz, nzc = g.nzc, plane = g.plane;
  const long long nreal = nx * ny * nz, nspec = nx * ny * nzc;
  MRL_TRY(ensure_work(ctx, 1, sizeof(cplx) * nx * plane));
  MRL_TRY(ensure_work(ctx, 2, sizeof(cplx) * nx * plane));
  cplx *w_c = reinterpret_cast<cplx *>(ctx->d_work[1]);
  cplx *w_mu = reinterpret_cast<cplx *>(ctx->d_work[2]);
  p2::ChDev chp{cp.family, cp.c0, cp.c1, cp.c2, {}};
  const unsigned lpp = g.zl.lpp, lpad = g.zl.pad;
  const double h = 16.0 * nspec;  // bytes of one complex half-spectrum array
  // A (z pass) and B (y pass) are both local to an x plane and can run chunk by chunk over x, so that B reads what A
  // has just written while it is still in the 256 MB Infinity Cache (B walks its tiles in reverse, starting where A ended)
  // (measured: 2 and 4 chunks are slower than one launch each -- the smaller grids cost more than the cache hits gain)
  const int nchunk = (ctx->exp & 16) ? 4 : ((ctx->exp & 8) ? 2 : 1);
  for (int ch = 0; ch < nchunk; ++ch) {
    const long long x0 = nx * ch / nchunk, x1 = nx * (ch + 1) / nchunk;
    const long long l0 = x0 * ny, nl = (x1 - x0) * ny;
    const double *cin = c_in + l0 * nz;
    cplx *wc = w_c + x0 * plane, *wm = w_mu + x0 * plane;
    double *muc = mu ? mu + l0 * nz : nullptr;
    if (spec) {  // mu only: two lines per complex transform, one field through the y pass
      {
        ProfScope ps(ctx, "ch_A_z_fwd", (8.0 * nreal + h + (mu ? 8.0 * nreal : 0.0)) / nchunk);
        if (cp.family == MRL_FE_PARSED) {
          MRL_TRY(parsed_z_fwd_launch(ctx, cp.parsed, (int)nz, 2, cin, wm, nullptr, muc, nl / 2, lpp, lpad));
        } else if (cp.family == MRL_FE_DOUBLE_WELL) {
          MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_fwd<NN, 2, MRL_FE_DOUBLE_WELL>(ctx, cin, wm, nullptr, muc, chp, nl / 2, g.zl))));
        } else {
          MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_fwd<NN, 2, MRL_FE_PFHUB>(ctx, cin, wm, nullptr, muc, chp, nl / 2, g.zl))));
        }
      }
      ProfScope ps(ctx, "ch_B_y_fwd", 2.0 * h / nchunk);
      MRL_TRY(pass_axis(ctx, 1, false, 1, w_mu, nullptr, w_mu, nullptr, true, x0, x1, plane));
      continue;
    }
    {
      ProfScope ps(ctx, "ch_A_z_fwd", (8.0 * nreal + 2.0 * h + (mu ? 8.0 * nreal : 0.0)) / nchunk);
      if (cp.family == MRL_FE_PARSED) {
        MRL_TRY(parsed_z_fwd_launch(ctx, cp.parsed, (int)nz, 1, cin, wc, wm, muc, nl, lpp, lpad));
      } else if (cp.family == MRL_FE_DOUBLE_WELL) {
        MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_fwd<NN, 1, MRL_FE_DOUBLE_WELL>(ctx, cin, wc, wm, muc, chp, nl, g.zl))));
      } else {
        MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_fwd<NN, 1, MRL_FE_PFHUB>(ctx, cin, wc, wm, muc, chp, nl, g.zl))));
      }
    }
    {
      ProfScope ps(ctx, "ch_B_y_fwd", 4.0 * h / nchunk);
      MRL_TRY(pass_axis(ctx, 1, false, 2, w_c, w_mu, w_c, w_mu, true, x0, x1, plane));
    }
  }
  {
    ProfScope ps(ctx, "ch_C_x_fused", ((spec ? 5.0 : 4.0) + order + (cbar && !spec ? 1.0 : 0.0)) * h);
    p2::FusedArgs a{};
    a.c.chat = w_c;
    a.c.muhat = w_mu;
    a.c.ubar = w_c;
    a.c.Nnew = reinterpret_cast<cplx *>(Nhat_new);
    a.c.cbar = carry == MRL_CARRY_NONE ? reinterpret_cast<cplx *>(cbar) : nullptr;
    a.c.carry = carry == MRL_CARRY_NONE ? nullptr : reinterpret_cast<cplx *>(cbar);
    for (int i = 0; i < order; ++i) a.c.Nold[i] = reinterpret_cast<const cplx *>(Nhat_old[i]);
    for (int i = 0; i <= order; ++i) a.c.coef[i] = sub_dt * kBetaAB[order][i];
    a.inner = ny * nzc;
    a.plane = plane;
    a.nzc = (int)nzc;
    a.kx = g.kx;
    a.ky = g.ky;
    a.kz = g.kz;
    a.c.M = cp.M;
    a.c.kappa = cp.kappa;
    a.c.dt = sub_dt;
    if (spec) {
      switch (order) {
        case 0: MRL_SWITCH_N(nx, MRL_TRY((p2::launch_xfused<NN, 0, true>(ctx, a, g.tw_x)))); break;
        case 1: MRL_SWITCH_N(nx, MRL_TRY((p2::launch_xfused<NN, 1, true>(ctx, a, g.tw_x)))); break;
        case 2: MRL_SWITCH_N(nx, MRL_TRY((p2::launch_xfused<NN, 2, true>(ctx, a, g.tw_x)))); break;
        case 3: MRL_SWITCH_N(nx, MRL_TRY((p2::launch_xfused<NN, 3, true>(ctx, a, g.tw_x)))); break;
        default: MRL_SWITCH_N(nx, MRL_TRY((p2::launch_xfused<NN, 4, true>(ctx, a, g.tw_x)))); break;
      }
    } else if (x400_two_stage(ctx, nx, carry, 16.0 * (double)nx * (double)plane)) {
      MRL_TRY(launch_x400(ctx, a, order, g.tw_x, nx == 192));
    } else {
      switch (order) {
        case 0: MRL_SWITCH_N(nx, MRL_TRY((p2::launch_xfused<NN, 0, false>(ctx, a, g.tw_x)))); break;
        case 1: MRL_SWITCH_N(nx, MRL_TRY((p2::launch_xfused<NN, 1, false>(ctx, a, g.tw_x)))); break;  // PRE = 8 (4, 12: same; 16 spills: -12 %)
        case 2: MRL_SWITCH_N(nx, MRL_TRY((p2::launch_xfused<NN, 2, false>(ctx, a, g.tw_x)))); break;
        case 3: MRL_SWITCH_N(nx, MRL_TRY((p2::launch_xfused<NN, 3, false>(ctx, a, g.tw_x)))); break;
        default: MRL_SWITCH_N(nx, MRL_TRY((p2::launch_xfused<NN, 4, false>(ctx, a, g.tw_x)))); break;
      }
    }
  }
  {
    ProfScope ps(ctx, "ch_D_y_inv", 2.0 * h);
    MRL_TRY(pass_axis(ctx, 1, true, 1, w_c, nullptr, w_c, nullptr, false, 0, -1, plane));
  }
  {
    ProfScope ps(ctx, "ch_E_z_inv", h + 8.0 * nreal);
    const double scale = 1.0 / ((double)nx * (double)ny * (double)nz);
    MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_inv<NN>(ctx, w_c, c_out, scale, nx * ny / 2, g.zl))));
  }
  return MRL_OK;
}

// `count` consecutive substeps of one TensorSolver::computeBuffer call (TensorSolver.C:93-109): between two substeps the z inverse
// of substep k and the z forward of substep k + 1 are ONE kernel (k_z_inv_fwd) and the intermediate real field never touches HBM.
// ring: `ring_size` = pred + 1 Nhat arrays; *head = slot of the newest history entry, *n_old = valid history entries;
// the substep writes its Nhat into slot (*head + 1) % ring_size and, if `advance`, that slot becomes the head before the next
// substep (TensorBuffer<T>::advanceState between substeps).  After the call the newest Nhat is in slot (*head + 1) % ring_size.
int ch_substeps_fused(mrl_ctx *ctx, const ChP &cp, const double *c_in, double *c_out, double *const *ring, int ring_size,
                      int *head, int *n_old, int pred, int count, int advance, double sub_dt, double *mu, bool dt_changed) {
  if (!fast_path_ok(ctx)) return MRL_ERR_UNSUPPORTED;
  const Geo g = geo_of(ctx);
  const long long nx = g.nx, ny = g.ny, nz = g.nz, nzc = g.nzc, plane = g.plane;
  if ((nx * ny) % 2) return MRL_ERR_UNSUPPORTED;
  const long long nreal = nx * ny * nz, nspec = nx * ny * nzc;
  const unsigned lpp = g.zl.lpp, lpad = g.zl.pad;
  MRL_TRY(ensure_work(ctx, 1, sizeof(cplx) * nx * plane));
  MRL_TRY(ensure_work(ctx, 2, sizeof(cplx) * nx * plane));
  cplx *w_c = reinterpret_cast<cplx *>(ctx->d_work[1]);
  cplx *w_mu = reinterpret_cast<cplx *>(ctx->d_work[2]);
  p2::ChDev chp{cp.family, cp.c0, cp.c1, cp.c2, {}};
  const double h = 16.0 * nspec;
  const double scale = 1.0 / ((double)nx * (double)ny * (double)nz);
  // Infinity-Cache-sized chunks of x planes (round 4, MRL_OPT_CACHE_CHUNK_MB; OFF by default).  Between the x-fused pass of one substep
  // and that of the next every pass works plane by plane -- inverse y, fused inverse + forward z, forward y -- so they can run chunk
  // after chunk, a chunk's c-hat and mu-hat planes being written by one pass and read by the next while they are still in the 256 MiB
  // cache (DRAM traffic per substep 14 h -> 8 h for arrays beyond the cache).  Same kernels on the same data in another order:
  // bit-identical fields.  Measured (tools/chunk_sweep.py, 512^3, one box): unchunked 2.885 ms, chunks of 32 / 64 / 96 / 128 / 160 /
  // 192 MB: 4.14 / 3.10 / 3.30 / 2.98 / 3.02 / 2.86 ms -- an in-cache stream is only ~25 % faster than a DRAM one (6.8 against 5.3-5.5
  // TB/s, tools/mall_probe.hip) and launches of 700-1400 workgroups on 512-1024 slots lose more than that in their last partial wave;
  // 256^3: 0.305 -> 0.34-0.43 ms.  Kept as an option for A/B runs on other parts, not used by default.
  long long xchunk = 0;
  if (ctx->opt_chunk_mb > 0 && ny > 1) {
    xchunk = (long long)((double)ctx->opt_chunk_mb * 1048576.0 / (2.0 * 16.0 * (double)plane));
    if (xchunk < 1) xchunk = 1;
    if ((xchunk * ny) % 2) xchunk += 1;   // the z kernels take line PAIRS
    if (xchunk >= nx) xchunk = 0;
  }
  auto launch_ea = [&](long long x0, long long x1, double *mu_k) -> int {   // fused inverse + forward z pass of the x planes [x0, x1)
    cplx *c = w_c + x0 * plane, *m = w_mu + x0 * plane;
    double *mk = mu_k ? mu_k + x0 * ny * nz : nullptr;
    const long long pairs = (x1 - x0) * ny / 2;
    ProfScope ps(ctx, "ch_EA_z_inv_fwd", (3.0 * h + (mu_k ? 8.0 * nreal : 0.0)) * (double)(x1 - x0) / (double)nx);
    if (cp.family == MRL_FE_PARSED) {
      MRL_TRY(parsed_z_inv_fwd_launch(ctx, cp.parsed, (int)nz, c, c, m, mk, scale, pairs, false, lpp, lpad));
    } else if (nz == 192 && !(ctx->exp & (1 << 29))) {
      // 192-point lines: the two-stage kernel 16 x 12 (fft_two_z.h) instead of k_z_inv_fwd<192> (12 x 4 x 4)
      if (cp.family == MRL_FE_DOUBLE_WELL) {
        MRL_TRY((p2::launch_z_inv_fwd2<192, MRL_FE_DOUBLE_WELL>(ctx, c, c, m, mk, chp, scale, pairs, g.zl)));
      } else {
        MRL_TRY((p2::launch_z_inv_fwd2<192, MRL_FE_PFHUB>(ctx, c, c, m, mk, chp, scale, pairs, g.zl)));
      }
    } else if (cp.family == MRL_FE_DOUBLE_WELL) {
      MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_inv_fwd<NN, MRL_FE_DOUBLE_WELL>(ctx, c, c, m, mk, chp, scale, pairs, g.zl))));
    } else {
      MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_inv_fwd<NN, MRL_FE_PFHUB>(ctx, c, c, m, mk, chp, scale, pairs, g.zl))));
    }
    return MRL_OK;
  };
  auto launch_b = [&](long long x0, long long x1) -> int {
    ProfScope ps(ctx, "ch_B_y_fwd", 4.0 * h * (double)(x1 - x0) / (double)nx);
    return pass_axis(ctx, 1, false, 2, w_c, w_mu, w_c, w_mu, true, x0, x1, plane);
  };
  auto launch_d = [&](long long x0, long long x1) -> int {
    ProfScope ps(ctx, "ch_D_y_inv", 2.0 * h * (double)(x1 - x0) / (double)nx);
    return pass_axis(ctx, 1, true, 1, w_c, nullptr, w_c, nullptr, false, x0, x1, plane);
  };
  for (int k = 0; k < count; ++k) {
    double *mu_k = (k == count - 1) ? mu : nullptr;   // the buffer `mu` holds f'(c) of the last substep's input field
    if (k == 0) {
      {
        ProfScope ps(ctx, "ch_A_z_fwd", 8.0 * nreal + 2.0 * h + (mu_k ? 8.0 * nreal : 0.0));
        if (cp.family == MRL_FE_PARSED) {
          MRL_TRY(parsed_z_fwd_launch(ctx, cp.parsed, (int)nz, 1, c_in, w_c, w_mu, mu_k, nx * ny, lpp, lpad));
        } else if (cp.family == MRL_FE_DOUBLE_WELL) {
          MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_fwd<NN, 1, MRL_FE_DOUBLE_WELL>(ctx, c_in, w_c, w_mu, mu_k, chp, nx * ny, g.zl))));
        } else {
          MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_fwd<NN, 1, MRL_FE_PFHUB>(ctx, c_in, w_c, w_mu, mu_k, chp, nx * ny, g.zl))));
        }
      }
      MRL_TRY(launch_b(0, nx));
    } else if (xchunk) {
      for (long long x0 = 0; x0 < nx; x0 += xchunk) {   // D of substep k - 1, then EA and B of substep k, chunk after chunk
        const long long x1 = x0 + xchunk < nx ? x0 + xchunk : nx;
        MRL_TRY(launch_d(x0, x1));
        MRL_TRY(launch_ea(x0, x1, mu_k));
        MRL_TRY(launch_b(x0, x1));
      }
    } else {
      MRL_TRY(launch_ea(0, nx, mu_k));
      MRL_TRY(launch_b(0, nx));
    }
    const int order = (dt_changed && k < pred) ? 0 : (*n_old < pred ? *n_old : pred);   // AdamsBashforthMoulton.C:90-91
    const int slot_new = (*head + 1) % ring_size;
    {
      ProfScope ps(ctx, "ch_C_x_fused", (4.0 + order) * h);
      p2::FusedArgs a{};
      a.c.chat = w_c;
      a.c.muhat = w_mu;
      a.c.ubar = w_c;
      a.c.Nnew = reinterpret_cast<cplx *>(ring[slot_new]);
      for (int i = 0; i < order; ++i) a.c.Nold[i] = reinterpret_cast<const cplx *>(ring[((*head - i) % ring_size + ring_size) % ring_size]);
      for (int i = 0; i <= order; ++i) a.c.coef[i] = sub_dt * kBetaAB[order][i];
      a.inner = ny * nzc;
      a.plane = plane;
      a.nzc = (int)nzc;
      a.kx = g.kx;
      a.ky = g.ky;
      a.kz = g.kz;
      a.c.M = cp.M;
      a.c.kappa = cp.kappa;
      a.c.dt = sub_dt;
      if (x400_two_stage(ctx, nx, MRL_CARRY_NONE, 16.0 * (double)nx * (double)plane)) {
        MRL_TRY(launch_x400(ctx, a, order, g.tw_x, nx == 192));
      } else {
        switch (order) {
          case 0: MRL_SWITCH_N(nx, MRL_TRY((p2::launch_xfused<NN, 0, false>(ctx, a, g.tw_x)))); break;
          case 1: MRL_SWITCH_N(nx, MRL_TRY((p2::launch_xfused<NN, 1, false>(ctx, a, g.tw_x)))); break;
          case 2: MRL_SWITCH_N(nx, MRL_TRY((p2::launch_xfused<NN, 2, false>(ctx, a, g.tw_x)))); break;
          case 3: MRL_SWITCH_N(nx, MRL_TRY((p2::launch_xfused<NN, 3, false>(ctx, a, g.tw_x)))); break;
          default: MRL_SWITCH_N(nx, MRL_TRY((p2::launch_xfused<NN, 4, false>(ctx, a, g.tw_x)))); break;
        }
      }
    }
    if (!xchunk) MRL_TRY(launch_d(0, nx));   // (chunked schedule: it runs in front of the next substep's z pass, or of the final one below)
    if (advance && k < count - 1) {   // TensorSolver.C:105-106
      *head = slot_new;
      if (*n_old < pred) *n_old += 1;
    }
  }
  if (xchunk) {   // the last substep's inverse y pass and the final inverse z pass, chunk after chunk
    for (long long x0 = 0; x0 < nx; x0 += xchunk) {
      const long long x1 = x0 + xchunk < nx ? x0 + xchunk : nx;
      MRL_TRY(launch_d(x0, x1));
      ProfScope ps(ctx, "ch_E_z_inv", (h + 8.0 * nreal) * (double)(x1 - x0) / (double)nx);
      MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_inv<NN>(ctx, w_c + x0 * plane, c_out + x0 * ny * nz, scale, (x1 - x0) * ny / 2, g.zl))));
    }
    return MRL_OK;
  }
  ProfScope ps(ctx, "ch_E_z_inv", h + 8.0 * nreal);
  MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_inv<NN>(ctx, w_c, c_out, scale, nx * ny / 2, g.zl))));
  return MRL_OK;
}

}  // namespace mrl
