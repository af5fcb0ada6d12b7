// Pass sequencing of the 1-3 D transforms: DomainAction::fftSerial / ifft
// (src/actions/DomainAction.C:853-867, 1049-1063).
#include "mrl_internal.h"

namespace mrl {

static void fill_radix(PassDesc &d, const AxisPlan &ax, int sign) {
  d.n = ax.n;
  d.npass = (int)ax.radix.size();
  for (int i = 0; i < d.npass; ++i) d.radix[i] = ax.radix[i];
  d.sign = sign;
  d.scale = 1.0;
  d.in_kind = 0;
  d.out_kind = 0;
  d.nout = ax.n;
  d.tile = 1;
}

// strides of a [A0][A1][Az] array with value-major batch B (B = 1 and sb = count for field-major)
struct Lay {
  long long s0, s1, s2, sb;
};
static Lay layout_of(long long A1, long long Az, long long count, long long batch, int layout) {
  Lay l;
  if (layout == 1) {
    l.s2 = batch;
    l.s1 = Az * batch;
    l.s0 = A1 * Az * batch;
    l.sb = 1;
  } else {
    l.s2 = 1;
    l.s1 = Az;
    l.s0 = A1 * Az;
    l.sb = count;
  }
  return l;
}

// z pass forward: real [A0*A1 lines][A2] -> complex [..][nzc]
int pass_z_forward(mrl_ctx *ctx, const double *d_in, double *d_out, long long A0, long long A1, long long batch, int layout) {
  const long long A2 = ctx->n[2], nzc = ctx->nrec[2];
  const Lay li = layout_of(A1, A2, A0 * A1 * A2, batch, layout);
  const Lay lo = layout_of(A1, nzc, A0 * A1 * nzc, batch, layout);
  PassDesc d{};
  fill_radix(d, ctx->ax[2], -1);
  d.inner = 1;
  d.outer = A0 * A1;
  d.in_so = li.s1;
  d.in_si = 0;
  d.in_sn = li.s2;
  d.in_sb = li.sb;
  d.out_so = lo.s1;
  d.out_si = 0;
  d.out_sn = lo.s2;
  d.out_sb = lo.sb;
  d.in_kind = 1;
  d.nout = (int)nzc;
  d.lines_fastest = 0;
  return launch_pass(ctx, d, d_in, d_out, ctx->ax[2].d_tw, batch);
}

// c2c pass along internal axis a (0 or 1) of a complex [A0][A1][nzc] array
int pass_strided(mrl_ctx *ctx, int a, int sign, const double *d_in, double *d_out, long long A0, long long A1,
                 long long nzc, long long batch, int layout) {
  const Lay l = layout_of(A1, nzc, A0 * A1 * nzc, batch, layout);
  PassDesc d{};
  fill_radix(d, ctx->ax[a], sign);
  if (a == 1) {
    d.inner = nzc;
    d.outer = A0;
    d.in_so = d.out_so = l.s0;
    d.in_si = d.out_si = l.s2;
    d.in_sn = d.out_sn = l.s1;
  } else {
    d.inner = A1 * nzc;
    d.outer = 1;
    d.in_so = d.out_so = 0;
    d.in_si = d.out_si = l.s2;
    d.in_sn = d.out_sn = l.s0;
  }
  d.in_sb = d.out_sb = l.sb;
  d.lines_fastest = (layout == 1) ? 0 : 1;
  if (layout == 1) d.lines_fastest = 1;  // consecutive lines are batch-strided but still closest in memory
  return launch_pass(ctx, d, d_in, d_out, ctx->ax[a].d_tw, batch);
}

// c2c pass along `axis` over an explicit set of lines: line (o, i), o < outer, i < inner, starts at o*so + i*si,
// successive points are sn apart (complex elements); used on kz sub-ranges by the slab pipeline
int pass_lines(mrl_ctx *ctx, int axis, int sign, const double *in, double *out, long long outer, long long inner,
               long long so, long long si, long long sn, int lines_fastest) {
  PassDesc d{};
  fill_radix(d, ctx->ax[axis], sign);
  d.inner = inner;
  d.outer = outer;
  d.in_so = d.out_so = so;
  d.in_si = d.out_si = si;
  d.in_sn = d.out_sn = sn;
  d.in_sb = d.out_sb = 0;
  d.lines_fastest = lines_fastest;
  return launch_pass(ctx, d, in, out, ctx->ax[axis].d_tw, 1);
}

// z pass inverse: complex [..][nzc] -> real [..][A2], scaled
int pass_z_inverse(mrl_ctx *ctx, const double *d_in, double *d_out, long long A0, long long A1, long long batch,
                   int layout, double scale) {
  const long long A2 = ctx->n[2], nzc = ctx->nrec[2];
  const Lay li = layout_of(A1, nzc, A0 * A1 * nzc, batch, layout);
  const Lay lo = layout_of(A1, A2, A0 * A1 * A2, batch, layout);
  PassDesc d{};
  fill_radix(d, ctx->ax[2], +1);
  d.inner = 1;
  d.outer = A0 * A1;
  d.in_so = li.s1;
  d.in_si = 0;
  d.in_sn = li.s2;
  d.in_sb = li.sb;
  d.out_so = lo.s1;
  d.out_si = 0;
  d.out_sn = lo.s2;
  d.out_sb = lo.sb;
  d.in_kind = ctx->spectrum == MRL_SPECTRUM_HALF ? 2 : 0;
  d.out_kind = 2;
  d.nout = (int)A2;
  d.scale = scale;
  d.lines_fastest = 0;
  return launch_pass(ctx, d, d_in, d_out, ctx->ax[2].d_tw, batch);
}

int fft_forward_serial(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch, int layout) {
  const long long A0 = ctx->n[0], A1 = ctx->n[1], nzc = ctx->nrec[2];
  {
    ProfScope ps(ctx, "fft_z_fwd_generic", (double)batch * (8.0 * A0 * A1 * ctx->n[2] + 16.0 * A0 * A1 * nzc));
    MRL_TRY(pass_z_forward(ctx, d_in, d_out, A0, A1, batch, layout));
  }
  if (A1 > 1) {
    ProfScope ps(ctx, "fft_y_generic", (double)batch * 32.0 * A0 * A1 * nzc);
    MRL_TRY(pass_strided(ctx, 1, -1, d_out, d_out, A0, A1, nzc, batch, layout));
  }
  if (A0 > 1) {
    ProfScope ps(ctx, "fft_x_generic", (double)batch * 32.0 * A0 * A1 * nzc);
    MRL_TRY(pass_strided(ctx, 0, -1, d_out, d_out, A0, A1, nzc, batch, layout));
  }
  return MRL_OK;
}

int fft_inverse_serial(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch, int layout) {
  const long long A0 = ctx->n[0], A1 = ctx->n[1], A2 = ctx->n[2], nzc = ctx->nrec[2];
  const double scale = 1.0 / ((double)A0 * (double)A1 * (double)A2);
  const double *cur = d_in;
  if (A0 > 1 || A1 > 1) {
    MRL_TRY(ensure_work(ctx, 0, sizeof(cplx) * A0 * A1 * nzc * batch));
    double *w = ctx->d_work[0];
    if (A0 > 1) {
      ProfScope ps(ctx, "fft_x_generic", (double)batch * 32.0 * A0 * A1 * nzc);
      MRL_TRY(pass_strided(ctx, 0, +1, cur, w, A0, A1, nzc, batch, layout));
      cur = w;
    }
    if (A1 > 1) {
      ProfScope ps(ctx, "fft_y_generic", (double)batch * 32.0 * A0 * A1 * nzc);
      MRL_TRY(pass_strided(ctx, 1, +1, cur, w, A0, A1, nzc, batch, layout));
      cur = w;
    }
  }
  ProfScope ps(ctx, "fft_z_inv_generic", (double)batch * (8.0 * A0 * A1 * A2 + 16.0 * A0 * A1 * nzc));
  return pass_z_inverse(ctx, cur, d_out, A0, A1, batch, layout, scale);
}

bool planned_unfused_ok(const mrl_ctx *ctx);                                               // ch_planned.hip
int fft_forward_planned(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch);
int fft_inverse_planned(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch);
int slab_fft_forward(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch);   // slab_driver.hip
int slab_fft_inverse(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch);
int pencil_fft_forward(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch);
int pencil_fft_inverse(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch);

}  // namespace mrl

using namespace mrl;

extern "C" {

int mrl_fft_r2c(mrl_ctx *ctx, const double *d_in, double *d_out, int64_t batch, int layout) {
  if (!ctx) return MRL_ERR_INVALID;
  if (!d_in || !d_out || batch < 1 || (layout != 0 && layout != 1))
    return set_error(ctx, MRL_ERR_INVALID, "mrl_fft_r2c: bad argument");
  if (ctx->pencil) {  // DomainAction::fftPencil (DomainAction.C:1021-1034)
    if (layout == 1 && batch > 1) return set_error(ctx, MRL_ERR_UNSUPPORTED, "mrl_fft_r2c on a pencil context: field-major batches only");
    return pencil_fft_forward(ctx, d_in, d_out, batch);
  }
  if (ctx->slab) {  // DomainAction::fftSlab with the library-owned exchange (communicator attached)
    if (layout == 1 && batch > 1) return set_error(ctx, MRL_ERR_UNSUPPORTED, "mrl_fft_r2c on a slab context: field-major batches only");
    return slab_fft_forward(ctx, d_in, d_out, batch);
  }
  if (fast_path_ok(ctx) && (layout == 0 || batch == 1)) return fft_forward_fast(ctx, d_in, d_out, batch);
  if (planned_unfused_ok(ctx) && (layout == 0 || batch == 1)) return fft_forward_planned(ctx, d_in, d_out, batch);
  return fft_forward_serial(ctx, d_in, d_out, batch, layout);
}

int mrl_fft_c2r(mrl_ctx *ctx, const double *d_in, double *d_out, int64_t batch, int layout) {
  if (!ctx) return MRL_ERR_INVALID;
  if (!d_in || !d_out || batch < 1 || (layout != 0 && layout != 1))
    return set_error(ctx, MRL_ERR_INVALID, "mrl_fft_c2r: bad argument");
  if (ctx->pencil) {  // DomainAction::ifftPencil (DomainAction.C:1036-1047)
    if (layout == 1 && batch > 1) return set_error(ctx, MRL_ERR_UNSUPPORTED, "mrl_fft_c2r on a pencil context: field-major batches only");
    return pencil_fft_inverse(ctx, d_in, d_out, batch);
  }
  if (ctx->slab) {  // DomainAction::ifftSlab
    if (layout == 1 && batch > 1) return set_error(ctx, MRL_ERR_UNSUPPORTED, "mrl_fft_c2r on a slab context: field-major batches only");
    return slab_fft_inverse(ctx, d_in, d_out, batch);
  }
  if (fast_path_ok(ctx) && (layout == 0 || batch == 1)) return fft_inverse_fast(ctx, d_in, d_out, batch);
  if (planned_unfused_ok(ctx) && (layout == 0 || batch == 1)) return fft_inverse_planned(ctx, d_in, d_out, batch);
  return fft_inverse_serial(ctx, d_in, d_out, batch, layout);
}

}  // extern "C"
