// Mechanics post-processing on the spectral path: ComputeDisplacements (src/tensor_computes/ComputeDisplacements.C:53-107)
// and ComputeVonMisesStress (src/tensor_computes/ComputeVonMisesStress.C:31-66).
//
// Displacements: u = (<F> - I) X + ifft( (fft(F - <F>) . (-i q)) / |q|^2 ), then interpolated from the n cell centres to
// n + 1 "nodes" per axis (torch interpolate, align_corners = true).  Here: one component reduction for <F>, the batched
// forward transform of F (the k = 0 mode, the only one <F> touches, is zeroed by the k-space kernel as the reference's
// where(denom == 0) does), one k-space contraction kernel writing D spectra instead of D*D, the batched inverse transform and one
// kernel that adds the affine part and interpolates -- the reference materialises ~10 full-size temporaries on the way.
#include "mrl_internal.h"

namespace mrl {

static int grid_of(long long n) {
  long long b = (n + 255) / 256;
  if (b > 65536) b = 65536;
  return b < 1 ? 1 : (int)b;
}

// ubar_i = sum_j H_ij (-i q_j) / |q|^2, 0 at q = 0 ; value-major in [e][D][D], out [e][D]
template <int D>
__global__ void __launch_bounds__(256) k_disp_kspace(const double2 *__restrict__ spec, double2 *__restrict__ ubar, long long n0,
                                                      long long n1, long long n2, const double *__restrict__ k0,
                                                      const double *__restrict__ k1, const double *__restrict__ k2) {
#pragma clang fp contract(off)
  const long long total = n0 * n1 * n2;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long i2 = e % n2, t = e / n2, i1 = t % n1, i0 = t / n1;
    const double kk[3] = {k0[i0], k1[i1], k2[i2]};
    double q[D], Q = 0.0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      q[d] = kk[3 - D + d];
      Q += q[d] * q[d];
    }
    const double inv = (Q == 0.0) ? 0.0 : 1.0 / Q;
#pragma unroll
    for (int i = 0; i < D; ++i) {
      double2 s = make_double2(0.0, 0.0);
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const double2 h = spec[e * D * D + i * D + j];
        s.x += h.y * q[j];      // (a + b i)(-i q) = b q - a q i
        s.y -= h.x * q[j];
      }
      ubar[e * D + i] = make_double2(s.x * inv, s.y * inv);
    }
  }
}

// node values: linear interpolation (align_corners) of  u_aff + u_per  from cells to n + 1 points per axis.
// internal axes are right-aligned (user axis d = internal 3 - D + d); sums[] = component sums of F (device scalars)
template <int D>
__global__ void __launch_bounds__(256) k_disp_nodes(const double *__restrict__ uper, const double *__restrict__ sums,
                                                     double inv_npts, long long n0, long long n1, long long n2,
                                                     const double *__restrict__ x0, const double *__restrict__ x1,
                                                     const double *__restrict__ x2, double *__restrict__ out) {
#pragma clang fp contract(off)
  const long long nn[3] = {n0, n1, n2};
  long long m[3];  // node counts per internal axis (unused leading axes stay 1)
#pragma unroll
  for (int a = 0; a < 3; ++a) m[a] = (a >= 3 - D) ? nn[a] + 1 : 1;
  double A[D][D];
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) A[i][j] = sums[i * D + j] * inv_npts - (i == j ? 1.0 : 0.0);
  const long long total = m[0] * m[1] * m[2];
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    long long o[3];
    o[2] = e % m[2];
    o[1] = (e / m[2]) % m[1];
    o[0] = e / (m[2] * m[1]);
    long long i0[3], i1[3];
    double w0[3], w1[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (a >= 3 - D) {
        // at::native::area_pixel_compute_scale / guard_index_and_lambda with align_corners = true
        const double scale = nn[a] > 0 ? (double)(nn[a] - 1) / (double)nn[a] : 0.0;   // (in - 1) / (out - 1), out = in + 1
        const double src = scale * (double)o[a];
        long long lo = (long long)src;
        if (lo > nn[a] - 1) lo = nn[a] - 1;
        const long long hi = lo + 1 < nn[a] ? lo + 1 : nn[a] - 1;
        double l1 = src - (double)lo;
        l1 = l1 < 0.0 ? 0.0 : (l1 > 1.0 ? 1.0 : l1);
        i0[a] = lo;
        i1[a] = hi;
        w1[a] = l1;
        w0[a] = 1.0 - l1;
      } else {
        i0[a] = i1[a] = 0;
        w0[a] = 1.0;
        w1[a] = 0.0;
      }
    }
    // nested lerp, outermost = first user axis (as cpu_upsample_linear's recursive interpolate<> does)
    double acc0[D], acc1[D], acc2[D];
#pragma unroll
    for (int c = 0; c < D; ++c) acc0[c] = 0.0;
    for (int s0 = 0; s0 < (D >= 3 ? 2 : 1); ++s0) {
      const long long c0 = s0 ? i1[0] : i0[0];
#pragma unroll
      for (int c = 0; c < D; ++c) acc1[c] = 0.0;
      for (int s1 = 0; s1 < (D >= 2 ? 2 : 1); ++s1) {
        const long long c1 = s1 ? i1[1] : i0[1];
#pragma unroll
        for (int c = 0; c < D; ++c) acc2[c] = 0.0;
        for (int s2 = 0; s2 < 2; ++s2) {
          const long long c2 = s2 ? i1[2] : i0[2];
          const long long cell = (c0 * n1 + c1) * n2 + c2;
          const double xc[3] = {x0[c0], x1[c1], x2[c2]};
          const double ws = s2 ? w1[2] : w0[2];
#pragma unroll
          for (int c = 0; c < D; ++c) {
            double aff = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) aff += A[c][j] * xc[3 - D + j];
            const double v = aff + uper[cell * D + c];
            acc2[c] = s2 ? acc2[c] + ws * v : ws * v;
          }
        }
        const double wm = (D >= 2) ? (s1 ? w1[1] : w0[1]) : 1.0;
#pragma unroll
        for (int c = 0; c < D; ++c) acc1[c] = (D >= 2) ? (s1 ? acc1[c] + wm * acc2[c] : wm * acc2[c]) : acc2[c];
      }
      const double wo = (D >= 3) ? (s0 ? w1[0] : w0[0]) : 1.0;
#pragma unroll
      for (int c = 0; c < D; ++c) acc0[c] = (D >= 3) ? (s0 ? acc0[c] + wo * acc1[c] : wo * acc1[c]) : acc1[c];
    }
#pragma unroll
    for (int c = 0; c < D; ++c) out[e * D + c] = acc0[c];
  }
}

template <int D>
__global__ void __launch_bounds__(256) k_von_mises(const double *__restrict__ S, double *__restrict__ out, long long npts) {
#pragma clang fp contract(off)
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < npts; e += (long long)gridDim.x * 256) {
    const double *s = S + e * D * D;
    if (D == 3) {
      const double xx = s[0], yy = s[4], zz = s[8], xy = s[1], yz = s[5], zx = s[6];
      const double t1 = (xx - yy) * (xx - yy), t2 = (yy - zz) * (yy - zz), t3 = (zz - xx) * (zz - xx);
      const double t4 = 6.0 * ((xy * xy + yz * yz) + zx * zx);
      out[e] = sqrt(0.5 * (((t1 + t2) + t3) + t4));
    } else {
      const double xx = s[0], yy = s[3], xy = s[1];
      out[e] = sqrt(0.5 * ((xx - yy) * (xx - yy) + 6.0 * (xy * xy)));
    }
  }
}

}  // namespace mrl

using namespace mrl;

extern "C" {

int mrl_mech_displacements(mrl_ctx *ctx, const double *d_F, double *d_disp) {
  if (!ctx) return MRL_ERR_INVALID;
  if (ctx->dim != 2 && ctx->dim != 3) return set_error(ctx, MRL_ERR_UNSUPPORTED, "mrl_mech_displacements: 2-D or 3-D domains");
  if (ctx->slab || ctx->pencil || ctx->spectrum != MRL_SPECTRUM_HALF)
    return set_error(ctx, MRL_ERR_UNSUPPORTED, "mrl_mech_displacements: serial half-spectrum contexts only");
  if (!d_F || !d_disp) return set_error(ctx, MRL_ERR_INVALID, "mrl_mech_displacements: null buffer");
  const int D = ctx->dim, dd = D * D;
  const long long npts = real_count_local(ctx), nspec = spec_count_local(ctx);
  MRL_TRY(ensure_work(ctx, 4, sizeof(cplx) * nspec * dd));
  MRL_TRY(ensure_work(ctx, 5, sizeof(cplx) * nspec * D));
  MRL_TRY(ensure_work(ctx, 6, sizeof(double) * (size_t)(npts * D + 2)));
  double *spec = ctx->d_work[4], *ubar = ctx->d_work[5], *uper = ctx->d_work[6];
  double *sums = ctx->d_red + kScalarBase + 16;
  MRL_TRY(component_sums_async(ctx, d_F, npts, dd, sums));
  MRL_TRY(fft_forward_serial(ctx, d_F, spec, dd, 1));
  {
    ProfScope ps(ctx, "disp_kspace");
    const int nb = grid_of(nspec);
    if (D == 3)
      hipLaunchKernelGGL(k_disp_kspace<3>, dim3(nb), dim3(256), 0, ctx->stream, reinterpret_cast<const double2 *>(spec),
                         reinterpret_cast<double2 *>(ubar), ctx->nrec[0], ctx->nrec[1], ctx->nrec[2], ctx->d_k[0], ctx->d_k[1],
                         ctx->d_k[2]);
    else
      hipLaunchKernelGGL(k_disp_kspace<2>, dim3(nb), dim3(256), 0, ctx->stream, reinterpret_cast<const double2 *>(spec),
                         reinterpret_cast<double2 *>(ubar), ctx->nrec[0], ctx->nrec[1], ctx->nrec[2], ctx->d_k[0], ctx->d_k[1],
                         ctx->d_k[2]);
    MRL_HIP(ctx, hipGetLastError());
  }
  MRL_TRY(fft_inverse_serial(ctx, ubar, uper, D, 1));
  {
    ProfScope ps(ctx, "disp_nodes");
    long long nodes = 1;
    for (int a = 3 - D; a < 3; ++a) nodes *= ctx->nloc[a] + 1;
    const int nb = grid_of(nodes);
    if (D == 3)
      hipLaunchKernelGGL(k_disp_nodes<3>, dim3(nb), dim3(256), 0, ctx->stream, uper, sums, 1.0 / (double)npts, ctx->nloc[0],
                         ctx->nloc[1], ctx->nloc[2], ctx->d_x[0], ctx->d_x[1], ctx->d_x[2], d_disp);
    else
      hipLaunchKernelGGL(k_disp_nodes<2>, dim3(nb), dim3(256), 0, ctx->stream, uper, sums, 1.0 / (double)npts, ctx->nloc[0],
                         ctx->nloc[1], ctx->nloc[2], ctx->d_x[0], ctx->d_x[1], ctx->d_x[2], d_disp);
    MRL_HIP(ctx, hipGetLastError());
  }
  return MRL_OK;
}

int mrl_mech_von_mises(mrl_ctx *ctx, const double *d_stress, double *d_out) {
  if (!ctx) return MRL_ERR_INVALID;
  if (ctx->dim != 2 && ctx->dim != 3) return set_error(ctx, MRL_ERR_UNSUPPORTED, "Unsupported problem dimension %d", ctx->dim);
  if (!d_stress || !d_out) return set_error(ctx, MRL_ERR_INVALID, "mrl_mech_von_mises: null buffer");
  const long long npts = real_count_local(ctx);
  ProfScope ps(ctx, "von_mises");
  if (ctx->dim == 3)
    hipLaunchKernelGGL(k_von_mises<3>, dim3(grid_of(npts)), dim3(256), 0, ctx->stream, d_stress, d_out, npts);
  else
    hipLaunchKernelGGL(k_von_mises<2>, dim3(grid_of(npts)), dim3(256), 0, ctx->stream, d_stress, d_out, npts);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

}  // extern "C"
