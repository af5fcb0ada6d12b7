// Homogeneous small-strain elasticity coupled to a concentration field through a volumetric eigenstrain e0*c
// (test/tests/tensor_compute/coupled_pf_mech.i):
//   FFTQuasistaticElasticity   (src/tensor_computes/FFTQuasistaticElasticity.C:46-104): per k-point the 3 x 3 system
//       A u-hat = b,  A = (2 mu + lambda) k_i k_i (diagonal) + mu (k^2 - k_i k_i) ... as written there, k = 2 pi i * axis,
//       b = k * (2 e0 (3 lambda + mu) c-hat), A_ii = 1 and b = 0 at k = 0; the displacements are the inverse transforms of u-hat.
//   FFTElasticChemicalPotential (src/tensor_computes/FFTElasticChemicalPotential.C:47-61): the elastic contribution to the
//       chemical potential in reciprocal space from c-hat and the transformed displacements.
// The reference materialises A as a [grid][3][3] complex tensor (144 B per k-point) plus ~15 full-size temporaries and calls
// at::linalg_solve; every entry of A is real (products of two purely imaginary numbers), so here one kernel builds A in
// registers, solves by LU with partial pivoting (what gesv does) and writes the three spectra, 16 B read + 48 B written per k-point.
#include "mrl_internal.h"

namespace mrl {

static int pf_grid(long long n) {
  long long b = (n + 255) / 256;
  if (b > 65536) b = 65536;
  return b < 1 ? 1 : (int)b;
}

__global__ void __launch_bounds__(256) k_qs_elasticity(const double2 *__restrict__ cbar, double2 *__restrict__ ux,
                                                        double2 *__restrict__ uy, double2 *__restrict__ uz, long long n0,
                                                        long long n1, long long n2, const double *__restrict__ k0,
                                                        const double *__restrict__ k1, const double *__restrict__ k2,
                                                        double mu, double lambda, double e0) {
#pragma clang fp contract(off)
  const double two_pi = 2.0 * 3.14159265358979323846;
  const long long total = n0 * n1 * n2;
  const double ul = 2.0 * mu + lambda, lm = lambda + mu, ef = 3.0 * lambda + mu, te = 2.0 * e0;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long i2 = e % n2, t = e / n2, i1 = t % n1, i0 = t / n1;
    // k_d = (0, a_d): every product of two of them is the real number -a_i a_j
    const double a[3] = {two_pi * k0[i0], two_pi * k1[i1], two_pi * k2[i2]};
    double A[3][3];
    A[0][0] = (-((ul * a[0]) * a[0]) + -((mu * a[1]) * a[1])) + -((mu * a[2]) * a[2]);
    A[1][1] = (-((ul * a[1]) * a[1]) + -((mu * a[0]) * a[0])) + -((mu * a[2]) * a[2]);
    A[2][2] = (-((ul * a[2]) * a[2]) + -((mu * a[0]) * a[0])) + -((mu * a[1]) * a[1]);
    A[0][1] = A[1][0] = -((lm * a[0]) * a[1]);
    A[0][2] = A[2][0] = -((lm * a[0]) * a[2]);
    A[1][2] = A[2][1] = -((lm * a[1]) * a[2]);
    double2 ev = cbar[e];
    ev = make_double2((te * ev.x) * ef, (te * ev.y) * ef);
    if (e == 0) {  // |k| = 0
      A[0][0] = A[1][1] = A[2][2] = 1.0;
      ev = make_double2(0.0, 0.0);
    }
    // b_d = k_d * e = (0, a_d)(er, ei) = (-a_d ei, a_d er)
    double br[3], bi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      br[d] = -(a[d] * ev.y);
      bi[d] = a[d] * ev.x;
    }
    // LU with partial pivoting; row swaps by selects so that A stays in registers
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      int piv = c;
      double best = fabs(A[c][c]);
#pragma unroll
      for (int r = c + 1; r < 3; ++r)
        if (fabs(A[r][c]) > best) {
          best = fabs(A[r][c]);
          piv = r;
        }
#pragma unroll
      for (int r = c + 1; r < 3; ++r)
        if (piv == r) {
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const double tmp = A[c][j];
            A[c][j] = A[r][j];
            A[r][j] = tmp;
          }
          double tmp = br[c];
          br[c] = br[r];
          br[r] = tmp;
          tmp = bi[c];
          bi[c] = bi[r];
          bi[r] = tmp;
        }
      const double inv = 1.0 / A[c][c];
#pragma unroll
      for (int r = c + 1; r < 3; ++r) {
        const double f = A[r][c] * inv;
#pragma unroll
        for (int j = c + 1; j < 3; ++j) A[r][j] -= f * A[c][j];
        br[r] -= f * br[c];
        bi[r] -= f * bi[c];
      }
    }
    double xr[3], xi[3];
#pragma unroll
    for (int c = 2; c >= 0; --c) {
      double sr = br[c], si = bi[c];
#pragma unroll
      for (int j = c + 1; j < 3; ++j) {
        sr -= A[c][j] * xr[j];
        si -= A[c][j] * xi[j];
      }
      xr[c] = sr / A[c][c];
      xi[c] = si / A[c][c];
    }
    ux[e] = make_double2(xr[0], xi[0]);
    uy[e] = make_double2(xr[1], xi[1]);
    uz[e] = make_double2(xr[2], xi[2]);
  }
}

// out = -e0 * ( e0 * (9 lambda cbar + (mu 6) cbar) - (2 mu + 3 lambda) * ((kx ux + ky uy) + kz uz) ),  k_d = (0, a_d)
__global__ void __launch_bounds__(256) k_elastic_mu(const double2 *__restrict__ cbar, const double2 *__restrict__ ux,
                                                     const double2 *__restrict__ uy, const double2 *__restrict__ uz,
                                                     double2 *__restrict__ out, long long n0, long long n1, long long n2,
                                                     const double *__restrict__ k0, const double *__restrict__ k1,
                                                     const double *__restrict__ k2, double mu, double lambda, double e0) {
#pragma clang fp contract(off)
  const double two_pi = 2.0 * 3.14159265358979323846;
  const long long total = n0 * n1 * n2;
  const double l9 = 9.0 * lambda, m6 = mu * 6.0, f = 2.0 * mu + 3.0 * lambda, me0 = -e0;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long i2 = e % n2, t = e / n2, i1 = t % n1, i0 = t / n1;
    const double ax = two_pi * k0[i0], ay = two_pi * k1[i1], az = two_pi * k2[i2];
    const double2 c = cbar[e], x = ux[e], y = uy[e], z = uz[e];
    // (0, a)(ur, ui) = (-a ui, a ur)
    const double dr = (-(ax * x.y) + -(ay * y.y)) + -(az * z.y);
    const double di = (ax * x.x + ay * y.x) + az * z.x;
    const double tr = e0 * (l9 * c.x + m6 * c.x) - f * dr;
    const double ti = e0 * (l9 * c.y + m6 * c.y) - f * di;
    out[e] = make_double2(me0 * tr, me0 * ti);
  }
}

static int check_pf(mrl_ctx *ctx, const char *what) {
  if (ctx->dim != 3) return set_error(ctx, MRL_ERR_UNSUPPORTED, "%s: 3-D domains only (one displacement per dimension, k = 0 at index {0,0,0})", what);
  if (ctx->slab || ctx->pencil || ctx->spectrum != MRL_SPECTRUM_HALF)
    return set_error(ctx, MRL_ERR_UNSUPPORTED, "%s: serial half-spectrum contexts only", what);
  return MRL_OK;
}

}  // namespace mrl

using namespace mrl;

extern "C" {

int mrl_qs_elasticity(mrl_ctx *ctx, const double *d_cbar, double mu, double lambda, double e0, double *const *d_disp) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_TRY(check_pf(ctx, "mrl_qs_elasticity"));
  if (!d_cbar || !d_disp || !d_disp[0] || !d_disp[1] || !d_disp[2]) return set_error(ctx, MRL_ERR_INVALID, "mrl_qs_elasticity: null buffer");
  const long long nspec = spec_count_local(ctx);
  MRL_TRY(ensure_work(ctx, 4, sizeof(cplx) * nspec * 3));
  double2 *u = reinterpret_cast<double2 *>(ctx->d_work[4]);
  {
    ProfScope ps(ctx, "qs_elasticity_solve", 64.0 * nspec);
    hipLaunchKernelGGL(k_qs_elasticity, dim3(pf_grid(nspec)), dim3(256), 0, ctx->stream, reinterpret_cast<const double2 *>(d_cbar), u,
                       u + nspec, u + 2 * nspec, ctx->nrec[0], ctx->nrec[1], ctx->nrec[2], ctx->d_k[0], ctx->d_k[1], ctx->d_k[2], mu,
                       lambda, e0);
    MRL_HIP(ctx, hipGetLastError());
  }
  for (int d = 0; d < 3; ++d) MRL_TRY(fft_inverse_serial(ctx, reinterpret_cast<const double *>(u + d * nspec), d_disp[d], 1, 0));
  return MRL_OK;
}

int mrl_elastic_chemical_potential(mrl_ctx *ctx, const double *d_cbar, const double *const *d_disp, double mu, double lambda,
                                   double e0, double *d_out) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_TRY(check_pf(ctx, "mrl_elastic_chemical_potential"));
  if (!d_cbar || !d_out || !d_disp || !d_disp[0] || !d_disp[1] || !d_disp[2])
    return set_error(ctx, MRL_ERR_INVALID, "mrl_elastic_chemical_potential: null buffer");
  const long long nspec = spec_count_local(ctx);
  MRL_TRY(ensure_work(ctx, 4, sizeof(cplx) * nspec * 3));
  double2 *u = reinterpret_cast<double2 *>(ctx->d_work[4]);
  for (int d = 0; d < 3; ++d) MRL_TRY(fft_forward_serial(ctx, d_disp[d], reinterpret_cast<double *>(u + d * nspec), 1, 0));
  ProfScope ps(ctx, "elastic_chemical_potential", 80.0 * nspec);
  hipLaunchKernelGGL(k_elastic_mu, dim3(pf_grid(nspec)), dim3(256), 0, ctx->stream, reinterpret_cast<const double2 *>(d_cbar), u,
                     u + nspec, u + 2 * nspec, reinterpret_cast<double2 *>(d_out), ctx->nrec[0], ctx->nrec[1], ctx->nrec[2],
                     ctx->d_k[0], ctx->d_k[1], ctx->d_k[2], mu, lambda, e0);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

}  // extern "C"
