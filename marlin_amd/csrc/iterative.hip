// Iterative split-operator solvers: the k-space work of SecantSolver::substep (src/tensor_solver/SecantSolver.C:60-176)
// as two fused kernels per variable -- the reference spends ~15 elementwise ATen kernels, as many full-size temporaries
// and one host-synchronising norm per iteration; here one kernel reads u, u_prev, R_prev, N, L, u_old once, writes the
// new residual and the new iterate and leaves the partial sums of |R|^2 and |du|^2 for a single fold.
#include "mrl_internal.h"

namespace mrl {

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
#pragma clang fp contract(off)
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// c10::complex<double>::operator/= (Smith's algorithm as numpy does it)
__device__ __forceinline__ double2 cdiv(double2 n, double2 q) {
#pragma clang fp contract(off)
  const double a = n.x, b = n.y, c = q.x, d = q.y;
  const double ac = fabs(c), ad = fabs(d);
  if (ac >= ad) {
    if (ac == 0.0 && ad == 0.0) return make_double2(a / ac, b / ad);
    const double rat = d / c, scl = 1.0 / (c + d * rat);
    return make_double2((a + b * rat) * scl, (b - a * rat) * scl);
  }
  const double rat = c / d, scl = 1.0 / (d + c * rat);
  return make_double2((a * rat + b) * scl, (b * rat - a) * scl);
}

// R0 = (N + L*u)*dt ; guess = (u + eps*N) / (1 - eps*L)                                       SecantSolver.C:79-101
__global__ void __launch_bounds__(256) k_secant_begin(const double2 *__restrict__ u, const double2 *__restrict__ N,
                                                       const double *__restrict__ L, double dt, double eps,
                                                       double2 *__restrict__ R0, double2 *__restrict__ guess, long long n,
                                                       double *__restrict__ partial) {
#pragma clang fp contract(off)
  __shared__ double sh[4];
  double acc = 0.0;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const double2 ue = u[e], Ne = N[e];
    double2 r, g;
    if (L) {
      const double l = L[e];
      r = make_double2((Ne.x + l * ue.x) * dt, (Ne.y + l * ue.y) * dt);
      const double scl = 1.0 / (1.0 - eps * l);
      g = make_double2((ue.x + eps * Ne.x) * scl, (ue.y + eps * Ne.y) * scl);
    } else {
      r = make_double2(Ne.x * dt, Ne.y * dt);
      g = make_double2(ue.x + eps * Ne.x, ue.y + eps * Ne.y);
    }
    R0[e] = r;
    guess[e] = g;
    acc += r.x * r.x + r.y * r.y;
  }
  const double s = block_sum256(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// R = (N + L*u)*dt + u_old - u ; du = where(R - Rprev != 0, -R*(u - uprev)/(R - Rprev), 0) ; unew = u + du*damping   :121-140
__global__ void __launch_bounds__(256) k_secant_iterate(const double2 *__restrict__ u, const double2 *__restrict__ N,
                                                         const double *__restrict__ L, const double2 *__restrict__ uold,
                                                         const double2 *__restrict__ uprev, double2 *__restrict__ Rprev,
                                                         double dt, double damping, double2 *__restrict__ unew, long long n,
                                                         double *__restrict__ partial) {
#pragma clang fp contract(off)
  __shared__ double sh[4];
  double accR = 0.0, accD = 0.0;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const double2 ue = u[e], Ne = N[e], uo = uold[e], up = uprev[e], Rp = Rprev[e];
    double2 R;
    if (L) {
      const double l = L[e];
      R = make_double2((Ne.x + l * ue.x) * dt, (Ne.y + l * ue.y) * dt);
    } else {
      R = make_double2(Ne.x * dt, Ne.y * dt);
    }
    R = make_double2(R.x + uo.x - ue.x, R.y + uo.y - ue.y);
    const double2 dx = make_double2(ue.x - up.x, ue.y - up.y), dy = make_double2(R.x - Rp.x, R.y - Rp.y);
    double2 du = make_double2(0.0, 0.0);
    if (dy.x != 0.0 || dy.y != 0.0) du = cdiv(cmul(make_double2(-R.x, -R.y), dx), dy);
    Rprev[e] = R;
    unew[e] = damping == 1.0 ? make_double2(ue.x + du.x, ue.y + du.y)
                             : make_double2(ue.x + du.x * damping, ue.y + du.y * damping);
    accR += R.x * R.x + R.y * R.y;
    accD += du.x * du.x + du.y * du.y;
  }
  const double sR = block_sum256(accR, sh);
  const double sD = block_sum256(accD, sh);
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = sR;
    partial[2 * blockIdx.x + 1] = sD;
  }
}

static int blocks_for(long long n) {
  long long b = (n + 255) / 256;
  if (b > kRedBlocks / 2) b = kRedBlocks / 2;
  return b < 1 ? 1 : (int)b;
}

}  // namespace mrl

using namespace mrl;

extern "C" {

int mrl_secant_begin(mrl_ctx *ctx, const double *d_u, const double *d_N, const double *d_L, double sub_dt, double dt_epsilon,
                     double *d_R0_out, double *d_guess_out, double *h_sumsq, int64_t n_spec) {
  if (!ctx) return MRL_ERR_INVALID;
  if (!d_u || !d_N || !d_R0_out || !d_guess_out || !h_sumsq || n_spec < 0)
    return set_error(ctx, MRL_ERR_INVALID, "mrl_secant_begin: bad argument");
  h_sumsq[0] = 0.0;
  if (n_spec == 0) return MRL_OK;
  const int nb = blocks_for(n_spec);
  double *slot = ctx->d_red + kScalarBase;
  {
    ProfScope ps(ctx, "secant_begin");
    hipLaunchKernelGGL(k_secant_begin, dim3(nb), dim3(256), 0, ctx->stream, reinterpret_cast<const double2 *>(d_u),
                       reinterpret_cast<const double2 *>(d_N), d_L, sub_dt, dt_epsilon, reinterpret_cast<double2 *>(d_R0_out),
                       reinterpret_cast<double2 *>(d_guess_out), (long long)n_spec, ctx->d_red);
    MRL_HIP(ctx, hipGetLastError());
  }
  MRL_TRY(reduce_finalize(ctx, nb, 1, slot));
  return read_scalars(ctx, slot, 1, h_sumsq);
}

int mrl_secant_iterate(mrl_ctx *ctx, const double *d_u, const double *d_N, const double *d_L, const double *d_u_old,
                       const double *d_u_prev, double *d_R_prev, double sub_dt, double damping, double *d_u_new,
                       double *h_sumsq, int64_t n_spec) {
  if (!ctx) return MRL_ERR_INVALID;
  if (!d_u || !d_N || !d_u_old || !d_u_prev || !d_R_prev || !d_u_new || !h_sumsq || n_spec < 0)
    return set_error(ctx, MRL_ERR_INVALID, "mrl_secant_iterate: bad argument");
  if (d_u_new == d_u_prev || d_u_new == d_u_old)
    return set_error(ctx, MRL_ERR_INVALID, "mrl_secant_iterate: the new iterate may alias d_u only");
  h_sumsq[0] = h_sumsq[1] = 0.0;
  if (n_spec == 0) return MRL_OK;
  const int nb = blocks_for(n_spec);
  double *slot = ctx->d_red + kScalarBase;
  {
    ProfScope ps(ctx, "secant_iterate");
    hipLaunchKernelGGL(k_secant_iterate, dim3(nb), dim3(256), 0, ctx->stream, reinterpret_cast<const double2 *>(d_u),
                       reinterpret_cast<const double2 *>(d_N), d_L, reinterpret_cast<const double2 *>(d_u_old),
                       reinterpret_cast<const double2 *>(d_u_prev), reinterpret_cast<double2 *>(d_R_prev), sub_dt, damping,
                       reinterpret_cast<double2 *>(d_u_new), (long long)n_spec, ctx->d_red);
    MRL_HIP(ctx, hipGetLastError());
  }
  MRL_TRY(reduce_finalize(ctx, nb, 2, slot));
  return read_scalars(ctx, slot, 2, h_sumsq);
}

}  // extern "C"
