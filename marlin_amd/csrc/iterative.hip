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

// ------------------------------------------------------------------------------------------------------------------
// BroydenSolver::substep (src/tensor_solver/BroydenSolver.C:63-176): a "good Broyden" iteration per reciprocal grid point
// on the stacked residual R(u) = (N + L u) dt + u_old - u of nvar coupled variables, with a persistent nvar x nvar complex
// approximation M of the inverse Jacobian per k-point.  The reference stacks u, N, L into [grid, nvar] tensors and runs ~25
// batched matmul / where / norm kernels per iteration; here an iteration is two kernels around the compute group:
//   predict: s = -M R ; u_out_i = u_i + step * s_i
//   update : Rnew = (N + L u) dt + u_old - u ; y = Rnew - R ; d = s^T y (no conjugation, as torch::matmul) ;
//            M += where(|d| > 1e-12, (s - M y) s^T / d, 0) ; R <- Rnew ; sum |Rnew|^2
// M, R, s are field-major ([nvar*nvar][n], [nvar][n] complex) so that every access is coalesced.
namespace mrl {

constexpr int kBroydenMax = 32;   // (the reference: any N; per-thread vectors are sized 8 or 32: kernels below are instantiated for both)
struct BroydenPtrs {
  const double2 *u[kBroydenMax], *N[kBroydenMax], *uold[kBroydenMax];
  const double *L[kBroydenMax];
  double2 *uout[kBroydenMax];
};

__global__ void __launch_bounds__(256) k_broyden_init(int nv, double factor, double2 *__restrict__ M, long long n) {
  const long long total = (long long)nv * nv * n;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int ij = (int)(e / n);
    M[e] = make_double2((ij / nv == ij % nv) ? factor : 0.0, 0.0);
  }
}

// initial residual (uold == nullptr: (N + L u) dt, BroydenSolver.C:99) or the full one
__global__ void __launch_bounds__(256) k_broyden_residual(int nv, BroydenPtrs p, int with_old, double dt, double2 *__restrict__ R,
                                                           long long n, double *__restrict__ partial) {
#pragma clang fp contract(off)
  __shared__ double sh[4];
  double acc = 0.0;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    for (int i = 0; i < nv; ++i) {
      const double2 u = p.u[i][e], Nn = p.N[i][e];
      const double l = p.L[i] ? p.L[i][e] : 0.0;
      double2 r = make_double2((Nn.x + l * u.x) * dt, (Nn.y + l * u.y) * dt);
      if (with_old) {
        const double2 uo = p.uold[i][e];
        r = make_double2(r.x + uo.x - u.x, r.y + uo.y - u.y);
      }
      R[(long long)i * n + e] = r;
      acc += r.x * r.x + r.y * r.y;
    }
  }
  const double s = block_sum256(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

template <int MAXV>
__global__ void __launch_bounds__(256) k_broyden_predict(int nv, BroydenPtrs p, const double2 *__restrict__ M,
                                                          const double2 *__restrict__ R, double2 *__restrict__ S, double step,
                                                          long long n) {
#pragma clang fp contract(off)
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    double2 r[MAXV];
    for (int j = 0; j < nv; ++j) r[j] = R[(long long)j * n + e];
    for (int i = 0; i < nv; ++i) {
      double2 s = make_double2(0.0, 0.0);
      for (int j = 0; j < nv; ++j) {
        const double2 m = cmul(M[(long long)(i * nv + j) * n + e], r[j]);
        s.x += m.x;
        s.y += m.y;
      }
      s = make_double2(-s.x, -s.y);
      S[(long long)i * n + e] = s;
      const double2 u = p.u[i][e];
      p.uout[i][e] = make_double2(u.x + s.x * step, u.y + s.y * step);
    }
  }
}

template <int MAXV>
__global__ void __launch_bounds__(256) k_broyden_update(int nv, BroydenPtrs p, double2 *__restrict__ M, double2 *__restrict__ R,
                                                         const double2 *__restrict__ S, double dt, long long n,
                                                         double *__restrict__ partial) {
#pragma clang fp contract(off)
  __shared__ double sh[4];
  double acc = 0.0;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    double2 y[MAXV], s[MAXV];
    double2 d = make_double2(0.0, 0.0);
    for (int i = 0; i < nv; ++i) {
      const double2 u = p.u[i][e], Nn = p.N[i][e], uo = p.uold[i][e];
      const double l = p.L[i] ? p.L[i][e] : 0.0;
      double2 rn = make_double2((Nn.x + l * u.x) * dt, (Nn.y + l * u.y) * dt);
      rn = make_double2(rn.x + uo.x - u.x, rn.y + uo.y - u.y);
      const double2 ro = R[(long long)i * n + e];
      y[i] = make_double2(rn.x - ro.x, rn.y - ro.y);
      R[(long long)i * n + e] = rn;
      acc += rn.x * rn.x + rn.y * rn.y;
      s[i] = S[(long long)i * n + e];
      const double2 sy = cmul(s[i], y[i]);
      d.x += sy.x;
      d.y += sy.y;
    }
    if (hypot(d.x, d.y) > 1e-12) {  // torch::abs(denom) > 1e-12 (BroydenSolver.C:158)
      for (int i = 0; i < nv; ++i) {
        double2 my = make_double2(0.0, 0.0);
        for (int j = 0; j < nv; ++j) {
          const double2 t = cmul(M[(long long)(i * nv + j) * n + e], y[j]);
          my.x += t.x;
          my.y += t.y;
        }
        const double2 c = make_double2(s[i].x - my.x, s[i].y - my.y);
        for (int j = 0; j < nv; ++j) {
          const double2 upd = cdiv(cmul(c, s[j]), d);
          double2 *m = M + (long long)(i * nv + j) * n + e;
          *m = make_double2(m->x + upd.x, m->y + upd.y);
        }
      }
    }
  }
  const double sR = block_sum256(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = sR;
}

static int broyden_ptrs(mrl_ctx *ctx, int nv, const double *const *u, const double *const *N, const double *const *L,
                        const double *const *uold, double *const *uout, BroydenPtrs &p) {
  if (nv < 1 || nv > kBroydenMax) return set_error(ctx, MRL_ERR_INVALID, "Broyden: 1 <= nvar <= %d", kBroydenMax);
  for (int i = 0; i < nv; ++i) {
    if ((u && !u[i]) || (N && !N[i]) || (uold && !uold[i]) || (uout && !uout[i]))
      return set_error(ctx, MRL_ERR_INVALID, "Broyden: null buffer for variable %d", i);
    p.u[i] = u ? reinterpret_cast<const double2 *>(u[i]) : nullptr;
    p.N[i] = N ? reinterpret_cast<const double2 *>(N[i]) : nullptr;
    p.uold[i] = uold ? reinterpret_cast<const double2 *>(uold[i]) : nullptr;
    p.L[i] = L ? L[i] : nullptr;
    p.uout[i] = uout ? reinterpret_cast<double2 *>(uout[i]) : nullptr;
  }
  return MRL_OK;
}

}  // namespace mrl

extern "C" {

int mrl_broyden_init(mrl_ctx *ctx, int nvar, double factor, double *d_M, int64_t n_spec) {
  if (!ctx) return MRL_ERR_INVALID;
  if (nvar < 1 || nvar > kBroydenMax || !d_M || n_spec < 0) return set_error(ctx, MRL_ERR_INVALID, "mrl_broyden_init: bad argument");
  if (n_spec == 0) return MRL_OK;
  hipLaunchKernelGGL(k_broyden_init, dim3(blocks_for((long long)nvar * nvar * n_spec)), dim3(256), 0, ctx->stream, nvar, factor,
                     reinterpret_cast<double2 *>(d_M), (long long)n_spec);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

int mrl_broyden_residual(mrl_ctx *ctx, int nvar, const double *const *d_u, const double *const *d_N, const double *const *d_L,
                         const double *const *d_u_old, double sub_dt, double *d_R, double *h_sumsq, int64_t n_spec) {
  if (!ctx) return MRL_ERR_INVALID;
  if (!d_u || !d_N || !d_R || !h_sumsq || n_spec < 0) return set_error(ctx, MRL_ERR_INVALID, "mrl_broyden_residual: bad argument");
  BroydenPtrs p{};
  MRL_TRY(broyden_ptrs(ctx, nvar, d_u, d_N, d_L, d_u_old, nullptr, p));
  h_sumsq[0] = 0.0;
  if (n_spec == 0) return MRL_OK;
  const int nb = blocks_for(n_spec);
  double *slot = ctx->d_red + kScalarBase;
  {
    ProfScope ps(ctx, "broyden_residual");
    hipLaunchKernelGGL(k_broyden_residual, dim3(nb), dim3(256), 0, ctx->stream, nvar, p, d_u_old ? 1 : 0, sub_dt,
                       reinterpret_cast<double2 *>(d_R), (long long)n_spec, ctx->d_red);
    MRL_HIP(ctx, hipGetLastError());
  }
  MRL_TRY(reduce_finalize(ctx, nb, 1, slot));
  return read_scalars(ctx, slot, 1, h_sumsq);
}

int mrl_broyden_predict(mrl_ctx *ctx, int nvar, const double *d_M, const double *d_R, const double *const *d_u, double step,
                        double *d_S, double *const *d_u_out, int64_t n_spec) {
  if (!ctx) return MRL_ERR_INVALID;
  if (!d_M || !d_R || !d_u || !d_S || !d_u_out || n_spec < 0) return set_error(ctx, MRL_ERR_INVALID, "mrl_broyden_predict: bad argument");
  BroydenPtrs p{};
  MRL_TRY(broyden_ptrs(ctx, nvar, d_u, nullptr, nullptr, nullptr, d_u_out, p));
  if (n_spec == 0) return MRL_OK;
  ProfScope ps(ctx, "broyden_predict");
  hipLaunchKernelGGL(nvar <= 8 ? k_broyden_predict<8> : k_broyden_predict<kBroydenMax>, dim3(blocks_for(n_spec)), dim3(256), 0,
                     ctx->stream, nvar, p, reinterpret_cast<const double2 *>(d_M), reinterpret_cast<const double2 *>(d_R),
                     reinterpret_cast<double2 *>(d_S), step, (long long)n_spec);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

int mrl_broyden_update(mrl_ctx *ctx, int nvar, double *d_M, double *d_R, const double *d_S, const double *const *d_u,
                       const double *const *d_N, const double *const *d_L, const double *const *d_u_old, double sub_dt,
                       double *h_sumsq, int64_t n_spec) {
  if (!ctx) return MRL_ERR_INVALID;
  if (!d_M || !d_R || !d_S || !d_u || !d_N || !d_u_old || !h_sumsq || n_spec < 0)
    return set_error(ctx, MRL_ERR_INVALID, "mrl_broyden_update: bad argument");
  BroydenPtrs p{};
  MRL_TRY(broyden_ptrs(ctx, nvar, d_u, d_N, d_L, d_u_old, nullptr, p));
  h_sumsq[0] = 0.0;
  if (n_spec == 0) return MRL_OK;
  const int nb = blocks_for(n_spec);
  double *slot = ctx->d_red + kScalarBase;
  {
    ProfScope ps(ctx, "broyden_update");
    hipLaunchKernelGGL(nvar <= 8 ? k_broyden_update<8> : k_broyden_update<kBroydenMax>, dim3(nb), dim3(256), 0, ctx->stream, nvar, p,
                       reinterpret_cast<double2 *>(d_M), reinterpret_cast<double2 *>(d_R), reinterpret_cast<const double2 *>(d_S), sub_dt,
                       (long long)n_spec, ctx->d_red);
    MRL_HIP(ctx, hipGetLastError());
  }
  MRL_TRY(reduce_finalize(ctx, nb, 1, slot));
  return read_scalars(ctx, slot, 1, h_sumsq);
}

}  // extern "C"
