// Cahn-Hilliard pointwise operators and the substep driver.
//   mu      : ParsedCompute f'(c)            (src/tensor_computes/ParsedCompute.C:184-265; op tree SURVEY A.3)
//   k-space : Mbar*mubar + ABM predictor     (src/tensor_solver/AdamsBashforthMoulton.C:94-99)
// k, Mbar = -k^2 M and Lbar = k^2 k^2 f are recomputed from the 1-D reciprocal axes (no full arrays).
#include "mrl_internal.h"
#include <cstring>
#include <vector>

namespace mrl {

__device__ __forceinline__ double ch_mu_eval(const ChP &p, double c) {
#pragma clang fp contract(off)
  if (p.family == MRL_FE_DOUBLE_WELL) {
    // (A*(2*c)) * pow(c-1,2) + (A*pow(c,2)) * (2*(c-1))
    const double cm1 = c - 1.0;
    return (p.c0 * (2.0 * c)) * (cm1 * cm1) + (p.c0 * (c * c)) * (2.0 * cm1);
  } else {
    // (rho*(2*(c-ca))) * pow(cb-c,2) + (rho*pow(c-ca,2)) * ((2*(cb-c))*-1)
    const double a = c - p.c1;
    const double b = p.c2 - c;
    return (p.c0 * (2.0 * a)) * (b * b) + (p.c0 * (a * a)) * ((2.0 * b) * -1.0);
  }
}

__global__ void __launch_bounds__(256) k_ch_mu(ChP p, const double *__restrict__ c, double *__restrict__ mu, long long n) {
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  const long long stride = (long long)gridDim.x * blockDim.x * 2;
  for (; i + 1 < n; i += stride) {
    const double2 v = *reinterpret_cast<const double2 *>(c + i);
    double2 r;
    r.x = ch_mu_eval(p, v.x);
    r.y = ch_mu_eval(p, v.y);
    *reinterpret_cast<double2 *>(mu + i) = r;
  }
  if (i < n) mu[i] = ch_mu_eval(p, c[i]);
}

struct KspaceArgs {
  int dim;
  long long kz0, ksub;   // kz sub-range handled by this launch (kz0 = 0, ksub = n2: everything)
  long long n0, n1, n2;  // local reciprocal extents
  const double *k0, *k1, *k2;
  double M, kappa, dt;
  int order;
  double coef[6];        // dt*beta[order][i]
  const double *Nold[5];
};

// kx*kx + ky*ky + kz*kz with unused axes = {0} (DomainAction.C:1503-1509): adding the exact zeros of
// the unused axes changes nothing, so one expression serves every dimension and axis alignment.
__device__ __forceinline__ double ksq(int, double a, double b, double c) {
#pragma clang fp contract(off)
  return a * a + b * b + c * c;
}

// one thread per spectral point; cbar / mubar interleaved complex
__global__ void __launch_bounds__(256) k_ch_kspace(KspaceArgs a, const double2 *cbar /* may alias ubar */,
                                                   const double2 *__restrict__ mubar, double2 *__restrict__ Nhat,
                                                   double2 *ubar) {
#pragma clang fp contract(off)
  const long long total = a.n0 * a.n1 * a.ksub;
  for (long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += (long long)gridDim.x * blockDim.x) {
    const long long i2 = a.kz0 + s % a.ksub;
    const long long t = s / a.ksub;
    const long long i1 = t % a.n1;
    const long long i0 = t / a.n1;
    const long long e = (i0 * a.n1 + i1) * a.n2 + i2;
    const double k2v = ksq(a.dim, a.k0[i0], a.k1[i1], a.k2[i2]);
    const double Mbar = -k2v * a.M;
    const double L = k2v * k2v * a.kappa;
    const double2 m = mubar[e];
    double2 N = make_double2(Mbar * m.x, Mbar * m.y);
    Nhat[e] = N;
    double2 u = cbar[e];
    u.x = u.x + a.coef[0] * N.x;
    u.y = u.y + a.coef[0] * N.y;
    for (int i = 0; i < a.order; ++i) {
      const double2 o = reinterpret_cast<const double2 *>(a.Nold[i])[e];
      u.x += a.coef[i + 1] * o.x;
      u.y += a.coef[i + 1] * o.y;
    }
    const double den = 1.0 - a.dt * L;
    const double scl = 1.0 / den;
    u.x = u.x * scl;
    u.y = u.y * scl;
    ubar[e] = u;
  }
}

// ReciprocalLaplacianFactor (-k^2 * f, ReciprocalLaplacianFactor.C:28-31) / ReciprocalLaplacianSquareFactor
// (k^2 * k^2 * f, ReciprocalLaplacianSquareFactor.C:28-32) as full reciprocal-grid arrays, for solvers that take
// their linear operator as a buffer
__global__ void __launch_bounds__(256) k_recip_factor(int power, double factor, long long n0, long long n1, long long n2,
                                                       const double *__restrict__ k0, const double *__restrict__ k1,
                                                       const double *__restrict__ k2, double *__restrict__ out) {
#pragma clang fp contract(off)
  const long long total = n0 * n1 * n2;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long i2 = e % n2, t = e / n2, i1 = t % n1, i0 = t / n1;
    const double q = ksq(3, k0[i0], k1[i1], k2[i2]);
    out[e] = power == 1 ? -q * factor : q * q * factor;
  }
}

struct AbmArgs {
  int nterms;
  double coef[8];
  const double *N[8];
  double dt;
};

__global__ void __launch_bounds__(256) k_kspace_abm(AbmArgs a, double2 *__restrict__ out, const double2 *__restrict__ u0,
                                                    const double *__restrict__ L, long long n) {
#pragma clang fp contract(off)
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    double2 u = u0[e];
    for (int i = 0; i < a.nterms; ++i) {
      const double2 o = reinterpret_cast<const double2 *>(a.N[i])[e];
      u.x += a.coef[i] * o.x;
      u.y += a.coef[i] * o.y;
    }
    if (L) {
      const double scl = 1.0 / (1.0 - a.dt * L[e]);
      u.x *= scl;
      u.y *= scl;
    }
    out[e] = u;
  }
}

// AdamsBashforthMoultonCoupled (AdamsBashforthMoultonCoupled.C:141-186): per reciprocal grid point, rhs_i = ubar0_i + sum_t coef_it N_it
// and the dense nvar x nvar solve of (I - dt*L) ubar = rhs by LU with partial pivoting (what at::linalg_solve's gesv does).
// The reference stacks the columns of each row on the last axis and then the rows on a new last axis (:160-176), so the
// matrix it hands to the solver is A[a][b] = delta_ab - dt * L_ba (the transpose of the operator table of the input file),
// and it casts the complex right-hand side to the real dtype of L (:183), i.e. only Re(rhs) enters the solve; both are
// reproduced unless the corresponding flag asks for the operator as written / the full complex solve.
constexpr int kCoupledMax = 8;  // variables of one coupled solve (the reference: any N; none of its inputs has more than 3)
struct CoupledArgs {
  int nterms[kCoupledMax];
  double coef[kCoupledMax][6];
  const double *N[kCoupledMax][6];
  const double *u0[kCoupledMax];
  const double *L[kCoupledMax * kCoupledMax];
  double *out[kCoupledMax];
  double dt;
  int flags;
};

template <int NV>
__global__ void __launch_bounds__(256) k_kspace_coupled(CoupledArgs a, long long n) {
#pragma clang fp contract(off)
  const bool real_rhs = !(a.flags & MRL_COUPLED_COMPLEX_RHS), transposed = !(a.flags & MRL_COUPLED_L_AS_WRITTEN);
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    double A[NV][NV], br[NV], bi[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double2 u = reinterpret_cast<const double2 *>(a.u0[i])[e];
      for (int t = 0; t < a.nterms[i]; ++t) {
        const double2 o = reinterpret_cast<const double2 *>(a.N[i][t])[e];
        u.x += a.coef[i][t] * o.x;
        u.y += a.coef[i][t] * o.y;
      }
      br[i] = u.x;
      bi[i] = real_rhs ? 0.0 : u.y;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const double *Lp = transposed ? a.L[j * NV + i] : a.L[i * NV + j];
        const double l = Lp ? Lp[e] : 0.0;
        A[i][j] = (i == j ? 1.0 : 0.0) - a.dt * l;
      }
    }
    // LU with partial pivoting (row swaps done with selects so that A stays in registers)
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      int piv = c;
      double best = fabs(A[c][c]);
#pragma unroll
      for (int r = c + 1; r < NV; ++r)
        if (fabs(A[r][c]) > best) {
          best = fabs(A[r][c]);
          piv = r;
        }
#pragma unroll
      for (int r = c + 1; r < NV; ++r)
        if (piv == r) {
#pragma unroll
          for (int j = 0; j < NV; ++j) {
            const double t = A[c][j];
            A[c][j] = A[r][j];
            A[r][j] = t;
          }
          double t = br[c];
          br[c] = br[r];
          br[r] = t;
          t = bi[c];
          bi[c] = bi[r];
          bi[r] = t;
        }
      const double inv = 1.0 / A[c][c];
#pragma unroll
      for (int r = c + 1; r < NV; ++r) {
        const double f = A[r][c] * inv;
#pragma unroll
        for (int j = c + 1; j < NV; ++j) A[r][j] -= f * A[c][j];
        br[r] -= f * br[c];
        bi[r] -= f * bi[c];
      }
    }
#pragma unroll
    for (int c = NV - 1; c >= 0; --c) {
#pragma unroll
      for (int j = c + 1; j < NV; ++j) {
        br[c] -= A[c][j] * br[j];
        bi[c] -= A[c][j] * bi[j];
      }
      br[c] /= A[c][c];
      bi[c] /= A[c][c];
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) reinterpret_cast<double2 *>(a.out[i])[e] = make_double2(br[i], bi[i]);
  }
}

// The same solve for more variables than a register file holds (9 ... kCoupledAnyMax; the reference: any N): the matrix and the
// right-hand side of a thread live in a workspace [(nv * nv + 2 nv)][lanes] doubles -- entry-major, so that the lanes of a wave touch
// consecutive addresses -- and the pointer tables in device memory.  Same operations in the same order as k_kspace_coupled (row swaps
// are real swaps here): equal results bit for bit where both apply (tests/test_coupled_gpu.py).
constexpr int kCoupledAnyMax = 32;
struct CoupledAnyArgs {
  int nv, flags;
  double dt;
  const int *nterms;        // [nv]
  const double *coef;       // [nv][6]
  const double *const *N;   // [nv][6]
  const double *const *u0;  // [nv]
  const double *const *L;   // [nv][nv] as the caller names them (row = equation, column = variable)
  double *const *out;       // [nv]
  double *ws;
  long long lanes;
};

__global__ void __launch_bounds__(256) k_kspace_coupled_any(CoupledAnyArgs a, long long n) {
#pragma clang fp contract(off)
  const int nv = a.nv;
  const bool real_rhs = !(a.flags & MRL_COUPLED_COMPLEX_RHS), transposed = !(a.flags & MRL_COUPLED_L_AS_WRITTEN);
  const long long lane = (long long)blockIdx.x * blockDim.x + threadIdx.x, W = a.lanes;
  double *A = a.ws + lane;                       // A(i, j) = A[(i * nv + j) * W]
  double *br = A + (long long)nv * nv * W, *bi = br + (long long)nv * W;
  for (long long e = lane; e < n; e += W) {
    for (int i = 0; i < nv; ++i) {
      double2 u = reinterpret_cast<const double2 *>(a.u0[i])[e];
      for (int t = 0; t < a.nterms[i]; ++t) {
        const double2 o = reinterpret_cast<const double2 *>(a.N[i * 6 + t])[e];
        const double c = a.coef[i * 6 + t];
        u.x += c * o.x;
        u.y += c * o.y;
      }
      br[i * W] = u.x;
      bi[i * W] = real_rhs ? 0.0 : u.y;
      for (int j = 0; j < nv; ++j) {
        const double *Lp = transposed ? a.L[j * nv + i] : a.L[i * nv + j];
        const double l = Lp ? Lp[e] : 0.0;
        A[(long long)(i * nv + j) * W] = (i == j ? 1.0 : 0.0) - a.dt * l;
      }
    }
    for (int c = 0; c < nv; ++c) {
      int piv = c;
      double best = fabs(A[(long long)(c * nv + c) * W]);
      for (int r = c + 1; r < nv; ++r) {
        const double v = fabs(A[(long long)(r * nv + c) * W]);
        if (v > best) {
          best = v;
          piv = r;
        }
      }
      if (piv != c) {
        for (int j = 0; j < nv; ++j) {
          const double t = A[(long long)(c * nv + j) * W];
          A[(long long)(c * nv + j) * W] = A[(long long)(piv * nv + j) * W];
          A[(long long)(piv * nv + j) * W] = t;
        }
        double t = br[c * W];
        br[c * W] = br[piv * W];
        br[piv * W] = t;
        t = bi[c * W];
        bi[c * W] = bi[piv * W];
        bi[piv * W] = t;
      }
      const double inv = 1.0 / A[(long long)(c * nv + c) * W];
      const double brc = br[c * W], bic = bi[c * W];
      for (int r = c + 1; r < nv; ++r) {
        const double f = A[(long long)(r * nv + c) * W] * inv;
        for (int j = c + 1; j < nv; ++j) A[(long long)(r * nv + j) * W] -= f * A[(long long)(c * nv + j) * W];
        br[r * W] -= f * brc;
        bi[r * W] -= f * bic;
      }
    }
    for (int c = nv - 1; c >= 0; --c) {
      double xr = br[c * W], xi = bi[c * W];
      for (int j = c + 1; j < nv; ++j) {
        const double m = A[(long long)(c * nv + j) * W];
        xr -= m * br[j * W];
        xi -= m * bi[j * W];
      }
      const double d = A[(long long)(c * nv + c) * W];
      xr /= d;
      xi /= d;
      br[c * W] = xr;
      bi[c * W] = xi;
    }
    for (int i = 0; i < nv; ++i) reinterpret_cast<double2 *>(a.out[i])[e] = make_double2(br[i * W], bi[i * W]);
  }
}

// Adams-Bashforth coefficients (src/tensor_solver/AdamsBashforthMoulton.C:67-73, incl. the AB5 190/720 entry)
static const double kBeta[5][5] = {
    {1.0, 0.0, 0.0, 0.0, 0.0},
    {3.0 / 2.0, -1.0 / 2.0, 0.0, 0.0, 0.0},
    {23.0 / 12.0, -16.0 / 12.0, 5.0 / 12.0, 0.0, 0.0},
    {55.0 / 24.0, -59.0 / 24.0, 37.0 / 24.0, -9.0 / 24.0, 0.0},
    {190.0 / 720.0, -2774.0 / 720.0, 2616.0 / 720.0, -1274.0 / 720.0, 251.0 / 720.0},
};

int ch_check_params(mrl_ctx *ctx, const mrl_ch_params *p, ChP &cp) {
  if (!p) return set_error(ctx, MRL_ERR_INVALID, "null mrl_ch_params");
  if (p->family != MRL_FE_DOUBLE_WELL && p->family != MRL_FE_PFHUB && p->family != MRL_FE_PARSED)
    return set_error(ctx, MRL_ERR_INVALID, "unknown free energy family %d", p->family);
  cp.parsed = nullptr;
  if (p->family == MRL_FE_PARSED) {
    MRL_TRY(parsed_check_mu(ctx, p->parsed));
    cp.parsed = p->parsed;
  }
  cp.family = p->family;
  cp.c0 = p->coef[0];
  cp.c1 = p->coef[1];
  cp.c2 = p->coef[2];
  cp.M = p->mobility;
  cp.kappa = p->kappa;
  return MRL_OK;
}

static inline int grid_for(long long n, int per_thread = 1) {
  long long b = (n / per_thread + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

int ch_mu_launch(mrl_ctx *ctx, const ChP &cp, const double *c, double *mu, long long count) {
  if (count == 0) return MRL_OK;
  if (cp.family == MRL_FE_PARSED) return parsed_eval1(cp.parsed, c, mu, count);
  if ((reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(mu)) & 15)
    return set_error(ctx, MRL_ERR_INVALID, "mrl_ch_mu: pointers must be 16-byte aligned");
  ProfScope ps(ctx, "ch_mu", 16.0 * (double)count);
  hipLaunchKernelGGL(k_ch_mu, dim3(grid_for(count, 2)), dim3(256), 0, ctx->stream, cp, c, mu, count);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

int ch_kspace_sub_launch(mrl_ctx *ctx, const ChP &cp, const double *cbar, const double *mubar, double *Nhat, double *ubar,
                         const double *const *Nold, int order, double sub_dt, long long k0, long long ksub) {
  KspaceArgs a{};
  a.dim = ctx->dim;
  a.kz0 = k0;
  a.ksub = ksub;
  a.n0 = ctx->nrec[0];
  a.n1 = ctx->nrec[1];
  a.n2 = ctx->nrec[2];
  a.k0 = ctx->d_k[0];
  a.k1 = ctx->d_k[1];
  a.k2 = ctx->d_k[2];
  a.M = cp.M;
  a.kappa = cp.kappa;
  a.dt = sub_dt;
  a.order = order;
  for (int i = 0; i <= order; ++i) a.coef[i] = sub_dt * kBeta[order][i];
  for (int i = 0; i < order; ++i) a.Nold[i] = Nold[i];
  const long long total = a.n0 * a.n1 * a.ksub;
  hipLaunchKernelGGL(k_ch_kspace, dim3(grid_for(total)), dim3(256), 0, ctx->stream, a,
                     reinterpret_cast<const double2 *>(cbar), reinterpret_cast<const double2 *>(mubar),
                     reinterpret_cast<double2 *>(Nhat), reinterpret_cast<double2 *>(ubar));
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

int ch_kspace_launch(mrl_ctx *ctx, const ChP &cp, const double *cbar, const double *mubar, double *Nhat, double *ubar,
                     const double *const *Nold, int order, double sub_dt) {
  return ch_kspace_sub_launch(ctx, cp, cbar, mubar, Nhat, ubar, Nold, order, sub_dt, 0, ctx->nrec[2]);
}

int ch_substeps_fused(mrl_ctx *ctx, const ChP &cp, const double *c_in, double *c_out, double *const *ring, int ring_size,
                      int *head, int *n_old, int pred, int count, int advance, double sub_dt, double *mu, bool dt_changed);
// fast fused path (ch_fused.hip); returns MRL_ERR_UNSUPPORTED when the shape has no fast kernels
int ch_substep_fused(mrl_ctx *ctx, const ChP &cp, const double *c_in, double *c_out, double *Nhat_new,
                     const double *const *Nhat_old, int order, double sub_dt, double *cbar, double *mu, int carry);
// extents with plans for the plain transforms only (ch_planned.hip)
int ch_substep_planned(mrl_ctx *ctx, const ChP &cp, const double *c_in, double *c_out, double *Nhat_new, const double *const *Nhat_old,
                       int order, double sub_dt, double *cbar, double *mu, int carry);
// slab contexts with a communicator (slab_driver.hip)
int ch_substeps_planned(mrl_ctx *ctx, const ChP &cp, const double *c_in, double *c_out, double *const *ring, int ring_size, int *head,
                        int *n_old, int pred, int count, int advance, double sub_dt, double *mu, bool dt_changed);
int slab_ch_substeps(mrl_ctx *ctx, const mrl_ch_params *p, const double *c_in, double *c_out, double *const *ring, int ring_size,
                     int *head, int *n_old, int pred, int count, int advance, double sub_dt, double *mu, bool dt_changed);

}  // namespace mrl

using namespace mrl;

extern "C" {

int mrl_ch_mu(mrl_ctx *ctx, const mrl_ch_params *p, const double *d_c, double *d_mu, int64_t count) {
  if (!ctx) return MRL_ERR_INVALID;
  ChP cp;
  MRL_TRY(ch_check_params(ctx, p, cp));
  if (!d_c || !d_mu || count < 0) return set_error(ctx, MRL_ERR_INVALID, "mrl_ch_mu: bad argument");
  return ch_mu_launch(ctx, cp, d_c, d_mu, (long long)count);
}

int mrl_ch_substep(mrl_ctx *ctx, const mrl_ch_params *p, const double *d_c_in, double *d_c_out, double *d_Nhat_new,
                   const double *const *d_Nhat_old, int order, double sub_dt, double *d_cbar, double *d_mu, int carry) {
  if (!ctx) return MRL_ERR_INVALID;
  ChP cp;
  MRL_TRY(ch_check_params(ctx, p, cp));
  if (!d_c_in || !d_c_out || !d_Nhat_new) return set_error(ctx, MRL_ERR_INVALID, "mrl_ch_substep: null buffer");
  if (order < 0 || order > 4) return set_error(ctx, MRL_ERR_INVALID, "predictor order %d out of range", order + 1);
  if (order > 0 && !d_Nhat_old) return set_error(ctx, MRL_ERR_INVALID, "history pointers missing");
  for (int i = 0; i < order; ++i)
    if (!d_Nhat_old[i]) return set_error(ctx, MRL_ERR_INVALID, "history entry %d missing", i);
  if (ctx->slab) {
    // one substep of the library-owned slab pipeline: the explicit history pointers as a ring {old[order-1] ... old[0], new}
    if (carry != MRL_CARRY_NONE || d_cbar)
      return set_error(ctx, MRL_ERR_UNSUPPORTED, "mrl_ch_substep on a slab context: cbar / carry are handled inside mrl_ch_substeps (MRL_OPT_SLAB_CARRY)");
    if (order == 0) {
      double *ring0[1] = {d_Nhat_new};
      int head = 0, n_old = 0;
      return slab_ch_substeps(ctx, p, d_c_in, d_c_out, ring0, 1, &head, &n_old, 0, 1, 0, sub_dt, d_mu, false);
    }
    double *ring[5];
    for (int i = 0; i < order; ++i) ring[order - 1 - i] = const_cast<double *>(d_Nhat_old[i]);
    ring[order] = d_Nhat_new;
    int head = order - 1, n_old = order;
    return slab_ch_substeps(ctx, p, d_c_in, d_c_out, ring, order + 1, &head, &n_old, order, 1, 0, sub_dt, d_mu, false);
  }

  if (carry != MRL_CARRY_NONE && carry != MRL_CARRY_OUT && carry != MRL_CARRY_IN)
    return set_error(ctx, MRL_ERR_INVALID, "mrl_ch_substep: carry must be MRL_CARRY_NONE, _OUT or _IN");
  if (carry != MRL_CARRY_NONE && !d_cbar)
    return set_error(ctx, MRL_ERR_INVALID, "mrl_ch_substep: the carry-over needs the d_cbar array");

  int rc = ch_substep_fused(ctx, cp, d_c_in, d_c_out, d_Nhat_new, d_Nhat_old, order, sub_dt, d_cbar, d_mu, carry);
  if (rc != MRL_ERR_UNSUPPORTED) return rc;
  rc = ch_substep_planned(ctx, cp, d_c_in, d_c_out, d_Nhat_new, d_Nhat_old, order, sub_dt, d_cbar, d_mu, carry);
  if (rc != MRL_ERR_UNSUPPORTED) return rc;

  // generic sequence: separate pointwise kernels around the generic transforms
  const long long nreal = real_count_local(ctx), nspec = spec_count_local(ctx);
  MRL_TRY(ensure_work(ctx, 1, sizeof(cplx) * nspec));
  MRL_TRY(ensure_work(ctx, 2, sizeof(cplx) * nspec));
  double *mu = d_mu;
  if (!mu) {
    MRL_TRY(ensure_work(ctx, 3, sizeof(double) * (nreal + 2)));
    mu = ctx->d_work[3];
  }
  MRL_TRY(mrl_ch_mu(ctx, p, d_c_in, mu, nreal));
  double *mubar = ctx->d_work[1];
  // carry-over: MRL_CARRY_IN reads c-hat from d_cbar (no transform of c) and both carry modes leave ubar there
  double *cbar = (d_cbar && carry != MRL_CARRY_OUT) ? d_cbar : ctx->d_work[2];
  double *ubar = carry == MRL_CARRY_NONE ? ctx->d_work[2] : d_cbar;
  // pencil contexts (parallel_mode = FFT_PENCIL): the same operator sequence over the staged pencil transforms -- in the reference
  // every solver reaches the decomposition only through DomainAction::fft / ifft (AdamsBashforthMoulton.C:88-101); the k-space
  // kernel works on this rank's reciprocal block with its own slices of the reciprocal axes (ctx->nrec, ctx->d_k)
  auto forward = [ctx](const double *in, double *out) {
    return ctx->pencil ? pencil_fft_forward(ctx, in, out, 1) : fft_forward_serial(ctx, in, out, 1, 0);
  };
  auto inverse = [ctx](const double *in, double *out) {
    return ctx->pencil ? pencil_fft_inverse(ctx, in, out, 1) : fft_inverse_serial(ctx, in, out, 1, 0);
  };
  MRL_TRY(forward(mu, mubar));
  if (carry != MRL_CARRY_IN) MRL_TRY(forward(d_c_in, cbar));
  {
    ProfScope ps(ctx, "ch_kspace", 16.0 * (double)nspec * (4 + order));
    MRL_TRY(ch_kspace_launch(ctx, cp, cbar, mubar, d_Nhat_new, ubar, d_Nhat_old, order, sub_dt));  // elementwise: ubar may alias cbar
  }
  if (carry == MRL_CARRY_NONE) return inverse(ubar, d_c_out);
  // the inverse transform may overwrite its input: run it on a copy so that d_cbar keeps ubar
  MRL_HIP(ctx, hipMemcpyAsync(mubar, ubar, sizeof(cplx) * nspec, hipMemcpyDeviceToDevice, ctx->stream));
  return inverse(mubar, d_c_out);
}

int mrl_reciprocal_laplacian(mrl_ctx *ctx, int power, double factor, double *d_out) {
  if (!ctx) return MRL_ERR_INVALID;
  if (!d_out || (power != 1 && power != 2))
    return set_error(ctx, MRL_ERR_INVALID, "mrl_reciprocal_laplacian: power must be 1 (-k^2 f) or 2 (k^4 f)");
  const long long total = spec_count_local(ctx);
  ProfScope ps(ctx, "recip_factor", 8.0 * total);
  hipLaunchKernelGGL(k_recip_factor, dim3(grid_for(total)), dim3(256), 0, ctx->stream, power, factor, ctx->nrec[0],
                     ctx->nrec[1], ctx->nrec[2], ctx->d_k[0], ctx->d_k[1], ctx->d_k[2], d_out);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

int mrl_kspace_abm(mrl_ctx *ctx, double *d_ubar_out, const double *d_ubar0, const double *const *d_N,
                   const double *h_coef, int nterms, const double *d_L, double dt, int64_t n_spec) {
  if (!ctx) return MRL_ERR_INVALID;
  if (!d_ubar_out || !d_ubar0 || nterms < 0 || nterms > 8 || n_spec < 0 || (nterms > 0 && (!d_N || !h_coef)))
    return set_error(ctx, MRL_ERR_INVALID, "mrl_kspace_abm: bad argument");
  if (n_spec == 0) return MRL_OK;
  AbmArgs a{};
  a.nterms = nterms;
  a.dt = dt;
  for (int i = 0; i < nterms; ++i) {
    a.coef[i] = h_coef[i];
    a.N[i] = d_N[i];
  }
  ProfScope ps(ctx, "kspace_abm");
  hipLaunchKernelGGL(k_kspace_abm, dim3(grid_for(n_spec)), dim3(256), 0, ctx->stream, a,
                     reinterpret_cast<double2 *>(d_ubar_out), reinterpret_cast<const double2 *>(d_ubar0), d_L,
                     (long long)n_spec);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

int mrl_kspace_coupled(mrl_ctx *ctx, int nvar, double *const *d_ubar_out, const double *const *d_ubar0,
                       const double *const *d_N, const double *h_coef, const int *h_nterms, const double *const *d_L,
                       double dt, int flags, int64_t n_spec) {
  if (!ctx) return MRL_ERR_INVALID;
  if (nvar < 1 || nvar > kCoupledAnyMax || !d_ubar_out || !d_ubar0 || !d_L || !h_nterms || n_spec < 0)
    return set_error(ctx, MRL_ERR_INVALID, "mrl_kspace_coupled: bad argument (1 <= nvar <= %d)", kCoupledAnyMax);
  if (n_spec == 0) return MRL_OK;
  if (nvar > kCoupledMax || (flags & MRL_COUPLED_GENERAL)) {
    // pointer tables and the per-thread matrices in device memory (slots 16 / 17 belong to this entry point)
    std::vector<unsigned char> tab;
    auto put = [&tab](const void *src, size_t bytes) {
      const size_t at = (tab.size() + 15) & ~size_t(15);
      tab.resize(at + bytes);
      std::memcpy(tab.data() + at, src, bytes);
      return at;
    };
    std::vector<int> nt(nvar);
    std::vector<double> coef((size_t)nvar * 6, 0.0);
    std::vector<const double *> Np((size_t)nvar * 6, nullptr), u0(nvar), Lp((size_t)nvar * nvar);
    std::vector<double *> outp(nvar);
    int at = 0;
    for (int i = 0; i < nvar; ++i) {
      if (!d_ubar_out[i] || !d_ubar0[i] || h_nterms[i] < 0 || h_nterms[i] > 6 || (h_nterms[i] > 0 && (!d_N || !h_coef)))
        return set_error(ctx, MRL_ERR_INVALID, "mrl_kspace_coupled: bad variable %d (at most 6 terms each)", i);
      nt[i] = h_nterms[i];
      u0[i] = d_ubar0[i];
      outp[i] = d_ubar_out[i];
      for (int t = 0; t < h_nterms[i]; ++t, ++at) {
        coef[(size_t)i * 6 + t] = h_coef[at];
        Np[(size_t)i * 6 + t] = d_N[at];
      }
      for (int j = 0; j < nvar; ++j) Lp[(size_t)i * nvar + j] = d_L[i * nvar + j];
    }
    const size_t o_nt = put(nt.data(), sizeof(int) * nvar), o_cf = put(coef.data(), sizeof(double) * coef.size()),
                 o_N = put(Np.data(), sizeof(void *) * Np.size()), o_u = put(u0.data(), sizeof(void *) * nvar),
                 o_L = put(Lp.data(), sizeof(void *) * Lp.size()), o_o = put(outp.data(), sizeof(void *) * nvar);
    // workgroups: as many as the grid needs, at most 256, and no more than a 256 MB workspace holds (32 variables: 108 instead of 256;
    // the kernel walks the grid with a stride of `lanes`)
    const size_t per_lane = sizeof(double) * (size_t)(nvar * nvar + 2 * nvar);
    long long nb = (n_spec + 255) / 256;
    if (nb > 256) nb = 256;
    const long long cap_nb = (long long)((256u << 20) / (per_lane * 256));
    if (nb > cap_nb) nb = cap_nb < 1 ? 1 : cap_nb;
    const long long lanes = nb * 256;
    MRL_TRY(ensure_work(ctx, 16, per_lane * (size_t)lanes));
    // the tables travel through a pinned staging slot; the slot is reused only after the launch that read its device copy has
    // finished (an event per slot: with four slots the wait is on a launch four calls back) -- no synchronisation per call
    auto &slot = ctx->tab_ring[ctx->tab_next];
    ctx->tab_next = (ctx->tab_next + 1) % 4;
    if (slot.done) MRL_HIP(ctx, hipEventSynchronize(slot.done));
    if (slot.cap < tab.size()) {
      if (slot.h) (void)hipHostFree(slot.h);
      if (slot.d) (void)hipFree(slot.d);
      slot.h = slot.d = nullptr;
      slot.cap = 0;
      const size_t cap = (tab.size() + 4095) & ~size_t(4095);
      MRL_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&slot.h), cap));
      MRL_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&slot.d), cap));
      slot.cap = cap;
    }
    if (!slot.done) MRL_HIP(ctx, hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
    std::memcpy(slot.h, tab.data(), tab.size());
    unsigned char *dt_ = slot.d;
    MRL_HIP(ctx, hipMemcpyAsync(dt_, slot.h, tab.size(), hipMemcpyHostToDevice, ctx->stream));
    CoupledAnyArgs g{};
    g.nv = nvar;
    g.flags = flags;
    g.dt = dt;
    g.nterms = reinterpret_cast<const int *>(dt_ + o_nt);
    g.coef = reinterpret_cast<const double *>(dt_ + o_cf);
    g.N = reinterpret_cast<const double *const *>(dt_ + o_N);
    g.u0 = reinterpret_cast<const double *const *>(dt_ + o_u);
    g.L = reinterpret_cast<const double *const *>(dt_ + o_L);
    g.out = reinterpret_cast<double *const *>(dt_ + o_o);
    g.ws = ctx->d_work[16];
    g.lanes = lanes;
    ProfScope ps(ctx, "kspace_coupled_general");
    hipLaunchKernelGGL(k_kspace_coupled_any, dim3((unsigned)nb), dim3(256), 0, ctx->stream, g, (long long)n_spec);
    MRL_HIP(ctx, hipGetLastError());
    MRL_HIP(ctx, hipEventRecord(slot.done, ctx->stream));
    return MRL_OK;
  }
  CoupledArgs a{};
  a.dt = dt;
  a.flags = flags;
  int at = 0;
  for (int i = 0; i < nvar; ++i) {
    if (!d_ubar_out[i] || !d_ubar0[i] || h_nterms[i] < 0 || h_nterms[i] > 6 || (h_nterms[i] > 0 && (!d_N || !h_coef)))
      return set_error(ctx, MRL_ERR_INVALID, "mrl_kspace_coupled: bad variable %d (at most 6 terms each)", i);
    a.nterms[i] = h_nterms[i];
    a.u0[i] = d_ubar0[i];
    a.out[i] = d_ubar_out[i];
    for (int t = 0; t < h_nterms[i]; ++t, ++at) {
      a.coef[i][t] = h_coef[at];
      a.N[i][t] = d_N[at];
    }
    for (int j = 0; j < nvar; ++j) a.L[i * nvar + j] = d_L[i * nvar + j];
  }
  ProfScope ps(ctx, "kspace_coupled");
  const dim3 grid(grid_for(n_spec)), block(256);
  switch (nvar) {
    case 1: hipLaunchKernelGGL(k_kspace_coupled<1>, grid, block, 0, ctx->stream, a, (long long)n_spec); break;
    case 2: hipLaunchKernelGGL(k_kspace_coupled<2>, grid, block, 0, ctx->stream, a, (long long)n_spec); break;
    case 3: hipLaunchKernelGGL(k_kspace_coupled<3>, grid, block, 0, ctx->stream, a, (long long)n_spec); break;
    case 4: hipLaunchKernelGGL(k_kspace_coupled<4>, grid, block, 0, ctx->stream, a, (long long)n_spec); break;
    case 5: hipLaunchKernelGGL(k_kspace_coupled<5>, grid, block, 0, ctx->stream, a, (long long)n_spec); break;
    case 6: hipLaunchKernelGGL(k_kspace_coupled<6>, grid, block, 0, ctx->stream, a, (long long)n_spec); break;
    case 7: hipLaunchKernelGGL(k_kspace_coupled<7>, grid, block, 0, ctx->stream, a, (long long)n_spec); break;
    default: hipLaunchKernelGGL(k_kspace_coupled<8>, grid, block, 0, ctx->stream, a, (long long)n_spec); break;
  }
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

int mrl_ch_substeps(mrl_ctx *ctx, const mrl_ch_params *p, const double *d_c_in, double *d_c_out, double *const *d_Nhat_ring,
                    int ring_size, int *head, int *n_old, int predictor_order, int count, int advance, double sub_dt,
                    double *d_mu) {
  if (!ctx) return MRL_ERR_INVALID;
  ChP cp;
  MRL_TRY(ch_check_params(ctx, p, cp));
  const int pred = predictor_order - 1;
  const bool dt_changed = (advance & MRL_SUBSTEPS_DT_CHANGED) != 0;
  advance &= MRL_SUBSTEPS_ADVANCE;
  if (!d_c_in || !d_c_out || !d_Nhat_ring || !head || !n_old || count < 1)
    return set_error(ctx, MRL_ERR_INVALID, "mrl_ch_substeps: bad argument");
  if (predictor_order < 1 || predictor_order > 5) return set_error(ctx, MRL_ERR_INVALID, "predictor order %d out of range", predictor_order);
  if (ring_size < pred + 1 || *head < 0 || *head >= ring_size || *n_old < 0 || *n_old > pred)
    return set_error(ctx, MRL_ERR_INVALID, "mrl_ch_substeps: history ring of %d arrays needed (head %d, n_old %d)", pred + 1, *head, *n_old);
  for (int i = 0; i < ring_size; ++i)
    if (!d_Nhat_ring[i]) return set_error(ctx, MRL_ERR_INVALID, "history ring entry %d missing", i);
  if (ctx->slab)  // the library owns the exchanges (communicator attached with mrl_ctx_attach_comm)
    return slab_ch_substeps(ctx, p, d_c_in, d_c_out, d_Nhat_ring, ring_size, head, n_old, pred, count, advance, sub_dt, d_mu, dt_changed);
  int rc = ch_substeps_fused(ctx, cp, d_c_in, d_c_out, d_Nhat_ring, ring_size, head, n_old, pred, count, advance, sub_dt, d_mu, dt_changed);
  if (rc != MRL_ERR_UNSUPPORTED) return rc;
  rc = ch_substeps_planned(ctx, cp, d_c_in, d_c_out, d_Nhat_ring, ring_size, head, n_old, pred, count, advance, sub_dt, d_mu, dt_changed);
  if (rc != MRL_ERR_UNSUPPORTED) return rc;
  // generic shapes: one mrl_ch_substep per substep, the intermediate fields ping-pong between d_c_out and a scratch array
  const long long nreal = real_count_local(ctx);
  MRL_TRY(ensure_work(ctx, 15, sizeof(double) * (size_t)(nreal + 2)));   // (slot 15: a slab slot, free on serial contexts)
  double *tmp = ctx->d_work[15];
  const double *src = d_c_in;
  for (int k = 0; k < count; ++k) {
    double *dst = ((count - 1 - k) % 2 == 0) ? d_c_out : tmp;   // the last substep lands in d_c_out
    const int order = (dt_changed && k < pred) ? 0 : (*n_old < pred ? *n_old : pred);   // AdamsBashforthMoulton.C:90-91
    const int slot_new = (*head + 1) % ring_size;
    const double *old[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < order; ++i) old[i] = d_Nhat_ring[((*head - i) % ring_size + ring_size) % ring_size];
    MRL_TRY(mrl_ch_substep(ctx, p, src, dst, d_Nhat_ring[slot_new], old, order, sub_dt, nullptr, k == count - 1 ? d_mu : nullptr,
                           MRL_CARRY_NONE));
    src = dst;
    if (advance && k < count - 1) {
      *head = slot_new;
      if (*n_old < pred) *n_old += 1;
    }
  }
  return MRL_OK;
}

}  // extern "C"
