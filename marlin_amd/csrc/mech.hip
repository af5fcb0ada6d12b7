// de Geus finite-strain mechanics: Gamma-operator projection, St-Venant-Kirchhoff stress / tangent
// application and the Newton-CG driver.
//   G(A)    = ifft( Ghat4 : fft(A) )                      src/tensor_computes/FFTMechanics.C:74-84,105-106
//   K_dF    = trans2(ddot42(K4, trans2(dF)))              src/tensor_computes/FFTMechanics.C:107-108
//   P, K4   : HyperElasticIsotropic::computeBuffer        src/tensor_computes/HyperElasticIsotropic.C:42-52
//   CG      : MooseTensor::conjugateGradientSolve         include/utils/MarlinUtils.h:55-131
//   Newton  : FFTMechanics::computeBuffer                 src/tensor_computes/FFTMechanics.C:96-163
//
// Nothing 4th-order is ever stored.  With Ghat4_ijlm = delta_im q_j q_l / |q|^2 the projection is
//   (Ghat4 : A)_ij = q_j (sum_k A_ik q_k) / |q|^2   (0 at q = 0)
// and with C4 = K II + 2 mu (I4s - II/3), S = C4 : E, E = (F^T F - I)/2 the tangent applied to dF is
//   K_dF(dF) = dF.S + F.Y,   W = F^T dF,  Y = K tr(W) I + 2 mu (sym(W) - tr(W)/3 I)
// (the contraction of K4 = S.I4 + I4rt:((F.C4).F^T):I4rt written out).  The "1/3" is
// the reference's literal 1./3. in every dimension.
// Fields are value-major [grid][D][D] as in the reference.
#include "mech_math.h"
#include "slab_stages.h"

namespace mrl {

template <int D, bool SOA>
__global__ void __launch_bounds__(256) k_mech_stress(const double *__restrict__ F, const double *__restrict__ K,
                                                      const double *__restrict__ mu, double *__restrict__ P,
                                                      long long npts) {
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npts; p += (long long)gridDim.x * 256) {
    const Mat<D> f = load_mat<D, SOA>(F, p, npts);
    const Mat<D> S = svk_S<D>(f, K[p], mu[p]);
    Mat<D> o;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) s += f.a[i][k] * S.a[k][j];
        o.a[i][j] = s;
      }
    store_mat<D, SOA>(P, p, npts, o);
  }
}


// out = K_dF(dF) ; bcast: dF is ONE tensor (D*D doubles) broadcast over the grid.
// DIR fuses the CG direction update into the operator application: dF <- r + beta*dF (beta = S[i_num]/S[i_den],
// written back: dF is the search direction p), then out = K_dF(dF)    (MarlinUtils.h:116 + FFTMechanics.C:107-108)
template <int D, bool SOA, bool DIR>
__global__ void __launch_bounds__(256) k_mech_tangent(const double *__restrict__ F, const double *__restrict__ K,
                                                       const double *__restrict__ mu, double *__restrict__ dF,
                                                       int bcast, double *__restrict__ out, long long npts,
                                                       const double *__restrict__ r, const double *__restrict__ S,
                                                       int i_num, int i_den) {
#pragma clang fp contract(off)
  double beta = 0.0;
  if (DIR) beta = S[i_num] / S[i_den];
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npts; p += (long long)gridDim.x * 256) {
    const Mat<D> f = load_mat<D, SOA>(F, p, npts);
    Mat<D> d = bcast ? load_mat<D, false>(dF, 0, 1) : load_mat<D, SOA>(dF, p, npts);
    if (DIR) {
      const Mat<D> rv = load_mat<D, SOA>(r, p, npts);
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) d.a[i][j] = rv.a[i][j] + beta * d.a[i][j];
      store_mat<D, SOA>(dF, p, npts, d);
    }
    store_mat<D, SOA>(out, p, npts, svk_tangent<D>(f, d, K[p], mu[p]));
  }
}

// field-major, two grid points per thread: every component stream is read / written 16 B per lane
// NT: stream F, K, mu (read once per CG iteration) and the direction store past the Infinity Cache -- pays off when the 9-field
// vectors are large against its 256 MB (128^3: 151 MB per vector, 0.771 -> 0.737 ms per iteration), loses where they would have
// stayed resident (64^3), so the launchers choose per size
template <bool DIR, bool NT>
__global__ void __launch_bounds__(256) k_mech_tangent_fm2(const double *__restrict__ F, const double *__restrict__ K,
                                                           const double *__restrict__ mu, double *__restrict__ dF,
                                                           int bcast, double *__restrict__ out, long long npts,
                                                           const double *__restrict__ r, const double *__restrict__ S,
                                                           int i_num, int i_den) {
#pragma clang fp contract(off)
  double beta = 0.0;
  if (DIR) beta = S[i_num] / S[i_den];
  const long long half = npts >> 1;  // npts is even on every fast-path shape
  for (long long h = (long long)blockIdx.x * 256 + threadIdx.x; h < half; h += (long long)gridDim.x * 256) {
    Mat<3> f[2], d[2];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const double2 *fp = reinterpret_cast<const double2 *>(F + (long long)c * npts) + h;
      const double2 fv = NT ? ld_nt(fp) : *fp;
      f[0].a[c / 3][c % 3] = fv.x;
      f[1].a[c / 3][c % 3] = fv.y;
      double2 dv;
      if (bcast) {
        dv = make_double2(dF[c], dF[c]);
      } else {
        dv = reinterpret_cast<const double2 *>(dF + (long long)c * npts)[h];
      }
      if (DIR) {
        const double2 rv = reinterpret_cast<const double2 *>(r + (long long)c * npts)[h];
        dv = make_double2(rv.x + beta * dv.x, rv.y + beta * dv.y);
        double2 *pp = reinterpret_cast<double2 *>(dF + (long long)c * npts) + h;   // p: next read at the far end of the Gamma passes
        if (NT)
          st_nt(pp, dv);
        else
          *pp = dv;
      }
      d[0].a[c / 3][c % 3] = dv.x;
      d[1].a[c / 3][c % 3] = dv.y;
    }
    const double2 *Kp = reinterpret_cast<const double2 *>(K) + h, *mp = reinterpret_cast<const double2 *>(mu) + h;
    const double2 Kv = NT ? ld_nt(Kp) : *Kp, mv = NT ? ld_nt(mp) : *mp;
    const Mat<3> o0 = svk_tangent<3>(f[0], d[0], Kv.x, mv.x);
    const Mat<3> o1 = svk_tangent<3>(f[1], d[1], Kv.y, mv.y);
#pragma unroll
    for (int c = 0; c < 9; ++c)
      reinterpret_cast<double2 *>(out + (long long)c * npts)[h] = make_double2(o0.a[c / 3][c % 3], o1.a[c / 3][c % 3]);
  }
}

// in-place projection of a value-major spectrum [n0][n1][n2][D][D] (complex), times `scale`
template <int D>
__global__ void __launch_bounds__(256) k_gamma_project(double2 *__restrict__ spec, long long n0, long long n1,
                                                        long long n2, const double *__restrict__ k0,
                                                        const double *__restrict__ k1, const double *__restrict__ k2,
                                                        double scale) {
  const long long total = n0 * n1 * n2;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long i2 = e % n2, t = e / n2, i1 = t % n1, i0 = t / n1;
    double q[3];
    // internal axes (A0, A1, A2) carry the user axes right-aligned: user axis d = internal 3 - D + d
    const double kk[3] = {k0[i0], k1[i1], k2[i2]};
#pragma unroll
    for (int d = 0; d < D; ++d) q[d] = kk[3 - D + d];
    double Q = 0.0;
#pragma unroll
    for (int d = 0; d < D; ++d) Q += q[d] * q[d];
    const double inv = (Q == 0.0) ? 0.0 : scale / Q;
    double2 *A = spec + e * D * D;
#pragma unroll
    for (int i = 0; i < D; ++i) {
      double2 s = make_double2(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const double2 v = A[i * D + k];
        s.x += v.x * q[k];
        s.y += v.y * q[k];
      }
      s.x *= inv;
      s.y *= inv;
#pragma unroll
      for (int j = 0; j < D; ++j) A[i * D + j] = make_double2(s.x * q[j], s.y * q[j]);
    }
  }
}

// in-place projection of FIELD-MAJOR spectra [D*D][n0][n1][n2] (complex), times `scale`; `off` = internal axis of
// user axis 0 (slab contexts are left-aligned, serial ones right-aligned)
template <int D>
__global__ void __launch_bounds__(256) k_gamma_project_fm(double2 *__restrict__ spec, long long n0, long long n1,
                                                           long long n2, const double *__restrict__ k0,
                                                           const double *__restrict__ k1, const double *__restrict__ k2,
                                                           int off, double scale) {
  const long long total = n0 * n1 * n2;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long i2 = e % n2, t = e / n2, i1 = t % n1, i0 = t / n1;
    const double kk[3] = {k0[i0], k1[i1], k2[i2]};
    double q[3];
#pragma unroll
    for (int d = 0; d < D; ++d) q[d] = kk[off + d];
    double Q = 0.0;
#pragma unroll
    for (int d = 0; d < D; ++d) Q += q[d] * q[d];
    const double inv = (Q == 0.0) ? 0.0 : scale / Q;
#pragma unroll
    for (int i = 0; i < D; ++i) {
      double2 s = make_double2(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const double2 v = spec[(long long)(i * D + k) * total + e];
        s.x += v.x * q[k];
        s.y += v.y * q[k];
      }
      s.x *= inv;
      s.y *= inv;
#pragma unroll
      for (int j = 0; j < D; ++j) spec[(long long)(i * D + j) * total + e] = make_double2(s.x * q[j], s.y * q[j]);
    }
  }
}

// out = a*x + b*y
__global__ void __launch_bounds__(256) k_axpby(double a, const double *x, double b, const double *y, double *out, long long n) {
#pragma clang fp contract(off)
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    out[i] = a * x[i] + b * y[i];
}

// ---- CG vector kernels (device-resident scalars S) ---------------------------------------------
// r = b - Ax ; p = r ; partial sum r.r
__global__ void __launch_bounds__(256) k_cg_init(const double *__restrict__ b, const double *__restrict__ Ax,
                                                  double *__restrict__ r, double *__restrict__ p, long long n,
                                                  double *__restrict__ partial) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const double v = b[i] - Ax[i];
    r[i] = v;
    p[i] = v;
    acc += v * v;
  }
  acc = wave_sum_f64(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// alpha = S[i_rz] / S[i_pAp] ; x += alpha p ; r -= alpha Ap ; partial sum r.r      (16 B per lane per access)
// SKIP_X: r and r.r only -- the solution update is deferred into the next iteration's direction kernel (k_gamma_z_fwd_tangent<XUPD>)
template <bool NT, bool SKIP_X = false>
__global__ void __launch_bounds__(256) k_cg_update(const double *__restrict__ S, int i_rz, int i_pAp,
                                                    double *__restrict__ x, double *__restrict__ r,
                                                    const double *__restrict__ p, const double *__restrict__ Ap,
                                                    long long n, double *__restrict__ partial, const int *__restrict__ stop = nullptr) {
#pragma clang fp contract(off)
  __shared__ double sh[4];
  if (stop && *stop) return;  // a solve that converged while this iteration was already enqueued (k_reduce_final_cg)
  const double alpha = S[i_rz] / S[i_pAp];
  double acc = 0.0;
  const long long n2 = n >> 1;
  double2 *x2 = reinterpret_cast<double2 *>(x), *r2 = reinterpret_cast<double2 *>(r);
  const double2 *p2 = reinterpret_cast<const double2 *>(p), *A2 = reinterpret_cast<const double2 *>(Ap);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long long)gridDim.x * 256) {
    const double2 rv = r2[i], av = A2[i];
    if (!SKIP_X) {
      const double2 xv = NT ? ld_nt(x2 + i) : x2[i], pv = NT ? ld_nt(p2 + i) : p2[i];   // NT: x, p are not re-read
      const double2 xn = make_double2(xv.x + alpha * pv.x, xv.y + alpha * pv.y);         // before ~3 GB of other traffic
      if (NT)
        st_nt(x2 + i, xn);
      else
        x2[i] = xn;
    }
    const double2 v = make_double2(rv.x - alpha * av.x, rv.y - alpha * av.y);
    r2[i] = v;
    acc += v.x * v.x + v.y * v.y;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const long long i = n - 1;
    if (!SKIP_X) x[i] = x[i] + alpha * p[i];
    const double v = r[i] - alpha * Ap[i];
    r[i] = v;
    acc += v * v;
  }
  acc = wave_sum_f64(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// x += (S[i_a] / S[i_b]) p : the last, still pending solution update of a CG solve with deferred updates
__global__ void __launch_bounds__(256) k_axpy_ratio(const double *__restrict__ S, int i_a, int i_b, double *__restrict__ x,
                                                     const double *__restrict__ p, long long n) {
#pragma clang fp contract(off)
  const double alpha = S[i_a] / S[i_b];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) x[i] = x[i] + alpha * p[i];
}

// y[i] += x[i]                       (period = 0, block = 0)
// y[i] += x[i % period]              (value-major broadcast of one tensor, period = D*D)
// y[i] += x[i / block]               (field-major broadcast, block = number of grid points)
__global__ void __launch_bounds__(256) k_add(double *__restrict__ y, const double *__restrict__ x, long long n,
                                              long long period, long long block) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    y[i] = y[i] + x[block ? i / block : (period ? i % period : i)];
}

// value-major [npts][dd] <-> field-major [dd][npts]: one thread per grid point moves all dd components, so the
// field-major side is coalesced per component and the value-major side is touched in whole 8*dd-byte records by one
// wave within a few instructions (one thread per ELEMENT leaves the value-major side as isolated 8-byte accesses)
template <bool TO_SOA, int DD>
__global__ void __launch_bounds__(256) k_relayout(const double *__restrict__ in, double *__restrict__ out, long long npts) {
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npts; p += (long long)gridDim.x * 256) {
    double v[DD];
    if (TO_SOA) {
#pragma unroll
      for (int c = 0; c < DD; ++c) v[c] = in[p * DD + c];
#pragma unroll
      for (int c = 0; c < DD; ++c) out[(long long)c * npts + p] = v[c];
    } else {
#pragma unroll
      for (int c = 0; c < DD; ++c) v[c] = in[(long long)c * npts + p];
#pragma unroll
      for (int c = 0; c < DD; ++c) out[p * DD + c] = v[c];
    }
  }
}

static int relayout_any(mrl_ctx *ctx, bool to_soa, const double *in, double *out, long long npts, int dd);

static inline int grid_for(long long n) {
  long long b = (n + 255) / 256;
  if (b > kRedBlocks) b = kRedBlocks;
  if (b < 1) b = 1;
  return (int)b;
}

// 9-field vectors of at least 96 MB do not survive in the 256 MB Infinity Cache from one use to the next: stream them
static inline bool mech_stream_vectors(long long npts) { return 72.0 * (double)npts >= 96.0e6; }

// serial = false: pointwise operators, valid on the local slab of any context
static int check_dim(mrl_ctx *ctx, const char *what, bool serial = true) {
  if (ctx->dim != 2 && ctx->dim != 3)
    return set_error(ctx, MRL_ERR_UNSUPPORTED, "%s: mechanics needs a 2-D or 3-D domain", what);
  if (!serial) return MRL_OK;
  if (ctx->slab)
    return set_error(ctx, MRL_ERR_UNSUPPORTED, "%s: serial contexts only (slab contexts: mrl_slab_gamma_project + the slab FFT stages)", what);
  if (ctx->spectrum != MRL_SPECTRUM_HALF)
    return set_error(ctx, MRL_ERR_UNSUPPORTED, "%s: needs spectrum = MRL_SPECTRUM_HALF", what);
  return MRL_OK;
}

bool mech_fast_ok(const mrl_ctx *ctx);
bool gamma_tangent_fusable(const mrl_ctx *ctx);
int gamma_fast_tangent_dir(mrl_ctx *ctx, const double *F, const double *K, const double *mu, double *p, const double *r,
                           const double *S, int i_num, int i_den, double *out, double *d_dot, bool nt, double *x, int i_arz,
                           int i_apAp, const int *stop);
int reduce_finalize_cg(mrl_ctx *ctx, int nb, double *d_scalar, int slot, double thr, int *stop);
int gamma_fast(mrl_ctx *ctx, const double *A, double *out, double scale, const double *dotv = nullptr,
               double *d_dot = nullptr);

int stress_launch(mrl_ctx *ctx, const double *F, const double *K, const double *mu, double *P, bool soa) {
  const long long npts = real_count_local(ctx);
  ProfScope ps(ctx, "mech_stress", (double)npts * 8.0 * (2 * ctx->dim * ctx->dim + 2));
  const dim3 g(grid_for(npts)), b(256);
  if (ctx->dim == 3 && soa)
    hipLaunchKernelGGL((k_mech_stress<3, true>), g, b, 0, ctx->stream, F, K, mu, P, npts);
  else if (ctx->dim == 3)
    hipLaunchKernelGGL((k_mech_stress<3, false>), g, b, 0, ctx->stream, F, K, mu, P, npts);
  else
    hipLaunchKernelGGL((k_mech_stress<2, false>), g, b, 0, ctx->stream, F, K, mu, P, npts);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

int tangent_launch(mrl_ctx *ctx, const double *F, const double *K, const double *mu, const double *dF, bool bcast,
                   double *out, bool soa) {
  const long long npts = real_count_local(ctx);
  const int dd = ctx->dim * ctx->dim;
  ProfScope ps(ctx, "mech_tangent", (double)npts * 8.0 * ((bcast ? 2 : 3) * dd + 2));
  const dim3 g(grid_for(npts)), b(256);
  double *d = const_cast<double *>(dF);  // only written by the DIR variant
  if (ctx->dim == 3 && soa)
  {
    if (mech_stream_vectors(npts))
      hipLaunchKernelGGL((k_mech_tangent_fm2<false, true>), dim3(grid_for(npts / 2)), b, 0, ctx->stream, F, K, mu, d,
                         bcast ? 1 : 0, out, npts, nullptr, nullptr, 0, 0);
    else
      hipLaunchKernelGGL((k_mech_tangent_fm2<false, false>), dim3(grid_for(npts / 2)), b, 0, ctx->stream, F, K, mu, d,
                         bcast ? 1 : 0, out, npts, nullptr, nullptr, 0, 0);
  }
  else if (ctx->dim == 3)
    hipLaunchKernelGGL((k_mech_tangent<3, false, false>), g, b, 0, ctx->stream, F, K, mu, d, bcast ? 1 : 0, out, npts,
                       nullptr, nullptr, 0, 0);
  else
    hipLaunchKernelGGL((k_mech_tangent<2, false, false>), g, b, 0, ctx->stream, F, K, mu, d, bcast ? 1 : 0, out, npts,
                       nullptr, nullptr, 0, 0);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

// p <- r + (S[i_num]/S[i_den]) p ; out = K_dF(p)
int tangent_dir_launch(mrl_ctx *ctx, const double *F, const double *K, const double *mu, double *p, const double *r,
                       const double *S, int i_num, int i_den, double *out, bool soa) {
  const long long npts = real_count_local(ctx);
  const int dd = ctx->dim * ctx->dim;
  ProfScope ps(ctx, "mech_tangent_dir", (double)npts * 8.0 * (5 * dd + 2));
  const dim3 g(grid_for(npts)), b(256);
  if (ctx->dim == 3 && soa)
  {
    if (mech_stream_vectors(npts))
      hipLaunchKernelGGL((k_mech_tangent_fm2<true, true>), dim3(grid_for(npts / 2)), b, 0, ctx->stream, F, K, mu, p, 0, out, npts,
                         r, S, i_num, i_den);
    else
      hipLaunchKernelGGL((k_mech_tangent_fm2<true, false>), dim3(grid_for(npts / 2)), b, 0, ctx->stream, F, K, mu, p, 0, out, npts,
                         r, S, i_num, i_den);
  }
  else if (ctx->dim == 3)
    hipLaunchKernelGGL((k_mech_tangent<3, false, true>), g, b, 0, ctx->stream, F, K, mu, p, 0, out, npts, r, S, i_num, i_den);
  else
    hipLaunchKernelGGL((k_mech_tangent<2, false, true>), g, b, 0, ctx->stream, F, K, mu, p, 0, out, npts, r, S, i_num, i_den);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

static int relayout_any(mrl_ctx *ctx, bool to_soa, const double *in, double *out, long long npts, int dd) {
  const dim3 g(grid_for(npts)), b(256);
#define MRL_RL(DD)                                                                                  \
  if (to_soa)                                                                                       \
    hipLaunchKernelGGL((k_relayout<true, DD>), g, b, 0, ctx->stream, in, out, npts);                \
  else                                                                                              \
    hipLaunchKernelGGL((k_relayout<false, DD>), g, b, 0, ctx->stream, in, out, npts);
  switch (dd) {
    case 1: MRL_RL(1) break;
    case 2: MRL_RL(2) break;
    case 3: MRL_RL(3) break;
    case 4: MRL_RL(4) break;
    case 6: MRL_RL(6) break;
    case 9: MRL_RL(9) break;
    default: return set_error(ctx, MRL_ERR_UNSUPPORTED, "relayout: %d components per point not supported (1,2,3,4,6,9)", dd);
  }
#undef MRL_RL
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

int relayout_launch(mrl_ctx *ctx, bool to_soa, const double *in, double *out) {
  const long long npts = real_count_local(ctx);
  const int dd = ctx->dim * ctx->dim;
  ProfScope ps(ctx, to_soa ? "mech_to_field_major" : "mech_to_value_major", 16.0 * npts * dd);
  return relayout_any(ctx, to_soa, in, out, npts, dd);
}

// out = scale * G(A), value-major fields through the generic batched transforms
int gamma_launch(mrl_ctx *ctx, const double *A, double *out, double scale) {
  const int dd = ctx->dim * ctx->dim;
  const long long nspec = spec_count_local(ctx);
  MRL_TRY(ensure_work(ctx, 4, sizeof(cplx) * nspec * dd));
  double *spec = ctx->d_work[4];
  MRL_TRY(fft_forward_serial(ctx, A, spec, dd, 1));
  {
    ProfScope ps(ctx, "mech_gamma_project", 32.0 * (double)nspec * dd);
    const int nb = grid_for(nspec);
    if (ctx->dim == 3)
      hipLaunchKernelGGL(k_gamma_project<3>, dim3(nb), dim3(256), 0, ctx->stream, reinterpret_cast<double2 *>(spec),
                         ctx->nrec[0], ctx->nrec[1], ctx->nrec[2], ctx->d_k[0], ctx->d_k[1], ctx->d_k[2], scale);
    else
      hipLaunchKernelGGL(k_gamma_project<2>, dim3(nb), dim3(256), 0, ctx->stream, reinterpret_cast<double2 *>(spec),
                         ctx->nrec[0], ctx->nrec[1], ctx->nrec[2], ctx->d_k[0], ctx->d_k[1], ctx->d_k[2], scale);
    MRL_HIP(ctx, hipGetLastError());
  }
  return fft_inverse_serial(ctx, spec, out, dd, 1);
}

}  // namespace mrl

using namespace mrl;

extern "C" {

int mrl_gamma_apply(mrl_ctx *ctx, const double *d_A, double *d_out) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_NO_PENCIL(ctx, "mrl_gamma_apply");
  MRL_TRY(check_dim(ctx, "mrl_gamma_apply"));
  if (!d_A || !d_out) return set_error(ctx, MRL_ERR_INVALID, "mrl_gamma_apply: null buffer");
  if (mech_fast_ok(ctx)) {
    // the fused kernels work on field-major data: convert at this (value-major) boundary
    const size_t vb = sizeof(double) * (size_t)(real_count_local(ctx) * 9 + 2);
    MRL_TRY(ensure_work(ctx, 9, vb));
    MRL_TRY(relayout_launch(ctx, true, d_A, ctx->d_work[9]));
    MRL_TRY(gamma_fast(ctx, ctx->d_work[9], ctx->d_work[9], 1.0));
    return relayout_launch(ctx, false, ctx->d_work[9], d_out);
  }
  return gamma_launch(ctx, d_A, d_out, 1.0);
}

int mrl_slab_gamma_project(mrl_ctx *ctx, double *d_spec, double scale) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_TRY(check_dim(ctx, "mrl_slab_gamma_project", false));
  if (!d_spec) return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_gamma_project: null buffer");
  const long long nspec = spec_count_local(ctx);
  ProfScope ps(ctx, "gamma_project_fm", 32.0 * (double)nspec * ctx->dim * ctx->dim);
  const int nb = grid_for(nspec);
  if (ctx->dim == 3)
    hipLaunchKernelGGL(k_gamma_project_fm<3>, dim3(nb), dim3(256), 0, ctx->stream, reinterpret_cast<double2 *>(d_spec),
                       ctx->nrec[0], ctx->nrec[1], ctx->nrec[2], ctx->d_k[0], ctx->d_k[1], ctx->d_k[2], ctx->off, scale);
  else
    hipLaunchKernelGGL(k_gamma_project_fm<2>, dim3(nb), dim3(256), 0, ctx->stream, reinterpret_cast<double2 *>(d_spec),
                       ctx->nrec[0], ctx->nrec[1], ctx->nrec[2], ctx->d_k[0], ctx->d_k[1], ctx->d_k[2], ctx->off, scale);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

int mrl_relayout(mrl_ctx *ctx, int to_field_major, const double *d_in, double *d_out, int64_t npts, int32_t ncomp) {
  if (!ctx) return MRL_ERR_INVALID;
  if (!d_in || !d_out || d_in == d_out || npts < 0 || ncomp < 1)
    return set_error(ctx, MRL_ERR_INVALID, "mrl_relayout: bad argument (in and out must be distinct)");
  if (npts == 0) return MRL_OK;
  ProfScope ps(ctx, to_field_major ? "to_field_major" : "to_value_major", 16.0 * npts * ncomp);
  return relayout_any(ctx, to_field_major != 0, d_in, d_out, (long long)npts, ncomp);
}

int mrl_axpby(mrl_ctx *ctx, double a, const double *d_x, double b, const double *d_y, double *d_out, int64_t n) {
  if (!ctx) return MRL_ERR_INVALID;
  if (!d_x || !d_y || !d_out || n < 0) return set_error(ctx, MRL_ERR_INVALID, "mrl_axpby: bad argument");
  if (n == 0) return MRL_OK;
  ProfScope ps(ctx, "axpby", 24.0 * n);
  hipLaunchKernelGGL(k_axpby, dim3(grid_for(n)), dim3(256), 0, ctx->stream, a, d_x, b, d_y, d_out, (long long)n);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

int mrl_axpy(mrl_ctx *ctx, double a, const double *d_x, double *d_y, int64_t n) { return mrl_axpby(ctx, a, d_x, 1.0, d_y, d_y, n); }

int mrl_mech_stress(mrl_ctx *ctx, const double *d_F, const double *d_K, const double *d_mu, double *d_P) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_TRY(check_dim(ctx, "mrl_mech_stress", false));
  if (!d_F || !d_K || !d_mu || !d_P) return set_error(ctx, MRL_ERR_INVALID, "mrl_mech_stress: null buffer");
  return stress_launch(ctx, d_F, d_K, d_mu, d_P, false);
}

int mrl_mech_tangent_apply(mrl_ctx *ctx, const double *d_F, const double *d_K, const double *d_mu,
                           const double *d_dF, double *d_out) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_TRY(check_dim(ctx, "mrl_mech_tangent_apply", false));
  if (!d_F || !d_K || !d_mu || !d_dF || !d_out)
    return set_error(ctx, MRL_ERR_INVALID, "mrl_mech_tangent_apply: null buffer");
  return tangent_launch(ctx, d_F, d_K, d_mu, d_dF, false, d_out, false);
}

int mrl_mech_stress_fm(mrl_ctx *ctx, const double *d_F, const double *d_K, const double *d_mu, double *d_P) {
  if (!ctx) return MRL_ERR_INVALID;
  if (ctx->dim != 3) return set_error(ctx, MRL_ERR_UNSUPPORTED, "mrl_mech_stress_fm: 3-D only");
  if (!d_F || !d_K || !d_mu || !d_P) return set_error(ctx, MRL_ERR_INVALID, "mrl_mech_stress_fm: null buffer");
  return stress_launch(ctx, d_F, d_K, d_mu, d_P, true);
}

int mrl_mech_tangent_apply_fm(mrl_ctx *ctx, const double *d_F, const double *d_K, const double *d_mu, const double *d_dF,
                              double *d_out) {
  if (!ctx) return MRL_ERR_INVALID;
  if (ctx->dim != 3) return set_error(ctx, MRL_ERR_UNSUPPORTED, "mrl_mech_tangent_apply_fm: 3-D only");
  if (real_count_local(ctx) % 2) return set_error(ctx, MRL_ERR_UNSUPPORTED, "mrl_mech_tangent_apply_fm: odd number of local points");
  if (!d_F || !d_K || !d_mu || !d_dF || !d_out)
    return set_error(ctx, MRL_ERR_INVALID, "mrl_mech_tangent_apply_fm: null buffer");
  return tangent_launch(ctx, d_F, d_K, d_mu, d_dF, false, d_out, true);
}

// CG building blocks for callers that own the iteration (slab contexts: the scalars are all-reduced between the calls)
static int put_scalars(mrl_ctx *ctx, double a, double b, double **S) {
  *S = ctx->d_red + kScalarBase + 8;
  // source in PINNED host memory (a ring of eight pairs in the context's scratch: every caller synchronises at least once per CG
  // iteration, so a pair is never overwritten before its copy has run) -- no reliance on how the runtime stages pageable copies
  static_assert(32 + 2 * 8 <= 64, "pinned scratch too small");
  double *hs = ctx->h_red + 32 + 2 * (ctx->scalar_ring++ & 7);
  hs[0] = a;
  hs[1] = b;
  MRL_HIP(ctx, hipMemcpyAsync(*S, hs, 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  return MRL_OK;
}

int mrl_cg_update(mrl_ctx *ctx, double alpha, double *d_x, double *d_r, const double *d_p, const double *d_Ap, int64_t n,
                  double *h_rr_local) {
  if (!ctx) return MRL_ERR_INVALID;
  if (!d_x || !d_r || !d_p || !d_Ap || !h_rr_local || n < 0) return set_error(ctx, MRL_ERR_INVALID, "mrl_cg_update: bad argument");
  double *S;
  MRL_TRY(put_scalars(ctx, alpha, 1.0, &S));  // the kernel forms alpha = S[0] / S[1]
  int nb = (int)((n / 2 + 255) / 256);
  nb = nb < 1 ? 1 : (nb > kRedBlocks ? kRedBlocks : nb);
  {
    ProfScope ps(ctx, "cg_update", 48.0 * (double)n);
    if (mech_stream_vectors(n / 9))
      hipLaunchKernelGGL(k_cg_update<true>, dim3(nb), dim3(256), 0, ctx->stream, S, 0, 1, d_x, d_r, d_p, d_Ap, (long long)n, ctx->d_red);
    else
      hipLaunchKernelGGL(k_cg_update<false>, dim3(nb), dim3(256), 0, ctx->stream, S, 0, 1, d_x, d_r, d_p, d_Ap, (long long)n, ctx->d_red);
    MRL_HIP(ctx, hipGetLastError());
  }
  double *slot = ctx->d_red + kScalarBase;
  return reduce_finalize_to_host(ctx, nb, 1, slot, h_rr_local);
}

int mrl_cg_update_r(mrl_ctx *ctx, double alpha, double *d_r, const double *d_Ap, int64_t n, double *h_rr_local) {
  if (!ctx) return MRL_ERR_INVALID;
  if (!d_r || !d_Ap || !h_rr_local || n < 0) return set_error(ctx, MRL_ERR_INVALID, "mrl_cg_update_r: bad argument");
  double *S;
  MRL_TRY(put_scalars(ctx, alpha, 1.0, &S));
  int nb = (int)((n / 2 + 255) / 256);
  nb = nb < 1 ? 1 : (nb > kRedBlocks ? kRedBlocks : nb);
  {
    ProfScope ps(ctx, "cg_update_r", 24.0 * (double)n);
    hipLaunchKernelGGL((k_cg_update<false, true>), dim3(nb), dim3(256), 0, ctx->stream, S, 0, 1, nullptr, d_r, nullptr, d_Ap, (long long)n,
                       ctx->d_red);
    MRL_HIP(ctx, hipGetLastError());
  }
  double *slot = ctx->d_red + kScalarBase;
  return reduce_finalize_to_host(ctx, nb, 1, slot, h_rr_local);
}

int mrl_mech_tangent_dir_fm(mrl_ctx *ctx, const double *d_F, const double *d_K, const double *d_mu, double *d_p,
                            const double *d_r, double beta, double *d_out) {
  if (!ctx) return MRL_ERR_INVALID;
  if (ctx->dim != 3) return set_error(ctx, MRL_ERR_UNSUPPORTED, "mrl_mech_tangent_dir_fm: 3-D only");
  if (real_count_local(ctx) % 2) return set_error(ctx, MRL_ERR_UNSUPPORTED, "mrl_mech_tangent_dir_fm: odd number of local points");
  if (!d_F || !d_K || !d_mu || !d_p || !d_r || !d_out) return set_error(ctx, MRL_ERR_INVALID, "mrl_mech_tangent_dir_fm: null buffer");
  double *S;
  MRL_TRY(put_scalars(ctx, beta, 1.0, &S));
  return tangent_dir_launch(ctx, d_F, d_K, d_mu, d_p, d_r, S, 0, 1, d_out, true);
}

}  // extern "C"

// value-major identity field [npts][dim][dim]
__global__ void k_fill_identity(double *out, long long n, int dim) {
  const int dd = dim * dim;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % dd);
    out[e] = (c / dim == c % dim) ? 1.0 : 0.0;
  }
}

// FFTMechanics::computeBuffer.  small: stop after the first linear solve and report sigma = C4 : (F - I) = K_dF(I; F - I) instead of
// P(F) (mrl_mech_small_strain; the caller passes the identity field as d_F)
static int newton_cg_impl(mrl_ctx *ctx, const mrl_mech_params *prm, const double *d_F, const double *d_K, const double *d_mu,
                          const double *d_applied, double *d_Fnew, double *d_P, mrl_mech_stats *stats, bool small) {
  // slab contexts: the same solve on the rank's y-slab, the Gamma operator over the library-owned exchanges and every norm / dot
  // product summed over the ranks on the device (the reference's norms are serial-only, DomainAction.C:1564-1567)
  const bool dist = ctx->slab;
  MRL_TRY(check_dim(ctx, "mrl_mech_newton_cg", !dist));
  if (dist && !ctx->comm)
    return set_error(ctx, MRL_ERR_INVALID, "mrl_mech_newton_cg on a slab context needs a communicator (mrl_ctx_attach_comm)");
  if (!prm || !d_F || !d_K || !d_mu || !d_Fnew || !d_P)
    return set_error(ctx, MRL_ERR_INVALID, "mrl_mech_newton_cg: null argument");
  if (d_Fnew == d_F) return set_error(ctx, MRL_ERR_INVALID, "mrl_mech_newton_cg: Fnew must not alias F");
  const int dd = ctx->dim * ctx->dim;
  const long long npts = real_count_local(ctx);
  const long long n = npts * dd;
  const size_t vb = sizeof(double) * (size_t)(n + 2);
  for (int s = 5; s <= 10; ++s) MRL_TRY(ensure_work(ctx, s, vb));
  double *b = ctx->d_work[5], *r = ctx->d_work[6], *p = ctx->d_work[7], *Ap = ctx->d_work[8], *tmp = ctx->d_work[9],
         *x = ctx->d_work[10];
  // Fast-path shapes run the whole solve on field-major vectors [9][grid] (coalesced streams for every kernel);
  // the caller's value-major F is converted once on entry, Fnew and P once on exit.
  const bool soa = dist ? (ctx->dim == 3 && slab_mech_soa(ctx)) : mech_fast_ok(ctx);
  double *Fin = nullptr, *Fwork = d_Fnew, *Pwork = d_P;
  if (soa) {
    for (int s = 11; s <= 13; ++s) MRL_TRY(ensure_work(ctx, s, vb));
    Fin = ctx->d_work[11];
    Fwork = ctx->d_work[12];
    Pwork = ctx->d_work[13];
    MRL_TRY(relayout_launch(ctx, true, d_F, Fin));
  }
  auto gamma = [&](const double *A, double *out, double scale) -> int {
    if (dist) return soa ? slab_gamma_fm(ctx, A, out, scale, nullptr, nullptr) : slab_gamma_vm(ctx, A, out, scale);
    return soa ? gamma_fast(ctx, A, out, scale) : gamma_launch(ctx, A, out, scale);
  };
  // a device scalar produced on the stream becomes the sum over the ranks (in place); h != nullptr: also read back (one sync)
  auto global = [&](double *slot, double *h) -> int {
    if (dist) return slab_allreduce_scalars(ctx, slot, 1, slot, h);
    return h ? read_scalars(ctx, slot, 1, h) : MRL_OK;
  };
  double *S = ctx->d_red + kScalarBase + 32;  // device scalars: [0],[2] r.r ping-pong, [1] p.Ap, [3] scratch
  const int nb = grid_for(n);
  const long long l_max_its = prm->l_max_its > 0 ? prm->l_max_its : ctx->n[0] * ctx->n[1] * ctx->n[2];
  mrl_mech_stats st{};
  double h[4];

  // _u = _tF ; constitutive at F (the tangent of the first linear solve stays linearised at F)
  const double *lin = soa ? Fin : d_F;
  MRL_HIP(ctx, hipMemcpyAsync(Fwork, lin, sizeof(double) * n, hipMemcpyDeviceToDevice, ctx->stream));
  // b = -G_K_dF(applied.expand) | -G_K_dF(0)                              FFTMechanics.C:116-117
  if (d_applied) {
    MRL_TRY(tangent_launch(ctx, lin, d_K, d_mu, d_applied, true, tmp, soa));
    MRL_TRY(gamma(tmp, b, -1.0));
    hipLaunchKernelGGL(k_add, dim3(nb), dim3(256), 0, ctx->stream, Fwork, d_applied, n, soa ? 0LL : (long long)dd,
                       soa ? npts : 0LL);  // :120-121
  } else {
    MRL_HIP(ctx, hipMemsetAsync(b, 0, sizeof(double) * n, ctx->stream));
  }
  MRL_TRY(reduce_async(ctx, 2, Fwork, Fwork, n, S + 3));
  MRL_TRY(global(S + 3, h));
  const double Fn = sqrt(h[0]);  // :123-124
  st.Fn = Fn;
  MRL_HIP(ctx, hipMemsetAsync(x, 0, sizeof(double) * n, ctx->stream));  // dFm = zeros_like(b)
  bool x_is_zero = true;  // until the first solve has run: G_K_dF(0) is exactly 0 (tangent, transforms and projection of zeros), so the
                          // initial residual b - A x0 of that solve is b itself -- no operator application needed for it

  auto apply_A = [&](const double *v, double *out) -> int {  // G_K_dF
    MRL_TRY(tangent_launch(ctx, lin, d_K, d_mu, v, false, tmp, soa));
    return gamma(tmp, out, 1.0);
  };

  const bool fuse_dir = soa && (dist ? slab_gamma_tangent_fusable(ctx) != 0 : gamma_tangent_fusable(ctx)) && !(ctx->exp & 32);
  // Look-ahead (round 4, one GPU, fused direction kernel): iteration k + 1 is enqueued BEFORE the host reads the residual norm of
  // iteration k, so the GPU never idles behind the per-iteration read (128^3: the kernels of an iteration sum to 0.49 ms, the loop ran
  // at 0.61 ms).  The convergence test of MarlinUtils.h:118-121 is taken on the device (k_reduce_final_cg); once it holds, every kernel
  // of the iterations enqueued ahead returns at once, so x, p, r and the scalars are exactly those of the converged iteration: same
  // iteration count, same solution as with the blocking read (experiment bit 1 << 26 restores it, A/B).
  // (not while the per-kernel profile is being taken: the iteration enqueued past convergence launches kernels that return at once,
  // and their microsecond launches would be averaged into the profile slots as if they had moved their algorithmic bytes -- ADVICE r04;
  // the timed solves run with the look-ahead, the profile pass with the blocking read, same iteration counts and fields)
  const bool look = fuse_dir && !dist && ctx->d_h_red != nullptr && !(ctx->exp & (1 << 26)) && !ctx->profiling;
  int *stop = reinterpret_cast<int *>(S + 8);
  if (look) {
    for (int e = 0; e < 2; ++e)
      if (!ctx->cg_ev[e]) MRL_HIP(ctx, hipEventCreateWithFlags(&ctx->cg_ev[e], hipEventDisableTiming));
  }
  int iiter = 0;
  while (true) {
    // ---- conjugateGradientSolve(G_K_dF, b, dFm, l_tol, l_max_its)        MarlinUtils.h:55-123
    int its = 0;
    double res_norm = 0.0;
    MRL_TRY(reduce_async(ctx, 2, b, b, n, S + 3));
    MRL_TRY(global(S + 3, h));
    const double b_norm = sqrt(h[0]);
    if (b_norm != 0.0) {
      if (x_is_zero && !(ctx->exp & (1 << 27))) {
        MRL_HIP(ctx, hipMemsetAsync(Ap, 0, sizeof(double) * n, ctx->stream));  // = A x0 for x0 = 0, bit for bit
      } else {
        MRL_TRY(apply_A(x, Ap));
      }
      x_is_zero = false;
      hipLaunchKernelGGL(k_cg_init, dim3(nb), dim3(256), 0, ctx->stream, b, Ap, r, p, n, ctx->d_red);
      MRL_TRY(reduce_finalize(ctx, nb, 1, S + 0));
      MRL_TRY(global(S + 0, nullptr));
      int i_old = 0, i_new = 2;
      int pend_rz = -1;  // slot of r.r of the iteration whose x update is still pending (deferred updates, fuse_dir)
      its = (int)l_max_its;
      const double thr = prm->l_tol * b_norm;
      const int *stop_k = look ? stop : nullptr;
      if (look) MRL_HIP(ctx, hipMemsetAsync(stop, 0, sizeof(double), ctx->stream));
      bool seen = false;  // look-ahead: the verdict of the last enqueued iteration has been read
      for (long long k = 0; k < l_max_its; ++k) {
        const int pend_before = pend_rz;   // (look-ahead) what is pending if this iteration turns out to be a no-op
        if (k == 0) {
          MRL_TRY(apply_A(p, Ap));
        } else {
          // p = r + beta p (beta = rr_new / rr_old of the previous iteration) fused into the operator application
          if (fuse_dir) {  // ... and into the forward z pass of G: K4:p is never written; the pending x update rides along
            if (dist) {
              MRL_TRY(slab_gamma_tangent_z(ctx, lin, d_K, d_mu, p, r, S, i_old, i_new, pend_rz >= 0 ? x : nullptr, pend_rz, 1));
              MRL_TRY(slab_gamma_fm(ctx, nullptr, Ap, 1.0, p, S + 1));
            } else {
              MRL_TRY(gamma_fast_tangent_dir(ctx, lin, d_K, d_mu, p, r, S, i_old, i_new, Ap, S + 1, mech_stream_vectors(npts), x,
                                             pend_rz, 1, stop_k));
            }
            pend_rz = -1;
          } else {
            MRL_TRY(tangent_dir_launch(ctx, lin, d_K, d_mu, p, r, S, i_old, i_new, tmp, soa));
            if (soa) {  // p.Ap taken in the last pass of G
              MRL_TRY(dist ? slab_gamma_fm(ctx, tmp, Ap, 1.0, p, S + 1) : gamma_fast(ctx, tmp, Ap, 1.0, p, S + 1));
            } else {
              MRL_TRY(gamma(tmp, Ap, 1.0));
            }
          }
        }
        if (!(soa && k > 0)) {
          ProfScope ps(ctx, "cg_dot_pAp", 16.0 * n);
          MRL_TRY(reduce_async(ctx, 1, p, Ap, n, S + 1));
        }
        MRL_TRY(global(S + 1, nullptr));
        {
          ProfScope ps(ctx, "cg_update_x_r", (fuse_dir ? 24.0 : 48.0) * n);
          if (fuse_dir) {  // x += alpha p is deferred into the next direction kernel (or the k_axpy_ratio after the loop)
            hipLaunchKernelGGL((k_cg_update<false, true>), dim3(nb), dim3(256), 0, ctx->stream, S, i_old, 1, x, r, p, Ap, n, ctx->d_red,
                               stop_k);
            pend_rz = i_old;
          } else if (mech_stream_vectors(npts))
            hipLaunchKernelGGL(k_cg_update<true>, dim3(nb), dim3(256), 0, ctx->stream, S, i_old, 1, x, r, p, Ap, n, ctx->d_red);
          else
            hipLaunchKernelGGL(k_cg_update<false>, dim3(nb), dim3(256), 0, ctx->stream, S, i_old, 1, x, r, p, Ap, n, ctx->d_red);
          if (look) {
            // r.r and the verdict of iteration k go to the pinned pair k & 1; the host reads the pair of iteration k - 1 now, with
            // iteration k already in the queue
            MRL_TRY(reduce_finalize_cg(ctx, nb, S + i_new, (int)(k & 1), thr, stop));
            MRL_HIP(ctx, hipEventRecord(ctx->cg_ev[k & 1], ctx->stream));
          } else if (dist) {  // the one host sync of the iteration
            MRL_TRY(reduce_finalize(ctx, nb, 1, S + i_new));
            MRL_TRY(global(S + i_new, h));
          } else {
            MRL_TRY(reduce_finalize_to_host(ctx, nb, 1, S + i_new, h));  // (no copy command)
          }
        }
        if (look) {
          if (k == 0) {
            const int t0 = i_old;
            i_old = i_new;
            i_new = t0;
            continue;
          }
          MRL_HIP(ctx, hipEventSynchronize(ctx->cg_ev[(k - 1) & 1]));
          res_norm = sqrt(ctx->h_red[8 + 2 * ((k - 1) & 1)]);
          if (ctx->h_red[9 + 2 * ((k - 1) & 1)] != 0.0) {  // iteration k - 1 converged: iteration k (enqueued) does nothing on the device
            its = (int)k;
            pend_rz = pend_before;
            seen = true;
            break;
          }
          const int t1 = i_old;
          i_old = i_new;
          i_new = t1;
          continue;
        }
        res_norm = sqrt(h[0]);
        if (res_norm <= thr) {
          its = (int)k + 1;
          break;
        }
        const int t = i_old;  // rr_new becomes rr_old; the next iteration's beta = S[i_old] / S[i_new]
        i_old = i_new;
        i_new = t;
      }
      if (look && !seen) {  // the loop ran to l_max_its: the last iteration's residual has not been looked at yet
        MRL_HIP(ctx, hipEventSynchronize(ctx->cg_ev[(l_max_its - 1) & 1]));
        res_norm = sqrt(ctx->h_red[8 + 2 * ((l_max_its - 1) & 1)]);
      }
      if (pend_rz >= 0) {
        hipLaunchKernelGGL(k_axpy_ratio, dim3(nb), dim3(256), 0, ctx->stream, S, pend_rz, 1, x, p, n);
        MRL_HIP(ctx, hipGetLastError());
      }
    }
    if (st.newton_its < 64) st.cg_its[st.newton_its] = its;
    st.cg_its_total += its;
    st.newton_its += 1;

    // _u = _u + dFm ; constitutive ; b = -G(P)                              FFTMechanics.C:137-143
    hipLaunchKernelGGL(k_add, dim3(nb), dim3(256), 0, ctx->stream, Fwork, x, n, 0LL, 0LL);
    if (small) {  // linear problem: eps = F - I is the solution, sigma = C4 : eps (the tangent at F = I applied to eps)
      MRL_TRY(mrl_axpby(ctx, 1.0, Fwork, -1.0, lin, tmp, n));
      MRL_TRY(tangent_launch(ctx, lin, d_K, d_mu, tmp, false, Pwork, soa));
      MRL_TRY(reduce_async(ctx, 2, x, x, n, S + 3));
      MRL_TRY(global(S + 3, h));
      st.last_anorm = sqrt(h[0]);
      st.last_rnorm = Fn > 0.0 ? st.last_anorm / Fn : 0.0;
      break;
    }
    lin = Fwork;
    MRL_TRY(stress_launch(ctx, Fwork, d_K, d_mu, Pwork, soa));
    MRL_TRY(gamma(Pwork, b, -1.0));
    MRL_TRY(reduce_async(ctx, 2, x, x, n, S + 3));
    MRL_TRY(global(S + 3, h));
    const double anorm = sqrt(h[0]);
    const double rnorm = anorm / Fn;
    st.last_anorm = anorm;
    st.last_rnorm = rnorm;
    if ((rnorm < prm->nl_rel_tol || anorm < prm->nl_abs_tol) && iiter > 0) break;  // :151-155
    iiter++;
    if (iiter > prm->nl_max_its) {
      if (stats) *stats = st;
      if (soa) {
        relayout_launch(ctx, false, Fwork, d_Fnew);
        relayout_launch(ctx, false, Pwork, d_P);
      }
      return set_error(ctx, MRL_ERR_NOT_CONVERGED,
                       "Exceeded the maximum number of nonlinear iterations without converging.");
    }
  }
  if (soa) {
    MRL_TRY(relayout_launch(ctx, false, Fwork, d_Fnew));
    MRL_TRY(relayout_launch(ctx, false, Pwork, d_P));
  }
  if (stats) *stats = st;
  return MRL_OK;
}

extern "C" {

int mrl_mech_newton_cg(mrl_ctx *ctx, const mrl_mech_params *prm, const double *d_F, const double *d_K,
                       const double *d_mu, const double *d_applied, double *d_Fnew, double *d_P,
                       mrl_mech_stats *stats) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_NO_PENCIL(ctx, "mrl_mech_newton_cg");
  return newton_cg_impl(ctx, prm, d_F, d_K, d_mu, d_applied, d_Fnew, d_P, stats, false);
}

int mrl_mech_small_strain(mrl_ctx *ctx, const mrl_mech_params *prm, const double *d_K, const double *d_mu, const double *d_E,
                          double *d_eps, double *d_sigma, mrl_mech_stats *stats) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_NO_PENCIL(ctx, "mrl_mech_small_strain");
  if (!prm || !d_K || !d_mu || !d_E || !d_eps || !d_sigma) return set_error(ctx, MRL_ERR_INVALID, "mrl_mech_small_strain: null argument");
  MRL_TRY(check_dim(ctx, "mrl_mech_small_strain", !ctx->slab));
  const long long n = real_count_local(ctx) * ctx->dim * ctx->dim;
  MRL_TRY(ensure_work(ctx, 23, sizeof(double) * (size_t)(n + 2)));
  double *ident = ctx->d_work[23];
  hipLaunchKernelGGL(k_fill_identity, dim3(grid_for(n)), dim3(256), 0, ctx->stream, ident, n, ctx->dim);
  MRL_HIP(ctx, hipGetLastError());
  // d_eps receives I + eps first (the impl's Fnew), then eps
  MRL_TRY(newton_cg_impl(ctx, prm, ident, d_K, d_mu, d_E, d_eps, d_sigma, stats, true));
  return mrl_axpby(ctx, 1.0, d_eps, -1.0, ident, d_eps, n);
}

}  // extern "C"
