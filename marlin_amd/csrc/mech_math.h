// Pointwise St-Venant-Kirchhoff helpers shared by the mechanics kernels (HyperElasticIsotropic.C:42-52, MarlinUtils.C:147-187).
#pragma once
#include "mrl_internal.h"

namespace mrl {

template <int D>
struct Mat {
  double a[D][D];
};

// tensor of grid point p: value-major (reference layout) base[p*D*D + c] or field-major base[c*npts + p]
template <int D, bool SOA>
__device__ __forceinline__ Mat<D> load_mat(const double *base, long long p, long long npts) {
  Mat<D> m;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) m.a[i][j] = SOA ? base[(long long)(i * D + j) * npts + p] : base[p * D * D + i * D + j];
  return m;
}

template <int D, bool SOA>
__device__ __forceinline__ void store_mat(double *base, long long p, long long npts, const Mat<D> &m) {
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      if (SOA)
        base[(long long)(i * D + j) * npts + p] = m.a[i][j];
      else
        base[p * D * D + i * D + j] = m.a[i][j];
    }
}

// second Piola-Kirchhoff stress S = C4 : (F^T F - I)/2
template <int D>
__device__ __forceinline__ Mat<D> svk_S(const Mat<D> &F, double K, double mu) {
  Mat<D> E;
  double tr = 0.0;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) s += F.a[k][i] * F.a[k][j];
      E.a[i][j] = 0.5 * (s - (i == j ? 1.0 : 0.0));
    }
#pragma unroll
  for (int i = 0; i < D; ++i) tr += E.a[i][i];
  Mat<D> S;
  const double two_mu = 2.0 * mu;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const double dev = 0.5 * (E.a[i][j] + E.a[j][i]) - (i == j ? (1.0 / 3.0) * tr : 0.0);
      S.a[i][j] = (i == j ? K * tr : 0.0) + two_mu * dev;
    }
  return S;
}

// out = dF.S + F.Y for one grid point
template <int D>
__device__ __forceinline__ Mat<D> svk_tangent(const Mat<D> &f, const Mat<D> &d, double Kp, double mup) {
  const Mat<D> S = svk_S<D>(f, Kp, mup);
  Mat<D> W;
  double tr = 0.0;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) s += f.a[k][i] * d.a[k][j];
      W.a[i][j] = s;
    }
#pragma unroll
  for (int i = 0; i < D; ++i) tr += W.a[i][i];
  Mat<D> Y;
  const double two_mu = 2.0 * mup;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const double dev = 0.5 * (W.a[i][j] + W.a[j][i]) - (i == j ? (1.0 / 3.0) * tr : 0.0);
      Y.a[i][j] = (i == j ? Kp * tr : 0.0) + two_mu * dev;
    }
  Mat<D> o;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) s += d.a[i][k] * S.a[k][j] + f.a[i][k] * Y.a[k][j];
      o.a[i][j] = s;
    }
  return o;
}

}  // namespace mrl
