// Pencil-decomposed transforms (parallel_mode = FFT_PENCIL): the rank-local stages of
//   DomainAction::fftPencil  (src/actions/DomainAction.C:1021-1034): rfft along x -> stage 1 -> fft along y -> stage 2 -> fft along z
//   DomainAction::ifftPencil (:1036-1047):                           ifft z -> stage 2 -> ifft y -> stage 1 -> irfft along x
// on the partition of DomainAction::partitionPencils (:568-742): rank r = (py, pz) = (r % Py, r / Py) holds the real block
// [nx][ny_py][nz_pz] and the reciprocal block [kx_py][ky_pz][nz] (kx = the nx/2+1 points of the r2c transform along x split over
// Py, ky = ny split over Pz).  The exchanges themselves -- stage 1 inside a group of equal pz, stage 2 inside a group of equal px,
// MPI_Isend / MPI_Recv of host tensors in the reference (:1105-1404) -- belong to slab_driver.hip (library-owned channels).
//
// This is the functional path of the last SURVEY 8(f) item (the reference tests it with 4 ranks on one host,
// test/tests/gradient/tests:21-29), built from the any-length passes: the r2c transform along the STRIDED x axis is a c2c
// transform of the promoted real block of which the first nx/2+1 planes are kept; the c2r transform extends the received half
// spectrum by Hermitian symmetry, transforms c2c and keeps the real part (= irfft: the imaginary parts of the self-conjugate bins
// drop out).  Not tuned: the headline paths are the serial and the slab pipelines.
#include "mrl_internal.h"

namespace mrl {

int pass_lines(mrl_ctx *ctx, int axis, int sign, const double *in, double *out, long long outer, long long inner, long long so,
               long long si, long long sn, int lines_fastest);

__global__ void __launch_bounds__(256) k_pen_promote(const double *__restrict__ in, double2 *__restrict__ out, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = make_double2(in[i], 0.0);
}
// planes x = nxc .. nx-1 of a [nx][plane] array from its first nxc = nx/2+1 planes: X[x] = conj(X[nx - x])
__global__ void __launch_bounds__(256) k_pen_hermitian_x(double2 *__restrict__ w, long long nx, long long nxc, long long plane) {
  const long long total = (nx - nxc) * plane;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long x = nxc + e / plane, i = e % plane;
    const double2 v = w[(nx - x) * plane + i];
    w[x * plane + i] = make_double2(v.x, -v.y);
  }
}
__global__ void __launch_bounds__(256) k_pen_real_scale(const double2 *__restrict__ in, double *__restrict__ out, long long n, double scale) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = in[i].x * scale;
}
// copy a [n0][n1][n2] block (n2 contiguous) between two strided complex arrays
__global__ void __launch_bounds__(256) k_pen_copy3(const double2 *__restrict__ src, double2 *__restrict__ dst, long long n0, long long n1,
                                                   long long n2, long long ss0, long long ss1, long long ds0, long long ds1) {
  const long long total = n0 * n1 * n2;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long c = e % n2, r = e / n2, b = r % n1, a = r / n1;
    dst[a * ds0 + b * ds1 + c] = src[a * ss0 + b * ss1 + c];
  }
}

static unsigned blocks_for(long long n) {
  long long nb = (n + 255) / 256;
  return (unsigned)(nb < 1 ? 1 : (nb > 8192 ? 8192 : nb));
}
static int copy3(mrl_ctx *ctx, const double *src, double *dst, long long n0, long long n1, long long n2, long long ss0, long long ss1,
                 long long ds0, long long ds1) {
  if (n0 * n1 * n2 == 0) return MRL_OK;
  hipLaunchKernelGGL(k_pen_copy3, dim3(blocks_for(n0 * n1 * n2)), dim3(256), 0, ctx->stream, reinterpret_cast<const double2 *>(src),
                     reinterpret_cast<double2 *>(dst), n0, n1, n2, ss0, ss1, ds0, ds1);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

static long long begin_of(const std::vector<long long> &v, int i) {
  long long b = 0;
  for (int r = 0; r < i; ++r) b += v[r];
  return b;
}

// Message sizes (complex elements) of the four staged exchanges, per peer rank (zero outside the group):
//   stage 1: the Py ranks of equal pz      forward: my kx chunk for px' <-> their y blocks          (:1105-1180)  inverse (:1331-1404)
//   stage 2: the Pz ranks of equal px      forward: my ky chunk for pz' <-> their z blocks          (:1182-1256)  inverse (:1258-1329)
int pencil_counts(const mrl_ctx *ctx, int stage, int forward, long long *send, long long *recv) {
  // ONE implementation of the message sizes: the host-only ABI entry point (context.hip), which the CPU tests drive with real
  // gloo messages (tests/test_pencil_gloo.py); the inverse stages swap send and receive
  const int64_t n[3] = {ctx->n[0], ctx->n[1], ctx->n[2]};
  std::vector<int64_t> a(ctx->nranks), b(ctx->nranks);
  int rc;
  if (stage == 1)
    rc = mrl_pencil_layout(ctx->nranks, ctx->rank, n, nullptr, nullptr, nullptr, nullptr, a.data(), b.data(), nullptr, nullptr);
  else
    rc = mrl_pencil_layout(ctx->nranks, ctx->rank, n, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, a.data(), b.data());
  if (rc != MRL_OK) return set_error(ctx, rc, "%s", mrl_last_error(nullptr));
  for (int p = 0; p < ctx->nranks; ++p) {
    send[p] = forward ? a[p] : b[p];
    recv[p] = forward ? b[p] : a[p];
  }
  return MRL_OK;
}

// work slots: 11 = the promoted real block / x-transformed block [nx][nyl][nzl]; 13 = [kxl][ny][nzl]; 14 = [kxl][kyl][nz]
static int pen_work(mrl_ctx *ctx) {
  const long long nx = ctx->n[0], ny = ctx->n[1], nz = ctx->n[2];
  MRL_TRY(ensure_work(ctx, 11, sizeof(cplx) * (size_t)(nx * ctx->nloc[1] * ctx->nloc[2])));
  MRL_TRY(ensure_work(ctx, 13, sizeof(cplx) * (size_t)(ctx->nrec[0] * ny * ctx->nloc[2])));
  MRL_TRY(ensure_work(ctx, 14, sizeof(cplx) * (size_t)(ctx->nrec[0] * ctx->nrec[1] * nz)));
  return MRL_OK;
}

// ---- forward ------------------------------------------------------------------------------------------------------------------
// real block -> send buffer of stage 1: [px'][kx chunk px'][nyl][nzl] = the first nx/2+1 x planes of the transformed block
int pencil_fwd_x(mrl_ctx *ctx, const double *real_in, double *send1) {
  const long long nx = ctx->n[0], nxc = ctx->nrec_glob[0], plane = ctx->nloc[1] * ctx->nloc[2];
  MRL_TRY(pen_work(ctx));
  double *w = ctx->d_work[11];
  ProfScope ps(ctx, "pencil_x_fwd", 8.0 * nx * plane + 48.0 * nx * plane + 32.0 * nxc * plane);
  hipLaunchKernelGGL(k_pen_promote, dim3(blocks_for(nx * plane)), dim3(256), 0, ctx->stream, real_in, reinterpret_cast<double2 *>(w), nx * plane);
  MRL_HIP(ctx, hipGetLastError());
  MRL_TRY(pass_lines(ctx, 0, -1, w, w, 1, plane, 0, 1, plane, 1));
  MRL_HIP(ctx, hipMemcpyAsync(send1, w, sizeof(cplx) * (size_t)(nxc * plane), hipMemcpyDeviceToDevice, ctx->stream));
  return MRL_OK;
}

// receive buffer of stage 1 [py'][kxl][ny_py'][nzl] -> [kxl][ny][nzl], transform along y, -> send buffer of stage 2
// [pz'][kxl][ky chunk pz'][nzl]
int pencil_fwd_y(mrl_ctx *ctx, const double *recv1, double *send2) {
  const long long ny = ctx->n[1], nzl = ctx->nloc[2], kxl = ctx->nrec[0];
  MRL_TRY(pen_work(ctx));
  double *a1 = ctx->d_work[13];
  ProfScope ps(ctx, "pencil_y_fwd", 96.0 * kxl * ny * nzl);
  long long off = 0;
  for (int q = 0; q < ctx->pen_py; ++q) {
    const long long nyq = ctx->pen_y[q], yb = begin_of(ctx->pen_y, q);
    MRL_TRY(copy3(ctx, recv1 + 2 * off, a1 + 2 * (yb * nzl), kxl, nyq, nzl, nyq * nzl, nzl, ny * nzl, nzl));
    off += kxl * nyq * nzl;
  }
  MRL_TRY(pass_lines(ctx, 1, -1, a1, a1, kxl, nzl, ny * nzl, 1, nzl, 1));
  off = 0;
  for (int q = 0; q < ctx->pen_pz; ++q) {
    const long long kyq = ctx->pen_ky[q], kb = begin_of(ctx->pen_ky, q);
    MRL_TRY(copy3(ctx, a1 + 2 * (kb * nzl), send2 + 2 * off, kxl, kyq, nzl, ny * nzl, nzl, kyq * nzl, nzl));
    off += kxl * kyq * nzl;
  }
  return MRL_OK;
}

// receive buffer of stage 2 [pz'][kxl][kyl][nz_pz'] -> spectrum [kxl][kyl][nz], transform along z
int pencil_fwd_z(mrl_ctx *ctx, const double *recv2, double *spec_out) {
  const long long nz = ctx->n[2], kxl = ctx->nrec[0], kyl = ctx->nrec[1];
  ProfScope ps(ctx, "pencil_z_fwd", 64.0 * kxl * kyl * nz);
  long long off = 0;
  for (int q = 0; q < ctx->pen_pz; ++q) {
    const long long nzq = ctx->pen_z[q], zb = begin_of(ctx->pen_z, q);
    MRL_TRY(copy3(ctx, recv2 + 2 * off, spec_out + 2 * zb, kxl, kyl, nzq, kyl * nzq, nzq, kyl * nz, nz));
    off += kxl * kyl * nzq;
  }
  return pass_lines(ctx, 2, -1, spec_out, spec_out, kxl * kyl, 1, nz, 1, 1, 0);
}

// ---- inverse ------------------------------------------------------------------------------------------------------------------
// spectrum [kxl][kyl][nz] -> inverse transform along z (unnormalised) -> send buffer of stage 2: [pz'][kxl][kyl][nz_pz']
int pencil_inv_z(mrl_ctx *ctx, const double *spec_in, double *send2) {
  const long long nz = ctx->n[2], kxl = ctx->nrec[0], kyl = ctx->nrec[1];
  MRL_TRY(pen_work(ctx));
  double *a2 = ctx->d_work[14];
  ProfScope ps(ctx, "pencil_z_inv", 64.0 * kxl * kyl * nz);
  MRL_TRY(pass_lines(ctx, 2, +1, spec_in, a2, kxl * kyl, 1, nz, 1, 1, 0));
  long long off = 0;
  for (int q = 0; q < ctx->pen_pz; ++q) {
    const long long nzq = ctx->pen_z[q], zb = begin_of(ctx->pen_z, q);
    MRL_TRY(copy3(ctx, a2 + 2 * zb, send2 + 2 * off, kxl, kyl, nzq, kyl * nz, nz, kyl * nzq, nzq));
    off += kxl * kyl * nzq;
  }
  return MRL_OK;
}

// receive buffer of stage 2 [pz'][kxl][ky chunk pz'][nzl] -> [kxl][ny][nzl], inverse transform along y, -> send buffer of stage 1:
// [py'][kxl][ny_py'][nzl]
int pencil_inv_y(mrl_ctx *ctx, const double *recv2, double *send1) {
  const long long ny = ctx->n[1], nzl = ctx->nloc[2], kxl = ctx->nrec[0];
  MRL_TRY(pen_work(ctx));
  double *a1 = ctx->d_work[13];
  ProfScope ps(ctx, "pencil_y_inv", 96.0 * kxl * ny * nzl);
  long long off = 0;
  for (int q = 0; q < ctx->pen_pz; ++q) {
    const long long kyq = ctx->pen_ky[q], kb = begin_of(ctx->pen_ky, q);
    MRL_TRY(copy3(ctx, recv2 + 2 * off, a1 + 2 * (kb * nzl), kxl, kyq, nzl, kyq * nzl, nzl, ny * nzl, nzl));
    off += kxl * kyq * nzl;
  }
  MRL_TRY(pass_lines(ctx, 1, +1, a1, a1, kxl, nzl, ny * nzl, 1, nzl, 1));
  off = 0;
  for (int q = 0; q < ctx->pen_py; ++q) {
    const long long nyq = ctx->pen_y[q], yb = begin_of(ctx->pen_y, q);
    MRL_TRY(copy3(ctx, a1 + 2 * (yb * nzl), send1 + 2 * off, kxl, nyq, nzl, ny * nzl, nzl, nyq * nzl, nzl));
    off += kxl * nyq * nzl;
  }
  return MRL_OK;
}

// receive buffer of stage 1 [px'][kx chunk px'][nyl][nzl] = the half spectrum [nx/2+1][nyl][nzl] along x -> real block, * 1/N
int pencil_inv_x(mrl_ctx *ctx, const double *recv1, double *real_out) {
  const long long nx = ctx->n[0], nxc = ctx->nrec_glob[0], plane = ctx->nloc[1] * ctx->nloc[2];
  MRL_TRY(pen_work(ctx));
  double *w = ctx->d_work[11];
  ProfScope ps(ctx, "pencil_x_inv", 32.0 * nxc * plane + 48.0 * nx * plane + 8.0 * nx * plane);
  MRL_HIP(ctx, hipMemcpyAsync(w, recv1, sizeof(cplx) * (size_t)(nxc * plane), hipMemcpyDeviceToDevice, ctx->stream));
  if (nx > nxc) {
    hipLaunchKernelGGL(k_pen_hermitian_x, dim3(blocks_for((nx - nxc) * plane)), dim3(256), 0, ctx->stream, reinterpret_cast<double2 *>(w), nx, nxc, plane);
    MRL_HIP(ctx, hipGetLastError());
  }
  MRL_TRY(pass_lines(ctx, 0, +1, w, w, 1, plane, 0, 1, plane, 1));
  const double scale = 1.0 / ((double)ctx->n[0] * (double)ctx->n[1] * (double)ctx->n[2]);
  hipLaunchKernelGGL(k_pen_real_scale, dim3(blocks_for(nx * plane)), dim3(256), 0, ctx->stream, reinterpret_cast<const double2 *>(w), real_out,
                     nx * plane, scale);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

}  // namespace mrl
