// Generic any-length Stockham FFT pass sequence along one axis (gfx950).
//
// Correctness backbone: serves every size / layout the reference's FFT service accepts
// (odd and prime lengths, 1-3 D, FULL and HALF spectra, batched value dimensions).  The
// power-of-two fast path (fft_pow2.hip) overrides it for the benchmark sizes.
//
// One workgroup holds `tile` complete lines of length n in LDS (ping-pong buffers) plus the
// twiddle table exp(-+2 pi i k/n) staged into LDS.  Each radix-r Stockham pass is executed
// "one thread per output element": y[o] = sum_t x[j + t n/r] W_n^{t (k n/(Ns r) + u n/r)},
// which needs no register arrays and works for any (also large prime) radix.
#include "mrl_internal.h"

namespace mrl {

__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

__global__ void __launch_bounds__(256) k_fft_generic(PassDesc d, const double *__restrict__ in,
                                                     double *__restrict__ out, const cplx *__restrict__ tw) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx *W = reinterpret_cast<cplx *>(smem);
  cplx *A = W + d.n;
  cplx *B = A + (long long)d.tile * d.n;

  const int tid = threadIdx.x;
  const int nt = blockDim.x;
  const int n = d.n;
  const long long nlines = d.inner * d.outer;
  const long long L0 = (long long)blockIdx.x * d.tile;
  const int nl = (int)min((long long)d.tile, nlines - L0);
  // batch offsets are in elements of the respective kind (real: double, complex: double2)
  const double *inb = in + (long long)blockIdx.y * d.in_sb * (d.in_kind == 1 ? 1 : 2);
  double *outb = out + (long long)blockIdx.y * d.out_sb * (d.out_kind == 2 ? 1 : 2);

  for (int i = tid; i < n; i += nt) {
    cplx w = tw[i];
    if (d.sign > 0) w.y = -w.y;
    W[i] = w;
  }

  const int total = n * nl;
  // ---- load
  for (int e = tid; e < total; e += nt) {
    int l, j;
    if (d.lines_fastest) {
      l = e % nl;
      j = e / nl;
    } else {
      l = e / n;
      j = e % n;
    }
    const long long L = L0 + l;
    const long long o = L / d.inner, i = L % d.inner;
    const long long base = o * d.in_so + i * d.in_si;
    cplx v;
    if (d.in_kind == 1) {
      v = make_double2(inb[base + (long long)j * d.in_sn], 0.0);
    } else if (d.in_kind == 2) {
      // hermitian half line: X[n-k] = conj(X[k]); imaginary parts of X[0] (and Nyquist) are
      // ignored as a c2r transform does
      const int nh = n / 2;
      if (j <= nh) {
        const double2 *p = reinterpret_cast<const double2 *>(inb) + base + (long long)j * d.in_sn;
        v = *p;
        if (j == 0 || (2 * j == n)) v.y = 0.0;
      } else {
        const double2 *p = reinterpret_cast<const double2 *>(inb) + base + (long long)(n - j) * d.in_sn;
        v = *p;
        v.y = -v.y;
      }
    } else {
      const double2 *p = reinterpret_cast<const double2 *>(inb) + base + (long long)j * d.in_sn;
      v = *p;
    }
    A[l * n + j] = v;
  }
  __syncthreads();

  // ---- Stockham passes
  int Ns = 1;
  for (int p = 0; p < d.npass; ++p) {
    const int r = d.radix[p];
    const int m = n / r;            // distance between the r inputs of a butterfly
    const int step1 = n / (Ns * r); // twiddle step of the inter-stage factor
    for (int e = tid; e < total; e += nt) {
      const int l = e / n;
      const int o = e - l * n;
      const int k = o % Ns;
      const int u = (o / Ns) % r;
      const int jhi = o / (Ns * r);
      const int j = jhi * Ns + k;
      const int base = (int)(((long long)k * step1 + (long long)u * m) % n);
      const cplx *x = A + l * n + j;
      cplx acc = x[0];
      int idx = 0;
      for (int t = 1; t < r; ++t) {
        idx += base;
        if (idx >= n) idx -= n;
        const cplx w = W[idx];
        const cplx v = x[t * m];
        acc.x += v.x * w.x - v.y * w.y;
        acc.y += v.x * w.y + v.y * w.x;
      }
      B[l * n + o] = acc;
    }
    __syncthreads();
    cplx *tmp = A;
    A = B;
    B = tmp;
    Ns *= r;
  }

  // ---- store
  const int nout = d.nout;
  const int total_out = nout * nl;
  for (int e = tid; e < total_out; e += nt) {
    int l, j;
    if (d.lines_fastest) {
      l = e % nl;
      j = e / nl;
    } else {
      l = e / nout;
      j = e % nout;
    }
    const long long L = L0 + l;
    const long long o = L / d.inner, i = L % d.inner;
    const long long base = o * d.out_so + i * d.out_si;
    cplx v = A[l * n + j];
    if (d.out_kind == 2) {
      outb[base + (long long)j * d.out_sn] = v.x * d.scale;
    } else {
      v.x *= d.scale;
      v.y *= d.scale;
      double2 *p = reinterpret_cast<double2 *>(outb) + base + (long long)j * d.out_sn;
      *p = v;
    }
  }
}

int launch_pass(mrl_ctx *ctx, const PassDesc &d0, const double *in, double *out, const cplx *d_tw, long long nbatch) {
  PassDesc d = d0;
  const long long nlines = d.inner * d.outer;
  if (nlines <= 0 || nbatch <= 0) return MRL_OK;
  // tile: as many lines as fit a 64 KiB LDS budget (max 16), at least 1
  const size_t per_line = (size_t)d.n * sizeof(cplx) * 2;
  const size_t tw_bytes = (size_t)d.n * sizeof(cplx);
  size_t budget = 64 * 1024;
  int tile = 1;
  if (per_line + tw_bytes > budget) {
    budget = 160 * 1024;
    if (per_line + tw_bytes > budget)
      return set_error(ctx, MRL_ERR_UNSUPPORTED, "FFT length %d exceeds the LDS-resident line limit", d.n);
  } else {
    tile = (int)((budget - tw_bytes) / per_line);
    if (tile > 16) tile = 16;
    // keep enough workgroups in flight
    while (tile > 1 && (nlines + tile - 1) / tile * nbatch < 1024) tile >>= 1;
    if (tile < 1) tile = 1;
  }
  if (d.lines_fastest && d.inner < tile) {
    // a tile must not straddle `outer` when lines are consecutive along the contiguous direction only
    // (it may: lines are addressed individually, so straddling is still correct)
  }
  d.tile = tile;
  const size_t lds = tw_bytes + per_line * tile;
  if (lds > 64 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      MRL_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_fft_generic),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set = true;
    }
  }
  const long long nblocks = (nlines + tile - 1) / tile;
  if (nblocks > 2147483647LL || nbatch > 65535)
    return set_error(ctx, MRL_ERR_UNSUPPORTED, "grid too large for the generic FFT pass");
  dim3 grid((unsigned)nblocks, (unsigned)nbatch);
  hipLaunchKernelGGL(k_fft_generic, grid, dim3(256), lds, ctx->stream, d, in, out, d_tw);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

}  // namespace mrl
