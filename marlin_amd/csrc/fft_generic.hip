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
#include <atomic>

namespace mrl {

__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// a / d and a % d for 0 <= a < 2^24, d >= 1 with a precomputed float reciprocal: the quotient estimate is off by at most one.
// (hipcc expands a run-time integer division into ~30 instructions, a 64-bit modulo into a loop; this kernel did four of the
// former and one of the latter per element and pass, which is what bounded it.)
__device__ __forceinline__ int fast_divmod(int a, int d, float inv, int &rem) {
  int q = (int)((float)a * inv);
  int r = a - q * d;
  if (r < 0) {
    --q;
    r += d;
  } else if (r >= d) {
    ++q;
    r -= d;
  }
  rem = r;
  return q;
}

// One radix-R Stockham pass (R = 2, 3, 4, 5) over the `nl` lines of the tile, one BUTTERFLY per thread: the R inputs
// x[b + t n/R] are read once, twiddled by W^(t k step1), combined by a length-R DFT in registers and written to their R output
// positions -- one LDS read, one twiddle read and one write per point, where the one-output-per-thread form below reads R inputs
// and R-1 twiddles per point.  W already holds the conjugated table for the inverse transform.
template <int R>
__device__ __forceinline__ void pass_small_radix(const cplx *__restrict__ A, cplx *__restrict__ B, const cplx *__restrict__ W,
                                                 int n, int nl, int Ns, int sign, int tid, int nt) {
  const int m = n / R, step1 = n / (Ns * R);
  const float inv_m = 1.0f / (float)m, inv_Ns = 1.0f / (float)Ns;
  cplx wr[R];  // wr[j] = exp(-+ 2 pi i j / R)
#pragma unroll
  for (int j = 1; j < R; ++j) wr[j] = W[j * m];
  const int nb = m * nl;
  for (int e = tid; e < nb; e += nt) {
    int b, k;
    const int l = fast_divmod(e, m, inv_m, b);
    const int jhi = fast_divmod(b, Ns, inv_Ns, k);
    const cplx *x = A + l * n + b;
    cplx a[R];
#pragma unroll
    for (int t = 0; t < R; ++t) a[t] = x[t * m];
    const int st = k * step1;  // t * st < n for every t < R
#pragma unroll
    for (int t = 1; t < R; ++t) a[t] = cmul(a[t], W[t * st]);
    cplx *y = B + l * n + jhi * Ns * R + k;
    if (R == 2) {
      y[0] = make_double2(a[0].x + a[1].x, a[0].y + a[1].y);
      y[Ns] = make_double2(a[0].x - a[1].x, a[0].y - a[1].y);
    } else if (R == 4) {
      const cplx s02 = make_double2(a[0].x + a[2].x, a[0].y + a[2].y), d02 = make_double2(a[0].x - a[2].x, a[0].y - a[2].y);
      const cplx s13 = make_double2(a[1].x + a[3].x, a[1].y + a[3].y), d13 = make_double2(a[1].x - a[3].x, a[1].y - a[3].y);
      // -i d13 (forward) or +i d13 (inverse)
      const cplx r13 = sign < 0 ? make_double2(d13.y, -d13.x) : make_double2(-d13.y, d13.x);
      y[0] = make_double2(s02.x + s13.x, s02.y + s13.y);
      y[Ns] = make_double2(d02.x + r13.x, d02.y + r13.y);
      y[2 * Ns] = make_double2(s02.x - s13.x, s02.y - s13.y);
      y[3 * Ns] = make_double2(d02.x - r13.x, d02.y - r13.y);
    } else {
#pragma unroll
      for (int u = 0; u < R; ++u) {
        cplx acc = a[0];
#pragma unroll
        for (int t = 1; t < R; ++t) {
          if (u == 0) {
            acc.x += a[t].x;
            acc.y += a[t].y;
          } else {
            const cplx w = wr[(t * u) % R];
            acc.x += a[t].x * w.x - a[t].y * w.y;
            acc.y += a[t].x * w.y + a[t].y * w.x;
          }
        }
        y[u * Ns] = acc;
      }
    }
  }
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
  const long long nlines = d.pair ? (d.nreal_lines + 1) / 2 : d.inner * d.outer;
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

  const long long o0 = L0 / d.inner, i0 = L0 % d.inner;
  const int total = n * nl;
  const float inv_n = 1.0f / (float)n, inv_nl = 1.0f / (float)nl;
  // ---- load
  for (int e = tid; e < total; e += nt) {
    int l, j;
    if (d.lines_fastest) {
      j = fast_divmod(e, nl, inv_nl, l);
    } else {
      l = fast_divmod(e, n, inv_n, j);
    }
    if (d.pair) {
      // two real lines (forward) or two hermitian half lines (inverse) in one complex transform: z = a + i b
      const long long La = 2 * (L0 + l), Lb = La + 1;
      const bool hb = Lb < d.nreal_lines;
      const long long ba = La * d.in_so, bb = Lb * d.in_so;
      cplx v;
      if (d.in_kind == 1) {
        v = make_double2(inb[ba + (long long)j * d.in_sn], hb ? inb[bb + (long long)j * d.in_sn] : 0.0);
      } else {
        const int nh = n / 2;
        const bool lo = j <= nh;
        const int k = lo ? j : n - j;
        const double2 *pa = reinterpret_cast<const double2 *>(inb) + ba + (long long)k * d.in_sn;
        const double2 *pb = reinterpret_cast<const double2 *>(inb) + bb + (long long)k * d.in_sn;
        cplx a = *pa, b = hb ? *pb : make_double2(0.0, 0.0);
        if (k == 0 || 2 * k == n) {  // a c2r transform ignores the imaginary parts of the self-conjugate bins
          a.y = 0.0;
          b.y = 0.0;
        }
        v = lo ? make_double2(a.x - b.y, a.y + b.x) : make_double2(a.x + b.y, b.x - a.y);
      }
      A[l * n + j] = v;
      continue;
    }
    // line L0 + l -> (outer, inner) index from the block's first line (one 64-bit division per workgroup instead of two per element)
    long long o = o0, i = i0 + l;
    if (d.inner == 1) {
      o += l;
      i = 0;
    } else {
      while (i >= d.inner) {
        i -= d.inner;
        ++o;
      }
    }
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
    if (r <= 5) {
      switch (r) {
        case 2: pass_small_radix<2>(A, B, W, n, nl, Ns, d.sign, tid, nt); break;
        case 3: pass_small_radix<3>(A, B, W, n, nl, Ns, d.sign, tid, nt); break;
        case 4: pass_small_radix<4>(A, B, W, n, nl, Ns, d.sign, tid, nt); break;
        default: pass_small_radix<5>(A, B, W, n, nl, Ns, d.sign, tid, nt); break;
      }
      __syncthreads();
      cplx *tmp = A;
      A = B;
      B = tmp;
      Ns *= r;
      continue;
    }
    const int m = n / r;            // distance between the r inputs of a butterfly
    const int step1 = n / (Ns * r); // twiddle step of the inter-stage factor
    const float inv_Ns = 1.0f / (float)Ns, inv_r = 1.0f / (float)r;
    for (int e = tid; e < total; e += nt) {
      int o, k, u;
      const int l = fast_divmod(e, n, inv_n, o);
      const int qn = fast_divmod(o, Ns, inv_Ns, k);
      const int jhi = fast_divmod(qn, r, inv_r, u);
      const int j = jhi * Ns + k;
      int base = k * step1 + u * m;   // k step1 < n / r and u m < n: one conditional subtraction reduces it mod n
      if (base >= n) base -= n;
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
  const float inv_nout = 1.0f / (float)nout;
  for (int e = tid; e < total_out; e += nt) {
    int l, j;
    if (d.lines_fastest) {
      j = fast_divmod(e, nl, inv_nl, l);
    } else {
      l = fast_divmod(e, nout, inv_nout, j);
    }
    if (d.pair) {
      const long long La = 2 * (L0 + l), Lb = La + 1;
      const bool hb = Lb < d.nreal_lines;
      const long long ba = La * d.out_so, bb = Lb * d.out_so;
      const cplx z = A[l * n + j];
      if (d.out_kind == 2) {  // inverse: the two real lines are the real and the imaginary part
        outb[ba + (long long)j * d.out_sn] = z.x * d.scale;
        if (hb) outb[bb + (long long)j * d.out_sn] = z.y * d.scale;
      } else {                // forward: X_a[k] = (Z[k] + conj Z[n-k]) / 2, X_b[k] = (Z[k] - conj Z[n-k]) / (2i)
        const cplx zc = A[l * n + (j == 0 ? 0 : n - j)];
        double2 *pa = reinterpret_cast<double2 *>(outb) + ba + (long long)j * d.out_sn;
        *pa = make_double2(0.5 * (z.x + zc.x) * d.scale, 0.5 * (z.y - zc.y) * d.scale);
        if (hb) {
          double2 *pb = reinterpret_cast<double2 *>(outb) + bb + (long long)j * d.out_sn;
          *pb = make_double2(0.5 * (z.y + zc.y) * d.scale, -0.5 * (z.x - zc.x) * d.scale);
        }
      }
      continue;
    }
    // line L0 + l -> (outer, inner) index from the block's first line (one 64-bit division per workgroup instead of two per element)
    long long o = o0, i = i0 + l;
    if (d.inner == 1) {
      o += l;
      i = 0;
    } else {
      while (i >= d.inner) {
        i -= d.inner;
        ++o;
      }
    }
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
  long long nlines = d.inner * d.outer;
  if (nlines <= 0 || nbatch <= 0) return MRL_OK;
  // real <-> half-spectrum passes over contiguous line sets: two lines per complex transform (half the LDS passes)
  d.pair = (d.inner == 1 && d.in_si == 0 && d.out_si == 0 && ((d.in_kind == 1 && d.out_kind == 0) || (d.in_kind == 2 && d.out_kind == 2))) ? 1 : 0;
  d.nreal_lines = nlines;
  if (d.pair) nlines = (nlines + 1) / 2;
  // tile: as many lines as fit a 32 KiB LDS budget (max 16), at least 1.  The budget sets the occupancy: a workgroup alternates
  // between a load, several LDS-only passes and a store, so HBM is only busy if other workgroups of the CU are in another phase
  // (swept 20 ... 64 KiB over 120^3 ... 405^2 x 40: 32 KiB is best or within 10 % of it; 240^3 substep 2.19 -> 1.38 ms vs 64 KiB)
  const size_t per_line = (size_t)d.n * sizeof(cplx) * 2;
  const size_t tw_bytes = (size_t)d.n * sizeof(cplx);
  size_t budget = 32 * 1024;
  int tile = 1;
  if (per_line + tw_bytes > budget) {
    budget = 160 * 1024;
    if (per_line + tw_bytes > budget)
      return set_error(ctx, MRL_ERR_UNSUPPORTED, "FFT length %d exceeds the LDS-resident line limit", d.n);
  } else {
    if (d.lines_fastest) budget = 40 * 1024;  // strided passes gather tile x 16 B per row: a little more room buys whole 64-byte sectors
    tile = (int)((budget - tw_bytes) / per_line);
    if (tile > 16) tile = 16;
    if (d.lines_fastest && tile >= 4) tile &= ~3;  // 4, 8, 12 or 16 lines = 64 ... 256 contiguous bytes per row
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
    static std::atomic<bool> attr_set{false};
    if (!attr_set.load(std::memory_order_acquire)) {
      MRL_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_fft_generic),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set.store(true, std::memory_order_release);
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
