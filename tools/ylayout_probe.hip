// What would a tile-ordered exchange layout buy the rank-local kernels of 512^3 / 8?  Copy-only movers with the thread ->
// element ownership and the access patterns of the three strided slab kernels, on the real geometry (nx = ny = 512, 64 local planes,
// 257 -> 264 columns), with today's layouts (A) and with exchange buffers + history in the consuming y pass's tile order (B):
//   A  exchange chunk [p][field][ix][jl][264]            history [ix][j][264]
//   B  exchange chunk [p][field][ix][kt][jl][8]          history [ix][kt][j][8]          (kt = kz / 8: 33 tiles)
//   x forward  : 16 columns x 512 x points per workgroup (32 points per thread, 256-byte pieces), work array -> exchange layout
//   y fused    : 8 columns x 512 y points per workgroup (16 points per thread), 2 exchange fields + old history -> new history + exchange
//   x inverse  : 16 columns x 512 x points per workgroup, exchange layout -> work array
// Buffers rotate through a pool so that no launch finds its input in the Infinity Cache (as on a real node, where every input of
// these kernels arrives from a peer).  No transforms: what the memory patterns alone allow.
//   hipcc -O3 --offload-arch=gfx950 tools/ylayout_probe.hip -o marlin_amd/lib/ylayout_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

typedef double2 cplx;
constexpr int NX = 512, NY = 512, P = 8, NXL = NX / P, NYL = NY / P, NZC = 257, KP = 264, NKT = KP / 8;
constexpr long long WP = (long long)NYL * NZC + 16;       // work-array x-plane pitch: 1029 pieces of 256 bytes (odd), as the product pads it
constexpr long long XP = (long long)NYL * KP + 16;        // x-plane pitch inside a chunk, layout A
constexpr long long CHUNK = (long long)NXL * XP;          // one field of one chunk (both layouts: B needs NXL * NKT * NYL * 8 = NXL * NYL * KP <= this)
constexpr long long HIST = (long long)NXL * NY * KP;      // history array

template <bool B>
__device__ __forceinline__ long long xoff(int p, int f, int nf, int ixl, int jl, int col) {  // element of exchange buffer
  const long long base = ((long long)p * nf + f) * CHUNK;
  if (B) return base + ((((long long)ixl * NKT + (col >> 3)) * NYL + jl) << 3) + (col & 7);
  return base + (long long)ixl * XP + (long long)jl * KP + col;
}
template <bool B>
__device__ __forceinline__ long long hoff(int ixl, int j, int col) {
  if (B) return ((((long long)ixl * NKT + (col >> 3)) * NY + j) << 3) + (col & 7);
  return ((long long)ixl * NY + j) * KP + col;
}

// the product's block -> tile map: every XCD (blockIdx % 8) walks a contiguous range of tiles (neighbouring tiles share partial lines
// of the 257-column work-array rows: without it the x movers below run 2 x slower, profiles/r05_ab_xcd_remap_vs_identity.txt)
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nb) {
  const unsigned q8 = nb >> 3, r8 = nb & 7, xcd = b & 7, idx = b >> 3;
  return (xcd < r8) ? xcd * (q8 + 1) + idx : r8 * (q8 + 1) + (xcd - r8) * q8 + idx;
}

// x forward: tiles over the flattened (jl, col) index of the padded rows, 16 consecutive columns per tile; thread (l, q) holds x = q + 16 m
template <bool B>
__global__ void __launch_bounds__(256, 2) k_xfwd(const cplx *__restrict__ w0, const cplx *__restrict__ w1, cplx *__restrict__ dst) {
  const int l = threadIdx.x & 15, q = threadIdx.x >> 4;
  const unsigned nb = NYL * KP / 16;
  const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned f = logical >= nb ? 1u : 0u, t = logical - f * nb;
  const unsigned i = t * 16 + l, row = i / KP, col = i - row * KP;
  const cplx *src = f ? w1 : w0;
  cplx v[32];
#pragma unroll
  for (int m = 0; m < 32; ++m) v[m] = src[(long long)(q + 16 * m) * WP + row * NZC + min(col, (unsigned)NZC - 1u)];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 32; ++m) {
    const int x = q + 16 * m;
    dst[xoff<B>(x >> 6, f, 2, x & 63, row, col)] = v[m];
  }
}

// y fused: tile = (ix, kt); thread (l, q) holds j = q + 32 m
template <bool B>
__global__ void __launch_bounds__(256, 2) k_yfused(const cplx *__restrict__ recv, const cplx *__restrict__ nold, cplx *__restrict__ nnew,
                                                   cplx *__restrict__ send) {
  const int l = threadIdx.x & 7, q = threadIdx.x >> 3;
  const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
  const int ix = logical / NKT, kt = logical - ix * NKT, col = kt * 8 + l;
  cplx a[16], b[16], c[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int j = q + 32 * m;
    a[m] = recv[xoff<B>(j >> 6, 0, 2, ix, j & 63, col)];
    b[m] = recv[xoff<B>(j >> 6, 1, 2, ix, j & 63, col)];
  }
#pragma unroll
  for (int m = 0; m < 16; ++m) c[m] = nold[hoff<B>(ix, q + 32 * m, col)];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int j = q + 32 * m;
    nnew[hoff<B>(ix, j, col)] = make_double2(b[m].x * 2.0, b[m].y * 2.0);
    send[xoff<B>(j >> 6, 0, 1, ix, j & 63, col)] = make_double2(a[m].x + b[m].x * 1.5 - 0.5 * c[m].x, a[m].y + b[m].y * 1.5 - 0.5 * c[m].y);
  }
}

// x inverse: exchange layout -> work array
template <bool B>
__global__ void __launch_bounds__(256, 2) k_xinv(const cplx *__restrict__ recv, cplx *__restrict__ w) {
  const int l = threadIdx.x & 15, q = threadIdx.x >> 4;
  const unsigned i = xcd_remap(blockIdx.x, gridDim.x) * 16 + l, row = i / KP, col = i - row * KP;
  cplx v[32];
#pragma unroll
  for (int m = 0; m < 32; ++m) {
    const int x = q + 16 * m;
    v[m] = recv[xoff<B>(x >> 6, 0, 1, x & 63, row, col)];
  }
  __syncthreads();
  if (col < (unsigned)NZC) {
#pragma unroll
    for (int m = 0; m < 32; ++m) w[(long long)(q + 16 * m) * WP + row * NZC + col] = v[m];
  }
}

__global__ void k_fill(double *p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0 + (double)(i & 1023) * 1e-3;
}

int main() {
  const size_t work = (size_t)NX * WP * 16, ex2 = (size_t)P * 2 * CHUNK * 16, ex1 = (size_t)P * CHUNK * 16, hist = (size_t)HIST * 16;
  const int R = 3;  // copies of every buffer, walked round-robin
  std::vector<char *> W0(R), W1(R), E2(R), E1(R), H0(R), H1(R);
  for (int r = 0; r < R; ++r) {
    CK(hipMalloc(&W0[r], work));
    CK(hipMalloc(&W1[r], work));
    CK(hipMalloc(&E2[r], ex2));
    CK(hipMalloc(&E1[r], ex1));
    CK(hipMalloc(&H0[r], hist));
    CK(hipMalloc(&H1[r], hist));
    for (auto pr : {std::make_pair(W0[r], work), std::make_pair(W1[r], work), std::make_pair(E2[r], ex2), std::make_pair(E1[r], ex1),
                    std::make_pair(H0[r], hist), std::make_pair(H1[r], hist)})
      k_fill<<<2048, 256>>>((double *)pr.first, pr.second / 8);
  }
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto timeit = [&](auto &&launch) {
    std::vector<float> t;
    for (int it = 0; it < 15; ++it) {
      CK(hipEventRecord(e0));
      launch(it % R);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (it >= 3) t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
  };
  const double mb_x2 = 2.0 * 2.0 * NX * NYL * (double)NZC * 16 / 1e6, mb_y = 5.0 * NXL * NY * (double)NZC * 16 / 1e6, mb_x1 = mb_x2 / 2;
  printf("copy-only movers on the slab-local geometry of 512^3 / 8; us per launch (median of 12), TB/s by algorithmic bytes\n");
  for (int rep = 0; rep < 2; ++rep) {
    const float xa = timeit([&](int r) { k_xfwd<false><<<2 * NYL * KP / 16, 256>>>((cplx *)W0[r], (cplx *)W1[r], (cplx *)E2[(r + 1) % R]); });
    const float xb = timeit([&](int r) { k_xfwd<true><<<2 * NYL * KP / 16, 256>>>((cplx *)W0[r], (cplx *)W1[r], (cplx *)E2[(r + 1) % R]); });
    const float ya = timeit([&](int r) { k_yfused<false><<<NXL * NKT, 256>>>((cplx *)E2[r], (cplx *)H0[r], (cplx *)H1[(r + 1) % R], (cplx *)E1[(r + 1) % R]); });
    const float yb = timeit([&](int r) { k_yfused<true><<<NXL * NKT, 256>>>((cplx *)E2[r], (cplx *)H0[r], (cplx *)H1[(r + 1) % R], (cplx *)E1[(r + 1) % R]); });
    const float ia = timeit([&](int r) { k_xinv<false><<<NYL * KP / 16, 256>>>((cplx *)E1[r], (cplx *)W0[(r + 1) % R]); });
    const float ib = timeit([&](int r) { k_xinv<true><<<NYL * KP / 16, 256>>>((cplx *)E1[r], (cplx *)W0[(r + 1) % R]); });
    printf("round %d                today's layouts (A)        tile-ordered exchange + history (B)\n", rep + 1);
    printf("  x forward, 2 fields   %7.1f us  %5.2f TB/s      %7.1f us  %5.2f TB/s\n", xa, mb_x2 / xa, xb, mb_x2 / xb);
    printf("  y fused (5 streams)   %7.1f us  %5.2f TB/s      %7.1f us  %5.2f TB/s\n", ya, mb_y / ya, yb, mb_y / yb);
    printf("  x inverse             %7.1f us  %5.2f TB/s      %7.1f us  %5.2f TB/s\n", ia, mb_x1 / ia, ib, mb_x1 / ib);
    printf("  sum                   %7.1f us                  %7.1f us\n", xa + ya + ia, xb + yb + ib);
  }
  return 0;
}
