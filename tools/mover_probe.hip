// The ceiling the bandwidth fractions are read against: a TUNED large-array mover (VERDICT r04 "next" 2), replacing the naive
// one-load-in-flight copy of tools/mall_probe.hip.  Every combination of
//   U    = 1, 2, 4, 8 independent 16-byte loads in flight per lane (all issued before the first store),
//   map  = grid-stride (consecutive workgroups touch consecutive 4 KB x U pieces) | one contiguous chunk per workgroup |
//          one contiguous 1/8 of the array per XCD (blockIdx % 8 = XCD), grid-stride inside it,
//   nt   = plain | non-temporal loads + stores,
//   grid = one workgroup per piece (no loop) | 2048 persistent workgroups,
// at 135 MB (one 256^3 half spectrum = the rank-local array of 512^3 / 8), 541 MB and 1 GB per buffer, out of place (a -> b) and in
// place (a -> a, what the serial passes do), the buffer pair rotated through a 6 GB pool so that no launch finds its lines in the
// Infinity Cache.  Also read-only and write-only streams with the best shape.
//   hipcc -O3 --offload-arch=gfx950 tools/mover_probe.hip -o marlin_amd/lib/mover_probe
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

template <bool NT>
__device__ __forceinline__ double2 ld(const double2 *p) {
  if constexpr (NT) {
    double2 v;
    v.x = __builtin_nontemporal_load(&p->x);
    v.y = __builtin_nontemporal_load(&p->y);
    return v;
  } else {
    return *p;
  }
}
template <bool NT>
__device__ __forceinline__ void st(double2 *p, double2 v) {
  if constexpr (NT) {
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
  } else {
    *p = v;
  }
}

// MAP 0: grid-stride.  MAP 1: contiguous chunk per workgroup.  MAP 2: contiguous eighth per XCD, grid-stride inside.
// n is a multiple of 256 * U * gridDim.x in the timed calls (the host rounds the size down), so no tail handling.
template <int U, int MAP, bool NT>
__global__ void __launch_bounds__(256) k_move(const double2 *__restrict__ in, double2 *__restrict__ out, size_t n) {
  const size_t piece = 256 * (size_t)U;
  const size_t npieces = n / piece;
  size_t first, step, count;
  if constexpr (MAP == 0) {
    first = blockIdx.x;
    step = gridDim.x;
    count = (npieces - first + step - 1) / step;
  } else if constexpr (MAP == 1) {
    const size_t per = npieces / gridDim.x;
    first = blockIdx.x * per;
    step = 1;
    count = per;
  } else {
    const size_t xcd = blockIdx.x & 7, loc = blockIdx.x >> 3, gl = gridDim.x >> 3;
    const size_t per = npieces / 8;
    first = xcd * per + loc;
    step = gl;
    count = (per - loc + gl - 1) / gl;
  }
  for (size_t c = 0; c < count; ++c) {
    const size_t base = (first + c * step) * piece + threadIdx.x;
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<NT>(in + base + (size_t)u * 256);
#pragma unroll
    for (int u = 0; u < U; ++u) st<NT>(out + base + (size_t)u * 256, v[u]);
  }
}

template <int U>
__global__ void __launch_bounds__(256) k_read(const double2 *__restrict__ in, double *sink, size_t n) {
  const size_t piece = 256 * (size_t)U, npieces = n / piece;
  double acc = 0;
  for (size_t p = blockIdx.x; p < npieces; p += gridDim.x) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[p * piece + threadIdx.x + (size_t)u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
  }
  if (acc == 1.2345e300) sink[0] = acc;
}
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_write(double2 *__restrict__ out, size_t n, double x) {
  const size_t piece = 256 * (size_t)U, npieces = n / piece;
  for (size_t p = blockIdx.x; p < npieces; p += gridDim.x)
#pragma unroll
    for (int u = 0; u < U; ++u) st<NT>(out + p * piece + threadIdx.x + (size_t)u * 256, make_double2(x, x));
}

struct Pool {
  char *base;
  size_t bytes, cur = 0;
  // a fresh, 2 MB-aligned window of `sz` bytes each call, walking the pool
  double2 *next(size_t sz) {
    sz = (sz + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
    if (cur + sz > bytes) cur = 0;
    char *p = base + cur;
    cur += sz;
    return (double2 *)p;
  }
};

static hipEvent_t e0, e1;

template <int U, int MAP, bool NT>
static double run_move(Pool &pool, size_t bytes, bool inplace, bool persistent) {
  const size_t piece = 256 * (size_t)U;
  size_t n = bytes / 16;
  int grid;
  if (persistent) {
    grid = 2048;
    n = n / (piece * grid) * (piece * grid);
  } else {
    n = n / (piece * 8) * (piece * 8);
    grid = (int)(n / piece);
    if (MAP == 1) grid = (int)(n / piece);  // chunk per workgroup degenerates to one piece per workgroup
  }
  const int reps = 12;
  std::vector<float> t;
  for (int r = 0; r < reps + 3; ++r) {
    double2 *a = pool.next(n * 16), *b = inplace ? a : pool.next(n * 16);
    CK(hipEventRecord(e0));
    k_move<U, MAP, NT><<<grid, 256>>>(a, b, n);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 3) t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  return 2.0 * n * 16 / (t[t.size() / 2] * 1e-3) / 1e9;
}

template <int U, bool NT>
static void row(Pool &pool, size_t bytes, bool inplace) {
  printf("  U=%d %-5s |", U, NT ? "nt" : "plain");
  printf(" %6.0f %6.0f %6.0f |", run_move<U, 0, NT>(pool, bytes, inplace, false), run_move<U, 1, NT>(pool, bytes, inplace, true),
         run_move<U, 2, NT>(pool, bytes, inplace, false));
  printf(" %6.0f %6.0f\n", run_move<U, 0, NT>(pool, bytes, inplace, true), run_move<U, 2, NT>(pool, bytes, inplace, true));
  fflush(stdout);
}

int main() {
  Pool pool;
  pool.bytes = (size_t)6 << 30;
  CK(hipMalloc(&pool.base, pool.bytes));
  double *sink;
  CK(hipMalloc(&sink, 64));
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  k_write<4, false><<<4096, 256>>>((double2 *)pool.base, pool.bytes / 16, 1.0);
  for (int r = 0; r < 200; ++r) k_move<4, 0, false><<<8192, 256>>>((double2 *)pool.base, (double2 *)pool.base + (1 << 26), 1 << 23);  // clocks
  CK(hipDeviceSynchronize());
  const size_t sizes[3] = {(size_t)256 * 256 * 129 * 16, (size_t)4 * 256 * 256 * 129 * 16, (size_t)1 << 30};
  for (int inplace = 0; inplace < 2; ++inplace)
    for (size_t bytes : sizes) {
      printf("\n%s, %.0f MB per buffer: GB/s (read + written bytes), median of 12, buffers rotated through a 6 GB pool\n",
             inplace ? "IN PLACE (a -> a)" : "OUT OF PLACE (a -> b)", bytes / 1e6);
      printf("                | one workgroup per piece:          | 2048 persistent workgroups:\n");
      printf("                | stride  chunk*  xcd-8 | stride  xcd-8      (* chunk per workgroup is always persistent)\n");
      row<1, false>(pool, bytes, inplace);
      row<2, false>(pool, bytes, inplace);
      row<4, false>(pool, bytes, inplace);
      row<8, false>(pool, bytes, inplace);
      row<4, true>(pool, bytes, inplace);
      row<8, true>(pool, bytes, inplace);
    }
  printf("\nread-only / write-only streams (U = 4, grid-stride, 4096 workgroups), GB/s\n");
  for (size_t bytes : sizes) {
    const size_t n = bytes / 16 / (1024 * 4096) * (1024 * 4096);
    float ms[3];
    for (int mode = 0; mode < 3; ++mode) {
      std::vector<float> t;
      for (int r = 0; r < 12; ++r) {
        double2 *a = pool.next(n * 16);
        CK(hipEventRecord(e0));
        if (mode == 0) k_read<4><<<4096, 256>>>(a, sink, n);
        if (mode == 1) k_write<4, false><<<4096, 256>>>(a, n, 2.0);
        if (mode == 2) k_write<4, true><<<4096, 256>>>(a, n, 3.0);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float x;
        CK(hipEventElapsedTime(&x, e0, e1));
        if (r >= 2) t.push_back(x);
      }
      std::sort(t.begin(), t.end());
      ms[mode] = t[t.size() / 2];
    }
    printf("  %6.0f MB: read %6.0f   write %6.0f   write nt %6.0f\n", bytes / 1e6, n * 16 / (ms[0] * 1e-3) / 1e9,
           n * 16 / (ms[1] * 1e-3) / 1e9, n * 16 / (ms[2] * 1e-3) / 1e9);
  }
  return 0;
}
