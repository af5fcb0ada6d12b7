// Where does the fused Cahn-Hilliard k-space pass spend its time at 512 points?  The product body (ch_fused_body.h) on the product
// geometry of the serial x pass, with ablations:
//   full          the kernel as it is
//   compute only  the same body reading a 64 KB window (L2 hits) and storing nothing: what transforms + LDS + pointwise work cost
//   memory only   built with -DPROBE_NOFFT: the three transforms are skipped, loads and stores as in the product
// usage: xfused_probe [lines_of_plane]      (plane = inner elements per x plane; default 64 * 257 = the slab-local 512^3 / 8 size)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Imarlin_amd/csrc tools/xfused_probe.hip -o marlin_amd/lib/xfused_probe
//   hipcc ... -DPROBE_NOFFT ... -o marlin_amd/lib/xfused_probe_nofft
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fft_pow2.h"
#ifdef PROBE_NOFFT
namespace mrl {
namespace p2 {
template <int N, class Map>
__device__ __forceinline__ void fft_line_skip(kcplx (&v)[Plan<N>::P], int q, int l, kcplx *X, const kcplx *W) {
  // keep the LDS staging of W / KL visible to the workgroup (the first exchange of the real transform does that)
  __syncthreads();
}
}  // namespace p2
}  // namespace mrl
#define fft_line fft_line_skip
#endif
#include "ch_fused_body.h"

using namespace mrl;
using namespace mrl::p2;

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

struct Args {
  FusedCommon c;
  long long inner, plane;
  int nzc;
  const kreal *kx, *ky, *kz;
};

template <int N, int ORDER, int PRE, bool NT, int ABL, int WPS = 2>
__global__ void __launch_bounds__(Plan<N>::NT, WPS) k_xf(Args a, const kcplx *__restrict__ tw) {
  constexpr int TPL = Plan<N>::TPL, T = Plan<N>::T;
  using Map = MapStrided<N>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  kcplx *W = reinterpret_cast<kcplx *>(smem);
  kcplx *X = W + N;
  kreal *KX = reinterpret_cast<kreal *>(X + Map::size);
  const int l = threadIdx.x % T, q = threadIdx.x / T;
  const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
  const long long i = (long long)logical * T + l;
  const bool valid = ABL == 2 ? false : i < a.inner;
  const long long iv = (i < a.inner) ? i : 0;
  kcplx *const ubar = a.c.ubar;
  if (ABL == 2) {
    // a 64 KB window shared by every workgroup: the loads hit L2 / L1
    const unsigned off0 = (unsigned)(threadIdx.x * 16u), step = 4096u;
    auto off = [=](int m) { return off0 + (unsigned)m * step; };
    auto stu = [=](int m, kcplx val) { stc(ubar, off0 + (unsigned)m * step, val); };
    ch_fused_body<N, ORDER, true, PRE, false, false, false, false>(a.c, tw, a.kx, a.ky, a.kz, valid, q, l, off, OffSame{}, off, stu, W, X, KX);
  } else {
    const unsigned off0 = (unsigned)(iv + (long long)q * a.plane) * (unsigned)sizeof(kcplx), step = (unsigned)(TPL * a.plane) * (unsigned)sizeof(kcplx);
    auto off = [=](int m) { return off0 + (unsigned)m * step; };
    auto stu = [=](int m, kcplx val) { stc(ubar, off0 + (unsigned)m * step, val); };
    ch_fused_body<N, ORDER, true, PRE, false, NT, NT, NT>(a.c, tw, a.kx, a.ky + iv / a.nzc, a.kz + iv % a.nzc, valid, q, l, off, OffSame{}, off, stu, W, X, KX);
  }
}

__global__ void __launch_bounds__(256) k_touch(double2 *p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    double2 v = p[i];
    v.x += 1e-300;
    p[i] = v;
  }
}
__global__ void k_fill(double2 *p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_double2(0.5 + 1e-3 * (double)(i % 977), 0.1);
}

template <int N, int ORDER, int PRE, bool NT, int ABL, int WPS = 2>
static void run(const char *name, long long inner, int nzc) {
  using Map = MapStrided<N>;
  constexpr int T = Plan<N>::T;
  long long plane = (inner + 15) / 16 * 16;
  if (((plane / 16) & 1) == 0) plane += 16;
  const size_t elems = (size_t)N * plane + 64;
  double2 *buf[5];  // chat, muhat, Nnew, Nold, (ubar = chat)
  for (auto &b : buf) {
    CK(hipMalloc(&b, elems * sizeof(double2)));
    k_fill<<<2048, 256>>>(b, elems);
  }
  std::vector<double2> tw(N);
  for (int k = 0; k < N; ++k) {
    const long double ang = -2.0L * M_PIl * k / N;
    tw[k] = make_double2((double)cosl(ang), (double)sinl(ang));
  }
  double2 *d_tw;
  double *d_k;
  CK(hipMalloc(&d_tw, N * sizeof(double2)));
  CK(hipMemcpy(d_tw, tw.data(), N * sizeof(double2), hipMemcpyHostToDevice));
  std::vector<double> k(1024);
  for (int i = 0; i < 1024; ++i) k[i] = 0.05 * (i % 512 < 256 ? i % 512 : i % 512 - 512);
  CK(hipMalloc(&d_k, 1024 * sizeof(double)));
  CK(hipMemcpy(d_k, k.data(), 1024 * sizeof(double), hipMemcpyHostToDevice));
  Args a{};
  a.c.chat = buf[0];
  a.c.muhat = buf[1];
  a.c.ubar = buf[0];
  a.c.Nnew = buf[2];
  a.c.Nold[0] = buf[3];
  a.c.coef[0] = 1.5e-3;
  a.c.coef[1] = -0.5e-3;
  a.c.M = 0.2;
  a.c.kappa = -0.001;
  a.c.dt = 1e-3;
  a.inner = inner;
  a.plane = plane;
  a.nzc = nzc;
  a.kx = d_k;
  a.ky = d_k;
  a.kz = d_k;
  const size_t lds = sizeof(kcplx) * (N + Map::size) + sizeof(kreal) * N;
  auto K = k_xf<N, ORDER, PRE, NT, ABL, WPS>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, K, Plan<N>::NT, lds));
  const unsigned nb = (unsigned)((inner + T - 1) / T);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float tot = 0.f, best = 1e9f;
  const int reps = 16;
  for (int r = -3; r < reps; ++r) {
    // what precedes the pass in the substep: the forward y pass has just rewritten both work arrays
    k_touch<<<4096, 256>>>(buf[0], elems);
    k_touch<<<4096, 256>>>(buf[1], elems);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(K, dim3(nb), dim3(Plan<N>::NT), lds, 0, a, d_tw);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 0) {
      tot += ms;
      best = ms < best ? ms : best;
    }
  }
  CK(hipGetLastError());
  const double bytes = (4.0 + ORDER) * 16.0 * (double)N * (double)inner;
  const double us = tot / reps * 1e3;
  printf("%-40s N=%d order=%d nt=%d abl=%d  blocks/CU %d lds %zu  avg %8.1f us  best %8.1f us  %6.0f GB/s\n", name, N, ORDER, (int)NT, ABL, occ, lds, us,
         best * 1e3, bytes / us * 1e-3);
  fflush(stdout);
  for (auto &b : buf) CK(hipFree(b));
  CK(hipFree(d_tw));
  CK(hipFree(d_k));
}

int main(int argc, char **argv) {
  const long long rows = argc > 1 ? atoll(argv[1]) : 64;
  {  // clock warm-up
    double2 *w;
    const size_t n = 16u << 20;
    CK(hipMalloc(&w, n * sizeof(double2)));
    for (int r = 0; r < 1500; ++r) k_touch<<<4096, 256>>>(w, n);
    CK(hipDeviceSynchronize());
    CK(hipFree(w));
  }
#ifdef PROBE_NOFFT
  printf("== memory only (transforms skipped), %lld x 257 lines per x plane\n", rows);
  run<512, 1, 4, false, 0>("512, default cache policy", rows * 257, 257);
  run<512, 1, 4, true, 0>("512, streaming history", rows * 257, 257);
  run<256, 1, 8, true, 0>("256 (256 x 129 plane)", 256 * 129, 129);
#else
  printf("== %lld x 257 lines per x plane\n", rows);
  run<512, 1, 4, false, 0>("512 product, default cache policy", rows * 257, 257);
  run<512, 1, 4, true, 0>("512 product, streaming history", rows * 257, 257);
  run<512, 1, 4, false, 2>("512 compute only", rows * 257, 257);
  run<512, 1, 8, false, 0>("512 PRE 8", rows * 257, 257);
  run<512, 1, 2, false, 0>("512 PRE 2", rows * 257, 257);
  run<256, 1, 8, true, 0>("256 product (256 x 129 plane)", 256 * 129, 129);
  run<256, 1, 8, false, 2>("256 compute only", 256 * 129, 129);
#endif
  return 0;
}
