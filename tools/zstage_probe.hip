// Stage-by-stage check of fft_line<N, Map> (fft_pow2.h) on one workgroup against a host emulation of the same Stockham plan: how the
// 240-point miscompile of round 5 (stage(): `b % NS`) was found.  hipcc -O3 --offload-arch=gfx950 -Iinclude -Imarlin_amd/csrc tools/zstage_probe.hip -o marlin_amd/lib/zstage_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <complex>
#include <cstdio>
#include <vector>
#include "../marlin_amd/csrc/mrl_internal.h"
#include "../marlin_amd/csrc/fft_pow2_kernels.h"
using namespace mrl;
using namespace mrl::p2;
typedef std::complex<double> cd;
template <int N, class Map, int STOP>
__global__ void __launch_bounds__(ZPlan<N>::NT, 2) k_probe(const kcplx *in, kcplx *out, const kcplx *tw) {
  constexpr int P = Plan<N>::P, TPL = Plan<N>::TPL;
  using Pl = Plan<N>;
  constexpr bool ST = Map::staged_tw;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  kcplx *W = reinterpret_cast<kcplx *>(smem);
  kcplx *X = W + N;
  const int q = threadIdx.x % TPL, l = threadIdx.x / TPL;
  TwRegs<N, ZPlan<N>::NT> twr;
  if (ST) tw_issue_staged<N>(twr, tw); else tw_issue<N>(twr, tw);
  kcplx v[P];
  for (int m = 0; m < P; ++m) v[m] = in[l * N + q + m * TPL];
  tw_commit<N>(twr, W);
  stage<N, Pl::r0, 1, ST>(v, q, W);
  exchange<N, Pl::r0, 1, Map>(v, q, l, X);
  if (STOP >= 2) stage<N, Pl::r1, Pl::r0, ST>(v, q, W);
  if (STOP >= 3) {
    exchange<N, Pl::r1, Pl::r0, Map>(v, q, l, X);
    stage<N, Pl::r2, Pl::r0 * Pl::r1, ST>(v, q, W);
  }
  if (STOP >= 4) {
    exchange<N, Pl::r2, Pl::r0 * Pl::r1, Map>(v, q, l, X);
    stage<N, Pl::r3, Pl::r0 * Pl::r1 * Pl::r2, ST>(v, q, W);
  }
  for (int m = 0; m < P; ++m) out[l * N + q + m * TPL] = v[m];
}
template <int N>
void emulate(std::vector<cd> &v, int stop) {  // v indexed [q + m TPL] as the registers are
  constexpr int P = Plan<N>::P, TPL = Plan<N>::TPL;
  const int rad[4] = {Plan<N>::r0, Plan<N>::r1, Plan<N>::r2, Plan<N>::r3};
  int NS = 1;
  for (int s = 0; s < stop; ++s) {
    const int R = rad[s], S = P / R;
    if (s > 0) {  // exchange of the stage before
      const int Rp = rad[s - 1], Sp = P / Rp, NSp = NS / Rp;
      std::vector<cd> X(N);
      for (int q = 0; q < TPL; ++q)
        for (int i = 0; i < Sp; ++i) {
          const int b = q + i * TPL, p0 = (b / NSp) * NSp * Rp + (b % NSp);
          for (int t = 0; t < Rp; ++t) X[p0 + t * NSp] = v[q + (i + Sp * t) * TPL];
        }
      v = X;
    }
    for (int q = 0; q < TPL; ++q)
      for (int i = 0; i < S; ++i) {
        std::vector<cd> a(R), o(R);
        const int b = q + i * TPL, k = b % NS;
        for (int t = 0; t < R; ++t) a[t] = v[q + (i + S * t) * TPL] * std::polar(1.0, -2.0 * M_PI * (double)(t * k) / (double)(NS * R));
        for (int u = 0; u < R; ++u) {
          o[u] = 0;
          for (int t = 0; t < R; ++t) o[u] += a[t] * std::polar(1.0, -2.0 * M_PI * (double)((u * t) % R) / R);
        }
        for (int t = 0; t < R; ++t) v[q + (i + S * t) * TPL] = o[t];
      }
    NS *= R;
  }
}
template <int N, class Map, int STOP>
void run(const char *name) {
  constexpr int T = ZPlan<N>::T;
  std::vector<cd> x(T * N), X(T * N), tw(N);
  for (int i = 0; i < T * N; ++i) x[i] = {std::sin(0.37 * i) + 0.1, std::cos(0.11 * i * i)};
  for (int k = 0; k < N; ++k) tw[k] = std::polar(1.0, -2.0 * M_PI * k / N);
  kcplx *din, *dout, *dtw;
  hipMalloc(&din, 16 * T * N); hipMalloc(&dout, 16 * T * N); hipMalloc(&dtw, 16 * N);
  hipMemcpy(din, x.data(), 16 * T * N, hipMemcpyHostToDevice);
  hipMemcpy(dtw, tw.data(), 16 * N, hipMemcpyHostToDevice);
  const size_t lds = 16 * (N + 16 * 260);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k_probe<N, Map, STOP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((k_probe<N, Map, STOP>), dim3(1), dim3(ZPlan<N>::NT), lds, 0, din, dout, dtw);
  hipError_t e = hipDeviceSynchronize();
  hipMemcpy(X.data(), dout, 16 * T * N, hipMemcpyDeviceToHost);
  std::vector<cd> ref(x.begin(), x.begin() + N);
  emulate<N>(ref, STOP);
  printf("%s N=%d stop after stage %d err=%d: bad register slots (q,m):", name, N, STOP, (int)e);
  int nbad = 0;
  for (int p = 0; p < N; ++p)
    if (std::abs(ref[p] - X[p]) > 1e-9) { if (nbad < 40) printf(" (%d,%d)", p % Plan<N>::TPL, p / Plan<N>::TPL); ++nbad; }
  printf("  [%d bad]\n", nbad);
}
template <int N> struct StridedSt { static constexpr bool staged_tw = true; __device__ static int at(int p, int l) { return MapStrided<N>::at(p, l); } };
int main() {
  run<240, MapLine<240>, 1>("line+staged"); run<240, MapLine<240>, 2>("line+staged"); run<240, MapLine<240>, 3>("line+staged"); run<240, MapLine<240>, 4>("line+staged");
  run<240, MapStrided<240>, 2>("strided+natural"); run<240, MapStrided<240>, 4>("strided+natural");
  run<120, MapLine<120>, 3>("line+staged");
  return 0;
}
