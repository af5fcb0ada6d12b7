// What a chain of dependent small kernels costs on one stream: plain launches against a captured hipGraph replayed.
//   hipcc -O3 --offload-arch=gfx950 tools/launch_probe.hip -o marlin_amd/lib/launch_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

__global__ void k_small(double *p, int n, int work) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = p[i];
  for (int w = 0; w < work; ++w) v = v * 1.0000001 + 1e-9;
  p[i] = v;
}

int main() {
  const int n = 128 * 128;
  double *d;
  CK(hipMalloc(&d, n * sizeof(double)));
  CK(hipMemset(d, 0, n * sizeof(double)));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  for (int work : {0, 200, 2000}) {
    for (int chain : {3, 4}) {
      const int reps = 3000;
      for (int r = 0; r < 300; ++r) hipLaunchKernelGGL(k_small, dim3(n / 256), dim3(256), 0, s, d, n, work);
      CK(hipStreamSynchronize(s));
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < reps * chain; ++r) hipLaunchKernelGGL(k_small, dim3(n / 256), dim3(256), 0, s, d, n, work);
      CK(hipStreamSynchronize(s));
      const double plain = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
      // one "substep" = `chain` dependent kernels; graphs of 1, 2, 8 and 32 substeps
      for (int per : {1, 2, 8, 32}) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int r = 0; r < per * chain; ++r) hipLaunchKernelGGL(k_small, dim3(n / 256), dim3(256), 0, s, d, n, work);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps / per; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        const double gr = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps / per * per);
        printf("work %5d  chain of %d kernels: plain launches %6.2f us per substep | graph of %2d substeps %6.2f us per substep\n", work, chain,
               plain, per, gr);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
      }
    }
  }
  return 0;
}
