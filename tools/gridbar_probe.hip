// What a grid-wide barrier costs inside a persistent kernel on MI355X -- the building block of a one-launch substep for the small 2-D
// grids.  G workgroups, all resident, R rounds of: write own slot, barrier, read the neighbour's slot
// (checked), barrier.  Spread: workgroup i works (round-robin over the 8 XCDs, their L2s are not coherent with each other: agent-scope
// release / acquire write back and invalidate).  One XCD: 8 G workgroups are launched and only those with blockIdx % 8 == 0 work.
// Every wait is bounded by the wall clock (20 ms): a lost arrival ends the kernel with an error flag, never a hung GPU.
//   hipcc -O3 --offload-arch=gfx950 tools/gridbar_probe.hip -o marlin_amd/lib/gridbar_probe
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

struct Bar {
  unsigned count;   // arrivals
  unsigned abort_;  // set by a waiter whose time ran out
};

__device__ __forceinline__ bool grid_barrier(Bar *b, unsigned target) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(&b->count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (__hip_atomic_load(&b->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || wall_clock64() - t0 > 2000000ull) {  // 20 ms at 100 MHz
        __hip_atomic_store(&b->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  ok = __syncthreads_and(ok);
  return ok;
}

// work: floating-point operations per thread and phase (stands for the transforms of a phase)
__global__ void __launch_bounds__(256) k_rounds(Bar *bar, double *slots, int G, int rounds, int stride, int work, unsigned *errors) {
  if (blockIdx.x % stride) return;
  const int g = blockIdx.x / stride;
  unsigned target = 0;
  double v = g;
  for (int r = 0; r < rounds; ++r) {
    for (int w = 0; w < work; ++w) v = v * 1.0000001 + 1e-9;
    if (threadIdx.x == 0) slots[g * 32] = (double)(r * 1000 + g) + (v > 1e300 ? 1.0 : 0.0);
    target += G;
    if (!grid_barrier(bar, target)) return;
    if (threadIdx.x == 0) {
      const int nb = (g + 1) % G;
      const double got = __hip_atomic_load(&slots[nb * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (got != (double)(r * 1000 + nb)) atomicAdd(errors, 1u);
    }
    target += G;
    if (!grid_barrier(bar, target)) return;
  }
}

int main() {
  Bar *bar;
  double *slots;
  unsigned *errors;
  CK(hipMalloc(&bar, sizeof(Bar)));
  CK(hipMalloc(&slots, 1024 * 32 * sizeof(double)));
  CK(hipMalloc(&errors, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("%-12s %5s %6s %8s %12s %8s %8s\n", "placement", "G", "work", "rounds", "us/barrier", "errors", "aborted");
  for (int stride : {1, 8}) {
    for (int G : {8, 16, 32, 64, 128}) {
      if (stride == 8 && G > 32 * 2) continue;   // one XCD holds 32 CUs: at most 64 workgroups of 256 threads (2 per CU assumed resident)
      for (int work : {0, 2000}) {
        const int rounds = 2000;
        CK(hipMemset(bar, 0, sizeof(Bar)));
        CK(hipMemset(errors, 0, 4));
        CK(hipMemset(slots, 0, 1024 * 32 * sizeof(double)));
        k_rounds<<<G * stride, 256>>>(bar, slots, G, 20, stride, work, errors);   // warm-up
        CK(hipDeviceSynchronize());
        CK(hipMemset(bar, 0, sizeof(Bar)));
        CK(hipMemset(errors, 0, 4));
        CK(hipEventRecord(e0));
        k_rounds<<<G * stride, 256>>>(bar, slots, G, rounds, stride, work, errors);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned err = 0;
        Bar hb{};
        CK(hipMemcpy(&err, errors, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&hb, bar, sizeof(Bar), hipMemcpyDeviceToHost));
        printf("%-12s %5d %6d %8d %12.2f %8u %8u\n", stride == 1 ? "8 XCDs" : "one XCD", G, work, rounds, ms * 1e3 / (2.0 * rounds), err, hb.abort_);
        fflush(stdout);
      }
    }
  }
  return 0;
}
