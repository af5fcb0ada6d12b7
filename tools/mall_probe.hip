// How fast is a copy whose working set fits the 256 MiB Infinity Cache?  Plain 16-byte grid-stride copies in -> out, the same two
// buffers every launch, sizes from 8 MB to 1 GB per buffer; also read-only (sum) and write-only (fill) streams.
//   hipcc -O3 --offload-arch=gfx950 tools/mall_probe.hip -o marlin_amd/lib/mall_probe
#include <hip/hip_runtime.h>
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

__global__ void __launch_bounds__(256) k_copy(const double2 *__restrict__ in, double2 *__restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void __launch_bounds__(256) k_read(const double2 *__restrict__ in, double *__restrict__ sink, size_t n) {
  double acc = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double2 v = in[i];
    acc += v.x + v.y;
  }
  if (acc == 1.2345e300) sink[0] = acc;
}
__global__ void __launch_bounds__(256) k_fill(double2 *__restrict__ out, size_t n, double v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = make_double2(v, v);
}

// the same streams with non-temporal accesses (nt bit: stream past the Infinity Cache)
__global__ void __launch_bounds__(256) k_copy_nt(const double2 *__restrict__ in, double2 *__restrict__ out, size_t n, int ntl, int nts) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    double2 v;
    if (ntl) {
      v.x = __builtin_nontemporal_load(&in[i].x);
      v.y = __builtin_nontemporal_load(&in[i].y);
    } else {
      v = in[i];
    }
    if (nts) {
      __builtin_nontemporal_store(v.x, &out[i].x);
      __builtin_nontemporal_store(v.y, &out[i].y);
    } else {
      out[i] = v;
    }
  }
}
__global__ void __launch_bounds__(256) k_fill_nt(double2 *__restrict__ out, size_t n, double v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    __builtin_nontemporal_store(v, &out[i].x);
    __builtin_nontemporal_store(v, &out[i].y);
  }
}

int main() {
  const size_t cap = (size_t)1 << 30;
  double2 *a, *b;
  double *sink;
  CK(hipMalloc(&a, cap));
  CK(hipMalloc(&b, cap));
  CK(hipMalloc(&sink, 64));
  k_fill<<<4096, 256>>>(a, cap / 16, 1.0);
  k_fill<<<4096, 256>>>(b, cap / 16, 2.0);
  for (int r = 0; r < 300; ++r) k_copy<<<4096, 256>>>(a, b, cap / 64);   // clocks
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("%10s %14s %14s %14s   (GB/s; copy counts read + written bytes)\n", "MB/buffer", "copy", "read only", "write only");
  for (size_t mb : {8, 16, 32, 48, 64, 96, 112, 128, 160, 192, 256, 384, 512, 1024}) {
    const size_t n = mb * 1024 * 1024 / 16;
    const int reps = mb <= 64 ? 400 : 100;
    float ms[3];
    for (int mode = 0; mode < 3; ++mode) {
      for (int r = 0; r < 10; ++r) {
        if (mode == 0) k_copy<<<4096, 256>>>(a, b, n);
        if (mode == 1) k_read<<<4096, 256>>>(a, sink, n);
        if (mode == 2) k_fill<<<4096, 256>>>(b, n, 3.0);
      }
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) {
        if (mode == 0) k_copy<<<4096, 256>>>(a, b, n);
        if (mode == 1) k_read<<<4096, 256>>>(a, sink, n);
        if (mode == 2) k_fill<<<4096, 256>>>(b, n, 3.0);
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms[mode], e0, e1));
      ms[mode] /= reps;
    }
    const double bytes = (double)mb * 1024 * 1024;
    printf("%10zu %14.0f %14.0f %14.0f\n", mb, 2 * bytes / ms[0] * 1e-6, bytes / ms[1] * 1e-6, bytes / ms[2] * 1e-6);
  }
  printf("\n%10s %14s %14s %14s %14s   (non-temporal: write only | copy nt stores | copy nt loads | copy both)\n", "MB/buffer", "write nt", "copy st-nt", "copy ld-nt", "copy both-nt");
  for (size_t mb : {64, 128, 256, 512, 1024}) {
    const size_t n = mb * 1024 * 1024 / 16;
    const int reps = 100;
    float ms[4];
    for (int mode = 0; mode < 4; ++mode) {
      auto go = [&]() {
        if (mode == 0) k_fill_nt<<<4096, 256>>>(b, n, 3.0);
        if (mode == 1) k_copy_nt<<<4096, 256>>>(a, b, n, 0, 1);
        if (mode == 2) k_copy_nt<<<4096, 256>>>(a, b, n, 1, 0);
        if (mode == 3) k_copy_nt<<<4096, 256>>>(a, b, n, 1, 1);
      };
      for (int r = 0; r < 10; ++r) go();
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) go();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms[mode], e0, e1));
      ms[mode] /= reps;
    }
    const double bytes = (double)mb * 1024 * 1024;
    printf("%10zu %14.0f %14.0f %14.0f %14.0f\n", mb, bytes / ms[0] * 1e-6, 2 * bytes / ms[1] * 1e-6, 2 * bytes / ms[2] * 1e-6, 2 * bytes / ms[3] * 1e-6);
  }
  return 0;
}
