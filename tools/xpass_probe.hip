// Memory-pattern probe for the slab x passes (no arithmetic): every workgroup moves a tile of T lines x N points exactly as
// k_pass_sub_w does (thread (q, l): points q + m * TPL of line l), from an input with a given line stride / row pitch to an output
// with another.  Prints GB/s per variant so that the pattern's ceiling can be separated from the transform's latency structure.
//   hipcc -O3 --offload-arch=gfx950 tools/xpass_probe.hip -o /tmp/xpass_probe && /tmp/xpass_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

struct Args {
  const double2 *in;
  double2 *out;
  unsigned rows, cols, tcols;       // tile space rows x tcols, valid columns < cols
  unsigned pitch_in, pitch_out;     // row pitch (elements)
  unsigned sn_in, sn_out;           // stride between consecutive points of a line (elements)
  unsigned rs_in;                   // stride between rows on the input side if it is not pitch_in ([y][x][k] layout): 0 = pitch_in
};

__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned n) {
  const unsigned per = (n + 7) / 8;
  const unsigned x = b % 8, i = b / 8;
  const unsigned r = x * per + i;
  return r < n ? r : b;   // (good enough for a probe: n is a multiple of 8 in every variant)
}

template <int N, int P, int T, int MODE = 0>
__global__ void __launch_bounds__(256, 2) k_move(Args a) {
  constexpr int TPL = N / P;
  const unsigned l = threadIdx.x % T, q = threadIdx.x / T;
  const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned i = logical * T + l;
  const bool valid = i < a.rows * a.tcols;
  const unsigned ic = valid ? i : 0u;
  const unsigned row = ic / a.tcols, col = ic - row * a.tcols;
  const unsigned cc = min(col, a.cols - 1u);
  const size_t bi = (size_t)row * (a.rs_in ? a.rs_in : a.pitch_in) + cc, bo = (size_t)row * a.pitch_out + col;
  double2 v[P];
  if (MODE == 2) {  // two halves: load 16, store 16, load 16, store 16
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int m = h * P / 2; m < (h + 1) * P / 2; ++m) v[m] = a.in[bi + (size_t)(q + m * TPL) * a.sn_in];
      if (valid) {
#pragma unroll
        for (int m = h * P / 2; m < (h + 1) * P / 2; ++m) a.out[bo + (size_t)(q + m * TPL) * a.sn_out] = v[m];
      }
    }
    return;
  }
  const unsigned rot = MODE == 1 ? logical : 0u;  // MODE 1: every tile walks its points in a different rotation
#pragma unroll
  for (int m = 0; m < P; ++m) v[m] = a.in[bi + (size_t)(q + ((m + rot) % P) * TPL) * a.sn_in];
  __syncthreads();
  if (valid) {
#pragma unroll
    for (int m = 0; m < P; ++m) a.out[bo + (size_t)(q + ((m + rot) % P) * TPL) * a.sn_out] = v[m];
  }
}

__global__ void __launch_bounds__(256) k_copy(const double2 *__restrict__ in, double2 *__restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

__global__ void k_fill(double2 *p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_double2(1.0, 2.0);
}

int main(int argc, char **argv) {
  const unsigned nx = 512, nyl = argc > 1 ? atoi(argv[1]) : 64, nzc = 257, kp = 264;
  printf("nyl = %u\n", nyl);
  const size_t cap = (size_t)nx * nyl * 272 + 4096;
  const int NBUF = nyl > 64 ? 2 : 6;  // rotate over more than 256 MB so that nothing survives in the Infinity Cache
  double2 *in[6], *out[6];
  for (int b = 0; b < NBUF; ++b) {
    CK(hipMalloc(&in[b], cap * sizeof(double2)));
    CK(hipMalloc(&out[b], cap * sizeof(double2)));
    k_fill<<<2048, 256>>>(in[b], cap);
    k_fill<<<2048, 256>>>(out[b], cap);
  }
  CK(hipDeviceSynchronize());
  struct V {
    const char *name;
    unsigned tcols, pitch_in, pitch_out, sn_in, sn_out, rs_in;
    int rot;  // 1 = rotate buffers (uncached), 0 = same buffers every launch
  };
  std::vector<V> vs = {
      {"fwd now:   in [x][y][257] -> out [x][y][264], tiles over 264 (aligned stores)", kp, nzc, kp, nyl * nzc, nyl * kp, 0, 1},
      {"fwd dense: in [x][y][257] -> out [x][y][264], tiles over 257 (aligned loads)", nzc, nzc, kp, nyl * nzc, nyl * kp, 0, 1},
      {"both 264:  in [x][y][264] -> out [x][y][264]", kp, kp, kp, nyl * kp, nyl * kp, 0, 1},
      {"both 272:  in [x][y][272] -> out [x][y][272]", 272, 272, 272, nyl * 272, nyl * 272, 0, 1},
      {"in [y][x][257] (x stride 4 KB) -> out [x][y][264]", kp, nzc, kp, nzc, nyl * kp, nx * nzc, 1},
      {"in [y][x][264] (x stride 4 KB) -> out [x][y][264]", kp, kp, kp, kp, nyl * kp, nx * kp, 1},
      {"in [x][y][264] -> out [y][x][264] (x stride 4 KB on the store side)", kp, kp, nx * kp, nyl * kp, kp, 0, 1},
      {"in [y][x][264] -> out [y][x][264] (4 KB both)", kp, kp, nx * kp, kp, kp, nx * kp, 1},
      {"contiguous: tile = 128 KB contiguous on both sides", 16, 16, 16, 16, 16, 0, 1},
      {"fwd now, same buffers every launch (Infinity Cache)", kp, nzc, kp, nyl * nzc, nyl * kp, 0, 0},
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  {  // warm the clocks up (0.3 s of copies), then calibrate: a plain grid-stride copy of the same byte count
    const size_t n = (size_t)nx * nyl * nzc;
    for (int r = 0; r < 3000; ++r) k_copy<<<4096, 256>>>(in[r % NBUF], out[r % NBUF], n);
    for (unsigned g : {1024u, 2048u, 4096u, 16384u}) {
      CK(hipEventRecord(e0));
      for (int r = 0; r < 24; ++r) k_copy<<<g, 256>>>(in[r % NBUF], out[r % NBUF], n);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("plain copy, %5u workgroups: %7.1f us %6.0f GB/s\n", g, ms * 1e3 / 24, 32.0 * n / (ms * 1e3 / 24) * 1e-3);
    }
  }
  for (int pass = 0; pass < 2; ++pass)
    for (auto &v : vs) {
      Args a{};
      a.rows = nyl;
      a.cols = nzc;
      a.tcols = v.tcols;
      const bool contig = v.tcols == 16;
      a.pitch_in = v.pitch_in;
      a.pitch_out = v.pitch_out;
      a.sn_in = v.sn_in;
      a.sn_out = v.sn_out;
      a.rs_in = v.rs_in;
      unsigned nb = (a.rows * a.tcols + 15) / 16;
      if (contig) {  // rows x 16 columns with pitch 16 * 512: row r = tile r, points 16 elements apart
        a.rows = nyl * nzc / 16; a.cols = 16; a.pitch_in = a.pitch_out = 16 * 512; nb = a.rows;
      }
      const int reps = 24;
      for (int w = 0; w < 3; ++w) {
        a.in = in[w % NBUF];
        a.out = out[w % NBUF];
        hipLaunchKernelGGL((k_move<512, 32, 16>), dim3(nb), dim3(256), 0, 0, a);
      }
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) {
        a.in = in[v.rot ? r % NBUF : 0];
        a.out = out[v.rot ? r % NBUF : 0];
        hipLaunchKernelGGL((k_move<512, 32, 16>), dim3(nb), dim3(256), 0, 0, a);
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / reps;
      const double bytes = 2.0 * 16.0 * nx * nyl * nzc;
      if (pass) printf("%-82s %7.1f us  %6.0f GB/s\n", v.name, us, bytes / us * 1e-3);
    }
  // the same with 16 points per thread and 8-line tiles (the 16-point plan of 512-point lines)
  {
    Args a{};
    a.rows = nyl; a.cols = nzc; a.tcols = kp; a.pitch_in = nzc; a.pitch_out = kp; a.sn_in = nyl * nzc; a.sn_out = nyl * kp;
    const unsigned nb = (a.rows * a.tcols + 7) / 8;
    CK(hipEventRecord(e0));
    for (int r = 0; r < 24; ++r) {
      a.in = in[r % NBUF];
      a.out = out[r % NBUF];
      hipLaunchKernelGGL((k_move<512, 16, 8>), dim3(nb), dim3(256), 0, 0, a);
    }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / 24;
    printf("%-82s %7.1f us  %6.0f GB/s\n", "fwd now with 16 points per thread, 8-line tiles", us, 2.0 * 16.0 * nx * nyl * nzc / us * 1e-3);
  }
  auto run = [&](const char *name, auto kern, unsigned T, unsigned tc, unsigned pin, unsigned pout, unsigned snin, unsigned snout, int contig) {
    Args a{};
    a.rows = nyl; a.cols = nzc; a.tcols = tc; a.pitch_in = pin; a.pitch_out = pout; a.sn_in = snin; a.sn_out = snout;
    unsigned nb = (a.rows * a.tcols + T - 1) / T;
    if (contig) { a.rows = nyl * nzc / T; a.cols = a.tcols = T; a.pitch_in = a.pitch_out = T * 512; a.sn_in = a.sn_out = T; nb = a.rows; }
    for (int pass = 0; pass < 2; ++pass) {
      CK(hipEventRecord(e0));
      for (int r = 0; r < 24; ++r) {
        a.in = in[r % NBUF];
        a.out = out[r % NBUF];
        hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, 0, a);
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / 24;
    printf("%-82s %7.1f us  %6.0f GB/s\n", name, us, 2.0 * 16.0 * nx * nyl * nzc / us * 1e-3);
  };
  run("rotated walk, fwd now", k_move<512, 32, 16, 1>, 16, kp, nzc, kp, nyl * nzc, nyl * kp, 0);
  run("rotated walk, both 264", k_move<512, 32, 16, 1>, 16, kp, kp, kp, nyl * kp, nyl * kp, 0);
  run("rotated walk, contiguous", k_move<512, 32, 16, 1>, 16, 16, 16, 16, 16, 16, 1);
  run("two halves, fwd now", k_move<512, 32, 16, 2>, 16, kp, nzc, kp, nyl * nzc, nyl * kp, 0);
  run("two halves, both 264", k_move<512, 32, 16, 2>, 16, kp, kp, kp, nyl * kp, nyl * kp, 0);
  run("two halves, contiguous", k_move<512, 32, 16, 2>, 16, 16, 16, 16, 16, 16, 1);
  run("16 points per thread / 8-line tiles, both 264", k_move<512, 16, 8, 0>, 8, kp, kp, kp, nyl * kp, nyl * kp, 0);
  run("16 points per thread / 8-line tiles, contiguous", k_move<512, 16, 8, 0>, 8, 8, 8, 8, 8, 8, 1);
  run("8 points per thread / 4-line tiles, both 264", k_move<512, 8, 4, 0>, 4, kp, kp, kp, nyl * kp, nyl * kp, 0);
  run("8 points per thread / 4-line tiles, contiguous", k_move<512, 8, 4, 0>, 4, 4, 4, 4, 4, 4, 1);
  return 0;
}
