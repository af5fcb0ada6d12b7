// LDS-DMA loader / consumer probe for the strided FFT passes (no arithmetic, optional synthetic work): persistent workgroups of
// NCW consumer waves + one loader wave.  The loader streams the tiles of the workgroup through a ring of LDS slots with
// global_load_lds_dwordx4 (no VGPR destination: the in-flight bytes cost no registers); the consumers copy a landed slot into
// registers (the points q + m*TPL of their line, exactly the ownership of the register-radix FFT plans), run `work` dependent FMAs
// per value and store the tile with the pass's store pattern.  All waves walk the same sequence of s_barriers (the loader's DMAs stay
// in flight across them: a barrier does not drain VMEM), so there is no flag polling and every wave reaches the end of the grid.
//
// Compared in the same process with the register-staged form the product kernels use today (k_move: all loads, barrier, all stores,
// two workgroups per CU) and with a plain grid-stride copy, on the geometries of
//   slab  : k_pass_sub_w<Wide512, fwd> of 512^3 / 8   (nx 512 lines of stride nyl*257, 16-line tiles, dense 257 -> padded 264 rows)
//   serial: the x pass of 256^3                      (nx 256, tiles over the flattened (y, kz) index of 256*129 columns)
//   hipcc -O3 --offload-arch=gfx950 tools/ldsdma_probe.hip -o marlin_amd/lib/ldsdma_probe && marlin_amd/lib/ldsdma_probe
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

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
template <bool NT>
__device__ __forceinline__ void glds16(const void *g, void *l) {
  __builtin_amdgcn_global_load_lds((glb_void *)g, (lds_void *)l, 16, 0, NT ? 2 : 0);
}

struct Args {
  const double2 *in;
  double2 *out;
  unsigned rows, cols, tcols;    // tile space rows x tcols, valid columns < cols
  unsigned pitch_in, pitch_out;  // row pitch (elements)
  unsigned sn_in, sn_out;        // stride between consecutive points of a line (elements)
  unsigned ntiles;
  int work;                      // dependent FMAs per value between the slot read and the store
  int clip;                      // 1: store only columns < cols (tiles that overhang a row: the fused y pass)
  int remap;                     // 1: every XCD owns a contiguous range of tiles (the product kernels' xcd_remap)
};

// bijective XCD-aware remap: hardware deals block b to XCD b % 8; give each XCD a contiguous range of logical tiles
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nb) {
  const unsigned q8 = nb >> 3, r8 = nb & 7, xcd = b & 7, idx = b >> 3;
  return (xcd < r8) ? xcd * (q8 + 1) + idx : r8 * (q8 + 1) + (xcd - r8) * q8 + idx;
}

template <int N>
__device__ __forceinline__ void vm_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// N points per line, T lines per tile (T columns of 16 bytes = one coalesced piece per point), XS points of every line per ring slot,
// NSLOT ring slots, 4 consumer waves (256 threads own the tile: P = N*T/256 points each) + 1 loader wave
template <int N, int T, int XS, int NSLOT, bool NT>
__global__ void __launch_bounds__(320) k_dma_move(Args a) {
  constexpr int TPL = 256 / T, P = N / TPL, SPT = N / XS, PPS = XS / TPL;  // slots per tile, points per thread and slot
  constexpr int DPS = XS * T / 64;                                         // DMA wave-instructions per slot (1 KiB each)
  static_assert(XS % TPL == 0 && (XS * T) % 64 == 0 && 64 % T == 0, "slot shape");
  static_assert((NSLOT - 2) * DPS <= 63 && NSLOT >= 3, "vmcnt is a 6-bit counter");
  extern __shared__ double2 ring[];  // [NSLOT][XS][T]
  const unsigned G = gridDim.x, wg = blockIdx.x;
  // remap: XCD x = wg % 8 owns the tiles [x * per, (x + 1) * per); its G/8 workgroups walk them interleaved
  const unsigned per = (a.ntiles + 7) / 8, xcd = wg & 7, wi = wg >> 3, gx = G >> 3;
  const unsigned xlo = xcd * per, xhi = min(a.ntiles, xlo + per), xn = xhi > xlo ? xhi - xlo : 0u;
  const unsigned mytiles = a.remap ? (wi < xn ? (xn - wi + gx - 1) / gx : 0u) : (wg < a.ntiles ? (a.ntiles - wg + G - 1) / G : 0u);
  auto tile_of = [&](unsigned t) { return a.remap ? xlo + wi + t * gx : wg + t * G; };
  const unsigned F = mytiles * SPT;  // slot fills of this workgroup
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;

  if (wave == 4) {
    // ---------------- loader wave
    const unsigned l = lane % T, xr = lane / T;  // one instruction = 64/T consecutive points x T columns
    auto issue = [&](unsigned f) {
      const unsigned tile = tile_of(f / SPT), j = f % SPT;
      const unsigned i = tile * T + l;
      const unsigned ic = i < a.rows * a.tcols ? i : 0u;
      const unsigned row = ic / a.tcols, col = ic - row * a.tcols;
      const unsigned cc = col < a.cols ? col : a.cols - 1u;
      const double2 *src = a.in + (size_t)row * a.pitch_in + cc + (size_t)(j * XS + xr) * a.sn_in;
      double2 *dst = ring + (size_t)(f % NSLOT) * (XS * T);
#pragma unroll
      for (int k = 0; k < DPS; ++k) glds16<NT>(src + (size_t)k * (64 / T) * a.sn_in, dst + k * 64);
    };
    for (unsigned f = 0; f < (unsigned)(NSLOT - 1) && f < F; ++f) issue(f);
    for (unsigned f = 0; f < F; ++f) {
      // fills issued so far: min(F, f + NSLOT - 1); fill f has landed once at most NSLOT - 2 younger fills are outstanding
      if (f + NSLOT - 1 <= F)
        vm_wait<(NSLOT - 2) * DPS>();
      else
        vm_wait<0>();
      __builtin_amdgcn_s_barrier();  // B(f): slot f is published; the consumers have finished reading fill f - 1
      if (f + NSLOT - 1 < F) issue(f + NSLOT - 1);
      if (f % SPT == SPT - 1) {  // the consumers' two exchange barriers at the end of a tile
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
      }
    }
    return;
  }

  // ---------------- consumer waves
  const unsigned l = threadIdx.x % T, q = threadIdx.x / T;
  for (unsigned t = 0; t < mytiles; ++t) {
    const unsigned tile = tile_of(t);
    double2 v[P];
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      const unsigned f = t * SPT + j;
      __syncthreads();  // B(f)   (lgkmcnt(0) + s_barrier: the reads of fill f - 1 are complete)
      const double2 *slot = ring + (size_t)(f % NSLOT) * (XS * T);
#pragma unroll
      for (int m = 0; m < PPS; ++m) v[j * PPS + m] = slot[(q + m * TPL) * T + l];
    }
    __syncthreads();
    for (int w = 0; w < a.work; ++w) {
#pragma unroll
      for (int m = 0; m < P; ++m) {
        v[m].x = v[m].x * 1.0000001 + v[(m + 1) % P].y;
        v[m].y = v[m].y * 0.9999999 - v[(m + 1) % P].x;
      }
    }
    __syncthreads();
    const unsigned i = tile * T + l;
    const bool valid0 = i < a.rows * a.tcols;
    const unsigned ic = valid0 ? i : 0u;
    const unsigned row = ic / a.tcols, col = ic - row * a.tcols;
    const bool valid = valid0 && (!a.clip || col < a.cols);
    double2 *o = a.out + (size_t)row * a.pitch_out + col;
    if (valid) {
#pragma unroll
      for (int m = 0; m < P; ++m) o[(size_t)(q + m * TPL) * a.sn_out] = v[m];
    }
  }
}

// the register-staged form of the product kernels: load the whole tile, barrier, [work], store; two workgroups per CU
template <int N, int P, int T>
__global__ void __launch_bounds__(256, 2) k_move(Args a) {
  constexpr int TPL = N / P;
  const unsigned l = threadIdx.x % T, q = threadIdx.x / T;
  const unsigned i = (a.remap ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x) * T + l;
  const bool valid0 = i < a.rows * a.tcols;
  const unsigned ic = valid0 ? i : 0u;
  const unsigned row = ic / a.tcols, col = ic - row * a.tcols;
  const bool valid = valid0 && (!a.clip || col < a.cols);
  const unsigned cc = min(col, a.cols - 1u);
  const size_t bi = (size_t)row * a.pitch_in + cc, bo = (size_t)row * a.pitch_out + col;
  double2 v[P];
#pragma unroll
  for (int m = 0; m < P; ++m) v[m] = a.in[bi + (size_t)(q + m * TPL) * a.sn_in];
  __syncthreads();
  for (int w = 0; w < a.work; ++w) {
#pragma unroll
    for (int m = 0; m < P; ++m) {
      v[m].x = v[m].x * 1.0000001 + v[(m + 1) % P].y;
      v[m].y = v[m].y * 0.9999999 - v[(m + 1) % P].x;
    }
  }
  __syncthreads();
  if (valid) {
#pragma unroll
    for (int m = 0; m < P; ++m) a.out[bo + (size_t)(q + m * TPL) * a.sn_out] = v[m];
  }
}

__global__ void __launch_bounds__(256) k_copy(const double2 *__restrict__ in, double2 *__restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void k_fill(double2 *p, size_t n, double s) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_double2(s * (double)(i % 1000003), -s * (double)(i % 7919));
}
__global__ void k_diff(const double2 *a, const double2 *b, size_t n, unsigned long long *bad) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (a[i].x != b[i].x || a[i].y != b[i].y) atomicAdd(bad, 1ull);
}

// mode 8: the register-staged mover with the workgroup shape as template parameters (threads = T N / P), occupancy set through the
// dynamic LDS size as in the product kernels (their Stockham tile): which tile shape suits 512-point lines of a 512^3 grid on one GPU
template <int N, int P, int T, int MINB>
__global__ void __launch_bounds__(T * (N / P), MINB) k_move_g(Args a) {
  constexpr int TPL = N / P;
  extern __shared__ double2 dyn_lds[];
  const unsigned l = threadIdx.x % T, q = threadIdx.x / T;
  const unsigned i = (a.remap ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x) * T + l;
  const bool valid = i < a.rows * a.tcols;
  const unsigned ic = valid ? i : 0u;
  const unsigned row = ic / a.tcols, col = ic - row * a.tcols;
  const size_t bi = (size_t)row * a.pitch_in + col, bo = (size_t)row * a.pitch_out + col;
  double2 v[P];
#pragma unroll
  for (int m = 0; m < P; ++m) v[m] = a.in[bi + (size_t)(q + m * TPL) * a.sn_in];
  if (a.work < 0) dyn_lds[threadIdx.x] = v[0];  // (keeps the allocation alive)
  __syncthreads();
  for (int w = 0; w < a.work; ++w) {
#pragma unroll
    for (int m = 0; m < P; ++m) {
      v[m].x = v[m].x * 1.0000001 + v[(m + 1) % P].y;
      v[m].y = v[m].y * 0.9999999 - v[(m + 1) % P].x;
    }
  }
  __syncthreads();
  if (valid) {
#pragma unroll
    for (int m = 0; m < P; ++m) a.out[bo + (size_t)(q + m * TPL) * a.sn_out] = v[m];
  }
}
__global__ void __launch_bounds__(256) k_copy12(const double2 *__restrict__ in, double2 *__restrict__ o1, double2 *__restrict__ o2, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double2 v = in[i];
    o1[i] = v;
    o2[i] = make_double2(v.y, v.x);
  }
}
__global__ void __launch_bounds__(256) k_copy21(const double2 *__restrict__ i1, const double2 *__restrict__ i2, double2 *__restrict__ o, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double2 v = i1[i], w = i2[i];
    o[i] = make_double2(v.y + w.x, v.x - w.y);
  }
}

int main(int argc, char **argv) {
  const int which = argc > 1 ? atoi(argv[1]) : 0;  // 0 both, 1 slab geometry, 2 serial geometry
  const int NBUF = 3;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  unsigned long long *d_bad;
  CK(hipMalloc(&d_bad, 8));

  auto bench = [&](const char *geom, Args base, size_t cap, size_t payload_elems, auto k_reg, unsigned reg_blocks, auto k_dma_a, auto k_dma_b,
                   size_t lds_a, size_t lds_b, const char *name_a, const char *name_b) {
    std::vector<double2 *> in(NBUF), out(NBUF);
    double2 *ref;
    for (int b = 0; b < NBUF; ++b) {
      CK(hipMalloc(&in[b], cap * sizeof(double2)));
      CK(hipMalloc(&out[b], cap * sizeof(double2)));
      k_fill<<<2048, 256>>>(in[b], cap, 1.0 + b);
      CK(hipMemset(out[b], 0, cap * sizeof(double2)));
    }
    CK(hipMalloc(&ref, cap * sizeof(double2)));
    CK(hipDeviceSynchronize());
    printf("== %s: %zu MB moved per launch (read + write)\n", geom, payload_elems * 32 / 1000000);
    for (int r = 0; r < 600; ++r) k_copy<<<4096, 256>>>(in[r % NBUF], out[r % NBUF], payload_elems);  // clocks up
    auto time_it = [&](auto launch, int reps) {
      for (int w = 0; w < 3; ++w) launch(w);
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) launch(r);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      return ms * 1e3 / reps;
    };
    {
      const double us = time_it([&](int r) { k_copy<<<4096, 256>>>(in[r % NBUF], out[r % NBUF], payload_elems); }, 30);
      printf("%-64s %8.1f us %6.0f GB/s\n", "plain grid-stride copy of the same bytes", us, 32.0 * payload_elems / us * 1e-3);
    }
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dma_a), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dma_b), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
    // correctness of the LDS-DMA forms against the register form (work = 0: a pure permutation-free move)
    {
      Args a = base;
      a.work = 0;
      a.in = in[0];
      a.out = ref;
      CK(hipMemset(ref, 0, cap * sizeof(double2)));
      hipLaunchKernelGGL(k_reg, dim3(reg_blocks), dim3(256), 0, 0, a);
      for (int v = 0; v < 2; ++v) {
        a.out = out[0];
        CK(hipMemset(out[0], 0, cap * sizeof(double2)));
        if (v == 0)
          hipLaunchKernelGGL(k_dma_a, dim3(256), dim3(320), lds_a, 0, a);
        else
          hipLaunchKernelGGL(k_dma_b, dim3(256), dim3(320), lds_b, 0, a);
        CK(hipMemset(d_bad, 0, 8));
        k_diff<<<2048, 256>>>(ref, out[0], cap, d_bad);
        unsigned long long bad = 0;
        CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
        printf("   %s vs register form: %llu differing elements%s\n", v ? name_b : name_a, bad, bad ? "  <-- WRONG" : "");
      }
    }
    for (int remap : {0, 1}) {
      const int work = 4;
      Args a = base;
      a.work = work;
      a.remap = remap;
      char nm[160];
      double us = time_it([&](int r) { a.in = in[r % NBUF]; a.out = out[r % NBUF]; hipLaunchKernelGGL(k_reg, dim3(reg_blocks), dim3(256), 0, 0, a); }, 30);
      snprintf(nm, sizeof nm, "register-staged tiles, 2 WG/CU, xcd remap %d", remap);
      printf("%-64s %8.1f us %6.0f GB/s\n", nm, us, 32.0 * payload_elems / us * 1e-3);
      for (unsigned g : {256u, 512u}) {
        us = time_it([&](int r) { a.in = in[r % NBUF]; a.out = out[r % NBUF]; hipLaunchKernelGGL(k_dma_a, dim3(g), dim3(320), lds_a, 0, a); }, 30);
        snprintf(nm, sizeof nm, "%s, %u persistent WGs, xcd remap %d", name_a, g, remap);
        printf("%-64s %8.1f us %6.0f GB/s\n", nm, us, 32.0 * payload_elems / us * 1e-3);
        us = time_it([&](int r) { a.in = in[r % NBUF]; a.out = out[r % NBUF]; hipLaunchKernelGGL(k_dma_b, dim3(g), dim3(320), lds_b, 0, a); }, 30);
        snprintf(nm, sizeof nm, "%s, %u persistent WGs, xcd remap %d", name_b, g, remap);
        printf("%-64s %8.1f us %6.0f GB/s\n", nm, us, 32.0 * payload_elems / us * 1e-3);
      }
    }
    for (int b = 0; b < NBUF; ++b) {
      CK(hipFree(in[b]));
      CK(hipFree(out[b]));
    }
    CK(hipFree(ref));
  };

  if (which == 0 || which == 1) {
    // rank-local forward x pass of 512^3 / 8: [512][64][257] -> [512][64][264], tiles over the padded output rows
    const unsigned nx = 512, nyl = 64, nzc = 257, kp = 264;
    Args a{};
    a.rows = nyl; a.cols = nzc; a.tcols = kp; a.pitch_in = nzc; a.pitch_out = kp; a.sn_in = nyl * nzc; a.sn_out = nyl * kp;
    a.ntiles = (a.rows * a.tcols + 15) / 16;
    const size_t cap = (size_t)nx * nyl * kp + 4096;
    bench("slab x pass 512 x (64 x 257 -> 264), 16-line tiles", a, cap, (size_t)nx * nyl * nzc, k_move<512, 32, 16>, a.ntiles,
          k_dma_move<512, 16, 64, 5, false>, k_dma_move<512, 16, 64, 5, true>, 5 * 64 * 16 * 16, 5 * 64 * 16 * 16, "LDS-DMA 5 x 16 KB ring",
          "LDS-DMA 5 x 16 KB ring, nt loads");
  }
  if (which == 0 || which == 2) {
    // x pass of 256^3: lines of 256 points, stride 256 * 129, tiles over the flattened (y, kz) index
    const unsigned nx = 256, ncol = 256 * 129;
    Args a{};
    a.rows = 1; a.cols = ncol; a.tcols = ncol; a.pitch_in = ncol; a.pitch_out = ncol; a.sn_in = ncol; a.sn_out = ncol;
    a.ntiles = (ncol + 15) / 16;
    const size_t cap = (size_t)nx * ncol + 4096;
    bench("serial x pass 256 x (256 x 129), 16-line tiles", a, cap, (size_t)nx * ncol, k_move<256, 16, 16>, a.ntiles,
          k_dma_move<256, 16, 64, 5, false>, k_dma_move<256, 16, 32, 8, false>, 5 * 64 * 16 * 16, 8 * 32 * 16 * 16, "LDS-DMA 5 x 16 KB ring",
          "LDS-DMA 8 x 8 KB ring");
  }
  if (which == 0 || which == 3) {
    // the same bytes in BLOCKED layouts [(y,kz)/16][x][16]: a tile is one contiguous 64 KB block on the blocked side(s)
    const unsigned nx = 256, ncol = 256 * 129, nt = ncol / 16;
    const size_t cap = (size_t)nx * ncol + 4096;
    for (int g = 0; g < 3; ++g) {
      Args a{};
      a.rows = nt; a.cols = 16; a.tcols = 16; a.ntiles = nt;
      const bool in_blocked = g != 2, out_blocked = g != 1;
      a.pitch_in = in_blocked ? 16 * nx : 16;  a.sn_in = in_blocked ? 16 : ncol;
      a.pitch_out = out_blocked ? 16 * nx : 16; a.sn_out = out_blocked ? 16 : ncol;
      const char *nm[3] = {"256^3 x pass, blocked -> blocked (64 KB contiguous tiles both sides)", "256^3 x pass, blocked reads -> strided 256-B stores",
                           "256^3 x pass, strided 256-B reads -> blocked stores"};
      bench(nm[g], a, cap, (size_t)nx * ncol, k_move<256, 16, 16>, a.ntiles, k_dma_move<256, 16, 64, 5, false>, k_dma_move<256, 16, 32, 8, false>,
            5 * 64 * 16 * 16, 8 * 32 * 16 * 16, "LDS-DMA 5 x 16 KB ring", "LDS-DMA 8 x 8 KB ring");
    }
  }
  if (which == 4) {
    // the strided x pass of 256^3 again with a PADDED plane pitch: consecutive points of a line are (2064 + pad/16) 256-byte pieces apart
    const unsigned nx = 256, ncol = 256 * 129;
    for (unsigned pad : {0u, 8u, 16u, 48u, 80u, 272u}) {
      Args a{};
      const unsigned sn = ncol + pad;
      a.rows = 1; a.cols = ncol; a.tcols = ncol; a.pitch_in = sn; a.pitch_out = sn; a.sn_in = sn; a.sn_out = sn;
      a.ntiles = (ncol + 15) / 16;
      const size_t cap = (size_t)nx * sn + 4096;
      char nm[128];
      snprintf(nm, sizeof nm, "serial x pass 256 x (256 x 129), plane pitch + %u elements (%u B)", pad, pad * 16);
      bench(nm, a, cap, (size_t)nx * ncol, k_move<256, 16, 16>, a.ntiles, k_dma_move<256, 16, 64, 5, false>, k_dma_move<256, 16, 32, 8, false>,
            5 * 64 * 16 * 16, 8 * 32 * 16 * 16, "LDS-DMA 5 x 16 KB ring", "LDS-DMA 8 x 8 KB ring");
    }
  }
  if (which == 5) {
    // slab x pass with padded line strides on both sides
    const unsigned nx = 512, nyl = 64, nzc = 257, kp = 264;
    for (unsigned pad : {0u, 16u, 48u}) {
      Args a{};
      a.rows = nyl; a.cols = nzc; a.tcols = kp; a.pitch_in = nzc; a.pitch_out = kp; a.sn_in = nyl * nzc + pad + (pad ? 15 - (nyl * nzc + 15) % 16 : 0); a.sn_out = nyl * kp + pad;
      a.ntiles = (a.rows * a.tcols + 15) / 16;
      const size_t cap = (size_t)nx * (nyl * kp + pad) + 4096;
      char nm[128];
      snprintf(nm, sizeof nm, "slab x pass 512 x (64 x 257 -> 264), line strides %u / %u elements", a.sn_in, a.sn_out);
      bench(nm, a, cap, (size_t)nx * nyl * nzc, k_move<512, 32, 16>, a.ntiles, k_dma_move<512, 16, 64, 5, false>, k_dma_move<512, 16, 64, 5, true>,
            5 * 64 * 16 * 16, 5 * 64 * 16 * 16, "LDS-DMA 5 x 16 KB ring", "LDS-DMA 5 x 16 KB ring, nt loads");
    }
  }
  if (which == 6) {
    // the fused y pass of 512^3 / 8 on the rank-local arrays [64 x planes][512 y][K]: lines along y (stride K), 17 tiles of 16 kz per x plane
    const unsigned nxl = 64, ny = 512, nzc = 257;
    for (unsigned K : {257u, 264u, 272u, 280u}) {
      Args a{};
      a.rows = nxl; a.cols = nzc; a.tcols = 272; a.clip = 1;
      a.pitch_in = a.pitch_out = ny * K; a.sn_in = a.sn_out = K;
      a.ntiles = (a.rows * a.tcols + 15) / 16;
      const size_t cap = (size_t)nxl * ny * K + 4096;
      char nm[128];
      snprintf(nm, sizeof nm, "slab y pass 512 x (64 x 257), row pitch K = %u elements (%u B)", K, K * 16);
      bench(nm, a, cap, (size_t)nxl * ny * nzc, k_move<512, 32, 16>, a.ntiles, k_dma_move<512, 16, 64, 5, false>, k_dma_move<512, 16, 64, 5, true>,
            5 * 64 * 16 * 16, 5 * 64 * 16 * 16, "LDS-DMA 5 x 16 KB ring", "LDS-DMA 5 x 16 KB ring, nt loads");
    }
  }
  if (which == 7) {
    // light run for rocprofv3 --pmc (tools/pmc_probe.sh): the 256^3 x pass with plane pitch + argv[2] elements, four launches of the
    // register-staged form and four of the LDS-DMA form, nothing else
    const unsigned nx = 256, ncol = 256 * 129, pad = argc > 2 ? atoi(argv[2]) : 0, sn = ncol + pad;
    Args a{};
    a.rows = 1; a.cols = ncol; a.tcols = ncol; a.pitch_in = sn; a.pitch_out = sn; a.sn_in = sn; a.sn_out = sn;
    a.ntiles = (ncol + 15) / 16; a.work = 4; a.remap = 1;
    const size_t cap = (size_t)nx * sn + 4096;
    double2 *in[2], *out[2];
    for (int b = 0; b < 2; ++b) {
      CK(hipMalloc(&in[b], cap * sizeof(double2)));
      CK(hipMalloc(&out[b], cap * sizeof(double2)));
      k_fill<<<2048, 256>>>(in[b], cap, 1.0 + b);
    }
    auto kd = k_dma_move<256, 16, 64, 5, false>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kd), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 64 * 16 * 16));
    for (int r = 0; r < 4; ++r) {
      a.in = in[r % 2]; a.out = out[r % 2];
      hipLaunchKernelGGL((k_move<256, 16, 16>), dim3(a.ntiles), dim3(256), 0, 0, a);
    }
    for (int r = 0; r < 4; ++r) {
      a.in = in[r % 2]; a.out = out[r % 2];
      hipLaunchKernelGGL(kd, dim3(512), dim3(320), 5 * 64 * 16 * 16, 0, a);
    }
    CK(hipDeviceSynchronize());
    printf("pad %u done\n", pad);
  }
  if (which == 8) {
    // 512^3 on one GPU: x pass over [512][512 x 257 (+ pad)] and y pass over [512 planes][512][257]; 1.08 GB per array (no cache holds it)
    const unsigned n = 512, nzc = 257;
    const size_t plane = (size_t)n * nzc;
    double2 *in[2], *out[2];
    const size_t cap = (size_t)n * (plane + 64) + 4096;
    for (int b = 0; b < 2; ++b) {
      CK(hipMalloc(&in[b], cap * sizeof(double2)));
      CK(hipMalloc(&out[b], cap * sizeof(double2)));
      k_fill<<<2048, 256>>>(in[b], cap, 1.0 + b);
      CK(hipMemset(out[b], 0, cap * sizeof(double2)));
    }
    CK(hipDeviceSynchronize());
    const size_t payload = (size_t)n * plane;
    for (int r = 0; r < 100; ++r) k_copy<<<4096, 256>>>(in[r % 2], out[r % 2], payload);
    auto time_it = [&](auto launch, int reps) {
      for (int w = 0; w < 2; ++w) launch(w);
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) launch(r);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      return ms * 1e3 / reps;
    };
    double us = time_it([&](int r) { k_copy<<<4096, 256>>>(in[r % 2], out[r % 2], payload); }, 20);
    printf("%-72s %8.1f us %6.0f GB/s\n", "grid-stride copy 1 read : 1 write, 1.08 GB arrays", us, 32.0 * payload / us * 1e-3);
    us = time_it([&](int r) { k_copy<<<16384, 256>>>(in[r % 2], out[r % 2], payload); }, 20);
    printf("%-72s %8.1f us %6.0f GB/s\n", "  ... 16384 workgroups", us, 32.0 * payload / us * 1e-3);
    us = time_it([&](int r) { k_copy12<<<4096, 256>>>(in[r % 2], out[0], out[1], payload); }, 20);
    printf("%-72s %8.1f us %6.0f GB/s\n", "copy 1 read : 2 writes (the fused z passes' ratio)", us, 48.0 * payload / us * 1e-3);
    us = time_it([&](int r) { k_copy21<<<4096, 256>>>(in[0], in[1], out[r % 2], payload); }, 20);
    printf("%-72s %8.1f us %6.0f GB/s\n", "copy 2 reads : 1 write", us, 48.0 * payload / us * 1e-3);
    auto run = [&](const char *nm, auto kern, unsigned T, unsigned threads, size_t lds, Args a) {
      CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      a.ntiles = (unsigned)(((size_t)a.rows * a.tcols + T - 1) / T);
      for (int work : {0, 4, 12}) {
        a.work = work;
        const double t = time_it([&](int r) { a.in = in[r % 2]; a.out = out[r % 2]; hipLaunchKernelGGL(kern, dim3(a.ntiles), dim3(threads), lds, 0, a); }, 12);
        printf("%-60s work %2d  %8.1f us %6.0f GB/s\n", nm, work, t, 32.0 * payload / t * 1e-3);
      }
    };
    for (unsigned pad : {0u, 16u}) {
      for (int pass = 0; pass < 2; ++pass) {
        Args a{};
        a.remap = 1;
        if (pass == 0) {  // x pass: lines across planes, tiles over the flattened (y, kz) index of a plane
          const unsigned sn = (unsigned)plane + pad;
          a.rows = 1; a.cols = (unsigned)plane; a.tcols = (unsigned)plane; a.pitch_in = a.pitch_out = sn; a.sn_in = a.sn_out = sn;
          printf("== x pass 512 x (512 x 257), plane pitch + %u elements\n", pad);
        } else {          // y pass: lines along y inside an x plane (stride nzc), tiles over (x, kz)
          if (pad) continue;
          a.rows = n; a.cols = nzc; a.tcols = nzc; a.pitch_in = a.pitch_out = (unsigned)plane; a.sn_in = a.sn_out = nzc;
          printf("== y pass 512 x (512 planes x 257)\n");
        }
        run("16 pts/thread,  8 lines (128-B pieces), 256 thr, 2 WG/CU", k_move_g<512, 16, 8, 2>, 8, 256, 64 * 1024, a);
        run("16 pts/thread,  8 lines (128-B pieces), 256 thr, 4 WG/CU", k_move_g<512, 16, 8, 4>, 8, 256, 32 * 1024, a);
        run("16 pts/thread, 16 lines (256-B pieces), 512 thr, 1 WG/CU", k_move_g<512, 16, 16, 1>, 16, 512, 128 * 1024, a);
        run("16 pts/thread, 16 lines (256-B pieces), 512 thr, 2 WG/CU", k_move_g<512, 16, 16, 2>, 16, 512, 64 * 1024, a);
        run("32 pts/thread, 16 lines (256-B pieces), 256 thr, 2 WG/CU", k_move_g<512, 32, 16, 2>, 16, 256, 64 * 1024, a);
        run("32 pts/thread, 32 lines (512-B pieces), 512 thr, 1 WG/CU", k_move_g<512, 32, 32, 1>, 32, 512, 128 * 1024, a);
        run("8 pts/thread,   8 lines (128-B pieces), 512 thr, 2 WG/CU", k_move_g<512, 8, 8, 2>, 8, 512, 64 * 1024, a);
      }
    }
  }
  return 0;
}
