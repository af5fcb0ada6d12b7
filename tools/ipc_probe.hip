// Feasibility probe for the slab transport: several processes, ONE GPU (or one GPU each), HIP IPC in dmabuf mode.
//   ipc_probe <nranks> [same_device=1]
// The parent never touches the GPU; it starts one child per rank (fork + exec of this binary with MRL_PROBE_RANK set).
// Each child: hipMalloc a buffer + a flag word, publish the IPC handles through a /dev/shm file, map the peers' buffers,
// launch a kernel that stores a pattern into every peer's buffer and then raises a flag there, wait (kernel with a bounded
// spin) for all peers' flags, verify.  Also times peer copies through the copy engines.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e = (x);                                                                         \
    if (e != hipSuccess) {                                                                      \
      fprintf(stderr, "[rank %d] %s failed: %s (line %d)\n", g_rank, #x, hipGetErrorString(e), __LINE__); \
      exit(2);                                                                                  \
    }                                                                                           \
  } while (0)

static int g_rank = -1;

struct Shared {
  std::atomic<int> arrived[16];
  hipIpcMemHandle_t buf[16];
  hipIpcMemHandle_t flag[16];
  hipIpcMemHandle_t buf2[16];
};

__global__ void k_fill(double *dst, long long n, double base) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dst[i] = base + (double)i;
}
__global__ void k_signal(unsigned long long *flag, unsigned long long v) {
  __threadfence_system();
  __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_wait(const unsigned long long *flags, int n, unsigned long long v, int *status, long long max_ticks) {
  const long long t0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
    while (__hip_atomic_load(flags + i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) {
      if (wall_clock64() - t0 > max_ticks) {
        *status = 1 + i;
        return;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  *status = 0;
}

static void barrier(Shared *sh, int slot, int n) {
  sh->arrived[slot].fetch_add(1);
  const auto t0 = std::chrono::steady_clock::now();
  while (sh->arrived[slot].load() < n) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) {
      fprintf(stderr, "[rank %d] host barrier %d timed out\n", g_rank, slot);
      exit(3);
    }
    usleep(50);
  }
}

static long long g_extra_gib = 0;
static long long g_second_mib = 0;  // argv[4]: a second symmetric buffer of this many MiB, mapped after the first
static long long g_slot_doubles = 4 << 20;  // doubles per peer slot (32 MiB; argv[3] = MiB per slot)

static int child(int rank, int nranks, int same_device, const char *shm_name) {
  g_rank = rank;
  int fd = shm_open(shm_name, O_RDWR, 0600);
  if (fd < 0) { perror("shm_open"); return 2; }
  Shared *sh = (Shared *)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  int ndev = 0;
  CK(hipGetDeviceCount(&ndev));
  const int dev = same_device ? 0 : rank % ndev;
  CK(hipSetDevice(dev));
  const long long n = g_slot_doubles;
  if (g_extra_gib > 0) {  // ballast: other device memory the process holds (argv[5] GiB in 1 GiB pieces, touched)
    for (long long k = 0; k < g_extra_gib; ++k) {
      void *b = nullptr;
      CK(hipMalloc(&b, 1ull << 30));
      CK(hipMemset(b, 0, 1ull << 30));
    }
    CK(hipDeviceSynchronize());
    fprintf(stderr, "[rank %d] %lld GiB of ballast allocated\n", rank, g_extra_gib);
  }
  if (g_second_mib > 0) {  // as the library does before its exchange buffers: two small symmetric buffers first, one of them uncached
    for (int k = 0; k < 2; ++k) {
      void *sm = nullptr;
      if (k == 0) CK(hipExtMallocWithFlags(&sm, 1 << 19, hipDeviceMallocUncached)); else CK(hipMalloc(&sm, 1 << 17));
      CK(hipMemset(sm, 0, 1 << 17));
      CK(hipDeviceSynchronize());
      CK(hipIpcGetMemHandle(&sh->buf2[rank], sm));
      barrier(sh, 6 + k, nranks);
      for (int p = 0; p < nranks; ++p) {
        if (p == rank) continue;
        void *m = nullptr;
        CK(hipIpcOpenMemHandle(&m, sh->buf2[p], hipIpcMemLazyEnablePeerAccess));
      }
      barrier(sh, 8 + k, nranks);
    }
    fprintf(stderr, "[rank %d] two small symmetric buffers mapped first\n", rank);
  }
  double *buf = nullptr;
  unsigned long long *flag = nullptr;
  CK(hipMalloc(&buf, sizeof(double) * n * nranks));
  CK(hipMalloc(&flag, 4096));
  CK(hipMemset(flag, 0, 4096));
  CK(hipMemset(buf, 0, sizeof(double) * n * nranks));
  CK(hipDeviceSynchronize());
  CK(hipIpcGetMemHandle(&sh->buf[rank], buf));
  CK(hipIpcGetMemHandle(&sh->flag[rank], flag));
  barrier(sh, 0, nranks);
  fprintf(stderr, "[rank %d] buffer of %.2f GiB allocated, handles published\n", rank, sizeof(double) * (double)n * nranks / 1073741824.0);
  std::vector<double *> pbuf(nranks);
  std::vector<unsigned long long *> pflag(nranks);
  for (int p = 0; p < nranks; ++p) {
    if (p == rank) { pbuf[p] = buf; pflag[p] = flag; continue; }
    CK(hipIpcOpenMemHandle((void **)&pbuf[p], sh->buf[p], hipIpcMemLazyEnablePeerAccess));
    CK(hipIpcOpenMemHandle((void **)&pflag[p], sh->flag[p], hipIpcMemLazyEnablePeerAccess));
  }
  fprintf(stderr, "[rank %d] peer buffers mapped\n", rank);
  if (g_second_mib > 0) {
    void *b2 = nullptr;
    CK(hipMalloc(&b2, (size_t)g_second_mib << 20));
    CK(hipMemset(b2, 0, (size_t)g_second_mib << 20));
    CK(hipDeviceSynchronize());
    CK(hipIpcGetMemHandle(&sh->buf2[rank], b2));
    barrier(sh, 4, nranks);
    fprintf(stderr, "[rank %d] second buffer of %lld MiB published\n", rank, g_second_mib);
    for (int p = 0; p < nranks; ++p) {
      if (p == rank) continue;
      void *m = nullptr;
      CK(hipIpcOpenMemHandle(&m, sh->buf2[p], hipIpcMemLazyEnablePeerAccess));
    }
    fprintf(stderr, "[rank %d] second buffer mapped\n", rank);
    barrier(sh, 5, nranks);
  }
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  int *status;
  CK(hipHostMalloc(&status, sizeof(int)));
  *status = -1;
  int clk_khz = 100000;
  CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, dev));
  // 1. direct stores into every peer's slot [rank], then a flag
  for (int p = 0; p < nranks; ++p) {
    hipLaunchKernelGGL(k_fill, dim3(512), dim3(256), 0, st, pbuf[p] + (long long)rank * n, n, 1000.0 * rank);
    hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, st, pflag[p] + rank, 1ull);
  }
  hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, st, flag, nranks, 1ull, status, (long long)clk_khz * 1000 * 20);
  CK(hipStreamSynchronize(st));
  if (*status != 0) { fprintf(stderr, "[rank %d] wait kernel timed out on peer %d\n", rank, *status - 1); return 4; }
  std::vector<double> h(n);
  int bad = 0;
  for (int p = 0; p < nranks; ++p) {
    CK(hipMemcpy(h.data(), buf + (long long)p * n, sizeof(double) * n, hipMemcpyDeviceToHost));
    for (long long i = 0; i < n; i += 4097)
      if (h[i] != 1000.0 * p + (double)i) ++bad;
  }
  printf("[rank %d] direct stores + flags: %s\n", rank, bad ? "MISMATCH" : "ok");
  barrier(sh, 1, nranks);
  // 2. copy-engine pushes into the peers' slots, timed
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  double *src;
  CK(hipMalloc(&src, sizeof(double) * n));
  hipLaunchKernelGGL(k_fill, dim3(512), dim3(256), 0, st, src, n, 5000.0 * rank);
  CK(hipStreamSynchronize(st));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int p = 0; p < nranks; ++p)
      CK(hipMemcpyAsync(pbuf[p] + (long long)rank * n, src, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("[rank %d] %d peer copies of %.0f MiB: %.3f ms (%.1f GB/s)\n", rank, nranks, n * 8.0 / 1048576, ms,
                    nranks * n * 8.0 / ms * 1e-6);
  }
  for (int p = 0; p < nranks; ++p) hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, st, pflag[p] + rank, 2ull);
  hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, st, flag, nranks, 2ull, status, (long long)clk_khz * 1000 * 20);
  CK(hipStreamSynchronize(st));
  bad = *status != 0;
  for (int p = 0; p < nranks && !bad; ++p) {
    CK(hipMemcpy(h.data(), buf + (long long)p * n, sizeof(double) * n, hipMemcpyDeviceToHost));
    for (long long i = 0; i < n; i += 4097)
      if (h[i] != 5000.0 * p + (double)i) ++bad;
  }
  printf("[rank %d] peer copies + flags: %s\n", rank, bad ? "MISMATCH" : "ok");
  // 3. hipStreamWaitValue64 on a peer-written flag
  for (int p = 0; p < nranks; ++p) hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, st, pflag[p] + rank, 3ull);
  hipError_t e = hipSuccess;
  for (int p = 0; p < nranks && e == hipSuccess; ++p) e = hipStreamWaitValue64(st, flag + p, 3ull, hipStreamWaitValueGte, ~0ull);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  printf("[rank %d] hipStreamWaitValue64 on device flags: %s\n", rank, e == hipSuccess ? "ok" : hipGetErrorString(e));
  barrier(sh, 2, nranks);
  for (int p = 0; p < nranks; ++p)
    if (p != rank) {
      CK(hipIpcCloseMemHandle(pbuf[p]));
      CK(hipIpcCloseMemHandle(pflag[p]));
    }
  barrier(sh, 3, nranks);
  CK(hipFree(buf));
  CK(hipFree(flag));
  return bad ? 5 : 0;
}

int main(int argc, char **argv) {
  // started once per rank by tools/ipc_probe.sh (a launcher that never touches the GPU)
  const char *r = getenv("MRL_PROBE_RANK");
  const int nranks = argc > 1 ? atoi(argv[1]) : 2;
  const int same = argc > 2 ? atoi(argv[2]) : 1;
  if (argc > 3) g_slot_doubles = (long long)atoi(argv[3]) * (1 << 17);
  if (argc > 4) g_second_mib = atoi(argv[4]);
  if (argc > 5) g_extra_gib = atoi(argv[5]);
  if (!r || !getenv("MRL_PROBE_SHM")) {
    fprintf(stderr, "run through tools/ipc_probe.sh\n");
    return 1;
  }
  return child(atoi(r), nranks, same, getenv("MRL_PROBE_SHM"));
}
