// Device reductions: the torch::sum / torch::norm call sites of the CG and the Newton loop
// (include/utils/MarlinUtils.h:63,82,92,99,109; src/tensor_computes/FFTMechanics.C:124,146) and
// DomainAction::average (src/actions/DomainAction.C:1558-1574).
//
// Two deterministic stages (no atomics): up to kRedBlocks workgroups write one partial each into
// ctx->d_red, a single workgroup folds the partials into a device scalar slot (ctx->d_red + kScalarBase).
// Scalars stay on the device for kernels that consume them (CG step sizes); the synchronous C ABI
// entry points copy them to pinned host memory.
#include "comm.h"
#include "mrl_internal.h"

namespace mrl {

// slab contexts with a communicator: the value over all ranks (the reference's norms / sums are serial-only,
// DomainAction.C:1564-1567).  op 0 sum, 1 min, 2 max.
static int global_values(mrl_ctx *ctx, double *h, int n, int op) {
  if (!ctx->comm || ctx->comm->nranks == 1) return MRL_OK;
  int rc = MRL_OK;
  for (int base = 0; base < n && rc == MRL_OK; base += 16)  // (the bootstrap all-reduce carries 16 values per call)
    rc = comm_allreduce_host(ctx->comm, h + base, n - base < 16 ? n - base : 16, op);
  if (rc != MRL_OK) set_error(ctx, rc, "%s", ctx->comm->err.c_str());
  return rc;
}

// OP 0: sum a ; 1: sum a*b ; 2: sum a*a
template <int OP>
__global__ void __launch_bounds__(256) k_reduce_partial(const double *__restrict__ a, const double *__restrict__ b,
                                                         long long n, double *__restrict__ partial) {
  __shared__ double sh[4];
  double acc = 0.0;
  const long long stride = (long long)gridDim.x * 256 * 2;
  long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 2;
  for (; i + 1 < n; i += stride) {
    const double2 x = *reinterpret_cast<const double2 *>(a + i);
    if (OP == 0) {
      acc += x.x + x.y;
    } else if (OP == 1) {
      const double2 y = *reinterpret_cast<const double2 *>(b + i);
      acc += x.x * y.x + x.y * y.y;
    } else {
      acc += x.x * x.x + x.y * x.y;
    }
  }
  if (i < n) {
    const double x = a[i];
    acc += (OP == 0) ? x : (OP == 1 ? x * b[i] : x * x);
  }
  const double r = block_sum256(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// fold `nb` partials of `nslots` interleaved quantities (partial[b*nslots + s]) into out[s]
// `mirror` (optional): the same values also go to pinned host memory, so that a host that needs them only has to wait for the
// stream -- no device-to-host copy command (one per CG iteration otherwise)
// `stop` (optional): non-zero = leave everything as it is (see k_reduce_final_cg)
__global__ void __launch_bounds__(256) k_reduce_final(const double *__restrict__ partial, int nb, int nslots,
                                                       double *__restrict__ out, double *__restrict__ mirror = nullptr,
                                                       const int *__restrict__ stop = nullptr) {
  __shared__ double sh[4];
  if (stop && *stop) return;
  for (int s = 0; s < nslots; ++s) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) acc += partial[(long long)i * nslots + s];
    const double r = block_sum256(acc, sh);
    if (threadIdx.x == 0) {
      out[s] = r;
      if (mirror) mirror[s] = r;
    }
  }
}

// The r.r of a conjugate-gradient iteration whose successor is enqueued BEFORE the host has seen this value (mech.hip: the host reads
// one iteration late, so the GPU never waits for it).  out[0] = r.r ; mirror[0] = r.r and mirror[1] = 1.0 / 0.0 = converged or not, in
// pinned host memory ; converged: sqrt(r.r) <= thr, the test of MarlinUtils.h:118-121 with the host's thr = l_tol * |b| (IEEE sqrt on
// both sides) -- then *stop = 1, and every kernel of the iterations already enqueued returns at once (k_pass, k_gamma_xfused, k_z_inv,
// k_gamma_z_fwd_tangent, k_cg_update, k_reduce_final): r, p, x and the scalars stay those of the converged iteration.
__global__ void __launch_bounds__(256) k_reduce_final_cg(const double *__restrict__ partial, int nb, double *__restrict__ out,
                                                          double *__restrict__ mirror, double thr, int *__restrict__ stop) {
  __shared__ double sh[4];
  if (*stop) return;
  double acc = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) acc += partial[i];
  const double r = block_sum256(acc, sh);
  if (threadIdx.x == 0) {
    const bool conv = sqrt(r) <= thr;
    out[0] = r;
    mirror[0] = r;
    mirror[1] = conv ? 1.0 : 0.0;
    if (conv) *stop = 1;
  }
}

// per-component sums of a value-major field [npts][ncomp]; partial[b*ncomp + c]
__global__ void __launch_bounds__(256) k_component_sums(const double *__restrict__ a, long long npts, int ncomp,
                                                         double *__restrict__ partial) {
  __shared__ double sh[4];
  double acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = 0.0;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npts; p += (long long)gridDim.x * 256) {
    const double *q = a + p * ncomp;
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c < ncomp) acc[c] += q[c];
  }
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    if (c < ncomp) {
      const double r = block_sum256(acc[c], sh);
      if (threadIdx.x == 0) partial[(long long)blockIdx.x * ncomp + c] = r;
    }
  }
}

// min and max: partial[2b], partial[2b+1]
__global__ void __launch_bounds__(256) k_minmax_partial(const double *__restrict__ a, long long n, double *__restrict__ partial) {
  __shared__ double smin[4], smax[4];
  double lo = INFINITY, hi = -INFINITY;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const double v = a[i];
    lo = fmin(lo, v);
    hi = fmax(hi, v);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = fmin(lo, __shfl_down(lo, off, 64));
    hi = fmax(hi, __shfl_down(hi, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    smin[threadIdx.x >> 6] = lo;
    smax[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = fmin(fmin(smin[0], smin[1]), fmin(smin[2], smin[3]));
    partial[2 * blockIdx.x + 1] = fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
  }
}

__global__ void __launch_bounds__(64) k_minmax_final(const double *__restrict__ partial, int nb, double *__restrict__ out) {
  double lo = INFINITY, hi = -INFINITY;
  for (int i = threadIdx.x; i < nb; i += 64) {
    lo = fmin(lo, partial[2 * i]);
    hi = fmax(hi, partial[2 * i + 1]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = fmin(lo, __shfl_down(lo, off, 64));
    hi = fmax(hi, __shfl_down(hi, off, 64));
  }
  if (threadIdx.x == 0) {
    out[0] = lo;
    out[1] = hi;
  }
}

static int red_blocks(long long n) {
  long long b = (n / 2 + 255) / 256;
  if (b > kRedBlocks) b = kRedBlocks;
  if (b < 1) b = 1;
  return (int)b;
}

// enqueue: d_scalar[0] = reduction result
int reduce_async(mrl_ctx *ctx, int op, const double *a, const double *b, long long n, double *d_scalar) {
  if (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) != 0)
    return set_error(ctx, MRL_ERR_INVALID, "reduction operands must be 16-byte aligned");
  const int nb = red_blocks(n);
  switch (op) {
    case 0: hipLaunchKernelGGL(k_reduce_partial<0>, dim3(nb), dim3(256), 0, ctx->stream, a, b, n, ctx->d_red); break;
    case 1: hipLaunchKernelGGL(k_reduce_partial<1>, dim3(nb), dim3(256), 0, ctx->stream, a, b, n, ctx->d_red); break;
    default: hipLaunchKernelGGL(k_reduce_partial<2>, dim3(nb), dim3(256), 0, ctx->stream, a, b, n, ctx->d_red); break;
  }
  hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, ctx->stream, ctx->d_red, nb, 1, d_scalar);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

// fold partials that another kernel left in ctx->d_red (nb blocks x nslots) into d_scalar[0..nslots)
int reduce_finalize(mrl_ctx *ctx, int nb, int nslots, double *d_scalar) {
  hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, ctx->stream, ctx->d_red, nb, nslots, d_scalar);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

// the same for partials in a caller-supplied buffer (any number of workgroups)
int reduce_finalize_from(mrl_ctx *ctx, const double *partial, int nb, double *d_scalar) {
  hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, ctx->stream, partial, nb, 1, d_scalar);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}
// ... that leaves d_scalar alone once *stop is set (a CG solve that has converged: k_reduce_final_cg)
int reduce_finalize_from_guarded(mrl_ctx *ctx, const double *partial, int nb, double *d_scalar, const int *stop) {
  hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, ctx->stream, partial, nb, 1, d_scalar, nullptr, stop);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

// r.r of a CG iteration with the convergence verdict taken on the device (k_reduce_final_cg); `slot` 0 / 1: which pair of the pinned
// mirror (ctx->h_red[8 + 2 slot], [9 + 2 slot]) receives {r.r, verdict}.  Nothing is waited for here.
int reduce_finalize_cg(mrl_ctx *ctx, int nb, double *d_scalar, int slot, double thr, int *stop) {
  if (!ctx->d_h_red) return set_error(ctx, MRL_ERR_UNSUPPORTED, "reduce_finalize_cg: the pinned scratch is not device-mapped");
  hipLaunchKernelGGL(k_reduce_final_cg, dim3(1), dim3(256), 0, ctx->stream, ctx->d_red, nb, d_scalar, ctx->d_h_red + 8 + 2 * slot, thr, stop);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

int read_scalars(mrl_ctx *ctx, const double *d_scalar, int count, double *h_out);
// reduce_finalize + the values on the host after the stream has drained (at most 64 slots): the final kernel writes them to the
// context's pinned, device-mapped scratch itself
int reduce_finalize_to_host(mrl_ctx *ctx, int nb, int nslots, double *d_scalar, double *h_out) {
  if (!ctx->d_h_red) {
    MRL_TRY(reduce_finalize(ctx, nb, nslots, d_scalar));
    return read_scalars(ctx, d_scalar, nslots, h_out);
  }
  hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, ctx->stream, ctx->d_red, nb, nslots, d_scalar, ctx->d_h_red);
  MRL_HIP(ctx, hipGetLastError());
  MRL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < nslots; ++i) h_out[i] = ctx->h_red[i];
  return MRL_OK;
}

// copy `count` device scalars starting at d_scalar to host (synchronises the stream)
int read_scalars(mrl_ctx *ctx, const double *d_scalar, int count, double *h_out) {
  MRL_HIP(ctx, hipMemcpyAsync(ctx->h_red, d_scalar, sizeof(double) * count, hipMemcpyDeviceToHost, ctx->stream));
  MRL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < count; ++i) h_out[i] = ctx->h_red[i];
  return MRL_OK;
}

int component_sums_async(mrl_ctx *ctx, const double *a, long long npts, int ncomp, double *d_scalar) {
  long long nb = (npts + 255) / 256;
  if (nb > kRedBlocks / 16) nb = kRedBlocks / 16;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(k_component_sums, dim3((unsigned)nb), dim3(256), 0, ctx->stream, a, npts, ncomp, ctx->d_red);
  hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, ctx->stream, ctx->d_red, (int)nb, ncomp, d_scalar);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

}  // namespace mrl

using namespace mrl;

static int sync_reduce(mrl_ctx *ctx, int op, const double *a, const double *b, int64_t n, double *h_out,
                       const char *what) {
  if (!ctx) return MRL_ERR_INVALID;
  if (!a || (op == 1 && !b) || n < 0 || !h_out) return set_error(ctx, MRL_ERR_INVALID, "%s: bad argument", what);
  if (n == 0) {
    *h_out = 0.0;
    return global_values(ctx, h_out, 1, 0);
  }
  double *slot = ctx->d_red + kScalarBase;
  MRL_TRY(reduce_async(ctx, op, a, op == 1 ? b : a, n, slot));
  MRL_TRY(read_scalars(ctx, slot, 1, h_out));
  return global_values(ctx, h_out, 1, 0);
}

extern "C" {

int mrl_dot(mrl_ctx *ctx, const double *d_a, const double *d_b, int64_t n, double *h_out) {
  return sync_reduce(ctx, 1, d_a, d_b, n, h_out, "mrl_dot");
}

int mrl_norm2(mrl_ctx *ctx, const double *d_a, int64_t n, double *h_out) {
  int rc = sync_reduce(ctx, 2, d_a, nullptr, n, h_out, "mrl_norm2");
  if (rc == MRL_OK) *h_out = sqrt(*h_out);
  return rc;
}

int mrl_sum(mrl_ctx *ctx, const double *d_a, int64_t n, double *h_out) {
  return sync_reduce(ctx, 0, d_a, nullptr, n, h_out, "mrl_sum");
}

/* TensorExtremeValuePostprocessor (src/postprocessors/TensorExtremeValuePostprocessor.C:30-44) */
int mrl_minmax(mrl_ctx *ctx, const double *d_a, int64_t n, double *h_min, double *h_max) {
  if (!ctx) return MRL_ERR_INVALID;
  if (!d_a || n < 1 || !h_min || !h_max) return set_error(ctx, MRL_ERR_INVALID, "mrl_minmax: bad argument");
  long long nb = (n + 255) / 256;
  if (nb > kRedBlocks / 2) nb = kRedBlocks / 2;
  double *slot = ctx->d_red + kScalarBase;
  hipLaunchKernelGGL(k_minmax_partial, dim3((unsigned)nb), dim3(256), 0, ctx->stream, d_a, (long long)n, ctx->d_red);
  hipLaunchKernelGGL(k_minmax_final, dim3(1), dim3(64), 0, ctx->stream, ctx->d_red, (int)nb, slot);
  MRL_HIP(ctx, hipGetLastError());
  double h[2];
  MRL_TRY(read_scalars(ctx, slot, 2, h));
  *h_min = h[0];
  *h_max = h[1];
  MRL_TRY(global_values(ctx, h_min, 1, 1));
  return global_values(ctx, h_max, 1, 2);
}

int mrl_average(mrl_ctx *ctx, const double *d_a, int64_t ncomp, double *h_out) {
  if (!ctx) return MRL_ERR_INVALID;
  if (!d_a || !h_out || ncomp < 1 || ncomp > 16)
    return set_error(ctx, MRL_ERR_INVALID, "mrl_average: need 1 <= ncomp <= 16");
  // slab contexts: the local sum over the GLOBAL point count, so that the sum over ranks is the average
  const long long npts = real_count_local(ctx);
  const double nglob = (double)ctx->n[0] * (double)ctx->n[1] * (double)ctx->n[2];
  double *slot = ctx->d_red + kScalarBase;
  MRL_TRY(component_sums_async(ctx, d_a, npts, (int)ncomp, slot));
  MRL_TRY(read_scalars(ctx, slot, (int)ncomp, h_out));
  for (int c = 0; c < ncomp; ++c) h_out[c] /= nglob;
  return global_values(ctx, h_out, (int)ncomp, 0);  // with a communicator: the global average
}

}  // extern "C"

// ---- TensorHistogram (src/vectorpostprocessors/TensorHistogram.C:48-79 -> at::native::histogramdd with explicit bin edges):
// bin i = [edge_i, edge_i+1), the last bin closed on the right, values outside [edge_0, edge_nbins] not counted
namespace mrl {

__global__ void __launch_bounds__(256) k_histogram(const double *__restrict__ a, long long n, const double *__restrict__ edges,
                                                    int nbins, unsigned long long *__restrict__ counts) {
  extern __shared__ unsigned long long sh_counts[];
  double *sh_edges = reinterpret_cast<double *>(sh_counts + nbins);
  for (int i = threadIdx.x; i < nbins; i += 256) sh_counts[i] = 0ull;
  for (int i = threadIdx.x; i <= nbins; i += 256) sh_edges[i] = edges[i];
  __syncthreads();
  const double lo = sh_edges[0], hi = sh_edges[nbins];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const double v = a[i];
    if (!(v >= lo && v <= hi)) continue;   // also drops NaN
    // upper_bound(edges, v) - 1, the right-most edge counted into the last bin
    int l = 0, r = nbins + 1;
    while (l < r) {
      const int m = (l + r) >> 1;
      if (sh_edges[m] <= v)
        l = m + 1;
      else
        r = m;
    }
    int b = l - 1;
    if (b >= nbins) b = nbins - 1;
    atomicAdd(&sh_counts[b], 1ull);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nbins; i += 256)
    if (sh_counts[i]) atomicAdd(&counts[i], sh_counts[i]);
}

}  // namespace mrl

extern "C" int mrl_histogram(mrl_ctx *ctx, const double *d_a, int64_t n, const double *h_edges, int nbins, int64_t *h_counts) {
  using namespace mrl;
  if (!ctx) return MRL_ERR_INVALID;
  if (!d_a || !h_edges || !h_counts || nbins < 1 || nbins > 4096 || n < 0)
    return set_error(ctx, MRL_ERR_INVALID, "mrl_histogram: bad argument (1 <= bins <= 4096)");
  for (int i = 0; i < nbins; ++i)
    if (!(h_edges[i] <= h_edges[i + 1])) return set_error(ctx, MRL_ERR_INVALID, "mrl_histogram: bin edges must not decrease");
  const size_t cb = sizeof(unsigned long long) * (size_t)nbins, eb = sizeof(double) * (size_t)(nbins + 1);
  MRL_TRY(ensure_work(ctx, 3, cb + eb + 16));
  unsigned long long *d_counts = reinterpret_cast<unsigned long long *>(ctx->d_work[3]);
  double *d_edges = reinterpret_cast<double *>(d_counts + nbins);
  MRL_HIP(ctx, hipMemsetAsync(d_counts, 0, cb, ctx->stream));
  MRL_HIP(ctx, hipMemcpyAsync(d_edges, h_edges, eb, hipMemcpyHostToDevice, ctx->stream));
  if (n > 0) {
    long long nb = (n + 255) / 256;
    if (nb > 1024) nb = 1024;
    if (cb + eb > 64 * 1024)  // 4096 bins need 65 544 bytes of dynamic LDS: above the 64 KiB default
      MRL_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_histogram), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(cb + eb)));
    hipLaunchKernelGGL(k_histogram, dim3((unsigned)nb), dim3(256), cb + eb, ctx->stream, d_a, (long long)n, d_edges, nbins, d_counts);
    MRL_HIP(ctx, hipGetLastError());
  }
  std::vector<unsigned long long> h((size_t)nbins);
  MRL_HIP(ctx, hipMemcpyAsync(h.data(), d_counts, cb, hipMemcpyDeviceToHost, ctx->stream));
  MRL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < nbins; ++i) h_counts[i] = (int64_t)h[i];
  return MRL_OK;
}
