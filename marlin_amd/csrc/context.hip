// Context, planning and the host-only helpers of the C ABI (mirrors DomainAction's set-up work).
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

#include "mrl_internal.h"

int g_mrl_trace = 0;

namespace mrl {

thread_local std::string g_create_error;
// one DEVICE per process (include/marlin_hip.h): the per-kernel attributes (dynamic LDS sizes) are set once per process on the
// device of the first context; a context on another device is refused instead of launching with unset attributes
static std::atomic<int> g_process_device{-1};

int set_error(const mrl_ctx *ctx, int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx)
    ctx->err = buf;
  else
    g_create_error = buf;
  return code;
}

int local_tab(mrl_ctx *ctx, int slot, void *base, size_t stride_bytes, cplx *const **out) {
  if (ctx->nranks > 64 || slot < 0 || slot > 7) return set_error(ctx, MRL_ERR_UNSUPPORTED, "pointer tables hold at most 64 ranks");
  if (!ctx->d_tabs) MRL_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_tabs), sizeof(char *) * 64 * 8));
  std::vector<size_t> off((size_t)ctx->nranks);
  for (int p = 0; p < ctx->nranks; ++p) off[p] = (size_t)p * stride_bytes;
  return local_tab_offsets(ctx, slot, base, off.data(), out);
}

struct TabOffsets {
  unsigned long long off[64];
};
__global__ void k_fill_tab_offsets(char **tab, char *base, TabOffsets o, int n) {
  if ((int)threadIdx.x < n) tab[threadIdx.x] = base + o.off[threadIdx.x];
}

// the same with explicit byte offsets per rank (chunks of different sizes: uneven partitions)
int local_tab_offsets(mrl_ctx *ctx, int slot, void *base, const size_t *byte_offsets, cplx *const **out) {
  if (ctx->nranks > 64 || slot < 0 || slot > 7) return set_error(ctx, MRL_ERR_UNSUPPORTED, "pointer tables hold at most 64 ranks");
  if (!ctx->d_tabs) MRL_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_tabs), sizeof(char *) * 64 * 8));
  char **t = ctx->d_tabs + 64 * slot;
  *out = reinterpret_cast<cplx *const *>(t);
  TabOffsets o{};
  bool same = ctx->tab_valid[slot] && ctx->tab_base[slot] == base;
  for (int p = 0; p < ctx->nranks; ++p) {
    o.off[p] = byte_offsets[p];
    same = same && ctx->tab_off[slot][p] == o.off[p];
  }
  if (same) return MRL_OK;  // (filled by an earlier launch on this stream from the same buffer: 5 us per call of the staged entry points)
  ctx->tab_valid[slot] = true;
  ctx->tab_base[slot] = base;
  for (int p = 0; p < ctx->nranks; ++p) ctx->tab_off[slot][p] = o.off[p];
  hipLaunchKernelGGL(k_fill_tab_offsets, dim3(1), dim3(64), 0, ctx->stream, t, static_cast<char *>(base), o, ctx->nranks);
  MRL_HIP(ctx, hipGetLastError());
  *out = reinterpret_cast<cplx *const *>(t);
  return MRL_OK;
}

int ensure_work(mrl_ctx *ctx, int slot, size_t bytes) {
  if (ctx->work_bytes[slot] >= bytes) return MRL_OK;
  if (ctx->d_work[slot]) {
    MRL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    MRL_HIP(ctx, hipFree(ctx->d_work[slot]));
    ctx->d_work[slot] = nullptr;
    ctx->work_bytes[slot] = 0;
  }
  void *p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess)
    return set_error(ctx, MRL_ERR_NOMEM, "hipMalloc of %zu scratch bytes failed", bytes);
  ctx->d_work[slot] = static_cast<double *>(p);
  ctx->work_bytes[slot] = bytes;
  return MRL_OK;
}

ProfScope::ProfScope(mrl_ctx *c, const char *name, double bytes) : ctx(c), slot(-1) {
  if (!c->profiling) return;
  for (size_t i = 0; i < c->prof.size(); ++i)
    if (c->prof[i].name == name || std::strcmp(c->prof[i].name, name) == 0) slot = (int)i;
  if (slot < 0) {
    c->prof.push_back(Profile{name, 0.0, 0, bytes});
    slot = (int)c->prof.size() - 1;
  }
  c->prof[slot].bytes = bytes;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipEventRecord(a, c->stream);
}

ProfScope::~ProfScope() {
  if (slot < 0) return;
  hipEventRecord(b, ctx->stream);
  ctx->prof_events.emplace_back(a, b);
  ctx->prof_slots.push_back(slot);
}

// radix sequence: 4s, then 2, then odd primes ascending (any n >= 1)
static std::vector<int> factorize(long long n) {
  std::vector<int> r;
  while (n % 4 == 0) {
    r.push_back(4);
    n /= 4;
  }
  while (n % 2 == 0) {
    r.push_back(2);
    n /= 2;
  }
  for (long long p = 3; p * p <= n; p += 2)
    while (n % p == 0) {
      r.push_back((int)p);
      n /= p;
    }
  if (n > 1) r.push_back((int)n);
  return r;
}

static int build_axis(mrl_ctx *ctx, AxisPlan &ax, long long n) {
  ax.n = (int)n;
  ax.radix = factorize(n);
  if ((int)ax.radix.size() > kMaxRadixPasses)
    return set_error(ctx, MRL_ERR_UNSUPPORTED, "too many radix passes for n=%lld", n);
  std::vector<cplx> tw(n);
  const long double two_pi = 6.283185307179586476925286766559005768L;
  for (long long k = 0; k < n; ++k) {
    const long double a = two_pi * (long double)k / (long double)n;
    tw[k] = make_double2((double)cosl(a), (double)(-sinl(a)));
  }
  MRL_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ax.d_tw), sizeof(cplx) * n));
  MRL_HIP(ctx, hipMemcpy(ax.d_tw, tw.data(), sizeof(cplx) * n, hipMemcpyHostToDevice));
  return MRL_OK;
}

// fp32 copies for the fp32 instantiation of the fused Cahn-Hilliard path (ch_fused_f32.hip), built on first use.
// Twiddles: the long-double table rounded once to float.  Reciprocal axis: the reference's sequence evaluated in float32 as libTorch
// does for a float32 run (MarlinUtils.C:39-44): fftfreq = arange * float(1 / (n d)), then * 2.0f, then * float(pi).
int axis_tw32(mrl_ctx *ctx, int axis) {
  AxisPlan &ax = ctx->ax[axis];
  if (ax.d_tw32) return MRL_OK;
  const long long n = ax.n;
  std::vector<float2> tw((size_t)n);
  const long double two_pi = 6.283185307179586476925286766559005768L;
  for (long long k = 0; k < n; ++k) {
    const long double a = two_pi * (long double)k / (long double)n;
    tw[k] = make_float2((float)cosl(a), (float)(-sinl(a)));
  }
  MRL_HIP(ctx, hipMalloc(&ax.d_tw32, sizeof(float2) * n));
  MRL_HIP(ctx, hipMemcpy(ax.d_tw32, tw.data(), sizeof(float2) * n, hipMemcpyHostToDevice));
  return MRL_OK;
}

int axis_k32(mrl_ctx *ctx, int axis) {
  AxisPlan &ax = ctx->ax[axis];
  if (ax.d_k32) return MRL_OK;
  const long long n = ctx->n[axis], cnt = ctx->nrec[axis];
  const bool rfft = (axis == 2) && ctx->spectrum == MRL_SPECTRUM_HALF;
  std::vector<float> k((size_t)cnt);
  const float scale = (float)(1.0 / ((double)n * ctx->dx[axis]));
  for (long long i = 0; i < cnt; ++i) {
    long long idx = i + ctx->kbeg[axis];
    if (!rfft && idx >= (n + 1) / 2) idx -= n;
    if (n == 1) idx = 0;
    const float f = (float)idx * scale;
    k[i] = f * 2.0f * (float)M_PI;
  }
  MRL_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ax.d_k32), sizeof(float) * cnt));
  MRL_HIP(ctx, hipMemcpy(ax.d_k32, k.data(), sizeof(float) * cnt, hipMemcpyHostToDevice));
  return MRL_OK;
}

// torch::fft::fftfreq / rfftfreq followed by `* 2.0 * pi` (DomainAction.C:284-293):
//   arange values times the scalar 1.0/(n*d), then *2.0, then *pi  -- same rounding sequence.
static void reciprocal_axis(long long n, double dx, bool rfft, std::vector<double> &out) {
  const double scale = 1.0 / ((double)n * dx);
  const long long cnt = rfft ? n / 2 + 1 : n;
  out.resize(cnt);
  for (long long i = 0; i < cnt; ++i) {
    long long idx = i;
    if (!rfft && i >= (n + 1) / 2) idx = i - n;
    const double f = (double)idx * scale;
    out[i] = f * 2.0 * M_PI;
  }
}

static int partition(long long total, int nranks, const int64_t *weights, std::vector<long long> &ns) {
  // DomainAction::partitionHepler (include/actions/DomainAction.h:247-280)
  ns.clear();
  long long remaining = 0;
  for (int r = 0; r < nranks; ++r) remaining += weights ? weights[r] : 1;
  for (int r = 0; r < nranks; ++r) {
    const long long w = weights ? weights[r] : 1;
    if (remaining == 0) return MRL_ERR_INVALID;
    long long nn = (total * w) / remaining;
    if (nn < 1) nn = 1;
    ns.push_back(nn);
    remaining -= w;
    if (total < nn) return MRL_ERR_INVALID;
    total -= nn;
  }
  ns.back() += total;
  return MRL_OK;
}

// DomainAction::partitionPencils, the choice of the process grid (DomainAction.C:574-618)
static bool pencil_factors(int nranks, long long nx, long long ny, long long nz, int *py_out, int *pz_out) {
  const long long nxc = nx / 2 + 1;
  auto can_use = [&](long long px, long long pz) {
    if (px < 2 || pz < 2) return false;
    if (px > ny || px > nxc) return false;
    if (pz > nz || pz > ny) return false;
    return true;
  };
  bool found = false;
  long long best_px = 0, best_pz = 0, best_cost = 0;
  auto consider = [&](long long px, long long pz) {
    if (!can_use(px, pz)) return;
    const long long cost = px > pz ? px - pz : pz - px;
    if (!found || cost < best_cost) {
      best_px = px;
      best_pz = pz;
      best_cost = cost;
      found = true;
    }
  };
  long long max_divisor = (long long)std::sqrt((double)nranks);
  if (max_divisor < 2) max_divisor = 2;
  for (long long d = 2; d <= max_divisor; ++d)
    if (nranks % d == 0) {
      consider(d, nranks / d);
      consider(nranks / d, d);
    }
  if (!found) return false;
  *py_out = (int)best_px;
  *pz_out = (int)best_pz;
  return true;
}

}  // namespace mrl

using namespace mrl;

extern "C" {

int mrl_abi_version(void) { return MRL_ABI_VERSION; }

int mrl_pencil_factors(int32_t nranks, const int64_t n[3], int32_t *py, int32_t *pz) {
  if (!n || !py || !pz || nranks < 1) return set_error(nullptr, MRL_ERR_INVALID, "mrl_pencil_factors: bad argument");
  int a = 0, b = 0;
  if (!pencil_factors(nranks, n[0], n[1], n[2], &a, &b))
    return set_error(nullptr, MRL_ERR_INVALID,
                     "FFT_PENCIL requires factoring the number of MPI ranks into two integers greater than one that fit the domain "
                     "(ranks = %d). Use FFT_SLAB or adjust the rank count.", nranks);
  *py = a;
  *pz = b;
  return MRL_OK;
}

int mrl_pencil_layout(int32_t nranks, int32_t rank, const int64_t n[3], int64_t real_n[3], int64_t real_begin[3], int64_t recip_n[3],
                      int64_t recip_begin[3], int64_t *s1s, int64_t *s1r, int64_t *s2s, int64_t *s2r) {
  if (!n || nranks < 1 || rank < 0 || rank >= nranks) return set_error(nullptr, MRL_ERR_INVALID, "mrl_pencil_layout: bad argument");
  int Py = 0, Pz = 0;
  if (!pencil_factors(nranks, n[0], n[1], n[2], &Py, &Pz))
    return set_error(nullptr, MRL_ERR_INVALID,
                     "FFT_PENCIL requires factoring the number of MPI ranks into two integers greater than one that fit the domain "
                     "(ranks = %d). Use FFT_SLAB or adjust the rank count.", nranks);
  std::vector<long long> y, z, kx, ky;
  if (partition(n[1], Py, nullptr, y) != MRL_OK || partition(n[2], Pz, nullptr, z) != MRL_OK ||
      partition(n[0] / 2 + 1, Py, nullptr, kx) != MRL_OK || partition(n[1], Pz, nullptr, ky) != MRL_OK)
    return set_error(nullptr, MRL_ERR_INVALID, "Internal partitioning error.");
  auto begin_of = [](const std::vector<long long> &v, int i) {
    long long b = 0;
    for (int r = 0; r < i; ++r) b += v[r];
    return b;
  };
  const int px = rank % Py, pz = rank / Py;
  const long long nyl = y[px], nzl = z[pz], kxl = kx[px], kyl = ky[pz];
  if (real_n) { real_n[0] = n[0]; real_n[1] = nyl; real_n[2] = nzl; }
  if (real_begin) { real_begin[0] = 0; real_begin[1] = begin_of(y, px); real_begin[2] = begin_of(z, pz); }
  if (recip_n) { recip_n[0] = kxl; recip_n[1] = kyl; recip_n[2] = n[2]; }
  if (recip_begin) { recip_begin[0] = begin_of(kx, px); recip_begin[1] = begin_of(ky, pz); recip_begin[2] = 0; }
  for (int p = 0; p < nranks; ++p) {
    if (s1s) s1s[p] = 0;
    if (s1r) s1r[p] = 0;
    if (s2s) s2s[p] = 0;
    if (s2r) s2r[p] = 0;
  }
  for (int q = 0; q < Py; ++q) {   // stage 1: my kx chunk for px' = q  <->  their y blocks
    const int peer = pz * Py + q;
    if (s1s) s1s[peer] = kx[q] * nyl * nzl;
    if (s1r) s1r[peer] = kxl * y[q] * nzl;
  }
  for (int q = 0; q < Pz; ++q) {   // stage 2: my ky chunk for pz' = q  <->  their z blocks
    const int peer = q * Py + px;
    if (s2s) s2s[peer] = kxl * ky[q] * nzl;
    if (s2r) s2r[peer] = kxl * kyl * z[q];
  }
  return MRL_OK;
}

int mrl_pencil_grid(const mrl_ctx *ctx, int32_t *py, int32_t *pz) {
  if (!ctx) return MRL_ERR_INVALID;
  if (!ctx->pencil) return set_error(ctx, MRL_ERR_INVALID, "mrl_pencil_grid: not a pencil context");
  if (py) *py = ctx->pen_py;
  if (pz) *pz = ctx->pen_pz;
  return MRL_OK;
}

const char *mrl_last_error(const mrl_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int mrl_reciprocal_axis(int64_t n, double dx, int rfft, double *h_out) {
  if (n < 1 || !h_out || !(dx > 0)) return set_error(nullptr, MRL_ERR_INVALID, "mrl_reciprocal_axis: bad argument");
  std::vector<double> v;
  reciprocal_axis(n, dx, rfft != 0, v);
  std::memcpy(h_out, v.data(), v.size() * sizeof(double));
  return MRL_OK;
}

int mrl_partition(int64_t total, int32_t nranks, const int64_t *weights, int64_t *h_counts) {
  if (nranks < 1 || !h_counts || total < nranks)
    return set_error(nullptr, MRL_ERR_INVALID, "mrl_partition: need total >= nranks >= 1");
  std::vector<long long> ns;
  if (partition(total, nranks, weights, ns) != MRL_OK)
    return set_error(nullptr, MRL_ERR_INVALID, "Internal partitioning error.");
  for (int r = 0; r < nranks; ++r) h_counts[r] = ns[r];
  return MRL_OK;
}

int mrl_ctx_create(mrl_ctx **out, const mrl_domain *dom) {
  if (!out || !dom) return set_error(nullptr, MRL_ERR_INVALID, "mrl_ctx_create: null argument");
  *out = nullptr;
  if (dom->dim < 1 || dom->dim > 3) return set_error(nullptr, MRL_ERR_INVALID, "Unsupported mesh dimension %d", dom->dim);
  if (dom->nranks < 1 || dom->rank < 0 || dom->rank >= dom->nranks)
    return set_error(nullptr, MRL_ERR_INVALID, "invalid rank %d of %d", dom->rank, dom->nranks);
  const bool pencil = (dom->flags & MRL_FLAG_PENCIL) != 0;
  int pen_py = 1, pen_pz = 1;
  if (pencil) {
    if (dom->dim < 3) return set_error(nullptr, MRL_ERR_INVALID, "Dimension must be 3 for pencil decomposition.");  // DomainAction.C:571-572
    if (dom->spectrum != MRL_SPECTRUM_HALF) return set_error(nullptr, MRL_ERR_UNSUPPORTED, "FFT_PENCIL needs spectrum = MRL_SPECTRUM_HALF (r2c along x)");
    if (dom->weights) return set_error(nullptr, MRL_ERR_UNSUPPORTED, "FFT_PENCIL partitions with equal weights (DomainAction.C:633-637)");
    if (dom->nranks > 64) return set_error(nullptr, MRL_ERR_UNSUPPORTED, "at most 64 ranks");
    if (!pencil_factors(dom->nranks, dom->n[0], dom->n[1], dom->n[2], &pen_py, &pen_pz))
      return set_error(nullptr, MRL_ERR_INVALID,
                       "FFT_PENCIL requires factoring the number of MPI ranks into two integers greater than one that fit the domain "
                       "(ranks = %d). Use FFT_SLAB or adjust the rank count.", dom->nranks);
  }
  const bool slab = !pencil && (dom->nranks > 1 || (dom->flags & MRL_FLAG_SLAB));
  if (slab && dom->dim < 2)
    return set_error(nullptr, MRL_ERR_INVALID, "Dimension must be 2 or 3 for slab decomposition.");
  if (slab && dom->dim == 2 && dom->spectrum != MRL_SPECTRUM_FULL)
    return set_error(nullptr, MRL_ERR_UNSUPPORTED, "2-D slab decomposition needs spectrum = MRL_SPECTRUM_FULL");
  for (int d = 0; d < dom->dim; ++d) {
    if (dom->n[d] < 1) return set_error(nullptr, MRL_ERR_INVALID, "grid size must be positive");
    if (!(dom->max[d] > dom->min[d]))
      return set_error(nullptr, MRL_ERR_INVALID, "Max coordinate must be larger than the min coordinate in every dimension");
  }
  mrl_ctx *c = new (std::nothrow) mrl_ctx();
  if (!c) return set_error(nullptr, MRL_ERR_NOMEM, "out of host memory");

  int rc = MRL_OK;
  auto fail = [&](int code) {
    g_create_error = c->err;
    mrl_ctx_destroy(c);
    return code;
  };

  c->dim = dom->dim;
  c->spectrum = dom->spectrum;
  c->nranks = dom->nranks;
  c->rank = dom->rank;
  c->slab = slab;
  c->pencil = pencil;
  c->pen_py = pen_py;
  c->pen_pz = pen_pz;
  // internal axes: serial contexts right-align the user axes (the r2c axis is always A2); slab
  // contexts left-align them so that x = A0 is the reciprocal split axis and y = A1 the real-space
  // split axis in 2-D and 3-D alike (a 2-D slab domain is [nx][ny][1]).
  c->off = (c->slab || c->pencil) ? 0 : 3 - c->dim;
  const int off = c->off;
  for (int a = 0; a < 3; ++a) {
    c->n[a] = 1;
    c->gmin[a] = 0.0;
    c->gmax[a] = 1.0;
    c->dx[a] = 1.0;
  }
  for (int d = 0; d < c->dim; ++d) {
    const int a = d + off;
    c->n[a] = dom->n[d];
    c->gmin[a] = dom->min[d];
    c->gmax[a] = dom->max[d];
    c->dx[a] = (dom->max[d] - dom->min[d]) / (double)dom->n[d];  // DomainAction.C:241
  }

  if (dom->device >= 0) {
    if (hipSetDevice(dom->device) != hipSuccess) {
      set_error(c, MRL_ERR_HIP, "hipSetDevice(%d) failed", dom->device);
      return fail(MRL_ERR_HIP);
    }
    c->device = dom->device;
  } else if (hipGetDevice(&c->device) != hipSuccess) {
    set_error(c, MRL_ERR_HIP, "no HIP device available (the HIP path has no CPU fallback)");
    return fail(MRL_ERR_HIP);
  }
  {
    int expected = -1;
    if (!g_process_device.compare_exchange_strong(expected, c->device) && expected != c->device) {
      set_error(c, MRL_ERR_UNSUPPORTED, "this process already runs contexts on device %d: one device per process (one MPI rank <-> one device, DomainAction.C:197-198)", expected);
      return fail(MRL_ERR_UNSUPPORTED);
    }
  }
  if (!(dom->flags & MRL_FLAG_OWN_STREAM)) {
    c->stream = static_cast<hipStream_t>(dom->stream);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
      set_error(c, MRL_ERR_HIP, "no HIP device available (the HIP path has no CPU fallback)");
      return fail(MRL_ERR_HIP);
    }
  } else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
      set_error(c, MRL_ERR_HIP, "hipStreamCreate failed (no usable GPU?)");
      return fail(MRL_ERR_HIP);
    }
    c->own_stream = true;
  }

  // reciprocal axes (global)
  for (int a = 0; a < 3; ++a) {
    if (a < off || a >= off + c->dim) {
      c->h_k[a] = {0.0};  // DomainAction.C:295-296
      c->nrec_glob[a] = 1;
    } else {
      // r2c axis: the last one (parallel_mode NONE, and this library's slab mode); FFT_PENCIL: x (DomainAction.C:282-284)
      const bool rfft = c->pencil ? (a == 0) : ((a == 2) && c->spectrum == MRL_SPECTRUM_HALF);
      reciprocal_axis(c->n[a], c->dx[a], rfft, c->h_k[a]);
      c->nrec_glob[a] = (long long)c->h_k[a].size();
    }
  }

  // real-space axes: torch::linspace(min + dx/2, max - dx/2, n) (DomainAction.C:246-251); ATen fills the lower half
  // as start + step*i and the upper half as end - step*(n-1-i)
  for (int a = 0; a < 3; ++a) {
    if (a < off || a >= off + c->dim) {
      c->h_x[a] = {0.0};
    } else {
      const long long n = c->n[a];
      const double lo = c->gmin[a] + c->dx[a] / 2.0, hi = c->gmax[a] - c->dx[a] / 2.0;
      const double step = n > 1 ? (hi - lo) / (double)(n - 1) : 0.0;
      c->h_x[a].resize(n);
      for (long long i = 0; i < n; ++i) c->h_x[a][i] = (i < n / 2) ? lo + step * (double)i : hi - step * (double)(n - 1 - i);
    }
  }

  // partition
  for (int a = 0; a < 3; ++a) {
    c->nloc[a] = c->n[a];
    c->rbeg[a] = 0;
    c->nrec[a] = c->nrec_glob[a];
    c->kbeg[a] = 0;
  }
  if (c->slab) {
    c->split_recip_axis = off + 0;  // x: DomainAction.C:519-520
    c->split_real_axis = off + 1;   // y: DomainAction.C:522-523
    if (c->n[c->split_recip_axis] < c->nranks || c->n[c->split_real_axis] < c->nranks) {
      set_error(c, MRL_ERR_INVALID, "slab decomposition needs at least one layer per rank");
      return fail(MRL_ERR_INVALID);
    }
    if (partition(c->nrec_glob[c->split_recip_axis], c->nranks, dom->weights, c->part_recip) != MRL_OK ||
        partition(c->n[c->split_real_axis], c->nranks, dom->weights, c->part_real) != MRL_OK) {
      set_error(c, MRL_ERR_INVALID, "Internal partitioning error.");
      return fail(MRL_ERR_INVALID);
    }
    long long b = 0;
    for (int r = 0; r < c->rank; ++r) b += c->part_real[r];
    c->rbeg[c->split_real_axis] = b;
    c->nloc[c->split_real_axis] = c->part_real[c->rank];
    b = 0;
    for (int r = 0; r < c->rank; ++r) b += c->part_recip[r];
    c->kbeg[c->split_recip_axis] = b;
    c->nrec[c->split_recip_axis] = c->part_recip[c->rank];
  }

  if (c->pencil) {  // DomainAction::partitionPencils (DomainAction.C:620-698)
    const int py = c->rank % c->pen_py, pz = c->rank / c->pen_py;
    if (partition(c->n[1], c->pen_py, nullptr, c->pen_y) != MRL_OK || partition(c->n[2], c->pen_pz, nullptr, c->pen_z) != MRL_OK ||
        partition(c->nrec_glob[0], c->pen_py, nullptr, c->pen_kx) != MRL_OK || partition(c->nrec_glob[1], c->pen_pz, nullptr, c->pen_ky) != MRL_OK) {
      set_error(c, MRL_ERR_INVALID, "Internal partitioning error.");
      return fail(MRL_ERR_INVALID);
    }
    auto begin_of = [](const std::vector<long long> &v, int i) {
      long long b = 0;
      for (int r = 0; r < i; ++r) b += v[r];
      return b;
    };
    c->nloc[1] = c->pen_y[py];
    c->rbeg[1] = begin_of(c->pen_y, py);
    c->nloc[2] = c->pen_z[pz];
    c->rbeg[2] = begin_of(c->pen_z, pz);
    c->nrec[0] = c->pen_kx[py];   // (px = rank % py partitions, :689-693)
    c->kbeg[0] = begin_of(c->pen_kx, py);
    c->nrec[1] = c->pen_ky[pz];   // (py_final = rank / py partitions, :690-697)
    c->kbeg[1] = begin_of(c->pen_ky, pz);
  }

  for (int a = 0; a < 3 && rc == MRL_OK; ++a) {
    rc = build_axis(c, c->ax[a], c->n[a]);
    if (rc != MRL_OK) break;
    const size_t bytes = sizeof(double) * c->nrec[a];
    if (hipMalloc(reinterpret_cast<void **>(&c->d_k[a]), bytes) != hipSuccess ||
        hipMemcpy(c->d_k[a], c->h_k[a].data() + c->kbeg[a], bytes, hipMemcpyHostToDevice) != hipSuccess) {
      set_error(c, MRL_ERR_HIP, "uploading reciprocal axis failed");
      rc = MRL_ERR_HIP;
    }
    const size_t xb = sizeof(double) * c->nloc[a];
    if (rc == MRL_OK && (hipMalloc(reinterpret_cast<void **>(&c->d_x[a]), xb) != hipSuccess ||
                         hipMemcpy(c->d_x[a], c->h_x[a].data() + c->rbeg[a], xb, hipMemcpyHostToDevice) != hipSuccess)) {
      set_error(c, MRL_ERR_HIP, "uploading real-space axis failed");
      rc = MRL_ERR_HIP;
    }
  }
  if (rc != MRL_OK) return fail(rc);

  // solver-private spectral layout of the fused serial path (mrl_ch_spec_elems): x planes padded to an odd number of 256-byte pieces
  if (!c->slab && !c->pencil && c->dim == 3 && !(dom->flags & MRL_FLAG_DENSE_SPECTRA) && fast_path_ok(c) && (c->n[0] * c->n[1]) % 2 == 0) {
    const long long inner = c->n[1] * c->nrec[2];
    long long plane = (inner + 15) / 16 * 16;
    if ((plane / 16) % 2 == 0) plane += 16;
    // (arrays of 4 GiB and more keep the dense layout: their 64-bit offset variants are rare enough not to be doubled)
    if (16.0 * (double)c->n[0] * (double)plane < 4294967296.0) c->spec_plane = plane;
  }

  if (hipMalloc(reinterpret_cast<void **>(&c->d_red), sizeof(double) * 4096) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void **>(&c->h_red), sizeof(double) * 64) != hipSuccess ||
      hipEventCreate(&c->ev_start) != hipSuccess || hipEventCreate(&c->ev_stop) != hipSuccess) {
    set_error(c, MRL_ERR_HIP, "allocating reduction scratch / events failed");
    return fail(MRL_ERR_HIP);
  }
  {  // device-side address of the pinned scratch (hipHostMalloc memory is mapped by default); optional
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, c->h_red, 0) == hipSuccess) c->d_h_red = static_cast<double *>(dp);
    else (void)hipGetLastError();
  }
  *out = c;
  return MRL_OK;
}

void mrl_ctx_destroy(mrl_ctx *c) {
  if (!c) return;
  if (c->stream) hipStreamSynchronize(c->stream);
  slab_pipes_destroy(c);  // (collective on a communicator with several ranks; a communicator destroyed first has done it already)
  slab_detach_comm(c);
  if (c->d_tabs) hipFree(c->d_tabs);
  for (auto &t : c->slab_tabs)
    if (t.d) hipFree(t.d);
  for (int a = 0; a < 3; ++a) {
    if (c->ax[a].d_tw) hipFree(c->ax[a].d_tw);
    if (c->ax[a].d_tw32) hipFree(c->ax[a].d_tw32);
    if (c->ax[a].d_k32) hipFree(c->ax[a].d_k32);
    if (c->d_k[a]) hipFree(c->d_k[a]);
    if (c->d_x[a]) hipFree(c->d_x[a]);
  }
  for (int s = 0; s < kWorkSlots; ++s)
    if (c->d_work[s]) hipFree(c->d_work[s]);
  if (c->d_red) hipFree(c->d_red);
  if (c->h_red) hipHostFree(c->h_red);
  if (c->ev_start) hipEventDestroy(c->ev_start);
  for (int e = 0; e < 2; ++e)
    if (c->cg_ev[e]) hipEventDestroy(c->cg_ev[e]);
  for (auto &t : c->tab_ring) {
    if (t.h) hipHostFree(t.h);
    if (t.d) hipFree(t.d);
    if (t.done) hipEventDestroy(t.done);
  }
  if (c->ev_stop) hipEventDestroy(c->ev_stop);
  for (auto &p : c->prof_events) {
    hipEventDestroy(p.first);
    hipEventDestroy(p.second);
  }
  if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
  delete c;
}

int mrl_sync(mrl_ctx *ctx) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return slab_comm_check(ctx);  // a device-side wait that timed out surfaces here
}

int mrl_ctx_set_option(mrl_ctx *ctx, int option, int64_t value) {
  if (!ctx) return MRL_ERR_INVALID;
  switch (option) {
    case MRL_OPT_EXPERIMENT:
      if ((value & 2048) && ctx->spec_plane)
        return set_error(ctx, MRL_ERR_UNSUPPORTED, "experiment 2048 (any-length path on a planned shape) needs a context created with MRL_FLAG_DENSE_SPECTRA");
      ctx->exp = (int)value;
      g_mrl_trace = (value & (1 << 20)) ? 1 : 0;
      return MRL_OK;
    case MRL_OPT_SLAB_NSUB:
      if (value < 1 || value > 64) return set_error(ctx, MRL_ERR_INVALID, "MRL_OPT_SLAB_NSUB must be in 1..64");
      ctx->opt_nsub = (int)value;
      return MRL_OK;
    case MRL_OPT_SLAB_CARRY: ctx->opt_carry = value != 0; return MRL_OK;
    case MRL_OPT_CACHE_CHUNK_MB:
      if (value < 0 || value > 1024) return set_error(ctx, MRL_ERR_INVALID, "MRL_OPT_CACHE_CHUNK_MB must be 0 (off) or 1..1024");
      ctx->opt_chunk_mb = (int)value;
      return MRL_OK;
    case MRL_OPT_VERIFY_EXCHANGE: ctx->opt_verify = value != 0; return MRL_OK;
    case MRL_OPT_VERIFY_MISMATCHES:  // (only 0 may be stored: resets the counter)
      if (value != 0) return set_error(ctx, MRL_ERR_INVALID, "MRL_OPT_VERIFY_MISMATCHES can only be reset to 0");
      return slab_verify_count(ctx, true) < 0 ? MRL_ERR_HIP : MRL_OK;
    default: return set_error(ctx, MRL_ERR_INVALID, "unknown option %d", option);
  }
}

int64_t mrl_ctx_get_option(const mrl_ctx *ctx, int option) {
  if (!ctx) return 0;
  switch (option) {
    case MRL_OPT_EXPERIMENT: return ctx->exp;
    case MRL_OPT_SLAB_NSUB: return ctx->opt_nsub;
    case MRL_OPT_SLAB_CARRY: return ctx->opt_carry;
    case MRL_OPT_CACHE_CHUNK_MB: return ctx->opt_chunk_mb;
    case MRL_OPT_VERIFY_EXCHANGE: return ctx->opt_verify;
    case MRL_OPT_VERIFY_MISMATCHES: return slab_verify_count(const_cast<mrl_ctx *>(ctx), false);
    default: return 0;
  }
}

int64_t mrl_ch_spec_elems(const mrl_ctx *ctx) {
  if (!ctx) return 0;
  if (ctx->slab && ctx->dim == 3) return ctx->nrec[0] * ctx->nrec[1] * mrl_slab_ch_spec_pitch(ctx);
  if (ctx->spec_plane) return ctx->n[0] * ctx->spec_plane;
  return ctx->nrec[0] * ctx->nrec[1] * ctx->nrec[2];
}

int mrl_ch_spec_layout(const mrl_ctx *ctx, int64_t *plane_pitch, int64_t *row_pitch) {
  if (!ctx) return MRL_ERR_INVALID;
  // (internal axes: serial contexts right-align the user axes, so the last two internal axes are always the rows of a plane)
  int64_t row = ctx->nrec[2];
  if (ctx->slab && ctx->dim == 3) row = mrl_slab_ch_spec_pitch(ctx);
  if (row_pitch) *row_pitch = row;
  if (plane_pitch) *plane_pitch = ctx->spec_plane ? ctx->spec_plane : ctx->nrec[1] * row;
  return MRL_OK;
}

int mrl_set_stream(mrl_ctx *ctx, void *stream) {
  if (!ctx) return MRL_ERR_INVALID;
  if (ctx->own_stream && ctx->stream) {
    hipStreamSynchronize(ctx->stream);
    hipStreamDestroy(ctx->stream);
    ctx->own_stream = false;
  }
  ctx->stream = static_cast<hipStream_t>(stream);
  for (bool &v : ctx->tab_valid) v = false;  // (the pointer tables were filled on the previous stream)
  return MRL_OK;
}

int mrl_local_shape(const mrl_ctx *ctx, int64_t real_n[3], int64_t real_begin[3], int64_t recip_n[3],
                    int64_t recip_begin[3]) {
  if (!ctx) return MRL_ERR_INVALID;
  const int off = ctx->off;
  for (int d = 0; d < 3; ++d) {
    const bool act = d < ctx->dim;
    if (real_n) real_n[d] = act ? ctx->nloc[d + off] : 1;
    if (real_begin) real_begin[d] = act ? ctx->rbeg[d + off] : 0;
    if (recip_n) recip_n[d] = act ? ctx->nrec[d + off] : 1;
    if (recip_begin) recip_begin[d] = act ? ctx->kbeg[d + off] : 0;
  }
  return MRL_OK;
}

int mrl_ctx_reciprocal_axis(const mrl_ctx *ctx, int axis, double *h_out, int64_t cap) {
  if (!ctx || axis < 0 || axis >= ctx->dim || !h_out) return set_error(ctx, MRL_ERR_INVALID, "bad axis");
  const int a = axis + ctx->off;
  if (cap < ctx->nrec[a]) return set_error(ctx, MRL_ERR_INVALID, "output capacity too small");
  std::memcpy(h_out, ctx->h_k[a].data() + ctx->kbeg[a], sizeof(double) * ctx->nrec[a]);
  return MRL_OK;
}

int mrl_timer_start(mrl_ctx *ctx) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_HIP(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
  return MRL_OK;
}

int mrl_timer_stop(mrl_ctx *ctx, float *h_ms) {
  if (!ctx || !h_ms) return MRL_ERR_INVALID;
  MRL_HIP(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
  MRL_HIP(ctx, hipEventSynchronize(ctx->ev_stop));
  MRL_HIP(ctx, hipEventElapsedTime(h_ms, ctx->ev_start, ctx->ev_stop));
  return MRL_OK;
}

int mrl_set_profiling(mrl_ctx *ctx, int on) {
  if (!ctx) return MRL_ERR_INVALID;
  ctx->profiling = on != 0;
  if (on) {
    for (auto &p : ctx->prof) {
      p.ms = 0;
      p.launches = 0;
    }
  }
  return MRL_OK;
}

int mrl_get_profile(mrl_ctx *ctx, int slot, const char **name, double *total_ms, int64_t *launches,
                    double *bytes_per_launch) {
  if (!ctx) return MRL_ERR_INVALID;
  // fold pending event pairs
  if (!ctx->prof_events.empty()) {
    MRL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < ctx->prof_events.size(); ++i) {
      float ms = 0.f;
      hipEventElapsedTime(&ms, ctx->prof_events[i].first, ctx->prof_events[i].second);
      ctx->prof[ctx->prof_slots[i]].ms += ms;
      ctx->prof[ctx->prof_slots[i]].launches += 1;
      hipEventDestroy(ctx->prof_events[i].first);
      hipEventDestroy(ctx->prof_events[i].second);
    }
    ctx->prof_events.clear();
    ctx->prof_slots.clear();
  }
  if (slot < 0 || slot >= (int)ctx->prof.size()) return MRL_ERR_INVALID;
  if (name) *name = ctx->prof[slot].name;
  if (total_ms) *total_ms = ctx->prof[slot].ms;
  if (launches) *launches = ctx->prof[slot].launches;
  if (bytes_per_launch) *bytes_per_launch = ctx->prof[slot].bytes;
  return MRL_OK;
}

int mrl_get_timing(mrl_ctx *ctx, mrl_timing *out) {
  if (!ctx || !out) return MRL_ERR_INVALID;
  *out = mrl_timing{};
  const int rc = mrl_get_profile(ctx, 0, nullptr, nullptr, nullptr, nullptr);  // folds the pending event pairs
  if (rc != MRL_OK && !ctx->prof.empty()) return rc;
  for (const auto &s : ctx->prof) {
    if (s.launches <= 0) continue;
    out->kernel_classes += 1;
    out->launches += s.launches;
    out->device_ms += s.ms;
    out->algorithmic_bytes += s.bytes * (double)s.launches;
    if (!out->dominant || s.ms > out->dominant_ms) {
      out->dominant = s.name;
      out->dominant_ms = s.ms;
    }
  }
  return MRL_OK;
}

}  // extern "C"
