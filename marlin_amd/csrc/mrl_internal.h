// Internal declarations shared by the host planner and the HIP kernels (gfx950 only).
#pragma once
#include "mrl_trace.h"
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "marlin_hip.h"

struct mrl_comm;

namespace mrl {

typedef double2 cplx;  // interleaved (re, im) complex128
struct SlabPipes;

constexpr int kMaxRadixPasses = 24;
constexpr int kWorkSlots = 24;
constexpr int kRedBlocks = 2048;    // partial sums of a reduction live in ctx->d_red[0, kRedBlocks)
constexpr int kScalarBase = 3072;   // device scalar slots: ctx->d_red[kScalarBase, 4096)

// One generic Stockham pass sequence along one axis (see fft_generic.hip).
struct PassDesc {
  int n;                      // transform length
  int npass;                  // number of radix passes
  int radix[kMaxRadixPasses];
  int sign;                   // -1 forward, +1 inverse
  // line enumeration: line L in [0, outer*inner): o = L / inner, i = L % inner
  long long inner, outer;
  long long in_so, in_si, in_sn;     // element offsets (in units of the input scalar kind)
  long long out_so, out_si, out_sn;
  long long in_sb, out_sb;           // batch strides (blockIdx.y)
  int in_kind;                // 0 complex, 1 real (imag = 0), 2 hermitian half line (n/2+1 stored)
  int out_kind;               // 0 complex, 2 real part
  int nout;                   // number of output elements stored per line (<= n)
  double scale;               // applied on store
  int tile;                   // lines per workgroup
  int lines_fastest;          // thread->(line,element) mapping for global access
  int pair;                   // real <-> half-spectrum passes with inner == 1: one complex transform carries lines 2L and 2L+1
  long long nreal_lines;      // (pair) number of real lines; the last pair may lack its second line
};

struct AxisPlan {
  int n = 1;
  std::vector<int> radix;
  cplx *d_tw = nullptr;  // exp(-2 pi i k / n), k = 0..n-1
  void *d_tw32 = nullptr;  // the same as float2 (fp32 instantiation of the fused Cahn-Hilliard path), built on first use
  float *d_k32 = nullptr;  // LOCAL reciprocal axis as float, likewise
};

// Device tables of the table-driven slab pipeline (slab_fused.hip: partitions that are not equal powers of two), one set per row
// pitch kp of the exchange layouts; element units, 32 bits
struct SlabTabs {
  long long kp = -1;
  int dense = -1;           // MRL_OPT_EXPERIMENT bit 1 << 23 when the set was built
  int nf = 0;               // 0: Cahn-Hilliard layouts (padded x planes; yA2 / yA1 = two / one field per forward chunk);
                            // 3 / 9: Gamma-operator layouts (dense planes, nf fields per chunk in both directions; yA2 only)
  unsigned *d = nullptr;    // one allocation holding all tables
  const unsigned *xch = nullptr, *xoff = nullptr, *fsz = nullptr, *cofi = nullptr, *xin = nullptr, *xfs = nullptr;  // x passes
  const unsigned *ych = nullptr, *yD = nullptr, *yB = nullptr, *yC = nullptr, *yA2 = nullptr, *yA1 = nullptr;  // fused y pass
};

struct Profile {
  const char *name;
  double ms;
  long long launches;
  double bytes;  // algorithmic HBM bytes of one launch (DESIGN.md "Kernels")
};

}  // namespace mrl

struct mrl_ctx {
  int dim = 0;
  int off = 0;                     // internal axis of user axis d is d + off
  long long n[3] = {1, 1, 1};      // global real extents in internal order (A0, A1, A2), A2 contiguous
  double gmin[3], gmax[3], dx[3];
  int spectrum = MRL_SPECTRUM_HALF;
  int nranks = 1, rank = 0;
  bool slab = false;            // FFT_SLAB layout and staged entry points (nranks > 1, or MRL_FLAG_SLAB)
  bool pencil = false;          // FFT_PENCIL (MRL_FLAG_PENCIL): y / z split in real space, kx / ky split in reciprocal space (pencil.hip)
  int pen_py = 1, pen_pz = 1;   // process grid: rank r = (r % pen_py, r / pen_py)
  std::vector<long long> pen_y, pen_z;    // real-space counts of the py y blocks / the pz z blocks (partitionHepler, equal weights)
  std::vector<long long> pen_kx, pen_ky;  // reciprocal counts: kx = nx/2+1 over py, ky = ny over pz
  bool gamma_z_ready = false;   // d_work[18] holds the 9 z spectra written by mrl_slab_gamma_tangent_z_fwd
  int gamma_dot_nb = 0;         // workgroups per row of the fused slab Gamma z pass that left dot-product partials (d_work[3])
  int device = 0;
  long long nloc[3] = {1, 1, 1};   // local real extents
  long long rbeg[3] = {0, 0, 0};   // local real begin
  long long nrec[3] = {1, 1, 1};   // local reciprocal extents
  long long kbeg[3] = {0, 0, 0};   // local reciprocal begin
  long long nrec_glob[3] = {1, 1, 1};
  std::vector<long long> part_real;   // split of the real-space slab axis per rank
  std::vector<long long> part_recip;  // split of the reciprocal slab axis per rank
  int split_real_axis = -1, split_recip_axis = -1;  // internal axes (slab mode)

  hipStream_t stream = nullptr;
  bool own_stream = false;
  mrl::AxisPlan ax[3];
  std::vector<double> h_k[3];   // reciprocal axis values (global), internal axis order
  double *d_k[3] = {nullptr, nullptr, nullptr};  // device copies of the LOCAL reciprocal axes
  std::vector<double> h_x[3];   // real-space axis values (global): linspace(min+dx/2, max-dx/2, n), DomainAction.C:246-251
  double *d_x[3] = {nullptr, nullptr, nullptr};  // device copies of the LOCAL real-space axes

  // scratch (complex spectra), grown on demand
  // slots: 0 inverse-transform scratch, 1-3 Cahn-Hilliard, 4-10 mechanics, 11-15 slab stages, 16-18 slab Gamma rows, 19-20 slab Gamma generic
  double *d_work[mrl::kWorkSlots] = {};
  size_t work_bytes[mrl::kWorkSlots] = {};
  double *d_red = nullptr;      // reduction scratch
  double *h_red = nullptr;      // pinned host scratch: [0,32) reduction results, [32,48) / [48,64) rings of host scalars on their way to the device
  unsigned scalar_ring = 0;
  double *d_h_red = nullptr;    // its device-side address (kernels may write results there), nullptr if not mappable

  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  hipEvent_t cg_ev[2] = {nullptr, nullptr};   // look-ahead conjugate-gradient loop (mech.hip): end of iteration k, k & 1
  // small argument tables that do not fit a kernel's argument block (mrl_kspace_coupled beyond 8 variables): a ring of pinned host
  // staging slots, each with its device copy and an event that says when the launch that read it has finished -- no host
  // synchronisation per call, capture-safe ordering
  struct TabSlot {
    unsigned char *h = nullptr, *d = nullptr;
    size_t cap = 0;
    hipEvent_t done = nullptr;
  } tab_ring[4];
  int tab_next = 0;
  bool profiling = false;
  std::vector<mrl::Profile> prof;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  std::vector<int> prof_slots;

  int exp = 0;  // experiment switches (MRL_OPT_EXPERIMENT, bit mask): A/B testing of kernel variants inside one process
  int opt_nsub = 1;    // MRL_OPT_SLAB_NSUB
  int opt_chunk_mb = 0;   // MRL_OPT_CACHE_CHUNK_MB: 0 = off, > 0 = MB of c-hat + mu-hat planes per chunk
  int opt_carry = 0;   // MRL_OPT_SLAB_CARRY
  int opt_verify = 0;  // MRL_OPT_VERIFY_EXCHANGE
  // serial contexts on the fused fast path: elements between two x planes of the solver-private spectral arrays (work arrays, Nhat
  // history, cbar).  ny * nzc rounded up so that the plane pitch is an ODD number of 256-byte pieces (fft_pow2_kernels.h: ZLay);
  // 0 = dense (every other context, and MRL_FLAG_DENSE_SPECTRA)
  long long spec_plane = 0;

  // multi-GPU (slab contexts): the attached communicator (not owned) and the exchange pipelines built on it (slab_driver.hip)
  mrl_comm *comm = nullptr;
  struct mrl::SlabPipes *pipes = nullptr;
  char **d_tabs = nullptr;  // 8 device pointer tables of 64 entries for the staged entry points (caller-owned send buffers)
  const void *tab_base[8] = {};          // what each table was last filled from: the same buffer and offsets need no new fill
  unsigned long long tab_off[8][64] = {};
  bool tab_valid[8] = {};
  std::vector<mrl::SlabTabs> slab_tabs;

  mutable std::string err;
};

namespace mrl {

int set_error(const mrl_ctx *ctx, int code, const char *fmt, ...);
extern thread_local std::string g_create_error;

#define MRL_HIP(ctx, expr)                                                                      \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return mrl::set_error(ctx, MRL_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                            __FILE__, __LINE__);                                                \
  } while (0)

#define MRL_TRY(expr)          \
  do {                         \
    int rc_ = (expr);          \
    if (rc_ != MRL_OK) return rc_; \
  } while (0)

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// block sum of `v` over 256 threads -> valid in thread 0
__device__ __forceinline__ double block_sum256(double v, double *sh) {
  v = wave_sum_f64(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return r;
}

// streaming (non-temporal) accesses for arrays that are not re-read while they could still be in the 256 MB Infinity Cache: they
// should not displace the arrays that the next kernel does re-read
typedef double mrl_ntv2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ld_nt(const double2 *p) {
  const mrl_ntv2 v = __builtin_nontemporal_load(reinterpret_cast<const mrl_ntv2 *>(p));
  return make_double2(v.x, v.y);
}
__device__ __forceinline__ void st_nt(double2 *p, double2 v) {
  mrl_ntv2 w;
  w.x = v.x;
  w.y = v.y;
  __builtin_nontemporal_store(w, reinterpret_cast<mrl_ntv2 *>(p));
}
typedef float mrl_ntv2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 ld_nt(const float2 *p) {
  const mrl_ntv2f v = __builtin_nontemporal_load(reinterpret_cast<const mrl_ntv2f *>(p));
  return make_float2(v.x, v.y);
}
__device__ __forceinline__ void st_nt(float2 *p, float2 v) {
  mrl_ntv2f w;
  w.x = v.x;
  w.y = v.y;
  __builtin_nontemporal_store(w, reinterpret_cast<mrl_ntv2f *>(p));
}

// scratch management
int ensure_work(mrl_ctx *ctx, int slot, size_t bytes);
// device pointer table `slot` (0..7) filled on the context's stream with base + p * stride_bytes, p = 0..nranks-1: the
// destination table of a scatter-capable kernel when the chunks go to one contiguous local buffer
int axis_tw32(mrl_ctx *ctx, int axis);  // fp32 twiddle table / local reciprocal axis of an internal axis, built on first use
int axis_k32(mrl_ctx *ctx, int axis);
int local_tab(mrl_ctx *ctx, int slot, void *base, size_t stride_bytes, cplx *const **out);
int local_tab_offsets(mrl_ctx *ctx, int slot, void *base, const size_t *byte_offsets, cplx *const **out);
void slab_pipes_destroy(mrl_ctx *ctx);
void slab_detach_comm(mrl_ctx *ctx);
long long slab_verify_count(mrl_ctx *ctx, bool reset);  // MRL_OPT_VERIFY_MISMATCHES: synchronises the stream; -1 on a HIP error
int slab_comm_check(mrl_ctx *ctx);  // MRL_ERR_COMM if a device-side wait of the attached communicator timed out

// generic pass launcher (fft_generic.hip)
int launch_pass(mrl_ctx *ctx, const PassDesc &d, const double *in, double *out, const cplx *d_tw, long long nbatch);

// spectral sizes
// entry points that run their own (serial or slab) transform pipelines refuse FFT_PENCIL contexts: those offer the transforms
// (mrl_fft_r2c / mrl_fft_c2r), the reductions and the pointwise entry points
#define MRL_NO_PENCIL(ctx, what)                                                                                                  \
  do {                                                                                                                            \
    if ((ctx)->pencil)                                                                                                            \
      return mrl::set_error(ctx, MRL_ERR_UNSUPPORTED, "%s: not available on FFT_PENCIL contexts (use mrl_fft_r2c / mrl_fft_c2r and " \
                                                      "the pointwise entry points, or FFT_SLAB)", what);                         \
  } while (0)
inline long long spec_count_local(const mrl_ctx *c) { return c->nrec[0] * c->nrec[1] * c->nrec[2]; }
inline long long real_count_local(const mrl_ctx *c) { return c->nloc[0] * c->nloc[1] * c->nloc[2]; }

// profiling scope
struct ProfScope {
  mrl_ctx *ctx;
  int slot;
  hipEvent_t a = nullptr, b = nullptr;
  ProfScope(mrl_ctx *c, const char *name, double bytes = 0.0);
  ~ProfScope();
};

// Cahn-Hilliard parameters as the kernels take them
struct ChP {
  int family;
  double c0, c1, c2;
  double M, kappa;
  mrl_parsed *parsed;  // MRL_FE_PARSED
};

// parsed free energies (expr.hip)
int parsed_check_mu(mrl_ctx *ctx, const mrl_parsed *p);                     // one real input, real output, same context
int parsed_eval1(mrl_parsed *p, const double *c, double *mu, long long n);  // mu = expression(c), pointwise
// k_z_fwd<N, mode, PARSED> (mode 1: c and mu per line, 2: mu of two lines) with the generated chemical potential compiled in (hiprtc)
// lay_lpp / lay_pad: p2::ZLay of the spectral side (rows per x plane, extra elements per plane; 0, 0 = dense)
int parsed_z_fwd_launch(mrl_ctx *ctx, mrl_parsed *p, int N, int mode, const double *in, cplx *out0, cplx *out1, double *mu_out,
                        long long nlines, unsigned lay_lpp = 0, unsigned lay_pad = 0);
int parsed_z_inv_fwd_launch(mrl_ctx *ctx, mrl_parsed *p, int N, const cplx *in, cplx *out0, cplx *out1, double *mu_out, double scale,
                            long long nlines, bool mu_only = false, unsigned lay_lpp = 0, unsigned lay_pad = 0);

// power-of-two fast path (ch_fused.hip)
bool fast_path_ok(const mrl_ctx *ctx);
int fft_forward_fast(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch);
int fft_inverse_fast(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch);

// reductions (reduce.hip): op 0 sum(a), 1 sum(a*b), 2 sum(a*a); results stay on the device
int reduce_async(mrl_ctx *ctx, int op, const double *a, const double *b, long long n, double *d_scalar);
int reduce_finalize(mrl_ctx *ctx, int nb, int nslots, double *d_scalar);
int read_scalars(mrl_ctx *ctx, const double *d_scalar, int count, double *h_out);
int reduce_finalize_to_host(mrl_ctx *ctx, int nb, int nslots, double *d_scalar, double *h_out);
int component_sums_async(mrl_ctx *ctx, const double *a, long long npts, int ncomp, double *d_scalar);

// serial transforms (fft_plan.hip)
int fft_forward_serial(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch, int layout);
int fft_inverse_serial(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch, int layout);
int pencil_fft_forward(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch);   // slab_driver.hip (pencil contexts)
int pencil_fft_inverse(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch);

}  // namespace mrl
