// Fast path of the Gamma-operator application on a slab-decomposed grid (BASELINE configs[4]: de Geus mechanics over the
// GPUs of a node): G(A) = ifftSlab( Ghat4 : fftSlab(A) ) (FFTMechanics.C:74-84,105-106 over DomainAction.C:869-1019) on
// FIELD-MAJOR data [9][nx][nyl][nz], 3-D power-of-two extents, equal partitions.
//
// out_ij = q_j (sum_k A^_ik q_k)/|q|^2 couples only the three components of one tensor ROW i, so the three rows are
// independent from the first z pass to the last one and form the natural pipeline: while row i is on the wire, row i+1 is
// being transformed (full z lines per row, so -- unlike the kz sub-blocks of the Cahn-Hilliard pipeline -- the z passes
// overlap with communication too).  Per row:
//   fwd   k_z_fwd<PAIR> of the row's 3 fields, forward x pass written straight into the exchange layout
//         [p][3 fields][x_p][y_me][nzc]                      (one message per peer and row, no pack kernel)
//   -- all-to-all --
//   mid   k_gamma_yfused: gathers lines along y from [p][3][x_me][y_p][nzc], forward y of the 3 components,
//         s = (A_i0 kx + A_i1 ky + A_i2 kz)/|q|^2, out_ij = s q_j, inverse y, written back IN PLACE (the inverse
//         exchange layout is the same chunked layout), Ghat4 (1296 B per k-point in the reference) is never formed
//   -- all-to-all --
//   inv   inverse x pass from [p][3][x_p][y_me][nzc] into the work arrays, k_z_inv<PAIR> * 1/N into the row's 3 output fields
#include "fft_pow2_launch.h"
#include <atomic>
#include "slab_stages.h"
#include "slab_tables.h"

// register budget of k_gamma_yfused: see k_gamma_xfused (mech_fused.hip); here the full prefetch of the third component measured
// best once the output transforms are no longer unrolled into each other (rank-local 256^3 / 8: 60 -> 45 us per row)
#ifndef MRL_GAMMAY_PRE
#define MRL_GAMMAY_PRE 16
#endif
#ifndef MRL_GAMMA_JUNROLL
#define MRL_GAMMA_JUNROLL 1
#endif

namespace mrl {
int gamma_z_fwd_tangent_launch(mrl_ctx *ctx, const double *F, const double *K, const double *mu, double *p, const double *r,
                               const double *S, int i_num, int i_den, cplx *spec, long long npts, long long rows, int nz, bool nt,
                               double *x, int i_arz, int i_apAp);
}

namespace mrl {

int slab_mech_fast_ok(const mrl_ctx *ctx);
int reduce_finalize_from(mrl_ctx *ctx, const double *partial, int nb, double *d_scalar);

namespace p2 {

struct GammaYArgs {
  const cplx *buf;    // received [p][3][nxl][nyl_p][nzc]
  cplx *const *otab;  // projected output: chunk p (same layout) starts at otab[p] -- in place (otab[p] = buf + p*3*chunk), the local
                      // send buffer, or rank p's receive buffer of the inverse exchange (direct peer stores)
  SignalArgs sig;
  int nxl, nzc;
  int nyl_shift;      // log2(ny / P)
  unsigned chunk;     // nxl * nyl * nzc: elements of one field of one chunk
  int nf;             // fields per chunk: 3 (one tensor row per exchange) or 9 (all rows in one exchange, f = 3 row + component)
  unsigned rowblk;    // workgroups per tensor row (gridDim = rows * rowblk)
  const double *kx, *ky, *kz;  // local reciprocal axes (kx already offset to this rank's x range)
  double scale;
};

// AL: TPL | ny/P (the usual case), so the chunk index and the row within the chunk of point j = q + m TPL split into a wave-uniform
// part that depends on m only and q: every load / store is "uniform 64-bit base + one per-thread 32-bit offset" (the general form keeps
// 16 per-thread offsets and a per-element table load alive across the six transforms and spilled 47-96 VGPRs)
template <int N, bool AL>
__global__ void __launch_bounds__(Plan<N>::NT, 2) k_gamma_yfused(GammaYArgs a, const cplx *__restrict__ tw) {
#pragma clang fp contract(off)
  constexpr int P = Plan<N>::P, TPL = Plan<N>::TPL, T = Plan<N>::T, NT = Plan<N>::NT, CNT = (N + NT - 1) / NT;
  constexpr int GPRE = MRL_GAMMAY_PRE < P ? MRL_GAMMAY_PRE : P;
  using Map = MapStrided<N>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx *W = reinterpret_cast<cplx *>(smem);
  cplx *X = W + N;
  double *KY = reinterpret_cast<double *>(X + Map::size);
  const int l = threadIdx.x % T, q = threadIdx.x / T;
  unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned trow = logical / a.rowblk;  // tensor row of this workgroup (0 when every row has its own exchange)
  logical -= trow * a.rowblk;
  // tiles run over the flattened (x, kz) index: with one tile row per x, the last tile of a row held nzc mod T lines (1 of 32 at
  // 128^3, 1 of 16 at 256^3 -- 10 to 30 % of the lanes idle)
  const unsigned li = logical * T + l;
  const bool valid = li < (unsigned)(a.nxl * a.nzc);
  const int ix = valid ? (int)(li / (unsigned)a.nzc) : 0;
  const int kl = valid ? (int)(li - (unsigned)ix * (unsigned)a.nzc) : 0;
  cplx twv[CNT];
  double kyv[CNT];
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    const int idx = threadIdx.x + j * NT;
    twv[j] = idx < N ? tw[idx] : make_double2(0.0, 0.0);
    kyv[j] = idx < N ? a.ky[idx] : 0.0;
  }
  const double kx = a.kx[ix], kz = a.kz[kl];
  // element (f, ix, j, kl): (j >> sh) * 3*chunk + f*chunk + ((ix << sh) + (j & msk)) * nzc + kl     [byte offsets]
  const int sh = a.nyl_shift, msk = (1 << sh) - 1;
  const unsigned chB = a.chunk * 16u, rowB = (unsigned)a.nzc * 16u, klB = (unsigned)kl * 16u + 3u * trow * chB;
  const unsigned peerB = (unsigned)a.nf * chB;
  const unsigned tb = (unsigned)((ix << sh) + (AL ? q : 0)) * rowB + klB;  // per-thread part
  auto off = [=](int m) {
    if (AL) return (unsigned)((m * TPL) >> sh) * peerB + (unsigned)((m * TPL) & msk) * rowB;  // wave-uniform
    const int j = q + m * TPL;
    return (unsigned)(j >> sh) * peerB + (unsigned)(j & msk) * rowB;
  };
  cplx *const *otab = a.otab;
  auto ld = [=](unsigned f, int m) {
    return *reinterpret_cast<const cplx *>(reinterpret_cast<const char *>(a.buf) + (size_t)(f * chB + off(m)) + tb);
  };
  cplx v0[P], v1[P];
#pragma unroll
  for (int m = 0; m < P; ++m) v0[m] = ld(0u, m);
#pragma unroll
  for (int m = 0; m < P; ++m) v1[m] = ld(1u, m);
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    const int idx = threadIdx.x + j * NT;
    if (idx < N) {
      W[idx] = twv[j];
      KY[idx] = kyv[j];
    }
  }
  // s = sum_k A_ik q_k with q = (kx, ky along the line, kz); same association as the serial kernel (k_gamma_xfused)
  cplx s[P];
  fft_line<N, Map>(v0, q, l, X, W);
#pragma unroll
  for (int m = 0; m < P; ++m) s[m] = make_double2(v0[m].x * kx, v0[m].y * kx);
#pragma unroll
  for (int m = 0; m < GPRE; ++m) v0[m] = ld(2u, m);  // third component: GPRE values in flight during the second transform
  fft_line<N, Map>(v1, q, l, X, W);
#pragma unroll
  for (int m = GPRE; m < P; ++m) v0[m] = ld(2u, m);
#pragma unroll
  for (int m = 0; m < P; ++m) {
    const double ky = KY[q + m * TPL];
    s[m].x += v1[m].x * ky;
    s[m].y += v1[m].y * ky;
  }
  fft_line<N, Map>(v0, q, l, X, W);
#pragma unroll
  for (int m = 0; m < P; ++m) {
    const double ky = KY[q + m * TPL];
    const double Q = kx * kx + (ky * ky + kz * kz);
    const double inv = (Q == 0.0) ? 0.0 : a.scale / Q;
    s[m].x = (s[m].x + v0[m].x * kz) * inv;
    s[m].y = (s[m].y + v0[m].y * kz) * inv;
  }
  // out_ij = s q_j, inverse y (unnormalised; 1/N applied by the z pass); swap trick for the inverse
#pragma unroll MRL_GAMMA_JUNROLL
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int m = 0; m < P; ++m) {
      const double qj = (j == 0) ? kx : (j == 1 ? KY[q + m * TPL] : kz);
      v0[m] = make_double2(s[m].y * qj, s[m].x * qj);
    }
    fft_line<N, Map>(v0, q, l, X, W);
    if (valid) {
#pragma unroll
      for (int m = 0; m < P; ++m) {
        const int jj = AL ? m * TPL : q + m * TPL;
        *reinterpret_cast<cplx *>(reinterpret_cast<char *>(otab[jj >> sh]) + (size_t)((unsigned)j * chB + (unsigned)(jj & msk) * rowB) + tb) = cswap(v0[m]);
      }
    }
  }
  signal_tail(a.sig);
}

template <int N>
static int launch_gamma_yfused(mrl_ctx *ctx, GammaYArgs a) {
  static std::atomic<bool> attr{false};  // (two host threads may both set the attribute: harmless, and no torn flag)
  constexpr size_t lds = lds_strided<N>() + sizeof(double) * N;
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY((set_lds_attr(ctx, k_gamma_yfused<N, true>, lds)));
    attr.store(true, std::memory_order_release);
  }
  constexpr int T = Plan<N>::T;
  a.rowblk = (unsigned)(((long long)a.nxl * a.nzc + T - 1) / T);
  const long long nb = (long long)(a.nf / 3) * a.rowblk;
  a.sig.expected = (unsigned)nb;
  if ((1 << a.nyl_shift) % Plan<N>::TPL)  // (slab_fast_shift sends such partitions to k_gamma_yfused_t)
    return set_error(ctx, MRL_ERR_UNSUPPORTED, "slab Gamma y pass: %d rows per chunk are not a multiple of the %d threads of a line", 1 << a.nyl_shift, Plan<N>::TPL);
  hipLaunchKernelGGL((k_gamma_yfused<N, true>), dim3((unsigned)nb), dim3(Plan<N>::NT), lds, ctx->stream, a, ctx->ax[1].d_tw);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

// Table-addressed form of k_gamma_yfused (slab_mech_table_ok): the same transforms and the same projection arithmetic in the same order;
// the chunk of y row j and its offsets come from the tables (built for nf fields per chunk, dense planes: slab_tabs_get(ctx, nzc, nf)).
//   received element (f, ix, j, kl): yA[j] + f yC[j] + ix yB[j] + kl ;   in the chunk for rank ych[j]: f yC[j] + ix yB[j] + yD[j] + kl
template <int N>
__global__ void __launch_bounds__(Plan<N>::NT, 2) k_gamma_yfused_t(GammaYArgs a, YTabs t, const cplx *__restrict__ tw) {
#pragma clang fp contract(off)
  constexpr int P = Plan<N>::P, TPL = Plan<N>::TPL, T = Plan<N>::T, NT = Plan<N>::NT, CNT = (N + NT - 1) / NT;
  using Map = MapStrided<N>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx *W = reinterpret_cast<cplx *>(smem);
  cplx *X = W + N;
  double *KY = reinterpret_cast<double *>(X + Map::size);
  const int l = threadIdx.x % T, q = threadIdx.x / T;
  unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned trow = logical / a.rowblk;
  logical -= trow * a.rowblk;
  const unsigned li = logical * T + l;
  const bool valid = li < (unsigned)(a.nxl * a.nzc);
  const unsigned ix = valid ? li / (unsigned)a.nzc : 0u;
  const unsigned kl = valid ? li - ix * (unsigned)a.nzc : 0u;
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    const int idx = threadIdx.x + j * NT;
    if (idx < N) {
      W[idx] = tw[idx];
      KY[idx] = a.ky[idx];
    }
  }
  const double kx = a.kx[ix], kz = a.kz[kl];
  const unsigned f0 = 3u * trow;
  auto ld = [=](unsigned f, int m) {
    const unsigned j = q + m * TPL;
    return a.buf[t.yA[j] + (f0 + f) * t.yC[j] + ix * t.yB[j] + kl];
  };
  cplx v0[P], v1[P];
#pragma unroll
  for (int m = 0; m < P; ++m) v0[m] = ld(0u, m);
#pragma unroll
  for (int m = 0; m < P; ++m) v1[m] = ld(1u, m);
  // (fft_line starts with a barrier-protected exchange through X only; W / KY are read after its first __syncthreads)
  __syncthreads();
  cplx s[P];
  fft_line<N, Map>(v0, q, l, X, W);
#pragma unroll
  for (int m = 0; m < P; ++m) s[m] = make_double2(v0[m].x * kx, v0[m].y * kx);
#pragma unroll
  for (int m = 0; m < P; ++m) v0[m] = ld(2u, m);
  fft_line<N, Map>(v1, q, l, X, W);
#pragma unroll
  for (int m = 0; m < P; ++m) {
    const double ky = KY[q + m * TPL];
    s[m].x += v1[m].x * ky;
    s[m].y += v1[m].y * ky;
  }
  fft_line<N, Map>(v0, q, l, X, W);
#pragma unroll
  for (int m = 0; m < P; ++m) {
    const double ky = KY[q + m * TPL];
    const double Q = kx * kx + (ky * ky + kz * kz);
    const double inv = (Q == 0.0) ? 0.0 : a.scale / Q;
    s[m].x = (s[m].x + v0[m].x * kz) * inv;
    s[m].y = (s[m].y + v0[m].y * kz) * inv;
  }
#pragma unroll 1
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int m = 0; m < P; ++m) {
      const double qj = (j == 0) ? kx : (j == 1 ? KY[q + m * TPL] : kz);
      v0[m] = make_double2(s[m].y * qj, s[m].x * qj);
    }
    fft_line<N, Map>(v0, q, l, X, W);
    if (valid) {
#pragma unroll
      for (int m = 0; m < P; ++m) {
        const unsigned jj = q + m * TPL;
        a.otab[t.ych[jj]][(f0 + (unsigned)j) * t.yC[jj] + ix * t.yB[jj] + t.yD[jj] + kl] = cswap(v0[m]);
      }
    }
  }
  signal_tail(a.sig);
}

template <int N>
static int launch_gamma_yfused_t(mrl_ctx *ctx, GammaYArgs a, const YTabs &t) {
  static std::atomic<bool> attr{false};  // (two host threads may both set the attribute: harmless, and no torn flag)
  constexpr size_t lds = lds_strided<N>() + sizeof(double) * N;
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY((set_lds_attr(ctx, k_gamma_yfused_t<N>, lds)));
    attr.store(true, std::memory_order_release);
  }
  constexpr int T = Plan<N>::T;
  a.rowblk = (unsigned)(((long long)a.nxl * a.nzc + T - 1) / T);
  const long long nb = (long long)(a.nf / 3) * a.rowblk;
  a.sig.expected = (unsigned)nb;
  hipLaunchKernelGGL((k_gamma_yfused_t<N>), dim3((unsigned)nb), dim3(Plan<N>::NT), lds, ctx->stream, a, t, ctx->ax[1].d_tw);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

}  // namespace p2

static int ilog2(long long v) {
  int s = 0;
  while ((1LL << s) < v) ++s;
  return s;
}

// table-addressed x passes of nf fields (k_pass_sub_mft): forward from the dense work array w into the exchange layout through otab,
// inverse from the received chunks into w
static int x_pass_tab(mrl_ctx *ctx, bool inverse, int nf, const cplx *src, cplx *dst, cplx *const *otab, const SignalArgs &sig) {
  const long long nx = ctx->n[0], nyl = ctx->nloc[1], nzc = ctx->nrec[2];
  const SlabTabs *tb;
  MRL_TRY(slab_tabs_get(ctx, nzc, nf, &tb));
  p2::SubPassArgs a{};
  a.rows = (int)nyl;
  a.cols = (int)nzc;
  a.pitch_in = a.pitch_out = (unsigned)nzc;
  a.sn_in = a.sn_out = (unsigned)(nyl * nzc);
  a.fdense = (unsigned)(nx * nyl * nzc);
  a.in[0] = src;
  a.out[0] = dst;
  a.otab = otab;
  a.sig = sig;
  const p2::SubPassTabs t{tb->xch, tb->xoff, tb->fsz, tb->xin, tb->xfs};
  if (inverse) {
    MRL_SWITCH_N(nx, MRL_TRY((p2::launch_pass_sub_mft<NN, true>(ctx, a, t, ctx->ax[0].d_tw, nf))));
  } else {
    MRL_SWITCH_N(nx, MRL_TRY((p2::launch_pass_sub_mft<NN, false>(ctx, a, t, ctx->ax[0].d_tw, nf))));
  }
  return MRL_OK;
}

static int y_fused_tab(mrl_ctx *ctx, int nf, const double *recv, cplx *const *otab, const SignalArgs &sig, double scale) {
  const long long nxl = ctx->nrec[0], ny = ctx->n[1], nzc = ctx->nrec[2];
  const SlabTabs *tb;
  MRL_TRY(slab_tabs_get(ctx, nzc, nf, &tb));
  p2::GammaYArgs a{};
  a.buf = reinterpret_cast<const cplx *>(recv);
  a.otab = otab;
  a.sig = sig;
  a.nxl = (int)nxl;
  a.nzc = (int)nzc;
  a.nf = nf;
  a.kx = ctx->d_k[0];
  a.ky = ctx->d_k[1];
  a.kz = ctx->d_k[2];
  a.scale = scale;
  const p2::YTabs t{tb->ych, tb->yD, tb->yB, tb->yC, tb->yA2};
  MRL_SWITCH_N(ny, MRL_TRY((p2::launch_gamma_yfused_t<NN>(ctx, a, t))));
  return MRL_OK;
}

// work arrays: slot 16 = the three z/x-transformed fields of a row [3][nx][nyl][nzc] (one row at a time per direction:
// slot 16 forward, slot 17 inverse, so that the forward stage of row i+1 can run while row i's inverse stage is pending)
static int row_work(mrl_ctx *ctx, int slot, cplx **w) {
  const size_t bytes = sizeof(cplx) * (size_t)(3 * ctx->n[0] * ctx->nloc[1] * ctx->nrec[2]);
  MRL_TRY(ensure_work(ctx, slot, bytes));
  *w = reinterpret_cast<cplx *>(ctx->d_work[slot]);
  return MRL_OK;
}

// the staged entry points (caller-owned exchange: chunks of one size at a uniform stride in the caller's buffers): the shift-addressed
// pipeline, or the table-addressed one on EQUAL partitions (e.g. fewer rows per chunk than the threads of a line)
static bool staged_ok(const mrl_ctx *ctx) {
  if (slab_mech_fast_ok(ctx)) return ctx->nloc[1] % 2 == 0;
  if (!slab_mech_table_ok(ctx)) return false;
  for (int p = 0; p < ctx->nranks; ++p)
    if (ctx->part_real[p] != ctx->part_real[0] || ctx->part_recip[p] != ctx->part_recip[0]) return false;
  return true;
}

static int check_fast(mrl_ctx *ctx, const char *what, int row) {
  if (!staged_ok(ctx))
    return set_error(ctx, MRL_ERR_UNSUPPORTED, "%s: needs a 3-D slab context with planned extents, equal partitions and an even number of local z lines", what);
  if (row < 0 || row > 2) return set_error(ctx, MRL_ERR_INVALID, "%s: row %d out of range", what, row);
  return MRL_OK;
}

int slab_gamma_row_fwd(mrl_ctx *ctx, int row, const double *d_A_fm, cplx *const *otab, const SignalArgs &sig) {
  if (!d_A_fm && !ctx->gamma_z_ready)
    return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_gamma_row_fwd: no input field and no spectra from mrl_slab_gamma_tangent_z_fwd");
  const long long nx = ctx->n[0], nyl = ctx->nloc[1], nz = ctx->n[2], nzc = ctx->nrec[2], nxl = ctx->nrec[0];
  const long long nreal = nx * nyl * nz, nspec = nx * nyl * nzc;
  cplx *w;
  if (d_A_fm) {
    MRL_TRY(row_work(ctx, 16, &w));
    ProfScope ps(ctx, "slab_gamma_z_fwd", 3.0 * (8.0 * nreal + 16.0 * nspec));
    p2::ChDev none{};
    MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_fwd<NN, 0, 0>(ctx, d_A_fm + 3 * row * nreal, w, nullptr, nullptr, none, 3 * nx * nyl / 2))));
  } else {
    w = reinterpret_cast<cplx *>(ctx->d_work[18]) + 3 * row * nspec;  // z spectra left by mrl_slab_gamma_tangent_z_fwd
    if (row == 2) ctx->gamma_z_ready = false;                         // consumed: a later call needs a new fused z pass
  }
  if (!slab_fast_shift(ctx)) {
    ProfScope ps(ctx, "slab_gamma_x_fwd", 3.0 * 32.0 * nspec);
    return x_pass_tab(ctx, false, 3, w, nullptr, otab, sig);
  }
  const unsigned chunk = (unsigned)(nxl * nyl * nzc);
  p2::SubPassArgs a{};
  a.rows = (int)nyl;
  a.cols = (int)nzc;
  a.pitch_in = a.pitch_out = (unsigned)nzc;
  a.sn_in = a.sn_out = (unsigned)(nyl * nzc);
  a.sh_in = 31;
  a.sh_out = ilog2(nxl);
  a.otab = otab;
  a.fs_out = chunk;
  a.sig = sig;  // the three fields of the row go out in two launches that count towards ONE arrival flag
  unsigned nb = 0;
  MRL_SWITCH_N(nx, nb = pass_sub_blocks<NN>(nyl, nzc));
  a.sig.expected = 2u * nb;
  ProfScope ps(ctx, "slab_gamma_x_fwd", 3.0 * 32.0 * nspec);
  a.in[0] = w;
  a.in[1] = w + nspec;
  a.fo_out = 0;
  MRL_SWITCH_N(nx, MRL_TRY((p2::launch_pass_sub<NN, false, 2>(ctx, a, ctx->ax[0].d_tw))));
  a.in[0] = w + 2 * nspec;
  a.fo_out = 2u * chunk;
  MRL_SWITCH_N(nx, MRL_TRY((p2::launch_pass_sub<NN, false, 1>(ctx, a, ctx->ax[0].d_tw))));
  return MRL_OK;
}

int slab_gamma_row_mid(mrl_ctx *ctx, const double *recv, cplx *const *otab, const SignalArgs &sig, double scale) {
  const long long nxl = ctx->nrec[0], ny = ctx->n[1], nzc = ctx->nrec[2], nyl = ny / ctx->nranks;
  if (!slab_fast_shift(ctx)) {
    ProfScope ps(ctx, "slab_gamma_y_fused", 3.0 * 32.0 * nxl * ny * nzc);
    return y_fused_tab(ctx, 3, recv, otab, sig, scale);
  }
  p2::GammaYArgs a{};
  a.buf = reinterpret_cast<const cplx *>(recv);
  a.otab = otab;
  a.sig = sig;
  a.nxl = (int)nxl;
  a.nzc = (int)nzc;
  a.nf = 3;
  a.nyl_shift = ilog2(nyl);
  a.chunk = (unsigned)(nxl * nyl * nzc);
  a.kx = ctx->d_k[0];
  a.ky = ctx->d_k[1];
  a.kz = ctx->d_k[2];
  a.scale = scale;
  ProfScope ps(ctx, "slab_gamma_y_fused", 3.0 * 32.0 * nxl * ny * nzc);
  MRL_SWITCH_N(ny, MRL_TRY((p2::launch_gamma_yfused<NN>(ctx, a))));
  return MRL_OK;
}

int slab_gamma_row_inv(mrl_ctx *ctx, int row, const double *d_recv, double *d_out_fm, const double *d_dotv_fm) {
  const long long nx = ctx->n[0], nyl = ctx->nloc[1], nz = ctx->n[2], nzc = ctx->nrec[2], nxl = ctx->nrec[0];
  const long long nreal = nx * nyl * nz, nspec = nx * nyl * nzc;
  cplx *w;
  MRL_TRY(row_work(ctx, 17, &w));
  const unsigned chunk = (unsigned)(nxl * nyl * nzc);
  if (!slab_fast_shift(ctx)) {
    ProfScope ps(ctx, "slab_gamma_x_inv", 3.0 * 32.0 * nspec);
    MRL_TRY(x_pass_tab(ctx, true, 3, reinterpret_cast<const cplx *>(d_recv), w, nullptr, SignalArgs{}));
  } else {
    p2::SubPassArgs a{};
    a.rows = (int)nyl;
    a.cols = (int)nzc;
    a.pitch_in = a.pitch_out = (unsigned)nzc;
    a.sn_in = a.sn_out = (unsigned)(nyl * nzc);
    a.sh_in = ilog2(nxl);
    a.cs_in = 3u * chunk;
    a.sh_out = 31;
    ProfScope ps(ctx, "slab_gamma_x_inv", 3.0 * 32.0 * nspec);
    for (int f = 0; f < 3; ++f) {
      a.in[0] = reinterpret_cast<const cplx *>(d_recv) + (long long)f * chunk;
      a.out[0] = w + (long long)f * nspec;
      MRL_SWITCH_N(nx, MRL_TRY((p2::launch_pass_sub<NN, true, 1>(ctx, a, ctx->ax[0].d_tw))));
    }
  }
  ProfScope ps(ctx, "slab_gamma_z_inv", 3.0 * (16.0 * nspec + 8.0 * nreal) + (d_dotv_fm ? 3.0 * 8.0 * nreal : 0.0));
  const double scale = 1.0 / ((double)nx * (double)ctx->n[1] * (double)nz);
  if (!d_dotv_fm) {
    MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_inv<NN>(ctx, w, d_out_fm + 3 * row * nreal, scale, 3 * nx * nyl / 2))));
    return MRL_OK;
  }
  // sum(out * dotv) of this row while `out` is still in registers: one partial per workgroup, rows back to back
  const long long max_blocks = 3 * nx * nyl / 2;  // >= the number of workgroups of one row for every plan
  MRL_TRY(ensure_work(ctx, 3, sizeof(double) * (size_t)(3 * max_blocks)));
  int nb = 0;
  if (row > 0 && ctx->gamma_dot_nb == 0)
    return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_gamma_row_inv: dot-product rows must start with row 0");
  const long long at = row == 0 ? 0 : (long long)row * ctx->gamma_dot_nb;
  MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_inv_dot<NN>(ctx, w, d_out_fm + 3 * row * nreal, scale, 3 * nx * nyl / 2,
                                                       d_dotv_fm + 3 * row * nreal, ctx->d_work[3] + at, &nb))));
  ctx->gamma_dot_nb = nb;
  return MRL_OK;
}

// ---- all three tensor rows per launch (one exchange of nine fields per direction) ---------------------------------------------
// The row pipeline above overlaps the exchange of row r with the transforms of row r + 1, which pays when the exchange is a copy
// (copy engines, RCCL).  With peer stores the producing kernels ARE the exchange, and at the rank-local sizes of BASELINE
// configs[4] (256^3 over 8 GPUs: 17 MB per field) a one-row launch is half a wave of workgroups (258 on 512 slots): the
// one-rank 128^3 job spent 541 us per CG iteration in 33 such launches where the serial solver needs 274 us in 4.  Here the
// nine fields go through ONE launch per stage; chunk layout [p][9 fields][x][y][nzc], f = 3 row + component.
bool slab_gamma_batched_ok(const mrl_ctx *ctx) {
  long long nxl = ctx->nrec[0];
  for (long long v : ctx->part_recip) nxl = v > nxl ? v : nxl;  // (the same verdict on every rank)
  const long long ny = ctx->n[1], nzc = ctx->nrec[2];
  return 9.0 * 16.0 * (double)(nxl * ny * nzc) < 4294967296.0;  // 32-bit byte offsets within the nine-field exchange buffers
}

static int rows_work(mrl_ctx *ctx, int slot, cplx **w) {
  const size_t bytes = sizeof(cplx) * (size_t)(9 * ctx->n[0] * ctx->nloc[1] * ctx->nrec[2]);
  MRL_TRY(ensure_work(ctx, slot, bytes));
  *w = reinterpret_cast<cplx *>(ctx->d_work[slot]);
  return MRL_OK;
}

int slab_gamma_rows_fwd(mrl_ctx *ctx, const double *d_A_fm, cplx *const *otab, const SignalArgs &sig) {
  if (!d_A_fm && !ctx->gamma_z_ready)
    return set_error(ctx, MRL_ERR_INVALID, "slab Gamma operator: no input field and no spectra from the fused tangent / z pass");
  const long long nx = ctx->n[0], nyl = ctx->nloc[1], nz = ctx->n[2], nzc = ctx->nrec[2], nxl = ctx->nrec[0];
  const long long nreal = nx * nyl * nz, nspec = nx * nyl * nzc;
  cplx *w;
  if (d_A_fm) {
    MRL_TRY(rows_work(ctx, 16, &w));
    ProfScope ps(ctx, "slab_gamma_z_fwd", 9.0 * (8.0 * nreal + 16.0 * nspec));
    p2::ChDev none{};
    MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_fwd<NN, 0, 0>(ctx, d_A_fm, w, nullptr, nullptr, none, 9 * nx * nyl / 2))));
  } else {
    w = reinterpret_cast<cplx *>(ctx->d_work[18]);  // z spectra of the nine fields left by slab_gamma_tangent_z
    ctx->gamma_z_ready = false;
  }
  if (!slab_fast_shift(ctx)) {
    ProfScope ps(ctx, "slab_gamma_x_fwd", 9.0 * 32.0 * nspec);
    return x_pass_tab(ctx, false, 9, w, nullptr, otab, sig);
  }
  p2::SubPassArgs a{};
  a.rows = (int)nyl;
  a.cols = (int)nzc;
  a.pitch_in = a.pitch_out = (unsigned)nzc;
  a.sn_in = a.sn_out = (unsigned)(nyl * nzc);
  a.sh_in = 31;
  a.sh_out = ilog2(nxl);
  a.otab = otab;
  a.fs_out = (unsigned)(nxl * nyl * nzc);
  a.fdense = (unsigned)nspec;
  a.sig = sig;
  a.in[0] = w;
  ProfScope ps(ctx, "slab_gamma_x_fwd", 9.0 * 32.0 * nspec);
  MRL_SWITCH_N(nx, MRL_TRY((p2::launch_pass_sub_mf<NN, false>(ctx, a, ctx->ax[0].d_tw, 9))));
  return MRL_OK;
}

int slab_gamma_rows_mid(mrl_ctx *ctx, const double *recv, cplx *const *otab, const SignalArgs &sig, double scale) {
  const long long nxl = ctx->nrec[0], ny = ctx->n[1], nzc = ctx->nrec[2], nyl = ny / ctx->nranks;
  if (!slab_fast_shift(ctx)) {
    ProfScope ps(ctx, "slab_gamma_y_fused", 9.0 * 32.0 * nxl * ny * nzc);
    return y_fused_tab(ctx, 9, recv, otab, sig, scale);
  }
  p2::GammaYArgs a{};
  a.buf = reinterpret_cast<const cplx *>(recv);
  a.otab = otab;
  a.sig = sig;
  a.nxl = (int)nxl;
  a.nzc = (int)nzc;
  a.nf = 9;
  a.nyl_shift = ilog2(nyl);
  a.chunk = (unsigned)(nxl * nyl * nzc);
  a.kx = ctx->d_k[0];
  a.ky = ctx->d_k[1];
  a.kz = ctx->d_k[2];
  a.scale = scale;
  ProfScope ps(ctx, "slab_gamma_y_fused", 9.0 * 32.0 * nxl * ny * nzc);
  MRL_SWITCH_N(ny, MRL_TRY((p2::launch_gamma_yfused<NN>(ctx, a))));
  return MRL_OK;
}

int slab_gamma_rows_inv(mrl_ctx *ctx, const double *d_recv, double *d_out_fm, const double *d_dotv_fm) {
  const long long nx = ctx->n[0], nyl = ctx->nloc[1], nz = ctx->n[2], nzc = ctx->nrec[2], nxl = ctx->nrec[0];
  const long long nreal = nx * nyl * nz, nspec = nx * nyl * nzc;
  cplx *w;
  MRL_TRY(rows_work(ctx, 17, &w));
  if (!slab_fast_shift(ctx)) {
    ProfScope ps(ctx, "slab_gamma_x_inv", 9.0 * 32.0 * nspec);
    MRL_TRY(x_pass_tab(ctx, true, 9, reinterpret_cast<const cplx *>(d_recv), w, nullptr, SignalArgs{}));
  } else {
    p2::SubPassArgs a{};
    a.rows = (int)nyl;
    a.cols = (int)nzc;
    a.pitch_in = a.pitch_out = (unsigned)nzc;
    a.sn_in = a.sn_out = (unsigned)(nyl * nzc);
    a.sh_in = ilog2(nxl);
    a.fs_out = (unsigned)(nxl * nyl * nzc);
    a.cs_in = 9u * a.fs_out;
    a.sh_out = 31;
    a.fdense = (unsigned)nspec;
    a.in[0] = reinterpret_cast<const cplx *>(d_recv);
    a.out[0] = w;
    ProfScope ps(ctx, "slab_gamma_x_inv", 9.0 * 32.0 * nspec);
    MRL_SWITCH_N(nx, MRL_TRY((p2::launch_pass_sub_mf<NN, true>(ctx, a, ctx->ax[0].d_tw, 9))));
  }
  ProfScope ps(ctx, "slab_gamma_z_inv", 9.0 * (16.0 * nspec + 8.0 * nreal) + (d_dotv_fm ? 9.0 * 8.0 * nreal : 0.0));
  const double scale = 1.0 / ((double)nx * (double)ctx->n[1] * (double)nz);
  if (!d_dotv_fm) {
    MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_inv<NN>(ctx, w, d_out_fm, scale, 9 * nx * nyl / 2))));
    return MRL_OK;
  }
  const long long max_blocks = 9 * nx * nyl / 2;
  MRL_TRY(ensure_work(ctx, 3, sizeof(double) * (size_t)max_blocks));
  int nb = 0;
  MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_inv_dot<NN>(ctx, w, d_out_fm, scale, 9 * nx * nyl / 2, d_dotv_fm, ctx->d_work[3], &nb))));
  ctx->gamma_dot_nb = nb;  // all rows: the caller finalises nb partials
  return MRL_OK;
}

// the fused CG direction + tangent + forward z pass: z lines of 32 ... 256 points in whole 512-point tiles, on EVERY rank (the ranks must
// agree on the pipeline: with the fused form the forward exchange starts from the spectra the z pass left in the context)
int slab_gamma_tangent_fusable(const mrl_ctx *ctx) {
  if (!ctx->slab || ctx->dim != 3 || !slab_mech_any_ok(ctx)) return 0;
  const long long nz = ctx->n[2];
  if (!(nz == 32 || nz == 64 || nz == 128 || nz == 256)) return 0;
  for (int p = 0; p < ctx->nranks; ++p)
    if ((ctx->n[0] * ctx->part_real[p]) % (512 / nz)) return 0;
  return 1;
}

}  // namespace mrl

using namespace mrl;

extern "C" {

int mrl_slab_fast_path(const mrl_ctx *ctx) { return ctx && ctx->slab && ctx->dim == 3 && staged_ok(ctx) ? 1 : 0; }

int mrl_slab_gamma_counts(const mrl_ctx *ctx, int forward, int64_t *h_send_counts, int64_t *h_recv_counts) {
  if (!ctx) return MRL_ERR_INVALID;
  if (!ctx->slab || ctx->dim != 3) return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_gamma_counts: needs a 3-D slab context");
  const long long nzc = ctx->nrec[2];
  for (int p = 0; p < ctx->nranks; ++p) {
    const long long x_p_y_me = 3 * ctx->part_recip[p] * ctx->nloc[1] * nzc, x_me_y_p = 3 * ctx->nrec[0] * ctx->part_real[p] * nzc;
    if (h_send_counts) h_send_counts[p] = forward ? x_p_y_me : x_me_y_p;
    if (h_recv_counts) h_recv_counts[p] = forward ? x_me_y_p : x_p_y_me;
  }
  return MRL_OK;
}

int mrl_slab_gamma_tangent_fusable(const mrl_ctx *ctx) {
  if (!ctx || !ctx->slab || !mrl_slab_fast_path(ctx)) return 0;
  return slab_gamma_tangent_fusable(ctx);
}

int mrl_slab_gamma_tangent_z_fwd(mrl_ctx *ctx, const double *d_F, const double *d_K, const double *d_mu, double *d_p,
                                 const double *d_r, double beta, double *d_x, double alpha_prev) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_TRY(check_fast(ctx, "mrl_slab_gamma_tangent_z_fwd", 0));
  if (!mrl_slab_gamma_tangent_fusable(ctx))
    return set_error(ctx, MRL_ERR_UNSUPPORTED, "mrl_slab_gamma_tangent_z_fwd: z lines of 32 ... 256 points in whole 512-point tiles only");
  if (!d_F || !d_K || !d_mu || !d_p || !d_r) return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_gamma_tangent_z_fwd: null buffer");
  const long long nx = ctx->n[0], nyl = ctx->nloc[1], nz = ctx->n[2], nzc = ctx->nrec[2];
  const long long npts = nx * nyl * nz, nspec = nx * nyl * nzc;
  MRL_TRY(ensure_work(ctx, 18, sizeof(cplx) * (size_t)(9 * nspec)));
  double *S = ctx->d_red + kScalarBase + 8;
  double *hs = ctx->h_red + 48 + 4 * (ctx->scalar_ring++ & 3);  // pinned source (ring of four quadruples, see put_scalars in mech.hip)
  hs[0] = beta;
  hs[1] = 1.0;
  hs[2] = alpha_prev;
  hs[3] = 1.0;
  MRL_HIP(ctx, hipMemcpyAsync(S, hs, 4 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ProfScope ps(ctx, "slab_gamma_z_fwd_tangent_dir", 8.0 * npts * ((d_x ? 6 : 4) * 9 + 2) + 16.0 * nspec * 9);
  MRL_TRY(gamma_z_fwd_tangent_launch(ctx, d_F, d_K, d_mu, d_p, d_r, S, 0, 1, reinterpret_cast<cplx *>(ctx->d_work[18]), npts,
                                     nx * nyl, (int)nz, 72.0 * (double)npts >= 96.0e6, d_x, 2, 3));
  ctx->gamma_z_ready = true;
  return MRL_OK;
}

int mrl_slab_gamma_row_fwd(mrl_ctx *ctx, int row, const double *d_A_fm, double *d_send) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_TRY(check_fast(ctx, "mrl_slab_gamma_row_fwd", row));
  if (!d_send) return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_gamma_row_fwd: null buffer");
  cplx *const *tab;
  MRL_TRY(local_tab(ctx, 2, d_send, sizeof(cplx) * 3 * (size_t)(ctx->nrec[0] * ctx->nloc[1] * ctx->nrec[2]), &tab));
  return slab_gamma_row_fwd(ctx, row, d_A_fm, tab, SignalArgs{});
}

int mrl_slab_gamma_row_mid(mrl_ctx *ctx, double *d_recv_inout, double scale) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_TRY(check_fast(ctx, "mrl_slab_gamma_row_mid", 0));
  if (!d_recv_inout) return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_gamma_row_mid: null buffer");
  cplx *const *tab;  // in place: chunk p is written back where it was read
  MRL_TRY(local_tab(ctx, 3, d_recv_inout, sizeof(cplx) * 3 * (size_t)(ctx->nrec[0] * (ctx->n[1] / ctx->nranks) * ctx->nrec[2]), &tab));
  return slab_gamma_row_mid(ctx, d_recv_inout, tab, SignalArgs{}, scale);
}

int mrl_slab_gamma_row_inv(mrl_ctx *ctx, int row, const double *d_recv, double *d_out_fm, const double *d_dotv_fm) {
  if (!ctx) return MRL_ERR_INVALID;
  MRL_TRY(check_fast(ctx, "mrl_slab_gamma_row_inv", row));
  if (!d_recv || !d_out_fm) return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_gamma_row_inv: null buffer");
  return slab_gamma_row_inv(ctx, row, d_recv, d_out_fm, d_dotv_fm);
}

int mrl_slab_gamma_dot(mrl_ctx *ctx, double *h_local) {
  if (!ctx) return MRL_ERR_INVALID;
  if (!h_local || ctx->gamma_dot_nb <= 0)
    return set_error(ctx, MRL_ERR_INVALID, "mrl_slab_gamma_dot: no dot-product rows pending");
  double *slot = ctx->d_red + kScalarBase;
  MRL_TRY(mrl::reduce_finalize_from(ctx, ctx->d_work[3], 3 * ctx->gamma_dot_nb, slot));
  ctx->gamma_dot_nb = 0;
  return read_scalars(ctx, slot, 1, h_local);
}

}  // extern "C"
