// Fast path of the slab-decomposed Cahn-Hilliard substep (3-D, power-of-two extents, r2c on z, equal
// partitions), pipelined over `nsub` sub-blocks of the kz axis.  After the z pass every kz plane is an
// independent 2-D problem (x pass -> exchange -> y pass with the k-space update -> exchange -> inverse x pass),
// so sub-block s can be on the wire while sub-block s+1 is being transformed:
//   Z   k_z_fwd<CH>      c -> (c-hat_z, mu-hat_z) on the real slab, into work arrays [nx][nyl][nzc]
//   A_s k_pass_sub<x>    forward x of both fields for kz in K_s, written straight into the exchange layout
//                        [p][field][nxl_p][nyl][ksub]      (ordered by destination rank: no pack kernel)
//   -- all-to-all s (both fields in one message per peer) --
//   B_s k_ch_yfused      gathers lines along y from [p][field][nxl][nyl_p][ksub], forward y on both fields,
//                        Nhat = Mbar*mu-hat (dense reference layout), ABM predictor, 1/(1-dt*Lbar), inverse y,
//                        scattered into the inverse exchange layout [p][nxl][nyl_p][ksub]
//   -- all-to-all s --
//   C_s k_pass_sub<x>    inverse x from [p][nxl_p][nyl][ksub] into the work array [nx][nyl][nzc]
//   E   k_z_inv          c2r along z, 1/N
// (AdamsBashforthMoulton.C:60-101 with DomainAction::fftSlab/ifftSlab, DomainAction.C:869-1019.)
//
// Spectral carry-over (carry = MRL_CARRY_IN): the reference recomputes c-hat = fftSlab(c) every substep although c =
// ifftSlab(ubar) of the previous one; rfftn(irfftn(.)) is the identity up to rounding, so each rank keeps ubar
// [nxl][ny][nzc] (it is produced right there, in B_s) and uses it as the next c-hat.  Z then transforms mu only, A_s and the
// forward all-to-all carry ONE field instead of two: the exchange volume of a substep drops from 3 to 2 slab transposes,
// which is what bounds the multi-GPU rate on point-to-point xGMI links.
#include "ch_fused_body.h"
#include <atomic>

#include "fft_pow2_launch.h"
#include "slab_tables.h"

namespace mrl {

namespace p2 {

struct YFusedArgs {
  FusedCommon c;      // chat: received chunks (field 0 of each chunk; field 1 = mu-hat follows at +chunk)
  int nxl, nzc;       // local x extent, kz pitch of the rank-local spectral arrays (Nhat, cbar): mrl_slab_ch_spec_pitch
  int k0, ksub;       // kz sub-block
  int kp;             // row pitch of the exchange layouts (>= ksub: rows start on 128-byte lines)
  int nyl_shift;      // log2(ny / P)
  unsigned xp;        // elements between two x planes of a chunk (>= nyl * kp: slab_xplane)
  unsigned chunk;     // nxl * xp: elements of one field of one chunk
  int tiles_per_x;
  int idle_kl;        // column that the idle lanes of a row's last tile load (ksub - 1; experiment bit 1 << 25: 0, the round-2 behaviour)
  const double *kx, *ky, *kz;  // local reciprocal axes
  cplx *const *utab;  // ubar output: chunk p of the inverse exchange layout starts at utab[p] (a peer's receive buffer or the local send buffer)
  SignalArgs sig;     // arrival flags raised by the last workgroup (direct peer stores), or none
};

// BIG (with ALIGNED): exchange buffers or rank-local spectral arrays of 4 GiB and more (1024^3 on 2 or 4 GPUs: 4.3 / 2.2 GB per array,
// two fields per forward buffer).  The chunk index, the x plane and the row group of element j = q + m TPL are wave-uniform: they
// go into a 64-bit scalar part of the offset, the per-thread part (q rows + kz) stays 32-bit -- same access form, same registers.
// NTH (experiment bit 1 << 28, 512 points): stream the old / new Nhat arrays past the Infinity Cache as the serial x pass does
template <int N, int ORDER, bool SPEC_C, bool ALIGNED, bool BIG = false, bool NTH = false>
__global__ void __launch_bounds__(Plan<N>::NT, 2) k_ch_yfused(YFusedArgs a, const cplx *__restrict__ tw) {
  static_assert(!BIG || ALIGNED, "the 64-bit variant needs wave-uniform chunk offsets");
  constexpr int TPL = Plan<N>::TPL, T = Plan<N>::T;
  using Map = MapStrided<N>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx *W = reinterpret_cast<cplx *>(smem);
  cplx *X = W + N;
  double *KY = reinterpret_cast<double *>(X + Map::size);
  const int l = threadIdx.x % T, q = threadIdx.x / T;
  const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
  const int ix = logical / a.tiles_per_x;
  const int kl0 = (logical % a.tiles_per_x) * T + l;
  const bool valid = kl0 < a.ksub;
  const int kl = valid ? kl0 : a.idle_kl;  // (idle lanes of the last tile re-load its last valid column: the same 128-byte line, not another one)
  // element (ix, j, k0+kl), p = j >> nyl_shift, jl = j & (nyl-1)       [byte offsets]
  //   forward exchange layout  (p*2 + field)*chunk + ix*xp + jl*kp + kl
  //   inverse exchange layout   p*chunk            + ix*xp + jl*kp + kl
  //   dense layout              (ix*N + j)*nzc + k0 + kl
  const int sh = a.nyl_shift, msk = (1 << sh) - 1;
  const unsigned ksB = (unsigned)a.kp * 16u, chB = a.chunk * 16u, klB = (unsigned)kl * 16u;
  // the x plane of this workgroup is folded into the (wave-uniform) base pointers: the per-element offsets keep the shape they had
  // without padded planes (an extra uniform term in each of them cost 30 spilled VGPRs)
  const size_t xE = (size_t)ix * a.xp;
  FusedCommon fc = a.c;
  fc.chat += xE;
  fc.muhat += xE;
  // (with the carry-over only mu-hat is received: one field per chunk)
  // ALIGNED (ny/P is a multiple of the TPL threads of a line, the usual case): the chunk index and the row within the chunk of
  // element j = q + m TPL split into a wave-uniform part that depends on m only and the per-thread constant q ksB + klB, so the
  // 3 x 16 offsets are scalar arithmetic plus one VGPR instead of 48 VGPRs (which made the kernel spill 50-60 registers)
  const unsigned tq = (ALIGNED ? (unsigned)q * ksB : 0u) + klB;
  const int q0 = ALIGNED ? 0 : q;
  auto offf = [=](int m) {
    const int j = q0 + m * TPL;
    return (unsigned)(j >> sh) * (SPEC_C ? chB : 2u * chB) + (unsigned)(j & msk) * ksB + tq;
  };
  cplx *const *utab = a.utab;
  auto stu = [=](int m, cplx val) {  // (ALIGNED: the chunk index depends on m only -> the table entry is a scalar load)
    const int j = q0 + m * TPL;
    stc(utab[j >> sh] + xE, (unsigned)(j & msk) * ksB + tq, val);
  };
  // default cache policy for every stream: on the sub-block-sized working sets of the slab pipeline the non-temporal accesses of
  // the serial kernel cost 8-30 % (measured per variant with tools/slab_local_bench.py 8 256)
  if constexpr (BIG) {
    const unsigned long long ks64 = (unsigned long long)a.kp * 16ull, ch64 = (unsigned long long)a.chunk * 16ull;
    auto offf64 = [=](int m) {
      const int j = m * TPL;
      return BigOff{(unsigned long long)(j >> sh) * (SPEC_C ? ch64 : 2ull * ch64) + (unsigned long long)(j & msk) * ks64, tq};
    };
    auto stu64 = [=](int m, cplx val) {
      const int j = m * TPL;
      stc(utab[j >> sh] + xE, BigOff{(unsigned long long)(j & msk) * ks64, tq}, val);
    };
    const unsigned long long dx64 = (unsigned long long)ix * N * (unsigned long long)a.nzc * 16ull, dstep64 = (unsigned long long)(TPL * a.nzc) * 16ull;
    const unsigned dl = (unsigned)(q * a.nzc + a.k0 + kl) * 16u;
    auto offd64 = [=](int m) { return BigOff{dx64 + (unsigned long long)m * dstep64, dl}; };
    ch_fused_body<N, ORDER, false, Plan<N>::P / 2, SPEC_C, false, false, false>(fc, tw, a.ky, a.kx + ix, a.kz + a.k0 + kl, valid, q, l, offf64, OffSame{}, offd64, stu64, W, X, KY);
  } else {
    const unsigned d0 = (unsigned)(((long long)ix * N + q) * a.nzc + a.k0 + kl) * 16u, dstep = (unsigned)(TPL * a.nzc) * 16u;
    auto offd = [=](int m) { return d0 + (unsigned)m * dstep; };
    ch_fused_body<N, ORDER, false, Plan<N>::P / 2, SPEC_C, false, false, NTH>(fc, tw, a.ky, a.kx + ix, a.kz + a.k0 + kl, valid, q, l, offf, OffSame{}, offd, stu, W, X, KY);
  }
  signal_tail(a.sig);
}

template <int N, int ORDER, bool SPEC_C, bool ALIGNED, bool BIG = false, bool NTH = false>
static int launch_yfused_v(mrl_ctx *ctx, YFusedArgs a) {
  static std::atomic<bool> attr{false};  // (two host threads may both set the attribute: harmless, and no torn flag)
  constexpr size_t lds = lds_strided<N>() + sizeof(double) * N;
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY((set_lds_attr(ctx, k_ch_yfused<N, ORDER, SPEC_C, ALIGNED, BIG, NTH>, lds)));
    attr.store(true, std::memory_order_release);
  }
  constexpr int T = Plan<N>::T;
  a.tiles_per_x = (a.ksub + T - 1) / T;
  const long long nb = (long long)a.nxl * a.tiles_per_x;
  a.sig.expected = (unsigned)nb;
  hipLaunchKernelGGL((k_ch_yfused<N, ORDER, SPEC_C, ALIGNED, BIG, NTH>), dim3((unsigned)nb), dim3(Plan<N>::NT), lds, ctx->stream, a,
                     ctx->ax[1].d_tw);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

// y lengths for which the 64-bit variant is instantiated (long enough for >= 4 GiB exchange buffers at realistic shapes)
template <int N>
constexpr bool ybig_capable() {
  return N == 512 || N == 1024 || N == 2048;
}

// true when the byte offsets inside the forward exchange buffer (two fields) or a rank-local spectral array exceed 32 bits
static bool yfused_needs_big(const YFusedArgs &a, int nranks) {
  const double fwd = 2.0 * 16.0 * (double)a.chunk * nranks, dense = 16.0 * (double)a.nxl * (double)((1LL << a.nyl_shift) * nranks) * (double)a.nzc;
  return fwd >= 4294967296.0 || dense >= 4294967296.0;
}

template <int N, int ORDER, bool SPEC_C>
static int launch_yfused(mrl_ctx *ctx, YFusedArgs a) {
  constexpr int TPL = Plan<N>::TPL;
  const bool big = yfused_needs_big(a, ctx->nranks);
  if ((1 << a.nyl_shift) % TPL)  // (slab_fast_shift sends such partitions to the table-addressed kernel)
    return set_error(ctx, MRL_ERR_UNSUPPORTED, "slab y pass: %d rows per chunk are not a multiple of the %d threads of a line", 1 << a.nyl_shift, TPL);
  if constexpr (ybig_capable<N>()) {
    if (big) return launch_yfused_v<N, ORDER, SPEC_C, true, true>(ctx, a);
  }
  if (big) return set_error(ctx, MRL_ERR_UNSUPPORTED, "slab y pass: arrays of 4 GiB and more need ny in {512, 1024, 2048}");
  if constexpr (N == 512 && !SPEC_C) {
    // 512-point lines, the whole kz range in one sub-block, history arrays of 96 MB and more (512^3 on 8 GPUs: 138 MB): the old / new
    // Nhat arrays are streamed past the Infinity Cache as in the serial x pass, which leaves it to the exchange buffers the next pass
    // re-reads.  Same-box A/B (tools/ab_r04.sh, slab-local 512^3 / 8): y pass 160 -> 153-155 us, forward z pass 84 -> 79 us, sum of
    // the rank-local kernels 0.611 -> 0.599 ms; experiment bit 1 << 28 switches it off.  (Sub-block-sized working sets lose with
    // streaming accesses, profiles/HISTORY.md 3: hence only for nsub = 1.)
    const double hist_bytes = 16.0 * (double)a.nxl * (double)N * (double)a.nzc;
    if (a.k0 == 0 && a.ksub == ctx->nrec[2] && hist_bytes >= 96.0e6 && !(ctx->exp & (1 << 28)))
      return launch_yfused_v<N, ORDER, SPEC_C, true, false, true>(ctx, a);
  }
  return launch_yfused_v<N, ORDER, SPEC_C, true>(ctx, a);
}


// (fields of YFusedArgs used here: c, nxl, nzc, k0, ksub, tiles_per_x, kx, ky, kz, utab, sig)
template <int N, int ORDER, bool SPEC_C>
__global__ void __launch_bounds__(Plan<N>::NT, 2) k_ch_yfused_t(YFusedArgs a, YTabs t, const cplx *__restrict__ tw) {
  constexpr int TPL = Plan<N>::TPL, T = Plan<N>::T;
  using Map = MapStrided<N>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx *W = reinterpret_cast<cplx *>(smem);
  cplx *X = W + N;
  double *KY = reinterpret_cast<double *>(X + Map::size);
  const int l = threadIdx.x % T, q = threadIdx.x / T;
  const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
  const int ix = logical / a.tiles_per_x;
  const int kl0 = (logical % a.tiles_per_x) * T + l;
  const bool valid = kl0 < a.ksub;
  const int kl = valid ? kl0 : a.idle_kl;  // (idle lanes of the last tile re-load its last valid column: the same 128-byte line, not another one)
  const unsigned uix = (unsigned)ix, klB = (unsigned)kl * 16u;
  // element (field f, ix, j, k0 + kl) of the received forward buffer: yA[j] + f * yC[j] + ix * yB[j] + kl; of the chunk for rank
  // ych[j] in the inverse layout: ix * yB[j] + yD[j] + kl                                                        [byte offsets]
  auto offf = [=](int m) {
    const unsigned j = q + m * TPL;
    return (t.yA[j] + uix * t.yB[j]) * 16u + klB;
  };
  auto offm = [=](int m) {
    const unsigned j = q + m * TPL;
    return (t.yA[j] + (SPEC_C ? 0u : t.yC[j]) + uix * t.yB[j]) * 16u + klB;
  };
  cplx *const *utab = a.utab;
  auto stu = [=](int m, cplx val) {
    const unsigned j = q + m * TPL;
    stc(utab[t.ych[j]], (uix * t.yB[j] + t.yD[j]) * 16u + klB, val);
  };
  const unsigned d0 = (unsigned)(((long long)ix * N + q) * a.nzc + a.k0 + kl) * 16u, dstep = (unsigned)(TPL * a.nzc) * 16u;
  auto offd = [=](int m) { return d0 + (unsigned)m * dstep; };
  ch_fused_body<N, ORDER, false, Plan<N>::P / 2, SPEC_C, false, false, false>(a.c, tw, a.ky, a.kx + ix, a.kz + a.k0 + kl, valid, q, l, offf, offm, offd, stu, W, X, KY);
  signal_tail(a.sig);
}

template <int N, int ORDER, bool SPEC_C>
static int launch_yfused_t(mrl_ctx *ctx, YFusedArgs a, const YTabs &t) {
  static std::atomic<bool> attr{false};  // (two host threads may both set the attribute: harmless, and no torn flag)
  constexpr size_t lds = lds_strided<N>() + sizeof(double) * N;
  if (!attr.load(std::memory_order_acquire)) {
    MRL_TRY((set_lds_attr(ctx, k_ch_yfused_t<N, ORDER, SPEC_C>, lds)));
    attr.store(true, std::memory_order_release);
  }
  constexpr int T = Plan<N>::T;
  a.tiles_per_x = (a.ksub + T - 1) / T;
  const long long nb = (long long)a.nxl * a.tiles_per_x;
  a.sig.expected = (unsigned)nb;
  hipLaunchKernelGGL((k_ch_yfused_t<N, ORDER, SPEC_C>), dim3((unsigned)nb), dim3(Plan<N>::NT), lds, ctx->stream, a, t, ctx->ax[1].d_tw);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

}  // namespace p2

// threads per line of the plan of a fused-capable length
static int plan_tpl(long long n) {
  MRL_SWITCH_N(n, return p2::Plan<NN>::TPL);
  return 0;
}

int slab_fast_shift(const mrl_ctx *ctx) {
  if (!(ctx->dim == 3 && ctx->slab && ctx->spectrum == MRL_SPECTRUM_HALF && pow2_ok(ctx->n[0]) &&
        pow2_ok(ctx->n[1]) && pow2_ok(ctx->n[2])))
    return 0;
  if (ctx->exp & (1 << 24)) return 0;  // experiment: the table-addressed kernels on a shape that has the shift-addressed ones (A/B, parity)
  // equal power-of-two partitions: chunk addressing by shifts
  const long long nyl = ctx->n[1] / ctx->nranks, nxl = ctx->n[0] / ctx->nranks;
  if (nyl * ctx->nranks != ctx->n[1] || nxl * ctx->nranks != ctx->n[0] || (nyl & (nyl - 1)) || (nxl & (nxl - 1))) return 0;
  for (int p = 0; p < ctx->nranks; ++p)
    if (ctx->part_real[p] != nyl || ctx->part_recip[p] != nxl) return 0;
  // the y kernels want the rows of a chunk to be a multiple of the threads that share a line (chunk index and row offset of an element
  // then split into a wave-uniform and a per-thread part); chunks of fewer rows take the table-addressed kernels
  if (nyl % plan_tpl(ctx->n[1])) return 0;
  // the x passes index with 32-bit ELEMENT offsets inside a two-field exchange buffer: arrays below 32 GiB
  const double count = (double)ctx->n[0] * ((double)ctx->nloc[1] * (double)(ctx->nrec[2] + 8) + 32.0);
  if (2.0 * count >= 4294967296.0) return 0;
  // the y pass uses 32-bit BYTE offsets up to 4 GiB per two-field buffer; beyond that its 64-bit variant, which exists for
  // ny in {512, 1024, 2048} with ny/P a multiple of the threads per line
  if (32.0 * count >= 4294967296.0) {
    const long long ny = ctx->n[1];
    if (!(ny == 512 || ny == 1024 || ny == 2048) || nyl % (ny / 16) != 0) return 0;
  }
  return 1;
}

// Every other partition of a grid of planned lengths (the reference's own 200^3 example grid on 2 or 4 ranks: ny/P = 100 or 50; its
// 3-rank 64^3 test: 22 / 21 / 21 planes; device_weights) runs the same pipeline with TABLE-addressed chunks (k_pass_sub_t,
// k_ch_yfused_t): the chunk of an x plane / a y row and its offset inside the chunk come from small device tables instead of shifts.
// The verdict must be the same on every rank (the message sizes depend on it): it only uses global quantities.
static int slab_fast_table(const mrl_ctx *ctx) {
  if (!(ctx->dim == 3 && ctx->slab && ctx->spectrum == MRL_SPECTRUM_HALF && pow2_ok(ctx->n[0]) &&
        pow2_ok(ctx->n[1]) && pow2_ok(ctx->n[2])))
    return 0;
  if ((int)ctx->part_real.size() != ctx->nranks || (int)ctx->part_recip.size() != ctx->nranks) return 0;
  if (ctx->n[0] % 2) return 0;  // (the pair-wise z passes: nx * nyl even on every rank)
  long long ymax = 0, xmax = 0;
  for (int p = 0; p < ctx->nranks; ++p) {
    ymax = ctx->part_real[p] > ymax ? ctx->part_real[p] : ymax;
    xmax = ctx->part_recip[p] > xmax ? ctx->part_recip[p] : xmax;
  }
  // 32-bit BYTE offsets everywhere: the two-field forward buffer of the rank with the most rows, a rank-local spectral array
  const double fwd = 32.0 * (double)ctx->n[0] * ((double)ymax * (double)(ctx->nrec[2] + 8) + 32.0);
  const double dense = 16.0 * (double)xmax * (double)ctx->n[1] * (double)(ctx->nrec[2] + 8);
  return fwd < 4294967296.0 && dense < 4294967296.0;
}

int slab_fast_ok(const mrl_ctx *ctx) { return slab_fast_shift(ctx) || slab_fast_table(ctx); }

// the mechanics row pipelines keep 32-bit byte offsets throughout (three fields of a tensor row per exchange buffer)
int slab_mech_fast_ok(const mrl_ctx *ctx) {
  return slab_fast_shift(ctx) && 48.0 * (double)ctx->n[0] * ((double)ctx->nloc[1] * (double)(ctx->nrec[2] + 8) + 32.0) < 4294967296.0;
}
// ... the same pipelines with table-addressed chunks (k_pass_sub_mft, k_gamma_yfused_t): every other partition of planned extents whose
// nine-field exchange buffers fit 32-bit byte offsets.  Global quantities only: the same verdict on every rank.
int slab_mech_table_ok(const mrl_ctx *ctx) {
  if (slab_fast_shift(ctx) || !slab_fast_table(ctx)) return 0;
  long long ymax = 0, xmax = 0;
  for (int p = 0; p < ctx->nranks; ++p) {
    if ((ctx->n[0] * ctx->part_real[p]) % 2) return 0;  // pair-wise z passes on every rank
    ymax = ctx->part_real[p] > ymax ? ctx->part_real[p] : ymax;
    xmax = ctx->part_recip[p] > xmax ? ctx->part_recip[p] : xmax;
  }
  const double fwd = 9.0 * 16.0 * (double)ctx->n[0] * (double)ymax * (double)ctx->nrec[2];
  const double inv = 9.0 * 16.0 * (double)xmax * (double)ctx->n[1] * (double)ctx->nrec[2];
  return fwd < 4294967296.0 && inv < 4294967296.0;
}

// Row pitch (complex elements) of the exchange layouts of a kz sub-block of width ksub: rows start on 128-byte lines, so the x
// passes write (and the y pass gathers) whole lines instead of pieces that straddle two (the spectral extent nz/2+1 is odd).  With
// direct peer stores the padding never crosses a link; contiguous chunk pushes carry it (+2.7 % at 257 -> 264).
long long slab_kpitch(const mrl_ctx *ctx, long long ksub) { return slab_fast_ok(ctx) ? ((ksub + 7) & ~7LL) : ksub; }

// Pitch between two x planes (complex elements): [nyl][pitch] rows padded so that the plane pitch is an ODD number of 256-byte pieces.
// The x passes gather / scatter 256-byte pieces one plane apart; with the natural pitches of the power-of-two grids (512^3 / 8:
// 64 x 257 = 1028 pieces in the work arrays, 64 x 264 = 1056 pieces in the exchange layout) they fall on 32 resp. 4 of the 128
// memory channels (tools/ldsdma_probe.hip: -14 % time for the pass's bytes with both strides padded).  Experiment bit 1 << 23: dense.
static long long odd_plane(const mrl_ctx *ctx, long long elems) {
  if (ctx->exp & (1 << 23)) return elems;
  long long plane = (elems + 15) / 16 * 16;
  if ((plane / 16) % 2 == 0) plane += 16;
  return plane;
}
// x planes inside a chunk of the exchange layouts [p][field][x][y][kp]: what mrl_slab_ch_counts sizes the messages with
long long slab_xplane_of(const mrl_ctx *ctx, long long rows, long long kp) { return slab_fast_ok(ctx) ? odd_plane(ctx, rows * kp) : rows * kp; }
long long slab_xplane(const mrl_ctx *ctx, long long kp) { return slab_xplane_of(ctx, ctx->nloc[1], kp); }
// x planes of the rank-local work arrays [nx][nyl][nzc]
static long long slab_wplane(const mrl_ctx *ctx) { return odd_plane(ctx, ctx->nloc[1] * ctx->nrec[2]); }
static p2::ZLay slab_zlay(const mrl_ctx *ctx) {
  return p2::ZLay{(unsigned)ctx->nloc[1], (unsigned)(slab_wplane(ctx) - ctx->nloc[1] * ctx->nrec[2])};
}

static int ilog2(long long v) {
  int s = 0;
  while ((1LL << s) < v) ++s;
  return s;
}

// the device tables of the table-addressed kernels for exchange rows of pitch kp (built once per pitch, kept by the context)
int slab_tabs_get(mrl_ctx *ctx, long long kp, int nf, const SlabTabs **out) {
  const int dense = nf ? 1 : (ctx->exp >> 23) & 1;
  for (const SlabTabs &t : ctx->slab_tabs)
    if (t.kp == kp && t.dense == dense && t.nf == nf) {
      *out = &t;
      return MRL_OK;
    }
  const int P = ctx->nranks;
  const long long nx = ctx->n[0], ny = ctx->n[1], nxl = ctx->nrec[0], nyl = ctx->nloc[1];
  std::vector<unsigned> h((size_t)(4 * nx + 2 * P + 6 * ny));
  unsigned *xch = h.data(), *xoff = xch + nx, *fsz = xoff + nx, *cofi = fsz + P, *ych = cofi + P, *yD = ych + ny, *yB = yD + ny,
           *yC = yB + ny, *yA2 = yC + ny, *yA1 = yA2 + ny, *xin = yA1 + ny, *xfs = xin + nx;
  // x-plane pitch of a chunk with `rows` y rows, fields per chunk (forward / inverse)
  auto plane = [&](long long rows) { return nf ? rows * kp : slab_xplane_of(ctx, rows, kp); };
  const long long nff = nf ? nf : 2, nfi = nf ? nf : 1;
  const long long xp_me = plane(nyl);
  long long xb = 0, off = 0;
  for (int p = 0; p < P; ++p) {
    for (long long i = 0; i < ctx->part_recip[p]; ++i) {
      xch[xb + i] = (unsigned)p;
      xoff[xb + i] = (unsigned)(i * xp_me);
    }
    fsz[p] = (unsigned)(ctx->part_recip[p] * xp_me);
    cofi[p] = (unsigned)off;
    off += nfi * ctx->part_recip[p] * xp_me;
    xb += ctx->part_recip[p];
  }
  long long yb = 0, off2 = 0, off1 = 0;
  for (int p = 0; p < P; ++p) {
    const long long xp_p = plane(ctx->part_real[p]);
    for (long long i = 0; i < ctx->part_real[p]; ++i) {
      const long long j = yb + i;
      ych[j] = (unsigned)p;
      yD[j] = (unsigned)(i * kp);
      yB[j] = (unsigned)xp_p;
      yC[j] = (unsigned)(nxl * xp_p);
      yA2[j] = (unsigned)(off2 + i * kp);
      yA1[j] = (unsigned)(off1 + i * kp);
    }
    off2 += nff * nxl * xp_p;
    off1 += nxl * xp_p;
    yb += ctx->part_real[p];
  }
  if (xb != nx || yb != ny) return set_error(ctx, MRL_ERR_INVALID, "slab tables: the partitions do not cover the grid");
  for (long long n = 0; n < nx; ++n) {  // inverse x pass: ONE look-up between the plane index and the address (not chunk -> offset -> address)
    xin[n] = cofi[xch[n]] + xoff[n];
    xfs[n] = fsz[xch[n]];
  }
  if (16.0 * (double)off2 >= 4294967296.0 || 16.0 * (double)off >= 4294967296.0)
    return set_error(ctx, MRL_ERR_UNSUPPORTED, "slab tables: exchange buffers of 4 GiB and more");
  SlabTabs t;
  t.kp = kp;
  t.dense = dense;
  t.nf = nf;
  MRL_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&t.d), sizeof(unsigned) * h.size()));
  MRL_HIP(ctx, hipMemcpy(t.d, h.data(), sizeof(unsigned) * h.size(), hipMemcpyHostToDevice));
  t.xch = t.d;
  t.xoff = t.xch + nx;
  t.fsz = t.xoff + nx;
  t.cofi = t.fsz + P;
  t.ych = t.cofi + P;
  t.yD = t.ych + ny;
  t.yB = t.yD + ny;
  t.yC = t.yB + ny;
  t.yA2 = t.yC + ny;
  t.yA1 = t.yA2 + ny;
  t.xin = t.yA1 + ny;
  t.xfs = t.xin + nx;
  ctx->slab_tabs.push_back(t);
  *out = &ctx->slab_tabs.back();
  return MRL_OK;
}

// work arrays of the pipeline: slots 13, 14 = c-hat_z, mu-hat_z [nx][nyl][nzc].  The inverse x pass writes into the c-hat_z array
// again: its kz range s is dead once the forward x pass of sub-block s has run (which always precedes the inverse pass of s), and
// the fused z passes between two substeps then run in place (a workgroup reads its rows before it writes them) -- one array
// less to stream through the Infinity Cache.
static int slab_work(mrl_ctx *ctx, cplx **w_c, cplx **w_mu, cplx **w_inv) {
  const size_t bytes = sizeof(cplx) * (size_t)(ctx->n[0] * slab_wplane(ctx));
  MRL_TRY(ensure_work(ctx, 13, bytes));
  MRL_TRY(ensure_work(ctx, 14, bytes));
  *w_c = reinterpret_cast<cplx *>(ctx->d_work[13]);
  *w_mu = reinterpret_cast<cplx *>(ctx->d_work[14]);
  *w_inv = *w_c;
  return MRL_OK;
}

int slab_ch_z_fwd_fast(mrl_ctx *ctx, const ChP &cp, const double *c_in, double *mu, int carry) {
  const long long nx = ctx->n[0], nyl = ctx->nloc[1], nz = ctx->n[2], nzc = ctx->nrec[2];
  cplx *w_c, *w_mu, *w_inv;
  MRL_TRY(slab_work(ctx, &w_c, &w_mu, &w_inv));
  p2::ChDev chp{cp.family, cp.c0, cp.c1, cp.c2, {}};
  const p2::ZLay zl = slab_zlay(ctx);
  if (carry == MRL_CARRY_IN) {  // mu = f'(c) only, two lines per transform
    if ((nx * nyl) % 2) return set_error(ctx, MRL_ERR_UNSUPPORTED, "carry-over z pass needs an even number of local lines");
    ProfScope ps(ctx, "slab_Z_z_fwd", 8.0 * nx * nyl * nz + 16.0 * nx * nyl * nzc + (mu ? 8.0 * nx * nyl * nz : 0.0));
    if (cp.family == MRL_FE_PARSED) {
      MRL_TRY(parsed_z_fwd_launch(ctx, cp.parsed, (int)nz, 2, c_in, w_mu, nullptr, mu, nx * nyl / 2, zl.lpp, zl.pad));
    } else if (cp.family == MRL_FE_DOUBLE_WELL) {
      MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_fwd<NN, 2, MRL_FE_DOUBLE_WELL>(ctx, c_in, w_mu, nullptr, mu, chp, nx * nyl / 2, zl))));
    } else {
      MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_fwd<NN, 2, MRL_FE_PFHUB>(ctx, c_in, w_mu, nullptr, mu, chp, nx * nyl / 2, zl))));
    }
    return MRL_OK;
  }
  ProfScope ps(ctx, "slab_Z_z_fwd", 8.0 * nx * nyl * nz + 32.0 * nx * nyl * nzc + (mu ? 8.0 * nx * nyl * nz : 0.0));
  if (cp.family == MRL_FE_PARSED) {
    MRL_TRY(parsed_z_fwd_launch(ctx, cp.parsed, (int)nz, 1, c_in, w_c, w_mu, mu, nx * nyl, zl.lpp, zl.pad));
  } else if (cp.family == MRL_FE_DOUBLE_WELL) {
    MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_fwd<NN, 1, MRL_FE_DOUBLE_WELL>(ctx, c_in, w_c, w_mu, mu, chp, nx * nyl, zl))));
  } else {
    MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_fwd<NN, 1, MRL_FE_PFHUB>(ctx, c_in, w_c, w_mu, mu, chp, nx * nyl, zl))));
  }
  return MRL_OK;
}

int slab_ch_x_fwd_fast(mrl_ctx *ctx, int k0, int ksub, cplx *const *otab, const SignalArgs &sig, int carry) {
  const long long nx = ctx->n[0], nyl = ctx->nloc[1], nzc = ctx->nrec[2], nxl = ctx->nrec[0];
  cplx *w_c, *w_mu, *w_inv;
  MRL_TRY(slab_work(ctx, &w_c, &w_mu, &w_inv));
  const long long kp = slab_kpitch(ctx, ksub), xp = slab_xplane(ctx, kp), wp = slab_wplane(ctx);
  const unsigned chunk = (unsigned)(nxl * xp);
  const bool one = carry == MRL_CARRY_IN;  // mu-hat only
  if (!slab_fast_shift(ctx)) {  // table-addressed chunks (partitions that are not equal powers of two)
    const SlabTabs *tb;
    MRL_TRY(slab_tabs_get(ctx, kp, 0, &tb));
    p2::SubPassArgs a{};
    a.in[0] = (one ? w_mu : w_c) + k0;
    a.in[1] = w_mu + k0;
    a.otab = otab;
    a.sig = sig;
    a.rows = (int)nyl;
    a.cols = ksub;
    a.tcols = (int)kp;
    a.pitch_in = (unsigned)nzc;
    a.pitch_out = (unsigned)kp;
    a.sn_in = (unsigned)wp;
    const p2::SubPassTabs t{tb->xch, tb->xoff, tb->fsz, tb->xin, tb->xfs};
    ProfScope ps(ctx, "slab_A_x_fwd", (one ? 2.0 : 4.0) * 16.0 * nx * nyl * ksub);
    if (nx == 512 && !(ctx->exp & 1024))  // the wide plan, as on the shift-addressed path
      return p2::launch_pass_sub_wt<p2::Wide512, false>(ctx, a, t, ctx->ax[0].d_tw, one ? 1 : 2);
    if (one) {
      MRL_SWITCH_N(nx, MRL_TRY((p2::launch_pass_sub_t<NN, false, 1>(ctx, a, t, ctx->ax[0].d_tw))));
    } else {
      MRL_SWITCH_N(nx, MRL_TRY((p2::launch_pass_sub_t<NN, false, 2>(ctx, a, t, ctx->ax[0].d_tw))));
    }
    return MRL_OK;
  }
  p2::SubPassArgs a{};
  a.in[0] = (one ? w_mu : w_c) + k0;
  a.in[1] = w_mu + k0;
  a.otab = otab;
  a.fs_out = chunk;
  a.sig = sig;
  a.rows = (int)nyl;
  a.cols = ksub;
  a.pitch_in = (unsigned)nzc;
  a.pitch_out = (unsigned)kp;
  a.sn_in = (unsigned)wp;
  a.sn_out = (unsigned)xp;
  a.sh_in = 31;
  a.sh_out = ilog2(nxl);
  // tiles over the padded rows of the exchange layout: every wave stores whole 128-byte lines (the dense tiling stored 256-byte pieces
  // at 16-byte offsets, and the partial lines at their ends cost the memory a read-modify-write: 7 % of the write requests were short and
  // the read side stalled on DRAM credits 9 x as often as in the inverse pass).  Experiment bit 4096 = the dense tiling
  a.tcols = (ctx->exp & 4096) ? ksub : (int)kp;
  ProfScope ps(ctx, "slab_A_x_fwd", (one ? 2.0 : 4.0) * 16.0 * nx * nyl * ksub);
  // non-temporal stores of the exchange layout: A/B at 512^3 / 8 on one box -- rows of the natural odd pitch 182 / 162 -> 156 / 149 us,
  // rows padded to 128-byte lines (what is used) 171 -> 211 / 220 us: off
  a.nt_out = (ctx->exp & 512) ? 1 : 0;
  // 512-point lines: the wide plan (fft_pow2_wide.h: 32 points per thread, two stages, 256-byte segments), one field per launch, all
  // launches counting towards one arrival flag.  A/B at 512^3 / 8, same box, interleaved: forward pass 205-215 -> 182 us, inverse
  // pass 66-68 -> 59 us (experiment bit 1024 switches back to the 16-point plan)
  if (nx == 512 && !(ctx->exp & 1024)) {
    const int nf = one ? 1 : 2;
    if (!(ctx->exp & 8192)) return p2::launch_pass_sub_w<p2::Wide512, false>(ctx, a, ctx->ax[0].d_tw, nf);  // both fields in one launch
    a.sig.expected = (unsigned)nf * (unsigned)(((long long)a.rows * a.tcols + p2::Wide512::T - 1) / p2::Wide512::T);
    for (int f = 0; f < nf; ++f) {  // experiment bit 8192: one launch per field, all counting towards one arrival flag
      a.in[0] = f == 0 ? a.in[0] : a.in[1];
      a.fo_out = (unsigned)f * chunk;
      MRL_TRY((p2::launch_pass_sub_w<p2::Wide512, false>(ctx, a, ctx->ax[0].d_tw)));
    }
    return MRL_OK;
  }
  if (one) {
    MRL_SWITCH_N(nx, MRL_TRY((p2::launch_pass_sub<NN, false, 1>(ctx, a, ctx->ax[0].d_tw))));
  } else if (ctx->exp & 256) {   // experiment: one field per launch (the two launches count towards one arrival flag)
    unsigned nb = 0;
    MRL_SWITCH_N(nx, nb = pass_sub_blocks<NN>(a.rows, a.tcols));
    a.sig.expected = 2u * nb;
    MRL_SWITCH_N(nx, MRL_TRY((p2::launch_pass_sub<NN, false, 1>(ctx, a, ctx->ax[0].d_tw))));
    a.in[0] = a.in[1];
    a.fo_out = chunk;
    MRL_SWITCH_N(nx, MRL_TRY((p2::launch_pass_sub<NN, false, 1>(ctx, a, ctx->ax[0].d_tw))));
  } else {
    MRL_SWITCH_N(nx, MRL_TRY((p2::launch_pass_sub<NN, false, 2>(ctx, a, ctx->ax[0].d_tw))));
  }
  return MRL_OK;
}

int slab_ch_kspace_fast(mrl_ctx *ctx, const ChP &cp, int k0, int ksub, const double *recv, cplx *const *utab, const SignalArgs &sig,
                        double *Nhat_new, const double *const *Nhat_old, int order, double sub_dt, double *cbar, int carry) {
  const long long nxl = ctx->nrec[0], ny = ctx->n[1], nzc = ctx->nrec[2];
  const long long nyl = ny / ctx->nranks;
  const bool spec = carry == MRL_CARRY_IN;
  p2::YFusedArgs a{};
  a.kp = (int)slab_kpitch(ctx, ksub);
  a.xp = (unsigned)slab_xplane(ctx, a.kp);
  a.chunk = (unsigned)(nxl * a.xp);
  a.c.chat = reinterpret_cast<const cplx *>(recv);
  a.c.muhat = spec ? a.c.chat : a.c.chat + a.chunk;
  a.c.ubar = nullptr;  // scattered through utab
  a.utab = utab;
  a.sig = sig;
  a.c.Nnew = reinterpret_cast<cplx *>(Nhat_new);
  a.c.cbar = carry == MRL_CARRY_NONE ? reinterpret_cast<cplx *>(cbar) : nullptr;
  a.c.carry = carry == MRL_CARRY_NONE ? nullptr : reinterpret_cast<cplx *>(cbar);
  for (int i = 0; i < order; ++i) a.c.Nold[i] = reinterpret_cast<const cplx *>(Nhat_old[i]);
  for (int i = 0; i <= order; ++i) a.c.coef[i] = sub_dt * kBetaAB[order][i];
  a.nxl = (int)nxl;
  a.nzc = (int)((nzc + 7) & ~7LL);  // mrl_slab_ch_spec_pitch: rows of the rank-local spectral arrays start on 128-byte lines
  a.k0 = k0;
  a.ksub = ksub;
  a.idle_kl = (ctx->exp & (1 << 25)) ? 0 : ksub - 1;
  a.nyl_shift = ilog2(nyl);
  a.kx = ctx->d_k[0];
  a.ky = ctx->d_k[1];
  a.kz = ctx->d_k[2];
  a.c.M = cp.M;
  a.c.kappa = cp.kappa;
  a.c.dt = sub_dt;
  // NONE: recv 2, Nnew, send (+ cbar) ; OUT: + carry write ; IN: recv 1, carry read + write, Nnew, send
  ProfScope ps(ctx, "slab_B_y_fused", ((spec ? 5.0 : 4.0) + order + (cbar && !spec ? 1.0 : 0.0)) * 16.0 * nxl * ny * ksub);
  if (!slab_fast_shift(ctx)) {  // table-addressed chunks
    const SlabTabs *tb;
    MRL_TRY(slab_tabs_get(ctx, a.kp, 0, &tb));
    a.c.muhat = a.c.chat;  // (the field offset comes from the table: chunk-dependent)
    const p2::YTabs t{tb->ych, tb->yD, tb->yB, tb->yC, spec ? tb->yA1 : tb->yA2};
    if (spec) {
      switch (order) {
        case 0: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused_t<NN, 0, true>(ctx, a, t)))); break;
        case 1: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused_t<NN, 1, true>(ctx, a, t)))); break;
        case 2: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused_t<NN, 2, true>(ctx, a, t)))); break;
        case 3: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused_t<NN, 3, true>(ctx, a, t)))); break;
        default: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused_t<NN, 4, true>(ctx, a, t)))); break;
      }
      return MRL_OK;
    }
    switch (order) {
      case 0: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused_t<NN, 0, false>(ctx, a, t)))); break;
      case 1: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused_t<NN, 1, false>(ctx, a, t)))); break;
      case 2: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused_t<NN, 2, false>(ctx, a, t)))); break;
      case 3: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused_t<NN, 3, false>(ctx, a, t)))); break;
      default: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused_t<NN, 4, false>(ctx, a, t)))); break;
    }
    return MRL_OK;
  }
  if (spec) {
    switch (order) {
      case 0: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused<NN, 0, true>(ctx, a)))); break;
      case 1: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused<NN, 1, true>(ctx, a)))); break;
      case 2: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused<NN, 2, true>(ctx, a)))); break;
      case 3: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused<NN, 3, true>(ctx, a)))); break;
      default: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused<NN, 4, true>(ctx, a)))); break;
    }
    return MRL_OK;
  }
  switch (order) {
    case 0: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused<NN, 0, false>(ctx, a)))); break;
    case 1: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused<NN, 1, false>(ctx, a)))); break;
    case 2: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused<NN, 2, false>(ctx, a)))); break;
    case 3: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused<NN, 3, false>(ctx, a)))); break;
    default: MRL_SWITCH_N(ny, MRL_TRY((p2::launch_yfused<NN, 4, false>(ctx, a)))); break;
  }
  return MRL_OK;
}

int slab_ch_x_inv_fast(mrl_ctx *ctx, int k0, int ksub, const double *recv) {
  const long long nx = ctx->n[0], nyl = ctx->nloc[1], nzc = ctx->nrec[2], nxl = ctx->nrec[0];
  cplx *w_c, *w_mu, *w_inv;
  MRL_TRY(slab_work(ctx, &w_c, &w_mu, &w_inv));
  p2::SubPassArgs a{};
  a.in[0] = reinterpret_cast<const cplx *>(recv);
  a.out[0] = w_inv + k0;
  a.rows = (int)nyl;
  a.cols = ksub;
  const long long kp = slab_kpitch(ctx, ksub), xp = slab_xplane(ctx, kp);
  a.pitch_in = (unsigned)kp;
  a.pitch_out = (unsigned)nzc;
  a.sn_in = (unsigned)xp;
  a.sn_out = (unsigned)slab_wplane(ctx);
  a.sh_in = ilog2(nxl);
  a.cs_in = (unsigned)(nxl * xp);
  a.sh_out = 31;
  ProfScope ps(ctx, "slab_C_x_inv", 32.0 * nx * nyl * ksub);
  if (!slab_fast_shift(ctx)) {  // table-addressed chunks
    const SlabTabs *tb;
    MRL_TRY(slab_tabs_get(ctx, kp, 0, &tb));
    const p2::SubPassTabs t{tb->xch, tb->xoff, tb->fsz, tb->xin, tb->xfs};
    if (nx == 512 && !(ctx->exp & 1024)) return p2::launch_pass_sub_wt<p2::Wide512, true>(ctx, a, t, ctx->ax[0].d_tw);
    MRL_SWITCH_N(nx, MRL_TRY((p2::launch_pass_sub_t<NN, true, 1>(ctx, a, t, ctx->ax[0].d_tw))));
    return MRL_OK;
  }
  if (nx == 512 && !(ctx->exp & 1024)) return p2::launch_pass_sub_w<p2::Wide512, true>(ctx, a, ctx->ax[0].d_tw);
  MRL_SWITCH_N(nx, MRL_TRY((p2::launch_pass_sub<NN, true, 1>(ctx, a, ctx->ax[0].d_tw))));
  return MRL_OK;
}

// inverse z pass of substep k fused with the forward z pass of substep k + 1 (both are local to the rank: no exchange lies between
// them).  With the carry-over only mu travels forward, so the two real lines of a pair become ONE forward transform of (mu, mu').
int slab_ch_z_inv_fwd_fast(mrl_ctx *ctx, const ChP &cp, double *mu, int carry) {
  const long long nx = ctx->n[0], nyl = ctx->nloc[1], nz = ctx->n[2], nzc = ctx->nrec[2];
  if ((nx * nyl) % 2) return set_error(ctx, MRL_ERR_UNSUPPORTED, "fused z passes need an even number of local lines");
  cplx *w_c, *w_mu, *w_inv;
  MRL_TRY(slab_work(ctx, &w_c, &w_mu, &w_inv));
  const bool mu_only = carry == MRL_CARRY_IN;
  ProfScope ps(ctx, "slab_EZ_z_inv_fwd", (mu_only ? 2.0 : 3.0) * 16.0 * nx * nyl * nzc + (mu ? 8.0 * nx * nyl * nz : 0.0));
  const double scale = 1.0 / ((double)nx * (double)ctx->n[1] * (double)nz);
  p2::ChDev chp{cp.family, cp.c0, cp.c1, cp.c2, {}};
  const long long np = nx * nyl / 2;
  const p2::ZLay zl = slab_zlay(ctx);
  if (cp.family == MRL_FE_PARSED) {
    MRL_TRY(parsed_z_inv_fwd_launch(ctx, cp.parsed, (int)nz, w_inv, mu_only ? w_mu : w_c, w_mu, mu, scale, np, mu_only, zl.lpp, zl.pad));
  } else if (cp.family == MRL_FE_DOUBLE_WELL) {
    if (mu_only) {
      MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_inv_fwd<NN, MRL_FE_DOUBLE_WELL, true>(ctx, w_inv, w_mu, nullptr, mu, chp, scale, np, zl))));
    } else {
      MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_inv_fwd<NN, MRL_FE_DOUBLE_WELL, false>(ctx, w_inv, w_c, w_mu, mu, chp, scale, np, zl))));
    }
  } else {
    if (mu_only) {
      MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_inv_fwd<NN, MRL_FE_PFHUB, true>(ctx, w_inv, w_mu, nullptr, mu, chp, scale, np, zl))));
    } else {
      MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_inv_fwd<NN, MRL_FE_PFHUB, false>(ctx, w_inv, w_c, w_mu, mu, chp, scale, np, zl))));
    }
  }
  return MRL_OK;
}

int slab_ch_z_inv_fast(mrl_ctx *ctx, double *real_out) {
  const long long nx = ctx->n[0], nyl = ctx->nloc[1], nz = ctx->n[2], nzc = ctx->nrec[2];
  cplx *w_c, *w_mu, *w_inv;
  MRL_TRY(slab_work(ctx, &w_c, &w_mu, &w_inv));
  ProfScope ps(ctx, "slab_E_z_inv", 16.0 * nx * nyl * nzc + 8.0 * nx * nyl * nz);
  const double scale = 1.0 / ((double)nx * (double)ctx->n[1] * (double)nz);
  MRL_SWITCH_N(nz, MRL_TRY((p2::launch_z_inv<NN>(ctx, w_inv, real_out, scale, nx * nyl / 2, slab_zlay(ctx)))));
  return MRL_OK;
}

}  // namespace mrl
