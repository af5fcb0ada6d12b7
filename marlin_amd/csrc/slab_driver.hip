// Library-owned multi-GPU drivers on slab contexts with an attached communicator (comm.hip): the solver-level entry points
// (mrl_ch_substeps, mrl_fft_r2c / mrl_fft_c2r, mrl_mech_newton_cg, the reductions) then behave as on a serial context and the
// library performs the global transposes itself -- what DomainAction::fftSlab / ifftSlab do from C++ inside the reference
// (src/actions/DomainAction.C:869-1019), here without host staging:
//
//   Cahn-Hilliard substep (AdamsBashforthMoulton.C:60-101 over fftSlab / ifftSlab), per kz sub-block s:
//     Z / EZ  z pass(es)                                             local
//     A_s     forward x pass, scattered into the peers' receive buffers (PEER_STORE) or a send buffer   -> exchange F_s
//     B_s     y pass + k-space update + inverse y pass, scattered likewise                               -> exchange I_s
//     C_s     inverse x pass                                          local
//     E       inverse z pass
//   One HIP stream carries all kernels of a rank.  With PEER_STORE the x and y pass kernels ARE the exchange: their stores go
//   over xGMI while they compute, and their last workgroup raises the arrival flags.  Write-after-read safety of the receive
//   buffers needs no extra handshake: a rank overwrites F_s of a peer only after it has consumed that peer's I_s of the
//   previous substep, which the peer sent after it had finished reading F_s (and symmetrically for I_s).
//
//   Plain transforms (DomainAction::fft / ifft in FFT_SLAB mode) have no such request / response structure, so their
//   exchanges use an explicit acknowledgement flag per receive buffer.
#include <algorithm>

#include "comm.h"
#include "slab_stages.h"

namespace mrl {

int ch_check_params(mrl_ctx *ctx, const mrl_ch_params *p, ChP &cp);
int reduce_async(mrl_ctx *ctx, int op, const double *a, const double *b, long long n, double *d_scalar);

struct ChPipe {
  bool built = false;
  int nsub = 0, transport = -1;
  bool carry = false, fast = false;
  bool local_only = false;            // MRL_OPT_EXPERIMENT bit 64: no exchange at all (timing of the rank-local kernels)
  int dense_planes = -1;              // MRL_OPT_EXPERIMENT bit 1 << 23 at build time (it changes the message sizes)
  std::vector<Xchg> fwd2, fwd1, inv;  // two-field forward, one-field forward (carry-over), inverse; one per sub-block
  double *cbar = nullptr;             // carried spectrum [x_me][ny][pitch]
};

struct FftPipe {
  bool built = false;
  int transport = -1;
  Xchg fwd, inv;
  int ack_fwd = -1, ack_inv = -1;  // acknowledgement channels
  unsigned long long ack_fwd_epoch = 0, ack_inv_epoch = 0;
  unsigned long long ack_fwd_base = 0, ack_inv_base = 0;  // epochs the channels were taken over at (nothing to acquire yet)
};

struct MechPipe {
  bool built = false, built_all = false;
  int transport = -1, transport_all = -1;
  Xchg fwd[3], inv[3];     // row pipeline: one pair per tensor row
  Xchg fwd_all, inv_all;   // all rows per launch: one pair of nine-field exchanges
};

// FFT_PENCIL: the four staged exchanges of DomainAction::fftPencil / ifftPencil (DomainAction.C:1105-1404), each with its own
// acknowledgement channel (a receive buffer is only overwritten once every peer has consumed its previous contents)
struct PencilPipe {
  bool built = false;
  int transport = -1;
  Xchg x[4];                       // 0: stage 1 forward, 1: stage 2 forward, 2: stage 2 inverse, 3: stage 1 inverse
  int ack[4] = {-1, -1, -1, -1};
  unsigned long long ack_epoch[4] = {0, 0, 0, 0}, ack_base[4] = {0, 0, 0, 0};
};

struct SlabPipes {
  ChPipe ch;
  FftPipe fft;
  MechPipe mech;
  PencilPipe pencil;
};

static int comm_fail(mrl_ctx *ctx, int rc) {
  if (rc != MRL_OK && ctx->comm) set_error(ctx, rc, "%s", ctx->comm->err.c_str());
  return rc;
}
#define MRL_COMM(ctx, expr)                         \
  do {                                              \
    int rc__ = (expr);                              \
    if (rc__ != MRL_OK) return comm_fail(ctx, rc__); \
  } while (0)

long long slab_verify_count(mrl_ctx *ctx, bool reset) {
  if (!ctx->comm || !ctx->comm->d_verify) return 0;
  unsigned long long v = 0;
  if (hipStreamSynchronize(ctx->stream) != hipSuccess ||
      hipMemcpy(&v, ctx->comm->d_verify, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  if (reset && hipMemset(ctx->comm->d_verify, 0, sizeof(v)) != hipSuccess) return -1;
  return (long long)v;
}

int slab_comm_check(mrl_ctx *ctx) {
  if (!ctx->comm) return MRL_OK;
  return comm_fail(ctx, comm_check(ctx->comm));
}

static void ch_pipe_destroy(mrl_ctx *ctx, ChPipe &P) {
  if (!ctx->comm) return;
  for (auto &x : P.fwd2) xchg_destroy(ctx->comm, &x);
  for (auto &x : P.fwd1) xchg_destroy(ctx->comm, &x);
  for (auto &x : P.inv) xchg_destroy(ctx->comm, &x);
  P.fwd2.clear();
  P.fwd1.clear();
  P.inv.clear();
  if (P.cbar) (void)hipFree(P.cbar);
  P.cbar = nullptr;
  P.built = false;
}

void slab_pipes_destroy(mrl_ctx *ctx) {
  if (!ctx->pipes) return;
  ch_pipe_destroy(ctx, ctx->pipes->ch);
  if (ctx->comm) {
    xchg_destroy(ctx->comm, &ctx->pipes->fft.fwd);
    xchg_destroy(ctx->comm, &ctx->pipes->fft.inv);
    if (ctx->pipes->fft.ack_fwd >= 0) comm_free_channel(ctx->comm, ctx->pipes->fft.ack_fwd, ctx->pipes->fft.ack_fwd_epoch);
    if (ctx->pipes->fft.ack_inv >= 0) comm_free_channel(ctx->comm, ctx->pipes->fft.ack_inv, ctx->pipes->fft.ack_inv_epoch);
    for (int r = 0; r < 3; ++r) {
      xchg_destroy(ctx->comm, &ctx->pipes->mech.fwd[r]);
      xchg_destroy(ctx->comm, &ctx->pipes->mech.inv[r]);
    }
    xchg_destroy(ctx->comm, &ctx->pipes->mech.fwd_all);
    xchg_destroy(ctx->comm, &ctx->pipes->mech.inv_all);
    for (int i = 0; i < 4; ++i) {
      xchg_destroy(ctx->comm, &ctx->pipes->pencil.x[i]);
      if (ctx->pipes->pencil.ack[i] >= 0) comm_free_channel(ctx->comm, ctx->pipes->pencil.ack[i], ctx->pipes->pencil.ack_epoch[i]);
    }
  }
  delete ctx->pipes;
  ctx->pipes = nullptr;
}

// the context leaves its communicator's list (mrl_ctx_destroy, re-attachment); the pipes are gone by then
void slab_detach_comm(mrl_ctx *ctx) {
  if (!ctx->comm) return;
  auto &v = ctx->comm->attached;
  v.erase(std::remove(v.begin(), v.end(), ctx), v.end());
  ctx->comm = nullptr;
}

static int need_comm(mrl_ctx *ctx, const char *what) {
  if (!ctx->slab && !ctx->pencil) return set_error(ctx, MRL_ERR_INVALID, "%s: not a slab context", what);
  if (!ctx->comm)
    return set_error(ctx, MRL_ERR_INVALID, "%s on a slab context needs a communicator (mrl_ctx_attach_comm), or use the staged mrl_slab_* entry points", what);
  if (!ctx->pipes) ctx->pipes = new SlabPipes();
  return MRL_OK;
}

// an exchange whose producer scatters through the table: direct peer stores when the transport allows, a send buffer otherwise
static int prepare_table(mrl_ctx *ctx, Xchg *x, bool scatter_capable) {
  mrl_comm *c = ctx->comm;
  const bool direct = scatter_capable && xchg_direct(c) && (c->ipc_ok || c->nranks == 1);
  if (!direct && !x->send && x->send_bytes) {
    void *s = nullptr;
    if (hipMalloc(&s, x->send_bytes) != hipSuccess) return set_error(ctx, MRL_ERR_NOMEM, "hipMalloc of a %zu byte send buffer failed", x->send_bytes);
    x->send = static_cast<double *>(s);
  }
  MRL_COMM(ctx, xchg_build_table(c, x, direct));
  return MRL_OK;
}

// ---- Cahn-Hilliard pipeline ---------------------------------------------------------------------------------------------
static int ch_pipe_build(mrl_ctx *ctx) {
  ChPipe &P = ctx->pipes->ch;
  mrl_comm *c = ctx->comm;
  const long long nzc = ctx->dim == 3 ? ctx->nrec[2] : 1;
  int nsub = ctx->opt_nsub < 1 ? 1 : ctx->opt_nsub;
  if (nsub > nzc) nsub = (int)nzc;
  const bool carry = ctx->opt_carry != 0;
  const bool local_only = (ctx->exp & 64) != 0;
  const int dense_planes = (ctx->exp >> 23) & 1;
  if (P.built && P.nsub == nsub && P.carry == carry && P.transport == c->transport && P.local_only == local_only && P.dense_planes == dense_planes)
    return MRL_OK;
  if (P.built && (P.nsub != nsub || P.carry != carry || P.dense_planes != dense_planes)) {
    MRL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ch_pipe_destroy(ctx, P);
  }
  const int R = ctx->nranks;
  P.fast = slab_fast_ok(ctx) != 0;
  if (!P.built) {
    P.fwd2.resize(nsub);
    P.inv.resize(nsub);
    if (carry) P.fwd1.resize(nsub);
    std::vector<int64_t> sc(R), rc(R);
    std::vector<size_t> sb(R), rb(R);
    auto make = [&](Xchg *x, int s, int forward, int mode) -> int {
      MRL_TRY(mrl_slab_ch_counts(ctx, s, nsub, forward, mode, sc.data(), rc.data()));
      for (int p = 0; p < R; ++p) {
        sb[p] = sizeof(cplx) * (size_t)sc[p];
        rb[p] = sizeof(cplx) * (size_t)rc[p];
      }
      MRL_COMM(ctx, xchg_create(c, x, sb.data(), rb.data(), false));
      return MRL_OK;
    };
    for (int s = 0; s < nsub; ++s) {
      MRL_TRACE("ch_pipe_build: creating the forward exchange of sub-block %d", s);
      MRL_TRY(make(&P.fwd2[s], s, 1, MRL_CARRY_NONE));
      MRL_TRACE("ch_pipe_build: creating the inverse exchange of sub-block %d", s);
      MRL_TRY(make(&P.inv[s], s, 0, MRL_CARRY_NONE));
      MRL_TRACE("ch_pipe_build: exchanges of sub-block %d created", s);
      if (carry) MRL_TRY(make(&P.fwd1[s], s, 1, MRL_CARRY_IN));
    }
    if (carry) {
      const size_t bytes = sizeof(cplx) * (size_t)(ctx->nrec[0] * ctx->nrec[1] * (ctx->dim == 3 ? mrl_slab_ch_spec_pitch(ctx) : 1));
      MRL_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&P.cbar), bytes));
    }
  } else {
    MRL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    MRL_COMM(ctx, comm_barrier(c));
  }
  for (int s = 0; s < nsub; ++s) {
    MRL_TRY(prepare_table(ctx, &P.fwd2[s], P.fast && !local_only));
    MRL_TRY(prepare_table(ctx, &P.inv[s], P.fast && !local_only));
    if (carry) MRL_TRY(prepare_table(ctx, &P.fwd1[s], P.fast && !local_only));
  }
  MRL_TRACE("ch_pipe_build: tables ready");
  P.local_only = local_only;
  P.dense_planes = dense_planes;
  P.nsub = nsub;
  P.carry = carry;
  P.transport = c->transport;
  P.built = true;
  return MRL_OK;
}

// post / wait with their stream time visible to the profiler (the wait is where an exposed exchange shows up)
static int post(mrl_ctx *ctx, Xchg *x, bool kernel_signalled, bool local_only) {
  if (local_only) return MRL_OK;
  ProfScope ps(ctx, "slab_exchange_post");
  MRL_COMM(ctx, xchg_post(ctx->comm, x, ctx->stream, kernel_signalled));
  return MRL_OK;
}
// post and wait of one exchange back to back (one kz sub-block in flight: nothing is enqueued between them)
static int verify(mrl_ctx *ctx, Xchg *x) {  // MRL_OPT_VERIFY_EXCHANGE (debug): see xchg_verify
  if (!ctx->opt_verify) return MRL_OK;
  ProfScope ps(ctx, "slab_exchange_verify");
  MRL_COMM(ctx, xchg_verify(ctx->comm, x, ctx->stream));
  return MRL_OK;
}
static int post_wait(mrl_ctx *ctx, Xchg *x, bool local_only) {
  if (local_only) return MRL_OK;
  {
    ProfScope ps(ctx, "slab_exchange_wait");
    MRL_COMM(ctx, xchg_post_wait(ctx->comm, x, ctx->stream, !(ctx->exp & 262144)));
  }
  return verify(ctx, x);
}
static int wait(mrl_ctx *ctx, Xchg *x, bool local_only) {
  if (local_only) return MRL_OK;
  {
    ProfScope ps(ctx, "slab_exchange_wait");
    MRL_COMM(ctx, xchg_wait(ctx->comm, x, ctx->stream));
  }
  return verify(ctx, x);
}

// the substep loop of TensorSolver::computeBuffer (TensorSolver.C:93-109) over slab transforms; ring semantics as mrl_ch_substeps
int slab_ch_substeps(mrl_ctx *ctx, const mrl_ch_params *p, const double *c_in, double *c_out, double *const *ring, int ring_size,
                     int *head, int *n_old, int pred, int count, int advance, double sub_dt, double *mu, bool dt_changed) {
  MRL_TRY(need_comm(ctx, "mrl_ch_substeps"));
  ChP cp;
  MRL_TRY(ch_check_params(ctx, p, cp));
  MRL_TRY(ch_pipe_build(ctx));
  ChPipe &P = ctx->pipes->ch;
  mrl_comm *c = ctx->comm;
  c->kernel_signals = (ctx->exp & 128) != 0;
  hipStream_t st = ctx->stream;
  const int nsub = P.nsub;
  const bool lo = P.local_only;
  for (int k = 0; k < count; ++k) {
    const int mode = !P.carry ? MRL_CARRY_NONE : (k == 0 ? MRL_CARRY_OUT : MRL_CARRY_IN);
    double *mu_k = (k == count - 1) ? mu : nullptr;
    MRL_TRACE("substep %d: z pass", k);
    if (k == 0) {
      MRL_TRY(mrl_slab_ch_z_fwd(ctx, p, c_in, mu_k, mode));
    } else {
      MRL_TRY(mrl_slab_ch_z_inv_fwd(ctx, p, mu_k, mode));
    }
    const int order = (dt_changed && k < pred) ? 0 : (*n_old < pred ? *n_old : pred);   // AdamsBashforthMoulton.C:90-91
    const int slot_new = (*head + 1) % ring_size;
    const double *old[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < order; ++i) old[i] = ring[((*head - i) % ring_size + ring_size) % ring_size];
    std::vector<Xchg> &F = mode == MRL_CARRY_IN ? P.fwd1 : P.fwd2;
    for (int s = 0; s < nsub; ++s) {
      long long k0, ks;
      MRL_TRY(slab_sub_range(ctx, s, nsub, &k0, &ks));
      MRL_COMM(ctx, xchg_begin(c, &F[s], st));
      MRL_TRACE("substep %d: forward x pass (fast %d)", k, (int)P.fast);
      if (P.fast) {
        const SignalArgs sig = xchg_signal_args(c, &F[s], 0);
        MRL_TRY(slab_ch_x_fwd_fast(ctx, (int)k0, (int)ks, reinterpret_cast<cplx *const *>(F[s].d_tab), sig, mode));
        if (nsub == 1 && !sig.counter) {
          MRL_TRY(post_wait(ctx, &F[s], lo));
        } else {
          MRL_TRY(post(ctx, &F[s], sig.counter != nullptr, lo));
        }
      } else {
        MRL_TRY(gen_x_fwd(ctx, k0, ks, F[s].send, mode));
        MRL_TRY(post(ctx, &F[s], false, lo));
      }
    }
    for (int s = 0; s < nsub; ++s) {
      long long k0, ks;
      MRL_TRY(slab_sub_range(ctx, s, nsub, &k0, &ks));
      MRL_TRACE("substep %d: wait F, y pass", k);
      if (!(P.fast && nsub == 1 && !c->kernel_signals)) MRL_TRY(wait(ctx, &F[s], lo));   // (else: waited for by post_wait above)
      MRL_COMM(ctx, xchg_begin(c, &P.inv[s], st));
      const double *recv = static_cast<const double *>(F[s].recv.local);
      if (P.fast) {
        const SignalArgs sig = xchg_signal_args(c, &P.inv[s], 0);
        MRL_TRY(slab_ch_kspace_fast(ctx, cp, (int)k0, (int)ks, recv, reinterpret_cast<cplx *const *>(P.inv[s].d_tab), sig, ring[slot_new], old,
                                    order, sub_dt, P.cbar, mode));
        if (nsub == 1 && !sig.counter) {
          MRL_TRY(post_wait(ctx, &P.inv[s], lo));
        } else {
          MRL_TRY(post(ctx, &P.inv[s], sig.counter != nullptr, lo));
        }
      } else {
        MRL_TRY(gen_kspace(ctx, cp, k0, ks, recv, P.inv[s].send, ring[slot_new], old, order, sub_dt, P.cbar, mode));
        MRL_TRY(post(ctx, &P.inv[s], false, lo));
      }
    }
    for (int s = 0; s < nsub; ++s) {
      MRL_TRACE("substep %d: wait I, inverse x pass", k);
      if (!(P.fast && nsub == 1 && !c->kernel_signals)) MRL_TRY(wait(ctx, &P.inv[s], lo));
      MRL_TRY(mrl_slab_ch_x_inv(ctx, s, nsub, static_cast<const double *>(P.inv[s].recv.local)));
    }
    if (advance && k < count - 1) {  // TensorSolver.C:105-106
      *head = slot_new;
      if (*n_old < pred) *n_old += 1;
    }
  }
  return mrl_slab_ch_z_inv(ctx, c_out);
}

// ---- plain slab transforms: DomainAction::fft / ifft with parallel_mode = FFT_SLAB ------------------------------------------
__global__ void k_ack_signal(unsigned long long *const *tab, int nranks, int me, int row, unsigned long long epoch) {
  const int p = threadIdx.x;
  if (p < nranks) __hip_atomic_store(tab[p] + row + me, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_ack_wait(const unsigned long long *row, int nranks, unsigned long long epoch, int *status, long long max_ticks) {
  const int p = threadIdx.x;
  if (p >= nranks) return;
  if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;  // (an earlier wait already timed out)
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(row + p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
    if (wall_clock64() - t0 > max_ticks) {
      atomicMax(status, 1 + p);
      return;
    }
    __builtin_amdgcn_s_sleep(4);
  }
}

static int fft_pipe_build(mrl_ctx *ctx) {
  FftPipe &P = ctx->pipes->fft;
  mrl_comm *c = ctx->comm;
  const int R = ctx->nranks;
  if (!P.built) {
    std::vector<int64_t> sc(R), rc(R);
    std::vector<size_t> sb(R), rb(R);
    for (int dir = 1; dir >= 0; --dir) {
      MRL_TRY(mrl_slab_counts(ctx, dir, sc.data(), rc.data(), nullptr, nullptr));
      for (int p = 0; p < R; ++p) {
        sb[p] = sizeof(cplx) * (size_t)sc[p];
        rb[p] = sizeof(cplx) * (size_t)rc[p];
      }
      MRL_COMM(ctx, xchg_create(c, dir ? &P.fwd : &P.inv, sb.data(), rb.data(), true));
    }
    P.ack_fwd = comm_alloc_channel(c);
    P.ack_inv = comm_alloc_channel(c);
    if (P.ack_fwd < 0 || P.ack_inv < 0) return set_error(ctx, MRL_ERR_UNSUPPORTED, "out of exchange channels");
    P.ack_fwd_epoch = P.ack_fwd_base = c->chan_epoch[P.ack_fwd];  // (reused channels continue from their previous owner's epoch)
    P.ack_inv_epoch = P.ack_inv_base = c->chan_epoch[P.ack_inv];
    P.built = true;
  }
  if (P.transport != c->transport) {
    MRL_TRY(prepare_table(ctx, &P.fwd, false));
    MRL_TRY(prepare_table(ctx, &P.inv, false));
    P.transport = c->transport;
  }
  return MRL_OK;
}

// my receive buffer of `x` may be overwritten again: tell every peer (stream-ordered behind the consumer)
static int ack_release(mrl_ctx *ctx, int ack_channel, unsigned long long *epoch) {
  mrl_comm *c = ctx->comm;
  *epoch += 1;
  if (!c->ipc_ok && c->nranks > 1) return MRL_OK;  // RCCL only: its own rendezvous orders the buffers
  hipLaunchKernelGGL(k_ack_signal, dim3(1), dim3(64), 0, ctx->stream, c->d_flag_tab, c->nranks, c->rank, ack_channel * kFlagRow, *epoch);
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}
// before pushing into the peers' buffers: all of them have released the previous contents
static int ack_acquire(mrl_ctx *ctx, int ack_channel, unsigned long long epoch, unsigned long long base) {
  mrl_comm *c = ctx->comm;
  if (epoch == base || (!c->ipc_ok && c->nranks > 1) || c->transport == MRL_TRANSPORT_RCCL) return MRL_OK;
  hipLaunchKernelGGL(k_ack_wait, dim3(1), dim3(64), 0, ctx->stream,
                     static_cast<const unsigned long long *>(c->flags.local) + (size_t)ack_channel * kFlagRow, c->nranks, epoch,
                     c->d_status, (long long)(c->timeout_s * (double)c->wall_khz * 1000.0));
  MRL_HIP(ctx, hipGetLastError());
  return MRL_OK;
}

int slab_fft_forward(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch) {
  MRL_TRY(need_comm(ctx, "mrl_fft_r2c"));
  MRL_TRY(fft_pipe_build(ctx));
  FftPipe &P = ctx->pipes->fft;
  mrl_comm *c = ctx->comm;
  const long long nreal = real_count_local(ctx), nspec = spec_count_local(ctx);
  for (long long b = 0; b < batch; ++b) {
    MRL_COMM(ctx, xchg_begin(c, &P.fwd, ctx->stream));
    MRL_TRY(slab_fwd_local(ctx, d_in + b * nreal, P.fwd.send));
    MRL_TRY(ack_acquire(ctx, P.ack_fwd, P.ack_fwd_epoch, P.ack_fwd_base));
    MRL_COMM(ctx, xchg_post(c, &P.fwd, ctx->stream));
    MRL_COMM(ctx, xchg_wait(c, &P.fwd, ctx->stream));
    MRL_TRY(slab_fwd_finish(ctx, static_cast<const double *>(P.fwd.recv.local), d_out + 2 * b * nspec));
    MRL_TRY(ack_release(ctx, P.ack_fwd, &P.ack_fwd_epoch));
  }
  return MRL_OK;
}

int slab_fft_inverse(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch) {
  MRL_TRY(need_comm(ctx, "mrl_fft_c2r"));
  MRL_TRY(fft_pipe_build(ctx));
  FftPipe &P = ctx->pipes->fft;
  mrl_comm *c = ctx->comm;
  const long long nreal = real_count_local(ctx), nspec = spec_count_local(ctx);
  for (long long b = 0; b < batch; ++b) {
    MRL_COMM(ctx, xchg_begin(c, &P.inv, ctx->stream));
    MRL_TRY(slab_inv_local(ctx, d_in + 2 * b * nspec, P.inv.send));
    MRL_TRY(ack_acquire(ctx, P.ack_inv, P.ack_inv_epoch, P.ack_inv_base));
    MRL_COMM(ctx, xchg_post(c, &P.inv, ctx->stream));
    MRL_COMM(ctx, xchg_wait(c, &P.inv, ctx->stream));
    MRL_TRY(slab_inv_finish(ctx, static_cast<const double *>(P.inv.recv.local), d_out + b * nreal));
    MRL_TRY(ack_release(ctx, P.ack_inv, &P.ack_inv_epoch));
  }
  return MRL_OK;
}

// ---- FFT_PENCIL: DomainAction::fftPencil / ifftPencil with library-owned exchanges -------------------------------------------------
int pencil_counts(const mrl_ctx *ctx, int stage, int forward, long long *send, long long *recv);   // pencil.hip
int pencil_fwd_x(mrl_ctx *ctx, const double *real_in, double *send1);
int pencil_fwd_y(mrl_ctx *ctx, const double *recv1, double *send2);
int pencil_fwd_z(mrl_ctx *ctx, const double *recv2, double *spec_out);
int pencil_inv_z(mrl_ctx *ctx, const double *spec_in, double *send2);
int pencil_inv_y(mrl_ctx *ctx, const double *recv2, double *send1);
int pencil_inv_x(mrl_ctx *ctx, const double *recv1, double *real_out);

static int pencil_pipe_build(mrl_ctx *ctx) {
  PencilPipe &P = ctx->pipes->pencil;
  mrl_comm *c = ctx->comm;
  const int R = ctx->nranks;
  if (!P.built) {
    std::vector<long long> sc(R), rc(R);
    std::vector<size_t> sb(R), rb(R);
    const int stage_of[4] = {1, 2, 2, 1}, fwd_of[4] = {1, 1, 0, 0};
    for (int i = 0; i < 4; ++i) {
      MRL_TRY(pencil_counts(ctx, stage_of[i], fwd_of[i], sc.data(), rc.data()));
      for (int p = 0; p < R; ++p) {
        sb[p] = sizeof(cplx) * (size_t)sc[p];
        rb[p] = sizeof(cplx) * (size_t)rc[p];
      }
      MRL_COMM(ctx, xchg_create(c, &P.x[i], sb.data(), rb.data(), true));
      P.ack[i] = comm_alloc_channel(c);
      if (P.ack[i] < 0) return set_error(ctx, MRL_ERR_UNSUPPORTED, "out of exchange channels");
      P.ack_epoch[i] = P.ack_base[i] = c->chan_epoch[P.ack[i]];
    }
    P.built = true;
  }
  if (P.transport != c->transport) {
    for (int i = 0; i < 4; ++i) MRL_TRY(prepare_table(ctx, &P.x[i], false));
    P.transport = c->transport;
  }
  return MRL_OK;
}

// one staged exchange: `produce` fills the send buffer, `consume` reads the receive buffer
template <class Produce, class Consume>
static int pencil_exchange(mrl_ctx *ctx, int i, Produce produce, Consume consume) {
  PencilPipe &P = ctx->pipes->pencil;
  mrl_comm *c = ctx->comm;
  MRL_COMM(ctx, xchg_begin(c, &P.x[i], ctx->stream));
  MRL_TRY(produce(P.x[i].send));
  MRL_TRY(ack_acquire(ctx, P.ack[i], P.ack_epoch[i], P.ack_base[i]));
  MRL_COMM(ctx, xchg_post(c, &P.x[i], ctx->stream));
  MRL_COMM(ctx, xchg_wait(c, &P.x[i], ctx->stream));
  MRL_TRY(consume(static_cast<const double *>(P.x[i].recv.local)));
  return ack_release(ctx, P.ack[i], &P.ack_epoch[i]);
}

int pencil_fft_forward(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch) {
  if (!ctx->comm) return set_error(ctx, MRL_ERR_INVALID, "mrl_fft_r2c on a pencil context needs a communicator (mrl_ctx_attach_comm)");
  MRL_TRY(need_comm(ctx, "mrl_fft_r2c"));
  MRL_TRY(pencil_pipe_build(ctx));
  PencilPipe &P = ctx->pipes->pencil;
  const long long nreal = real_count_local(ctx), nspec = spec_count_local(ctx);
  for (long long b = 0; b < batch; ++b) {
    const double *in = d_in + b * nreal;
    double *out = d_out + 2 * b * nspec;
    // stage 1: the x-transformed block leaves, my kx chunk of every y block of the group arrives -> y transform -> stage 2 leaves
    MRL_TRY(pencil_exchange(ctx, 0, [&](double *send) { return pencil_fwd_x(ctx, in, send); },
                            [&](const double *recv) {
                              MRL_COMM(ctx, xchg_begin(ctx->comm, &P.x[1], ctx->stream));
                              return pencil_fwd_y(ctx, recv, P.x[1].send);
                            }));
    // stage 2 (its send buffer was filled by the consumer above)
    MRL_TRY(pencil_exchange(ctx, 1, [&](double *) { return (int)MRL_OK; }, [&](const double *recv) { return pencil_fwd_z(ctx, recv, out); }));
  }
  return MRL_OK;
}

int pencil_fft_inverse(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch) {
  if (!ctx->comm) return set_error(ctx, MRL_ERR_INVALID, "mrl_fft_c2r on a pencil context needs a communicator (mrl_ctx_attach_comm)");
  MRL_TRY(need_comm(ctx, "mrl_fft_c2r"));
  MRL_TRY(pencil_pipe_build(ctx));
  PencilPipe &P = ctx->pipes->pencil;
  const long long nreal = real_count_local(ctx), nspec = spec_count_local(ctx);
  for (long long b = 0; b < batch; ++b) {
    const double *in = d_in + 2 * b * nspec;
    double *out = d_out + b * nreal;
    MRL_TRY(pencil_exchange(ctx, 2, [&](double *send) { return pencil_inv_z(ctx, in, send); },
                            [&](const double *recv) {
                              MRL_COMM(ctx, xchg_begin(ctx->comm, &P.x[3], ctx->stream));
                              return pencil_inv_y(ctx, recv, P.x[3].send);
                            }));
    MRL_TRY(pencil_exchange(ctx, 3, [&](double *) { return (int)MRL_OK; }, [&](const double *recv) { return pencil_inv_x(ctx, recv, out); }));
  }
  return MRL_OK;
}

// ---- Gamma operator on a slab-decomposed grid (FFTMechanics.C:74-84,105-106 over fftSlab / ifftSlab) ------------------------
int reduce_finalize_from(mrl_ctx *ctx, const double *partial, int nb, double *d_scalar);
int gamma_z_fwd_tangent_launch(mrl_ctx *ctx, const double *F, const double *K, const double *mu, double *p, const double *r,
                               const double *S, int i_num, int i_den, cplx *spec, long long npts, long long rows, int nz, bool nt,
                               double *x, int i_arz, int i_apAp);

// peer stores: the producing kernels are the exchange, so nothing is gained by splitting them per row and much is lost (launches of
// half a wave of workgroups at the rank-local sizes of BASELINE configs[4]); copies (copy engines, RCCL) overlap with the next
// row's transforms in the row pipeline.  MRL_OPT_EXPERIMENT bits 16384 / 32768 force the row pipeline / the batched form.
static bool mech_batched(const mrl_ctx *ctx) {
  if (!slab_gamma_batched_ok(ctx) || (ctx->exp & 16384) || ctx->comm->kernel_signals) return false;
  if (ctx->exp & 32768) return true;
  return xchg_direct(ctx->comm) && (ctx->comm->ipc_ok || ctx->comm->nranks == 1);
}

static int mech_pipe_build_all(mrl_ctx *ctx) {
  MechPipe &P = ctx->pipes->mech;
  mrl_comm *c = ctx->comm;
  const int R = ctx->nranks;
  if (!P.built_all) {
    std::vector<int64_t> sc(R), rc(R);
    std::vector<size_t> sb(R), rb(R);
    MRL_TRY(mrl_slab_gamma_counts(ctx, 1, sc.data(), rc.data()));
    for (int p = 0; p < R; ++p) {
      sb[p] = 3 * sizeof(cplx) * (size_t)sc[p];
      rb[p] = 3 * sizeof(cplx) * (size_t)rc[p];
    }
    MRL_COMM(ctx, xchg_create(c, &P.fwd_all, sb.data(), rb.data(), false));
    MRL_COMM(ctx, xchg_create(c, &P.inv_all, rb.data(), sb.data(), false));
    P.built_all = true;
  } else if (P.transport_all != c->transport) {
    MRL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    MRL_COMM(ctx, comm_barrier(c));
  }
  if (P.transport_all != c->transport) {
    MRL_TRY(prepare_table(ctx, &P.fwd_all, true));
    MRL_TRY(prepare_table(ctx, &P.inv_all, true));
    P.transport_all = c->transport;
  }
  return MRL_OK;
}

static int mech_pipe_build(mrl_ctx *ctx) {
  MechPipe &P = ctx->pipes->mech;
  mrl_comm *c = ctx->comm;
  const int R = ctx->nranks;
  if (!P.built) {
    std::vector<int64_t> sc(R), rc(R);
    std::vector<size_t> sb(R), rb(R);
    MRL_TRY(mrl_slab_gamma_counts(ctx, 1, sc.data(), rc.data()));
    for (int p = 0; p < R; ++p) {
      sb[p] = sizeof(cplx) * (size_t)sc[p];
      rb[p] = sizeof(cplx) * (size_t)rc[p];
    }
    for (int r = 0; r < 3; ++r) {
      MRL_COMM(ctx, xchg_create(c, &P.fwd[r], sb.data(), rb.data(), false));
      MRL_COMM(ctx, xchg_create(c, &P.inv[r], rb.data(), sb.data(), false));
    }
    P.built = true;
  } else if (P.transport != c->transport) {
    MRL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    MRL_COMM(ctx, comm_barrier(c));
  }
  if (P.transport != c->transport) {
    for (int r = 0; r < 3; ++r) {
      MRL_TRY(prepare_table(ctx, &P.fwd[r], true));
      MRL_TRY(prepare_table(ctx, &P.inv[r], true));
    }
    P.transport = c->transport;
  }
  return MRL_OK;
}

// out = scale * G(A) on FIELD-MAJOR local slabs [9][nx][nyl][nz] (planned shapes, equal partitions): three tensor rows, each
//   z + x passes -> exchange -> y pass with the projection -> exchange -> inverse x + z passes;
// row r + 1 is transformed while row r is on the wire.  A_fm == nullptr: the z spectra of the nine fields were left in the context
// by slab_gamma_tangent_z (x passes only).  dotv_fm: *d_dot = LOCAL sum(out * dotv), taken while `out` is in registers.
// Write-after-read safety: a rank refills a peer's forward buffer of row r only after it has consumed that peer's inverse message of
// the previous application, which the peer sent after its y pass had read the forward buffer (and symmetrically).
int slab_gamma_fm(mrl_ctx *ctx, const double *A_fm, double *out_fm, double scale, const double *dotv_fm, double *d_dot) {
  MRL_TRY(need_comm(ctx, "slab Gamma operator"));
  mrl_comm *c = ctx->comm;
  hipStream_t st = ctx->stream;
  c->kernel_signals = (ctx->exp & 128) != 0;
  if (mech_batched(ctx)) {  // one launch per stage over all nine fields
    MRL_TRY(mech_pipe_build_all(ctx));
    MechPipe &P = ctx->pipes->mech;
    MRL_COMM(ctx, xchg_begin(c, &P.fwd_all, st));
    MRL_TRY(slab_gamma_rows_fwd(ctx, A_fm, reinterpret_cast<cplx *const *>(P.fwd_all.d_tab), SignalArgs{}));
    MRL_COMM(ctx, xchg_post_wait(c, &P.fwd_all, st, !(ctx->exp & 262144)));
    MRL_TRY(verify(ctx, &P.fwd_all));
    MRL_COMM(ctx, xchg_begin(c, &P.inv_all, st));
    MRL_TRY(slab_gamma_rows_mid(ctx, static_cast<const double *>(P.fwd_all.recv.local), reinterpret_cast<cplx *const *>(P.inv_all.d_tab), SignalArgs{},
                                scale));
    MRL_COMM(ctx, xchg_post_wait(c, &P.inv_all, st, !(ctx->exp & 262144)));
    MRL_TRY(verify(ctx, &P.inv_all));
    ctx->gamma_dot_nb = 0;
    MRL_TRY(slab_gamma_rows_inv(ctx, static_cast<const double *>(P.inv_all.recv.local), out_fm, dotv_fm));
    if (dotv_fm) {
      MRL_TRY(reduce_finalize_from(ctx, ctx->d_work[3], ctx->gamma_dot_nb, d_dot));
      ctx->gamma_dot_nb = 0;
    }
    return MRL_OK;
  }
  MRL_TRY(mech_pipe_build(ctx));
  MechPipe &P = ctx->pipes->mech;
  for (int r = 0; r < 3; ++r) {
    MRL_COMM(ctx, xchg_begin(c, &P.fwd[r], st));
    const SignalArgs sig = xchg_signal_args(c, &P.fwd[r], 0);
    MRL_TRY(slab_gamma_row_fwd(ctx, r, A_fm, reinterpret_cast<cplx *const *>(P.fwd[r].d_tab), sig));
    MRL_COMM(ctx, xchg_post(c, &P.fwd[r], st, sig.counter != nullptr));
  }
  for (int r = 0; r < 3; ++r) {
    MRL_COMM(ctx, xchg_wait(c, &P.fwd[r], st));
    MRL_TRY(verify(ctx, &P.fwd[r]));
    MRL_COMM(ctx, xchg_begin(c, &P.inv[r], st));
    const SignalArgs sig = xchg_signal_args(c, &P.inv[r], 0);
    MRL_TRY(slab_gamma_row_mid(ctx, static_cast<const double *>(P.fwd[r].recv.local), reinterpret_cast<cplx *const *>(P.inv[r].d_tab), sig, scale));
    MRL_COMM(ctx, xchg_post(c, &P.inv[r], st, sig.counter != nullptr));
  }
  ctx->gamma_dot_nb = 0;
  for (int r = 0; r < 3; ++r) {
    MRL_COMM(ctx, xchg_wait(c, &P.inv[r], st));
    MRL_TRY(verify(ctx, &P.inv[r]));
    MRL_TRY(slab_gamma_row_inv(ctx, r, static_cast<const double *>(P.inv[r].recv.local), out_fm, dotv_fm));
  }
  if (dotv_fm) {
    MRL_TRY(reduce_finalize_from(ctx, ctx->d_work[3], 3 * ctx->gamma_dot_nb, d_dot));
    ctx->gamma_dot_nb = 0;
  }
  return MRL_OK;
}

// [x += (S[i_arz]/S[i_apAp]) p ;] p <- r + (S[i_num]/S[i_den]) p ; z spectra of K_dF(p) for the nine fields -> context scratch
// (FFTMechanics.C:107-108 fused with MarlinUtils.h:112 and the forward z pass; K_dF(p) is never written)
int slab_gamma_tangent_z(mrl_ctx *ctx, const double *F, const double *K, const double *mu, double *p, const double *r, const double *S,
                         int i_num, int i_den, double *x, int i_arz, int i_apAp) {
  const long long nx = ctx->n[0], nyl = ctx->nloc[1], nz = ctx->n[2], nzc = ctx->nrec[2];
  const long long npts = nx * nyl * nz, nspec = nx * nyl * nzc;
  MRL_TRY(ensure_work(ctx, 18, sizeof(cplx) * (size_t)(9 * nspec)));
  ProfScope ps(ctx, "slab_gamma_z_fwd_tangent_dir", 8.0 * npts * ((x ? 6 : 4) * 9 + 2) + 16.0 * nspec * 9);
  MRL_TRY(gamma_z_fwd_tangent_launch(ctx, F, K, mu, p, r, S, i_num, i_den, reinterpret_cast<cplx *>(ctx->d_work[18]), npts, nx * nyl, (int)nz,
                                     72.0 * (double)npts >= 96.0e6, x, i_arz, i_apAp));
  ctx->gamma_z_ready = true;
  return MRL_OK;
}

// out = scale * G(A) on VALUE-MAJOR local slabs of any shape: per component slab transform, projection of the field-major
// spectra, inverse (the generic stages; 2 x D*D blocking exchanges)
int slab_gamma_vm(mrl_ctx *ctx, const double *A_vm, double *out_vm, double scale) {
  MRL_TRY(need_comm(ctx, "slab Gamma operator"));
  const int dd = ctx->dim * ctx->dim;
  const long long npts = real_count_local(ctx), nspec = spec_count_local(ctx);
  MRL_TRY(ensure_work(ctx, 19, sizeof(double) * (size_t)(npts * dd + 2)));
  MRL_TRY(ensure_work(ctx, 20, sizeof(cplx) * (size_t)(nspec * dd)));
  double *fm = ctx->d_work[19], *spec = ctx->d_work[20];
  MRL_TRY(mrl_relayout(ctx, 1, A_vm, fm, npts, dd));
  MRL_TRY(slab_fft_forward(ctx, fm, spec, dd));
  MRL_TRY(mrl_slab_gamma_project(ctx, spec, scale));
  MRL_TRY(slab_fft_inverse(ctx, spec, fm, dd));
  return mrl_relayout(ctx, 0, fm, out_vm, npts, dd);
}

// sum over ranks of a device scalar produced on the context's stream; result on the host (one synchronisation)
int slab_allreduce_scalars(mrl_ctx *ctx, const double *d_local, int n, double *d_global, double *h_out) {
  MRL_COMM(ctx, comm_allreduce_device(ctx->comm, ctx->stream, d_local, n, d_global, h_out));
  return MRL_OK;
}

}  // namespace mrl

using namespace mrl;

extern "C" {

int mrl_ctx_attach_comm(mrl_ctx *ctx, mrl_comm *comm) {
  if (!ctx) return MRL_ERR_INVALID;
  if (!ctx->slab && !ctx->pencil) return set_error(ctx, MRL_ERR_INVALID, "mrl_ctx_attach_comm: not a slab or pencil context");
  if (comm && (comm->nranks != ctx->nranks || comm->rank != ctx->rank))
    return set_error(ctx, MRL_ERR_INVALID, "mrl_ctx_attach_comm: communicator is rank %d of %d, context rank %d of %d", comm->rank,
                     comm->nranks, ctx->rank, ctx->nranks);
  if (comm && comm->device != ctx->device)
    return set_error(ctx, MRL_ERR_INVALID, "mrl_ctx_attach_comm: communicator lives on device %d, context on device %d", comm->device, ctx->device);
  if (ctx->comm && ctx->comm != comm) {
    MRL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    slab_pipes_destroy(ctx);
    slab_detach_comm(ctx);
  }
  ctx->comm = comm;
  if (comm && std::find(comm->attached.begin(), comm->attached.end(), ctx) == comm->attached.end()) comm->attached.push_back(ctx);
  return MRL_OK;
}

}  // extern "C"
