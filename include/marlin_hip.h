/*
 * marlin_hip.h -- C ABI of the MI355X-native FFT spectral-solver inner loop for Marlin.
 *
 * Drop-in boundary: these entry points are what Marlin's MOOSE objects bind to in place of
 * their libTorch call sequences (see INTEGRATION.md for the shim classes).  Every function
 * cites the reference interface it replaces (paths relative to the idaholab/marlin tree).
 *
 * Conventions
 *   - all pointers named d_* are DEVICE pointers (HBM) owned by the caller and only borrowed
 *     for the call (+ until the next mrl_sync for the asynchronous work it enqueued);
 *   - real fields are dense row-major double, last spatial axis fastest: [nx][ny][nz]
 *     (= torch `.contiguous()` of the reference's buffers); spectra are interleaved
 *     (re,im) complex128 [nx][ny][nzc] with nzc = nz/2+1 (HALF spectrum, the layout of
 *     torch::fft::rfftn that DomainAction::fftSerial returns) or nzc = nz (FULL spectrum, the
 *     c2c layout of DomainAction::fftSlab);
 *   - every function returns MRL_OK (0) or a negative error code; the message is available
 *     from mrl_last_error().  No exceptions cross the boundary, no C++ or torch types appear;
 *   - work is enqueued on the context's HIP stream and is asynchronous unless stated.
 *   - one context per rank / per GPU; a context is not thread-safe (the reference drives each
 *     rank from one host thread, src/problems/TensorProblem.C:154-197).
 *   - one DEVICE per process (DomainAction.C:197-198: one MPI rank <-> one device): all contexts of a process must live on
 *     the same GPU -- per-kernel attributes (dynamic LDS sizes) are set once per process, on the device current at first use.
 */
#ifndef MARLIN_HIP_H
#define MARLIN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRL_ABI_VERSION 3

enum mrl_status {
  MRL_OK = 0,
  MRL_ERR_INVALID = -1,      /* bad argument (paramError in the reference) */
  MRL_ERR_UNSUPPORTED = -2,  /* valid request this build cannot serve */
  MRL_ERR_HIP = -3,          /* HIP runtime failure */
  MRL_ERR_NOMEM = -4,
  MRL_ERR_NOT_CONVERGED = -5, /* nl_max_its exceeded (FFTMechanics.C:159-161) */
  MRL_ERR_COMM = -6,          /* multi-GPU transport failure (a peer did not arrive in time, bootstrap / IPC / RCCL error) */
  MRL_ERR_IO = -7             /* file output failed (mrl_h5_*) */
};

enum mrl_spectrum {
  MRL_SPECTRUM_HALF = 0, /* r2c on the last axis: DomainAction.C:273-275 (parallel_mode NONE) */
  MRL_SPECTRUM_FULL = 1  /* c2c on every axis:    DomainAction.C:279-281 (parallel_mode FFT_SLAB) */
};

typedef struct mrl_ctx mrl_ctx;

/* Mirrors the [Domain] block / DomainAction ctor + gridChanged (src/actions/DomainAction.C:94-338). */
typedef struct mrl_domain {
  int32_t dim;             /* 1..3 */
  int64_t n[3];            /* nx, ny, nz (entries >= dim ignored) */
  double min[3], max[3];   /* xmin..zmax */
  int32_t device;          /* HIP device ordinal; <0 = current device */
  int32_t nranks, rank;    /* FFT_SLAB decomposition (DomainAction.C:510-566); nranks=1 = serial */
  const int64_t *weights;  /* device_weights (nranks entries) or NULL = equal */
  int32_t spectrum;        /* enum mrl_spectrum */
  void *stream;            /* hipStream_t to enqueue on; NULL = the HIP null stream (unless MRL_FLAG_OWN_STREAM) */
  int32_t flags;           /* MRL_FLAG_* */
} mrl_domain;

#define MRL_FLAG_OWN_STREAM 1 /* the context creates (and owns) a non-blocking stream; `stream` is ignored */
#define MRL_FLAG_SLAB 2       /* slab context even with nranks = 1 (the staged mrl_slab_* pipeline on one rank; exchanges are self-copies) */
#define MRL_FLAG_DENSE_SPECTRA 4 /* the Cahn-Hilliard history arrays of this context are dense [nx][ny][nzc] (see mrl_ch_spec_elems) */
#define MRL_FLAG_PENCIL 8        /* parallel_mode = FFT_PENCIL (DomainAction::partitionPencils, DomainAction.C:568-742): 3-D only,
                                    nranks = py * pz with both factors >= 2 (the reference's choice: mrl_pencil_factors).  Real space
                                    [nx][ny / py][nz / pz] (rank r: y block r % py, z block r / py); reciprocal space
                                    [(nx/2+1) / py][ny / pz][nz]: the r2c transform runs along X (DomainAction.C:282-284), kx is split
                                    over py, ky over pz, kz is complete.  mrl_fft_r2c / mrl_fft_c2r are DomainAction::fftPencil /
                                    ifftPencil (:1021-1047) with their four staged exchanges (:1105-1404) owned by the library;
                                    reductions are global, pointwise entry points (mrl_parsed_*, mrl_axpby, ...) work on the local
                                    blocks; mrl_ch_substep(s) run as the reference's operator sequence over these transforms
                                    (unfused); the mechanics entry points (mrl_mech_*, mrl_gamma_apply) return MRL_ERR_UNSUPPORTED */

/* ---- context ------------------------------------------------------------------------ */
int mrl_abi_version(void);
/* DomainAction::DomainAction + gridChanged + partitionSlabs */
int mrl_ctx_create(mrl_ctx **out, const mrl_domain *dom);
void mrl_ctx_destroy(mrl_ctx *ctx);
/* DomainAction::partitionPencils' choice of the process grid (DomainAction.C:574-618): among the factorisations nranks = py * pz with
 * py, pz >= 2, py <= min(ny, nx/2+1), pz <= min(nz, ny), the one with the smallest |py - pz| (first found wins).  Host only.
 * MRL_ERR_INVALID with the reference's message when no factorisation fits. */
int mrl_pencil_factors(int32_t nranks, const int64_t n[3], int32_t *py, int32_t *pz);
/* Host only (no GPU, no context): the block layout and the message sizes of the four staged exchanges of an FFT_PENCIL job for rank
 * `rank` of `nranks` on an n[0] x n[1] x n[2] grid -- what a caller that keeps its own exchange (MPI, torch.distributed) needs, and what
 * the library's own pipeline uses.  real_n / real_begin / recip_n / recip_begin: as mrl_local_shape.  stage 1 = inside the group of
 * equal z block (DomainAction.C:1105-1180 forward, :1331-1404 inverse), stage 2 = inside the group of equal kx block (:1182-1256,
 * :1258-1329); counts in COMPLEX elements per peer rank (zero outside the group), nranks entries each; any output may be NULL. */
int mrl_pencil_layout(int32_t nranks, int32_t rank, const int64_t n[3], int64_t real_n[3], int64_t real_begin[3], int64_t recip_n[3],
                      int64_t recip_begin[3], int64_t *stage1_fwd_send, int64_t *stage1_fwd_recv, int64_t *stage2_fwd_send,
                      int64_t *stage2_fwd_recv);
/* the process grid of a pencil context (MRL_ERR_INVALID on other contexts) */
int mrl_pencil_grid(const mrl_ctx *ctx, int32_t *py, int32_t *pz);
/* message of the last failing call on ctx (ctx may be NULL: failure of mrl_ctx_create) */
const char *mrl_last_error(const mrl_ctx *ctx);
/* hipStreamSynchronize on the context stream */
int mrl_sync(mrl_ctx *ctx);
int mrl_set_stream(mrl_ctx *ctx, void *stream);

/* Run-time options of a context (no environment variables are read by the library). */
enum mrl_option {
  MRL_OPT_EXPERIMENT = 0,  /* bit mask of A/B switches for kernel variants (tools/ only; 0 = product defaults) */
  MRL_OPT_SLAB_NSUB = 1,   /* kz sub-blocks the slab Cahn-Hilliard substep is pipelined over (default 1) */
  MRL_OPT_SLAB_CARRY = 2,  /* 1: spectral carry-over inside mrl_ch_substeps on slab contexts (see mrl_slab_ch_*; default 0 =
                              the reference's data flow, three slab transposes per substep) */
  MRL_OPT_VERIFY_EXCHANGE = 3,   /* debug, slab contexts with a communicator: after every arrival wait the receive buffer is re-read
                                    once with plain loads and once with system-scope loads behind a system-scope acquire; 64-bit
                                    words that differ (a stale cache line on the consumer's GPU: the memory-model argument of the
                                    peer-store transport, profiles/HISTORY.md 4.1a, does not hold on this node) are counted */
  MRL_OPT_VERIFY_MISMATCHES = 4, /* get: that count for the communicator of this context (synchronises); set 0: reset */
  MRL_OPT_CACHE_CHUNK_MB = 5     /* serial fused Cahn-Hilliard path, A/B switch (0 = off, the default): the plane-wise passes between
                                    two x passes (inverse y, fused z, forward y) run over chunks of x planes whose c-hat + mu-hat
                                    planes take this many MB, to meet in the 256 MiB Infinity Cache; results are bit-identical
                                    (measured slower on MI355X: profiles/HISTORY.md section 5) */
};
int mrl_ctx_set_option(mrl_ctx *ctx, int option, int64_t value);
int64_t mrl_ctx_get_option(const mrl_ctx *ctx, int option);

/* ---- multi-GPU transport: one process per GPU, the GPUs of ONE node (xGMI) ---------------------------------------------
 * Replaces the host-staged MPI point-to-point transposes of DomainAction::fftSlab / ifftSlab (src/actions/DomainAction.C:
 * 869-938, 940-1019: MPI_Isend / blocking MPI_Recv of host tensors in rank order) and adds the sum over ranks that the
 * reference's norms lack in parallel (DomainAction.C:1564-1567).  A communicator is created collectively by the `nranks`
 * processes of a job; `name` identifies the job on the node (a POSIX shared-memory object "/name" is the bootstrap channel:
 * barriers, the exchange of HIP IPC handles and of the RCCL unique id -- the MOOSE shim broadcasts a unique string over MPI,
 * see INTEGRATION.md).  Every rank owns receive buffers that all peers map through HIP IPC; an exchange is
 *   MRL_TRANSPORT_PEER_STORE  the producing kernels store their output chunks straight into the peers' receive buffers
 *                             (no send buffer, no copy), then raise a per-source arrival flag there
 *   MRL_TRANSPORT_PEER_COPY   the kernels write a local send buffer and the copy engines push the chunks to the peers
 *   MRL_TRANSPORT_RCCL        grouped ncclSend / ncclRecv on a side stream (librccl.so.1 is loaded at run time)
 * MRL_TRANSPORT_AUTO = PEER_STORE where HIP IPC works, else RCCL.  All ranks must select the same transport.
 * Once a communicator is attached to a slab context, the whole-solver entry points work on it as they do on a serial context:
 * mrl_ch_substeps, mrl_mech_newton_cg, mrl_fft_r2c / mrl_fft_c2r, and mrl_dot / mrl_norm2 / mrl_sum / mrl_minmax / mrl_average
 * return the GLOBAL value.  Consumers wait for arrivals on the GPU with a bounded spin: a peer that never arrives makes the
 * next synchronising call return MRL_ERR_COMM instead of hanging the device. */
typedef struct mrl_comm mrl_comm;
enum mrl_transport {
  MRL_TRANSPORT_AUTO = 0,
  MRL_TRANSPORT_PEER_STORE = 1,
  MRL_TRANSPORT_PEER_COPY = 2,
  MRL_TRANSPORT_RCCL = 3
};
int mrl_comm_create(mrl_comm **out, const char *name, int32_t nranks, int32_t rank, int32_t device, int32_t transport);
void mrl_comm_destroy(mrl_comm *comm);
const char *mrl_comm_last_error(const mrl_comm *comm); /* comm may be NULL: failure of mrl_comm_create */
int mrl_comm_transport(const mrl_comm *comm);
/* collective; contexts attached to the communicator rebuild their exchange buffers on the next call */
int mrl_comm_set_transport(mrl_comm *comm, int32_t transport);
int mrl_comm_set_timeout(mrl_comm *comm, double seconds); /* bound of every host barrier and device-side wait (default 60 s) */
/* after MRL_ERR_COMM from a device-side wait (a peer's data never arrived) every later wait returns at once; once the caller has
 * destroyed the contexts whose exchanges were in flight (all ranks), this clears the condition so that the communicator can be
 * used again, e.g. with another transport */
int mrl_comm_reset_error(mrl_comm *comm); /* COLLECTIVE: also repairs the host barrier a timed-out rank left behind */
int mrl_comm_barrier(mrl_comm *comm);                     /* host barrier over the ranks */
/* in-place all-reduce of n <= 16 host values: op 0 sum (rank order: identical bits on every rank), 1 min, 2 max */
int mrl_comm_allreduce(mrl_comm *comm, double *h_values, int32_t n, int32_t op);
/* one JSON object describing what the communicator runs on: HIP runtime version and the libamdhip64 actually mapped into the
 * process, whether HIP IPC is usable, the RCCL library loaded (it is looked up beside that libamdhip64 first), its version and the
 * rank count RCCL itself reports for the communicator (-1: not initialised), exchange channels in use */
int mrl_comm_describe(const mrl_comm *comm, char *buf, size_t cap);
/* COLLECTIVE.  The RCCL bring-up alone, stage by stage with a collective verdict after each: load librccl beside the mapped HIP
 * runtime, draw the ncclUniqueId on rank 0 and broadcast it over the bootstrap segment, check the placement (one device per rank --
 * PCI bus ids gathered at creation), ncclCommInitRank, ncclCommCount == nranks.  MRL_OK: RCCL is ready (mrl_comm_set_transport(RCCL)
 * will not initialise anything further); MRL_ERR_UNSUPPORTED: it cannot work on this placement ("unavailable": several ranks share a
 * device, or no library) -- not an error of the job; MRL_ERR_COMM: a stage that should have worked failed.  mrl_comm_describe reports
 * rccl_status, rccl_unique_id_hash (equal on every rank once stage 2 ran), devices_per_rank, distinct_devices, rccl_comm_nranks.
 * Replaces nothing in the reference (its transposes are MPI, DomainAction.C:889-927); this is what lets N rank processes on one GPU
 * test everything up to the call that needs N GPUs. */
int mrl_comm_rccl_preflight(mrl_comm *comm);
/* cumulative count of exchanges posted and payload bytes sent to OTHER ranks by this rank */
int mrl_comm_stats(const mrl_comm *comm, int64_t *n_exchanges, double *bytes_sent);
/* host half of the transport alone, no GPU needed: `rounds` x { barrier, all-gather, sum / min / max all-reduce } over the bootstrap
 * segment "/name", each checked against its closed form; MRL_OK on every rank if all of them agree (CPU test tier) */
int mrl_comm_bootstrap_selftest(const char *name, int32_t nranks, int32_t rank, int32_t rounds);
/* slab contexts only; nranks / rank must match; the context does not own the communicator.  Lifetime: destroy the contexts first
 * (mrl_ctx_destroy tears their exchange pipelines down: collective over the ranks), then the communicator.  If the communicator is
 * destroyed first, mrl_comm_destroy tears the pipelines of every attached context down itself and detaches them; those contexts
 * stay valid for rank-local work and for mrl_ctx_destroy. */
int mrl_ctx_attach_comm(mrl_ctx *ctx, mrl_comm *comm);

/* Local extents: DomainAction::getLocalShape / getReciprocalShape and the partition getters
 * (_local_begin/_local_end, DomainAction.C:524-533).  real_n/recip_n: local extents per axis,
 * real_begin/recip_begin: global index of the first local entry. */
int mrl_local_shape(const mrl_ctx *ctx, int64_t real_n[3], int64_t real_begin[3], int64_t recip_n[3],
                    int64_t recip_begin[3]);
/* Host-only helpers (no GPU needed).
 * mrl_reciprocal_axis: DomainAction.C:268-293, k = 2*pi*fftfreq / rfftfreq, same rounding sequence.
 * mrl_partition:       DomainAction::partitionHepler, include/actions/DomainAction.h:247-280. */
int mrl_reciprocal_axis(int64_t n, double dx, int rfft, double *h_out /* n or n/2+1 */);
int mrl_partition(int64_t total, int32_t nranks, const int64_t *weights, int64_t *h_counts);
/* local reciprocal axis of this context (host copy), axis in [0,dim) */
int mrl_ctx_reciprocal_axis(const mrl_ctx *ctx, int axis, double *h_out, int64_t cap);

/* ---- FFT service: DomainAction::fft / ifft (src/actions/DomainAction.C:833-867, 1049-1078) -- */
/* batch: number of fields; layout 0: field-major [batch][grid], 1: value-major [grid][batch]
 * (the reference's trailing value dimensions, e.g. [n,n,n,3,3]).  Forward is unnormalised,
 * inverse scales by 1/N ("backward" norm); the inverse does not modify d_in. Serial contexts only. */
int mrl_fft_r2c(mrl_ctx *ctx, const double *d_in, double *d_out, int64_t batch, int layout);
int mrl_fft_c2r(mrl_ctx *ctx, const double *d_in, double *d_out, int64_t batch, int layout);

/* Slab-decomposed transform split at its exchange (DomainAction::fftSlab / ifftSlab,
 * DomainAction.C:869-1019).  The caller owns the exchange (RCCL all-to-all, MPI, ...):
 *   forward : mrl_slab_fwd_local -> exchange(send -> recv) -> mrl_slab_fwd_finish
 *   inverse : mrl_slab_inv_local -> exchange(send -> recv) -> mrl_slab_inv_finish
 * send/recv buffers hold nranks chunks, chunk p at the element offset mrl_slab_counts returns for it
 * (complex elements); chunk p of `send` goes to rank p, chunk p of `recv` came from rank p. */
int mrl_slab_counts(const mrl_ctx *ctx, int forward, int64_t *h_send_counts, int64_t *h_recv_counts,
                    int64_t *h_send_offsets, int64_t *h_recv_offsets); /* complex elements, nranks each */
int mrl_slab_fwd_local(mrl_ctx *ctx, const double *d_real_in, double *d_send);
int mrl_slab_fwd_finish(mrl_ctx *ctx, const double *d_recv, double *d_spec_out);
int mrl_slab_inv_local(mrl_ctx *ctx, const double *d_spec_in, double *d_send);
int mrl_slab_inv_finish(mrl_ctx *ctx, const double *d_recv, double *d_real_out);

/* ---- Cahn-Hilliard semi-implicit substep ------------------------------------------------ */
enum mrl_free_energy {
  MRL_FE_DOUBLE_WELL = 0, /* f = A*c^2*(c-1)^2           coef = {A}        (examples/cahn_hilliard/cahnhilliard2.i:74-80) */
  MRL_FE_PFHUB = 1,       /* f = rho*(c-ca)^2*(cb-c)^2   coef = {rho,ca,cb} (benchmarks/01_spinodal_decomposition/1a_solver.i:62-70) */
  MRL_FE_PARSED = 2       /* any ParsedCompute expression: `parsed` = mrl_parsed_create(free energy, inputs = {c},
                             derivatives = {c}); on fast-path shapes the generated chemical potential is compiled INTO the
                             forward z pass (hiprtc), so a user free energy runs at the speed of the built-in ones */
};
struct mrl_parsed;
typedef struct mrl_ch_params {
  int32_t family;  /* enum mrl_free_energy */
  double coef[4];
  double mobility; /* ReciprocalLaplacianFactor factor:        Mbar = -k^2 * M        (ReciprocalLaplacianFactor.C:28-31) */
  double kappa;    /* ReciprocalLaplacianSquareFactor factor:  Lbar = k^2 * k^2 * f   (ReciprocalLaplacianSquareFactor.C:28-32) */
  struct mrl_parsed *parsed; /* MRL_FE_PARSED only: one real input, real result; must outlive the calls that use it */
} mrl_ch_params;

/* ParsedCompute f'(c) (src/tensor_computes/ParsedCompute.C:184-265) for the built-in families */
int mrl_ch_mu(mrl_ctx *ctx, const mrl_ch_params *p, const double *d_c, double *d_mu, int64_t count);

/* Layout of the spectral arrays that belong to the Cahn-Hilliard solver (d_Nhat_new, d_Nhat_old[], d_Nhat_ring[], d_cbar): they are
 * produced and consumed by mrl_ch_substep / mrl_ch_substeps only -- the Nhat history (the reference's Mbarmubar states) and the
 * carried / exported cbar -- and live in a solver-private layout:
 *   element (ix, iy, kz) at ix * plane_pitch + iy * row_pitch + kz           (complex elements)
 * mrl_ch_spec_elems = complex elements to allocate per array, mrl_ch_spec_layout = the two pitches.  Generic shapes: dense
 * (plane_pitch = ny * nzc, row_pitch = nzc).  Serial 3-D contexts on the fused fast path pad the x planes so that the plane pitch is
 * an odd number of 256-byte pieces: the fused x pass gathers 256-byte pieces one plane apart, and the natural pitch of the
 * power-of-two grids puts them on a few memory channels only (256^3: 8 of 128; measured -17 % time for the same bytes with one piece
 * of padding per plane).  Slab contexts: row_pitch = mrl_slab_ch_spec_pitch, plane_pitch = ny * row_pitch.  The padding is never
 * read.  MRL_FLAG_DENSE_SPECTRA at context creation keeps every context dense (mrl_fft_r2c output can then be used as cbar). */
int64_t mrl_ch_spec_elems(const mrl_ctx *ctx);
int mrl_ch_spec_layout(const mrl_ctx *ctx, int64_t *plane_pitch, int64_t *row_pitch);

/* One AdamsBashforthMoulton::substep for the CH system including its root compute group
 * (src/tensor_solver/AdamsBashforthMoulton.C:60-101, src/tensor_computes/ComputeGroup.C:61-84):
 *   mu = f'(c); mubar = fft(mu); Nhat = Mbar*mubar; cbar = fft(c);
 *   ubar = (cbar + (dt*b0)*Nhat + sum_i (dt*b_{i+1})*Nhat_old[i]) / (1 - dt*Lbar);  c_out = ifft(ubar)
 * order = number of history terms used (0 = AB1 ... 4 = AB5), d_Nhat_old[i] = i-th old Mbarmubar.
 * d_Nhat_new is always written (it enters the history); d_cbar / d_mu may be NULL (not materialised).
 * c_out may alias c_in.  Serial contexts only; slab contexts use the mrl_slab_ch_* stages.
 * carry (spectral carry-over, opt-in; see the slab stages below for the rationale):
 *   MRL_CARRY_NONE  the reference's data flow, d_cbar = optional output (cbar of this substep)
 *   MRL_CARRY_OUT   the reference's data flow, ubar additionally written to d_cbar (required)
 *   MRL_CARRY_IN    d_cbar (required) holds cbar = ubar of the previous substep on entry and receives the new ubar; c_in is
 *                   only used for mu = f'(c): one forward transform instead of two (results agree to rounding) */
#define MRL_CARRY_NONE 0
#define MRL_CARRY_OUT 1
#define MRL_CARRY_IN 2
int mrl_ch_substep(mrl_ctx *ctx, const mrl_ch_params *p, const double *d_c_in, double *d_c_out,
                   double *d_Nhat_new, const double *const *d_Nhat_old, int order, double sub_dt,
                   double *d_cbar, double *d_mu, int carry);

/* `count` consecutive substeps: the substep loop of TensorSolver::computeBuffer (src/tensor_solver/TensorSolver.C:93-109) for the
 * Cahn-Hilliard system, i.e. count x { AdamsBashforthMoulton::substep ; advanceState between substeps }.  The reference's
 * examples run 1000 substeps per solver call; inside the loop the real field c of an intermediate substep is not visible to
 * anything (the buffer is rebound every substep, outputs and postprocessors run at the end of the time step), so on planned
 * shapes (the fused family, and z extents with a two-stage plan: 120 ... 320 points) the inverse z pass of substep k and the
 * forward z pass of substep k+1 are one kernel and that field never reaches HBM (bit-identical results; 12 % less traffic at 256^3).
 *   d_Nhat_ring : ring_size >= predictor_order arrays (complex, reciprocal grid).  *head = slot of the newest history entry
 *                 (Nhat_old[0]), *n_old = number of valid history entries.  A substep uses order = min(*n_old, predictor_order-1),
 *                 writes its Nhat into slot (*head + 1) % ring_size, and with advance != 0 that slot becomes the head before
 *                 the next substep (TensorBuffer<T>::advanceState; pass 0 while timeStep() <= 1, TensorProblem.C:455).
 *                 On return the newest Nhat is in slot (*head + 1) % ring_size -- the caller's own advanceState makes it the head.
 *   d_mu        : optional, f'(c) of the last substep's input field (what the mu buffer holds after the call).
 * c_in and c_out must not alias when count > 1 on unplanned shapes.
 *   advance     : bit mask.  MRL_SUBSTEPS_ADVANCE (1): rotate the history between the substeps (see above).
 *                 MRL_SUBSTEPS_DT_CHANGED (2): the time step size differs from the previous step's (_dt != _dt_old): the first
 *                 predictor_order - 1 substeps of the call run at first order whatever the history holds, the history itself keeps
 *                 advancing (AdamsBashforthMoulton.C:75, 88-91: order = min(_substep < _predictor_order && dt_changed ? 0 : n_old, ...)). */
#define MRL_SUBSTEPS_ADVANCE 1
#define MRL_SUBSTEPS_DT_CHANGED 2
int mrl_ch_substeps(mrl_ctx *ctx, const mrl_ch_params *p, const double *d_c_in, double *d_c_out, double *const *d_Nhat_ring,
                    int ring_size, int *head, int *n_old, int predictor_order, int count, int advance, double sub_dt,
                    double *d_mu);

/* fp32 form of mrl_ch_substeps.  The reference selects its floating-point precision per run (src/utils/MarlinUtils.C:39-44,
 * DomainAction.C:81,201), and its only published GPU numbers are fp32 (doc/content/installation.md:36-43).  Same kernels as the fp64
 * path instantiated for float (same butterflies, same pointwise expressions in the same association), on an ordinary context: the
 * shim calls this entry point when the run's float tensor options are float32.  All arrays are float / complex64:
 * d_c_in, d_c_out real [nx][ny][nz]; d_Nhat_ring arrays of mrl_ch_spec_elems_f32(ctx) complex64 values in the solver-private layout
 * mrl_ch_spec_layout_f32 (x planes padded as for fp64).  Scope: serial 3-D contexts with extents in {64, 100, 128, 200, 256, 400, 512},
 * built-in free-energy families, predictor_order <= 3; MRL_ERR_UNSUPPORTED otherwise (mrl_ch_spec_elems_f32 returns 0 then). */
int64_t mrl_ch_spec_elems_f32(const mrl_ctx *ctx);
int mrl_ch_spec_layout_f32(const mrl_ctx *ctx, int64_t *plane_pitch, int64_t *row_pitch);
int mrl_ch_substeps_f32(mrl_ctx *ctx, const mrl_ch_params *p, const float *d_c_in, float *d_c_out, float *const *d_Nhat_ring,
                        int ring_size, int *head, int *n_old, int predictor_order, int count, int advance, double sub_dt);

/* ReciprocalLaplacianFactor (power = 1: -k^2 * factor, ReciprocalLaplacianFactor.C:28-31) and
 * ReciprocalLaplacianSquareFactor (power = 2: k^2 * k^2 * factor, ReciprocalLaplacianSquareFactor.C:28-32) as real
 * arrays on the local reciprocal grid, for solvers that take their linear operator as a buffer. */
int mrl_reciprocal_laplacian(mrl_ctx *ctx, int power, double factor, double *d_out);

/* Generic k-space ABM update for caller-supplied reciprocal arrays (AdamsBashforthMoulton.C:94-99):
 *   ubar = (ubar0 + sum_i coef[i]*N[i]) / (1 - dt*L)   (L may be NULL: no division)
 * N[i] complex arrays of n_spec elements, L real array. */
int mrl_kspace_abm(mrl_ctx *ctx, double *d_ubar_out, const double *d_ubar0, const double *const *d_N,
                   const double *h_coef, int nterms, const double *d_L, double dt, int64_t n_spec);

/* Coupled k-space update of AdamsBashforthMoultonCoupled::substep (AdamsBashforthMoultonCoupled.C:118-186, corrector
 * :214-270): for nvar <= 32 variables (the reference solves any N; its inputs couple 2 or 3), at every reciprocal grid point
 *   rhs_i = ubar0_i + sum_t coef[i][t] * N[i][t]          (h_nterms[i] <= 6 terms; d_N / h_coef are the rows concatenated)
 *   solve (I - dt * Lhat) ubar = rhs                       (dense nvar x nvar, LU with partial pivoting, one thread per k)
 * d_L[i*nvar + j] = real array of the linear operator entry the input file names (row i, column j), NULL = zero.
 * Default flags (0) reproduce the reference bit for bit in structure, including two of its quirks that its gold files
 * (test/tests/solvers/gold/coupled_*.csv) pin: the matrix is assembled transposed (A_ab = delta_ab - dt*L_ba, :160-178)
 * and the complex right-hand side is cast to the real dtype of L, dropping Im(rhs) (:183; outputs then have Im = 0).
 * MRL_COUPLED_L_AS_WRITTEN uses A_ab = delta_ab - dt*L_ab; MRL_COUPLED_COMPLEX_RHS solves for the full complex rhs.
 * Up to 8 variables the matrix of a k-point lives in registers; 9 ... 32 (or MRL_COUPLED_GENERAL at any size) run the same
 * elimination, operation for operation, on a device workspace of (nvar^2 + 2 nvar) x 65536 doubles owned by the context. */
#define MRL_COUPLED_L_AS_WRITTEN 1
#define MRL_COUPLED_COMPLEX_RHS 2
#define MRL_COUPLED_GENERAL 4
int mrl_kspace_coupled(mrl_ctx *ctx, int nvar, double *const *d_ubar_out, const double *const *d_ubar0,
                       const double *const *d_N, const double *h_coef, const int *h_nterms, const double *const *d_L,
                       double dt, int flags, int64_t n_spec);

/* SecantSolver::substep (src/tensor_solver/SecantSolver.C:60-176), the reciprocal-space work for one variable; complex arrays
 * of n_spec elements, d_L real or NULL.  The caller (the solver object) keeps the reference's control flow: evaluate the
 * compute group, call these, inverse-transform the new iterate into the variable's buffer, test convergence.
 *   mrl_secant_begin  (:79-101): R0 = (N + L*u)*sub_dt ; guess = (u + dt_epsilon*N)/(1 - dt_epsilon*L) ; h_sumsq[0] = sum|R0|^2
 *   mrl_secant_iterate (:121-140): R = (N + L*u)*sub_dt + u_old - u ; du = where(R - R_prev != 0, -R*(u - u_prev)/(R - R_prev), 0) ;
 *        u_new = u + du*damping ; R_prev <- R in place ; h_sumsq = { sum|R|^2, sum|du|^2 }   (complex division as
 *        c10::complex does it).  The previous iterate is simply the caller's previous `u` buffer (`uprev[i] = u` is a handle
 *        copy in the reference too).  Sums are over this rank's reciprocal grid; torch::norm = sqrt of them.
 * Both calls synchronise the context's stream to return the sums (the reference's .item() calls). */
int mrl_secant_begin(mrl_ctx *ctx, const double *d_u, const double *d_N, const double *d_L, double sub_dt, double dt_epsilon,
                     double *d_R0_out, double *d_guess_out, double *h_sumsq, int64_t n_spec);
int mrl_secant_iterate(mrl_ctx *ctx, const double *d_u, const double *d_N, const double *d_L, const double *d_u_old,
                       const double *d_u_prev, double *d_R_prev, double sub_dt, double damping, double *d_u_new,
                       double *h_sumsq, int64_t n_spec);

/* BroydenSolver::substep (src/tensor_solver/BroydenSolver.C:63-176) for nvar <= 32 coupled variables: per reciprocal grid point
 * a Broyden iteration on R(u) = (N + L u) sub_dt + u_old - u with a persistent complex nvar x nvar approximation M of the inverse
 * Jacobian.  The caller keeps the reference's control flow (compute group, inverse transforms, convergence tests); per iteration:
 *   mrl_broyden_predict: S = -M R ; u_out_i = u_i + step * S_i                    (:124-133; the reference hard-wires step = 0.5)
 *   [u_i = ifft(u_out_i); compute group]
 *   mrl_broyden_update : Rnew = (N + L u) sub_dt + u_old - u ; y = Rnew - R ; d = S^T y ;
 *                        M += where(|d| > 1e-12, (S - M y) S^T / d, 0) ; R <- Rnew ; h_sumsq = sum |Rnew|^2      (:139-166)
 * mrl_broyden_residual fills R before the first iteration (d_u_old = NULL: (N + L u) sub_dt, :99) and mrl_broyden_init sets
 * M = factor * I (:57-63; M persists over substeps).  Arrays owned by the caller, field-major: d_M [nvar*nvar][n_spec],
 * d_R / d_S [nvar][n_spec] complex.  No regression test of the reference exercises this solver: parity is against the
 * oracle's restatement only. */
int mrl_broyden_init(mrl_ctx *ctx, int nvar, double factor, double *d_M, int64_t n_spec);
int mrl_broyden_residual(mrl_ctx *ctx, int nvar, const double *const *d_u, const double *const *d_N, const double *const *d_L,
                         const double *const *d_u_old, double sub_dt, double *d_R, double *h_sumsq, int64_t n_spec);
int mrl_broyden_predict(mrl_ctx *ctx, int nvar, const double *d_M, const double *d_R, const double *const *d_u, double step,
                        double *d_S, double *const *d_u_out, int64_t n_spec);
int mrl_broyden_update(mrl_ctx *ctx, int nvar, double *d_M, double *d_R, const double *d_S, const double *const *d_u,
                       const double *const *d_N, const double *const *d_L, const double *const *d_u_old, double sub_dt,
                       double *h_sumsq, int64_t n_spec);

/* Slab (multi-GPU) CH substep (AdamsBashforthMoulton::substep over DomainAction::fftSlab/ifftSlab), split at its
 * exchanges and pipelined over `nsub` sub-blocks of the kz axis: after the z pass every kz plane is an independent
 * 2-D problem, so the caller can put sub-block s on the wire while sub-block s+1 is being transformed.
 *   mrl_slab_ch_z_fwd                : mu = f'(c); z pass of c and mu (kept in context scratch)
 *   for s: mrl_slab_ch_x_fwd(s)      : forward x pass of both fields for kz in K_s       -> d_send
 *          [exchange s, forward]       one message per peer carrying both fields
 *   for s: mrl_slab_ch_kspace(s)     : forward y pass, Nhat / ubar update, inverse y pass -> d_send
 *          [exchange s, inverse]
 *   for s: mrl_slab_ch_x_inv(s)      : inverse x pass (into context scratch)
 *   mrl_slab_ch_z_inv                : inverse z pass, 1/N                                -> c_out
 * K_s = kz range s of mrl_partition(nzc, nsub).  Buffer layouts (complex elements, peer chunks back to back in rank
 * order, sizes from mrl_slab_ch_counts; x_p / y_p = the x / y range rank p owns, "me" = this rank):
 *   forward send  [p][field c, mu][x_p ][y_me][K_s]      forward recv  [p][field c, mu][x_me][y_p ][K_s]
 *   inverse send  [p][x_me][y_p ][K_s]                   inverse recv  [p][x_p ][y_me][K_s]
 * d_Nhat_new / d_Nhat_old / d_cbar are the dense reciprocal arrays [x_me][ny][nzc] of the reference.
 * The last index of the exchange layouts has the pitch mrl_slab_ch_k_pitch(ctx, s, nsub) >= |K_s| (planned shapes pad the rows to
 * 128-byte lines), and on planned shapes the x planes of a chunk lie an ODD number of 256-byte pieces apart (>= y extent x pitch:
 * the x passes gather / scatter 256-byte pieces one plane apart, and the natural pitch of the power-of-two grids puts them on a few
 * memory channels only); the padding is never read.  The buffers are opaque to the caller: mrl_slab_ch_counts returns the message
 * sizes that follow from the layout, chunks back to back in rank order.
 *
 * Spectral carry-over (`carry`).  The reference recomputes cbar = fft(c) in every substep although c = ifft(ubar) of
 * the previous one; fft(ifft(.)) is the identity up to rounding (1e-16 relative), so a rank can keep ubar -- it is produced
 * on that rank, in reciprocal layout -- and use it as the next cbar.  The forward stages then move ONE field (mu) instead of
 * two and a substep needs 2 slab transposes instead of 3, which is what bounds the multi-GPU rate.
 *   MRL_CARRY_NONE  the reference's data flow; d_cbar = optional output (cbar of this substep)
 *   MRL_CARRY_OUT   the reference's data flow, and ubar is additionally written to d_cbar (required): the bootstrap substep,
 *                   also to be used again whenever c was modified by anything but this solver
 *   MRL_CARRY_IN    d_cbar (required) holds cbar on entry and receives ubar; z_fwd / x_fwd / the forward exchange carry
 *                   mu only: forward layouts lose the field index, [p][x_p][y_me][K_s] / [p][x_me][y_p][K_s]
 * The same value has to be passed to the four calls of one substep.  Results differ from MRL_CARRY_NONE at rounding level
 * only (tests: <= 1e-13 after 20 substeps, and against the reference's gold file).  (MRL_CARRY_* are defined above.) */
/* Last-axis pitch, in complex elements, of the rank-local spectral arrays of the slab CH stages (d_Nhat_new, d_Nhat_old[], d_cbar):
 * [x_me][ny][pitch], element (ix, j, kz) at (ix*ny + j)*pitch + kz.  Equal to the reciprocal extent nz/2+1 on generic shapes; the
 * planned pipeline pads rows to a multiple of 8 elements so that every row starts on a 128-byte line (its y pass gathers 128-byte
 * row segments: with the odd natural pitch each of them straddles two lines and the pass moves 1.5x its algorithmic bytes, measured
 * with the FETCH_SIZE / WRITE_SIZE counters).  Allocate x_me*ny*pitch complex values per array; the padding is never read. */
int64_t mrl_slab_ch_spec_pitch(const mrl_ctx *ctx);
int64_t mrl_slab_ch_k_pitch(const mrl_ctx *ctx, int sub, int nsub);
int mrl_slab_ch_counts(const mrl_ctx *ctx, int sub, int nsub, int forward, int carry, int64_t *h_send_counts,
                       int64_t *h_recv_counts);
int mrl_slab_ch_z_fwd(mrl_ctx *ctx, const mrl_ch_params *p, const double *d_c_in, double *d_mu /* optional out */, int carry);
int mrl_slab_ch_x_fwd(mrl_ctx *ctx, int sub, int nsub, double *d_send, int carry);
int mrl_slab_ch_kspace(mrl_ctx *ctx, const mrl_ch_params *p, int sub, int nsub, const double *d_recv, double *d_send,
                       double *d_Nhat_new, const double *const *d_Nhat_old, int order, double sub_dt, double *d_cbar, int carry);
int mrl_slab_ch_x_inv(mrl_ctx *ctx, int sub, int nsub, const double *d_recv);
/* between two substeps of one solver call: mrl_slab_ch_z_inv of substep k fused with mrl_slab_ch_z_fwd of substep k + 1 (both
 * are local to the rank; the intermediate real field is not written; `carry` = the mode of substep k + 1) */
int mrl_slab_ch_z_inv_fwd(mrl_ctx *ctx, const mrl_ch_params *p, double *d_mu /* optional out */, int carry);
int mrl_slab_ch_z_inv(mrl_ctx *ctx, double *d_c_out);

/* ---- de Geus mechanics ------------------------------------------------------------------ */
/* Fields are value-major as in the reference: rank-2 [grid][3][3] (dim x dim in 2-D). */
/* G(A) = ifft( Ghat4 : fft(A) )   (FFTMechanics.C:74-84, 105-106) without materialising Ghat4 */
int mrl_gamma_apply(mrl_ctx *ctx, const double *d_A, double *d_out);
/* HyperElasticIsotropic::computeBuffer (HyperElasticIsotropic.C:42-52): P(F) ; K4 is not stored */
int mrl_mech_stress(mrl_ctx *ctx, const double *d_F, const double *d_K, const double *d_mu, double *d_P);
/* K_dF(dF) = trans2(ddot42(K4, trans2(dF)))  (FFTMechanics.C:107-108), K4 rebuilt on the fly from (F,K,mu) */
int mrl_mech_tangent_apply(mrl_ctx *ctx, const double *d_F, const double *d_K, const double *d_mu,
                           const double *d_dF, double *d_out);

/* Building blocks of the same solve on a SLAB context (BASELINE configs[4]); the Newton-CG driver, the exchanges and
 * the all-reduce of the CG scalars belong to the caller (marlin_amd/slab.py: SlabMechanics):
 *   G(A):  mrl_relayout(to field-major) -> per component mrl_slab_fwd_local / exchange / mrl_slab_fwd_finish
 *          -> mrl_slab_gamma_project -> per component mrl_slab_inv_local / exchange / mrl_slab_inv_finish -> mrl_relayout
 *   mrl_mech_stress / mrl_mech_tangent_apply are pointwise and work on the local real slab of any context;
 *   mrl_dot / mrl_norm2 / mrl_sum return the LOCAL value (the caller all-reduces), mrl_axpby is the vector update. */
/* in-place Gamma projection of field-major spectra [D*D][x_me][ny][nzc] (complex), times scale */
int mrl_slab_gamma_project(mrl_ctx *ctx, double *d_spec, double scale);
/* Fused slab Gamma operator on FIELD-MAJOR data [9][x][y_me][z] (3-D, planned extents, equal power-of-two partitions;
 * MRL_ERR_UNSUPPORTED otherwise -- use the per-component stages above).  out_ij = q_j (sum_k A_ik q_k)/|q|^2 couples only
 * the three components of one tensor row, so the rows are three independent pipelines (row r+1 is transformed while row r
 * is on the wire).  Per row r = 0..2:
 *   mrl_slab_gamma_row_fwd(r, A_fm, send)   z + x passes of fields 3r..3r+2           -> send  [p][3][x_p ][y_me][nzc]
 *   [exchange]                                                                          -> recv  [p][3][x_me][y_p ][nzc]
 *   mrl_slab_gamma_row_mid(recv, scale)     y pass, projection * scale, inverse y pass, IN PLACE (same chunked layout)
 *   [exchange, inverse: send = that buffer]                                             -> recv2 [p][3][x_p ][y_me][nzc]
 *   mrl_slab_gamma_row_inv(r, recv2, out_fm) inverse x + z passes, 1/N                -> fields 3r..3r+2 of out_fm
 * mrl_slab_gamma_counts: complex elements per peer of one row's message (forward / inverse direction). */
int mrl_slab_fast_path(const mrl_ctx *ctx); /* 1 if the fused slab mechanics kernels (mrl_slab_gamma_row_*, 32-bit byte offsets per row buffer) apply to this context */
int mrl_slab_gamma_counts(const mrl_ctx *ctx, int forward, int64_t *h_send_counts, int64_t *h_recv_counts);
int mrl_slab_gamma_row_fwd(mrl_ctx *ctx, int row, const double *d_A_fm, double *d_send);
int mrl_slab_gamma_row_mid(mrl_ctx *ctx, double *d_recv_inout, double scale);
int mrl_slab_gamma_row_inv(mrl_ctx *ctx, int row, const double *d_recv, double *d_out_fm,
                           const double *d_dotv_fm /* optional [9][npts]: accumulate sum(out * dotv), rows in order 0,1,2 */);
/* the local sum(out * dotv) over the three rows just inverted with d_dotv_fm (the p.Ap of the CG, taken while Ap is in
 * registers); synchronises the stream */
int mrl_slab_gamma_dot(mrl_ctx *ctx, double *h_local);
/* CG vector kernels for callers that own the iteration (slab contexts all-reduce the scalars between the calls):
 *   mrl_cg_update:           x += alpha p ; r -= alpha Ap ; *h_rr_local = sum r^2 over the local vector   (MarlinUtils.h:95-100)
 *   mrl_mech_tangent_dir_fm: p <- r + beta p ; out = K_dF(p)   (MarlinUtils.h:112 fused with FFTMechanics.C:107-108; field-major) */
/* The CG direction update and the tangent fused into the forward z pass of the row pipelines (z lines of 32 ... 256 points;
 * mrl_slab_gamma_tangent_fusable tells):  [d_x += alpha_prev * p, the previous iteration's solution update, if d_x != NULL;]
 * p <- r + beta p ; the z spectra of K_dF(p) for all nine fields stay in context scratch, and the following
 * mrl_slab_gamma_row_fwd(r, NULL, send) calls run only the x pass from there.  K_dF(p) itself is never written.
 * mrl_cg_update_r is mrl_cg_update without the x update (r -= alpha Ap ; sum r^2), for callers that defer it this way. */
int mrl_slab_gamma_tangent_fusable(const mrl_ctx *ctx);
int mrl_slab_gamma_tangent_z_fwd(mrl_ctx *ctx, const double *d_F, const double *d_K, const double *d_mu, double *d_p,
                                 const double *d_r, double beta, double *d_x /* optional */, double alpha_prev);
int mrl_cg_update_r(mrl_ctx *ctx, double alpha, double *d_r, const double *d_Ap, int64_t n, double *h_rr_local);
int mrl_cg_update(mrl_ctx *ctx, double alpha, double *d_x, double *d_r, const double *d_p, const double *d_Ap, int64_t n,
                  double *h_rr_local);
int mrl_mech_tangent_dir_fm(mrl_ctx *ctx, const double *d_F_fm, const double *d_K, const double *d_mu, double *d_p_fm,
                            const double *d_r_fm, double beta, double *d_out_fm);
/* mrl_mech_stress / mrl_mech_tangent_apply on field-major fields [9][npts] (3-D; K, mu are [npts]); npts must be even */
int mrl_mech_stress_fm(mrl_ctx *ctx, const double *d_F_fm, const double *d_K, const double *d_mu, double *d_P_fm);
int mrl_mech_tangent_apply_fm(mrl_ctx *ctx, const double *d_F_fm, const double *d_K, const double *d_mu, const double *d_dF_fm,
                              double *d_out_fm);
/* value-major [npts][ncomp] <-> field-major [ncomp][npts] */
int mrl_relayout(mrl_ctx *ctx, int to_field_major, const double *d_in, double *d_out, int64_t npts, int32_t ncomp);
/* out = a*x + b*y (out may alias x or y) */
int mrl_axpby(mrl_ctx *ctx, double a, const double *d_x, double b, const double *d_y, double *d_out, int64_t n);
/* y += a*x: the update of conjugateGradientSolve (x = x + alpha p, r = r - alpha Ap; src/utils/ConjugateGradientSolver.h) under the
 * name the survey's boundary table uses; = mrl_axpby(a, x, 1, y, y) */
int mrl_axpy(mrl_ctx *ctx, double a, const double *d_x, double *d_y, int64_t n);

typedef struct mrl_mech_params {
  double l_tol;       /* FFTMechanics l_tol */
  int64_t l_max_its;  /* 0 = number of cells (FFTMechanics.C:63-64) */
  double nl_rel_tol, nl_abs_tol;
  int32_t nl_max_its;
} mrl_mech_params;
typedef struct mrl_mech_stats {
  int32_t newton_its;
  int32_t cg_its_total;
  int32_t cg_its[64];  /* per Newton iteration (first 64) */
  double last_anorm, last_rnorm, Fn;
} mrl_mech_stats;
/* FFTMechanics::computeBuffer (FFTMechanics.C:96-163) with conjugateGradientSolve
 * (include/utils/MarlinUtils.h:55-131).  d_applied: [3][3] (dim x dim) device array or NULL.
 * Writes Fnew and the final stress P.  Synchronous (host reads CG scalars). */
int mrl_mech_newton_cg(mrl_ctx *ctx, const mrl_mech_params *p, const double *d_F, const double *d_K,
                       const double *d_mu, const double *d_applied, double *d_Fnew, double *d_P,
                       mrl_mech_stats *stats);
/* Small-strain linear-elastic RVE, the wording of BASELINE configs[2] (de Geus's Gamma-operator scheme with a constant tangent).  The
 * reference itself has only the finite-strain solve above; this is that solve's FIRST linear system taken at F = I, where the second
 * Piola-Kirchhoff stress vanishes and the tangent of HyperElasticIsotropic.C:42-52 reduces to C4 = K II + 2 mu (I4s - II/3):
 *   CG (MarlinUtils.h:55-123, l_tol / l_max_its of p) on  G(C4 : d_eps) = -G(C4 : E),   eps = E + d_eps,   sigma = C4 : eps
 * with the same kernels (tangent at the identity field, Gamma operator, fused CG vectors).  d_E: [dim][dim] device array (the applied
 * macroscopic strain); d_eps, d_sigma: [grid...][dim][dim].  stats: newton_its = 1, cg_its[0].  Serial and slab contexts. */
int mrl_mech_small_strain(mrl_ctx *ctx, const mrl_mech_params *p, const double *d_K, const double *d_mu, const double *d_E,
                          double *d_eps, double *d_sigma, mrl_mech_stats *stats);

/* ComputeDisplacements::computeBuffer (src/tensor_computes/ComputeDisplacements.C:53-107): displacement field of a deformation
 * gradient F [grid..., D, D]:  u = (<F> - I) X + ifft( fft(F - <F>) . (-i q) / |q|^2 )  (zero at q = 0), linearly interpolated
 * (torch interpolate, align_corners = true) from the n cell centres to n + 1 points per axis.
 * d_disp: [(n_0+1)...(n_{D-1}+1)][D] doubles, value-major.  Serial contexts. */
int mrl_mech_displacements(mrl_ctx *ctx, const double *d_F, double *d_disp);

/* ComputeVonMisesStress::computeBuffer (src/tensor_computes/ComputeVonMisesStress.C:31-66): von Mises measure of a
 * rank-two field [grid..., D, D] -> [grid...] with the reference's 2-D / 3-D formulas and term order. */
int mrl_mech_von_mises(mrl_ctx *ctx, const double *d_stress, double *d_out);

/* Homogeneous small-strain elasticity coupled to a concentration field by the volumetric eigenstrain e0*c (3-D, serial,
 * half-spectrum contexts; test/tests/tensor_compute/coupled_pf_mech.i).
 *   mrl_qs_elasticity               FFTQuasistaticElasticity::computeBuffer (src/tensor_computes/FFTQuasistaticElasticity.C:46-104):
 *                                   per k-point A u-hat = b with A_ij from (mu, lambda) and k = 2 pi i * reciprocal axis,
 *                                   b = k * 2 e0 (3 lambda + mu) c-hat, A_ii = 1 and b = 0 at k = 0; d_disp[0..2] (real
 *                                   [nx][ny][nz]) receive the inverse transforms.  d_cbar: half spectrum of c (mrl_fft_r2c).
 *   mrl_elastic_chemical_potential  FFTElasticChemicalPotential::computeBuffer (src/tensor_computes/FFTElasticChemicalPotential.C:47-61):
 *                                   d_out (half spectrum) = -e0 (e0 (9 lambda + 6 mu) c-hat - (2 mu + 3 lambda) k . u-hat). */
int mrl_qs_elasticity(mrl_ctx *ctx, const double *d_cbar, double mu, double lambda, double e0, double *const *d_disp);
int mrl_elastic_chemical_potential(mrl_ctx *ctx, const double *d_cbar, const double *const *d_disp, double mu, double lambda,
                                   double e0, double *d_out);

/* ---- parsed pointwise expressions: ParsedCompute (src/tensor_computes/ParsedCompute.C:50-265) ----------------
 * expression text -> AST -> d/d(derivatives[0]) d/d(derivatives[1]) ... -> simplify -> one fused HIP kernel (hiprtc).
 * Grammar, derivative and simplification rules follow the reference's parser (they fix the floating-point
 * evaluation order); see marlin_amd/csrc/expr.hip.  Inputs are full-size device arrays, real or interleaved complex;
 * named constants stay symbolic; with extra_symbols the names x y z kx ky kz k2 t pi e i are available and the
 * expression is evaluated on the whole real (space = 0) or reciprocal (space = 1) grid of the context.
 * ctx may be NULL: the expression is parsed / differentiated / simplified only (mrl_parsed_string), no GPU needed. */
typedef struct mrl_parsed mrl_parsed;
int mrl_parsed_create(mrl_ctx *ctx, mrl_parsed **out, const char *expression, int n_inputs, const char *const *input_names,
                      const int *input_is_complex, int n_constants, const char *const *constant_names,
                      const double *constant_values, int n_derivatives, const char *const *derivatives, int extra_symbols,
                      int space);
void mrl_parsed_destroy(mrl_parsed *p);
int mrl_parsed_is_complex(const mrl_parsed *p);      /* 1: the result is complex */
const char *mrl_parsed_string(const mrl_parsed *p);  /* the simplified tree, fully parenthesised */
const char *mrl_parsed_source(const mrl_parsed *p);  /* the generated HIP source */
int mrl_parsed_eval(mrl_parsed *p, const double *const *d_inputs, double *d_out, int64_t count, double time);

/* ---- reductions: torch::sum / torch::norm call sites of the CG (MarlinUtils.h:63,82,92,99,109) */
/* Synchronous: result returned in *h_out. */
int mrl_dot(mrl_ctx *ctx, const double *d_a, const double *d_b, int64_t n, double *h_out);
int mrl_norm2(mrl_ctx *ctx, const double *d_a, int64_t n, double *h_out);
int mrl_sum(mrl_ctx *ctx, const double *d_a, int64_t n, double *h_out);
/* TensorExtremeValuePostprocessor (src/postprocessors/TensorExtremeValuePostprocessor.C:30-44) */
int mrl_minmax(mrl_ctx *ctx, const double *d_a, int64_t n, double *h_min, double *h_max);
/* DomainAction::average over the grid of a value-major field [grid][ncomp] -> h_out[ncomp]; on a slab context the
 * local sum divided by the GLOBAL point count (the sum over ranks is the average).
 * With a communicator attached to a slab context, mrl_dot / mrl_norm2 / mrl_sum / mrl_minmax / mrl_average return GLOBAL values
 * (any ncomp); mrl_histogram stays rank-local (the reference gathers its vector postprocessors itself). */
int mrl_average(mrl_ctx *ctx, const double *d_a, int64_t ncomp, double *h_out);
/* TensorHistogram (src/vectorpostprocessors/TensorHistogram.C:48-79, at::native::histogramdd with explicit edges): counts of the
 * values of d_a per bin [edge_i, edge_i+1) -- the last bin closed on the right, values outside the edges ignored.  h_edges has
 * nbins + 1 non-decreasing entries (the reference uses torch::linspace(min, max, bins + 1)).  Local count; synchronises. */
int mrl_histogram(mrl_ctx *ctx, const double *d_a, int64_t n, const double *h_edges, int nbins, int64_t *h_counts);

/* ---- timing on the context stream (hipEvents): used by bench.py for roofline.achieved ---- */
int mrl_timer_start(mrl_ctx *ctx);
int mrl_timer_stop(mrl_ctx *ctx, float *h_ms); /* records, synchronises, returns elapsed ms */
/* per-kernel-class accumulated device time of the calls made between start/stop when
 * profiling is enabled via mrl_set_profiling(ctx,1) (adds an event pair per launch). */
int mrl_set_profiling(mrl_ctx *ctx, int on);
int mrl_get_profile(mrl_ctx *ctx, int slot, const char **name, double *total_ms, int64_t *launches,
                    double *bytes_per_launch /* algorithmic HBM bytes of one launch */);
/* the whole profile in one struct (the survey's mrl_get_timing): device time, launches and algorithmic bytes summed over the kernel
 * classes recorded since profiling was switched on, and the class with the largest share */
typedef struct mrl_timing {
  int32_t kernel_classes;      /* profile slots with at least one launch */
  int64_t launches;
  double device_ms;            /* sum of the per-launch HIP-event times */
  double algorithmic_bytes;    /* sum over launches of the bytes their kernels must move */
  const char *dominant;        /* name of the class with the largest device time (owned by the context), NULL when empty */
  double dominant_ms;
} mrl_timing;
int mrl_get_timing(mrl_ctx *ctx, mrl_timing *out);

/* ---- HDF5 container of XDMFTensorOutput (src/tensor_outputs/XDMFTensorOutput.C with enable_hdf5 = true) -- host code, no GPU ----
 * The reference stores every output component as a dataset "<name>.<frame>" in the root group of "<file_base>[.rankNNNN].h5"
 * through libhdf5: H5Fcreate (XDMFTensorOutput.C:152-160), addDataToHDF5 = H5Screate_simple + H5Dcreate + H5Dwrite
 * (:578-650, called from :323-343), H5Fflush per output step (:244-246), H5Fclose (:113-115).  The image has no libhdf5, so
 * these four entry points write the same container directly from the HDF5 file format specification (h5write.hip); the files
 * are ordinary HDF5 (h5dump / h5py / the reference's HDF5Diff tester read them).  dims are the dataset's dimensions, slowest
 * first, as H5Screate_simple takes them; host_data is dense row-major.  A dataset name that already exists is an error, as
 * in the reference (:593-594). */
typedef struct mrl_h5 mrl_h5;
enum { MRL_H5_F64 = 0, MRL_H5_F32 = 1, MRL_H5_I32 = 2, MRL_H5_I64 = 3 };   /* the types XDMFTensorOutput.C:331-341 accepts */
int mrl_h5_create(const char *path, mrl_h5 **out);   /* truncates; the empty file is already a valid HDF5 file */
int mrl_h5_write(mrl_h5 *file, const char *name, int dtype, int rank, const int64_t *dims, const void *host_data);
int mrl_h5_flush(mrl_h5 *file);                      /* the file on disk is valid and complete up to the last dataset */
int mrl_h5_close(mrl_h5 *file);
const char *mrl_h5_last_error(const mrl_h5 *file);

#ifdef __cplusplus
}
#endif
#endif /* MARLIN_HIP_H */
