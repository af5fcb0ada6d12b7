// Internal interface of the multi-GPU transport (comm.hip): one process per GPU on one node.
//
// The global transposes of DomainAction::fftSlab / ifftSlab (src/actions/DomainAction.C:869-1019) are host-staged MPI
// point-to-point messages in the reference.  Here every rank owns SYMMETRIC receive buffers that all peers map through HIP IPC,
// and an exchange is one of
//   PEER_STORE  the producing kernel stores its output chunks straight into the peers' receive buffers over xGMI (pointer
//               table per destination rank), followed by a flag write; no send buffer, no copy
//   PEER_COPY   the producing kernel writes a local send buffer; the chunks are pushed by the copy engines (hipMemcpyAsync to the
//               IPC-mapped peer buffers on side streams), followed by a flag write
//   RCCL        grouped ncclSend / ncclRecv on a side stream (librccl is loaded at run time)
// Arrival is a per-source 64-bit epoch flag in the receiver's memory; the consumer kernel is ordered behind a small wait
// kernel with a bounded spin (a lost peer ends in an error code, never in a hung GPU).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "comm_dev.h"
#include "marlin_hip.h"

namespace mrl {

constexpr int kMaxRanks = 64;
constexpr int kMaxChannels = 256;
constexpr int kFlagRow = kMaxRanks;  // 64-bit words per channel row

struct ShmSeg;   // host bootstrap segment (POSIX shared memory)
struct RcclApi;  // entry points of librccl resolved at run time

struct SymBuf {  // a device allocation of the same size on every rank, mapped by all peers
  void *local = nullptr;
  size_t bytes = 0;
  std::vector<void *> peer;  // peer[p] = address of rank p's allocation in THIS process (peer[rank] = local)
  std::vector<char> ipc_mapped;  // peer[p] came from hipIpcOpenMemHandle (to be closed); 0: rank p lives in this process (a thread), peer[p] is its pointer
};

// One exchange endpoint: a symmetric receive buffer, an optional local send buffer, one arrival channel.
struct Xchg {
  SymBuf recv;
  double *send = nullptr;  // local staging (PEER_COPY / RCCL, or producers that cannot scatter through a table)
  size_t send_bytes = 0;
  int channel = -1;
  uint64_t epoch = 0;
  std::vector<size_t> send_off, send_cnt;  // bytes: chunk for peer p in the send buffer
  std::vector<size_t> recv_off, recv_cnt;  // bytes: chunk from peer p in MY receive buffer
  std::vector<size_t> slot_at_peer;        // bytes: where my chunk lands in peer p's receive buffer
  unsigned int *d_counter = nullptr;       // device word: workgroups of the producing launch that have finished (signal_tail)
  char **d_tab = nullptr;                  // device table [nranks]: destination of the chunk for peer p (see tab_direct)
  bool tab_direct = false;                 // d_tab points into the peers' receive buffers (PEER_STORE) or into `send`
  std::vector<hipEvent_t> copy_done;       // PEER_COPY: per side stream, the last push out of `send`
  hipEvent_t rccl_done = nullptr;
  hipEvent_t release_ev = nullptr;         // PEER_STORE: system-scope release between the producing kernel and the flag kernel
  bool pending_send_guard = false;
};

}  // namespace mrl

struct mrl_comm {
  int nranks = 1, rank = 0, device = 0;
  int transport = MRL_TRANSPORT_PEER_STORE;
  bool ipc_ok = true;
  // PEER_STORE: the producing kernels raise the arrival flags themselves (every workgroup fences at system scope and counts itself,
  // comm_dev.h) instead of an event-ordered flag kernel after them.  Saves a launch per exchange but costs one L2 write-back per
  // workgroup -- measured 2x slower on one GPU, so off by default (MRL_OPT_EXPERIMENT bit 128 of an attached context turns it on)
  bool kernel_signals = false;
  double timeout_s = 60.0;
  mrl::ShmSeg *shm = nullptr;
  std::string shm_name;
  uint32_t red_parity = 0;
  // flags: [kMaxChannels][kFlagRow] 64-bit epochs, symmetric
  mrl::SymBuf flags;
  unsigned long long **d_flag_tab = nullptr;  // device table: peer flag bases
  int *h_status = nullptr;          // pinned: 0 ok, else 1 + rank whose flag timed out
  int *d_status = nullptr;
  long long wall_khz = 100000;
  int next_channel = 0;
  // channels returned by destroyed exchanges (every rank creates and destroys exchanges in the same order, so the lists agree);
  // a reused channel continues from the epoch its previous owner reached: flags left in the row can never satisfy a new wait
  std::vector<int> free_channels;
  std::vector<uint64_t> chan_epoch = std::vector<uint64_t>(mrl::kMaxChannels, 0);
  // contexts whose exchange pipelines live on this communicator (mrl_ctx_attach_comm): torn down by mrl_comm_destroy if the
  // caller destroys the communicator first
  std::vector<mrl_ctx *> attached;
  // MRL_OPT_VERIFY_EXCHANGE: mismatches between plain and system-scope re-reads of the receive buffers (device counter)
  unsigned long long *d_verify = nullptr;
  std::vector<hipStream_t> side;    // side streams (copy engines / RCCL)
  hipEvent_t ev_prod = nullptr;
  // device mailbox for scalar all-reduces: [2][kMaxRanks][16] doubles, symmetric
  mrl::SymBuf mbox;
  double **d_mbox_tab = nullptr;
  int mbox_channel = -1;
  uint64_t mbox_epoch = 0;
  double *h_mbox = nullptr;  // pinned result [16]
  double *d_h_mbox = nullptr;
  // which physical GPU every rank runs on (PCI bus ids gathered at creation): RCCL needs one device per rank, HIP IPC does not
  std::vector<std::string> pci;        // [nranks]
  int distinct_devices = 1;
  // RCCL bring-up, stage by stage (rccl_init): "not tried" | "ready" | "unavailable: ..." (cannot work on this placement: several
  // ranks on one device, no library) | "failed: ..." (it should have worked)
  std::string rccl_status = "not tried";
  unsigned long long rccl_id_hash = 0;  // FNV-1a of the ncclUniqueId this rank holds after the bootstrap broadcast (0: none yet)
  // RCCL (dlopen)
  void *rccl_lib = nullptr;
  void *rccl_comm = nullptr;
  mrl::RcclApi *rccl = nullptr;
  // statistics
  long long n_exchanges = 0;
  double bytes_sent = 0.0;
  mutable std::string err;
};

namespace mrl {

int comm_error(const mrl_comm *c, int code, const char *fmt, ...);

// host collectives over the bootstrap segment
int comm_barrier(mrl_comm *c);
int comm_allgather(mrl_comm *c, const void *mine, size_t bytes, void *all /* nranks * bytes */);  // bytes <= 256

// symmetric device memory (collective calls)
int sym_alloc(mrl_comm *c, size_t bytes, SymBuf *out, bool uncached = false);
int sym_free(mrl_comm *c, SymBuf *b);

// exchange endpoints (collective create / destroy).  counts in BYTES per peer.
int xchg_create(mrl_comm *c, Xchg *x, const size_t *send_cnt, const size_t *recv_cnt, bool want_send_buffer);
void xchg_destroy(mrl_comm *c, Xchg *x);
// (re)build the device pointer table for scatter-capable producers: entry p = where the chunk for rank p starts
// (direct: in peer p's receive buffer; otherwise in the local send buffer)
int xchg_build_table(mrl_comm *c, Xchg *x, bool direct);
// producer side.  xchg_begin: call BEFORE enqueueing the kernel that writes the send buffer (orders it behind pushes still reading it).
// xchg_post: call AFTER the producing kernel was enqueued on `stream`: pushes the data (PEER_COPY / RCCL) and raises the flags.
int xchg_begin(mrl_comm *c, Xchg *x, hipStream_t stream);
// kernel_signalled: the producing kernel was given xchg_signal_args() and raises the flags itself (PEER_STORE tables only).
int xchg_post(mrl_comm *c, Xchg *x, hipStream_t stream, bool kernel_signalled = false);
// arguments for a producer that scatters through a direct table and signals from its last workgroup; counter == nullptr
// (no in-kernel signalling) unless the table is direct.  Call BEFORE xchg_post (it names the epoch xchg_post will open).
SignalArgs xchg_signal_args(const mrl_comm *c, const Xchg *x, unsigned int nblocks);
// consumer side: orders `stream` behind the arrival of every peer's chunk of the current epoch
int xchg_wait(mrl_comm *c, Xchg *x, hipStream_t stream);
// xchg_post + xchg_wait back to back; fuse: one flag kernel for both when the table is direct (peer stores)
int xchg_post_wait(mrl_comm *c, Xchg *x, hipStream_t stream, bool fuse);
// debug (MRL_OPT_VERIFY_EXCHANGE): after xchg_wait, re-read the whole receive buffer with plain loads and with system-scope loads
// behind a system-scope acquire; differing 64-bit words are counted in c->d_verify (a stale cache line on the consumer's side)
int xchg_verify(mrl_comm *c, Xchg *x, hipStream_t stream);
int comm_alloc_channel(mrl_comm *c);
void comm_free_channel(mrl_comm *c, int channel, uint64_t epoch);
// true when producers should scatter through the table straight into the peers' buffers
inline bool xchg_direct(const mrl_comm *c) { return c->transport == MRL_TRANSPORT_PEER_STORE; }

// sum over ranks of n <= 16 device scalars (device-side: mailbox stores + flags, no host round trip); result in d_out[n]
// on every rank (identical bits: fixed summation order) and, if h_out, also read back (synchronises the stream)
int comm_allreduce_device(mrl_comm *c, hipStream_t stream, const double *d_in, int n, double *d_out, double *h_out);
// host-side all-reduce of n <= 16 host values through the bootstrap segment (op 0 sum, 1 min, 2 max)
int comm_allreduce_host(mrl_comm *c, double *h_values, int n, int op);
// check the device status word (after a stream synchronisation): timed-out waits become MRL_ERR_COMM
int comm_check(mrl_comm *c);

}  // namespace mrl
