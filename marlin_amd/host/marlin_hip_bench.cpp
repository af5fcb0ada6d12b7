// marlin-hip-bench: the multi-GPU benchmark as a native program -- C++ rank processes over the C ABI (include/marlin_hip.h) and the
// system HIP runtime, no Python / torch in any rank.  One process per GPU, as the reference assigns one MPI rank to one device from
// the host-local rank inside the program (src/actions/DomainAction.C:163-199): without rank=..., this process is only the launcher --
// it starts `gpus` copies of itself (rank=0..gpus-1, one job name) BEFORE anything touches the GPU and waits for them.
//
//   marlin-hip-bench workload=ch|mech gpus=N steps=K warmup=W [grid=n] [global_grid=G] [transport=tune|auto|1|2|3] [nsub=0|1|2|4]
//                    [carry=0|1] [exp=mask] [variants=1] [profile_steps=10] [tune_budget_s=60] [device=d] [slab=0|1] [verify=0|1]
//
//   workload=ch    Cahn-Hilliard AB2 substeps (AdamsBashforthMoulton.C:60-101) of the slab-decomposed grid through mrl_ch_substeps.
//                  Weak scaling: per-GPU work n^3 points (axes doubled y, x, z: N = 8 is the 512^3 grid of BASELINE configs[3]).
//   workload=mech  de Geus finite-strain RVE, Newton-CG through mrl_mech_newton_cg (FFTMechanics.C:96-163); a step = one CG iteration.
//
// The warm-up tunes everything the first contact with real xGMI links has to decide: {transport} x {kz sub-blocks in flight} x
// {event-ordered vs in-kernel arrival flags}.  Every candidate runs the same substeps from the same initial condition; the global
// checksums must agree; the fastest (max over ranks) carries the timed region.  verify=1 (default during tuning) additionally makes
// the consumers re-read their receive buffers with system-scope loads (MRL_OPT_VERIFY_EXCHANGE): a stale-line bug of the peer-store
// path then shows up as a count of differing elements, not as wrong physics.
// Rank 0 prints ONE JSON line; bench.py wraps it (adds nothing that was not measured here).
//
// PROFILERS.  The launcher starts its ranks with fork + exec, which is only safe while the launching process has not initialised the
// GPU.  A profiler preload (rocprofv3: LD_PRELOAD / ROCP_TOOL_LIBRARIES / HSA_TOOLS_LIB) initialises it before main(), so with
// gpus > 1 the launcher REFUSES to start under one (exit 2): profile ONE rank process directly, `rocprofv3 ... -- marlin-hip-bench
// gpus=N rank=r job=<name> ...` (the other ranks started unprofiled with the same job name), or a single-process run.  gpus=1 never
// forks or execs: the one rank runs in this process, so `rocprofv3 -- marlin-hip-bench gpus=1 ...` is fine.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "marlin_hip.h"

static std::map<std::string, std::string> g_args;
static std::string arg(const std::string & k, const std::string & d = "") { return g_args.count(k) ? g_args[k] : d; }
static long argi(const std::string & k, long d) { return g_args.count(k) ? std::atol(g_args[k].c_str()) : d; }

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

[[noreturn]] static void die(int rank, const std::string & msg)
{
  std::fprintf(stderr, "marlin-hip-bench rank %d: %s\n", rank, msg.c_str());
  std::fflush(stderr);
  _exit(1);
}

#define HIPCK(expr)                                                                                           \
  do                                                                                                          \
  {                                                                                                           \
    hipError_t e_ = (expr);                                                                                   \
    if (e_ != hipSuccess)                                                                                     \
      die(g_rank, std::string(#expr) + " failed: " + hipGetErrorString(e_));                                  \
  } while (0)

static int g_rank = -1;

static std::string jstr(const std::string & s)
{
  std::string o = "\"";
  for (char c : s)
  {
    if (c == '"' || c == '\\')
      o += '\\';
    if (c == '\n')
      o += "\\n";
    else
      o += c;
  }
  return o + "\"";
}
static std::string jnum(double v)
{
  if (!std::isfinite(v))
    return "null";
  char b[64];
  std::snprintf(b, sizeof b, "%.17g", v);
  return b;
}

static const char * transport_name(int t)
{
  switch (t)
  {
    case MRL_TRANSPORT_PEER_STORE: return "peer_store";
    case MRL_TRANSPORT_PEER_COPY: return "peer_copy";
    case MRL_TRANSPORT_RCCL: return "rccl";
    default: return "auto";
  }
}

// weak scaling: per-GPU work fixed at base^3 points; axes doubled in the order y, x, z (bench.py: grid_for)
static void grid_for(int ngpus, int64_t base, int64_t g[3])
{
  g[0] = g[1] = g[2] = base;
  const int order[3] = {1, 0, 2};
  for (int m = ngpus, k = 0; m > 1; m /= 2, ++k)
    g[order[k % 3]] *= 2;
}

// counter-based initial condition (bench.py: splitmix64_uniform): element i of the GLOBAL row-major field
static inline double splitmix(uint64_t i)
{
  uint64_t z = i * 0x9E3779B97F4A7C15ull + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return 0.44 + (0.56 - 0.44) * ((double)(z >> 11) * (1.0 / 9007199254740992.0));
}

struct Rank
{
  int world = 1, rank = 0, device = 0;
  mrl_comm * comm = nullptr;
  mrl_ctx * ctx = nullptr;
  bool slab = false;
  int64_t shape[3] = {1, 1, 1};
  double L[3] = {1, 1, 1};
  int64_t rn[3], rb[3], kn[3], kb[3];

  void ck(int rc, const char * what) const
  {
    if (rc != MRL_OK)
      die(rank, std::string(what) + ": " + (ctx ? mrl_last_error(ctx) : mrl_last_error(nullptr)));
  }
  void make_ctx()
  {
    mrl_domain d{};
    d.dim = 3;
    for (int i = 0; i < 3; ++i)
    {
      d.n[i] = shape[i];
      d.min[i] = 0.0;
      d.max[i] = L[i];
    }
    d.device = device;
    d.nranks = slab ? world : 1;
    d.rank = slab ? rank : 0;
    d.spectrum = MRL_SPECTRUM_HALF;
    d.flags = slab ? MRL_FLAG_SLAB : 0;
    if (mrl_ctx_create(&ctx, &d) != MRL_OK)
      die(rank, std::string("mrl_ctx_create: ") + mrl_last_error(nullptr));
    if (slab)
      ck(mrl_ctx_attach_comm(ctx, comm), "mrl_ctx_attach_comm");
    ck(mrl_local_shape(ctx, rn, rb, kn, kb), "mrl_local_shape");
  }
  void barrier() const
  {
    if (comm && mrl_comm_barrier(comm) != MRL_OK)
      die(rank, std::string("host barrier: ") + mrl_comm_last_error(comm));
  }
  double reduce(double v, int op) const
  {
    if (comm && mrl_comm_allreduce(comm, &v, 1, op) != MRL_OK)
      die(rank, std::string("host all-reduce: ") + mrl_comm_last_error(comm));
    return v;
  }
};

struct KernelRow
{
  std::string name;
  double ms;
  long long launches;
  double bytes;
};
static std::vector<KernelRow> read_profile(mrl_ctx * ctx)
{
  std::vector<KernelRow> rows;
  for (int slot = 0;; ++slot)
  {
    const char * name = nullptr;
    double ms = 0, bytes = 0;
    int64_t n = 0;
    if (mrl_get_profile(ctx, slot, &name, &ms, &n, &bytes) != MRL_OK)
      break;
    if (n > 0)
      rows.push_back({name, ms, (long long)n, bytes});
  }
  return rows;
}

static const double HBM_PEAK = 8000.0, HBM_COPY = 6290.0;

// ---------------------------------------------------------------------------------------------------------------------------------
struct ChState
{
  double * c[2] = {nullptr, nullptr};
  double * ring[2] = {nullptr, nullptr};
  int i = 0, head = 1, n_old = 0;
  bool started = false;
  std::vector<double> ic;  // this rank's slab of the initial condition
  size_t nreal = 0, nspec2 = 0;
};

static void ch_alloc(Rank & R, ChState & S)
{
  S.nreal = (size_t)(R.rn[0] * R.rn[1] * R.rn[2]);
  S.nspec2 = 2 * (size_t)mrl_ch_spec_elems(R.ctx);  // the solver's private layout (padded rows / x planes): include/marlin_hip.h
  for (int k = 0; k < 2; ++k)
  {
    HIPCK(hipMalloc(reinterpret_cast<void **>(&S.c[k]), sizeof(double) * S.nreal));
    HIPCK(hipMalloc(reinterpret_cast<void **>(&S.ring[k]), sizeof(double) * S.nspec2));
    HIPCK(hipMemset(S.ring[k], 0, sizeof(double) * S.nspec2));
  }
  S.ic.resize(S.nreal);
  const int64_t nx = R.rn[0], nyl = R.rn[1], nz = R.rn[2], ny = R.shape[1];
  for (int64_t ix = 0; ix < nx; ++ix)
    for (int64_t j = 0; j < nyl; ++j)
    {
      const uint64_t base = (uint64_t)((ix * ny + R.rb[1] + j) * nz);
      double * dst = S.ic.data() + (ix * nyl + j) * nz;
      for (int64_t k = 0; k < nz; ++k)
        dst[k] = splitmix(base + (uint64_t)k);
    }
}
static void ch_reset(ChState & S)
{
  HIPCK(hipMemcpy(S.c[0], S.ic.data(), sizeof(double) * S.nreal, hipMemcpyHostToDevice));
  S.i = 0;
  S.head = 1;
  S.n_old = 0;
  S.started = false;
}
// `count` substeps in one solver call; returns the library's return code (tuning tolerates failures)
static int ch_run(Rank & R, ChState & S, const mrl_ch_params & p, int count, double sub_dt)
{
  if (S.started)
  {  // TensorBuffer::advanceState between two solver calls
    S.head = (S.head + 1) % 2;
    S.n_old = 1;
  }
  const int rc = mrl_ch_substeps(R.ctx, &p, S.c[S.i], S.c[1 - S.i], S.ring, 2, &S.head, &S.n_old, 2, count, MRL_SUBSTEPS_ADVANCE, sub_dt, nullptr);
  S.i = 1 - S.i;
  S.started = true;
  return rc;
}

struct Cand
{
  int transport, nsub;
  long exp;
  std::string label;
};

// physical devices of the job as the communicator sees them (PCI bus ids gathered at its creation): mrl_comm_describe
static int comm_distinct_devices(mrl_comm * comm)
{
  if (!comm)
    return 1;
  char buf[2048] = "";
  if (mrl_comm_describe(comm, buf, sizeof buf) != MRL_OK)
    return 1;
  const char * p = std::strstr(buf, "\"distinct_devices\": ");
  return p ? std::max(1, std::atoi(p + 20)) : 1;
}
static std::string comm_devices_json(mrl_comm * comm)
{
  char buf[2048] = "";
  if (!comm || mrl_comm_describe(comm, buf, sizeof buf) != MRL_OK)
    return "[]";
  const char * p = std::strstr(buf, "\"devices_per_rank\": ");
  if (!p)
    return "[]";
  p += 20;
  const char * e = std::strchr(p, ']');
  return e ? std::string(p, e + 1) : "[]";
}

// The denominators of the parallel efficiency, measured IN THE SAME JOB: rank 0 runs the serial path (one
// context without a communicator, the N = 1 code) of grid g on its own GPU while the other ranks wait at a host barrier.  Soft
// failure: a grid that does not fit the free memory of the card is reported as such, never fatal.
struct SerialRef
{
  bool ok = false;
  double ms = 0.0;
  int64_t g[3] = {0, 0, 0};
  std::string why;
};
static SerialRef serial_reference(int device, const int64_t g[3], const mrl_ch_params & p, double sub_dt, int steps)
{
  SerialRef out;
  for (int i = 0; i < 3; ++i)
    out.g[i] = g[i];
  const double nreal = (double)g[0] * (double)g[1] * (double)g[2];
  const double hspec = 16.0 * (double)g[0] * (double)g[1] * (double)(g[2] / 2 + 1);
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess)
  {
    (void)hipGetLastError();
    out.why = "hipMemGetInfo failed";
    return out;
  }
  // two real fields, two history arrays, the context's work arrays (four spectra) + slack
  const double need = 2.0 * 8.0 * nreal + 6.5 * hspec;
  if (need > 0.9 * (double)free_b)
  {
    out.why = "needs " + jnum(need / 1e9) + " GB of device memory, " + jnum((double)free_b / 1e9) + " GB free on rank 0's card";
    return out;
  }
  mrl_domain d{};
  d.dim = 3;
  const double dx = 8.0 * M_PI / 200.0;
  for (int i = 0; i < 3; ++i)
  {
    d.n[i] = g[i];
    d.min[i] = 0.0;
    d.max[i] = (double)g[i] * dx;
  }
  d.device = device;
  d.nranks = 1;
  d.spectrum = MRL_SPECTRUM_HALF;
  mrl_ctx * ctx = nullptr;
  if (mrl_ctx_create(&ctx, &d) != MRL_OK)
  {
    out.why = std::string("mrl_ctx_create: ") + mrl_last_error(nullptr);
    return out;
  }
  double * c[2] = {nullptr, nullptr};
  double * ring[2] = {nullptr, nullptr};
  const size_t nr = (size_t)nreal, ns2 = 2 * (size_t)mrl_ch_spec_elems(ctx);
  bool alloc = true;
  for (int k = 0; k < 2; ++k)
  {
    alloc = alloc && hipMalloc(reinterpret_cast<void **>(&c[k]), sizeof(double) * nr) == hipSuccess;
    alloc = alloc && hipMalloc(reinterpret_cast<void **>(&ring[k]), sizeof(double) * ns2) == hipSuccess;
    if (alloc)
      alloc = hipMemset(ring[k], 0, sizeof(double) * ns2) == hipSuccess;
  }
  if (alloc)
  {
    std::vector<double> ic(nr);
    for (size_t i = 0; i < nr; ++i)
      ic[i] = splitmix((uint64_t)i);
    alloc = hipMemcpy(c[0], ic.data(), sizeof(double) * nr, hipMemcpyHostToDevice) == hipSuccess;
  }
  if (!alloc)
  {
    (void)hipGetLastError();
    out.why = "device allocation failed";
  }
  else
  {
    int head = 1, n_old = 0, cur = 0, rc = MRL_OK;
    auto run = [&](int count) {
      rc = mrl_ch_substeps(ctx, &p, c[cur], c[1 - cur], ring, 2, &head, &n_old, 2, count, MRL_SUBSTEPS_ADVANCE, sub_dt, nullptr);
      cur = 1 - cur;
      head = (head + 1) % 2;
      n_old = 1;
    };
    run(5);
    std::vector<double> t;
    const double t_warm = now_s();
    while (rc == MRL_OK && now_s() - t_warm < 0.15)  // clocks
      run(steps);
    (void)mrl_sync(ctx);
    for (int r = 0; r < 5 && rc == MRL_OK; ++r)
    {
      (void)hipDeviceSynchronize();
      const double t0 = now_s();
      run(steps);
      if (rc == MRL_OK)
        rc = mrl_sync(ctx);
      t.push_back((now_s() - t0) / steps * 1e3);
    }
    if (rc != MRL_OK)
      out.why = std::string("mrl_ch_substeps: ") + mrl_last_error(ctx);
    else
    {
      std::sort(t.begin(), t.end());
      out.ms = t[t.size() / 2];
      out.ok = true;
    }
  }
  for (int k = 0; k < 2; ++k)
  {
    if (c[k])
      (void)hipFree(c[k]);
    if (ring[k])
      (void)hipFree(ring[k]);
  }
  mrl_ctx_destroy(ctx);
  return out;
}

static int run_ch(Rank & R)
{
  const int steps = (int)argi("steps", 100), warmup = (int)argi("warmup", 10);
  const int64_t n = argi("grid", 256), G = argi("global_grid", 0);
  const bool carry = argi("carry", 0) != 0;
  const long exp_user = argi("exp", 0);
  const int profile_steps = (int)argi("profile_steps", 10);
  const double tune_budget = (double)argi("tune_budget_s", 45);
  double tuning_s = 0.0;
  const bool variants = argi("variants", 1) != 0;
  if (G)
    R.shape[0] = R.shape[1] = R.shape[2] = G;
  else
    grid_for(R.world, n, R.shape);
  const double dx = 8.0 * M_PI / 200.0;  // examples/cahn_hilliard/cahnhilliard2.i:8-13
  for (int i = 0; i < 3; ++i)
    R.L[i] = (double)R.shape[i] * dx;
  R.make_ctx();
  mrl_ch_params p{};
  p.family = MRL_FE_DOUBLE_WELL;  // f = 0.1 c^2 (c-1)^2, M = 0.2, kappa factor -0.001 (cahnhilliard2.i:61-91)
  p.coef[0] = 0.1;
  p.mobility = 0.2;
  p.kappa = -0.001;
  const double sub_dt = 1e-3;
  const double npts = (double)R.shape[0] * (double)R.shape[1] * (double)R.shape[2];
  ChState S;
  ch_alloc(R, S);

  auto set_opts = [&](int nsub, long exp, bool carry_on) {
    R.ck(mrl_ctx_set_option(R.ctx, MRL_OPT_SLAB_NSUB, nsub), "MRL_OPT_SLAB_NSUB");
    R.ck(mrl_ctx_set_option(R.ctx, MRL_OPT_SLAB_CARRY, carry_on ? 1 : 0), "MRL_OPT_SLAB_CARRY");
    R.ck(mrl_ctx_set_option(R.ctx, MRL_OPT_EXPERIMENT, exp), "MRL_OPT_EXPERIMENT");
  };
  auto checksum = [&]() {
    double s = 0.0;
    R.ck(mrl_dot(R.ctx, S.c[S.i], S.c[S.i], (int64_t)S.nreal, &s), "mrl_dot");  // global on slab contexts
    return s;
  };
  auto sum_c = [&]() {
    double s = 0.0;
    R.ck(mrl_sum(R.ctx, S.c[S.i], (int64_t)S.nreal, &s), "mrl_sum");
    return s;
  };
  // timed region: barrier + device synchronisation on both sides, max over ranks
  auto timed = [&](int count, int per_call) -> double {
    HIPCK(hipDeviceSynchronize());
    R.barrier();
    const double t0 = now_s();
    for (int done = 0; done < count;)
    {
      const int k = std::min(per_call, count - done);
      R.ck(ch_run(R, S, p, k, sub_dt), "mrl_ch_substeps");
      done += k;
    }
    R.ck(mrl_sync(R.ctx), "mrl_sync");
    HIPCK(hipDeviceSynchronize());
    R.barrier();
    return R.reduce(now_s() - t0, 2);
  };

  // ---- tuning ------------------------------------------------------------------------------------------------------------------
  std::ostringstream tuned;
  std::string selected = "serial", flag_variant = "n/a";
  int sel_transport = 0, sel_nsub = 0;
  long sel_exp = exp_user;
  long long verify_mismatches = -1;
  if (R.slab)
  {
    const std::string tr = arg("transport", "tune");
    const int nsub_user = (int)argi("nsub", 0);
    const int distinct_devices = comm_distinct_devices(R.comm);
    std::vector<Cand> cands;
    if (tr == "tune")
    {
      for (int ns : {1, 2, 4})
      {
        if (nsub_user && ns != nsub_user)
          continue;
        cands.push_back({MRL_TRANSPORT_PEER_STORE, ns, 0, "event-ordered flags"});
        cands.push_back({MRL_TRANSPORT_PEER_COPY, ns, 0, ""});
        // (RCCL before the in-kernel-flag variant: every multi-device job times the RCCL candidate at least once inside the budget
        // and reports runtime.rccl_comm_nranks, whichever transport wins)
        cands.push_back({MRL_TRANSPORT_RCCL, ns, 0, ""});
        // in-kernel arrival flags: 3 x slower than event-ordered ones in every run on one device (one L2 write-back per workgroup);
        // only worth a slot where the stores really cross a link
        if (distinct_devices > 1 || argi("tune_in_kernel_flags", 0) != 0)
          cands.push_back({MRL_TRANSPORT_PEER_STORE, ns, 128, "in-kernel flags"});
      }
    }
    else
    {
      const int t = tr == "auto" ? mrl_comm_transport(R.comm) : std::atoi(tr.c_str());
      cands.push_back({t, nsub_user ? nsub_user : 1, 0, ""});
    }
    struct Res
    {
      Cand c;
      double ms, sum;
      bool ok;
      std::string why;
      long long bad;
    };
    std::vector<Res> results;
    std::map<int, std::string> unavailable;
    std::set<std::pair<int, long>> failed_variants;
    const double t_tune0 = now_s();
    mrl_comm_set_timeout(R.comm, 20.0);
    const bool verify = argi("verify", 1) != 0;
    for (const Cand & c : cands)
    {
      // the budget is a collective decision (rank 0's clock)
      double over = (R.rank == 0 && now_s() - t_tune0 > tune_budget && !results.empty()) ? 1.0 : 0.0;
      if (R.reduce(over, 2) != 0.0)
        break;
      if (unavailable.count(c.transport) || failed_variants.count({c.transport, c.exp}))
        continue;  // (a variant that failed once is not tried again with another sub-block count: each failure costs a time-out)
      if (mrl_comm_transport(R.comm) != c.transport && mrl_comm_set_transport(R.comm, c.transport) != MRL_OK)
      {  // (collective verdict: every rank lands here)
        unavailable[c.transport] = mrl_comm_last_error(R.comm);
        continue;
      }
      Res r{c, 0.0, 0.0, true, "", -1};
      set_opts(c.nsub, exp_user | c.exp, carry);
      if (verify)
      {  // the first three substeps of every candidate: consumers re-read their receive buffers with system-scope loads
        R.ck(mrl_ctx_set_option(R.ctx, MRL_OPT_VERIFY_MISMATCHES, 0), "MRL_OPT_VERIFY_MISMATCHES");
        R.ck(mrl_ctx_set_option(R.ctx, MRL_OPT_VERIFY_EXCHANGE, 1), "MRL_OPT_VERIFY_EXCHANGE");
      }
      ch_reset(S);
      int rc = ch_run(R, S, p, 3, sub_dt);
      if (rc == MRL_OK)
        rc = mrl_sync(R.ctx);
      if (verify)
      {  // (collective on every rank, whatever its own return code was)
        const int64_t bad = rc == MRL_OK ? mrl_ctx_get_option(R.ctx, MRL_OPT_VERIFY_MISMATCHES) : 0;
        r.bad = (long long)R.reduce((double)(bad < 0 ? 0 : bad), 0);
        (void)mrl_ctx_set_option(R.ctx, MRL_OPT_VERIFY_EXCHANGE, 0);  // the timed substeps run without the re-reads
      }
      if (rc == MRL_OK)
      {
        HIPCK(hipDeviceSynchronize());
        R.barrier();
        const double t0 = now_s();
        rc = ch_run(R, S, p, 6, sub_dt);
        if (rc == MRL_OK)
          rc = mrl_sync(R.ctx);
        r.ms = (now_s() - t0) / 6 * 1e3;
      }
      if (rc != MRL_OK)
      {
        r.ok = false;
        r.why = mrl_last_error(R.ctx);
      }
      const double all_ok = R.reduce(r.ok ? 1.0 : 0.0, 1);
      if (all_ok == 0.0)
      {
        if (r.ok)
          r.why = "failed on another rank";
        r.ok = false;
        failed_variants.insert({c.transport, c.exp});
        // tear the pipeline down on every rank, clear the condition, start over with fresh exchange buffers
        (void)hipDeviceSynchronize();
        mrl_ctx_destroy(R.ctx);
        R.ctx = nullptr;
        mrl_comm_reset_error(R.comm);
        R.barrier();
        R.make_ctx();
      }
      else
      {
        r.ms = R.reduce(r.ms, 2);
        r.sum = checksum();
      }
      results.push_back(r);
    }
    mrl_comm_set_timeout(R.comm, 120.0);
    tuning_s = R.reduce(now_s() - t_tune0, 2);
    // reference checksum: the most conservative transport that ran -- RCCL (its own rendezvous and fences), else the copy engines
    // (hipMemcpyAsync between IPC mappings), else the median of the peer-store variants.  The peer-store candidates share one
    // memory-model argument (profiles/HISTORY.md 4.1a): if it failed on this node they could agree with each other and still be wrong, so
    // they must not outvote a transport that does not depend on it.  A candidate that disagrees is disqualified.
    double ref = 0.0;
    bool have_ref = false;
    for (int want : {MRL_TRANSPORT_RCCL, MRL_TRANSPORT_PEER_COPY})
      for (auto & r : results)
        if (!have_ref && r.ok && r.c.transport == want)
        {
          ref = r.sum;
          have_ref = true;
        }
    if (!have_ref)
    {
      std::vector<double> sums;
      for (auto & r : results)
        if (r.ok)
          sums.push_back(r.sum);
      std::sort(sums.begin(), sums.end());
      ref = sums.empty() ? 0.0 : sums[sums.size() / 2];
    }
    int best = -1;
    tuned << "[";
    for (size_t k = 0; k < results.size(); ++k)
    {
      auto & r = results[k];
      const bool agrees = r.ok && std::fabs(r.sum - ref) <= 1e-12 * std::fabs(ref) && r.bad <= 0;
      if (agrees && (best < 0 || r.ms < results[best].ms))
        best = (int)k;
      tuned << (k ? ", " : "") << "{\"transport\": " << jstr(transport_name(r.c.transport)) << ", \"nsub\": " << r.c.nsub;
      if (!r.c.label.empty())
        tuned << ", \"flags\": " << jstr(r.c.label);
      if (r.ok)
        tuned << ", \"ms_per_step\": " << jnum(r.ms) << ", \"checksum\": " << jnum(r.sum) << ", \"checksum_agrees\": " << (agrees ? "true" : "false");
      else
        tuned << ", \"failed\": " << jstr(r.why.substr(0, 200));
      if (r.bad >= 0)
        tuned << ", \"receive_buffer_reread_mismatches\": " << r.bad;
      tuned << "}";
    }
    for (auto & u : unavailable)
      tuned << (tuned.tellp() > 1 ? ", " : "") << "{\"transport\": " << jstr(transport_name(u.first)) << ", \"unavailable\": " << jstr(u.second.substr(0, 200)) << "}";
    tuned << "]";
    if (best < 0)
      die(R.rank, "no transport candidate produced an agreeing result on this node: " + tuned.str());
    sel_transport = results[best].c.transport;
    sel_nsub = results[best].c.nsub;
    sel_exp = exp_user | results[best].c.exp;
    flag_variant = sel_transport == MRL_TRANSPORT_PEER_STORE ? (results[best].c.exp & 128 ? "in-kernel" : "event-ordered") : "n/a";
    verify_mismatches = results[best].bad;
    selected = transport_name(sel_transport);
    if (mrl_comm_transport(R.comm) != sel_transport && mrl_comm_set_transport(R.comm, sel_transport) != MRL_OK)
      die(R.rank, std::string("re-selecting the tuned transport failed: ") + mrl_comm_last_error(R.comm));
    set_opts(sel_nsub, sel_exp, carry);
  }

  // ---- warm-up, timed region -------------------------------------------------------------------------------------------------
  ch_reset(S);
  const double mass0 = sum_c();
  const int per_call = argi("substeps_per_call", 0) > 0 ? (int)argi("substeps_per_call", 0) : steps;
  for (int done = 0; done < warmup;)
  {
    const int k = std::min(per_call, warmup - done);
    R.ck(ch_run(R, S, p, k, sub_dt), "mrl_ch_substeps (warm-up)");
    done += k;
  }
  HIPCK(hipDeviceSynchronize());
  int64_t ex0 = 0, ex1 = 0;
  double by0 = 0, by1 = 0;
  if (R.comm)
    mrl_comm_stats(R.comm, &ex0, &by0);
  // clock warm-up by TIME (untimed regions of `steps` substeps until clock_warmup_ms have passed; timed() returns the maximum over the
  // ranks, so every rank leaves the loop together), then `repeats` regions of exactly `steps` substeps back to back: the MEDIAN is
  // ms_per_step (median of repeats; VERDICT r03: a single 7 ms region after 2 ms of warm-up read 7 % slow)
  const double clock_warmup_s = (double)argi("clock_warmup_ms", 150) * 1e-3;
  const int repeats = std::max(1, (int)argi("repeats", 7));
  int warm_regions = 0;
  double warm_s = 0.0;
  while (warm_s < clock_warmup_s && warm_regions < 1000)
  {
    warm_s += timed(steps, per_call);
    ++warm_regions;
  }
  if (R.comm)
    mrl_comm_stats(R.comm, &ex0, &by0);
  std::vector<double> reps;
  for (int r = 0; r < repeats; ++r)
    reps.push_back(timed(steps, per_call));
  std::vector<double> sorted_reps = reps;
  std::sort(sorted_reps.begin(), sorted_reps.end());
  const double elapsed = sorted_reps[sorted_reps.size() / 2];
  const double timed_steps_total = (double)steps * (double)reps.size();
  if (R.comm)
    mrl_comm_stats(R.comm, &ex1, &by1);
  double single_ms = NAN;
  if (per_call != 1)
  {
    const int n1 = std::min(steps, 50);
    R.ck(ch_run(R, S, p, 1, sub_dt), "mrl_ch_substeps");
    single_ms = timed(n1, 1) / n1 * 1e3;
  }
  // the scheme conserves mass exactly (the k = 0 mode has Mbar = Lbar = 0): a wrong exchange or a missed dependency shows up here
  const double mass1 = sum_c();
  double lo = 0, hi = 0;
  R.ck(mrl_minmax(R.ctx, S.c[S.i], (int64_t)S.nreal, &lo, &hi), "mrl_minmax");
  if (!(std::fabs(mass1 - mass0) <= 1e-11 * std::fabs(mass0)) || !(lo > 0.0 && hi < 1.0))
    die(R.rank, "sanity check failed: mass " + jnum(mass0) + " -> " + jnum(mass1) + ", field range [" + jnum(lo) + ", " + jnum(hi) + "]");

  // ---- per-kernel device time (HIP events on the launch stream) -----------------------------------------------------------
  R.ck(mrl_set_profiling(R.ctx, 1), "mrl_set_profiling");
  R.ck(ch_run(R, S, p, profile_steps, sub_dt), "mrl_ch_substeps (profile)");
  R.ck(mrl_sync(R.ctx), "mrl_sync");
  const std::vector<KernelRow> kernels = read_profile(R.ctx);
  R.ck(mrl_set_profiling(R.ctx, 0), "mrl_set_profiling");
  // checksum after a DETERMINISTIC number of substeps from the initial condition (the protocol above runs a box-dependent number of
  // warm-up regions): warmup + steps, in calls of per_call -- bench.py --driver python does the same, their lines are comparable
  ch_reset(S);
  for (int done = 0; done < warmup + steps;)
  {
    const int k = std::min(per_call, warmup + steps - done);
    R.ck(ch_run(R, S, p, k, sub_dt), "mrl_ch_substeps (checksum run)");
    done += k;
  }
  R.ck(mrl_sync(R.ctx), "mrl_sync");
  const double cs = checksum();

  // ---- variants --------------------------------------------------------------------------------------------------------------
  std::ostringstream var;
  double local_only_ms = NAN;
  if (variants && R.slab)
  {
    const int k = std::min(steps, 40);
    set_opts(sel_nsub, sel_exp, !carry);
    ch_reset(S);
    R.ck(ch_run(R, S, p, 3, sub_dt), "carry variant");
    const double ms_c = timed(k, k) / k * 1e3;
    set_opts(sel_nsub, sel_exp | 64, carry);
    ch_reset(S);
    R.ck(ch_run(R, S, p, 3, sub_dt), "local-only variant");
    const double ms_l = timed(k, k) / k * 1e3;
    local_only_ms = ms_l;
    set_opts(sel_nsub, sel_exp, carry);
    var << "{\"spectral_carry_over_" << (carry ? "off" : "on") << "\": {\"ms_per_step\": " << jnum(ms_c) << ", \"value\": " << jnum(npts / (ms_c * 1e-3))
        << "}, \"local_kernels_only\": {\"ms_per_step\": " << jnum(ms_l) << ", \"value\": " << jnum(npts / (ms_l * 1e-3))
        << ", \"note\": \"the same launches without any exchange or wait: what the rank-local work costs\"}}";
  }

  // ---- parallel efficiency: both denominators from this job ------------------------------------------------------------------
  SerialRef ref_weak, ref_strong;
  if (R.slab && R.world > 1 && argi("efficiency", 1) != 0)
  {
    HIPCK(hipDeviceSynchronize());
    R.barrier();
    if (R.rank == 0)
    {
      const int k = std::min(std::max(steps, 4), 20);
      const int64_t base[3] = {n, n, n};
      if (!G)
        ref_weak = serial_reference(R.device, base, p, sub_dt, k);  // what ONE GPU does with one rank's share of the points
      ref_strong = serial_reference(R.device, R.shape, p, sub_dt, k);  // what ONE GPU does with the whole grid of this job
    }
    mrl_comm_set_timeout(R.comm, 600.0);
    R.barrier();
    mrl_comm_set_timeout(R.comm, 120.0);
  }

  // devices per rank (all-gathered through the 16-value host all-reduce: sums of one-hot rows)
  std::vector<int> devs(R.world, R.device);
  if (R.comm)
    for (int base = 0; base < R.world; base += 16)
    {
      double v[16] = {};
      const int cnt = std::min(16, R.world - base);
      if (R.rank >= base && R.rank < base + cnt)
        v[R.rank - base] = (double)R.device;
      if (mrl_comm_allreduce(R.comm, v, cnt, 0) != MRL_OK)
        die(R.rank, "all-reduce of the device list failed");
      for (int i = 0; i < cnt; ++i)
        devs[base + i] = (int)v[i];
    }
  char describe[2048] = "{}";
  if (R.comm)
    mrl_comm_describe(R.comm, describe, sizeof describe);

  if (R.rank == 0)
  {
    const double value = npts * steps / elapsed;
    const double h = 8.0 * (1.0 + 2.0 / (double)R.shape[2]);
    const double bpu = 3.0 * (8.0 + 5.0 * h) + 1 * h;  // 3 B_fft(n) + n_old * 8 (1 + 2/n)
    std::ostringstream o;
    o << "{\"metric\": \"grid-point-updates/sec, 3-D Cahn-Hilliard semi-implicit spectral substep (AB2, fp64)\", \"value\": " << jnum(value)
      << ", \"unit\": \"grid-point-updates/s\", \"n_gpus\": " << R.world << ", \"steps\": " << steps << ", \"warmup\": " << warmup
      << ", \"ms_per_step\": " << jnum(elapsed / steps * 1e3) << ", \"higher_is_better\": true, \"scaling\": " << (G ? "\"strong\"" : "\"weak\"")
      << ", \"vs_baseline\": null, \"dtype\": \"f64\", \"data\": \"synthetic (splitmix64 uniform [0.44,0.56] initial concentration)\"";
    o << ", \"repeats_ms\": [";
    for (size_t r = 0; r < reps.size(); ++r)
      o << (r ? ", " : "") << jnum(reps[r] / steps * 1e3);
    o << "], \"timing_protocol\": \"SURVEY 8(d): warm-up, then the median of repeated regions, no host sync inside a region.  Here: " << warmup
      << " warm-up substeps + " << warm_regions << " untimed region(s) of " << steps << " substeps (" << (long)(warm_s * 1e3) << " ms, clock_warmup_ms "
      << (long)(clock_warmup_s * 1e3) << "), then " << reps.size() << " timed regions of exactly " << steps
      << " substeps back to back, each between barrier + device synchronisation, MAX over ranks; ms_per_step and value are the MEDIAN region\"";
    o << ", \"config\": {\"workload\": \"3D Cahn-Hilliard " << R.shape[0] << "x" << R.shape[1] << "x" << R.shape[2]
      << " fp64 semi-implicit spectral step, AB2, f=0.1c^2(c-1)^2, M=0.2, kappa=-0.001, sub_dt=1e-3\", \"grid\": [" << R.shape[0] << ", " << R.shape[1]
      << ", " << R.shape[2] << "], \"decomposition\": "
      << (R.slab ? jstr("slab x" + std::to_string(R.world) + ", library-owned exchange (mrl_comm), " + std::to_string(sel_nsub) + " kz sub-block(s) in flight")
                 : jstr("none"))
      << ", \"driver\": \"native (marlin-hip-bench: C++ rank processes over the C ABI, system HIP runtime)\", \"spectral_carry_over\": "
      << (carry ? "true" : "false") << ", \"substeps_per_library_call\": " << per_call << ", \"ms_per_step_with_one_call_per_substep\": " << jnum(single_ms)
      << "}";
    o << ", \"substep_algorithmic_bytes_per_update\": " << jnum(bpu) << ", \"substep_model_GBps\": " << jnum(value * bpu / 1e9)
      << ", \"substep_model_frac_of_hbm_peak\": " << jnum(value * bpu / 1e9 / HBM_PEAK / R.world)
      << ", \"substep_model_frac_of_copy_ceiling\": " << jnum(value * bpu / 1e9 / HBM_COPY / R.world);
    // dominant kernel
    const KernelRow * dom = nullptr;
    for (auto & k : kernels)
      if (k.bytes > 0 && (!dom || k.ms > dom->ms))
        dom = &k;
    if (dom)
    {
      const double avg = dom->ms / dom->launches, gbps = dom->bytes / (avg * 1e-3) / 1e9;
      o << ", \"roofline\": {\"bound\": \"hbm\", \"kernel\": " << jstr(dom->name) << ", \"achieved\": " << jnum(gbps) << ", \"peak\": " << HBM_PEAK
        << ", \"unit\": \"GB/s\", \"frac\": " << jnum(gbps / HBM_PEAK) << ", \"frac_of_measured_copy_ceiling\": " << jnum(gbps / HBM_COPY)
        << ", \"traffic\": null, \"avg_launch_ms\": " << jnum(avg) << ", \"algorithmic_bytes_per_launch\": " << jnum(dom->bytes)
        << (R.slab ? ", \"note\": \"kernels that store into peer memory or wait for it are timed with the exchange they carry; "
                     "variants.local_kernels_only has the rank-local cost\""
                   : "")
        << "}";
    }
    o << ", \"kernels\": [";
    double loc = 0, waits = 0;
    for (size_t k = 0; k < kernels.size(); ++k)
    {
      const double avg = kernels[k].ms / kernels[k].launches;
      o << (k ? ", " : "") << "{\"kernel\": " << jstr(kernels[k].name) << ", \"avg_ms\": " << jnum(avg) << ", \"launches_per_step\": "
        << jnum((double)kernels[k].launches / profile_steps) << ", \"algorithmic_GBps\": " << jnum(kernels[k].bytes > 0 ? kernels[k].bytes / (avg * 1e-3) / 1e9 : 0.0)
        << "}";
      if (kernels[k].bytes > 0)
        loc += kernels[k].ms / profile_steps;
      if (kernels[k].name == "slab_exchange_wait")
        waits += kernels[k].ms / profile_steps;
    }
    o << "]";
    if (R.slab)
    {
      o << ", \"exchange\": {\"ranks\": " << R.world << ", \"devices_per_rank\": [";
      for (int r = 0; r < R.world; ++r)
        o << (r ? ", " : "") << devs[r];
      int ndistinct = 0;
      {
        std::vector<int> d = devs;
        std::sort(d.begin(), d.end());
        ndistinct = (int)(std::unique(d.begin(), d.end()) - d.begin());
      }
      o << "], \"distinct_devices\": " << ndistinct << ", \"transport\": {\"selected\": " << jstr(selected) << ", \"nsub\": " << sel_nsub
        << ", \"arrival_flags\": " << jstr(flag_variant) << ", \"tuning_s\": " << jnum(tuning_s) << ", \"tune_budget_s\": " << jnum(tune_budget)
        << ", \"tuned\": " << (tuned.str().empty() ? "[]" : tuned.str()) << "}, \"physical_devices_per_rank\": " << comm_devices_json(R.comm);
      if (verify_mismatches >= 0)
        o << ", \"receive_buffer_reread_mismatches\": " << verify_mismatches;
      o << ", \"kernel_ms_per_step_incl_peer_stores\": " << jnum(loc) << ", \"exposed_wait_ms_per_step\": " << jnum(waits)
        << ", \"exchanges_per_step\": " << jnum((double)(ex1 - ex0) / timed_steps_total) << ", \"bytes_sent_to_peers_per_step_rank0\": " << jnum((by1 - by0) / timed_steps_total)
        << ", \"link_GBps_out_rank0\": " << jnum((by1 - by0) / timed_steps_total / (elapsed / steps) / 1e9) << ", \"runtime\": " << describe << "}";
    }
    if (!var.str().empty())
      o << ", \"variants\": " << var.str();
    if (R.slab)
    {
      // the two readings of north_star's "parallel efficiency on 512^3": WEAK = against one GPU doing one rank's share (grid^3
      // points), STRONG = against one GPU doing this job's whole grid.  efficiency = value(N) / (N * value(1)); both value(1)
      // were measured by rank 0 in this job (serial context, same kernels as the N = 1 line), nothing is taken from a file.
      auto side = [&](const char * name, const SerialRef & r) {
        o << "\"" << name << "\": ";
        if (r.ok)
        {
          const double pts = (double)r.g[0] * (double)r.g[1] * (double)r.g[2], v1 = pts / (r.ms * 1e-3);
          o << "{\"efficiency\": " << jnum(value / ((double)R.world * v1)) << ", \"one_gpu\": {\"grid\": [" << r.g[0] << ", " << r.g[1] << ", "
            << r.g[2] << "], \"ms_per_step\": " << jnum(r.ms) << ", \"value\": " << jnum(v1) << "}}";
        }
        else
          o << "{\"efficiency\": null, \"why\": " << jstr(r.why.empty() ? (R.world > 1 ? "not measured" : "one rank") : r.why) << "}";
      };
      o << ", \"parallel_efficiency\": {";
      side("weak", ref_weak);
      o << ", ";
      side("strong", ref_strong);
      o << ", \"definition\": \"value(N) / (N * value(1)); value(1) measured by rank 0 of this job on its own GPU with the serial path while the "
           "other ranks waited: weak = one GPU on one rank's share of the points (grid^3), strong = one GPU on this job's whole grid\"}";
      o << ", \"local_kernels_only_ms\": " << jnum(local_only_ms) << ", \"exposed_wait_ms_per_step\": " << jnum(waits);
    }
    o << ", \"field_checksum\": {\"sum_c_squared\": " << jnum(cs) << "}}";
    std::printf("%s\n", o.str().c_str());
    std::fflush(stdout);
  }
  for (int k = 0; k < 2; ++k)
  {
    (void)hipFree(S.c[k]);
    (void)hipFree(S.ring[k]);
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// BASELINE configs[2] (N = 1) / configs[4] (N = 8: 256^3): de Geus finite-strain RVE, Newton-CG with the FFT-applied Gamma operator.
// Cubic inclusion phase[-s:, :s, -s:] = 1, s = 9n/32 (test/src/tensor_computes/PhaseMechanicsTest.C:36-45), K = 0.833 / 8.33,
// mu = 0.386 / 3.86 (examples/degeus_mechanics/mech.i:23-38), shear ramp, l_tol = 1e-2, nl tolerances 2e-2.
static int run_mech(Rank & R)
{
  const int steps = (int)argi("steps", 3), warmup = 1;
  const int64_t n = argi("grid", 128), G = argi("global_grid", 0);
  if (G)
    R.shape[0] = R.shape[1] = R.shape[2] = G;
  else
    grid_for(R.world, n, R.shape);
  for (int i = 0; i < 3; ++i)
    R.L[i] = 2.0 * M_PI;
  R.make_ctx();
  R.ck(mrl_ctx_set_option(R.ctx, MRL_OPT_EXPERIMENT, argi("exp", 0)), "MRL_OPT_EXPERIMENT");
  const int64_t nx = R.shape[0], ny = R.shape[1], nz = R.shape[2], nyl = R.rn[1], yb = R.rb[1];
  const size_t npl = (size_t)(nx * nyl * nz);
  std::vector<double> hK(npl), hmu(npl), hF(npl * 9, 0.0);
  const int64_t sx = 9 * nx / 32, sy = 9 * ny / 32, sz = 9 * nz / 32;
  for (int64_t i = 0; i < nx; ++i)
    for (int64_t j = 0; j < nyl; ++j)
      for (int64_t k = 0; k < nz; ++k)
      {
        const bool inc = i >= nx - sx && (yb + j) < sy && k >= nz - sz;
        const size_t e = (size_t)((i * nyl + j) * nz + k);
        hK[e] = inc ? 8.33 : 0.833;
        hmu[e] = inc ? 3.86 : 0.386;
        hF[e * 9 + 0] = hF[e * 9 + 4] = hF[e * 9 + 8] = 1.0;
      }
  double *dK, *dmu, *dF[2], *dP, *dA;
  HIPCK(hipMalloc(reinterpret_cast<void **>(&dK), 8 * npl));
  HIPCK(hipMalloc(reinterpret_cast<void **>(&dmu), 8 * npl));
  HIPCK(hipMalloc(reinterpret_cast<void **>(&dF[0]), 72 * npl));
  HIPCK(hipMalloc(reinterpret_cast<void **>(&dF[1]), 72 * npl));
  HIPCK(hipMalloc(reinterpret_cast<void **>(&dP), 72 * npl));
  HIPCK(hipMalloc(reinterpret_cast<void **>(&dA), 72));
  HIPCK(hipMemcpy(dK, hK.data(), 8 * npl, hipMemcpyHostToDevice));
  HIPCK(hipMemcpy(dmu, hmu.data(), 8 * npl, hipMemcpyHostToDevice));
  HIPCK(hipMemcpy(dF[0], hF.data(), 72 * npl, hipMemcpyHostToDevice));
  mrl_mech_params mp{};
  mp.l_tol = 1e-2;
  mp.l_max_its = 0;
  mp.nl_rel_tol = mp.nl_abs_tol = 2e-2;
  mp.nl_max_its = 100;
  const double sub_dt = 0.01 / 10;
  int cur = 0;
  auto solve = [&](int it, mrl_mech_stats & st) -> double {
    double avg[9], applied[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    R.ck(mrl_average(R.ctx, dF[cur], 9, avg), "mrl_average");
    applied[1] += it * sub_dt;
    for (int i = 0; i < 9; ++i)
      applied[i] -= avg[i];
    HIPCK(hipMemcpy(dA, applied, 72, hipMemcpyHostToDevice));
    HIPCK(hipDeviceSynchronize());
    const double t0 = now_s();
    R.ck(mrl_mech_newton_cg(R.ctx, &mp, dF[cur], dK, dmu, dA, dF[1 - cur], dP, &st), "mrl_mech_newton_cg");
    HIPCK(hipDeviceSynchronize());
    cur = 1 - cur;
    return now_s() - t0;
  };
  // transports: time one solve with each (peer stores run all three tensor rows per launch, copies the row pipeline)
  std::ostringstream tuned;
  std::string selected = "serial";
  if (R.slab)
  {
    const std::string tr = arg("transport", "tune");
    std::vector<int> cands;
    if (tr == "tune")
      cands = {MRL_TRANSPORT_PEER_STORE, MRL_TRANSPORT_PEER_COPY, MRL_TRANSPORT_RCCL};
    else
      cands = {tr == "auto" ? mrl_comm_transport(R.comm) : std::atoi(tr.c_str())};
    int best = 0;
    double best_ms = 1e300;
    tuned << "[";
    bool first = true;
    for (int t : cands)
    {
      if (mrl_comm_transport(R.comm) != t && mrl_comm_set_transport(R.comm, t) != MRL_OK)
      {
        tuned << (first ? "" : ", ") << "{\"transport\": " << jstr(transport_name(t)) << ", \"unavailable\": " << jstr(std::string(mrl_comm_last_error(R.comm)).substr(0, 200)) << "}";
        first = false;
        continue;
      }
      HIPCK(hipMemcpy(dF[0], hF.data(), 72 * npl, hipMemcpyHostToDevice));
      cur = 0;
      mrl_mech_stats st{};
      solve(0, st);  // buffers, exchange pipes
      R.barrier();
      const double dt = solve(1, st);
      const double ms = R.reduce(dt, 2) / std::max(1, (int)st.cg_its_total) * 1e3;
      double fn = 0.0;
      R.ck(mrl_norm2(R.ctx, dF[cur], (int64_t)(9 * npl), &fn), "mrl_norm2");
      tuned << (first ? "" : ", ") << "{\"transport\": " << jstr(transport_name(t)) << ", \"ms_per_cg_iteration\": " << jnum(ms) << ", \"cg_iterations\": " << st.cg_its_total
            << ", \"norm_F\": " << jnum(fn) << "}";
      first = false;
      if (ms < best_ms)
      {
        best_ms = ms;
        best = t;
      }
    }
    tuned << "]";
    if (!best)
      die(R.rank, "no transport works on this node: " + tuned.str());
    if (mrl_comm_transport(R.comm) != best && mrl_comm_set_transport(R.comm, best) != MRL_OK)
      die(R.rank, "re-selecting the tuned transport failed");
    selected = transport_name(best);
  }
  HIPCK(hipMemcpy(dF[0], hF.data(), 72 * npl, hipMemcpyHostToDevice));
  cur = 0;
  mrl_mech_stats st{};
  solve(0, st);  // warm-up
  int64_t ex0 = 0, ex1 = 0;
  double by0 = 0, by1 = 0;
  if (R.comm)
    mrl_comm_stats(R.comm, &ex0, &by0);
  double tot_t = 0.0;
  long tot_its = 0;
  std::ostringstream newton;
  const int substeps = std::max(1, std::min(steps, 3));
  for (int it = 1; it <= substeps; ++it)
  {
    R.barrier();
    tot_t += solve(it, st);
    tot_its += st.cg_its_total;
    newton << (it > 1 ? ", " : "") << st.newton_its;
  }
  if (R.comm)
    mrl_comm_stats(R.comm, &ex1, &by1);
  tot_t = R.reduce(tot_t, 2);
  double fn = 0.0;
  R.ck(mrl_norm2(R.ctx, dF[cur], (int64_t)(9 * npl), &fn), "mrl_norm2");
  // one more solve with per-kernel event timing: the exposed arrival waits per CG iteration (after the checksum: not part of it)
  double waits_ms = 0.0;
  long prof_its = 0;
  if (R.slab)
  {
    R.ck(mrl_set_profiling(R.ctx, 1), "mrl_set_profiling");
    mrl_mech_stats pst{};
    solve(substeps + 1, pst);
    prof_its = pst.cg_its_total;
    for (const KernelRow & k : read_profile(R.ctx))
      if (k.name == "slab_exchange_wait")
        waits_ms += k.ms;
    R.ck(mrl_set_profiling(R.ctx, 0), "mrl_set_profiling");
  }
  char describe[2048] = "{}";
  if (R.comm)
    mrl_comm_describe(R.comm, describe, sizeof describe);
  if (R.rank == 0)
  {
    const double npts = (double)nx * ny * nz;
    const double h = 8.0 * (1.0 + 2.0 / (double)nz);
    const double bpi = 2 * 9 * (8.0 + 5.0 * h) + 232 + 504;
    const double ms = tot_t / std::max(1l, tot_its) * 1e3;
    const double gbps = bpi * npts * tot_its / tot_t / 1e9 / R.world;
    std::ostringstream o;
    o << "{\"metric\": \"grid-point CG-iteration updates/sec, de Geus finite-strain RVE Newton-CG (fp64)\", \"value\": " << jnum(npts * tot_its / tot_t)
      << ", \"unit\": \"grid-point-CG-iterations/s\", \"n_gpus\": " << R.world << ", \"steps\": " << tot_its << ", \"warmup\": " << warmup
      << ", \"ms_per_step\": " << jnum(ms) << ", \"higher_is_better\": true, \"scaling\": " << (G ? "\"strong\"" : "\"weak\"")
      << ", \"vs_baseline\": null, \"dtype\": \"f64\", \"data\": \"synthetic (cubic inclusion RVE)\", \"config\": {\"workload\": \"de Geus finite-strain hyperelastic RVE "
      << nx << "x" << ny << "x" << nz << ", Newton-CG with FFT-applied Gamma operator\", \"grid\": [" << nx << ", " << ny << ", " << nz << "], \"decomposition\": "
      << (R.slab ? jstr("slab x" + std::to_string(R.world) + ", library-owned exchange (mrl_comm)") : jstr("none"))
      << ", \"driver\": \"native (marlin-hip-bench)\", \"newton_iterations_per_substep\": [" << newton.str() << "], \"cg_iterations\": " << tot_its
      << ", \"transport\": " << jstr(selected) << "}, \"algorithmic_bytes_per_point_per_cg_iteration\": " << jnum(bpi) << ", \"model_GBps_per_gpu\": " << jnum(gbps)
      << ", \"model_frac_of_hbm_peak\": " << jnum(gbps / HBM_PEAK) << ", \"model_frac_of_copy_ceiling\": " << jnum(gbps / HBM_COPY)
      << ", \"roofline\": {\"bound\": \"hbm\", \"kernel\": \"one CG iteration (all kernels; SURVEY 8(d) byte model)\", \"achieved\": " << jnum(gbps) << ", \"peak\": " << HBM_PEAK
      << ", \"unit\": \"GB/s\", \"frac\": " << jnum(gbps / HBM_PEAK) << ", \"traffic\": null}";
    if (R.slab)
      o << ", \"exchange\": {\"ranks\": " << R.world << ", \"devices_per_rank\": " << comm_devices_json(R.comm) << ", \"distinct_devices\": "
        << comm_distinct_devices(R.comm) << ", \"transport\": {\"selected\": " << jstr(selected) << ", \"tuned\": " << tuned.str()
        << "}, \"exchanges_per_step\": " << jnum((double)(ex1 - ex0) / std::max(1l, tot_its)) << ", \"bytes_sent_to_peers_per_step_rank0\": "
        << jnum((by1 - by0) / std::max(1l, tot_its)) << ", \"link_GBps_out_rank0\": " << jnum((by1 - by0) / tot_t / 1e9)
        << ", \"exposed_wait_ms_per_step\": " << jnum(prof_its ? waits_ms / prof_its : 0.0) << ", \"runtime\": " << describe << "}";
    o << ", \"field_checksum\": {\"norm_F\": " << jnum(fn) << "}}";
    std::printf("%s\n", o.str().c_str());
    std::fflush(stdout);
  }
  return 0;
}

// a profiler (or any tool library) that is loaded into this process before main() and initialises the GPU there
static const char * profiler_preload()
{
  static const char * vars[] = {"LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_REGISTER_FORCE_LOAD", nullptr};
  for (int i = 0; vars[i]; ++i)
  {
    const char * v = std::getenv(vars[i]);
    if (v && (std::strstr(v, "rocprof") || std::strstr(v, "roctracer") || std::strstr(v, "rocprofiler")))
      return vars[i];
  }
  return nullptr;
}

// start one child per rank (fork + exec of this binary; the parent has made no HIP call) and return the worst exit code
static int launch_ranks(int argc, char ** argv, int nranks)
{
  if (const char * why = profiler_preload())
  {
    // the tool library has initialised the GPU before main(): an exec from here takes the machine down on this pool
    std::fprintf(stderr,
                 "marlin-hip-bench: refusing to fork + exec %d rank processes under a profiler preload (%s is set): the preloaded tool "
                 "has initialised the GPU in this process.  Profile ONE rank directly: rocprofv3 ... -- marlin-hip-bench gpus=%d rank=r "
                 "job=<name> ... (start the other ranks unprofiled with the same job=), or run gpus=1.\n",
                 nranks, why, nranks);
    return 2;
  }
  const std::string job = "job=mrlbench_" + std::to_string((long)getpid());
  std::vector<pid_t> kids;
  for (int r = 0; r < nranks; ++r)
  {
    const pid_t pid = fork();
    if (pid < 0)
    {
      std::perror("fork");
      return 1;
    }
    if (pid == 0)
    {
      const std::string rk = "rank=" + std::to_string(r);
      std::vector<char *> av(argv, argv + argc);
      av.push_back(const_cast<char *>(rk.c_str()));
      av.push_back(const_cast<char *>(job.c_str()));
      av.push_back(nullptr);
      execv("/proc/self/exe", av.data());
      std::perror("execv");
      _exit(127);
    }
    kids.push_back(pid);
  }
  // reap in any order; once a rank has failed the others can only run into their communicator time-outs (120 s): give them a grace
  // period, then end exactly the processes started here (by pid) so that the caller's fallback starts without that wait
  const double grace_s = (double)argi("launch_grace_s", 20);
  int rc = 0;
  double first_fail = -1.0;
  bool killed = false;
  std::vector<pid_t> left = kids;
  while (!left.empty())
  {
    int st = 0;
    const pid_t p = waitpid(-1, &st, WNOHANG);
    if (p > 0)
    {
      const auto it = std::find(left.begin(), left.end(), p);
      if (it == left.end())
        continue;
      left.erase(it);
      if (!WIFEXITED(st) || WEXITSTATUS(st) != 0)
      {
        rc = 1;
        if (first_fail < 0)
          first_fail = now_s();
      }
      continue;
    }
    if (p < 0 && errno != EINTR)
    {
      rc = 1;
      break;
    }
    if (first_fail >= 0 && !killed && now_s() - first_fail > grace_s)
    {
      for (const pid_t k : left)
        kill(k, SIGKILL);
      killed = true;
    }
    usleep(20000);
  }
  shm_unlink(("/mrlbench_" + std::to_string((long)getpid())).c_str());  // (left behind only if the ranks died before the bootstrap)
  return rc;
}

int main(int argc, char ** argv)
{
  for (int i = 1; i < argc; ++i)
  {
    const std::string a = argv[i];
    const auto eq = a.find('=');
    if (eq == std::string::npos)
    {
      std::fprintf(stderr, "expected key=value, got '%s'\n", a.c_str());
      return 2;
    }
    g_args[a.substr(0, eq)] = a.substr(eq + 1);
  }
  const int gpus = (int)argi("gpus", 1);
  if (gpus < 1 || gpus > 64)
  {
    std::fprintf(stderr, "1 <= gpus <= 64\n");
    return 2;
  }
  if (!g_args.count("rank") && gpus > 1)
    return launch_ranks(argc, argv, gpus);  // nothing has touched the GPU yet
  // gpus == 1 without rank=: this process IS the rank -- no fork, no exec (safe under rocprofv3, whose preload initialises the GPU
  // before main)
  if (!g_args.count("job"))
    g_args["job"] = "mrlbench_" + std::to_string((long)getpid());

  Rank R;
  R.world = gpus;
  R.rank = g_rank = (int)argi("rank", 0);
  R.slab = gpus > 1 || argi("slab", 0) != 0;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    die(R.rank, "no HIP device available");
  // host-local rank -> device (DomainAction.C:163-199); device=d puts every rank on GPU d (single-GPU functional runs)
  R.device = g_args.count("device") ? (int)argi("device", 0) : R.rank % ndev;
  HIPCK(hipSetDevice(R.device));
  if (R.slab)
  {
    const std::string tr = arg("transport", "tune");
    const int want = (tr == "tune" || tr == "auto") ? MRL_TRANSPORT_AUTO : std::atoi(tr.c_str());
    if (mrl_comm_create(&R.comm, arg("job", "mrlbench").c_str(), R.world, R.rank, R.device, want) != MRL_OK)
      die(R.rank, std::string("mrl_comm_create: ") + mrl_comm_last_error(nullptr));
    mrl_comm_set_timeout(R.comm, 120.0);
    if (g_args.count("test_die_rank") && argi("test_die_rank", -1) == R.rank)
      _exit(3);  // (test hook: a rank lost after the bootstrap -- the launcher must not wait for the survivors' time-outs)
  }
  const std::string workload = arg("workload", "ch");
  int rc = 2;
  if (workload == "ch")
    rc = run_ch(R);
  else if (workload == "mech")
    rc = run_mech(R);
  else
    std::fprintf(stderr, "workload=ch|mech\n");
  if (R.ctx)
    mrl_ctx_destroy(R.ctx);
  if (R.comm)
    mrl_comm_destroy(R.comm);
  return rc;
}
