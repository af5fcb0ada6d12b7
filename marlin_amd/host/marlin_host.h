// Host-side mirror (C++17) of the part of Marlin's MOOSE-object API that drives the hot path, over the
// C ABI of include/marlin_hip.h.  Same names, argument meaning and error behaviour as the reference:
//   DeviceTensor                    torch::Tensor handle semantics (assignment rebinds, copies share storage)
//   TensorBuffer                    include/tensor_buffers/TensorBuffer.h:16-116 (history by handle copy)
//   TensorProblem                   src/problems/TensorProblem.C:154-197, 451-472 (advanceState rule, sub time)
//   DomainAction                    src/actions/DomainAction.C (fft / ifft / average / axes)
//   TensorOperatorBase, ComputeGroup        include/tensor_computes/TensorOperatorBase.h:27-171, ComputeGroup.C:50-87
//   TensorSolver                    src/tensor_solver/TensorSolver.C:86-109
//   AdamsBashforthMoulton           src/tensor_solver/AdamsBashforthMoulton.C:48-101  (Cahn-Hilliard system, fused)
//   ForwardEulerSolver              src/tensor_solver/ForwardEulerSolver.C:29-38
//   MacroscopicShearTensor          test/src/tensor_computes/MacroscopicShearTensor.C:31-41
//   FFTMechanics                    src/tensor_computes/FFTMechanics.C:96-163 (+ HyperElasticIsotropic)
//   ParsedCompute, ForwardFFT, ReciprocalLaplacianFactor    src/tensor_computes/{ParsedCompute,PerformFFT,ReciprocalLaplacianFactor}.C
//   SplitOperatorABM                src/tensor_solver/AdamsBashforthMoulton.C:60-178 for any number of variables, with
//                                   the Adams-Moulton corrector (compute group evaluated operator by operator)
//   TensorExtremeValuePostprocessor, TensorIntegralPostprocessor    src/postprocessors/*.C
//   Transient                       MOOSE's executioner loop as far as the path needs it
// The MOOSE factory / input parser is NOT mirrored: objects are constructed directly (marlin_hip_run.cpp).
// All arithmetic happens in libmarlin_hip.so; this layer only owns handles, history and control flow.
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <functional>
#include <map>
#include <sstream>
#include <thread>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "marlin_hip.h"

namespace marlin_host {

[[noreturn]] inline void mooseError(const std::string & msg) { throw std::runtime_error(msg); }
[[noreturn]] inline void paramError(const std::string & param, const std::string & msg)
{
  throw std::runtime_error(param + ": " + msg);
}

/// shared handle to a device array of doubles (complex spectra are interleaved re,im)
class DeviceTensor
{
public:
  DeviceTensor() = default;
  static DeviceTensor empty(std::size_t count)
  {
    DeviceTensor t;
    double * p = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&p), sizeof(double) * (count ? count : 1)) != hipSuccess)
      mooseError("hipMalloc failed");
    t._p = std::shared_ptr<double>(p, [](double * q) { (void)hipFree(q); });
    t._n = count;
    return t;
  }
  static DeviceTensor zeros(std::size_t count)
  {
    auto t = empty(count);
    if (hipMemset(t.data(), 0, sizeof(double) * count) != hipSuccess)
      mooseError("hipMemset failed");
    return t;
  }
  static DeviceTensor fromHost(const std::vector<double> & h)
  {
    auto t = empty(h.size());
    if (hipMemcpy(t.data(), h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice) != hipSuccess)
      mooseError("hipMemcpy (host to device) failed");
    return t;
  }
  std::vector<double> toHost() const
  {
    std::vector<double> h(_n);
    if (hipMemcpy(h.data(), _p.get(), sizeof(double) * _n, hipMemcpyDeviceToHost) != hipSuccess)
      mooseError("hipMemcpy (device to host) failed");
    return h;
  }
  bool defined() const { return (bool)_p; }
  double * data() const { return _p.get(); }
  std::size_t numel() const { return _n; }

private:
  std::shared_ptr<double> _p;
  std::size_t _n = 0;
};

/// TensorBuffer<T>: current tensor + history of handles
class TensorBuffer
{
public:
  DeviceTensor & getTensor() { return _u; }
  std::size_t advanceState()
  {
    if (_u_old.size() < _max_states)
      _u_old.resize(_u_old.size() + 1);
    if (!_u_old.empty())
    {
      for (std::size_t i = _u_old.size() - 1; i > 0; --i)
        _u_old[i] = _u_old[i - 1];
      _u_old[0] = _u;
    }
    return _u_old.size();
  }
  const std::vector<DeviceTensor> & getOldTensor(std::size_t states_requested)
  {
    _max_states = std::max(_max_states, states_requested);
    return _u_old;
  }
  void clearStates() { _u_old.clear(); }
  std::size_t maxStates() const { return _max_states; }
  /// rebind the current tensor and the whole history at once (a solver that ran several substeps in one library call)
  void setStates(const DeviceTensor & u, const std::vector<DeviceTensor> & old)
  {
    _u = u;
    _u_old = old;
  }

private:
  DeviceTensor _u;
  std::vector<DeviceTensor> _u_old;
  std::size_t _max_states = 0;
};

/// [Domain] parallel_mode and the placement of this process in the job
enum class DomainParallelMode
{
  NONE,
  FFT_SLAB,
  FFT_PENCIL  ///< 3-D, nranks = py * pz (DomainAction::partitionPencils): transforms, reductions and pointwise computes
};
struct DomainParallel
{
  DomainParallelMode mode = DomainParallelMode::NONE;
  int nranks = 1, rank = 0;
  std::string job;  ///< identifies the job on the node (the MOOSE shim broadcasts one string over MPI)
  int transport = MRL_TRANSPORT_AUTO;
  int device = -1;  ///< HIP device of this rank (-1: the current one)
  /// the Cahn-Hilliard solver's spectral arrays (Mbarmubar states, cbar) dense [nx][ny][nzc] instead of the library's private layout
  /// (mrl_ch_spec_elems: x planes padded on fused fast-path grids): needed when another object reads them
  bool dense_spectra = false;
};

/// DomainAction as a math service.  parallel_mode = FFT_SLAB (DomainAction.C:510-566): this process is rank `rank` of `nranks`
/// (one process per GPU); the domain owns the library's communicator (mrl_comm_*: HIP IPC peer stores / copy engines / RCCL in
/// place of the reference's host-staged MPI transposes, DomainAction.C:869-1019) and every buffer is the rank's LOCAL block:
/// real space split along y, reciprocal space along x.
class DomainAction
{
public:
  using ParallelMode = DomainParallelMode;
  using Parallel = DomainParallel;
  DomainAction(int dim, const std::vector<int64_t> & n, const std::vector<double> & max,
               const std::vector<double> & min = {0, 0, 0}, const Parallel & par = Parallel())
    : _dim(dim), _n(n), _par(par)
  {
    mrl_domain d{};
    d.dim = dim;
    for (int i = 0; i < dim; ++i)
    {
      d.n[i] = n[i];
      d.min[i] = i < (int)min.size() ? min[i] : 0.0;
      d.max[i] = max[i];
      _dx.push_back((d.max[i] - d.min[i]) / n[i]);
      _min.push_back(d.min[i]);
      _max.push_back(d.max[i]);
    }
    d.device = par.device;
    d.nranks = 1;
    d.rank = 0;
    d.spectrum = MRL_SPECTRUM_HALF;
    d.stream = nullptr;  // the HIP null stream: hipMemcpy/hipMemset of this layer are ordered with the kernels
    d.flags = par.dense_spectra ? MRL_FLAG_DENSE_SPECTRA : 0;
    const bool slab = par.mode == ParallelMode::FFT_SLAB;
    if (slab)
    {
      if (dim < 2)
        mooseError("Dimension must be 2 or 3 for slab decomposition.");  // DomainAction.C:514-515
      d.nranks = par.nranks;
      d.rank = par.rank;
      d.flags |= MRL_FLAG_SLAB;
      // 3-D keeps the r2c transform along z (half the exchange volume of the reference's c2c, identical fields);
      // 2-D uses the reference's full c2c layout (DomainAction.C:279-281)
      d.spectrum = dim == 3 ? MRL_SPECTRUM_HALF : MRL_SPECTRUM_FULL;
      if (mrl_comm_create(&_comm, par.job.c_str(), par.nranks, par.rank, par.device, par.transport) != MRL_OK)
        mooseError(std::string("DomainAction: ") + mrl_comm_last_error(nullptr));
    }
    const bool pencil = par.mode == ParallelMode::FFT_PENCIL;
    if (pencil)
    {
      if (dim < 3)
        mooseError("Dimension must be 3 for pencil decomposition.");  // DomainAction.C:571-572
      d.nranks = par.nranks;
      d.rank = par.rank;
      d.flags |= MRL_FLAG_PENCIL;  // r2c along x, kx split over py, ky over pz (DomainAction.C:282-284, 620-698)
      int32_t py = 0, pz = 0;
      if (mrl_pencil_factors(par.nranks, d.n, &py, &pz) != MRL_OK)
        paramError("parallel_mode", mrl_last_error(nullptr));  // DomainAction.C:611-616
      if (mrl_comm_create(&_comm, par.job.c_str(), par.nranks, par.rank, par.device, par.transport) != MRL_OK)
        mooseError(std::string("DomainAction: ") + mrl_comm_last_error(nullptr));
    }
    if (mrl_ctx_create(&_ctx, &d) != MRL_OK)
      mooseError(std::string("DomainAction: ") + mrl_last_error(nullptr));
    if (slab || pencil)
      check(mrl_ctx_attach_comm(_ctx, _comm));
    int64_t rn[3], rb[3], kn[3], kb[3];
    check(mrl_local_shape(_ctx, rn, rb, kn, kb));
    _n_global = 1;
    _n_real = 1;
    _n_recip = 1;
    for (int i = 0; i < dim; ++i)
    {
      _n_global *= n[i];
      _n_real *= rn[i];
      _n_recip *= kn[i];
      _local_shape.push_back(rn[i]);
      _local_begin.push_back(rb[i]);
      _recip_shape.push_back(kn[i]);
      _recip_begin.push_back(kb[i]);
    }
    // the Cahn-Hilliard solver's own spectral arrays (history ring) may carry a padded last-axis pitch on slab contexts
    _n_recip_solver = mrl_ch_spec_elems(_ctx);
  }
  ~DomainAction()
  {
    mrl_ctx_destroy(_ctx);
    mrl_comm_destroy(_comm);
  }
  DomainAction(const DomainAction &) = delete;

  mrl_ctx * ctx() const { return _ctx; }
  mrl_comm * comm() const { return _comm; }
  bool isSlab() const { return _par.mode == ParallelMode::FFT_SLAB; }
  bool isPencil() const { return _par.mode == ParallelMode::FFT_PENCIL; }
  int rank() const { return _par.rank; }
  int nranks() const { return _par.nranks; }
  int getDim() const { return _dim; }
  const std::vector<int64_t> & getShape() const { return _n; }                 ///< global grid
  const std::vector<int64_t> & getLocalShape() const { return _local_shape; }  ///< DomainAction::getLocalShape
  const std::vector<int64_t> & getLocalBegin() const { return _local_begin; }
  const std::vector<int64_t> & getReciprocalShape() const { return _recip_shape; }
  /// cells of this rank's real-space block (= the global count in parallel_mode NONE): the size of every real buffer
  int64_t getNumberOfCells() const { return _n_real; }
  int64_t getGlobalNumberOfCells() const { return _n_global; }
  double getExtent(int d) const { return _max[d] - _min[d]; }
  /// points of this rank's reciprocal block: the size (in complex values) of every reciprocal buffer
  int64_t getReciprocalSize() const { return _n_recip; }
  /// ... of the arrays private to the fused Cahn-Hilliard solver (mrl_slab_ch_spec_pitch)
  int64_t getSolverReciprocalSize() const { return _n_recip_solver; }
  /// real-space axis: linspace(min + dx/2, max - dx/2, n)  (DomainAction.C:246-251), global
  std::vector<double> getAxis(int d) const
  {
    std::vector<double> a(_n[d]);
    const double lo = _min[d] + _dx[d] / 2.0, hi = _max[d] - _dx[d] / 2.0;
    const double step = _n[d] > 1 ? (hi - lo) / (double)(_n[d] - 1) : 0.0;
    // torch::linspace fills the upper half from the end (start + step*i below the midpoint, end - step*(n-1-i) above)
    for (int64_t i = 0; i < _n[d]; ++i)
      a[i] = (i < _n[d] / 2) ? lo + step * (double)i : hi - step * (double)(_n[d] - 1 - i);
    return a;
  }
  /// reciprocal axis d (2 pi fftfreq / rfftfreq, DomainAction.C:259-303) as the library holds it (the local part)
  std::vector<double> getReciprocalAxis(int d) const
  {
    const int64_t n = _recip_shape[d];
    std::vector<double> k((std::size_t)n);
    check(mrl_ctx_reciprocal_axis(_ctx, d, k.data(), n));
    return k;
  }
  void check(int rc) const
  {
    if (rc != MRL_OK)
      mooseError(std::string("marlin_hip: ") + mrl_last_error(_ctx));
  }
  DeviceTensor fft(const DeviceTensor & t, int64_t batch = 1) const
  {
    auto out = DeviceTensor::empty(2 * _n_recip * batch);
    check(mrl_fft_r2c(_ctx, t.data(), out.data(), batch, batch > 1));
    return out;
  }
  DeviceTensor ifft(const DeviceTensor & t, int64_t batch = 1) const
  {
    auto out = DeviceTensor::empty(_n_real * batch);
    check(mrl_fft_c2r(_ctx, t.data(), out.data(), batch, batch > 1));
    return out;
  }
  std::vector<double> average(const DeviceTensor & t, int ncomp) const
  {
    std::vector<double> a(ncomp);
    check(mrl_average(_ctx, t.data(), ncomp, a.data()));
    return a;
  }

private:
  const int _dim;
  std::vector<int64_t> _n;
  const Parallel _par;
  std::vector<double> _dx, _min, _max;
  mrl_ctx * _ctx = nullptr;
  mrl_comm * _comm = nullptr;
  std::vector<int64_t> _local_shape, _local_begin, _recip_shape, _recip_begin;
  int64_t _n_real, _n_recip, _n_recip_solver, _n_global;
};

/// the part of TensorProblem the solvers talk to: buffer registry, time bookkeeping, history advance
class TensorProblem
{
public:
  explicit TensorProblem(DomainAction & domain) : _domain(domain) {}
  DomainAction & domain() { return _domain; }
  TensorBuffer & getBufferObject(const std::string & name) { return _tensor_buffer[name]; }
  DeviceTensor & getBuffer(const std::string & name) { return _tensor_buffer[name].getTensor(); }
  const std::vector<DeviceTensor> & getBufferOld(const std::string & name, unsigned int max_states)
  {
    return _tensor_buffer[name].getOldTensor(max_states);
  }
  /// TensorProblem::advanceState (TensorProblem.C:451-472): a no-op while timeStep() <= 1
  void advanceState()
  {
    if (_t_step <= 1)
      return;
    for (auto & pair : _tensor_buffer)
      pair.second.advanceState();
  }
  /// true if `name` is the only buffer whose history anything asked for
  bool onlyHistoryOf(const std::string & name) const
  {
    for (const auto & pair : _tensor_buffer)
      if (pair.second.maxStates() > 0 && pair.first != name)
        return false;
    return true;
  }
  int & timeStep() { return _t_step; }
  double & time() { return _time; }
  double & timeOld() { return _time_old; }
  double & dt() { return _dt; }
  double & dtOld() { return _dt_old; }  ///< FEProblem::dtOld(): the previous time step's dt (TensorSolver.C:48)
  double & subDt() { return _sub_dt; }
  double & subTime() { return _sub_time; }

private:
  DomainAction & _domain;
  std::map<std::string, TensorBuffer> _tensor_buffer;
  int _t_step = 0;
  double _time = 0.0, _time_old = 0.0, _dt = 0.0, _dt_old = 0.0, _sub_dt = 0.0, _sub_time = 0.0;
};

class TensorOperatorBase
{
public:
  TensorOperatorBase(TensorProblem & problem, std::string name)
    : _tensor_problem(problem), _domain(problem.domain()), _name(std::move(name)), _time(problem.subTime())
  {
  }
  virtual ~TensorOperatorBase() = default;
  virtual void computeBuffer() = 0;
  virtual void init() {}
  virtual bool supportsJIT() const { return false; }
  const std::string & name() const { return _name; }

protected:
  DeviceTensor & getInputBuffer(const std::string & buffer) { return _tensor_problem.getBuffer(buffer); }
  DeviceTensor & getOutputBuffer(const std::string & buffer) { return _tensor_problem.getBuffer(buffer); }
  TensorProblem & _tensor_problem;
  DomainAction & _domain;
  const std::string _name;
  const double & _time;  // TensorOperatorBase.C:40: operators see the sub time
};

/// ComputeGroup: runs its computes in the given (already dependency-sorted) order
class ComputeGroup : public TensorOperatorBase
{
public:
  using TensorOperatorBase::TensorOperatorBase;
  void add(std::shared_ptr<TensorOperatorBase> cmp) { _computes.push_back(std::move(cmp)); }
  void computeBuffer() override
  {
    _compute_count++;                                                                // ComputeGroup.C:53
    for (auto & cmp : _computes)
    {
      try
      {
        cmp->computeBuffer();
      }
      catch (const std::exception & e)
      {
        mooseError("Exception in compute '" + cmp->name() + "': " + e.what());  // ComputeGroup.C:73-83
      }
    }
  }

  unsigned int getComputeCount() const { return _compute_count; }                   // ComputeGroupExecutionCount

private:
  std::vector<std::shared_ptr<TensorOperatorBase>> _computes;
  unsigned int _compute_count = 0;
};

class TensorSolver : public TensorOperatorBase
{
public:
  TensorSolver(TensorProblem & problem, std::string name, unsigned int substeps,
               std::shared_ptr<TensorOperatorBase> root_compute)
    : TensorOperatorBase(problem, std::move(name)), _substeps(substeps), _sub_dt(problem.subDt()),
      _sub_time(problem.subTime()), _dt(problem.dt()), _dt_old(problem.dtOld()), _compute(std::move(root_compute))
  {
  }
  /// TensorSolver::computeBuffer (TensorSolver.C:93-109)
  void computeBuffer() override
  {
    _sub_dt = _dt / _substeps;
    for (_substep = 0; _substep < _substeps; _substep++)
    {
      substep();
      if (_substep < _substeps - 1)
        _tensor_problem.advanceState();
      _sub_time += _sub_dt;
    }
  }
  void addForwardBuffer(const std::string & forward_buffer, const std::string & forward_buffer_new)
  {
    _forwarded.emplace_back(forward_buffer, forward_buffer_new);
  }

protected:
  void forwardBuffers()
  {
    for (auto & [dst, src] : _forwarded)
      _tensor_problem.getBuffer(dst) = _tensor_problem.getBuffer(src);  // handle assignment (TensorSolver.C:86-90)
  }
  virtual void substep() = 0;
  const unsigned int _substeps;
  unsigned int _substep = 0;
  double & _sub_dt;
  double & _sub_time;
  const double & _dt;
  const double & _dt_old;
  /// "If dt changes between steps, we start at first order again" (AdamsBashforthMoulton.C:75, AdamsBashforthMoultonCoupled.C:110)
  bool dtChanged() const { return _dt != _dt_old; }
  std::shared_ptr<TensorOperatorBase> _compute;
  std::vector<std::pair<std::string, std::string>> _forwarded;
};

/// AdamsBashforthMoulton for the Cahn-Hilliard system: variable {buffer=c, reciprocal_buffer=cbar,
/// linear_reciprocal=kappabarbar, nonlinear_reciprocal=Mbarmubar}; the root compute group (mu, mubar,
/// Mbarmubar, cbar) is fused into the substep kernel sequence, its buffers are still published.
class AdamsBashforthMoulton : public TensorSolver
{
public:
  struct Params
  {
    std::string buffer = "c", reciprocal_buffer = "cbar", nonlinear_reciprocal = "Mbarmubar", mu = "mu";
    unsigned int substeps = 1;
    std::size_t predictor_order = 2;
    mrl_ch_params ch{};
    bool publish_mu = true, publish_cbar = false;
    /// opt-in: cbar of a substep = ubar of the previous one (MRL_CARRY_*), valid while `buffer` is only written by this solver
    bool spectral_carry = false;
    /// one library call per substep (the reference's loop, operator by operator) instead of one per computeBuffer
    bool substep_calls = false;
  };
  AdamsBashforthMoulton(TensorProblem & problem, const std::string & name, const Params & p)
    : TensorSolver(problem, name, p.substeps, nullptr), _p(p), _predictor_order(p.predictor_order - 1),
      _u(problem.getBuffer(p.buffer)), _nonlinear(problem.getBuffer(p.nonlinear_reciprocal)),
      _old_nonlinear(problem.getBufferOld(p.nonlinear_reciprocal, (unsigned int)(p.predictor_order - 1)))
  {
    if (p.predictor_order < 1 || p.predictor_order > 5)
      paramError("predictor_order", "predictor_order > 0 & predictor_order <= 5");
    if (p.spectral_carry && p.publish_cbar)
      paramError("spectral_carry", "cbar is not materialised separately when it is carried over");
    if (p.publish_cbar && _domain.getSolverReciprocalSize() != _domain.getReciprocalSize())
      paramError("publish_cbar", "cbar would be published in the solver's private (padded) layout: create the Domain with dense_spectra = true");
    if (_domain.isSlab())
    {
      if (p.publish_cbar)
        paramError("publish_cbar", "not available with parallel_mode = FFT_SLAB");
      // the carried spectrum lives inside the library's slab pipeline (one mrl_ch_substeps call = one computeBuffer)
      _domain.check(mrl_ctx_set_option(_domain.ctx(), MRL_OPT_SLAB_CARRY, p.spectral_carry ? 1 : 0));
    }
  }

  /// TensorSolver::computeBuffer (TensorSolver.C:93-109).  When nothing else in the problem keeps a history and no per-substep
  /// output is requested, the whole substep loop is ONE library call (mrl_ch_substeps): the history ring it rotates is handed
  /// back to the TensorBuffer afterwards, so that the next advanceState sees exactly the reference's handles.
  void computeBuffer() override
  {
    const bool slab = _domain.isSlab();
    if ((!slab && (_substeps < 2 || _p.spectral_carry)) || _p.publish_cbar || _p.substep_calls ||
        !_tensor_problem.onlyHistoryOf(_p.nonlinear_reciprocal))
      return TensorSolver::computeBuffer();
    _sub_dt = _dt / _substeps;
    const std::size_t nreal = _domain.getNumberOfCells(), nspec = 2 * _domain.getSolverReciprocalSize();
    const int size = (int)_predictor_order + 1;   // history depth + the array being written
    // ring slot (size-1-i) = Nhat_old[i]; the remaining slots are scratch arrays that hold no live state
    std::vector<DeviceTensor> ring(size);
    const int n_old0 = (int)std::min(_old_nonlinear.size(), _predictor_order);
    for (int i = 0; i < n_old0; ++i)
      ring[size - 1 - i] = _old_nonlinear[i];
    for (int j = 0; j < size - n_old0; ++j)
    {
      // reuse an array of the pool that is not part of the history
      for (auto & cand : _pool)
      {
        bool used = false;
        for (int q = 0; q < size; ++q)
          used = used || (ring[q].defined() && ring[q].data() == cand.data());
        if (!used)
        {
          ring[j] = cand;
          break;
        }
      }
      if (!ring[j].defined())
      {
        ring[j] = DeviceTensor::empty(nspec);
        _pool.push_back(ring[j]);
      }
    }
    std::vector<double *> ptr(size);
    for (int q = 0; q < size; ++q)
      ptr[q] = ring[q].data();
    int head = size - 1, n_old = n_old0;
    auto c_out = DeviceTensor::empty(nreal);
    DeviceTensor mu;
    if (_p.publish_mu)
      mu = DeviceTensor::empty(nreal);
    // TensorProblem::advanceState is a no-op while timeStep() <= 1; a changed dt restarts the order (AdamsBashforthMoulton.C:75,88-91)
    const int advance = (_tensor_problem.timeStep() > 1 ? MRL_SUBSTEPS_ADVANCE : 0) | (dtChanged() ? MRL_SUBSTEPS_DT_CHANGED : 0);
    _domain.check(mrl_ch_substeps(_domain.ctx(), &_p.ch, _u.data(), c_out.data(), ptr.data(), size, &head, &n_old,
                                  (int)_predictor_order + 1, (int)_substeps, advance, _sub_dt, mu.defined() ? mu.data() : nullptr));
    std::vector<DeviceTensor> old;
    for (int i = 0; i < n_old; ++i)
      old.push_back(ring[((head - i) % size + size) % size]);
    _tensor_problem.getBufferObject(_p.nonlinear_reciprocal).setStates(ring[(head + 1) % size], old);
    if (mu.defined())
      _tensor_problem.getBuffer(_p.mu) = mu;
    _u = c_out;
    _sub_time += _sub_dt * _substeps;
  }

protected:
  void substep() override
  {
    const std::size_t n_old = _old_nonlinear.size();
    const int order = (int)std::min(_substep < _predictor_order && dtChanged() ? (std::size_t)0 : n_old, _predictor_order);  // AdamsBashforthMoulton.C:88-91
    const std::size_t nreal = _domain.getNumberOfCells(), nspec = 2 * _domain.getSolverReciprocalSize();
    auto c_out = DeviceTensor::empty(nreal);
    auto Nnew = DeviceTensor::empty(nspec);
    DeviceTensor mu, cbar;
    if (_p.publish_mu)
      mu = DeviceTensor::empty(nreal);
    if (_p.publish_cbar)
      cbar = DeviceTensor::empty(nspec);
    std::vector<const double *> old(order > 0 ? order : 1, nullptr);
    for (int i = 0; i < order; ++i)
      old[i] = _old_nonlinear[i].data();
    int carry = MRL_CARRY_NONE;
    if (_p.spectral_carry && !_domain.isSlab())
    {
      // the carried spectrum belongs to the tensor this solver published last; any other writer rebinds the handle
      const bool valid = _carry.defined() && _u.data() == _last_c;
      if (!valid)
        _carry = DeviceTensor::empty(nspec);
      carry = valid ? MRL_CARRY_IN : MRL_CARRY_OUT;
    }
    _domain.check(mrl_ch_substep(_domain.ctx(), &_p.ch, _u.data(), c_out.data(), Nnew.data(), old.data(), order, _sub_dt,
                                 carry != MRL_CARRY_NONE ? _carry.data() : (cbar.defined() ? cbar.data() : nullptr),
                                 mu.defined() ? mu.data() : nullptr, carry));
    _last_c = c_out.data();
    _nonlinear = Nnew;  // what the compute group assigns to Mbarmubar
    if (mu.defined())
      _tensor_problem.getBuffer(_p.mu) = mu;
    if (cbar.defined())
      _tensor_problem.getBuffer(_p.reciprocal_buffer) = cbar;
    _u = c_out;  // AdamsBashforthMoulton.C:101
  }
  const Params _p;
  const std::size_t _predictor_order;
  DeviceTensor & _u;
  DeviceTensor & _nonlinear;
  const std::vector<DeviceTensor> & _old_nonlinear;
  DeviceTensor _carry;
  const double * _last_c = nullptr;
  std::vector<DeviceTensor> _pool;   // Nhat arrays of the multi-substep path that are not (yet / any more) in the history
};

/// ForwardEulerSolver (src/tensor_solver/ForwardEulerSolver.C:27-38, ExplicitSolverBase.C:33-51): root compute, forward
/// buffers, then u = ifft(reciprocal_buffer + sub_dt * time_derivative_reciprocal) per variable (none for the mechanics driver)
class ForwardEulerSolver : public TensorSolver
{
public:
  struct VariableNames
  {
    std::string buffer, reciprocal_buffer, time_derivative_reciprocal;
  };
  ForwardEulerSolver(TensorProblem & problem, const std::string & name, unsigned int substeps,
                     std::shared_ptr<TensorOperatorBase> root_compute, const std::vector<VariableNames> & vars = {})
    : TensorSolver(problem, name, substeps, std::move(root_compute))
  {
    for (const auto & v : vars)
      _variables.push_back(Variable{problem.getBuffer(v.buffer), problem.getBuffer(v.reciprocal_buffer),
                                    problem.getBuffer(v.time_derivative_reciprocal)});
  }

protected:
  struct Variable
  {
    DeviceTensor & _buffer;
    const DeviceTensor & _reciprocal_buffer;
    const DeviceTensor & _time_derivative_reciprocal;
  };
  void substep() override
  {
    _compute->computeBuffer();
    forwardBuffers();
    for (auto & v : _variables)
    {
      auto ubar = DeviceTensor::empty(v._reciprocal_buffer.numel());
      const double * N[] = {v._time_derivative_reciprocal.data()};
      const double coef[] = {_sub_dt};
      _domain.check(mrl_kspace_abm(_domain.ctx(), ubar.data(), v._reciprocal_buffer.data(), N, coef, 1, nullptr, _sub_dt,
                                   _domain.getReciprocalSize()));
      v._buffer = _domain.ifft(ubar);
    }
  }
  std::vector<Variable> _variables;
};

/// test object: DbarF = (I + t e0 x e1) - <F>
class MacroscopicShearTensor : public TensorOperatorBase
{
public:
  MacroscopicShearTensor(TensorProblem & problem, const std::string & name, const std::string & buffer,
                         const std::string & F)
    : TensorOperatorBase(problem, name), _u(getOutputBuffer(buffer)), _tF(getInputBuffer(F))
  {
  }
  void computeBuffer() override
  {
    const int d = _domain.getDim();
    const auto avg = _domain.average(_tF, d * d);
    std::vector<double> a(d * d, 0.0);
    for (int i = 0; i < d; ++i)
      a[i * d + i] = 1.0;
    a[1] = a[1] + _time;
    for (int i = 0; i < d * d; ++i)
      a[i] = a[i] - avg[i];
    _u = DeviceTensor::fromHost(a);
  }

private:
  DeviceTensor & _u;
  DeviceTensor & _tF;
};

/// FFTMechanics with the HyperElasticIsotropic constitutive model
class FFTMechanics : public TensorOperatorBase
{
public:
  struct Params
  {
    std::string buffer = "Fnew", F = "F", K = "K", mu = "mu", stress = "stress", applied_macroscopic_strain;
    double l_tol = 1e-2, nl_rel_tol = 1e-5, nl_abs_tol = 1e-8;
    int64_t l_max_its = 0;
    unsigned int nl_max_its = 100;
  };
  FFTMechanics(TensorProblem & problem, const std::string & name, const Params & p)
    : TensorOperatorBase(problem, name), _p(p), _u(getOutputBuffer(p.buffer)), _tF(getInputBuffer(p.F)),
      _tK(getInputBuffer(p.K)), _tmu(getInputBuffer(p.mu)), _tP(getOutputBuffer(p.stress)),
      _applied(p.applied_macroscopic_strain.empty() ? nullptr : &getInputBuffer(p.applied_macroscopic_strain))
  {
  }
  void computeBuffer() override
  {
    mrl_mech_params prm{_p.l_tol, _p.l_max_its, _p.nl_rel_tol, _p.nl_abs_tol, (int32_t)_p.nl_max_its};
    auto Fnew = DeviceTensor::empty(_tF.numel());
    auto P = DeviceTensor::empty(_tF.numel());
    const int rc = mrl_mech_newton_cg(_domain.ctx(), &prm, _tF.data(), _tK.data(), _tmu.data(),
                                      _applied ? _applied->data() : nullptr, Fnew.data(), P.data(), &_stats);
    if (rc == MRL_ERR_NOT_CONVERGED)
      paramError("nl_max_its", mrl_last_error(_domain.ctx()));  // FFTMechanics.C:159-161
    _domain.check(rc);
    _u = Fnew;
    _tP = P;
  }
  const mrl_mech_stats & stats() const { return _stats; }

private:
  const Params _p;
  DeviceTensor & _u;
  DeviceTensor & _tF;
  DeviceTensor & _tK;
  DeviceTensor & _tmu;
  DeviceTensor & _tP;
  DeviceTensor * _applied;
  mrl_mech_stats _stats{};
};


/// ComputeDisplacements (src/tensor_computes/ComputeDisplacements.C:53-107): node displacements [(n+1)..., dim] of F
class ComputeDisplacements : public TensorOperatorBase
{
public:
  ComputeDisplacements(TensorProblem & problem, const std::string & name, const std::string & buffer, const std::string & F)
    : TensorOperatorBase(problem, name), _u(getOutputBuffer(buffer)), _F(getInputBuffer(F))
  {
  }
  void computeBuffer() override
  {
    if (!_F.defined())
      return;
    std::size_t nodes = 1;
    for (int d = 0; d < _domain.getDim(); ++d)
      nodes *= (std::size_t)_domain.getShape()[d] + 1;
    auto out = DeviceTensor::empty(nodes * _domain.getDim());
    _domain.check(mrl_mech_displacements(_domain.ctx(), _F.data(), out.data()));
    _u = out;
  }

private:
  DeviceTensor & _u;
  DeviceTensor & _F;
};

/// FFTQuasistaticElasticity (src/tensor_computes/FFTQuasistaticElasticity.C:30-104): homogeneous small-strain equilibrium with the
/// volumetric eigenstrain e0*c; writes one displacement buffer per mesh dimension
class FFTQuasistaticElasticity : public TensorOperatorBase
{
public:
  FFTQuasistaticElasticity(TensorProblem & problem, const std::string & name, const std::vector<std::string> & displacements,
                           const std::string & cbar, double mu, double lambda, double e0)
    : TensorOperatorBase(problem, name), _mu(mu), _lambda(lambda), _e0(e0), _cbar(getInputBuffer(cbar))
  {
    for (const auto & d : displacements)
      _displacements.push_back(&getOutputBuffer(d));
    if ((std::size_t)_domain.getDim() != _displacements.size())
      paramError("displacements", "Need one displacement variable per mesh dimension");
  }
  void computeBuffer() override
  {
    double * out[3] = {nullptr, nullptr, nullptr};
    std::vector<DeviceTensor> fresh;
    for (std::size_t i = 0; i < _displacements.size(); ++i)
    {
      fresh.push_back(DeviceTensor::empty(_domain.getNumberOfCells()));
      out[i] = fresh.back().data();
    }
    _domain.check(mrl_qs_elasticity(_domain.ctx(), _cbar.data(), _mu, _lambda, _e0, out));
    for (std::size_t i = 0; i < _displacements.size(); ++i)
      *_displacements[i] = fresh[i];
  }

private:
  std::vector<DeviceTensor *> _displacements;
  const double _mu, _lambda, _e0;
  DeviceTensor & _cbar;
};

/// FFTElasticChemicalPotential (src/tensor_computes/FFTElasticChemicalPotential.C:29-61): reciprocal-space elastic contribution to
/// the chemical potential from cbar and the displacement buffers
class FFTElasticChemicalPotential : public TensorOperatorBase
{
public:
  FFTElasticChemicalPotential(TensorProblem & problem, const std::string & name, const std::string & buffer,
                              const std::vector<std::string> & displacements, const std::string & cbar, double mu, double lambda, double e0)
    : TensorOperatorBase(problem, name), _u(getOutputBuffer(buffer)), _mu(mu), _lambda(lambda), _e0(e0), _cbar(getInputBuffer(cbar))
  {
    for (const auto & d : displacements)
      _displacements.push_back(&getInputBuffer(d));
    if ((std::size_t)_domain.getDim() != _displacements.size())
      paramError("displacements", "Need one displacement variable per mesh dimension");
  }
  void computeBuffer() override
  {
    const double * in[3] = {nullptr, nullptr, nullptr};
    for (std::size_t i = 0; i < _displacements.size(); ++i)
      in[i] = _displacements[i]->data();
    auto out = DeviceTensor::empty(2 * _domain.getReciprocalSize());
    _domain.check(mrl_elastic_chemical_potential(_domain.ctx(), _cbar.data(), in, _mu, _lambda, _e0, out.data()));
    _u = out;
  }

private:
  DeviceTensor & _u;
  std::vector<DeviceTensor *> _displacements;
  const double _mu, _lambda, _e0;
  DeviceTensor & _cbar;
};

/// ComputeVonMisesStress (src/tensor_computes/ComputeVonMisesStress.C:31-66)
class ComputeVonMisesStress : public TensorOperatorBase
{
public:
  ComputeVonMisesStress(TensorProblem & problem, const std::string & name, const std::string & buffer,
                        const std::string & stress = "stress")
    : TensorOperatorBase(problem, name), _u(getOutputBuffer(buffer)), _stress(getInputBuffer(stress))
  {
  }
  void computeBuffer() override
  {
    if (!_stress.defined())
      return;
    auto out = DeviceTensor::empty(_domain.getNumberOfCells());
    _domain.check(mrl_mech_von_mises(_domain.ctx(), _stress.data(), out.data()));
    _u = out;
  }

private:
  DeviceTensor & _u;
  DeviceTensor & _stress;
};

/// ParsedCompute: pointwise expression of input buffers (and x, y, z, kx, ky, kz, k2, t with extra_symbols)
class ParsedCompute : public TensorOperatorBase
{
public:
  struct Params
  {
    std::string buffer, expression;
    std::vector<std::string> inputs, complex_inputs, derivatives;
    std::vector<std::pair<std::string, double>> constants;
    bool extra_symbols = false, reciprocal = false;
  };
  ParsedCompute(TensorProblem & problem, const std::string & name, const Params & p)
    : TensorOperatorBase(problem, name), _p(p), _u(getOutputBuffer(p.buffer))
  {
    std::vector<const char *> in, cn, dn;
    std::vector<int> cplx;
    std::vector<double> cv;
    for (auto & n : _p.inputs)
    {
      in.push_back(n.c_str());
      cplx.push_back(std::count(_p.complex_inputs.begin(), _p.complex_inputs.end(), n) ? 1 : 0);
      _params.push_back(&getInputBuffer(n));
    }
    for (auto & c : _p.constants)
    {
      cn.push_back(c.first.c_str());
      cv.push_back(c.second);
    }
    for (auto & d : _p.derivatives)
      dn.push_back(d.c_str());
    if (mrl_parsed_create(_domain.ctx(), &_parsed, _p.expression.c_str(), (int)in.size(), in.data(), cplx.data(), (int)cn.size(),
                          cn.data(), cv.data(), (int)dn.size(), dn.data(), _p.extra_symbols, _p.reciprocal) != MRL_OK)
      paramError("expression", mrl_last_error(_domain.ctx()));
  }
  ~ParsedCompute() { mrl_parsed_destroy(_parsed); }
  void computeBuffer() override
  {
    const int64_t count = _p.reciprocal ? _domain.getReciprocalSize() : _domain.getNumberOfCells();
    auto out = DeviceTensor::empty(count * (mrl_parsed_is_complex(_parsed) ? 2 : 1));
    std::vector<const double *> in;
    for (auto * t : _params)
      in.push_back(t->data());
    _domain.check(mrl_parsed_eval(_parsed, in.data(), out.data(), count, _time));
    _u = out;
  }

private:
  const Params _p;
  DeviceTensor & _u;
  std::vector<DeviceTensor *> _params;
  mrl_parsed * _parsed = nullptr;
};

/// PerformFFT<true>: _u = _domain.fft(_input)   (PerformFFT.C:34-40)
class ForwardFFT : public TensorOperatorBase
{
public:
  ForwardFFT(TensorProblem & problem, const std::string & name, const std::string & buffer, const std::string & input)
    : TensorOperatorBase(problem, name), _u(getOutputBuffer(buffer)), _input(getInputBuffer(input))
  {
  }
  void computeBuffer() override { _u = _domain.fft(_input); }

private:
  DeviceTensor & _u;
  DeviceTensor & _input;
};

/// PerformFFT<false>: _u = _domain.ifft(_input)   (PerformFFT.C:34-40)
class InverseFFT : public TensorOperatorBase
{
public:
  InverseFFT(TensorProblem & problem, const std::string & name, const std::string & buffer, const std::string & input)
    : TensorOperatorBase(problem, name), _u(getOutputBuffer(buffer)), _input(getInputBuffer(input))
  {
  }
  void computeBuffer() override { _u = _domain.ifft(_input); }

private:
  DeviceTensor & _u;
  DeviceTensor & _input;
};

/// FFTGradient::computeBuffer (src/tensor_computes/FFTGradient.C:36-40): _u = ifft(fft(input) * k_direction * i),
/// the k-space product as one fused parsed kernel
class FFTGradient : public TensorOperatorBase
{
public:
  FFTGradient(TensorProblem & problem, const std::string & name, const std::string & buffer, const std::string & input,
              int direction)
    : TensorOperatorBase(problem, name), _u(getOutputBuffer(buffer)), _input(getInputBuffer(input))
  {
    static const char * k[] = {"kx", "ky", "kz"};
    const std::string expr = std::string("abar*") + k[direction] + "*i";
    const char * in[] = {"abar"};
    const int cplx[] = {1};
    if (mrl_parsed_create(_domain.ctx(), &_parsed, expr.c_str(), 1, in, cplx, 0, nullptr, nullptr, 0, nullptr, 1, 1) != MRL_OK)
      paramError("direction", mrl_last_error(_domain.ctx()));
  }
  ~FFTGradient() { mrl_parsed_destroy(_parsed); }
  void computeBuffer() override
  {
    const auto abar = _domain.fft(_input);
    auto gbar = DeviceTensor::empty(abar.numel());
    const double * in[] = {abar.data()};
    _domain.check(mrl_parsed_eval(_parsed, in, gbar.data(), _domain.getReciprocalSize(), 0.0));
    _u = _domain.ifft(gbar);
  }

private:
  DeviceTensor & _u;
  DeviceTensor & _input;
  mrl_parsed * _parsed = nullptr;
};


/// FFTGradientSquare::computeBuffer (src/tensor_computes/FFTGradientSquare.C:37-50): _u = factor * sum_d ifft(fft(input) k_d i)^2;
/// one forward transform, per axis one generated k-space kernel + inverse transform, one generated kernel for the squares
class FFTGradientSquare : public TensorOperatorBase
{
public:
  FFTGradientSquare(TensorProblem & problem, const std::string & name, const std::string & buffer, const std::string & input,
                    double factor = 1.0, bool input_is_reciprocal = false)
    : TensorOperatorBase(problem, name), _u(getOutputBuffer(buffer)), _input(getInputBuffer(input)),
      _input_is_reciprocal(input_is_reciprocal)
  {
    static const char * k[] = {"kx", "ky", "kz"};
    static const char * g[] = {"gx", "gy", "gz"};
    const int dim = _domain.getDim();
    const char * in[] = {"abar"};
    const int cplx[] = {1};
    std::string sq;
    for (int d = 0; d < dim; ++d)
    {
      mrl_parsed * p = nullptr;
      const std::string expr = std::string("abar*") + k[d] + "*i";
      if (mrl_parsed_create(_domain.ctx(), &p, expr.c_str(), 1, in, cplx, 0, nullptr, nullptr, 0, nullptr, 1, 1) != MRL_OK)
        paramError("input", mrl_last_error(_domain.ctx()));
      _grad.push_back(p);
      sq += std::string(d ? "+" : "") + g[d] + "*" + g[d];
    }
    if (factor != 1.0)
      sq = "(" + sq + ")*f";
    const int real[] = {0, 0, 0};
    const char * cn[] = {"f"};
    if (mrl_parsed_create(_domain.ctx(), &_square, sq.c_str(), dim, g, real, 1, cn, &factor, 0, nullptr, 0, 0) != MRL_OK)
      paramError("factor", mrl_last_error(_domain.ctx()));
  }
  ~FFTGradientSquare()
  {
    for (auto * p : _grad)
      mrl_parsed_destroy(p);
    mrl_parsed_destroy(_square);
  }
  void computeBuffer() override
  {
    const auto r = _input_is_reciprocal ? _input : _domain.fft(_input);
    std::vector<DeviceTensor> g;
    std::vector<const double *> gp;
    for (auto * p : _grad)
    {
      auto gbar = DeviceTensor::empty(r.numel());
      const double * in[] = {r.data()};
      _domain.check(mrl_parsed_eval(p, in, gbar.data(), _domain.getReciprocalSize(), 0.0));
      g.push_back(_domain.ifft(gbar));
      gp.push_back(g.back().data());
    }
    auto out = DeviceTensor::empty(_domain.getNumberOfCells());
    _domain.check(mrl_parsed_eval(_square, gp.data(), out.data(), _domain.getNumberOfCells(), 0.0));
    _u = out;
  }

private:
  DeviceTensor & _u;
  DeviceTensor & _input;
  const bool _input_is_reciprocal;
  std::vector<mrl_parsed *> _grad;
  mrl_parsed * _square = nullptr;
};

/// ReciprocalLaplacianFactor (-k^2 f) / ReciprocalLaplacianSquareFactor (k^4 f)
class ReciprocalLaplacianFactor : public TensorOperatorBase
{
public:
  ReciprocalLaplacianFactor(TensorProblem & problem, const std::string & name, const std::string & buffer, double factor,
                            int power = 1)
    : TensorOperatorBase(problem, name), _u(getOutputBuffer(buffer)), _factor(factor), _power(power)
  {
  }
  void computeBuffer() override
  {
    auto out = DeviceTensor::empty(_domain.getReciprocalSize());
    _domain.check(mrl_reciprocal_laplacian(_domain.ctx(), _power, _factor, out.data()));
    _u = out;
  }

private:
  DeviceTensor & _u;
  const double _factor;
  const int _power;
};

/// AdamsBashforthMoulton for any number of split-operator variables, with the Adams-Moulton corrector
class SplitOperatorABM : public TensorSolver
{
public:
  struct VariableNames
  {
    std::string buffer, reciprocal_buffer, linear_reciprocal /* "0" = none */, nonlinear_reciprocal;
  };
  SplitOperatorABM(TensorProblem & problem, const std::string & name, unsigned int substeps,
                   std::shared_ptr<TensorOperatorBase> root_compute, const std::vector<VariableNames> & vars,
                   std::size_t predictor_order, std::size_t corrector_order, std::size_t corrector_steps)
    : TensorSolver(problem, name, substeps, std::move(root_compute)), _predictor_order(predictor_order - 1),
      _corrector_order(corrector_order - 1), _corrector_steps(corrector_steps)
  {
    if (predictor_order < 1 || predictor_order > 5)
      paramError("predictor_order", "predictor_order > 0 & predictor_order <= 5");
    if (corrector_order < 1 || corrector_order > 5)
      paramError("corrector_order", "corrector_order > 0 & corrector_order <= 5");
    const auto history = (unsigned int)std::max(_predictor_order, _corrector_order);  // AdamsBashforthMoulton.C:55-56
    for (const auto & v : vars)
      _variables.push_back(Variable{problem.getBuffer(v.buffer), problem.getBuffer(v.reciprocal_buffer),
                                    v.linear_reciprocal == "0" ? nullptr : &problem.getBuffer(v.linear_reciprocal),
                                    problem.getBuffer(v.nonlinear_reciprocal),
                                    problem.getBufferOld(v.nonlinear_reciprocal, history)});
  }

protected:
  struct Variable  // SplitOperatorBase.h:27-34
  {
    DeviceTensor & _buffer;
    const DeviceTensor & _reciprocal_buffer;
    const DeviceTensor * _linear_reciprocal;
    const DeviceTensor & _nonlinear_reciprocal;
    const std::vector<DeviceTensor> & _old_nonlinear_reciprocal;
  };

  static double abBeta(std::size_t order, std::size_t i)  // AdamsBashforthMoulton.C:67-73 (incl. the AB5 190/720 entry)
  {
    static const double beta[5][5] = {{1.0, 0.0, 0.0, 0.0, 0.0},
                                      {3.0 / 2.0, -1.0 / 2.0, 0.0, 0.0, 0.0},
                                      {23.0 / 12.0, -16.0 / 12.0, 5.0 / 12.0, 0.0, 0.0},
                                      {55.0 / 24.0, -59.0 / 24.0, 37.0 / 24.0, -9.0 / 24.0, 0.0},
                                      {190.0 / 720.0, -2774.0 / 720.0, 2616.0 / 720.0, -1274.0 / 720.0, 251.0 / 720.0}};
    return beta[order][i];
  }
  static double amAlpha(std::size_t order, std::size_t i)  // :108-114
  {
    static const double alpha[5][5] = {{1.0, 0.0, 0.0, 0.0, 0.0},
                                       {0.5, 0.5, 0.0, 0.0, 0.0},
                                       {5.0 / 12.0, 8.0 / 12.0, -1.0 / 12.0, 0.0, 0.0},
                                       {9.0 / 24.0, 19.0 / 24.0, -5.0 / 24.0, 1.0 / 24.0, 0.0},
                                       {251.0 / 720.0, 646.0 / 720.0, -264.0 / 720.0, 106.0 / 720.0, -19.0 / 720.0}};
    return alpha[order][i];
  }

  /// u = ifft( (ubar0 + sum coef_i N_i) / (1 - dt L) )
  void update(Variable & v, const DeviceTensor & ubar0, const std::vector<const double *> & N, const std::vector<double> & coef)
  {
    auto ubar = DeviceTensor::empty(ubar0.numel());
    _domain.check(mrl_kspace_abm(_domain.ctx(), ubar.data(), ubar0.data(), N.data(), coef.data(), (int)N.size(),
                                 v._linear_reciprocal ? v._linear_reciprocal->data() : nullptr, _sub_dt,
                                 _domain.getReciprocalSize()));
    v._buffer = _domain.ifft(ubar);
  }

  void substep() override
  {
    _compute->computeBuffer();
    forwardBuffers();
    // Adams-Bashforth predictor on all variables (constant dt)                       AdamsBashforthMoulton.C:80-102
    for (auto & v : _variables)
    {
      const std::size_t n_old = v._old_nonlinear_reciprocal.size();
      const std::size_t order = std::min(_substep < _predictor_order && dtChanged() ? (std::size_t)0 : n_old, _predictor_order);
      std::vector<const double *> N{v._nonlinear_reciprocal.data()};
      std::vector<double> coef{_sub_dt * abBeta(order, 0)};
      for (std::size_t i = 0; i < order; ++i)
      {
        N.push_back(v._old_nonlinear_reciprocal[i].data());
        coef.push_back(_sub_dt * abBeta(order, i + 1));
      }
      update(v, v._reciprocal_buffer, N, coef);
    }
    if (_corrector_steps)                                                             // :117-177
    {
      _sub_time += _sub_dt;
      std::vector<DeviceTensor> ubar_n, N_n;  // handle copies keep the step-n tensors alive
      for (auto & v : _variables)
      {
        ubar_n.push_back(v._reciprocal_buffer);
        N_n.push_back(v._nonlinear_reciprocal);
      }
      for (std::size_t j = 0; j < _corrector_steps; ++j)
      {
        _compute->computeBuffer();
        forwardBuffers();
        for (std::size_t k = 0; k < _variables.size(); ++k)
        {
          auto & v = _variables[k];
          const std::size_t n_old = v._old_nonlinear_reciprocal.size();
          const std::size_t order = std::min(_substep < _corrector_order && dtChanged() ? (std::size_t)1 : n_old + 1, _corrector_order);
          if (order == 0)
            continue;
          std::vector<const double *> N{v._nonlinear_reciprocal.data(), N_n[k].data()};
          std::vector<double> coef{_sub_dt * amAlpha(order, 0), _sub_dt * amAlpha(order, 1)};
          for (std::size_t i = 0; i + 1 < order; ++i)
          {
            N.push_back(v._old_nonlinear_reciprocal[i].data());
            coef.push_back(_sub_dt * amAlpha(order, i + 2));
          }
          update(v, ubar_n[k], N, coef);
        }
      }
      _sub_time -= _sub_dt;
    }
  }

  const std::size_t _predictor_order, _corrector_order, _corrector_steps;
  std::vector<Variable> _variables;
};


/// AdamsBashforthMoultonCoupled: ABM right-hand sides for all variables + one dense per-k solve with off-diagonal
/// linear operators (AdamsBashforthMoultonCoupled.C:84-272).  `flags` = 0 reproduces the reference (and its gold files),
/// see mrl_kspace_coupled.
class AdamsBashforthMoultonCoupled : public SplitOperatorABM
{
public:
  struct OffDiagonal
  {
    unsigned int row, col;
    std::string buffer;
  };
  AdamsBashforthMoultonCoupled(TensorProblem & problem, const std::string & name, unsigned int substeps,
                               std::shared_ptr<TensorOperatorBase> root_compute, const std::vector<VariableNames> & vars,
                               const std::vector<OffDiagonal> & offdiag, bool assume_symmetric, std::size_t predictor_order,
                               std::size_t corrector_order, std::size_t corrector_steps, int flags = 0)
    : SplitOperatorABM(problem, name, substeps, std::move(root_compute), vars, predictor_order, corrector_order, corrector_steps),
      _flags(flags)
  {
    const std::size_t N = _variables.size();
    if (N > 32)
      paramError("buffer", "at most 32 coupled variables");
    _L.assign(N * N, nullptr);
    for (std::size_t i = 0; i < N; ++i)
      _L[i * N + i] = _variables[i]._linear_reciprocal;
    for (const auto & o : offdiag)
    {
      if (o.row >= N)
        paramError("linear_offdiag_rows", "Off-diagonal indices out of range.");
      if (o.col >= N)
        paramError("linear_offdiag_cols", "Off-diagonal indices out of range.");
      _L[o.row * N + o.col] = &problem.getBuffer(o.buffer);
    }
    if (assume_symmetric)                                                            // :153-156
      for (const auto & o : offdiag)
        if (o.row != o.col && !_L[o.col * N + o.row])
          _L[o.col * N + o.row] = &problem.getBuffer(o.buffer);
  }

protected:
  void solve(const std::vector<const double *> & ubar0, const std::vector<std::vector<const double *>> & N,
             const std::vector<std::vector<double>> & coef)
  {
    const std::size_t nv = _variables.size();
    std::vector<DeviceTensor> ubar;
    std::vector<double *> out;
    std::vector<const double *> flatN, L;
    std::vector<double> flatc;
    std::vector<int> nterms;
    for (std::size_t i = 0; i < nv; ++i)
    {
      ubar.push_back(DeviceTensor::empty(2 * _domain.getReciprocalSize()));
      out.push_back(ubar.back().data());
      nterms.push_back((int)N[i].size());
      flatN.insert(flatN.end(), N[i].begin(), N[i].end());
      flatc.insert(flatc.end(), coef[i].begin(), coef[i].end());
    }
    for (auto * l : _L)
      L.push_back(l ? l->data() : nullptr);
    _domain.check(mrl_kspace_coupled(_domain.ctx(), (int)nv, out.data(), ubar0.data(), flatN.data(), flatc.data(), nterms.data(),
                                     L.data(), _sub_dt, _flags, _domain.getReciprocalSize()));
    for (std::size_t i = 0; i < nv; ++i)
      _variables[i]._buffer = _domain.ifft(ubar[i]);
  }

  void substep() override
  {
    _compute->computeBuffer();
    forwardBuffers();
    const std::size_t nv = _variables.size();
    std::vector<const double *> u0(nv);
    std::vector<std::vector<const double *>> N(nv);
    std::vector<std::vector<double>> coef(nv);
    for (std::size_t k = 0; k < nv; ++k)                                              // :118-138
    {
      auto & v = _variables[k];
      const std::size_t order =
          std::min(_substep < _predictor_order && dtChanged() ? (std::size_t)0 : v._old_nonlinear_reciprocal.size(), _predictor_order);
      u0[k] = v._reciprocal_buffer.data();
      N[k] = {v._nonlinear_reciprocal.data()};
      coef[k] = {_sub_dt * abBeta(order, 0)};
      for (std::size_t i = 0; i < order; ++i)
      {
        N[k].push_back(v._old_nonlinear_reciprocal[i].data());
        coef[k].push_back(_sub_dt * abBeta(order, i + 1));
      }
    }
    solve(u0, N, coef);
    if (!_corrector_steps)
      return;
    std::vector<DeviceTensor> ubar_n, N_n;                                            // :198-211
    for (auto & v : _variables)
    {
      ubar_n.push_back(v._reciprocal_buffer);
      N_n.push_back(v._nonlinear_reciprocal);
    }
    for (std::size_t j = 0; j < _corrector_steps; ++j)
    {
      _compute->computeBuffer();
      forwardBuffers();
      for (std::size_t k = 0; k < nv; ++k)
      {
        auto & v = _variables[k];
        const std::size_t order =
            std::min(_substep < _corrector_order && dtChanged() ? (std::size_t)1 : v._old_nonlinear_reciprocal.size() + 1, _corrector_order);
        u0[k] = ubar_n[k].data();
        N[k].clear();
        coef[k].clear();
        if (order == 0)                                                               // rhs = ubar_n, still solved (:225-229)
          continue;
        N[k] = {v._nonlinear_reciprocal.data(), N_n[k].data()};
        coef[k] = {_sub_dt * amAlpha(order, 0), _sub_dt * amAlpha(order, 1)};
        for (std::size_t i = 0; i + 1 < order; ++i)
        {
          N[k].push_back(v._old_nonlinear_reciprocal[i].data());
          coef[k].push_back(_sub_dt * amAlpha(order, i + 2));
        }
      }
      solve(u0, N, coef);
    }
  }

  const int _flags;
  std::vector<const DeviceTensor *> _L;
};


/// IterativeTensorSolverInterface (include/tensor_solver/IterativeTensorSolverInterface.h): what the time stepper queries
/// TensorPredictor (src/tensor_predictor/TensorPredictor.C:15-41): forward-predicts a solver output buffer from its old states
class TensorPredictor
{
public:
  TensorPredictor(TensorProblem & problem, const std::string & buffer, unsigned int history_size = 1)
    : _domain(problem.domain()), _u(problem.getBuffer(buffer)), _u_old(problem.getBufferOld(buffer, history_size))
  {
  }
  virtual ~TensorPredictor() = default;
  virtual void computeBuffer() = 0;

protected:
  DomainAction & _domain;
  DeviceTensor & _u;
  const std::vector<DeviceTensor> & _u_old;
};

/// LinearTensorPredictor (src/tensor_predictor/LinearTensorPredictor.C:19-46): u += scale * (u_old[0] - u_old[1]), history_size 2
class LinearTensorPredictor : public TensorPredictor
{
public:
  LinearTensorPredictor(TensorProblem & problem, const std::string & buffer, double scale = 1.0)
    : TensorPredictor(problem, buffer, 2), _scale(scale)
  {
  }
  void computeBuffer() override
  {
    if (_u_old.size() > 1)
    {
      const int64_t n = (int64_t)_u.numel();
      auto diff = DeviceTensor::empty(_u.numel());
      _domain.check(mrl_axpby(_domain.ctx(), 1.0, _u_old[0].data(), -1.0, _u_old[1].data(), diff.data(), n));   // :38
      auto out = DeviceTensor::empty(_u.numel());
      // :39-42 (diff * 1.0 is exact, so the two branches of the reference coincide)
      _domain.check(mrl_axpby(_domain.ctx(), 1.0, _u.data(), _scale, diff.data(), out.data(), n));
      _u = out;
    }
  }

private:
  const double _scale;
};

/// IterativeTensorSolverInterface (src/tensor_solver/IterativeTensorSolverInterface.C:12-31)
class IterativeTensorSolverInterface
{
public:
  const unsigned int & getIterations() const { return _iterations; }
  const bool & isConverged() const { return _is_converged; }
  void addPredictor(std::shared_ptr<TensorPredictor> predictor) { _predictors.push_back(std::move(predictor)); }

protected:
  void applyPredictors()
  {
    for (const auto & pred : _predictors)
      pred->computeBuffer();
  }
  unsigned int _iterations = 0;
  bool _is_converged = true;
  std::vector<std::shared_ptr<TensorPredictor>> _predictors;
};

/// SwiftHohenbergLinear: _u = r - alpha^2 (1 - k^2)^2   (SwiftHohenbergLinear.C:35-39) as one generated kernel
class SwiftHohenbergLinear : public TensorOperatorBase
{
public:
  SwiftHohenbergLinear(TensorProblem & problem, const std::string & name, const std::string & buffer, double r, double alpha)
    : TensorOperatorBase(problem, name), _u(getOutputBuffer(buffer))
  {
    const char * cn[] = {"r", "aa"};
    const double cv[] = {r, alpha * alpha};
    if (mrl_parsed_create(_domain.ctx(), &_parsed, "r-aa*(1-k2)*(1-k2)", 0, nullptr, nullptr, 2, cn, cv, 0, nullptr, 1, 1) != MRL_OK)
      paramError("buffer", mrl_last_error(_domain.ctx()));
  }
  ~SwiftHohenbergLinear() { mrl_parsed_destroy(_parsed); }
  void computeBuffer() override
  {
    auto out = DeviceTensor::empty(_domain.getReciprocalSize());
    _domain.check(mrl_parsed_eval(_parsed, nullptr, out.data(), _domain.getReciprocalSize(), _time));
    _u = out;
  }

private:
  DeviceTensor & _u;
  mrl_parsed * _parsed = nullptr;
};


/// DeAliasingTensor (src/tensor_computes/DeAliasingTensor.C:37-65) as one generated kernel on the reciprocal grid
class DeAliasingTensor : public TensorOperatorBase
{
public:
  DeAliasingTensor(TensorProblem & problem, const std::string & name, const std::string & buffer, const std::string & method,
                   double p = 16.0, double alpha = 36.0)
    : TensorOperatorBase(problem, name), _u(getOutputBuffer(buffer))
  {
    double mx[3] = {0.0, 0.0, 0.0};  // maximum |k| per axis; absent axes are {0}
    for (int d = 0; d < _domain.getDim(); ++d)
      for (double k : _domain.getReciprocalAxis(d))
        mx[d] = std::max(mx[d], std::fabs(k));
    std::string expr;
    std::vector<const char *> cn;
    std::vector<double> cv;
    if (method == "SHARP")
    {
      expr = "if((abs(kx) > cx) | (abs(ky) > cy) | (abs(kz) > cz), 0, 1)";
      cn = {"cx", "cy", "cz"};
      cv = {2 * mx[0] / 3, 2 * mx[1] / 3, 2 * mx[2] / 3};
    }
    else if (method == "HOULI")
    {
      expr = "exp(0-alpha*((abs(kx)/mx)^p + (abs(ky)/my)^p + (abs(kz)/mz)^p))";
      cn = {"alpha", "p", "mx", "my", "mz"};
      cv = {alpha, p, mx[0] ? mx[0] : 1.0, mx[1] ? mx[1] : 1.0, mx[2] ? mx[2] : 1.0};
    }
    else
      paramError("method", "SHARP or HOULI");
    if (mrl_parsed_create(_domain.ctx(), &_parsed, expr.c_str(), 0, nullptr, nullptr, (int)cn.size(), cn.data(), cv.data(), 0, nullptr,
                          1, 1) != MRL_OK)
      paramError("method", mrl_last_error(_domain.ctx()));
  }
  ~DeAliasingTensor() { mrl_parsed_destroy(_parsed); }
  void computeBuffer() override
  {
    auto out = DeviceTensor::empty(_domain.getReciprocalSize());
    _domain.check(mrl_parsed_eval(_parsed, nullptr, out.data(), _domain.getReciprocalSize(), _time));
    _u = out;
  }

private:
  DeviceTensor & _u;
  mrl_parsed * _parsed = nullptr;
};


/// SmoothRectangleCompute (src/tensor_computes/SmoothRectangleCompute.C:58-121): inside / outside value of an axis-aligned box
/// with a sharp, cosine or tanh profile; one generated kernel (clamp(v, lo, hi) = min(max(v, lo), hi) as ATen evaluates it)
class SmoothRectangleCompute : public TensorOperatorBase
{
public:
  struct Params
  {
    std::string buffer, profile = "COS";
    double x1 = 0, x2 = 0, y1 = 0, y2 = 0, z1 = 0, z2 = 0, int_width = 0, inside = 1, outside = 0;
  };
  SmoothRectangleCompute(TensorProblem & problem, const std::string & name, const Params & p)
    : TensorOperatorBase(problem, name), _u(getOutputBuffer(p.buffer))
  {
    if (p.int_width < 0.0)
      mooseError("Interface width must be a non-negative real number.");
    const int dim = _domain.getDim();
    static const char * ax[] = {"x", "y", "z"};
    std::string h;
    if (p.int_width <= 0.0)
    {
      std::string cond;
      for (int d = 0; d < dim; ++d)
        cond += std::string(d ? " & " : "") + "(" + ax[d] + " >= " + ax[d] + "1) & (" + ax[d] + " <= " + ax[d] + "2)";
      h = "if(" + cond + ", 1, 0)";
    }
    else
      for (int d = 0; d < dim; ++d)   // the factors of the absent axes are exactly 1 (sin(pi/2), tanh(>= 20))
      {
        const std::string m = std::string("min(") + ax[d] + " - " + ax[d] + "1, " + ax[d] + "2 - " + ax[d] + ")";
        const std::string f = p.profile == "TANH" ? "(0.5 + 0.5 * tanh(4 * " + m + " / w))"
                                                  : "(0.5 + 0.5 * sin(pi * min(max(" + m + ", 0 - w / 2), w / 2) / w))";
        h += (d ? " * " : "") + f;
      }
    const std::string expr = "h := " + h + "; h * inside + (1 - h) * outside";
    const char * cn[] = {"x1", "x2", "y1", "y2", "z1", "z2", "w", "inside", "outside"};
    const double cv[] = {p.x1, p.x2, p.y1, p.y2, p.z1, p.z2, p.int_width, p.inside, p.outside};
    if (mrl_parsed_create(_domain.ctx(), &_parsed, expr.c_str(), 0, nullptr, nullptr, 9, cn, cv, 0, nullptr, 1, 0) != MRL_OK)
      paramError("profile", mrl_last_error(_domain.ctx()));
  }
  ~SmoothRectangleCompute() { mrl_parsed_destroy(_parsed); }
  void computeBuffer() override
  {
    auto out = DeviceTensor::empty(_domain.getNumberOfCells());
    _domain.check(mrl_parsed_eval(_parsed, nullptr, out.data(), _domain.getNumberOfCells(), _time));
    _u = out;
  }

private:
  DeviceTensor & _u;
  mrl_parsed * _parsed = nullptr;
};

/// SecantSolver (src/tensor_solver/SecantSolver.C:42-185): control flow of the reference, k-space work in two fused kernels
class SecantSolver : public SplitOperatorABM, public IterativeTensorSolverInterface
{
public:
  struct Params
  {
    unsigned int substeps = 1, max_iterations = 30;
    double relative_tolerance = 1e-9, absolute_tolerance = 1e-9, damping = 1.0, dt_epsilon = 1e-4;
    bool verbose = false;
  };
  SecantSolver(TensorProblem & problem, const std::string & name, std::shared_ptr<TensorOperatorBase> root_compute,
               const std::vector<VariableNames> & vars, const Params & p)
    : SplitOperatorABM(problem, name, p.substeps, std::move(root_compute), vars, 1, 1, 0), _p(p)  // getVariables(0): no history
  {
  }

protected:
  void substep() override
  {
    const std::size_t n = _variables.size();
    const int64_t ns = _domain.getReciprocalSize();
    std::vector<DeviceTensor> u_old(n), Rprev(n), uprev(n);
    std::vector<double> R0norm(n);
    _compute->computeBuffer();                                                       // :73
    forwardBuffers();
    for (std::size_t i = 0; i < n; ++i)
    {
      auto & v = _variables[i];
      Rprev[i] = DeviceTensor::empty(2 * ns);
      auto guess = DeviceTensor::empty(2 * ns);
      double ss = 0.0;
      _domain.check(mrl_secant_begin(_domain.ctx(), v._reciprocal_buffer.data(), v._nonlinear_reciprocal.data(),
                                     v._linear_reciprocal ? v._linear_reciprocal->data() : nullptr, _sub_dt, _p.dt_epsilon,
                                     Rprev[i].data(), guess.data(), &ss, ns));
      R0norm[i] = std::sqrt(ss);
      uprev[i] = v._reciprocal_buffer;                                               // handle copies (:85,:90)
      u_old[i] = v._reciprocal_buffer;
      v._buffer = _domain.ifft(guess);
      if (_p.verbose)
        std::printf("|R0|=%g\n", R0norm[i]);
    }
    applyPredictors();                                                               // :100 (on solver outputs)
    bool all_converged = false;
    for (_iterations = 0; _iterations < _p.max_iterations; ++_iterations)           // :112-165
    {
      _compute->computeBuffer();
      forwardBuffers();
      all_converged = true;
      for (std::size_t i = 0; i < n; ++i)
      {
        auto & v = _variables[i];
        auto unew = DeviceTensor::empty(2 * ns);
        double ss[2];
        _domain.check(mrl_secant_iterate(_domain.ctx(), v._reciprocal_buffer.data(), v._nonlinear_reciprocal.data(),
                                         v._linear_reciprocal ? v._linear_reciprocal->data() : nullptr, u_old[i].data(),
                                         uprev[i].data(), Rprev[i].data(), _sub_dt, _p.damping, unew.data(), ss, ns));
        uprev[i] = v._reciprocal_buffer;
        v._buffer = _domain.ifft(unew);
        const double Rnorm = std::sqrt(ss[0]);
        if (_p.verbose)
          std::printf("%u |du| = %g |R|=%g\n", _iterations, std::sqrt(ss[1]), Rnorm);
        if (std::isnan(Rnorm))
        {
          all_converged = false;
          _iterations = _p.max_iterations;
          std::printf("NaN detected, aborting solve.\n");
          break;
        }
        all_converged = all_converged && (Rnorm < _p.absolute_tolerance || Rnorm / R0norm[i] < _p.relative_tolerance);
      }
      if (all_converged)
      {
        _is_converged = true;
        break;
      }
    }
    if (!all_converged)
    {
      std::printf("Solve not converged.\n");
      for (std::size_t i = 0; i < n; ++i)
        _variables[i]._buffer = _domain.ifft(u_old[i]);                              // :171-173
      _is_converged = false;
    }
  }

  const Params _p;
};


/// BroydenSolver (src/tensor_solver/BroydenSolver.C:34-176): the reference's control flow, two fused kernels per iteration
class BroydenSolver : public SplitOperatorABM, public IterativeTensorSolverInterface
{
public:
  struct Params
  {
    unsigned int substeps = 1, max_iterations = 5;
    double relative_tolerance = 1e-9, absolute_tolerance = 1e-9, initial_jacobian_guess = 1.0;
    bool verbose = false;
  };
  BroydenSolver(TensorProblem & problem, const std::string & name, std::shared_ptr<TensorOperatorBase> root_compute,
                const std::vector<VariableNames> & vars, const Params & p)
    : SplitOperatorABM(problem, name, p.substeps, std::move(root_compute), vars, 1, 1, 0), _p(p)
  {
    const int n = (int)_variables.size();
    if (n > 32)
      paramError("buffer", "at most 32 coupled variables");
    const int64_t ns = _domain.getReciprocalSize();
    _M = DeviceTensor::empty((std::size_t)(2 * n * n * ns));   // persists over substeps (:57-63)
    _domain.check(mrl_broyden_init(_domain.ctx(), n, _p.initial_jacobian_guess, _M.data(), ns));
  }

protected:
  void gather(std::vector<const double *> & u, std::vector<const double *> & N, std::vector<const double *> & L)
  {
    u.clear();
    N.clear();
    L.clear();
    for (auto & v : _variables)
    {
      u.push_back(v._reciprocal_buffer.data());
      N.push_back(v._nonlinear_reciprocal.data());
      L.push_back(v._linear_reciprocal ? v._linear_reciprocal->data() : nullptr);
    }
  }
  void substep() override
  {
    const int n = (int)_variables.size();
    const int64_t ns = _domain.getReciprocalSize();
    _compute->computeBuffer();
    forwardBuffers();
    std::vector<DeviceTensor> keep;                           // u_old handles (:71-77)
    std::vector<const double *> u, N, L, u_old;
    for (auto & v : _variables)
    {
      keep.push_back(v._reciprocal_buffer);
      u_old.push_back(keep.back().data());
    }
    gather(u, N, L);
    auto R = DeviceTensor::empty((std::size_t)(2 * n * ns)), S = DeviceTensor::empty((std::size_t)(2 * n * ns));
    double ss = 0.0;
    _domain.check(mrl_broyden_residual(_domain.ctx(), n, u.data(), N.data(), L.data(), nullptr, _sub_dt, R.data(), &ss, ns));
    const double R0norm = std::sqrt(ss);
    double Rnorm = R0norm;
    for (_iterations = 0; _iterations < _p.max_iterations; ++_iterations)
    {
      if (std::isnan(Rnorm))
        mooseError("NAN!");
      if (Rnorm < _p.absolute_tolerance || Rnorm / R0norm < _p.relative_tolerance)
      {
        _is_converged = true;
        return;
      }
      if (_p.verbose)
        std::printf("%u |R|=%g\n", _iterations, Rnorm);
      std::vector<DeviceTensor> out;
      std::vector<double *> outp;
      for (int i = 0; i < n; ++i)
      {
        out.push_back(DeviceTensor::empty((std::size_t)(2 * ns)));
        outp.push_back(out.back().data());
      }
      _domain.check(mrl_broyden_predict(_domain.ctx(), n, _M.data(), R.data(), u.data(), 0.5, S.data(), outp.data(), ns));
      for (int i = 0; i < n; ++i)
        _variables[i]._buffer = _domain.ifft(out[i]);
      _compute->computeBuffer();
      forwardBuffers();
      gather(u, N, L);
      _domain.check(mrl_broyden_update(_domain.ctx(), n, _M.data(), R.data(), S.data(), u.data(), N.data(), L.data(), u_old.data(),
                                       _sub_dt, &ss, ns));
      Rnorm = std::sqrt(ss);
    }
    std::fprintf(stderr, "Broyden solve did not converge within the maximum number of iterations.\n");
    _is_converged = false;
  }

  const Params _p;
  DeviceTensor _M;
};

/// TensorSolveIterationAdaptiveDT (src/timesteppers/TensorSolveIterationAdaptiveDT.C:66-88,162-175) + Transient's dtmax
class TensorSolveIterationAdaptiveDT
{
public:
  TensorSolveIterationAdaptiveDT(const IterativeTensorSolverInterface & solver, double dt, unsigned int min_iterations,
                                 unsigned int max_iterations, double growth_factor = 2.0, double cutback_factor = 0.5,
                                 double dtmax = 1e30)
    : _solver(solver), _input_dt(dt), _min_iterations(min_iterations), _max_iterations(max_iterations),
      _growth_factor(growth_factor), _cutback_factor(cutback_factor), _dtmax(dtmax)
  {
  }
  double computeDT(int t_step)
  {
    double dt = _input_dt;                       // computeInitialDT
    if (t_step > 1)
    {
      dt = _dt_old;
      const auto previous_iterations = _solver.getIterations();
      if (previous_iterations < _min_iterations)
        dt *= _growth_factor;
      else if (previous_iterations > _max_iterations)
        dt *= _cutback_factor;
    }
    dt = std::min(dt, _dtmax);
    _dt_old = dt;                                // acceptStep
    return dt;
  }

private:
  const IterativeTensorSolverInterface & _solver;
  const double _input_dt;
  const unsigned int _min_iterations, _max_iterations;
  const double _growth_factor, _cutback_factor, _dtmax;
  double _dt_old = 0.0;
};

/// a fused pointwise kernel over explicit device arrays (mrl_parsed_* with pointer inputs)
class FusedExpression
{
public:
  FusedExpression(DomainAction & domain, const std::string & expression, const std::vector<std::string> & inputs,
                  const std::vector<std::string> & complex_inputs, const std::vector<std::pair<std::string, double>> & constants,
                  bool extra_symbols = false, bool reciprocal = true)
    : _domain(domain)
  {
    std::vector<const char *> in, cn;
    std::vector<int> cplx;
    std::vector<double> cv;
    for (auto & n : inputs)
    {
      in.push_back(n.c_str());
      cplx.push_back(std::count(complex_inputs.begin(), complex_inputs.end(), n) ? 1 : 0);
    }
    for (auto & c : constants)
    {
      cn.push_back(c.first.c_str());
      cv.push_back(c.second);
    }
    if (mrl_parsed_create(domain.ctx(), &_p, expression.c_str(), (int)in.size(), in.data(), cplx.data(), (int)cn.size(), cn.data(),
                          cv.data(), 0, nullptr, extra_symbols ? 1 : 0, reciprocal ? 1 : 0) != MRL_OK)
      mooseError(std::string("FusedExpression: ") + mrl_last_error(domain.ctx()));
  }
  ~FusedExpression() { mrl_parsed_destroy(_p); }
  FusedExpression(const FusedExpression &) = delete;
  DeviceTensor operator()(const std::vector<const DeviceTensor *> & in, int64_t count) const
  {
    auto out = DeviceTensor::empty(count * (mrl_parsed_is_complex(_p) ? 2 : 1));
    std::vector<const double *> ptr;
    for (auto * t : in)
      ptr.push_back(t->data());
    _domain.check(mrl_parsed_eval(_p, ptr.data(), out.data(), count, 0.0));
    return out;
  }

private:
  DomainAction & _domain;
  mrl_parsed * _p = nullptr;
};


/// ReciprocalMatDiffusion (src/tensor_computes/ReciprocalMatDiffusion.C:44-66): divergence of the flux M grad(mu) with a no-flux
/// condition on the boundary of {psi > 0} (smooth boundary method), every pointwise step one generated kernel
class ReciprocalMatDiffusion : public TensorOperatorBase
{
public:
  ReciprocalMatDiffusion(TensorProblem & problem, const std::string & name, const std::string & buffer,
                         const std::string & chemical_potential, const std::string & mobility, const std::string & psi,
                         bool always_update_psi = false)
    : TensorOperatorBase(problem, name), _u(getOutputBuffer(buffer)), _chem_pot(getInputBuffer(chemical_potential)),
      _M(getInputBuffer(mobility)), _psi(getInputBuffer(psi)), _always_update_psi(always_update_psi), _dim(_domain.getDim())
  {
    static const char * k[] = {"kx", "ky", "kz"};
    static const char * g[] = {"gx", "gy", "gz"}, * j[] = {"jx", "jy", "jz"}, * a[] = {"ax", "ay", "az"};
    std::string div, nof;
    std::vector<std::string> ga, ja, aa;
    for (int d = 0; d < _dim; ++d)
    {
      _grad.emplace_back(new FusedExpression(_domain, std::string(k[d]) + "*a*i", {"a"}, {"a"}, {}, true, true));   // _i * fft(.) * _imag
      div += std::string(d ? "+" : "") + k[d] + "*" + a[d];
      nof += std::string(d ? "+" : "") + g[d] + "*" + j[d];
      ga.push_back(g[d]);
      ja.push_back(j[d]);
      aa.push_back(a[d]);
    }
    _by_psi.reset(new FusedExpression(_domain, "if(psi>0, g/psi, 0)", {"psi", "g"}, {}, {}, false, false));
    _flux.reset(new FusedExpression(_domain, "M*(psi>0)*g", {"M", "psi", "g"}, {}, {}, false, false));
    _div.reset(new FusedExpression(_domain, "i*(" + div + ")", aa, aa, {}, true, true));
    std::vector<std::string> gj = ga;
    gj.insert(gj.end(), ja.begin(), ja.end());
    _noflux.reset(new FusedExpression(_domain, nof, gj, {}, {}, false, false));
    _sum.reset(new FusedExpression(_domain, "a+b", {"a", "b"}, {"a", "b"}, {}, false, true));
  }
  void computeBuffer() override
  {
    const int64_t nr = _domain.getNumberOfCells(), ns = _domain.getReciprocalSize();
    if (_update_psi || _always_update_psi)
    {
      const auto psibar = _domain.fft(_psi);
      _grad_psi_by_psi.clear();
      for (int d = 0; d < _dim; ++d)
      {
        const auto g = _domain.ifft((*_grad[d])({&psibar}, ns));
        _grad_psi_by_psi.push_back((*_by_psi)({&_psi, &g}, nr));
      }
      _update_psi = false;
    }
    const auto mubar = _domain.fft(_chem_pot);
    std::vector<DeviceTensor> J, Jbar;
    for (int d = 0; d < _dim; ++d)
    {
      const auto g = _domain.ifft((*_grad[d])({&mubar}, ns));
      J.push_back((*_flux)({&_M, &_psi, &g}, nr));
      Jbar.push_back(_domain.fft(J.back()));
    }
    std::vector<const DeviceTensor *> in;
    for (auto & t : Jbar)
      in.push_back(&t);
    const auto div_J_hat = (*_div)(in, ns);
    in.clear();
    for (auto & t : _grad_psi_by_psi)
      in.push_back(&t);
    for (auto & t : J)
      in.push_back(&t);
    const auto no_flux_hat = _domain.fft((*_noflux)(in, nr));
    _u = (*_sum)({&div_J_hat, &no_flux_hat}, ns);
  }

private:
  DeviceTensor & _u;
  DeviceTensor & _chem_pot;
  DeviceTensor & _M;
  DeviceTensor & _psi;
  const bool _always_update_psi;
  const int _dim;
  bool _update_psi = true;
  std::vector<std::unique_ptr<FusedExpression>> _grad;
  std::unique_ptr<FusedExpression> _by_psi, _flux, _div, _noflux, _sum;
  std::vector<DeviceTensor> _grad_psi_by_psi;
};

/// ReciprocalAllenCahn (src/tensor_computes/ReciprocalAllenCahn.C:39-50): fft(where(psi > 0, -1 * L * dF_chem_deta, 0))
class ReciprocalAllenCahn : public TensorOperatorBase
{
public:
  ReciprocalAllenCahn(TensorProblem & problem, const std::string & name, const std::string & buffer, const std::string & dF_chem_deta,
                      const std::string & L, const std::string & psi)
    : TensorOperatorBase(problem, name), _u(getOutputBuffer(buffer)), _dF(getInputBuffer(dF_chem_deta)), _L(getInputBuffer(L)),
      _psi(getInputBuffer(psi)), _rate(_domain, "if(psi>0, -1*L*dF, 0)", {"psi", "L", "dF"}, {}, {}, false, false)
  {
  }
  void computeBuffer() override { _u = _domain.fft(_rate({&_psi, &_L, &_dF}, _domain.getNumberOfCells())); }

private:
  DeviceTensor & _u;
  DeviceTensor & _dF;
  DeviceTensor & _L;
  DeviceTensor & _psi;
  FusedExpression _rate;
};


/// FFTSemiImplicit (src/tensor_timeintegrators/FFTSemiImplicit.C:43-62, the legacy TensorTimeIntegrator form of the semi-implicit
/// update): ubar = (ubar0 + dt N)/(1 - dt L) without history, (ubar0 + dt/2 (3 N - N_old))/(1 - dt L) with it; u = ifft(ubar).
/// The same numbers as AdamsBashforthMoulton orders 1 and 2 up to rounding; one generated kernel per form
class FFTSemiImplicit : public TensorOperatorBase
{
public:
  FFTSemiImplicit(TensorProblem & problem, const std::string & name, const std::string & buffer, const std::string & reciprocal_buffer,
                  const std::string & linear_reciprocal, const std::string & nonlinear_reciprocal, unsigned int history_size = 1)
    : TensorOperatorBase(problem, name), _u(getOutputBuffer(buffer)), _reciprocal_buffer(getInputBuffer(reciprocal_buffer)),
      _linear_reciprocal(getInputBuffer(linear_reciprocal)), _non_linear_reciprocal(getInputBuffer(nonlinear_reciprocal)),
      _old_reciprocal_buffer(problem.getBufferOld(reciprocal_buffer, history_size)),
      _old_non_linear_reciprocal(problem.getBufferOld(nonlinear_reciprocal, history_size))
  {
  }
  void computeBuffer() override
  {
    const double dt = _tensor_problem.subDt();
    if (!_first || _dt_built != dt)
    {
      _first.reset(new FusedExpression(_domain, "(ubar + dt * N) / (1 - dt * L)", {"ubar", "N", "L"}, {"ubar", "N"}, {{"dt", dt}}));
      _second.reset(new FusedExpression(_domain, "(ubar + dt / 2 * (3 * N - No)) / (1 - dt * L)", {"ubar", "N", "No", "L"},
                                        {"ubar", "N", "No"}, {{"dt", dt}}));
      _dt_built = dt;
    }
    const auto n_old = std::min(_old_reciprocal_buffer.size(), _old_non_linear_reciprocal.size());
    const int64_t ns = _domain.getReciprocalSize();
    const auto ubar = n_old == 0 ? (*_first)({&_reciprocal_buffer, &_non_linear_reciprocal, &_linear_reciprocal}, ns)
                                 : (*_second)({&_reciprocal_buffer, &_non_linear_reciprocal, &_old_non_linear_reciprocal[0],
                                               &_linear_reciprocal}, ns);
    _u = _domain.ifft(ubar);
  }

private:
  DeviceTensor & _u;
  DeviceTensor & _reciprocal_buffer;
  DeviceTensor & _linear_reciprocal;
  DeviceTensor & _non_linear_reciprocal;
  const std::vector<DeviceTensor> & _old_reciprocal_buffer;
  const std::vector<DeviceTensor> & _old_non_linear_reciprocal;
  std::unique_ptr<FusedExpression> _first, _second;
  double _dt_built = 0.0;
};

/// drives TensorTimeIntegrators the way the pre-TensorSolver syntax did: per substep the root compute, then every integrator
class TimeIntegratorSolver : public TensorSolver
{
public:
  TimeIntegratorSolver(TensorProblem & problem, const std::string & name, unsigned int substeps,
                       std::shared_ptr<TensorOperatorBase> root_compute, std::vector<std::shared_ptr<TensorOperatorBase>> integrators)
    : TensorSolver(problem, name, substeps, std::move(root_compute)), _integrators(std::move(integrators))
  {
  }

protected:
  void substep() override
  {
    _compute->computeBuffer();
    forwardBuffers();
    for (auto & ti : _integrators)
      ti->computeBuffer();
  }
  std::vector<std::shared_ptr<TensorOperatorBase>> _integrators;
};

/// ETDRK4Solver::substep (src/tensor_solver/ETDRK4Solver.C:29-115).  Every k-space stage combination is ONE fused
/// kernel generated from the reference's own formulas (exp(L dt), the phi functions with their L dt == 0 limits and
/// the stage sums are evaluated in registers; the reference materialises ~25 full-size temporaries per variable).
class ETDRK4Solver : public TensorSolver
{
public:
  using VariableNames = SplitOperatorABM::VariableNames;
  ETDRK4Solver(TensorProblem & problem, const std::string & name, unsigned int substeps,
               std::shared_ptr<TensorOperatorBase> root_compute, const std::vector<VariableNames> & vars)
    : TensorSolver(problem, name, substeps, std::move(root_compute))
  {
    for (const auto & v : vars)
      _variables.push_back(Variable{problem.getBuffer(v.buffer), problem.getBuffer(v.reciprocal_buffer),
                                    v.linear_reciprocal == "0" ? nullptr : &problem.getBuffer(v.linear_reciprocal),
                                    problem.getBuffer(v.nonlinear_reciprocal)});
  }

protected:
  struct Variable
  {
    DeviceTensor & _buffer;
    const DeviceTensor & _reciprocal_buffer;
    const DeviceTensor * _linear_reciprocal;
    const DeviceTensor & _nonlinear_reciprocal;
  };

  void build()
  {
    const std::vector<std::pair<std::string, double>> c = {{"dt", _sub_dt}};
    _built_dt = _sub_dt;
    // ubar_b / ubar_c = expHalfLdt*ubar_n + 0.5*dt*N ; ubar_d = expLdt*ubar_n + dt*N                     :93,100,105
    _half = std::make_unique<FusedExpression>(_domain, "exp(L*dt/2.0)*ubar + 0.5*dt*N", std::vector<std::string>{"L", "ubar", "N"},
                                              std::vector<std::string>{"ubar", "N"}, c);
    _full = std::make_unique<FusedExpression>(_domain, "exp(L*dt)*ubar + dt*N", std::vector<std::string>{"L", "ubar", "N"},
                                              std::vector<std::string>{"ubar", "N"}, c);
    // phi functions with their Ldt == 0 limits (:75-91) and the final combination (:110-111)
    _final = std::make_unique<FusedExpression>(
        _domain,
        "Ldt := L*dt; E := exp(Ldt); den := Ldt*Ldt*Ldt;"
        "p1 := if(Ldt == 0.0, dt, dt*(-4.0 - 3.0*Ldt + E*(4.0 - Ldt))/den);"
        "p2 := if(Ldt == 0.0, dt*dt/2.0, dt*(2.0 + Ldt + E*(-2.0 + Ldt))/den);"
        "p3 := if(Ldt == 0.0, dt*dt/6.0, dt*(-4.0 - 3.0*Ldt - Ldt*Ldt + E*(4.0 - Ldt))/den);"
        "E*ubar + p1*N1 + 2.0*p2*(N2 + N3) + p3*N4",
        std::vector<std::string>{"L", "ubar", "N1", "N2", "N3", "N4"}, std::vector<std::string>{"ubar", "N1", "N2", "N3", "N4"}, c);
  }

  std::vector<DeviceTensor> evaluate_nonlinear(const std::vector<DeviceTensor> & ubar_stage)
  {
    for (std::size_t i = 0; i < _variables.size(); ++i)
      _variables[i]._buffer = _domain.ifft(ubar_stage[i]);
    _compute->computeBuffer();
    forwardBuffers();
    std::vector<DeviceTensor> nonlinear;
    for (auto & v : _variables)
      nonlinear.push_back(v._nonlinear_reciprocal);
    return nonlinear;
  }

  void substep() override
  {
    if (!_half || _built_dt != _sub_dt)
      build();
    _compute->computeBuffer();
    forwardBuffers();
    const int64_t nk = _domain.getReciprocalSize();
    const std::size_t nv = _variables.size();
    std::vector<DeviceTensor> ubar_n, linear, N1, stage(nv);
    for (auto & v : _variables)
    {
      ubar_n.push_back(v._reciprocal_buffer);
      N1.push_back(v._nonlinear_reciprocal);
      linear.push_back(v._linear_reciprocal ? *v._linear_reciprocal : DeviceTensor::zeros(nk));
    }
    for (std::size_t i = 0; i < nv; ++i)
      stage[i] = (*_half)({&linear[i], &ubar_n[i], &N1[i]}, nk);
    const auto N2 = evaluate_nonlinear(stage);
    for (std::size_t i = 0; i < nv; ++i)
      stage[i] = (*_half)({&linear[i], &ubar_n[i], &N2[i]}, nk);
    const auto N3 = evaluate_nonlinear(stage);
    for (std::size_t i = 0; i < nv; ++i)
      stage[i] = (*_full)({&linear[i], &ubar_n[i], &N3[i]}, nk);
    const auto N4 = evaluate_nonlinear(stage);
    for (std::size_t i = 0; i < nv; ++i)
      _variables[i]._buffer = _domain.ifft((*_final)({&linear[i], &ubar_n[i], &N1[i], &N2[i], &N3[i], &N4[i]}, nk));
  }

  std::vector<Variable> _variables;
  std::unique_ptr<FusedExpression> _half, _full, _final;
  double _built_dt = 0.0;
};

/// TensorExtremeValuePostprocessor / TensorIntegralPostprocessor
struct TensorPostprocessors
{
  static void extreme(DomainAction & d, const DeviceTensor & t, double & mn, double & mx)
  {
    d.check(mrl_minmax(d.ctx(), t.data(), (int64_t)t.numel(), &mn, &mx));
  }
  /// TensorAveragePostprocessor.C:37-51
  static double average(DomainAction & d, const DeviceTensor & t)
  {
    double s = 0.0;
    d.check(mrl_sum(d.ctx(), t.data(), (int64_t)t.numel(), &s));  // (the sum over all ranks on a slab domain)
    return s / (double)d.getGlobalNumberOfCells();
  }
  /// ReciprocalIntegral.C:28-47: Re(ubar[0,...,0]) / (number of cells) * volume (rank owning k = 0)
  static double reciprocalIntegral(DomainAction & d, const DeviceTensor & tbar, double volume)
  {
    double re = 0.0;
    d.check(mrl_sync(d.ctx()));
    if (hipMemcpy(&re, tbar.data(), sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
      mooseError("ReciprocalIntegral: hipMemcpy failed");
    return re / (double)d.getGlobalNumberOfCells() * volume;
  }
  /// integral = average * domain volume   (TensorIntegralPostprocessor.C:29-38)
  static double integral(DomainAction & d, const DeviceTensor & t, double volume)
  {
    double s = 0.0;
    d.check(mrl_sum(d.ctx(), t.data(), (int64_t)t.numel(), &s));
    return s / (double)d.getGlobalNumberOfCells() * volume;
  }
};


/// TensorInterfaceVelocityPostprocessor (src/postprocessors/TensorInterfaceVelocityPostprocessor.C:36-62): sqrt(max_cells sum_d
/// v_d^2), v_d = (u - u_old)/dt / grad_d(u) where |grad_d(u)| > 1e-3 (the reference hard-wires that threshold), else 0
class TensorInterfaceVelocityPostprocessor
{
public:
  TensorInterfaceVelocityPostprocessor(TensorProblem & problem, const std::string & buffer)
    : _problem(problem), _domain(problem.domain()), _u(problem.getBuffer(buffer)), _u_old(problem.getBufferOld(buffer, 1))
  {
    static const char * k[] = {"kx", "ky", "kz"};
    static const char * g[] = {"gx", "gy", "gz"};
    std::string expr = "du := (u - uo) / dt; ";
    std::vector<std::string> in = {"u", "uo"};
    std::string sum;
    for (int d = 0; d < _domain.getDim(); ++d)
    {
      _grad.emplace_back(new FusedExpression(_domain, std::string("ubar*") + k[d] + "*i", {"ubar"}, {"ubar"}, {}, true, true));
      expr += std::string("v") + g[d] + " := if(abs(" + g[d] + ") > 0.001, du / " + g[d] + ", 0); ";
      sum += std::string(d ? " + " : "") + "v" + g[d] + "*v" + g[d];
      in.push_back(g[d]);
    }
    _expr = expr + sum;
    _inputs = in;
  }
  double getValue()
  {
    if (_u_old.empty())
      return 0.0;
    const int64_t nr = _domain.getNumberOfCells(), ns = _domain.getReciprocalSize();
    if (!_vsq || _dt_built != _problem.dt())
    {
      _vsq.reset(new FusedExpression(_domain, _expr, _inputs, {}, {{"dt", _problem.dt()}}, false, false));
      _dt_built = _problem.dt();
    }
    const auto ubar = _domain.fft(_u);
    std::vector<DeviceTensor> g;
    for (auto & e : _grad)
      g.push_back(_domain.ifft((*e)({&ubar}, ns)));
    std::vector<const DeviceTensor *> in = {&_u, &_u_old[0]};
    for (auto & t : g)
      in.push_back(&t);
    const auto vsquare = (*_vsq)(in, nr);
    double mn, mx;
    TensorPostprocessors::extreme(_domain, vsquare, mn, mx);
    return std::sqrt(mx);
  }

private:
  TensorProblem & _problem;
  DomainAction & _domain;
  DeviceTensor & _u;
  const std::vector<DeviceTensor> & _u_old;
  std::vector<std::unique_ptr<FusedExpression>> _grad;
  std::unique_ptr<FusedExpression> _vsq;
  std::string _expr;
  std::vector<std::string> _inputs;
  double _dt_built = 0.0;
};

/// MOOSE Transient as far as the path sees it: advanceState, then the solver at EXEC_TIMESTEP_BEGIN
/// XDMFTensorOutput (src/tensor_outputs/XDMFTensorOutput.C:58-470, threading model of TensorOutput.C:66-81) in its raw-binary mode
/// (enable_hdf5 = false): one little-endian file per buffer component and frame, `<file_base>[.rankNNNN].<name>[_<comp>].<frame>.bin`
/// (:742-760), described by `<file_base>.xmf` (XDMF 2.2, temporal collection; in FFT_SLAB runs every time step is a spatial
/// collection of the ranks' blocks, :420-470).  CELL data; `transpose` (default true) swaps x <-> y (2-D) or x <-> z (3-D) as the
/// reference does for Paraview (:128-131, 278-292).
/// Device-speed path: startOutput() enqueues ASYNCHRONOUS device-to-host copies of the registered buffers into pinned staging memory
/// (two sets, used alternately) on a side stream that is ordered behind the solver's work, and hands the frame to a writer thread;
/// the solver's next time step starts at once.  Only the next startOutput() waits for the writer (TensorOutput::waitForCompletion).
class XDMFTensorOutput
{
public:
  struct Params
  {
    std::vector<std::string> buffer;       ///< names of the (real-space) buffers to write
    std::vector<int> components;           ///< values per grid point of each buffer (1 = scalar); empty = all scalar
    std::string file_base = "out";
    bool transpose = true;
    bool enable_hdf5 = false;              ///< one "<file_base>[.rankNNNN].h5" with datasets "<name>.<frame>" instead of raw files
    std::vector<std::string> output_mode;  ///< per buffer "CELL" (default), "NODE" (XDMFTensorOutput.C:41-51: nodal data, every
                                           ///< dimension extended by one with a copy of the slice at 0, extendTensor :530-557) or
                                           ///< "OVERSIZED_NODAL" (:47-49,287: the buffer already holds n + 1 points per dimension --
                                           ///< displacement fields -- and is written as it is, as nodal data; serial domains)
  };
  XDMFTensorOutput(TensorProblem & problem, const Params & p) : _problem(problem), _domain(problem.domain()), _p(p)
  {
    if (_p.components.empty())
      _p.components.assign(_p.buffer.size(), 1);
    if (_p.components.size() != _p.buffer.size())
      paramError("components", "one entry per buffer");
    if (_p.output_mode.empty())
      _p.output_mode.assign(_p.buffer.size(), "CELL");
    if (_p.output_mode.size() != _p.buffer.size())
      paramError("output_mode", "Specify one output mode per buffer.");                             // XDMFTensorOutput.C:80-82
    for (const auto & m : _p.output_mode)
    {
      if (m != "CELL" && m != "NODE" && m != "OVERSIZED_NODAL")
        paramError("output_mode", "CELL, NODE or OVERSIZED_NODAL");
      if (m == "OVERSIZED_NODAL" && _domain.isSlab())
        paramError("output_mode", "OVERSIZED_NODAL buffers are global fields (ComputeDisplacements): serial domains only");
    }
    if (hipStreamCreateWithFlags(&_copy_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&_ready[0]) != hipSuccess ||
        hipEventCreate(&_ready[1]) != hipSuccess || hipEventCreate(&_solver_done) != hipSuccess)
      mooseError("XDMFTensorOutput: creating the copy stream failed");
    std::size_t total = 0;
    for (std::size_t b = 0; b < _p.buffer.size(); ++b)
      total += pointsOf(b) * (std::size_t)_p.components[b];
    for (auto & st : _staging)
      if (hipHostMalloc(reinterpret_cast<void **>(&st), sizeof(double) * total) != hipSuccess)
        mooseError("XDMFTensorOutput: pinned staging allocation failed");
    if (_p.enable_hdf5 && mrl_h5_create(hdf5FileName(_domain.rank()).c_str(), &_h5) != MRL_OK)   // XDMFTensorOutput.C:152-160
      mooseError("Error opening HDF5 file '" + hdf5FileName(_domain.rank()) + "'.");
    if (_domain.rank() == 0)
      writeXMF();   // skeleton (valid XDMF with zero frames)
  }
  ~XDMFTensorOutput()
  {
    waitForCompletion();
    if (_h5)
      (void)mrl_h5_close(_h5);                                                                      // XDMFTensorOutput.C:113-115
    for (auto & st : _staging)
      (void)hipHostFree(st);
    (void)hipEventDestroy(_ready[0]);
    (void)hipEventDestroy(_ready[1]);
    (void)hipEventDestroy(_solver_done);
    (void)hipStreamDestroy(_copy_stream);
  }
  /// TensorOutput::startOutput: snapshot (asynchronously) and write in the background
  void startOutput()
  {
    // the staging set of this frame was last used two frames ago, whose writer was joined when the previous frame started: the
    // copies of this frame can go out while the previous frame is still being written
    double * stage = _staging[_frame % 2];
    // order the copies behind everything the solver has enqueued (it works on the context's stream = the HIP null stream here)
    if (hipEventRecord(_solver_done, nullptr) != hipSuccess || hipStreamWaitEvent(_copy_stream, _solver_done, 0) != hipSuccess)
      mooseError("XDMFTensorOutput: ordering the copy stream failed");
    std::size_t off = 0;
    auto & held = _held[_frame % 2];
    held.clear();
    for (std::size_t b = 0; b < _p.buffer.size(); ++b)
    {
      const DeviceTensor t = _problem.getBuffer(_p.buffer[b]);   // a handle: keeps the array alive while it is being copied
      const std::size_t n = pointsOf(b) * (std::size_t)_p.components[b];
      if (!t.defined() || t.numel() != n)
        mooseError("XDMFTensorOutput: buffer '" + _p.buffer[b] + "' is undefined or has an unexpected size");
      if (hipMemcpyAsync(stage + off, t.data(), sizeof(double) * n, hipMemcpyDeviceToHost, _copy_stream) != hipSuccess)
        mooseError("XDMFTensorOutput: hipMemcpyAsync failed");
      held.push_back(t);
      off += n;
    }
    if (hipEventRecord(_ready[_frame % 2], _copy_stream) != hipSuccess)
      mooseError("XDMFTensorOutput: hipEventRecord failed");
    waitForCompletion();   // "Output thread is already running. Must call waitForCompletion() first."
    const double time = _problem.time();   // a dedicated output time, not changed while the output runs
    const int frame = _frame++;
    _times.push_back(time);
    _thread = std::thread([this, stage, frame]() { this->output(stage, frame); });
  }
  void waitForCompletion()
  {
    if (_thread.joinable())
      _thread.join();
    if (!_error.empty())
      mooseError(_error);
  }
  int frames() const { return _frame; }
  double secondsWriting() const { return _seconds_writing; }

private:
  std::string rankTag() const
  {
    if (!_domain.isSlab())
      return "";
    char buf[32];
    std::snprintf(buf, sizeof buf, ".rank%04d", _domain.rank());
    return buf;
  }
  std::string hdf5FileName(int rank) const
  {
    char buf[32] = "";
    if (_domain.isSlab())
      std::snprintf(buf, sizeof buf, ".rank%04d", rank);
    return _p.file_base + buf + ".h5";
  }
  std::string binaryFileName(const std::string & setname, int rank) const
  {
    char buf[32] = "";
    if (_domain.isSlab())
      std::snprintf(buf, sizeof buf, ".rank%04d", rank);
    return _p.file_base + buf + "." + setname + ".bin";
  }
  static std::string componentName(const std::string & name, int comps, int c)
  {
    static const char * xyz[3] = {"x", "y", "z"};   // buildAttributeNames (XDMFTensorOutput.C:654-670): vectors _x _y _z, tensors _0 ...
    return comps == 1 ? name : name + "_" + (comps <= 3 ? std::string(xyz[c]) : std::to_string(c));
  }
  /// grid points of buffer b as it is stored: the cells of the domain, or n + 1 points per dimension for an OVERSIZED_NODAL buffer
  std::size_t pointsOf(std::size_t b) const
  {
    if (_p.output_mode[b] != "OVERSIZED_NODAL")
      return (std::size_t)_domain.getNumberOfCells();
    std::size_t n = 1;
    for (const auto v : _domain.getLocalShape())
      n *= (std::size_t)(v + 1);
    return n;
  }
  /// the writer thread: wait for the copies of this frame, transpose on the host, write the files, update the .xmf
  void output(const double * stage, int frame)
  {
    const auto t0 = std::chrono::steady_clock::now();
    if (hipEventSynchronize(_ready[frame % 2]) != hipSuccess)
    {
      _error = "XDMFTensorOutput: waiting for the device-to-host copies failed";
      return;
    }
    const auto & ls = _domain.getLocalShape();
    const int dim = _domain.getDim();
    const int64_t n0 = ls[0], n1 = dim > 1 ? ls[1] : 1, n2 = dim > 2 ? ls[2] : 1;
    std::vector<double> slice;
    std::size_t off = 0;
    for (std::size_t b = 0; b < _p.buffer.size(); ++b)
    {
      const int comps = _p.components[b];
      // NODE: every dimension one longer, the extra slice is a copy of slice 0 (extendTensor); OVERSIZED_NODAL: the stored field already
      // has the extra points (no periodic copy); e = extents of what is written, s = extents of what is stored
      const bool oversized = _p.output_mode[b] == "OVERSIZED_NODAL";
      const int64_t ex = (_p.output_mode[b] == "NODE" || oversized) ? 1 : 0;
      const int64_t e0 = n0 + ex, e1 = dim > 1 ? n1 + ex : 1, e2 = dim > 2 ? n2 + ex : 1;
      const int64_t s0 = oversized ? e0 : n0, s1 = oversized ? e1 : n1, s2 = oversized ? e2 : n2;
      slice.resize((std::size_t)(e0 * e1 * e2));
      for (int c = 0; c < comps; ++c)
      {
        const double * src = stage + off;
        auto at = [&](int64_t i, int64_t j, int64_t k) { return src[(std::size_t)((((i % s0) * s1 + (j % s1)) * s2 + (k % s2))) * comps + c]; };
        // component c of a value-major field, transposed x <-> last axis if requested
        if (dim == 3 && _p.transpose)
        {
          for (int64_t k = 0; k < e2; ++k)
            for (int64_t j = 0; j < e1; ++j)
              for (int64_t i = 0; i < e0; ++i)
                slice[(std::size_t)((k * e1 + j) * e0 + i)] = at(i, j, k);
        }
        else if (dim == 2 && _p.transpose)
        {
          for (int64_t j = 0; j < e1; ++j)
            for (int64_t i = 0; i < e0; ++i)
              slice[(std::size_t)(j * e0 + i)] = at(i, j, 0);
        }
        else
          for (int64_t i = 0; i < e0; ++i)
            for (int64_t j = 0; j < e1; ++j)
              for (int64_t k = 0; k < e2; ++k)
                slice[(std::size_t)((i * e1 + j) * e2 + k)] = at(i, j, k);
        const std::string setname = componentName(_p.buffer[b], comps, c) + "." + std::to_string(frame);
        if (_h5)
        {
          // addDataToHDF5 (XDMFTensorOutput.C:323-343): dims = the spatial sizes of the (transposed) buffer
          int64_t dims[3];
          for (int i = 0; i < dim; ++i)
            dims[i] = (_p.transpose ? ls[dim - 1 - i] : ls[i]) + ex;
          if (mrl_h5_write(_h5, setname.c_str(), MRL_H5_F64, dim, dims, slice.data()) != MRL_OK)
          {
            _error = std::string("XDMFTensorOutput: ") + mrl_h5_last_error(_h5);
            return;
          }
          continue;
        }
        std::ofstream f(binaryFileName(setname, _domain.rank()), std::ios::binary);
        if (!f || !f.write(reinterpret_cast<const char *>(slice.data()), sizeof(double) * slice.size()))
        {
          _error = "XDMFTensorOutput: cannot write " + binaryFileName(setname, _domain.rank());
          return;
        }
      }
      off += pointsOf(b) * (std::size_t)comps;
    }
    if (_h5 && mrl_h5_flush(_h5) != MRL_OK)                                                        // XDMFTensorOutput.C:244-246
    {
      _error = std::string("XDMFTensorOutput: ") + mrl_h5_last_error(_h5);
      return;
    }
    if (_domain.rank() == 0)
      writeXMF();
    _seconds_writing += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  /// the whole .xmf file for the frames written so far (rank 0; small, rewritten every frame as the reference's _doc.save_file)
  void writeXMF()
  {
    const int dim = _domain.getDim();
    const auto & g = _domain.getShape();
    auto axis = [&](int i) { return _p.transpose ? dim - 1 - i : i; };   // mappedAxis
    std::ostringstream x;
    x.precision(17);
    x << "<?xml version=\"1.0\"?>\n<Xdmf xmlns:xi=\"http://www.w3.org/2003/XInclude\" Version=\"2.2\">\n <Domain>\n";
    std::string nodes, cellsdim, origin, spacing, geo = "ORIGIN_";
    const char * dxyz[] = {"DX", "DY", "DZ"};
    for (int i = 0; i < dim; ++i)
    {
      const int j = axis(i);
      nodes += (i ? " " : "") + std::to_string(g[j] + 1);
      cellsdim += (i ? " " : "") + std::to_string(g[j]);
      std::ostringstream o, d;
      o.precision(17);
      d.precision(17);
      o << 0.0;
      d << _domain.getExtent(j) / (double)g[j];
      origin += (i ? " " : "") + o.str();
      spacing += (i ? " " : "") + d.str();
      geo += dxyz[i];
    }
    if (!_domain.isSlab())
    {
      x << "  <Topology TopologyType=\"" << dim << "DCoRectMesh\" Dimensions=\"" << nodes << "\"/>\n";
      x << "  <Geometry Type=\"" << geo << "\">\n   <DataItem Format=\"XML\" Dimensions=\"" << dim << "\">" << origin
        << "</DataItem>\n   <DataItem Format=\"XML\" Dimensions=\"" << dim << "\">" << spacing << "</DataItem>\n  </Geometry>\n";
    }
    x << "  <Grid Name=\"TimeSeries\" GridType=\"Collection\" CollectionType=\"Temporal\">\n";
    for (int f = 0; f < (int)_times.size(); ++f)
    {
      if (!_domain.isSlab())
      {
        x << "   <Grid Name=\"T" << f << "\" GridType=\"Uniform\">\n    <Time Value=\"" << _times[f] << "\"/>\n"
          << "    <xi:include xpointer=\"xpointer(//Xdmf/Domain/Topology)\"/>\n    <xi:include xpointer=\"xpointer(//Xdmf/Domain/Geometry)\"/>\n";
        attributes(x, f, 0, cellsdim, nodes);
        x << "   </Grid>\n";
        continue;
      }
      // FFT_SLAB: a spatial collection of the ranks' y-slabs (writeParallelXMF, XDMFTensorOutput.C:420-470)
      x << "   <Grid Name=\"T" << f << "\" GridType=\"Collection\" CollectionType=\"Spatial\">\n    <Time Value=\"" << _times[f] << "\"/>\n";
      int64_t ybeg = 0;
      for (int r = 0; r < _domain.nranks(); ++r)
      {
        std::vector<int64_t> counts(_domain.nranks());
        if (mrl_partition(g[1], _domain.nranks(), nullptr, counts.data()) != MRL_OK)
          mooseError("XDMFTensorOutput: partition failed");
        std::string rn, rc, ro;
        for (int i = 0; i < dim; ++i)
        {
          const int j = axis(i);
          const int64_t cnt = j == 1 ? counts[r] : g[j];
          rn += (i ? " " : "") + std::to_string(cnt + 1);
          rc += (i ? " " : "") + std::to_string(cnt);
          std::ostringstream o;
          o.precision(17);
          o << (j == 1 ? (double)ybeg * _domain.getExtent(1) / (double)g[1] : 0.0);
          ro += (i ? " " : "") + o.str();
        }
        x << "    <Grid Name=\"Rank" << r << "\" GridType=\"Uniform\">\n     <Topology TopologyType=\"" << dim << "DCoRectMesh\" Dimensions=\"" << rn
          << "\"/>\n     <Geometry Type=\"" << geo << "\">\n      <DataItem Format=\"XML\" Dimensions=\"" << dim << "\">" << ro
          << "</DataItem>\n      <DataItem Format=\"XML\" Dimensions=\"" << dim << "\">" << spacing << "</DataItem>\n     </Geometry>\n";
        attributes(x, f, r, rc, rn);
        x << "    </Grid>\n";
        ybeg += counts[r];
      }
      x << "   </Grid>\n";
    }
    x << "  </Grid>\n </Domain>\n</Xdmf>\n";
    std::ofstream f(_p.file_base + ".xmf");
    f << x.str();
  }
  void attributes(std::ostringstream & x, int frame, int rank, const std::string & celldims, const std::string & nodedims) const
  {
    for (std::size_t b = 0; b < _p.buffer.size(); ++b)
      for (int c = 0; c < _p.components[b]; ++c)
      {
        const bool node = _p.output_mode[b] != "CELL";                      // XDMFTensorOutput.C:375-394 (NODE and OVERSIZED_NODAL)
        const std::string & dims = node ? nodedims : celldims;
        const char * center = node ? "Node" : "Cell";
        const std::string name = componentName(_p.buffer[b], _p.components[b], c);
        std::string file = _p.enable_hdf5 ? hdf5FileName(rank) : binaryFileName(name + "." + std::to_string(frame), rank);
        const auto slash = file.find_last_of('/');
        if (slash != std::string::npos)
          file = file.substr(slash + 1);   // relative to the .xmf file
        if (_p.enable_hdf5)   // XDMFTensorOutput.C:408-412
          x << "     <Attribute Name=\"" << name << "\" Center=\"" << center << "\">\n      <DataItem DataType=\"Float\" Dimensions=\"" << dims
            << "\" Format=\"HDF\">" << file << ":/" << name << "." << frame << "</DataItem>\n     </Attribute>\n";
        else
          x << "     <Attribute Name=\"" << name << "\" Center=\"" << center << "\">\n      <DataItem DataType=\"Float\" Dimensions=\"" << dims
            << "\" Format=\"Binary\" Endian=\"Little\" Precision=\"8\">" << file << "</DataItem>\n     </Attribute>\n";
      }
  }

  TensorProblem & _problem;
  DomainAction & _domain;
  Params _p;
  hipStream_t _copy_stream = nullptr;
  hipEvent_t _ready[2] = {nullptr, nullptr}, _solver_done = nullptr;
  double * _staging[2] = {nullptr, nullptr};
  std::vector<DeviceTensor> _held[2];
  std::vector<double> _times;
  std::thread _thread;
  std::string _error;
  int _frame = 0;
  double _seconds_writing = 0.0;
  mrl_h5 * _h5 = nullptr;
};

class Transient
{
public:
  Transient(TensorProblem & problem, TensorSolver & solver, double dt) : _problem(problem), _solver(solver), _dt(dt) {}
  /// [TimeStepper]: dt for time step `t_step` (1-based); default = the constant dt
  void setTimeStepper(std::function<double(int)> stepper) { _stepper = std::move(stepper); }
  template <typename F>
  void execute(int num_steps, F && on_timestep_end)
  {
    for (int s = 0; s < num_steps; ++s)
    {
      _problem.timeOld() = _problem.time();
      _problem.timeStep() += 1;
      const double dt_prev = _problem.dt();              // TransientBase::takeStep: _dt_old = _dt
      if (_stepper)
        _dt = _stepper(_problem.timeStep());
      _problem.dt() = _dt;
      _problem.dtOld() = _problem.timeStep() > 1 ? dt_prev : _dt;  // (nothing has changed before the first step)
      _problem.time() = _problem.timeOld() + _dt;
      _problem.advanceState();                         // incrementStepOrReject -> advanceState
      _problem.subTime() = _problem.timeOld();         // TensorProblem.C:179
      _solver.computeBuffer();                         // EXEC_TIMESTEP_BEGIN
      on_timestep_end(_problem.timeStep());            // EXEC_TIMESTEP_END: outputs
    }
  }

private:
  TensorProblem & _problem;
  TensorSolver & _solver;
  double _dt;
  std::function<double(int)> _stepper;
};

}  // namespace marlin_host
