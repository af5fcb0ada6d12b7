// Marlin-side glue (built inside the MOOSE application, only when MOOSE_DIR exists): one libmarlin_hip context per DomainAction,
// shared by every Hip* object of the run.
//
// Binds include/marlin_hip.h to the objects Marlin already owns.  Nothing in Marlin is modified: these files are added to the
// application's source tree (marlin_plugin/marlin_plugin.mk), libmarlin_hip.so is linked, and input files pick
// `type = HipAdamsBashforthMoulton` / `type = HipFFTMechanics` / `type = HipForwardFFT` ...  libTorch is used for memory only
// (torch::empty, data_ptr).
//
// Reference interfaces bound here (paths relative to idaholab/marlin):
//   include/actions/DomainAction.h:31-39,69   getDim / getGridSize / getDomainMin / getDomainMax / getShape / getLocalBounds
//   src/actions/DomainAction.C:163-199        one MPI rank <-> one device, chosen from the host-local rank
//   src/actions/DomainAction.C:510-566        partitionSlabs (nranks > 1: parallel_mode = FFT_SLAB)
//   src/actions/DomainAction.C:568-742        partitionPencils (parallel_mode = FFT_PENCIL -> MRL_FLAG_PENCIL)
#pragma once

#include "DomainAction.h"
#include "MarlinUtils.h"
#include "MooseError.h"
#include "libmesh/parallel.h"

#include <c10/hip/HIPStream.h>
#include <array>
#include <ctime>
#include <map>
#include <memory>
#include <string>
#include <unistd.h>

#include "marlin_hip.h"

/// RAII owner of the context (and, in parallel, of the communicator: destroyed after the context)
class HipDomain
{
public:
  /// the context of this DomainAction (created by the first object that asks; the work arrays of the library exist once per run)
  static std::shared_ptr<HipDomain> get(const DomainAction & d, const libMesh::Parallel::Communicator & comm)
  {
    static std::map<const DomainAction *, std::weak_ptr<HipDomain>> live;
    auto & slot = live[&d];
    auto sp = slot.lock();
    if (!sp)
    {
      sp = std::shared_ptr<HipDomain>(new HipDomain(d, comm));
      slot = sp;
    }
    return sp;
  }

  ~HipDomain()
  {
    if (_ctx)
      mrl_ctx_destroy(_ctx);
    if (_comm)
      mrl_comm_destroy(_comm);
  }
  HipDomain(const HipDomain &) = delete;
  HipDomain & operator=(const HipDomain &) = delete;

  mrl_ctx * ctx() const { return _ctx; }
  unsigned int dim() const { return _dim; }
  bool parallel() const { return _nranks > 1; }

  /// local extents as the library partitions them == the DomainAction's (checked at construction)
  int64_t realCount() const { return _real_n[0] * _real_n[1] * _real_n[2]; }
  int64_t reciprocalCount() const { return _recip_n[0] * _recip_n[1] * _recip_n[2]; }
  std::vector<int64_t> realShape() const { return std::vector<int64_t>(_real_n.begin(), _real_n.begin() + _dim); }
  std::vector<int64_t> reciprocalShape() const { return std::vector<int64_t>(_recip_n.begin(), _recip_n.begin() + _dim); }

  /// A flat array in the solver-private spectral layout (mrl_ch_spec_elems complex values) seen as the local reciprocal block:
  /// consumers of the published buffer read values, never the padding.
  std::vector<int64_t> spectralStrides() const
  {
    int64_t plane = 0, row = 0;
    mrl_ch_spec_layout(_ctx, &plane, &row);
    if (_dim == 3)
      return {plane, row, 1};
    if (_dim == 2)
      return {row, 1};
    return {1};
  }
  torch::Tensor spectralView(const torch::Tensor & flat) const { return torch::as_strided(flat, reciprocalShape(), spectralStrides()); }
  /// is `t` one of those views (its storage is an array in the private layout)?
  bool isSpectralView(const torch::Tensor & t) const
  {
    return t.defined() && t.is_complex() && t.storage_offset() == 0 && (int64_t)t.storage().nbytes() >= 16 * mrl_ch_spec_elems(_ctx) &&
           t.sizes().vec() == reciprocalShape() && t.strides().vec() == spectralStrides();
  }

  /// host copy of this rank's reciprocal axis `a` (what `_i`, `_j`, `_k` hold in a TensorOperatorBase); absent axes are {0}
  std::vector<double> reciprocalAxis(unsigned int a) const
  {
    if (a >= _dim)
      return {0.0};
    std::vector<double> k((std::size_t)_recip_n[a]);
    check(mrl_ctx_reciprocal_axis(_ctx, (int)a, k.data(), (int64_t)k.size()), "marlin_hip");
    return k;
  }

  /// turn a return code into a mooseError carrying the library's message
  void check(int rc, const std::string & who) const
  {
    if (rc != MRL_OK)
      mooseError(who, ": ", mrl_last_error(_ctx));
  }

private:
  HipDomain(const DomainAction & d, const libMesh::Parallel::Communicator & comm) : _dim(d.getDim()), _nranks(comm.size())
  {
    if (MooseTensor::floatTensorOptions().dtype() != torch::kFloat64)
      mooseError("marlin_hip: the Hip* objects bind the double precision entry points; run with floating_precision = DOUBLE");
    if (!MooseTensor::floatTensorOptions().device().is_cuda())
      mooseError("marlin_hip: the Hip* objects need the tensors on the HIP device (libTorch device type 'cuda' on ROCm builds)");
    mrl_domain dom{};
    dom.dim = _dim;
    const auto & n = d.getGridSize();
    for (unsigned int i = 0; i < _dim; ++i)
    {
      dom.n[i] = n[i];
      dom.min[i] = d.getDomainMin()(i);
      dom.max[i] = d.getDomainMax()(i);
    }
    dom.device = MooseTensor::floatTensorOptions().device().index();
    if (dom.device < 0)
      dom.device = c10::hip::current_device();
    dom.nranks = _nranks;
    dom.rank = comm.rank();
    dom.weights = nullptr; // or the [Domain] device_weights vector

    // The DomainAction does not publish its parallel mode; its partition does.  partitionSlabs splits y only (DomainAction.C:524-533),
    // partitionPencils y and z (:590-640): a local z extent shorter than the global one means FFT_PENCIL.
    std::array<int64_t, 3> begin{{0, 0, 0}}, end{{1, 1, 1}};
    d.getLocalBounds(comm.rank(), begin, end);
    const auto & nl = d.getLocalGridSize();
    const bool pencil = _nranks > 1 && _dim == 3 && nl[2] != n[2];
    if (_nranks > 1 && !d.isParallelFFT())
      mooseError("marlin_hip: parallel_mode must be FFT_SLAB or FFT_PENCIL when running on several ranks");
    // 2-D slab runs keep the reference's c2c layout; everything else is r2c on the last axis (pencil: on x, as DomainAction.C:282-284).
    // In 3-D FFT_SLAB the library keeps r2c on z where the reference transforms c2c (DomainAction.C:278-280): same fields, half the
    // exchange volume, and a reciprocal block of nz/2+1 instead of nz entries along z -- spectral buffers published by the Hip*
    // objects have the library's shape there (reciprocalShape()), which checkLayout() reports.
    dom.spectrum = _nranks > 1 && _dim == 2 ? MRL_SPECTRUM_FULL : MRL_SPECTRUM_HALF;
    dom.stream = c10::hip::getCurrentHIPStream().stream(); // the stream libTorch enqueues on
    dom.flags = pencil ? MRL_FLAG_PENCIL : 0;
    if (mrl_ctx_create(&_ctx, &dom) != MRL_OK)
      mooseError("marlin_hip: ", mrl_last_error(nullptr));
    checkLayout(d, begin, end, pencil);
    if (_nranks > 1)
    {
      // the library owns the global transposes (HIP IPC peer stores / copy engines / RCCL) in place of the host-staged
      // MPI_Isend / MPI_Recv loops of DomainAction::fftSlab / ifftSlab (DomainAction.C:889-927, 960-1005).  MPI is only used
      // to agree on one job name.
      std::string job =
          comm.rank() == 0 ? "marlin_" + std::to_string(getpid()) + "_" + std::to_string(std::time(nullptr)) : "";
      comm.broadcast(job);
      if (mrl_comm_create(&_comm, job.c_str(), dom.nranks, dom.rank, dom.device, MRL_TRANSPORT_AUTO) != MRL_OK)
        mooseError("marlin_hip: ", mrl_comm_last_error(nullptr));
      if (mrl_ctx_attach_comm(_ctx, _comm) != MRL_OK)
        mooseError("marlin_hip: ", mrl_last_error(_ctx));
    }
  }

  /// The blocks the library works on must be the blocks Marlin's tensors hold: same local extents in real and reciprocal space on
  /// every axis, and in FFT_SLAB mode the same offsets (getLocalBounds hands out _local_begin / _local_end: the RECIPROCAL x range on
  /// axis 0, the real y range on axis 1, DomainAction.C:524-533).  A mismatch would be out-of-bounds access, so it is an error at
  /// construction, not a surprise in the first substep.
  void checkLayout(const DomainAction & d, const std::array<int64_t, 3> & begin, const std::array<int64_t, 3> & end, bool pencil)
  {
    int64_t rb[3], kb[3];
    if (mrl_local_shape(_ctx, _real_n.data(), rb, _recip_n.data(), kb) != MRL_OK)
      mooseError("marlin_hip: ", mrl_last_error(_ctx));
    const auto & nl = d.getLocalGridSize();
    const auto & kl = d.getLocalReciprocalGridSize();
    for (unsigned int a = 0; a < _dim; ++a)
    {
      if (_real_n[a] != nl[a])
        mooseError("marlin_hip: the library's real-space block differs from the DomainAction's along axis ", a, " (", _real_n[a],
                   " against ", nl[a], " entries): unsupported partition (device_weights?)");
      // (3-D FFT_SLAB: r2c on z here, c2c in the reference -- the one extent that differs by design, see the constructor)
      const bool r2c_kept = _nranks > 1 && !pencil && _dim == 3 && a == 2 && _recip_n[2] == nl[2] / 2 + 1;
      if (_recip_n[a] != kl[a] && !r2c_kept)
        mooseError("marlin_hip: the library's reciprocal block differs from the DomainAction's along axis ", a, " (", _recip_n[a],
                   " against ", kl[a], ")");
    }
    if (_nranks > 1 && !pencil)
    {
      if (kb[0] != begin[0] || kb[0] + _recip_n[0] != end[0])
        mooseError("marlin_hip: this rank's reciprocal x planes are [", kb[0], ", ", kb[0] + _recip_n[0], ") in the library and [", begin[0],
                   ", ", end[0], ") in the DomainAction");
      if (rb[1] != begin[1] || rb[1] + _real_n[1] != end[1])
        mooseError("marlin_hip: this rank's real-space y planes are [", rb[1], ", ", rb[1] + _real_n[1], ") in the library and [", begin[1],
                   ", ", end[1], ") in the DomainAction");
    }
    for (unsigned int a = _dim; a < 3; ++a)
      _real_n[a] = _recip_n[a] = 1;
  }

  const unsigned int _dim;
  const int _nranks;
  std::array<int64_t, 3> _real_n{{1, 1, 1}}, _recip_n{{1, 1, 1}};
  mrl_ctx * _ctx = nullptr;
  mrl_comm * _comm = nullptr;
};
