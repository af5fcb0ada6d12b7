// Marlin-side glue (built inside the MOOSE application, only when MOOSE_DIR exists): one libmarlin_hip context per DomainAction.
//
// Binds include/marlin_hip.h to the objects Marlin already owns.  Nothing in Marlin is modified: these files are added to the
// application's source tree (marlin_plugin/marlin_plugin.mk), libmarlin_hip.so is linked, and input files pick
// `type = HipAdamsBashforthMoulton` / `type = HipFFTMechanics`.  libTorch is used for memory only (torch::empty, data_ptr).
//
// Reference interfaces bound here (paths relative to idaholab/marlin):
//   include/actions/DomainAction.h:31-39,69   getDim / getGridSize / getDomainMin / getDomainMax / getShape
//   src/actions/DomainAction.C:163-199        one MPI rank <-> one device, chosen from the host-local rank
//   src/actions/DomainAction.C:510-566        partitionSlabs (nranks > 1: parallel_mode = FFT_SLAB)
//   src/actions/DomainAction.C:568-742        partitionPencils (MRL_FLAG_PENCIL: parallel_mode = FFT_PENCIL)
#pragma once

#include "DomainAction.h"
#include "MarlinUtils.h"
#include "MooseError.h"
#include "libmesh/parallel.h"

#include <c10/hip/HIPStream.h>
#include <ctime>
#include <string>
#include <unistd.h>

#include "marlin_hip.h"

/// RAII owner of the context (and, in parallel, of the communicator: destroyed after the context)
class HipDomain
{
public:
  HipDomain(const DomainAction & d, const libMesh::Parallel::Communicator & comm, bool pencil = false)
  {
    mrl_domain dom{};
    dom.dim = d.getDim();
    const auto & n = d.getGridSize();
    for (unsigned int i = 0; i < d.getDim(); ++i)
    {
      dom.n[i] = n[i];
      dom.min[i] = d.getDomainMin()(i);
      dom.max[i] = d.getDomainMax()(i);
    }
    dom.device = MooseTensor::floatTensorOptions().device().index();
    dom.nranks = comm.size();
    dom.rank = comm.rank();
    dom.weights = nullptr; // or the [Domain] device_weights vector
    // 2-D slab runs keep the reference's c2c layout; everything else is r2c on the last axis (pencil: on x, as DomainAction.C:282-284)
    dom.spectrum = dom.nranks > 1 && dom.dim == 2 ? MRL_SPECTRUM_FULL : MRL_SPECTRUM_HALF;
    dom.stream = c10::hip::getCurrentHIPStream().stream(); // the stream libTorch enqueues on
    dom.flags = pencil ? MRL_FLAG_PENCIL : 0;
    if (mrl_ctx_create(&_ctx, &dom) != MRL_OK)
      mooseError("marlin_hip: ", mrl_last_error(nullptr));
    if (dom.nranks > 1)
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

  /// turn a return code into a mooseError carrying the library's message
  void check(int rc, const std::string & who) const
  {
    if (rc != MRL_OK)
      mooseError(who, ": ", mrl_last_error(_ctx));
  }

private:
  mrl_ctx * _ctx = nullptr;
  mrl_comm * _comm = nullptr;
};
