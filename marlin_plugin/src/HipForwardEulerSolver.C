// HipForwardEulerSolver: ForwardEulerSolver::substep (src/tensor_solver/ForwardEulerSolver.C:28-38) on libmarlin_hip.
#include "HipForwardEulerSolver.h"
#include "TensorProblem.h"
#include "DomainAction.h"

registerMooseObject("MarlinApp", HipForwardEulerSolver);

InputParameters
HipForwardEulerSolver::validParams()
{
  InputParameters params = TensorSolver::validParams();
  params.addClassDescription("Explicit (forward Euler) time integration solver on libmarlin_hip (MI355X).");
  params.addParam<std::vector<TensorOutputBufferName>>("buffer", {}, "The buffer this solver is writing to");
  params.addParam<std::vector<TensorInputBufferName>>("reciprocal_buffer", {}, "Buffer with the reciprocal of the integrated buffer");
  params.addParam<std::vector<TensorInputBufferName>>("time_derivative_reciprocal", {}, "Buffer with the reciprocal of the time derivative function");
  return params;
}

HipForwardEulerSolver::HipForwardEulerSolver(const InputParameters & parameters)
  : TensorSolver(parameters), _hip(HipDomain::get(_domain, comm()))
{
  const auto & buffers = getParam<std::vector<TensorOutputBufferName>>("buffer");
  const auto & reciprocal = getParam<std::vector<TensorInputBufferName>>("reciprocal_buffer");
  const auto & rate = getParam<std::vector<TensorInputBufferName>>("time_derivative_reciprocal");
  if (reciprocal.size() != buffers.size() || rate.size() != buffers.size())
    paramError("buffer", "Must have the same number of entries as 'reciprocal_buffer' and 'time_derivative_reciprocal'.");
  for (std::size_t i = 0; i < buffers.size(); ++i)
    _variables.push_back(Variable{getOutputBufferByName(buffers[i]), getInputBufferByName(reciprocal[i]), getInputBufferByName(rate[i])});
}

void
HipForwardEulerSolver::substep()
{
  // re-evaluate the solve compute
  _compute->computeBuffer();
  forwardBuffers();
  const int64_t ns = _hip->reciprocalCount();
  for (auto & v : _variables)
  {
    const torch::Tensor u0 = v._reciprocal_buffer.contiguous(), N = v._time_derivative_reciprocal.expand(_hip->reciprocalShape()).contiguous();
    if (u0.numel() != ns || !u0.is_complex() || !N.is_complex())
      paramError("reciprocal_buffer", "expected spectra on the local reciprocal grid (", ns, " complex values)");
    torch::Tensor ubar = torch::empty_like(u0);
    const double * terms[] = {static_cast<const double *>(N.data_ptr())};
    const double coef[] = {_sub_dt};
    _hip->check(mrl_kspace_abm(_hip->ctx(), static_cast<double *>(ubar.data_ptr()), static_cast<const double *>(u0.data_ptr()), terms, coef, 1,
                               nullptr, _sub_dt, ns),
                name());
    torch::Tensor u = torch::empty(_hip->realShape(), MooseTensor::floatTensorOptions());
    _hip->check(mrl_fft_c2r(_hip->ctx(), static_cast<const double *>(ubar.data_ptr()), u.data_ptr<double>(), 1, 0), name());
    v._buffer = u;
  }
}
