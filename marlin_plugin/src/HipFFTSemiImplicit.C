// HipFFTSemiImplicit: FFTSemiImplicit::computeBuffer (src/tensor_timeintegrators/FFTSemiImplicit.C:43-62) on libmarlin_hip.
#include "HipFFTSemiImplicit.h"
#include "TensorProblem.h"
#include "DomainAction.h"

#include <algorithm>

registerMooseObject("MarlinApp", HipFFTSemiImplicit);

InputParameters
HipFFTSemiImplicit::validParams()
{
  InputParameters params = TensorTimeIntegrator<>::validParams();
  params.addClassDescription("Semi-implicit time integrator on libmarlin_hip (MI355X).");
  params.addRequiredParam<TensorInputBufferName>("reciprocal_buffer", "Buffer with the reciprocal of the integrated buffer");
  params.addRequiredParam<TensorInputBufferName>("linear_reciprocal", "Buffer with the reciprocal of the linear prefactor (e.g. kappa*k^2)");
  params.addRequiredParam<TensorInputBufferName>("nonlinear_reciprocal", "Buffer with the reciprocal of the non-linear contribution");
  params.addParam<unsigned int>("history_size", 1, "How many old states to use (determines time integration order).");
  return params;
}

HipFFTSemiImplicit::HipFFTSemiImplicit(const InputParameters & parameters)
  : TensorTimeIntegrator<>(parameters),
    _hip(HipDomain::get(_domain, comm())),
    _history_size(getParam<unsigned int>("history_size")),
    _reciprocal_buffer(getInputBuffer("reciprocal_buffer")),
    _linear_reciprocal(getInputBuffer("linear_reciprocal")),
    _non_linear_reciprocal(getInputBuffer("nonlinear_reciprocal")),
    _old_reciprocal_buffer(getBufferOld("reciprocal_buffer", _history_size)),
    _old_non_linear_reciprocal(getBufferOld("nonlinear_reciprocal", _history_size))
{
}

HipFFTSemiImplicit::~HipFFTSemiImplicit()
{
  if (_first)
    mrl_parsed_destroy(_first);
  if (_second)
    mrl_parsed_destroy(_second);
}

void
HipFFTSemiImplicit::build()
{
  if (_first)
    mrl_parsed_destroy(_first);
  if (_second)
    mrl_parsed_destroy(_second);
  _first = _second = nullptr;
  _built_dt = _sub_dt;
  const char * cn[] = {"dt"};
  const double cv[] = {_sub_dt};
  {
    const char * in[] = {"ubar", "N", "L"};
    const int is_complex[] = {1, 1, 0};
    _hip->check(mrl_parsed_create(_hip->ctx(), &_first, "(ubar + dt * N) / (1 - dt * L)", 3, in, is_complex, 1, cn, cv, 0, nullptr, 0, 1), name());
  }
  const char * in[] = {"ubar", "N", "No", "L"};
  const int is_complex[] = {1, 1, 1, 0};
  _hip->check(mrl_parsed_create(_hip->ctx(), &_second, "(ubar + dt / 2 * (3 * N - No)) / (1 - dt * L)", 4, in, is_complex, 1, cn, cv, 0,
                                nullptr, 0, 1),
              name());
}

void
HipFFTSemiImplicit::computeBuffer()
{
  if (!_first || _built_dt != _sub_dt)
    build();
  const auto n_old = std::min(_old_reciprocal_buffer.size(), _old_non_linear_reciprocal.size());
  const auto shape = _hip->reciprocalShape();
  const torch::Tensor u0 = _reciprocal_buffer.contiguous(), N = _non_linear_reciprocal.expand(shape).contiguous(),
                      L = _linear_reciprocal.expand(shape).contiguous();
  if (u0.numel() != _hip->reciprocalCount() || !u0.is_complex() || !N.is_complex() || L.is_complex())
    paramError("reciprocal_buffer", "expected spectra on the local reciprocal grid and a real linear operator");
  torch::Tensor ubar = torch::empty_like(u0), No;
  std::vector<const double *> ptr = {static_cast<const double *>(u0.data_ptr()), static_cast<const double *>(N.data_ptr())};
  if (n_old >= 1) // second order, FFTSemiImplicit.C:52-59
  {
    No = _old_non_linear_reciprocal[0].expand(shape).contiguous();
    ptr.push_back(static_cast<const double *>(No.data_ptr()));
  }
  ptr.push_back(L.data_ptr<double>());
  _hip->check(mrl_parsed_eval(n_old >= 1 ? _second : _first, ptr.data(), static_cast<double *>(ubar.data_ptr()), _hip->reciprocalCount(), 0.0),
              name());
  torch::Tensor u = torch::empty(_hip->realShape(), MooseTensor::floatTensorOptions());
  _hip->check(mrl_fft_c2r(_hip->ctx(), static_cast<const double *>(ubar.data_ptr()), u.data_ptr<double>(), 1, 0), name());
  _u = u; // :61
}
