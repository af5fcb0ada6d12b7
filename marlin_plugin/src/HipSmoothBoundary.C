// HipReciprocalMatDiffusion / HipReciprocalAllenCahn on libmarlin_hip.
#include "HipSmoothBoundary.h"
#include "DomainAction.h"

#include <algorithm>

registerMooseObject("MarlinApp", HipReciprocalMatDiffusion);
registerMooseObject("MarlinApp", HipReciprocalAllenCahn);

HipFusedKernel::HipFusedKernel(std::shared_ptr<HipDomain> hip, const std::string & expression, const std::vector<std::string> & inputs,
                               const std::vector<std::string> & complex_inputs, bool extra_symbols, bool reciprocal)
  : _hip(std::move(hip)), _reciprocal(reciprocal)
{
  std::vector<const char *> in;
  std::vector<int> is_complex;
  for (const auto & n : inputs)
  {
    in.push_back(n.c_str());
    is_complex.push_back(std::count(complex_inputs.begin(), complex_inputs.end(), n) ? 1 : 0);
  }
  if (mrl_parsed_create(_hip->ctx(), &_p, expression.c_str(), (int)in.size(), in.data(), is_complex.data(), 0, nullptr, nullptr, 0, nullptr,
                        extra_symbols ? 1 : 0, reciprocal ? 1 : 0) != MRL_OK)
    mooseError("marlin_hip: generated kernel '", expression, "': ", mrl_last_error(_hip->ctx()));
}

HipFusedKernel::~HipFusedKernel()
{
  if (_p)
    mrl_parsed_destroy(_p);
}

torch::Tensor
HipFusedKernel::operator()(const std::vector<torch::Tensor> & in) const
{
  const auto shape = _reciprocal ? _hip->reciprocalShape() : _hip->realShape();
  std::vector<torch::Tensor> keep;
  std::vector<const double *> ptr;
  for (const auto & t : in)
  {
    keep.push_back(t.expand(shape).contiguous());
    ptr.push_back(static_cast<const double *>(keep.back().data_ptr()));
  }
  torch::Tensor out = torch::empty(shape, mrl_parsed_is_complex(_p) ? MooseTensor::complexFloatTensorOptions() : MooseTensor::floatTensorOptions());
  _hip->check(mrl_parsed_eval(_p, ptr.data(), static_cast<double *>(out.data_ptr()), _reciprocal ? _hip->reciprocalCount() : _hip->realCount(), 0.0),
              "marlin_hip");
  return out;
}

InputParameters
HipReciprocalMatDiffusion::validParams()
{
  InputParameters params = TensorOperator<>::validParams();
  params.addClassDescription("Divergence of flux for a variable mobility in reciprocal space on libmarlin_hip (MI355X).");
  params.addRequiredParam<TensorInputBufferName>("chemical_potential", "Chemical potential buffer name");
  params.addRequiredParam<TensorInputBufferName>("mobility", "Mobility buffer name");
  params.addParam<TensorInputBufferName>("psi", "Variable to impose Neuamnn BC.");
  params.addParam<bool>("always_update_psi", false, "Set to true if the BC changes .");
  return params;
}

HipReciprocalMatDiffusion::HipReciprocalMatDiffusion(const InputParameters & parameters)
  : TensorOperator<>(parameters),
    _hip(HipDomain::get(_domain, comm())),
    _chem_pot(getInputBuffer("chemical_potential")),
    _M(getInputBuffer("mobility")),
    _psi(getInputBuffer("psi")),
    _update_psi(true),
    _always_update_psi(getParam<bool>("always_update_psi"))
{
  static const char * const k[] = {"kx", "ky", "kz"};
  static const char * const g[] = {"gx", "gy", "gz"};
  static const char * const j[] = {"jx", "jy", "jz"};
  static const char * const a[] = {"ax", "ay", "az"};
  std::string div, nof;
  std::vector<std::string> ga, ja, aa;
  for (unsigned int d = 0; d < _hip->dim(); ++d)
  {
    // _i * fft(.) * _imag                                                         ReciprocalMatDiffusion.C:49, 56
    _grad.emplace_back(new HipFusedKernel(_hip, std::string(k[d]) + "*a*i", {"a"}, {"a"}, true, true));
    div += std::string(d ? "+" : "") + k[d] + "*" + a[d];
    nof += std::string(d ? "+" : "") + g[d] + "*" + j[d];
    ga.push_back(g[d]);
    ja.push_back(j[d]);
    aa.push_back(a[d]);
  }
  _by_psi.reset(new HipFusedKernel(_hip, "if(psi>0, g/psi, 0)", {"psi", "g"}, {}, false, false));
  _flux.reset(new HipFusedKernel(_hip, "M*(psi>0)*g", {"M", "psi", "g"}, {}, false, false));
  _div.reset(new HipFusedKernel(_hip, "i*(" + div + ")", aa, aa, true, true));
  std::vector<std::string> gj = ga;
  gj.insert(gj.end(), ja.begin(), ja.end());
  _noflux.reset(new HipFusedKernel(_hip, nof, gj, {}, false, false));
  _sum.reset(new HipFusedKernel(_hip, "a+b", {"a", "b"}, {"a", "b"}, false, true));
}

torch::Tensor
HipReciprocalMatDiffusion::fft(const torch::Tensor & real) const
{
  const torch::Tensor in = real.expand(_hip->realShape()).contiguous();
  torch::Tensor out = torch::empty(_hip->reciprocalShape(), MooseTensor::complexFloatTensorOptions());
  _hip->check(mrl_fft_r2c(_hip->ctx(), in.data_ptr<double>(), static_cast<double *>(out.data_ptr()), 1, 0), name());
  return out;
}

torch::Tensor
HipReciprocalMatDiffusion::ifft(const torch::Tensor & spectrum) const
{
  torch::Tensor out = torch::empty(_hip->realShape(), MooseTensor::floatTensorOptions());
  _hip->check(mrl_fft_c2r(_hip->ctx(), static_cast<const double *>(spectrum.data_ptr()), out.data_ptr<double>(), 1, 0), name());
  return out;
}

void
HipReciprocalMatDiffusion::computeBuffer()
{
  const unsigned int dim = _hip->dim();
  if (_update_psi || _always_update_psi) // grad(psi) / psi inside the domain, :44-53
  {
    const torch::Tensor psibar = fft(_psi);
    _grad_psi_by_psi.clear();
    for (unsigned int d = 0; d < dim; ++d)
      _grad_psi_by_psi.push_back((*_by_psi)({_psi, ifft((*_grad[d])({psibar}))}));
    _update_psi = false;
  }
  const torch::Tensor mubar = fft(_chem_pot);
  std::vector<torch::Tensor> J, Jbar;
  for (unsigned int d = 0; d < dim; ++d) // J = M (psi > 0) grad(mu), :55-60
  {
    J.push_back((*_flux)({_M, _psi, ifft((*_grad[d])({mubar}))}));
    Jbar.push_back(fft(J.back()));
  }
  const torch::Tensor div_J_hat = (*_div)(Jbar);
  std::vector<torch::Tensor> in = _grad_psi_by_psi;
  in.insert(in.end(), J.begin(), J.end());
  const torch::Tensor no_flux_hat = fft((*_noflux)(in));
  _u = (*_sum)({div_J_hat, no_flux_hat}); // :62-65
}

InputParameters
HipReciprocalAllenCahn::validParams()
{
  InputParameters params = TensorOperator<>::validParams();
  params.addClassDescription("Allen-Cahn bulk driving force masked using psi on libmarlin_hip (MI355X).");
  params.addRequiredParam<TensorInputBufferName>("dF_chem_deta", "Driving force buffer name");
  params.addRequiredParam<TensorInputBufferName>("L", "Allen-Cahn mobility buffer name");
  params.addRequiredParam<TensorInputBufferName>("psi", "Variable to impose Neumann BC.");
  params.addParam<bool>("always_update_psi", false, "Set to true if the BC changes .");
  return params;
}

HipReciprocalAllenCahn::HipReciprocalAllenCahn(const InputParameters & parameters)
  : TensorOperator<>(parameters),
    _hip(HipDomain::get(_domain, comm())),
    _dF_chem_deta(getInputBuffer("dF_chem_deta")),
    _L(getInputBuffer("L")),
    _psi(getInputBuffer("psi")),
    _rate(_hip, "if(psi>0, -1*L*dF, 0)", {"psi", "L", "dF"}, {}, false, false)
{
}

void
HipReciprocalAllenCahn::computeBuffer()
{
  const torch::Tensor rate = _rate({_psi, _L, _dF_chem_deta}); // where(psi > 0, -1 * L * dF_chem_deta, 0), ReciprocalAllenCahn.C:46-49
  torch::Tensor out = torch::empty(_hip->reciprocalShape(), MooseTensor::complexFloatTensorOptions());
  _hip->check(mrl_fft_r2c(_hip->ctx(), rate.data_ptr<double>(), static_cast<double *>(out.data_ptr()), 1, 0), name());
  _u = out;
}
