// HipForwardFFT / HipInverseFFT / HipParsedCompute / HipReciprocalLaplacian(Square)Factor: Marlin's spectral compute objects on
// libmarlin_hip (mrl_fft_r2c / mrl_fft_c2r, mrl_parsed_*, mrl_reciprocal_laplacian).
#include "HipSpectralComputes.h"
#include "DomainAction.h"

#include <algorithm>
#include <set>

registerMooseObject("MarlinApp", HipForwardFFT);
registerMooseObject("MarlinApp", HipInverseFFT);
registerMooseObject("MarlinApp", HipParsedCompute);
registerMooseObject("MarlinApp", HipReciprocalLaplacianFactor);
registerMooseObject("MarlinApp", HipReciprocalLaplacianSquareFactor);

template <bool forward>
InputParameters
HipPerformFFTTempl<forward>::validParams()
{
  InputParameters params = TensorOperator<>::validParams();
  params.addClassDescription("FFT of a buffer on libmarlin_hip (MI355X).");
  params.addParam<TensorInputBufferName>("input", "Input buffer name");
  return params;
}

template <bool forward>
HipPerformFFTTempl<forward>::HipPerformFFTTempl(const InputParameters & parameters)
  : TensorOperator<>(parameters), _hip(HipDomain::get(_domain, comm())), _input(getInputBuffer("input"))
{
}

template <bool forward>
void
HipPerformFFTTempl<forward>::computeBuffer()
{
  // trailing value dimensions ([..., 3, 3]) are the batch of a value-major transform (DomainAction.C:853-867: dim = the grid axes)
  const torch::Tensor in = _input.contiguous();
  const int64_t grid = forward ? _hip->realCount() : _hip->reciprocalCount();
  if (in.numel() % grid != 0 || in.is_complex() == forward)
    paramError("input", "expected a ", forward ? "real" : "complex", " tensor on the local grid (", grid, " points), got ", in.numel(),
               " entries");
  const int64_t batch = in.numel() / grid;
  std::vector<int64_t> shape = forward ? _hip->reciprocalShape() : _hip->realShape();
  for (int64_t d = _hip->dim(); d < in.dim(); ++d)
    shape.push_back(in.size(d));
  torch::Tensor out = torch::empty(shape, forward ? MooseTensor::complexFloatTensorOptions() : MooseTensor::floatTensorOptions());
  if constexpr (forward)
    _hip->check(mrl_fft_r2c(_hip->ctx(), in.data_ptr<double>(), static_cast<double *>(out.data_ptr()), batch, /*value-major*/ 1), name());
  else
    _hip->check(mrl_fft_c2r(_hip->ctx(), static_cast<const double *>(in.data_ptr()), out.data_ptr<double>(), batch, 1), name());
  _u = out;
}

template class HipPerformFFTTempl<true>;
template class HipPerformFFTTempl<false>;

InputParameters
HipParsedCompute::validParams()
{
  InputParameters params = TensorOperator<>::validParams();
  params.addClassDescription("ParsedCompute on libmarlin_hip: the expression becomes one generated HIP kernel.");
  params.addRequiredParam<std::string>("expression", "Parsed expression");
  params.addParam<std::vector<TensorInputBufferName>>("inputs", {}, "Buffer names used in the expression");
  params.addParam<std::vector<TensorInputBufferName>>("derivatives", {}, "List of inputs to take the derivative w.r.t. (or none)");
  params.addParam<bool>("extra_symbols", false, "Provide i, kx, ky, kz, k2, x, y, z, t, pi and e.");
  params.addParam<std::vector<std::string>>("constant_names", {}, "Vector of constants used in the parsed function");
  params.addParam<std::vector<std::string>>("constant_expressions", {}, "Vector of values for the constants in constant_names");
  params.addParam<std::string>("expand", "NONE", "REAL | RECIPROCAL | NONE: the grid an expression of extra symbols only is evaluated on");
  return params;
}

HipParsedCompute::HipParsedCompute(const InputParameters & parameters)
  : TensorOperator<>(parameters),
    _hip(HipDomain::get(_domain, comm())),
    _names(getParam<std::vector<TensorInputBufferName>>("inputs")),
    _constant_names(getParam<std::vector<std::string>>("constant_names")),
    _extra_symbols(getParam<bool>("extra_symbols")),
    _expand(getParam<std::string>("expand"))
{
  if (std::set<std::string>(_names.begin(), _names.end()).size() != _names.size())
    paramError("inputs", "Duplicate buffer name.");
  for (const auto & n : _names)
    _params.push_back(&getInputBufferByName(n));
  const auto & text = getParam<std::vector<std::string>>("constant_expressions");
  if (text.size() != _constant_names.size())
    paramError("constant_names", "The parameter vectors constant_names (size ", _constant_names.size(),
               ") and constant_expressions (size ", text.size(), ") must have equal length.");
  // a constant is a number or an expression of the constants before it (ParsedCompute.C:100-123): the latter is evaluated by the
  // library too, as a one-point kernel
  for (std::size_t i = 0; i < text.size(); ++i)
  {
    std::size_t used = 0;
    double value = 0;
    try
    {
      value = std::stod(text[i], &used);
    }
    catch (...)
    {
      used = 0;
    }
    if (used != text[i].size())
    {
      std::vector<const char *> cn;
      for (std::size_t j = 0; j < i; ++j)
        cn.push_back(_constant_names[j].c_str());
      mrl_parsed * c = nullptr;
      if (mrl_parsed_create(_hip->ctx(), &c, text[i].c_str(), 0, nullptr, nullptr, (int)i, cn.data(), _constant_values.data(), 0,
                            nullptr, /*extra_symbols=*/0, 0) != MRL_OK)
        paramError("constant_expressions", "Invalid constant expression\n", text[i], "\n", mrl_last_error(_hip->ctx()));
      torch::Tensor one = torch::empty({1}, MooseTensor::floatTensorOptions());
      const int rc = mrl_parsed_eval(c, nullptr, one.data_ptr<double>(), 1, 0.0);
      mrl_parsed_destroy(c);
      _hip->check(rc, name());
      value = one.item<double>();
    }
    _constant_values.push_back(value);
  }
  for (const auto & d : getParam<std::vector<TensorInputBufferName>>("derivatives"))
    if (std::find(_names.begin(), _names.end(), d) == _names.end())
      paramError("derivatives", "Derivative w.r.t `", d, "` was requested, but it is not listed in `inputs`.");
}

HipParsedCompute::~HipParsedCompute()
{
  if (_parsed)
    mrl_parsed_destroy(_parsed);
}

void
HipParsedCompute::build()
{
  std::vector<const char *> in, cn, dn;
  std::vector<int> is_complex;
  bool any_complex = false;
  for (std::size_t i = 0; i < _names.size(); ++i)
  {
    if (!_params[i]->defined())
      mooseError(name(), ": input buffer '", _names[i], "' is not initialised");
    in.push_back(_names[i].c_str());
    is_complex.push_back(_params[i]->is_complex());
    any_complex = any_complex || _params[i]->is_complex();
  }
  for (const auto & c : _constant_names)
    cn.push_back(c.c_str());
  const auto & derivatives = getParam<std::vector<TensorInputBufferName>>("derivatives");
  for (const auto & d : derivatives)
    dn.push_back(d.c_str());
  // the grid: reciprocal if an input is a spectrum or `expand = RECIPROCAL` says so, else real
  _reciprocal = _expand == "RECIPROCAL" || (_expand != "REAL" && any_complex);
  if (mrl_parsed_create(_hip->ctx(), &_parsed, getParam<std::string>("expression").c_str(), (int)in.size(), in.data(),
                        is_complex.data(), (int)cn.size(), cn.data(), _constant_values.data(), (int)dn.size(), dn.data(),
                        _extra_symbols ? 1 : 0, _reciprocal ? 1 : 0) != MRL_OK)
    paramError("expression", "Invalid function: ", mrl_last_error(_hip->ctx()));
}

void
HipParsedCompute::computeBuffer()
{
  if (!_parsed)
    build();
  const int64_t count = _reciprocal ? _hip->reciprocalCount() : _hip->realCount();
  std::vector<torch::Tensor> keep;
  std::vector<const double *> in;
  const auto shape = _reciprocal ? _hip->reciprocalShape() : _hip->realShape();
  for (const auto * t : _params)
  {
    keep.push_back(t->expand(shape).contiguous()); // (inputs smaller than the grid broadcast, as in the reference's JIT graph)
    in.push_back(static_cast<const double *>(keep.back().data_ptr()));
  }
  torch::Tensor out = torch::empty(shape, mrl_parsed_is_complex(_parsed) ? MooseTensor::complexFloatTensorOptions()
                                                                         : MooseTensor::floatTensorOptions());
  _hip->check(mrl_parsed_eval(_parsed, in.data(), static_cast<double *>(out.data_ptr()), count, _time), name());
  _u = out;
}

template <int power>
InputParameters
HipReciprocalLaplacianTempl<power>::validParams()
{
  InputParameters params = TensorOperator<>::validParams();
  params.addClassDescription(power == 1 ? "Reciprocal space Laplacian IC on libmarlin_hip." : "Reciprocal space Laplacian squared IC on libmarlin_hip.");
  params.addParam<Real>("factor", 1.0, "Prefactor");
  return params;
}

template <int power>
HipReciprocalLaplacianTempl<power>::HipReciprocalLaplacianTempl(const InputParameters & parameters)
  : TensorOperator<>(parameters), _hip(HipDomain::get(_domain, comm())), _factor(getParam<Real>("factor"))
{
}

template <int power>
void
HipReciprocalLaplacianTempl<power>::computeBuffer()
{
  // -k^2 f (ReciprocalLaplacianFactor.C:28-31) or k^2 k^2 f (ReciprocalLaplacianSquareFactor.C:28-32), full local reciprocal grid
  torch::Tensor out = torch::empty(_hip->reciprocalShape(), MooseTensor::floatTensorOptions());
  _hip->check(mrl_reciprocal_laplacian(_hip->ctx(), power, _factor, out.data_ptr<double>()), name());
  _u = out;
}

template class HipReciprocalLaplacianTempl<1>;
template class HipReciprocalLaplacianTempl<2>;
