// HipFFTGradient / HipFFTGradientSquare / HipComputeDisplacements / HipComputeVonMisesStress on libmarlin_hip.
#include "HipSpectralOperators.h"
#include "DomainAction.h"

#include <algorithm>
#include <cmath>

registerMooseObject("MarlinApp", HipFFTGradient);
registerMooseObject("MarlinApp", HipFFTGradientSquare);
registerMooseObject("MarlinApp", HipComputeDisplacements);
registerMooseObject("MarlinApp", HipComputeVonMisesStress);
registerMooseObject("MarlinApp", HipDeAliasingTensor);
registerMooseObject("MarlinApp", HipSwiftHohenbergLinear);

namespace
{
const char * const k_axis[] = {"kx", "ky", "kz"};

/// forward transform of a real field into a fresh spectrum, or the spectrum itself
torch::Tensor
spectrumOf(const HipDomain & hip, const torch::Tensor & input, bool is_reciprocal, const std::string & who)
{
  if (is_reciprocal)
    return input.contiguous();
  const torch::Tensor in = input.contiguous();
  torch::Tensor out = torch::empty(hip.reciprocalShape(), MooseTensor::complexFloatTensorOptions());
  hip.check(mrl_fft_r2c(hip.ctx(), in.data_ptr<double>(), static_cast<double *>(out.data_ptr()), 1, 0), who);
  return out;
}
}

InputParameters
HipFFTGradient::validParams()
{
  InputParameters params = TensorOperator<>::validParams();
  params.addClassDescription("Tensor gradient on libmarlin_hip (MI355X).");
  params.addRequiredParam<TensorInputBufferName>("input", "Input buffer name");
  params.addParam<bool>("input_is_reciprocal", false, "Input buffer is already in reciprocal space");
  params.addRequiredParam<MooseEnum>("direction", MooseEnum("X=0 Y=1 Z=2"), "Which axis to take the gradient along.");
  return params;
}

HipFFTGradient::HipFFTGradient(const InputParameters & parameters)
  : TensorOperator<>(parameters),
    _hip(HipDomain::get(_domain, comm())),
    _input(getInputBuffer("input")),
    _input_is_reciprocal(getParam<bool>("input_is_reciprocal"))
{
  // the k-space product abar * k_direction * i as one generated kernel (left to right, as FFTGradient.C:38-39 multiplies)
  const int direction = getParam<MooseEnum>("direction");
  if (direction < 0 || direction >= (int)_hip->dim())
    paramError("direction", "the domain has ", _hip->dim(), " dimension(s)");
  const std::string expr = std::string("abar*") + k_axis[direction] + "*i";
  const char * in[] = {"abar"};
  const int is_complex[] = {1};
  if (mrl_parsed_create(_hip->ctx(), &_parsed, expr.c_str(), 1, in, is_complex, 0, nullptr, nullptr, 0, nullptr, /*extra_symbols=*/1,
                        /*reciprocal=*/1) != MRL_OK)
    paramError("direction", mrl_last_error(_hip->ctx()));
}

HipFFTGradient::~HipFFTGradient()
{
  if (_parsed)
    mrl_parsed_destroy(_parsed);
}

void
HipFFTGradient::computeBuffer()
{
  const torch::Tensor abar = spectrumOf(*_hip, _input, _input_is_reciprocal, name());
  torch::Tensor gbar = torch::empty_like(abar);
  const double * in[] = {static_cast<const double *>(abar.data_ptr())};
  _hip->check(mrl_parsed_eval(_parsed, in, static_cast<double *>(gbar.data_ptr()), _hip->reciprocalCount(), 0.0), name());
  torch::Tensor out = torch::empty(_hip->realShape(), MooseTensor::floatTensorOptions());
  _hip->check(mrl_fft_c2r(_hip->ctx(), static_cast<const double *>(gbar.data_ptr()), out.data_ptr<double>(), 1, 0), name());
  _u = out;
}

InputParameters
HipFFTGradientSquare::validParams()
{
  InputParameters params = TensorOperator<>::validParams();
  params.addClassDescription("Square of the tensor gradient on libmarlin_hip (MI355X).");
  params.addRequiredParam<TensorInputBufferName>("input", "Input buffer name");
  params.addParam<bool>("input_is_reciprocal", false, "Input buffer is already in reciprocal space");
  params.addParam<Real>("factor", 1.0, "Prefactor to the gradient square");
  return params;
}

HipFFTGradientSquare::HipFFTGradientSquare(const InputParameters & parameters)
  : TensorOperator<>(parameters),
    _hip(HipDomain::get(_domain, comm())),
    _input(getInputBuffer("input")),
    _input_is_reciprocal(getParam<bool>("input_is_reciprocal"))
{
  static const char * const g[] = {"gx", "gy", "gz"};
  const int dim = (int)_hip->dim();
  const char * in[] = {"abar"};
  const int is_complex[] = {1};
  std::string sq;
  for (int d = 0; d < dim; ++d)
  {
    mrl_parsed * p = nullptr;
    const std::string expr = std::string("abar*") + k_axis[d] + "*i";
    if (mrl_parsed_create(_hip->ctx(), &p, expr.c_str(), 1, in, is_complex, 0, nullptr, nullptr, 0, nullptr, 1, 1) != MRL_OK)
      paramError("input", mrl_last_error(_hip->ctx()));
    _grad.push_back(p);
    sq += std::string(d ? "+" : "") + g[d] + "*" + g[d]; // FFTGradientSquare.C:41-48: sqr(grad_x) + sqr(grad_y) + sqr(grad_z)
  }
  const Real factor = getParam<Real>("factor");
  if (factor != 1.0)
    sq = "(" + sq + ")*f";
  const int is_real[] = {0, 0, 0};
  const char * cn[] = {"f"};
  if (mrl_parsed_create(_hip->ctx(), &_square, sq.c_str(), dim, g, is_real, 1, cn, &factor, 0, nullptr, 0, 0) != MRL_OK)
    paramError("factor", mrl_last_error(_hip->ctx()));
}

HipFFTGradientSquare::~HipFFTGradientSquare()
{
  for (auto * p : _grad)
    mrl_parsed_destroy(p);
  if (_square)
    mrl_parsed_destroy(_square);
}

void
HipFFTGradientSquare::computeBuffer()
{
  // one forward transform, per axis one generated k-space kernel + inverse transform, one generated kernel for the squares
  const torch::Tensor abar = spectrumOf(*_hip, _input, _input_is_reciprocal, name());
  std::vector<torch::Tensor> grad;
  std::vector<const double *> gp;
  for (auto * p : _grad)
  {
    torch::Tensor gbar = torch::empty_like(abar);
    const double * in[] = {static_cast<const double *>(abar.data_ptr())};
    _hip->check(mrl_parsed_eval(p, in, static_cast<double *>(gbar.data_ptr()), _hip->reciprocalCount(), 0.0), name());
    grad.push_back(torch::empty(_hip->realShape(), MooseTensor::floatTensorOptions()));
    _hip->check(mrl_fft_c2r(_hip->ctx(), static_cast<const double *>(gbar.data_ptr()), grad.back().data_ptr<double>(), 1, 0), name());
    gp.push_back(grad.back().data_ptr<double>());
  }
  torch::Tensor out = torch::empty(_hip->realShape(), MooseTensor::floatTensorOptions());
  _hip->check(mrl_parsed_eval(_square, gp.data(), out.data_ptr<double>(), _hip->realCount(), 0.0), name());
  _u = out;
}

InputParameters
HipComputeDisplacements::validParams()
{
  InputParameters params = TensorOperator<>::validParams();
  params.addClassDescription("Compute updated displacements from the deformation gradient tensor on libmarlin_hip (MI355X).");
  params.addRequiredParam<TensorInputBufferName>("F", "Deformation gradient tensor.");
  return params;
}

HipComputeDisplacements::HipComputeDisplacements(const InputParameters & parameters)
  : TensorOperator<>(parameters), _hip(HipDomain::get(_domain, comm())), _deformation_gradient_tensor(getInputBuffer("F"))
{
  if (_hip->parallel())
    mooseError(name(), ": the nodal displacement field is interpolated over the global grid: serial domains only");
}

void
HipComputeDisplacements::computeBuffer()
{
  if (!_deformation_gradient_tensor.defined())
    return;
  // node displacements [(n_0 + 1) ... (n_{D-1} + 1)][D], value-major (ComputeDisplacements.C:100-106)
  const torch::Tensor F = _deformation_gradient_tensor.contiguous();
  std::vector<int64_t> shape;
  for (const auto n : _hip->realShape())
    shape.push_back(n + 1);
  shape.push_back(_hip->dim());
  torch::Tensor out = torch::empty(shape, MooseTensor::floatTensorOptions());
  _hip->check(mrl_mech_displacements(_hip->ctx(), F.data_ptr<double>(), out.data_ptr<double>()), name());
  _u = out;
}

InputParameters
HipComputeVonMisesStress::validParams()
{
  InputParameters params = TensorOperator<>::validParams();
  params.addClassDescription("Compute vonMises stress on libmarlin_hip (MI355X).");
  params.addParam<TensorInputBufferName>("stress", "stress", "Stress tensor.");
  return params;
}

HipComputeVonMisesStress::HipComputeVonMisesStress(const InputParameters & parameters)
  : TensorOperator<>(parameters), _hip(HipDomain::get(_domain, comm())), _stress(getInputBuffer("stress"))
{
}

void
HipComputeVonMisesStress::computeBuffer()
{
  if (!_stress.defined())
    return;
  const torch::Tensor s = _stress.contiguous();
  torch::Tensor out = torch::empty(_hip->realShape(), MooseTensor::floatTensorOptions());
  _hip->check(mrl_mech_von_mises(_hip->ctx(), s.data_ptr<double>(), out.data_ptr<double>()), name());
  _u = out;
}

HipReciprocalExpression::HipReciprocalExpression(const InputParameters & parameters)
  : TensorOperator<>(parameters), _hip(HipDomain::get(_domain, comm()))
{
}

HipReciprocalExpression::~HipReciprocalExpression()
{
  if (_parsed)
    mrl_parsed_destroy(_parsed);
}

void
HipReciprocalExpression::build(const std::string & expression, const std::vector<std::string> & names, const std::vector<double> & values)
{
  std::vector<const char *> cn;
  for (const auto & n : names)
    cn.push_back(n.c_str());
  if (mrl_parsed_create(_hip->ctx(), &_parsed, expression.c_str(), 0, nullptr, nullptr, (int)cn.size(), cn.data(), values.data(), 0,
                        nullptr, /*extra_symbols=*/1, /*reciprocal=*/1) != MRL_OK)
    mooseError(name(), ": ", mrl_last_error(_hip->ctx()));
}

void
HipReciprocalExpression::computeBuffer()
{
  torch::Tensor out = torch::empty(_hip->reciprocalShape(), MooseTensor::floatTensorOptions());
  _hip->check(mrl_parsed_eval(_parsed, nullptr, out.data_ptr<double>(), _hip->reciprocalCount(), _time), name());
  _u = out;
}

InputParameters
HipDeAliasingTensor::validParams()
{
  InputParameters params = TensorOperator<>::validParams();
  params.addClassDescription("Create a de-aliasing filter on libmarlin_hip (MI355X).");
  params.addRequiredParam<MooseEnum>("method", MooseEnum("SHARP HOULI"), "Filter: SHARP (2/3 rule) or HOULI (Hou-Li exponential)");
  params.addParam<Real>("p", 16, "Hou-Li filter exponent");
  params.addParam<Real>("alpha", 36, "Hou-Li filter pre-factor");
  return params;
}

HipDeAliasingTensor::HipDeAliasingTensor(const InputParameters & parameters) : HipReciprocalExpression(parameters)
{
  // maximum |frequency| of this rank's reciprocal axes (DeAliasingTensor.C:40-43: max(abs(_i)) ...)
  double mx[3] = {0.0, 0.0, 0.0};
  for (unsigned int d = 0; d < 3; ++d)
    for (const double k : _hip->reciprocalAxis(d))
      mx[d] = std::max(mx[d], std::fabs(k));
  const int method = getParam<MooseEnum>("method");
  if (method == 0) // SHARP, :47-52
    build("if((abs(kx) > cx) | (abs(ky) > cy) | (abs(kz) > cz), 0, 1)", {"cx", "cy", "cz"}, {2 * mx[0] / 3, 2 * mx[1] / 3, 2 * mx[2] / 3});
  else // HOULI, :54-60
    build("exp(0-alpha*((abs(kx)/mx)^p + (abs(ky)/my)^p + (abs(kz)/mz)^p))", {"alpha", "p", "mx", "my", "mz"},
          {getParam<Real>("alpha"), getParam<Real>("p"), mx[0] ? mx[0] : 1.0, mx[1] ? mx[1] : 1.0, mx[2] ? mx[2] : 1.0});
}

InputParameters
HipSwiftHohenbergLinear::validParams()
{
  InputParameters params = TensorOperator<>::validParams();
  params.addClassDescription("Swift-Hohenberg linear operator r - alpha^2 (1 - k^2)^2 on libmarlin_hip (MI355X).");
  params.addRequiredParam<Real>("r", "Control parameter");
  params.addRequiredParam<Real>("alpha", "Wave number scale");
  return params;
}

HipSwiftHohenbergLinear::HipSwiftHohenbergLinear(const InputParameters & parameters) : HipReciprocalExpression(parameters)
{
  const Real alpha = getParam<Real>("alpha");
  build("r-aa*(1-k2)*(1-k2)", {"r", "aa"}, {getParam<Real>("r"), alpha * alpha}); // SwiftHohenbergLinear.C:38
}
