// HipFFTQuasistaticElasticity / HipFFTElasticChemicalPotential on libmarlin_hip (mrl_qs_elasticity, mrl_elastic_chemical_potential).
#include "HipPhaseFieldMechanics.h"
#include "DomainAction.h"

registerMooseObject("MarlinApp", HipFFTQuasistaticElasticity);
registerMooseObject("MarlinApp", HipFFTElasticChemicalPotential);

InputParameters
HipFFTQuasistaticElasticity::validParams()
{
  InputParameters params = TensorOperatorBase::validParams();
  params.addClassDescription("FFT based monolithic homogeneous quasistatic elasticity solve on libmarlin_hip (MI355X).");
  params.addParam<std::vector<TensorOutputBufferName>>("displacements", "Displacements");
  params.addParam<TensorInputBufferName>("cbar", "FFT of concentration buffer");
  params.addRequiredParam<Real>("mu", "Lame mu");
  params.addRequiredParam<Real>("lambda", "Lame lambda");
  params.addRequiredParam<Real>("e0", "volumetric eigenstrain");
  return params;
}

HipFFTQuasistaticElasticity::HipFFTQuasistaticElasticity(const InputParameters & parameters)
  : TensorOperatorBase(parameters),
    _hip(HipDomain::get(_domain, comm())),
    _mu(getParam<Real>("mu")),
    _lambda(getParam<Real>("lambda")),
    _e0(getParam<Real>("e0")),
    _cbar(getInputBuffer("cbar"))
{
  for (const auto & name : getParam<std::vector<TensorOutputBufferName>>("displacements"))
    _displacements.push_back(&getOutputBufferByName(name));
  if (_domain.getDim() != _displacements.size())
    paramError("displacements", "Need one displacement variable per mesh dimension");
  if (_hip->dim() != 3 || _hip->parallel())
    paramError("displacements", "the library's quasistatic elasticity solve is 3-D and serial");
}

void
HipFFTQuasistaticElasticity::computeBuffer()
{
  const torch::Tensor cbar = _cbar.contiguous();
  std::vector<torch::Tensor> fresh;
  double * out[3] = {nullptr, nullptr, nullptr};
  for (std::size_t i = 0; i < _displacements.size(); ++i)
  {
    fresh.push_back(torch::empty(_hip->realShape(), MooseTensor::floatTensorOptions()));
    out[i] = fresh.back().data_ptr<double>();
  }
  _hip->check(mrl_qs_elasticity(_hip->ctx(), static_cast<const double *>(cbar.data_ptr()), _mu, _lambda, _e0, out), name());
  for (std::size_t i = 0; i < _displacements.size(); ++i)
    *_displacements[i] = fresh[i]; // FFTQuasistaticElasticity.C:101-103
}

InputParameters
HipFFTElasticChemicalPotential::validParams()
{
  InputParameters params = TensorOperator<>::validParams();
  params.addClassDescription("FFT based elastic strain energy chemical potential solve on libmarlin_hip (MI355X).");
  params.addParam<std::vector<TensorInputBufferName>>("displacements", "Displacements");
  params.addParam<TensorInputBufferName>("cbar", "FFT of concentration buffer");
  params.addRequiredParam<Real>("mu", "Lame mu");
  params.addRequiredParam<Real>("lambda", "Lame lambda");
  params.addRequiredParam<Real>("e0", "volumetric eigenstrain");
  return params;
}

HipFFTElasticChemicalPotential::HipFFTElasticChemicalPotential(const InputParameters & parameters)
  : TensorOperator<>(parameters),
    _hip(HipDomain::get(_domain, comm())),
    _mu(getParam<Real>("mu")),
    _lambda(getParam<Real>("lambda")),
    _e0(getParam<Real>("e0")),
    _cbar(getInputBuffer("cbar"))
{
  for (const auto & name : getParam<std::vector<TensorInputBufferName>>("displacements"))
    _displacements.push_back(&getInputBufferByName(name));
  if (_domain.getDim() != _displacements.size())
    paramError("displacements", "Need one displacement variable per mesh dimension");
  if (_hip->dim() != 3 || _hip->parallel())
    paramError("displacements", "the library's elastic chemical potential is 3-D and serial");
}

void
HipFFTElasticChemicalPotential::computeBuffer()
{
  const torch::Tensor cbar = _cbar.contiguous();
  std::vector<torch::Tensor> keep;
  const double * in[3] = {nullptr, nullptr, nullptr};
  for (std::size_t i = 0; i < _displacements.size(); ++i)
  {
    keep.push_back(_displacements[i]->expand(_hip->realShape()).contiguous());
    in[i] = keep.back().data_ptr<double>();
  }
  torch::Tensor out = torch::empty(_hip->reciprocalShape(), MooseTensor::complexFloatTensorOptions());
  _hip->check(mrl_elastic_chemical_potential(_hip->ctx(), static_cast<const double *>(cbar.data_ptr()), in, _mu, _lambda, _e0,
                                             static_cast<double *>(out.data_ptr())),
              name());
  _u = out;
}
