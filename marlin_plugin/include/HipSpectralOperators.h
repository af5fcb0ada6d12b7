// One-transform-pair operators of a spectral solve on libmarlin_hip -- same parameter names as the objects they replace:
//   HipFFTGradient             FFTGradient             (src/tensor_computes/FFTGradient.C:14-40:       _u = ifft(fft(input) * k_direction * i))
//   HipFFTGradientSquare       FFTGradientSquare       (src/tensor_computes/FFTGradientSquare.C:14-50:  _u = factor * sum_d ifft(fft(input) k_d i)^2)
//   HipComputeDisplacements    ComputeDisplacements    (src/tensor_computes/ComputeDisplacements.C:53-107)
//   HipComputeVonMisesStress   ComputeVonMisesStress   (src/tensor_computes/ComputeVonMisesStress.C:31-66)
#pragma once

#include "TensorOperator.h"
#include "HipDomain.h"

#include <memory>

class HipFFTGradient : public TensorOperator<>
{
public:
  static InputParameters validParams();
  HipFFTGradient(const InputParameters & parameters);
  ~HipFFTGradient();
  virtual void computeBuffer() override;
  virtual bool supportsJIT() const override { return false; }

protected:
  std::shared_ptr<HipDomain> _hip;
  const torch::Tensor & _input;
  const bool _input_is_reciprocal;
  mrl_parsed * _parsed = nullptr;
};

class HipFFTGradientSquare : public TensorOperator<>
{
public:
  static InputParameters validParams();
  HipFFTGradientSquare(const InputParameters & parameters);
  ~HipFFTGradientSquare();
  virtual void computeBuffer() override;
  virtual bool supportsJIT() const override { return false; }

protected:
  std::shared_ptr<HipDomain> _hip;
  const torch::Tensor & _input;
  const bool _input_is_reciprocal;
  std::vector<mrl_parsed *> _grad;
  mrl_parsed * _square = nullptr;
};

class HipComputeDisplacements : public TensorOperator<>
{
public:
  static InputParameters validParams();
  HipComputeDisplacements(const InputParameters & parameters);
  virtual void computeBuffer() override;
  virtual bool supportsJIT() const override { return false; }

protected:
  std::shared_ptr<HipDomain> _hip;
  const torch::Tensor & _deformation_gradient_tensor;
};

class HipComputeVonMisesStress : public TensorOperator<>
{
public:
  static InputParameters validParams();
  HipComputeVonMisesStress(const InputParameters & parameters);
  virtual void computeBuffer() override;

protected:
  std::shared_ptr<HipDomain> _hip;
  const torch::Tensor & _stress;
};
