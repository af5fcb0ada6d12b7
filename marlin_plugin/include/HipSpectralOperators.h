// One-transform-pair operators of a spectral solve on libmarlin_hip -- same parameter names as the objects they replace:
//   HipFFTGradient             FFTGradient             (src/tensor_computes/FFTGradient.C:14-40:       _u = ifft(fft(input) * k_direction * i))
//   HipFFTGradientSquare       FFTGradientSquare       (src/tensor_computes/FFTGradientSquare.C:14-50:  _u = factor * sum_d ifft(fft(input) k_d i)^2)
//   HipComputeDisplacements    ComputeDisplacements    (src/tensor_computes/ComputeDisplacements.C:53-107)
//   HipComputeVonMisesStress   ComputeVonMisesStress   (src/tensor_computes/ComputeVonMisesStress.C:31-66)
//   HipDeAliasingTensor        DeAliasingTensor        (src/tensor_computes/DeAliasingTensor.C:14-65: SHARP 2/3 rule, Hou-Li filter)
//   HipSwiftHohenbergLinear    SwiftHohenbergLinear    (src/tensor_computes/SwiftHohenbergLinear.C:14-40: r - alpha^2 (1 - k^2)^2)
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

/// a reciprocal-grid operator that is one generated kernel of the extra symbols (kx, ky, kz, k2) and named constants
class HipReciprocalExpression : public TensorOperator<>
{
public:
  HipReciprocalExpression(const InputParameters & parameters);
  ~HipReciprocalExpression();
  virtual void computeBuffer() override;

protected:
  /// derived constructors call this once their constants are known
  void build(const std::string & expression, const std::vector<std::string> & names, const std::vector<double> & values);

  std::shared_ptr<HipDomain> _hip;
  mrl_parsed * _parsed = nullptr;
};

class HipDeAliasingTensor : public HipReciprocalExpression
{
public:
  static InputParameters validParams();
  HipDeAliasingTensor(const InputParameters & parameters);
};

class HipSwiftHohenbergLinear : public HipReciprocalExpression
{
public:
  static InputParameters validParams();
  HipSwiftHohenbergLinear(const InputParameters & parameters);
};
