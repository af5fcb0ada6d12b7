// HipFFTMechanics -- replaces FFTMechanics + HyperElasticIsotropic (include/tensor_computes/FFTMechanics.h,
// src/tensor_computes/FFTMechanics.C:96-163, HyperElasticIsotropic.C:42-52, include/utils/MarlinUtils.h:55-123).
#pragma once

#include "TensorOperator.h"
#include "HipDomain.h"

#include <memory>

class HipFFTMechanics : public TensorOperator<>
{
public:
  static InputParameters validParams();
  HipFFTMechanics(const InputParameters & parameters);

  virtual void computeBuffer() override;
  /// Newton-CG: data-dependent control flow
  virtual bool supportsJIT() const override { return false; }

protected:
  std::shared_ptr<HipDomain> _hip;
  const torch::Tensor & _tF;
  const torch::Tensor & _tK;
  const torch::Tensor & _tmu;
  torch::Tensor & _tP;
  const torch::Tensor * const _applied_macroscopic_strain;
  mrl_mech_params _prm;
  const bool _verbose;
};
