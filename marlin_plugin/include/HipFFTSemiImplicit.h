// HipFFTSemiImplicit -- replaces FFTSemiImplicit (include/tensor_timeintegrators/FFTSemiImplicit.h,
// src/tensor_timeintegrators/FFTSemiImplicit.C:14-62), the legacy [TensorTimeIntegrators] form of the semi-implicit update:
//   no history : ubar = (ubar0 + dt N) / (1 - dt L)
//   history    : ubar = (ubar0 + dt/2 (3 N - N_old)) / (1 - dt L)         u = ifft(ubar)
// one generated kernel per form (the constant `dt` is part of the kernel: regenerated when the substep size changes).
#pragma once

#include "TensorTimeIntegrator.h"
#include "HipDomain.h"

#include <memory>

class HipFFTSemiImplicit : public TensorTimeIntegrator<>
{
public:
  static InputParameters validParams();
  HipFFTSemiImplicit(const InputParameters & parameters);
  ~HipFFTSemiImplicit();
  virtual void computeBuffer() override;
  virtual bool supportsJIT() const override { return false; }

protected:
  void build();

  std::shared_ptr<HipDomain> _hip;
  const unsigned int _history_size;
  const torch::Tensor & _reciprocal_buffer;
  const torch::Tensor & _linear_reciprocal;
  const torch::Tensor & _non_linear_reciprocal;
  const std::vector<torch::Tensor> & _old_reciprocal_buffer;
  const std::vector<torch::Tensor> & _old_non_linear_reciprocal;
  mrl_parsed * _first = nullptr;
  mrl_parsed * _second = nullptr;
  Real _built_dt = 0.0;
};
