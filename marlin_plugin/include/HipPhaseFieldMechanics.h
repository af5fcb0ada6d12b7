// Homogeneous small-strain elasticity coupled to a concentration field (test/tests/tensor_compute/coupled_pf_mech.i) on libmarlin_hip:
//   HipFFTQuasistaticElasticity     FFTQuasistaticElasticity     (src/tensor_computes/FFTQuasistaticElasticity.C:14-104)
//   HipFFTElasticChemicalPotential  FFTElasticChemicalPotential  (src/tensor_computes/FFTElasticChemicalPotential.C:14-61)
// The 3 x 3 system of a k-point is built and LU-solved in registers (mrl_qs_elasticity) instead of a stored [grid][3][3] complex
// tensor + batched linalg_solve.  3-D, serial, half-spectrum contexts.  The reference ships neither a test spec nor gold data for
// this input: parity of these two objects is against the oracle's restatement only (unpinned).
#pragma once

#include "TensorOperator.h"
#include "HipDomain.h"

#include <memory>

class HipFFTQuasistaticElasticity : public TensorOperatorBase
{
public:
  static InputParameters validParams();
  HipFFTQuasistaticElasticity(const InputParameters & parameters);
  virtual void computeBuffer() override;
  virtual bool supportsJIT() const override { return false; }

protected:
  std::shared_ptr<HipDomain> _hip;
  const Real _mu, _lambda, _e0;
  const torch::Tensor & _cbar;
  std::vector<torch::Tensor *> _displacements;
};

class HipFFTElasticChemicalPotential : public TensorOperator<>
{
public:
  static InputParameters validParams();
  HipFFTElasticChemicalPotential(const InputParameters & parameters);
  virtual void computeBuffer() override;
  virtual bool supportsJIT() const override { return false; }

protected:
  std::shared_ptr<HipDomain> _hip;
  const Real _mu, _lambda, _e0;
  const torch::Tensor & _cbar;
  std::vector<const torch::Tensor *> _displacements;
};
