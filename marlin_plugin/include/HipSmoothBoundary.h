// Smooth-boundary-method operators of the KKS inputs on libmarlin_hip (test/tests/kks/KKS_no_flux_bc.i):
//   HipReciprocalMatDiffusion  ReciprocalMatDiffusion  (src/tensor_computes/ReciprocalMatDiffusion.C:14-66): divergence of the flux
//                              M grad(mu) with a no-flux condition on the boundary of {psi > 0}, in reciprocal space
//   HipReciprocalAllenCahn     ReciprocalAllenCahn     (src/tensor_computes/ReciprocalAllenCahn.C:14-50): fft(where(psi > 0, -L dF/deta, 0))
// Every pointwise step is one generated kernel (the reference: one ATen kernel and one full-size temporary per operator).
#pragma once

#include "TensorOperator.h"
#include "HipDomain.h"

#include <memory>

/// one generated pointwise kernel over explicit tensors (mrl_parsed_* with pointer inputs)
class HipFusedKernel
{
public:
  HipFusedKernel(std::shared_ptr<HipDomain> hip, const std::string & expression, const std::vector<std::string> & inputs,
                 const std::vector<std::string> & complex_inputs, bool extra_symbols, bool reciprocal);
  ~HipFusedKernel();
  HipFusedKernel(const HipFusedKernel &) = delete;
  torch::Tensor operator()(const std::vector<torch::Tensor> & in) const;

private:
  std::shared_ptr<HipDomain> _hip;
  const bool _reciprocal;
  mrl_parsed * _p = nullptr;
};

class HipReciprocalMatDiffusion : public TensorOperator<>
{
public:
  static InputParameters validParams();
  HipReciprocalMatDiffusion(const InputParameters & parameters);
  virtual void computeBuffer() override;
  virtual bool supportsJIT() const override { return false; }

protected:
  torch::Tensor fft(const torch::Tensor & real) const;
  torch::Tensor ifft(const torch::Tensor & spectrum) const;

  std::shared_ptr<HipDomain> _hip;
  const torch::Tensor & _chem_pot;
  const torch::Tensor & _M;
  const torch::Tensor & _psi;
  bool _update_psi;
  const bool _always_update_psi;
  std::vector<std::unique_ptr<HipFusedKernel>> _grad;
  std::unique_ptr<HipFusedKernel> _by_psi, _flux, _div, _noflux, _sum;
  std::vector<torch::Tensor> _grad_psi_by_psi;
};

class HipReciprocalAllenCahn : public TensorOperator<>
{
public:
  static InputParameters validParams();
  HipReciprocalAllenCahn(const InputParameters & parameters);
  virtual void computeBuffer() override;
  virtual bool supportsJIT() const override { return false; }

protected:
  std::shared_ptr<HipDomain> _hip;
  const torch::Tensor & _dF_chem_deta;
  const torch::Tensor & _L;
  const torch::Tensor & _psi;
  HipFusedKernel _rate;
};
