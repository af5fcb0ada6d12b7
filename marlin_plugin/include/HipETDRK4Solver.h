// HipETDRK4Solver -- replaces ETDRK4Solver (include/tensor_solver/ETDRK4Solver.h, src/tensor_solver/ETDRK4Solver.C:29-115): fourth-order
// exponential time differencing (Cox-Matthews) for any number of variables.  The stage updates, the phi functions with their L dt = 0
// limits and the final combination are ONE generated kernel each (the reference materialises ~25 temporaries per variable and
// substep); the nonlinear terms come from `root_compute` as in the reference.
#pragma once

#include "SplitOperatorBase.h"
#include "HipDomain.h"

#include <memory>

class HipETDRK4Solver : public SplitOperatorBase
{
public:
  static InputParameters validParams();
  HipETDRK4Solver(const InputParameters & parameters);
  ~HipETDRK4Solver();

protected:
  virtual void substep() override;
  /// (re)generate the three kernels for the current sub_dt (a named constant of the expressions)
  void build();
  void destroy();
  /// one generated kernel over the local reciprocal grid; inputs: L (real), the rest complex
  torch::Tensor apply(mrl_parsed * kernel, const std::vector<torch::Tensor> & inputs) const;
  /// u_i = ifft(ubar_stage_i); re-evaluate the compute group; collect the nonlinear terms          ETDRK4Solver.C:35-48
  std::vector<torch::Tensor> evaluateNonlinear(const std::vector<torch::Tensor> & ubar_stage);

  std::shared_ptr<HipDomain> _hip;
  mrl_parsed * _half = nullptr;
  mrl_parsed * _full = nullptr;
  mrl_parsed * _final = nullptr;
  Real _built_dt = 0.0;
};
