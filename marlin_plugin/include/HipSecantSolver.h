// HipSecantSolver -- replaces SecantSolver (include/tensor_solver/SecantSolver.h, src/tensor_solver/SecantSolver.C:14-185): implicit
// secant iteration per variable.  The control flow stays on the host as in the reference (compute group -> residual -> convergence
// test, one `.item()`-like read per iteration); the reciprocal-space work of an iteration is ONE kernel (mrl_secant_iterate:
// residual, secant update with c10's complex division, damped new iterate, |R|^2 and |du|^2 partial sums) where the reference runs
// ~15 ATen kernels.
#pragma once

#include "SplitOperatorBase.h"
#include "IterativeTensorSolverInterface.h"
#include "HipDomain.h"

#include <memory>

class HipSecantSolver : public SplitOperatorBase, public IterativeTensorSolverInterface
{
public:
  static InputParameters validParams();
  HipSecantSolver(const InputParameters & parameters);

protected:
  virtual void substep() override;
  /// u = ifft(ubar) into a fresh tensor, rebinding the variable's handle
  void inverse(Variable & v, const torch::Tensor & ubar);

  std::shared_ptr<HipDomain> _hip;
  const unsigned int _max_iterations;
  const Real _relative_tolerance;
  const Real _absolute_tolerance;
  const bool _verbose;
  const Real _damping;
  const Real _dt_epsilon;
};
