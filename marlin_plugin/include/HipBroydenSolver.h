// HipBroydenSolver -- replaces BroydenSolver (include/tensor_solver/BroydenSolver.h, src/tensor_solver/BroydenSolver.C:14-176): implicit
// time integration of coupled variables by a Broyden iteration per reciprocal grid point, with a persistent approximation M of the
// inverse Jacobian.  Host control flow as in the reference; per iteration two kernels (mrl_broyden_predict: s = -M R and the new
// iterate; mrl_broyden_update: residual, s.y, rank-one update of M, |R|^2) on field-major state arrays instead of ~25 batched-matmul /
// where / norm kernels on [grid, n, n] tensors.  Up to 32 variables.  No regression test of the reference exercises this solver:
// parity is against the oracle's restatement only (unpinned).
#pragma once

#include "SplitOperatorBase.h"
#include "IterativeTensorSolverInterface.h"
#include "HipDomain.h"

#include <memory>

class HipBroydenSolver : public SplitOperatorBase, public IterativeTensorSolverInterface
{
public:
  static InputParameters validParams();
  HipBroydenSolver(const InputParameters & parameters);

protected:
  virtual void substep() override;
  /// dense copies of every variable's reciprocal buffer, nonlinear term and linear operator + their device pointers
  struct Operands
  {
    std::vector<torch::Tensor> keep;
    std::vector<const double *> u, N, L;
  };
  Operands gather();

  std::shared_ptr<HipDomain> _hip;
  const unsigned int _max_iterations;
  const Real _relative_tolerance;
  const Real _absolute_tolerance;
  const bool _verbose;
  /// inverse-Jacobian approximation, field-major [n * n][n_spec] complex; persists over substeps (BroydenSolver.C:57-63)
  torch::Tensor _M;
};
