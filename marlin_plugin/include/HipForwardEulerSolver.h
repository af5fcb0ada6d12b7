// HipForwardEulerSolver -- replaces ForwardEulerSolver (include/tensor_solver/ForwardEulerSolver.h, ExplicitSolverBase.h,
// src/tensor_solver/ExplicitSolverBase.C:13-51, ForwardEulerSolver.C:28-38): u = ifft(ubar + sub_dt * du/dt_bar) per variable after
// the compute group; with no variables it only drives `root_compute` and forwards buffers (mech3d.i:81-89).
#pragma once

#include "TensorSolver.h"
#include "HipDomain.h"

#include <memory>

class HipForwardEulerSolver : public TensorSolver
{
public:
  static InputParameters validParams();
  HipForwardEulerSolver(const InputParameters & parameters);

protected:
  virtual void substep() override;

  struct Variable
  {
    torch::Tensor & _buffer;
    const torch::Tensor & _reciprocal_buffer;
    const torch::Tensor & _time_derivative_reciprocal;
  };
  std::shared_ptr<HipDomain> _hip;
  std::vector<Variable> _variables;
};
