// HipAdamsBashforthMoulton -- replaces AdamsBashforthMoulton for the scalar Cahn-Hilliard system
// (include/tensor_solver/AdamsBashforthMoulton.h, SplitOperatorBase.h:27-34, TensorSolver.h).
#pragma once

#include "SplitOperatorBase.h"
#include "HipDomain.h"

#include <memory>

class HipAdamsBashforthMoulton : public SplitOperatorBase
{
public:
  static InputParameters validParams();
  HipAdamsBashforthMoulton(const InputParameters & parameters);
  ~HipAdamsBashforthMoulton()
  {
    if (_parsed)
      mrl_parsed_destroy(_parsed);
  }

  /// the whole substep loop of TensorSolver::computeBuffer (TensorSolver.C:93-109) in one library call when nothing can observe
  /// the intermediate fields, the inherited loop over substep() otherwise
  virtual void computeBuffer() override;

protected:
  virtual void substep() override;

  /// the buffers the compute group of the reference would have assigned
  void publish(const torch::Tensor & Nnew);

  std::unique_ptr<HipDomain> _hip;
  mrl_ch_params _p;
  mrl_parsed * _parsed = nullptr;
  const std::size_t _predictor_order;
  const bool _fuse_substeps;
  /// ring of predictor_order + 1 spectral arrays in the library's private layout (mrl_ch_spec_elems complex values each)
  std::vector<torch::Tensor> _ring;
  int _head = 0, _n_old = 0;
};
