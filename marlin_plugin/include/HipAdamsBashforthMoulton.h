// HipAdamsBashforthMoulton -- replaces AdamsBashforthMoulton (include/tensor_solver/AdamsBashforthMoulton.h, SplitOperatorBase.h:27-34,
// TensorSolver.h): predictor orders 1-5, Adams-Moulton corrector, any number of variables.
//
// Two ways to run, selected by the input file:
//  * `expression = <free energy f(c)>` + `mobility` + `kappa_factor`: the scalar Cahn-Hilliard system INCLUDING its compute group
//    (ParsedCompute mu = f'(c), ForwardFFT x 2, Mbar * mubar) as one library call per substep (mrl_ch_substep) or per solver call
//    (mrl_ch_substeps, `fuse_substeps = true`, the default) -- the headline path;
//  * no `expression`: the reference's structure -- `root_compute` is evaluated (any mix of Marlin's own and Hip* compute objects), the
//    k-space update of every variable is mrl_kspace_abm and the inverse transform mrl_fft_c2r.
#pragma once

#include "SplitOperatorBase.h"
#include "HipDomain.h"

#include <memory>

class HipAdamsBashforthMoulton : public SplitOperatorBase
{
public:
  static InputParameters validParams();
  HipAdamsBashforthMoulton(const InputParameters & parameters);
  ~HipAdamsBashforthMoulton();

  /// the whole substep loop of TensorSolver::computeBuffer (TensorSolver.C:93-109) in one library call when nothing can observe
  /// the intermediate fields, the inherited loop over substep() otherwise
  virtual void computeBuffer() override;

protected:
  virtual void substep() override;

  void substepCahnHilliard();
  void substepGeneric();
  /// the right-hand side of one variable: ubar0 + sum coef_i N_i   (`none`: the corrector found order 0 for it, AdamsBashforthMoulton.C:155-156)
  struct Terms
  {
    torch::Tensor ubar0;
    std::vector<torch::Tensor> N;
    std::vector<double> coef;
    bool none = false;
  };
  /// all variables: ubar_k = rhs_k / (1 - sub_dt L_k), u_k = ifft(ubar_k) (AdamsBashforthMoulton.C:94-101, 158-172); the coupled
  /// solver overrides it with one dense solve per k-point
  virtual void solve(const std::vector<Terms> & rhs);
  /// a dense copy of a reciprocal-space operand with the expected number of values
  torch::Tensor spectral(const torch::Tensor & t, const char * param) const;
  /// u = ifft(ubar) into a fresh tensor, rebinding the variable's handle
  void inverse(Variable & v, const torch::Tensor & ubar);
  /// a fresh array in the solver-private spectral layout
  torch::Tensor newSpectral() const;
  /// the buffers the compute group of the reference would have assigned
  void publish(const torch::Tensor & Nnew, const torch::Tensor & mu);

  static constexpr std::size_t max_order = 5;

  std::shared_ptr<HipDomain> _hip;
  const std::size_t _predictor_order; ///< user value - 1, as AdamsBashforthMoulton.C:48
  const std::size_t _corrector_order; ///< user value - 1, :49
  const std::size_t _corrector_steps;
  const bool _cahn_hilliard;
  const bool _fuse_substeps;
  const bool _verbose;
  mrl_ch_params _p;
  mrl_parsed * _parsed = nullptr;
  torch::Tensor * const _mu_out;
  /// what `linear_reciprocal` means to this solver ("the coupled solver" in an error message)
  virtual const char * flavour() const { return "HipAdamsBashforthMoulton"; }
  /// fused loop: ring of predictor_order spectral arrays in the library's private layout (mrl_ch_spec_elems complex values each)
  std::vector<torch::Tensor> _ring;
  int _head = 0, _n_old = 0;
  bool _have_new = false; ///< slot (_head + 1) holds an Nhat that the next advanceState turns into history
};

/// HipAdamsBashforthMoultonCoupled -- replaces AdamsBashforthMoultonCoupled (include/tensor_solver/AdamsBashforthMoultonCoupled.h,
/// src/tensor_solver/AdamsBashforthMoultonCoupled.C:84-272): the same right-hand sides, then ONE dense N x N solve per k-point with
/// the off-diagonal linear operators (mrl_kspace_coupled: LU with partial pivoting in registers up to 8 variables, on a workspace
/// up to 32).  `reference_quirks = true` (default) reproduces what the reference's gold files pin: the operator assembled transposed
/// and the imaginary part of the right-hand side dropped (:160-183); false solves the system as written.
class HipAdamsBashforthMoultonCoupled : public HipAdamsBashforthMoulton
{
public:
  static InputParameters validParams();
  HipAdamsBashforthMoultonCoupled(const InputParameters & parameters);

protected:
  virtual void solve(const std::vector<Terms> & rhs) override;

  std::vector<const torch::Tensor *> _L; ///< row-major N x N table of linear operator buffers (nullptr = zero)
  const int _flags;
};
