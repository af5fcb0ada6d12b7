// HipSecantSolver: SecantSolver::substep (src/tensor_solver/SecantSolver.C:60-185) on libmarlin_hip.
#include "HipSecantSolver.h"
#include "TensorProblem.h"
#include "DomainAction.h"

#include <cmath>

registerMooseObject("MarlinApp", HipSecantSolver);

InputParameters
HipSecantSolver::validParams()
{
  InputParameters params = SplitOperatorBase::validParams();
  params.addClassDescription("Implicit secant solver time integration on libmarlin_hip (MI355X).");
  params.addParam<unsigned int>("substeps", 1, "secant solver substeps per time step.");
  params.addParam<unsigned int>("max_iterations", 30, "Maximum number of secant solver iteration.");
  params.addParam<Real>("relative_tolerance", 1e-9, "Convergence tolerance.");
  params.addParam<Real>("absolute_tolerance", 1e-9, "Convergence tolerance.");
  params.addParam<Real>("damping", 1.0, "Damping factor for the update step.");
  params.addParam<Real>("dt_epsilon", 1e-4, "Semi-implicit stable timestep to bootstrap secant solve.");
  params.addParam<bool>("verbose", false, "Show convergence history.");
  return params;
}

HipSecantSolver::HipSecantSolver(const InputParameters & parameters)
  : SplitOperatorBase(parameters),
    IterativeTensorSolverInterface(),
    _hip(HipDomain::get(_domain, comm())),
    _max_iterations(getParam<unsigned int>("max_iterations")),
    _relative_tolerance(getParam<Real>("relative_tolerance")),
    _absolute_tolerance(getParam<Real>("absolute_tolerance")),
    _verbose(getParam<bool>("verbose")),
    _damping(getParam<Real>("damping")),
    _dt_epsilon(getParam<Real>("dt_epsilon"))
{
  getVariables(0); // no history required, SecantSolver.C:44
  if (_variables.size() > 1)
    paramWarning("buffer", "The secant solver only work well for uncoupled variables.");
}

void
HipSecantSolver::inverse(Variable & v, const torch::Tensor & ubar)
{
  torch::Tensor u = torch::empty(_hip->realShape(), MooseTensor::floatTensorOptions());
  _hip->check(mrl_fft_c2r(_hip->ctx(), static_cast<const double *>(ubar.data_ptr()), u.data_ptr<double>(), 1, 0), name());
  v._buffer = u;
}

void
HipSecantSolver::substep()
{
  const std::size_t n = _variables.size();
  const int64_t ns = _hip->reciprocalCount();
  const auto shape = _hip->reciprocalShape();
  const auto copt = MooseTensor::complexFloatTensorOptions();
  std::vector<torch::Tensor> u_old(n), Rprev(n), uprev(n), linear(n);
  std::vector<Real> R0norm(n);
  auto cptr = [](const torch::Tensor & t) { return static_cast<const double *>(t.data_ptr()); };

  // initial residual and semi-implicit bootstrap guess                                     SecantSolver.C:73-101
  _compute->computeBuffer();
  forwardBuffers();
  for (std::size_t i = 0; i < n; ++i)
  {
    auto & v = _variables[i];
    const torch::Tensor u = v._reciprocal_buffer.contiguous(), N = v._nonlinear_reciprocal.expand(shape).contiguous();
    if (u.numel() != ns || !u.is_complex())
      paramError("reciprocal_buffer", "expected ", ns, " complex values (the local reciprocal grid), got ", u.numel());
    if (v._linear_reciprocal)
      linear[i] = v._linear_reciprocal->expand(shape).contiguous();
    Rprev[i] = torch::empty(shape, copt);
    torch::Tensor guess = torch::empty(shape, copt);
    double ss = 0.0;
    _hip->check(mrl_secant_begin(_hip->ctx(), cptr(u), cptr(N), v._linear_reciprocal ? linear[i].data_ptr<double>() : nullptr, _sub_dt,
                                 _dt_epsilon, static_cast<double *>(Rprev[i].data_ptr()), static_cast<double *>(guess.data_ptr()), &ss, ns),
                name());
    R0norm[i] = std::sqrt(ss);
    uprev[i] = u; // handle copies, :85,:90
    u_old[i] = u;
    inverse(v, guess);
    if (_verbose)
      _console << "|R0|=" << R0norm[i] << std::endl;
  }
  applyPredictors(); // :100

  // secant iterations                                                                      :112-165
  bool all_converged = false;
  for (_iterations = 0; _iterations < _max_iterations; ++_iterations)
  {
    _compute->computeBuffer();
    forwardBuffers();
    all_converged = true;
    for (std::size_t i = 0; i < n; ++i)
    {
      auto & v = _variables[i];
      const torch::Tensor u = v._reciprocal_buffer.contiguous(), N = v._nonlinear_reciprocal.expand(shape).contiguous();
      torch::Tensor unew = torch::empty(shape, copt);
      double ss[2] = {0.0, 0.0};
      _hip->check(mrl_secant_iterate(_hip->ctx(), cptr(u), cptr(N), v._linear_reciprocal ? linear[i].data_ptr<double>() : nullptr,
                                     cptr(u_old[i]), cptr(uprev[i]), static_cast<double *>(Rprev[i].data_ptr()), _sub_dt, _damping,
                                     static_cast<double *>(unew.data_ptr()), ss, ns),
                  name());
      uprev[i] = u;
      inverse(v, unew);
      const Real Rnorm = std::sqrt(ss[0]);
      if (_verbose)
        _console << _iterations << " |du| = " << std::sqrt(ss[1]) << " |R|=" << Rnorm << std::endl;
      if (std::isnan(Rnorm)) // :152-159
      {
        all_converged = false;
        _iterations = _max_iterations;
        _console << "NaN detected, aborting solve.\n";
        break;
      }
      all_converged = all_converged && (Rnorm < _absolute_tolerance || Rnorm / R0norm[i] < _relative_tolerance);
    }
    if (all_converged)
    {
      _is_converged = true;
      break;
    }
  }
  if (!all_converged) // restore the old solution, :167-184
  {
    _console << "Solve not converged.\n";
    for (std::size_t i = 0; i < n; ++i)
      inverse(_variables[i], u_old[i]);
    _is_converged = false;
  }
}
