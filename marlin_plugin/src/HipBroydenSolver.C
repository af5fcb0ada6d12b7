// HipBroydenSolver: BroydenSolver::substep (src/tensor_solver/BroydenSolver.C:63-176) on libmarlin_hip.
#include "HipBroydenSolver.h"
#include "TensorProblem.h"
#include "DomainAction.h"

#include <cmath>

registerMooseObject("MarlinApp", HipBroydenSolver);

InputParameters
HipBroydenSolver::validParams()
{
  InputParameters params = SplitOperatorBase::validParams();
  params.addClassDescription("Implicit Broyden solver time integration on libmarlin_hip (MI355X).");
  params.addParam<unsigned int>("substeps", 1, "solver substeps per time step.");
  params.addParam<unsigned int>("max_iterations", 5, "Maximum number of Broyden iterations.");
  params.addParam<Real>("relative_tolerance", 1e-9, "Convergence tolerance.");
  params.addParam<Real>("absolute_tolerance", 1e-9, "Convergence tolerance.");
  params.addParam<Real>("initial_jacobian_guess", 1.0, "Factor for the initial inverse jacobian guess.");
  params.addParam<bool>("verbose", false, "Show convergence history.");
  return params;
}

HipBroydenSolver::HipBroydenSolver(const InputParameters & parameters)
  : SplitOperatorBase(parameters),
    IterativeTensorSolverInterface(),
    _hip(HipDomain::get(_domain, comm())),
    _max_iterations(getParam<unsigned int>("max_iterations")),
    _relative_tolerance(getParam<Real>("relative_tolerance")),
    _absolute_tolerance(getParam<Real>("absolute_tolerance")),
    _verbose(getParam<bool>("verbose"))
{
  getVariables(0); // no history required, BroydenSolver.C:47
  const int64_t n = (int64_t)_variables.size();
  if (n > 32)
    paramError("buffer", "at most 32 coupled variables");
  _M = torch::empty({n * n, _hip->reciprocalCount()}, MooseTensor::complexFloatTensorOptions());
  _hip->check(mrl_broyden_init(_hip->ctx(), (int)n, getParam<Real>("initial_jacobian_guess"), static_cast<double *>(_M.data_ptr()),
                               _hip->reciprocalCount()),
              name());
}

HipBroydenSolver::Operands
HipBroydenSolver::gather()
{
  Operands o;
  const auto shape = _hip->reciprocalShape();
  for (auto & v : _variables)
  {
    o.keep.push_back(v._reciprocal_buffer.contiguous());
    if (o.keep.back().numel() != _hip->reciprocalCount() || !o.keep.back().is_complex())
      paramError("reciprocal_buffer", "expected ", _hip->reciprocalCount(), " complex values (the local reciprocal grid)");
    o.u.push_back(static_cast<const double *>(o.keep.back().data_ptr()));
    o.keep.push_back(v._nonlinear_reciprocal.expand(shape).contiguous());
    o.N.push_back(static_cast<const double *>(o.keep.back().data_ptr()));
    if (v._linear_reciprocal)
    {
      o.keep.push_back(v._linear_reciprocal->expand(shape).contiguous());
      o.L.push_back(o.keep.back().data_ptr<double>());
    }
    else
      o.L.push_back(nullptr);
  }
  return o;
}

void
HipBroydenSolver::substep()
{
  const int n = (int)_variables.size();
  const int64_t ns = _hip->reciprocalCount();
  const auto copt = MooseTensor::complexFloatTensorOptions();

  _compute->computeBuffer();
  forwardBuffers();
  Operands cur = gather();
  // u_old: the reciprocal buffers at the start of the substep (handle copies, BroydenSolver.C:71-77)
  std::vector<torch::Tensor> old_keep;
  std::vector<const double *> u_old;
  for (auto & v : _variables)
  {
    old_keep.push_back(v._reciprocal_buffer.contiguous());
    u_old.push_back(static_cast<const double *>(old_keep.back().data_ptr()));
  }
  torch::Tensor R = torch::empty({(int64_t)n, ns}, copt), S = torch::empty({(int64_t)n, ns}, copt);
  double ss = 0.0;
  _hip->check(mrl_broyden_residual(_hip->ctx(), n, cur.u.data(), cur.N.data(), cur.L.data(), nullptr, _sub_dt,
                                   static_cast<double *>(R.data_ptr()), &ss, ns),
              name());
  const Real R0norm = std::sqrt(ss);
  Real Rnorm = R0norm;
  for (_iterations = 0; _iterations < _max_iterations; ++_iterations)
  {
    if (std::isnan(Rnorm))
      mooseError("NAN!"); // BroydenSolver.C:107-108
    if (Rnorm < _absolute_tolerance || Rnorm / R0norm < _relative_tolerance)
    {
      _is_converged = true;
      return;
    }
    if (_verbose)
      _console << _iterations << " |R|=" << Rnorm << std::endl;
    std::vector<torch::Tensor> out;
    std::vector<double *> outp;
    for (int i = 0; i < n; ++i)
    {
      out.push_back(torch::empty(_hip->reciprocalShape(), copt));
      outp.push_back(static_cast<double *>(out.back().data_ptr()));
    }
    // s = -M R, u <- u + 0.5 s (the reference hard-wires the step, :124-133)
    _hip->check(mrl_broyden_predict(_hip->ctx(), n, static_cast<const double *>(_M.data_ptr()), static_cast<const double *>(R.data_ptr()),
                                    cur.u.data(), 0.5, static_cast<double *>(S.data_ptr()), outp.data(), ns),
                name());
    for (int i = 0; i < n; ++i)
    {
      torch::Tensor u = torch::empty(_hip->realShape(), MooseTensor::floatTensorOptions());
      _hip->check(mrl_fft_c2r(_hip->ctx(), outp[i], u.data_ptr<double>(), 1, 0), name());
      _variables[i]._buffer = u;
    }
    _compute->computeBuffer();
    forwardBuffers();
    cur = gather();
    _hip->check(mrl_broyden_update(_hip->ctx(), n, static_cast<double *>(_M.data_ptr()), static_cast<double *>(R.data_ptr()),
                                   static_cast<const double *>(S.data_ptr()), cur.u.data(), cur.N.data(), cur.L.data(), u_old.data(),
                                   _sub_dt, &ss, ns),
                name());
    Rnorm = std::sqrt(ss);
  }
  _console << "Broyden solve did not converge within the maximum number of iterations.\n";
  _is_converged = false;
}
