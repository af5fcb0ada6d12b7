// HipETDRK4Solver: ETDRK4Solver::substep (src/tensor_solver/ETDRK4Solver.C:29-115) on libmarlin_hip.
#include "HipETDRK4Solver.h"
#include "TensorProblem.h"
#include "DomainAction.h"

registerMooseObject("MarlinApp", HipETDRK4Solver);

InputParameters
HipETDRK4Solver::validParams()
{
  InputParameters params = SplitOperatorBase::validParams();
  params.addClassDescription("Fourth-order exponential time differencing solver on libmarlin_hip (MI355X).");
  return params;
}

HipETDRK4Solver::HipETDRK4Solver(const InputParameters & parameters)
  : SplitOperatorBase(parameters), _hip(HipDomain::get(_domain, comm()))
{
  getVariables(1); // ETDRK4Solver.C:26
}

HipETDRK4Solver::~HipETDRK4Solver() { destroy(); }

void
HipETDRK4Solver::destroy()
{
  for (mrl_parsed ** p : {&_half, &_full, &_final})
    if (*p)
    {
      mrl_parsed_destroy(*p);
      *p = nullptr;
    }
}

void
HipETDRK4Solver::build()
{
  destroy();
  _built_dt = _sub_dt;
  const char * cn[] = {"dt"};
  const double cv[] = {_sub_dt};
  auto make = [&](mrl_parsed ** out, const std::string & expr, std::vector<const char *> in)
  {
    std::vector<int> is_complex(in.size(), 1);
    is_complex[0] = 0; // L is real
    _hip->check(mrl_parsed_create(_hip->ctx(), out, expr.c_str(), (int)in.size(), in.data(), is_complex.data(), 1, cn, cv, 0, nullptr,
                                  /*extra_symbols=*/0, /*reciprocal=*/1),
                name());
  };
  // ubar_b / ubar_c = expHalfLdt * ubar_n + 0.5 dt N ; ubar_d = expLdt * ubar_n + dt N               ETDRK4Solver.C:93,100,105
  make(&_half, "exp(L*dt/2.0)*ubar + 0.5*dt*N", {"L", "ubar", "N"});
  make(&_full, "exp(L*dt)*ubar + dt*N", {"L", "ubar", "N"});
  // the phi functions with their L dt == 0 limits (:75-91) and the final combination (:110-111)
  make(&_final,
       "Ldt := L*dt; E := exp(Ldt); den := Ldt*Ldt*Ldt;"
       "p1 := if(Ldt == 0.0, dt, dt*(-4.0 - 3.0*Ldt + E*(4.0 - Ldt))/den);"
       "p2 := if(Ldt == 0.0, dt*dt/2.0, dt*(2.0 + Ldt + E*(-2.0 + Ldt))/den);"
       "p3 := if(Ldt == 0.0, dt*dt/6.0, dt*(-4.0 - 3.0*Ldt - Ldt*Ldt + E*(4.0 - Ldt))/den);"
       "E*ubar + p1*N1 + 2.0*p2*(N2 + N3) + p3*N4",
       {"L", "ubar", "N1", "N2", "N3", "N4"});
}

torch::Tensor
HipETDRK4Solver::apply(mrl_parsed * kernel, const std::vector<torch::Tensor> & inputs) const
{
  std::vector<const double *> ptr;
  for (const auto & t : inputs)
    ptr.push_back(static_cast<const double *>(t.data_ptr()));
  torch::Tensor out = torch::empty(_hip->reciprocalShape(), MooseTensor::complexFloatTensorOptions());
  _hip->check(mrl_parsed_eval(kernel, ptr.data(), static_cast<double *>(out.data_ptr()), _hip->reciprocalCount(), 0.0), name());
  return out;
}

std::vector<torch::Tensor>
HipETDRK4Solver::evaluateNonlinear(const std::vector<torch::Tensor> & ubar_stage)
{
  for (std::size_t i = 0; i < _variables.size(); ++i)
  {
    torch::Tensor u = torch::empty(_hip->realShape(), MooseTensor::floatTensorOptions());
    _hip->check(mrl_fft_c2r(_hip->ctx(), static_cast<const double *>(ubar_stage[i].data_ptr()), u.data_ptr<double>(), 1, 0), name());
    _variables[i]._buffer = u;
  }
  _compute->computeBuffer();
  forwardBuffers();
  std::vector<torch::Tensor> nonlinear;
  for (auto & v : _variables)
    nonlinear.push_back(v._nonlinear_reciprocal.expand(_hip->reciprocalShape()).contiguous());
  return nonlinear;
}

void
HipETDRK4Solver::substep()
{
  if (!_half || _built_dt != _sub_dt)
    build();
  _compute->computeBuffer();
  forwardBuffers();

  const auto shape = _hip->reciprocalShape();
  const std::size_t nv = _variables.size();
  std::vector<torch::Tensor> ubar_n, linear, N1, stage(nv);
  for (auto & v : _variables)
  {
    ubar_n.push_back(v._reciprocal_buffer.contiguous());
    N1.push_back(v._nonlinear_reciprocal.expand(shape).contiguous());
    linear.push_back(v._linear_reciprocal ? v._linear_reciprocal->expand(shape).contiguous()
                                          : torch::zeros(shape, MooseTensor::floatTensorOptions())); // ETDRK4Solver.C:60-63
    if (ubar_n.back().numel() != _hip->reciprocalCount() || linear.back().is_complex())
      paramError("reciprocal_buffer", "expected spectra on the local reciprocal grid and real linear operators");
  }
  for (std::size_t i = 0; i < nv; ++i)
    stage[i] = apply(_half, {linear[i], ubar_n[i], N1[i]});
  const auto N2 = evaluateNonlinear(stage);
  for (std::size_t i = 0; i < nv; ++i)
    stage[i] = apply(_half, {linear[i], ubar_n[i], N2[i]});
  const auto N3 = evaluateNonlinear(stage);
  for (std::size_t i = 0; i < nv; ++i)
    stage[i] = apply(_full, {linear[i], ubar_n[i], N3[i]});
  const auto N4 = evaluateNonlinear(stage);
  for (std::size_t i = 0; i < nv; ++i)
  {
    const torch::Tensor ubar = apply(_final, {linear[i], ubar_n[i], N1[i], N2[i], N3[i], N4[i]});
    torch::Tensor u = torch::empty(_hip->realShape(), MooseTensor::floatTensorOptions());
    _hip->check(mrl_fft_c2r(_hip->ctx(), static_cast<const double *>(ubar.data_ptr()), u.data_ptr<double>(), 1, 0), name());
    _variables[i]._buffer = u; // ETDRK4Solver.C:112
  }
}
