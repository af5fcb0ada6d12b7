// HipAdamsBashforthMoulton: AdamsBashforthMoulton::substep (src/tensor_solver/AdamsBashforthMoulton.C:60-101) and the compute group it
// re-evaluates (ParsedCompute mu = f'(c), PerformFFT, ReciprocalLaplacianFactor Mbar, ReciprocalLaplacianSquareFactor kappabarbar)
// as ONE library call per substep -- or one per solver call -- on the tensors Marlin owns.
#include "HipAdamsBashforthMoulton.h"
#include "TensorProblem.h"
#include "DomainAction.h"

registerMooseObject("MarlinApp", HipAdamsBashforthMoulton);

InputParameters
HipAdamsBashforthMoulton::validParams()
{
  InputParameters params = SplitOperatorBase::validParams();
  params.addClassDescription("Adams-Bashforth semi-implicit Cahn-Hilliard solver on libmarlin_hip (MI355X).");
  params.addParam<unsigned int>("substeps", 1, "semi-implicit substeps per time step.");
  params.addRangeCheckedParam<std::size_t>(
      "predictor_order", 2, "predictor_order > 0 & predictor_order <= 5", "Order of the Adams-Bashforth predictor.");
  // copied verbatim from the [mu] ParsedCompute block of the input (cahnhilliard.i:61-69)
  params.addRequiredParam<std::string>("expression", "Free energy density f(c); mu = df/dc is derived symbolically");
  params.addParam<std::vector<std::string>>("constant_names", {}, "Named constants of the expression");
  params.addParam<std::vector<Real>>("constant_expressions", {}, "... and their values");
  params.addRequiredParam<Real>("mobility", "Factor of the ReciprocalLaplacianFactor block (Mbar = -k^2 M)");
  params.addRequiredParam<Real>("kappa_factor", "Factor of the ReciprocalLaplacianSquareFactor block (Lbar = k^4 kappa)");
  params.addParam<bool>("fuse_substeps", true, "Hand the whole substep loop of a solver call to the library (mrl_ch_substeps)");
  return params;
}

HipAdamsBashforthMoulton::HipAdamsBashforthMoulton(const InputParameters & parameters)
  : SplitOperatorBase(parameters),
    _hip(std::make_unique<HipDomain>(_domain, comm())),
    _predictor_order(getParam<std::size_t>("predictor_order") - 1), // AdamsBashforthMoulton.C:48
    _fuse_substeps(getParam<bool>("fuse_substeps"))
{
  getVariables(_predictor_order); // history depth, AdamsBashforthMoulton.C:55-56
  if (_variables.size() != 1)
    paramError("buffer", "HipAdamsBashforthMoulton solves one variable; use AdamsBashforthMoulton with mrl_kspace_abm otherwise");

  const auto names = getParam<std::vector<std::string>>("constant_names");
  const auto values = getParam<std::vector<Real>>("constant_expressions");
  if (names.size() != values.size())
    paramError("constant_names", "Need one value per named constant");
  std::vector<const char *> cn;
  for (const auto & s : names)
    cn.push_back(s.c_str());
  const char * inputs[] = {"c"};
  const int is_complex[] = {0};
  const char * wrt[] = {"c"};
  // symbolic derivative with the reference's rules (MarlinExpressionParser.C:50-235), compiled into the forward z pass
  _hip->check(mrl_parsed_create(_hip->ctx(), &_parsed, getParam<std::string>("expression").c_str(), 1, inputs, is_complex,
                                (int)cn.size(), cn.data(), values.data(), 1, wrt, /*extra_symbols=*/0, /*space=*/0),
              name());
  _p = mrl_ch_params{};
  _p.family = MRL_FE_PARSED;
  _p.parsed = _parsed;
  _p.mobility = getParam<Real>("mobility");
  _p.kappa = getParam<Real>("kappa_factor");

  const auto copt = MooseTensor::complexFloatTensorOptions();
  for (std::size_t i = 0; i < _predictor_order + 1; ++i)
    _ring.push_back(torch::zeros({mrl_ch_spec_elems(_hip->ctx())}, copt));
}

void
HipAdamsBashforthMoulton::publish(const torch::Tensor & Nnew)
{
  // what ComputeGroup would have assigned (Mbarmubar); consumers see the dense values through a strided view
  int64_t plane = 0, row = 0;
  mrl_ch_spec_layout(_hip->ctx(), &plane, &row);
  const auto shape = _domain.getReciprocalShape();
  auto & v = _variables[0];
  const_cast<torch::Tensor &>(v._nonlinear_reciprocal) =
      shape.size() == 3 ? torch::as_strided(Nnew, {shape[0], shape[1], shape[2]}, {plane, row, 1}) : Nnew;
}

void
HipAdamsBashforthMoulton::substep()
{
  auto & v = _variables[0];
  const auto & hist = v._old_nonlinear_reciprocal;
  // AdamsBashforthMoulton.C:75,88-91: a changed time step size restarts the predictor at first order
  const int order = (int)std::min<std::size_t>(_substep < _predictor_order && _dt != _dt_old ? 0 : hist.size(), _predictor_order);
  const torch::Tensor c_in = v._buffer.contiguous();
  torch::Tensor c_out = torch::empty_like(c_in);
  torch::Tensor Nnew = torch::empty({mrl_ch_spec_elems(_hip->ctx())}, MooseTensor::complexFloatTensorOptions());
  std::vector<const double *> old(order);
  std::vector<torch::Tensor> keep(order);
  for (int i = 0; i < order; ++i)
  {
    keep[i] = hist[i].contiguous();
    old[i] = static_cast<const double *>(keep[i].data_ptr());
  }
  _hip->check(mrl_ch_substep(_hip->ctx(), &_p, c_in.data_ptr<double>(), c_out.data_ptr<double>(),
                             static_cast<double *>(Nnew.data_ptr()), old.data(), order, _sub_dt, nullptr, nullptr, MRL_CARRY_NONE),
              name());
  publish(Nnew);
  v._buffer = c_out; // AdamsBashforthMoulton.C:101: rebinding the handle
}

void
HipAdamsBashforthMoulton::computeBuffer()
{
  // per-substep outputs or other objects with a history need every intermediate field: keep Marlin's own loop
  if (!_fuse_substeps || _substeps < 2)
  {
    TensorSolver::computeBuffer();
    return;
  }
  auto & v = _variables[0];
  const torch::Tensor c_in = v._buffer.contiguous();
  torch::Tensor c_out = torch::empty_like(c_in);
  std::vector<double *> ring;
  for (auto & t : _ring)
    ring.push_back(static_cast<double *>(t.data_ptr()));
  // TensorSolver.C:105-106: advanceState between substeps, a no-op for the buffers while timeStep() <= 1
  int advance = _tensor_problem.timeStep() > 1 ? 1 : 0;
  if (_dt != _dt_old)
    advance |= MRL_SUBSTEPS_DT_CHANGED;
  if (_tensor_problem.timeStep() > 1 && _n_old < (int)_predictor_order) // advanceState between two solver calls
  {
    _head = (_head + 1) % (int)_ring.size();
    _n_old += 1;
  }
  else if (_tensor_problem.timeStep() > 1)
    _head = (_head + 1) % (int)_ring.size();
  _hip->check(mrl_ch_substeps(_hip->ctx(), &_p, c_in.data_ptr<double>(), c_out.data_ptr<double>(), ring.data(), (int)_ring.size(),
                              &_head, &_n_old, (int)_predictor_order, (int)_substeps, advance, _sub_dt, nullptr),
              name());
  publish(_ring[(_head + 1) % _ring.size()]);
  v._buffer = c_out;
  _sub_time += _substeps * _sub_dt;
}
