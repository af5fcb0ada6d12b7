// HipAdamsBashforthMoulton: AdamsBashforthMoulton::substep (src/tensor_solver/AdamsBashforthMoulton.C:60-177) on libmarlin_hip.
#include "HipAdamsBashforthMoulton.h"
#include "TensorProblem.h"
#include "DomainAction.h"

#include <algorithm>

registerMooseObject("MarlinApp", HipAdamsBashforthMoulton);

InputParameters
HipAdamsBashforthMoulton::validParams()
{
  InputParameters params = SplitOperatorBase::validParams();
  params.addClassDescription("Adams-Bashforth-Moulton semi-implicit solver on libmarlin_hip (MI355X).");
  params.addParam<unsigned int>("substeps", 1, "semi-implicit substeps per time step.");
  params.addRangeCheckedParam<std::size_t>(
      "predictor_order", 2, "predictor_order > 0 & predictor_order <= 5", "Order of the Adams-Bashforth predictor.");
  params.addRangeCheckedParam<std::size_t>(
      "corrector_order", 2, "corrector_order > 0 & corrector_order <= 5", "Order of the Adams-Moulton corrector.");
  params.addParam<std::size_t>("corrector_steps", 0, "Number the Adams-Moulton corrector steps to take.");
  // Cahn-Hilliard form: copied verbatim from the [mu] ParsedCompute block and the two Laplacian factor blocks (cahnhilliard.i:33-69)
  params.addParam<std::string>("expression", "Free energy density f(c); mu = df/dc is derived symbolically. Selects the fused path");
  params.addParam<std::vector<std::string>>("constant_names", {}, "Named constants of the expression");
  params.addParam<std::vector<Real>>("constant_expressions", {}, "... and their values");
  params.addParam<Real>("mobility", "Factor of the ReciprocalLaplacianFactor block (Mbar = -k^2 M)");
  params.addParam<Real>("kappa_factor", "Factor of the ReciprocalLaplacianSquareFactor block (Lbar = k^4 kappa)");
  params.addParam<TensorOutputBufferName>("chemical_potential", "Buffer that receives mu = f'(c) (what the [mu] block would hold)");
  params.addParam<bool>("fuse_substeps", true, "Hand the whole substep loop of a solver call to the library (mrl_ch_substeps)");
  params.addParam<bool>("verbose", false, "Print the predictor order of every substep.");
  return params;
}

HipAdamsBashforthMoulton::HipAdamsBashforthMoulton(const InputParameters & parameters)
  : SplitOperatorBase(parameters),
    _hip(HipDomain::get(_domain, comm())),
    _predictor_order(getParam<std::size_t>("predictor_order") - 1),
    _corrector_order(getParam<std::size_t>("corrector_order") - 1),
    _corrector_steps(getParam<std::size_t>("corrector_steps")),
    _cahn_hilliard(isParamValid("expression")),
    _fuse_substeps(getParam<bool>("fuse_substeps")),
    _verbose(getParam<bool>("verbose")),
    _mu_out(isParamValid("chemical_potential") ? &getOutputBuffer("chemical_potential") : nullptr)
{
  if (_predictor_order >= max_order)
    paramError("predictor_order", "predictor_order > 0 & predictor_order <= 5");
  if (_corrector_order >= max_order)
    paramError("corrector_order", "corrector_order > 0 & corrector_order <= 5");
  // history consistent with the chosen orders, AdamsBashforthMoulton.C:55-56
  getVariables(_cahn_hilliard ? _predictor_order : std::max(_predictor_order, _corrector_order));
  if (!_cahn_hilliard)
    return;

  if (_variables.size() != 1)
    paramError("buffer", "`expression` selects the scalar Cahn-Hilliard path: one variable (leave it out to integrate several)");
  if (_corrector_steps)
    paramError("corrector_steps", "the fused Cahn-Hilliard path is predictor-only; leave `expression` out to use the corrector");
  if (!isParamValid("mobility") || !isParamValid("kappa_factor"))
    paramError("expression", "`mobility` and `kappa_factor` are needed with `expression`");
  const auto & names = getParam<std::vector<std::string>>("constant_names");
  const auto & values = getParam<std::vector<Real>>("constant_expressions");
  if (names.size() != values.size())
    paramError("constant_names", "Need one value per named constant");
  std::vector<const char *> cn;
  for (const auto & s : names)
    cn.push_back(s.c_str());
  // the variable's name in the expression is the name of its buffer (the `inputs = c` of the [mu] block)
  const std::string cname = getParam<std::vector<TensorOutputBufferName>>("buffer")[0];
  const char * inputs[] = {cname.c_str()};
  const int is_complex[] = {0};
  // symbolic derivative with the reference's rules (MarlinExpressionParser.C:50-235), compiled into the forward z pass
  _hip->check(mrl_parsed_create(_hip->ctx(), &_parsed, getParam<std::string>("expression").c_str(), 1, inputs, is_complex,
                                (int)cn.size(), cn.data(), values.data(), 1, inputs, /*extra_symbols=*/0, /*space=*/0),
              name());
  _p = mrl_ch_params{};
  _p.family = MRL_FE_PARSED;
  _p.parsed = _parsed;
  _p.mobility = getParam<Real>("mobility");
  _p.kappa = getParam<Real>("kappa_factor");

  if (_fuse_substeps)
    for (std::size_t i = 0; i < _predictor_order + 1; ++i)
      _ring.push_back(newSpectral().zero_());
}

HipAdamsBashforthMoulton::~HipAdamsBashforthMoulton()
{
  if (_parsed)
    mrl_parsed_destroy(_parsed);
}

torch::Tensor
HipAdamsBashforthMoulton::newSpectral() const
{
  return torch::empty({mrl_ch_spec_elems(_hip->ctx())}, MooseTensor::complexFloatTensorOptions());
}

void
HipAdamsBashforthMoulton::publish(const torch::Tensor & Nnew, const torch::Tensor & mu)
{
  // what ComputeGroup would have assigned (Mbarmubar): the dense values through a strided view of the private layout; the next
  // advanceState moves this handle into the history (TensorBuffer.h:62-79), where substepCahnHilliard() finds it again
  const_cast<torch::Tensor &>(_variables[0]._nonlinear_reciprocal) = _hip->spectralView(Nnew);
  if (_mu_out)
    *_mu_out = mu;
}

void
HipAdamsBashforthMoulton::substep()
{
  if (_cahn_hilliard)
    substepCahnHilliard();
  else
    substepGeneric();
}

void
HipAdamsBashforthMoulton::substepCahnHilliard()
{
  auto & v = _variables[0];
  const auto & hist = v._old_nonlinear_reciprocal;
  // AdamsBashforthMoulton.C:75,88-91: a changed time step size restarts the predictor at first order
  const std::size_t order = std::min(_substep < _predictor_order && _dt != _dt_old ? 0 : hist.size(), _predictor_order);
  if (_verbose)
    _console << name() << ": substep " << _substep << " order " << order << '\n';
  const torch::Tensor c_in = v._buffer.contiguous();
  torch::Tensor c_out = torch::empty_like(c_in);
  torch::Tensor Nnew = newSpectral();
  torch::Tensor mu = _mu_out ? torch::empty_like(c_in) : torch::Tensor();
  std::vector<const double *> old(order);
  std::vector<torch::Tensor> keep(order);
  for (std::size_t i = 0; i < order; ++i)
  {
    // history entries are the views publish() created; a tensor that some other object assigned to the buffer is repacked
    if (_hip->isSpectralView(hist[i]))
      keep[i] = hist[i];
    else
    {
      keep[i] = _hip->spectralView(newSpectral());
      keep[i].copy_(hist[i]);
    }
    old[i] = static_cast<const double *>(keep[i].data_ptr());
  }
  _hip->check(mrl_ch_substep(_hip->ctx(), &_p, c_in.data_ptr<double>(), c_out.data_ptr<double>(),
                             static_cast<double *>(Nnew.data_ptr()), old.data(), (int)order, _sub_dt, nullptr,
                             _mu_out ? mu.data_ptr<double>() : nullptr, MRL_CARRY_NONE),
              name());
  publish(Nnew, mu);
  v._buffer = c_out; // AdamsBashforthMoulton.C:101: rebinding the handle
}

void
HipAdamsBashforthMoulton::computeBuffer()
{
  // per-substep outputs or other objects with a history need every intermediate field: keep Marlin's own loop
  if (!_cahn_hilliard || !_fuse_substeps)
  {
    TensorSolver::computeBuffer();
    return;
  }
  _sub_time = _time;           // TensorSolver.C:95
  _sub_dt = _dt / _substeps;   // TensorSolver.C:96
  auto & v = _variables[0];
  const torch::Tensor c_in = v._buffer.contiguous();
  torch::Tensor c_out = torch::empty_like(c_in);
  torch::Tensor mu = _mu_out ? torch::empty_like(c_in) : torch::Tensor();
  std::vector<double *> ring;
  for (auto & t : _ring)
    ring.push_back(static_cast<double *>(t.data_ptr()));
  // MOOSE called advanceState between the previous solver call and this one; TensorSolver.C:105-106 calls it between substeps.
  // Both are no-ops for the buffers while timeStep() <= 1 (TensorProblem.C:455): all substeps of the first step are AB1.
  const bool advancing = _tensor_problem.timeStep() > 1;
  if (advancing && _have_new)
  {
    _head = (_head + 1) % (int)_ring.size();
    _n_old = std::min<int>(_n_old + 1, (int)_predictor_order);
  }
  int advance = advancing ? MRL_SUBSTEPS_ADVANCE : 0;
  if (_dt != _dt_old)
    advance |= MRL_SUBSTEPS_DT_CHANGED; // AdamsBashforthMoulton.C:75
  if (_verbose) // what the call below is asked to do: order = min(dt_changed && k < pred ? 0 : n_old, pred), n_old growing when advancing
    for (unsigned int k = 0, n_old = _n_old; k < _substeps; ++k)
    {
      _console << name() << ": substep " << k << " order "
               << std::min<std::size_t>(_dt != _dt_old && k < _predictor_order ? 0 : n_old, _predictor_order) << '\n';
      if (advancing && n_old < _predictor_order)
        ++n_old;
    }
  // the ABI takes the user-facing order (1 = AB1); _predictor_order holds it minus one
  _hip->check(mrl_ch_substeps(_hip->ctx(), &_p, c_in.data_ptr<double>(), c_out.data_ptr<double>(), ring.data(), (int)_ring.size(),
                              &_head, &_n_old, (int)_predictor_order + 1, (int)_substeps, advance, _sub_dt,
                              _mu_out ? mu.data_ptr<double>() : nullptr),
              name());
  _have_new = true;
  publish(_ring[(_head + 1) % _ring.size()], mu);
  v._buffer = c_out;
  _substep = _substeps;
  _sub_time += _substeps * _sub_dt; // TensorSolver.C:108, once per substep
}

torch::Tensor
HipAdamsBashforthMoulton::spectral(const torch::Tensor & t, const char * param) const
{
  const torch::Tensor d = t.contiguous();
  if (d.numel() != _hip->reciprocalCount() || !d.is_complex())
    paramError(param, "expected ", _hip->reciprocalCount(), " complex values (the local reciprocal grid), got ", d.numel());
  return d;
}

void
HipAdamsBashforthMoulton::inverse(Variable & v, const torch::Tensor & ubar)
{
  torch::Tensor u = torch::empty(_hip->realShape(), MooseTensor::floatTensorOptions());
  _hip->check(mrl_fft_c2r(_hip->ctx(), static_cast<const double *>(ubar.data_ptr()), u.data_ptr<double>(), 1, 0), name());
  v._buffer = u; // AdamsBashforthMoulton.C:101
}

void
HipAdamsBashforthMoulton::solve(const std::vector<Terms> & rhs)
{
  const int64_t n_spec = _hip->reciprocalCount();
  for (std::size_t k = 0; k < _variables.size(); ++k)
  {
    if (rhs[k].none)
      continue; // AdamsBashforthMoulton.C:155-156
    auto & v = _variables[k];
    const torch::Tensor u0 = spectral(rhs[k].ubar0, "reciprocal_buffer");
    std::vector<torch::Tensor> keep;
    std::vector<const double *> ptr;
    for (const auto & t : rhs[k].N)
    {
      keep.push_back(spectral(t, "nonlinear_reciprocal"));
      ptr.push_back(static_cast<const double *>(keep.back().data_ptr()));
    }
    torch::Tensor L;
    if (v._linear_reciprocal)
      L = v._linear_reciprocal->expand(u0.sizes()).contiguous(); // (a broadcast k-axis product is materialised here)
    torch::Tensor ubar = torch::empty_like(u0);
    _hip->check(mrl_kspace_abm(_hip->ctx(), static_cast<double *>(ubar.data_ptr()), static_cast<const double *>(u0.data_ptr()),
                               ptr.data(), rhs[k].coef.data(), (int)ptr.size(), v._linear_reciprocal ? L.data_ptr<double>() : nullptr,
                               _sub_dt, n_spec),
                name());
    inverse(v, ubar);
  }
}

void
HipAdamsBashforthMoulton::substepGeneric()
{
  // re-evaluate the solve compute                                                          AdamsBashforthMoulton.C:63-64
  _compute->computeBuffer();
  forwardBuffers();

  // (zero-padded tables of the reference, including its first AB5 entry 190/720, AdamsBashforthMoulton.C:67-73, 108-114)
  static const double beta[max_order][max_order] = {{1.0, 0.0, 0.0, 0.0, 0.0},
                                                    {3.0 / 2.0, -1.0 / 2.0, 0.0, 0.0, 0.0},
                                                    {23.0 / 12.0, -16.0 / 12.0, 5.0 / 12.0, 0.0, 0.0},
                                                    {55.0 / 24.0, -59.0 / 24.0, 37.0 / 24.0, -9.0 / 24.0, 0.0},
                                                    {190.0 / 720.0, -2774.0 / 720.0, 2616.0 / 720.0, -1274.0 / 720.0, 251.0 / 720.0}};
  static const double alpha[max_order][max_order] = {{1.0, 0.0, 0.0, 0.0, 0.0},
                                                     {0.5, 0.5, 0.0, 0.0, 0.0},
                                                     {5.0 / 12.0, 8.0 / 12.0, -1.0 / 12.0, 0.0, 0.0},
                                                     {9.0 / 24.0, 19.0 / 24.0, -5.0 / 24.0, 1.0 / 24.0, 0.0},
                                                     {251.0 / 720.0, 646.0 / 720.0, -264.0 / 720.0, 106.0 / 720.0, -19.0 / 720.0}};
  const bool dt_changed = (_dt != _dt_old);
  const std::size_t nv = _variables.size();

  // Adams-Bashforth predictor on all variables                                             :80-102
  std::vector<Terms> rhs(nv);
  for (std::size_t k = 0; k < nv; ++k)
  {
    auto & v = _variables[k];
    const auto & hist = v._old_nonlinear_reciprocal;
    const std::size_t order = std::min(_substep < _predictor_order && dt_changed ? 0 : hist.size(), _predictor_order);
    if (_verbose)
      _console << name() << ": substep " << _substep << " order " << order << '\n';
    rhs[k].ubar0 = v._reciprocal_buffer;
    rhs[k].N = {v._nonlinear_reciprocal};
    rhs[k].coef = {_sub_dt * beta[order][0]};
    for (std::size_t i = 0; i < order; ++i)
    {
      rhs[k].N.push_back(hist[i]);
      rhs[k].coef.push_back(_sub_dt * beta[order][i + 1]);
    }
  }
  solve(rhs);

  if (!_corrector_steps)
    return;

  // Adams-Moulton corrector                                                                :117-177
  _sub_time += _sub_dt;
  std::vector<torch::Tensor> ubar_n, N_n; // handle copies keep the step-n tensors alive
  for (auto & v : _variables)
  {
    ubar_n.push_back(v._reciprocal_buffer);
    N_n.push_back(v._nonlinear_reciprocal);
  }
  for (std::size_t j = 0; j < _corrector_steps; ++j)
  {
    _compute->computeBuffer();
    forwardBuffers();
    for (std::size_t k = 0; k < nv; ++k)
    {
      auto & v = _variables[k];
      const auto & hist = v._old_nonlinear_reciprocal;
      const std::size_t order = std::min(_substep < _corrector_order && dt_changed ? 1 : hist.size() + 1, _corrector_order);
      rhs[k] = Terms{};
      rhs[k].ubar0 = ubar_n[k];
      rhs[k].none = order == 0;
      if (order == 0)
        continue;
      rhs[k].N = {v._nonlinear_reciprocal, N_n[k]};
      rhs[k].coef = {_sub_dt * alpha[order][0], _sub_dt * alpha[order][1]};
      for (std::size_t i = 0; i + 1 < order; ++i)
      {
        rhs[k].N.push_back(hist[i]);
        rhs[k].coef.push_back(_sub_dt * alpha[order][i + 2]);
      }
    }
    solve(rhs);
  }
  _sub_time -= _sub_dt;
}

// ---- HipAdamsBashforthMoultonCoupled ---------------------------------------------------------------------------------------

registerMooseObject("MarlinApp", HipAdamsBashforthMoultonCoupled);

InputParameters
HipAdamsBashforthMoultonCoupled::validParams()
{
  InputParameters params = HipAdamsBashforthMoulton::validParams();
  params.addClassDescription("Coupled Adams-Bashforth-Moulton solver with a dense linear operator on libmarlin_hip (MI355X).");
  // Off-diagonal linear operator specification
  params.addParam<std::vector<unsigned int>>("linear_offdiag_rows", {}, "Row indices for L_ij.");
  params.addParam<std::vector<unsigned int>>("linear_offdiag_cols", {}, "Column indices for L_ij.");
  params.addParam<std::vector<TensorInputBufferName>>("linear_offdiag", {}, "Off-diagonal linear operator buffers.");
  params.addParam<bool>("assume_symmetric", false, "Mirror off-diagonal entries (i,j) into (j,i) if not explicitly provided.");
  params.addParam<bool>("reference_quirks", true,
                        "Reproduce AdamsBashforthMoultonCoupled.C:160-183 as written (operator transposed, Im of the right-hand side "
                        "dropped): what the reference's gold files pin.  false = the system as the input file states it.");
  return params;
}

HipAdamsBashforthMoultonCoupled::HipAdamsBashforthMoultonCoupled(const InputParameters & parameters)
  : HipAdamsBashforthMoulton(parameters),
    _flags(getParam<bool>("reference_quirks") ? 0 : (MRL_COUPLED_L_AS_WRITTEN | MRL_COUPLED_COMPLEX_RHS))
{
  if (isParamValid("expression"))
    paramError("expression", "the coupled solver evaluates `root_compute`; the fused Cahn-Hilliard form is HipAdamsBashforthMoulton's");
  const auto & rows = getParam<std::vector<unsigned int>>("linear_offdiag_rows");
  const auto & cols = getParam<std::vector<unsigned int>>("linear_offdiag_cols");
  const auto & names = getParam<std::vector<TensorInputBufferName>>("linear_offdiag");
  if (rows.size() != names.size() || cols.size() != names.size())
    paramError("linear_offdiag", "'linear_offdiag_rows', 'linear_offdiag_cols', and 'linear_offdiag' must all have the same length.");
  const std::size_t N = _variables.size();
  if (N > 32)
    paramError("buffer", "at most 32 coupled variables");
  _L.assign(N * N, nullptr);
  for (std::size_t i = 0; i < N; ++i)
    _L[i * N + i] = _variables[i]._linear_reciprocal;
  for (std::size_t e = 0; e < names.size(); ++e)
  {
    if (rows[e] >= N)
      paramError("linear_offdiag_rows", "Off-diagonal indices out of range.");
    if (cols[e] >= N)
      paramError("linear_offdiag_cols", "Off-diagonal indices out of range.");
    _L[rows[e] * N + cols[e]] = &getInputBufferByName(names[e]);
  }
  if (getParam<bool>("assume_symmetric")) // AdamsBashforthMoultonCoupled.C:153-156
    for (std::size_t e = 0; e < names.size(); ++e)
      if (rows[e] != cols[e] && !_L[cols[e] * N + rows[e]])
        _L[cols[e] * N + rows[e]] = _L[rows[e] * N + cols[e]];
}

void
HipAdamsBashforthMoultonCoupled::solve(const std::vector<Terms> & rhs)
{
  const std::size_t nv = _variables.size();
  const int64_t n_spec = _hip->reciprocalCount();
  std::vector<torch::Tensor> keep, ubar;
  std::vector<const double *> u0, flatN, L;
  std::vector<double *> out;
  std::vector<double> flatc;
  std::vector<int> nterms;
  for (std::size_t k = 0; k < nv; ++k)
  {
    keep.push_back(spectral(rhs[k].ubar0, "reciprocal_buffer"));
    u0.push_back(static_cast<const double *>(keep.back().data_ptr()));
    ubar.push_back(torch::empty_like(keep.back()));
    out.push_back(static_cast<double *>(ubar.back().data_ptr()));
    nterms.push_back((int)rhs[k].N.size()); // (a variable whose corrector order is 0 keeps rhs = ubar_n and is still solved, :225-229)
    for (const auto & t : rhs[k].N)
    {
      keep.push_back(spectral(t, "nonlinear_reciprocal"));
      flatN.push_back(static_cast<const double *>(keep.back().data_ptr()));
    }
    flatc.insert(flatc.end(), rhs[k].coef.begin(), rhs[k].coef.end());
  }
  for (const auto * l : _L)
  {
    if (!l)
    {
      L.push_back(nullptr);
      continue;
    }
    keep.push_back(l->expand(keep[0].sizes()).contiguous());
    if (keep.back().is_complex())
      paramError("linear_reciprocal", "linear operator buffers are real");
    L.push_back(keep.back().data_ptr<double>());
  }
  _hip->check(mrl_kspace_coupled(_hip->ctx(), (int)nv, out.data(), u0.data(), flatN.data(), flatc.data(), nterms.data(), L.data(),
                                 _sub_dt, _flags, n_spec),
              name());
  for (std::size_t k = 0; k < nv; ++k)
    inverse(_variables[k], ubar[k]);
}
