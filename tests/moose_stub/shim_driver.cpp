// TEST INFRASTRUCTURE: runs the reference's regression inputs through the marlin_plugin/ classes (compiled unchanged against
// moose_stub.h) the way MOOSE's Transient executioner + TensorProblem would: objects are created by type name from parameter blocks,
// TensorProblem::execute(TIMESTEP_BEGIN) calls the solver, advanceState runs between the time steps.
//
//   shim-driver case=cahnhilliard nx=20 ny=20 xmax=3 ymax=3 ic=c0.bin substeps=10 num_steps=10 dt=1e-3 [fuse_substeps=false]
//               [predictor_order=2] [dt_sequence=a,b,c] out=dir          (test/tests/cahnhilliard/cahnhilliard.i, [TensorSolver] type =
//               HipAdamsBashforthMoulton with expression / mobility / kappa_factor): writes c.<step>.bin, mu.<step>.bin, Nhat.<step>.bin
//   shim-driver case=brusselator nx=150 ny=150 xmax=2pi ymax=2pi ss=10 cs=0 order=2 num_steps=25 dt=0.5 out=dir
//               (test/tests/solvers/diagonal.i: two variables, HipForwardFFT / HipParsedCompute / HipReciprocalLaplacianFactor in the
//               compute group, HipAdamsBashforthMoulton without `expression`): writes brusselator.csv
//   shim-driver case=coupled nx=150 ny=150 xmax=2pi ymax=2pi ss=10 cs=0 order=2 num_steps=25 dt=10 out=dir
//               (test/tests/solvers/coupled.i: HipAdamsBashforthMoultonCoupled, dense 2 x 2 operator): writes coupled.csv
//   shim-driver case=etdrk4 nx=64 xmax=2pi D=0.05 k=1.0 ss=1 dt=10 num_steps=10 out=dir
//               (test/tests/solvers/etdrk4_diffusion.i: HipETDRK4Solver, 1-D): writes etdrk4.csv
//   shim-driver case=secant nx=40 ny=40 xmax=12pi ymax=<..> ic=psi0.bin substeps=3 num_steps=10 dt=1 out=dir
//               (test/tests/tensor_compute/rotating_grain_secant.i: HipSecantSolver, iteration-adaptive dt): writes psi.<step>.bin
//   shim-driver case=broyden nx=24 ny=24 xmax=2pi ymax=2pi num_steps=3 dt=0.05 out=dir
//               (two coupled Brusselator variables through HipBroydenSolver, one substep per step): writes u / v.<step>.bin
//   shim-driver case=explicit nx=50 ny=50 xmax=3 ymax=3 ic=c0.bin method=SHARP|HOULI substeps=50 num_steps=20 dt=0.5 out=dir
//               (test/tests/cahnhilliard/cahnhilliard_explicit_smooth.i: HipForwardEulerSolver + HipDeAliasingTensor): writes c / mu.<step>.bin
//   shim-driver case=kks nx=20 ny=20 xmin=-50 xmax=50 ymin=-50 ymax=50 c=c0.bin eta=eta0.bin psi=psi0.bin num_steps=10 dt=0.1 out=dir
//               (test/tests/kks/KKS_no_flux_bc.i: HipReciprocalMatDiffusion, HipReciprocalAllenCahn, two-variable HipAdamsBashforthMoulton order 3)
//   shim-driver case=mechanics nx=16 ny=16 nz=16 substeps=10 num_steps=3 dt=0.01 l_tol=1e-2 nl_rel_tol=2e-2 nl_abs_tol=2e-2 out=dir
//               (test/tests/mechanics/mech3d.i with [mech] type = HipFFTMechanics, [displacements] type = HipComputeDisplacements,
//               [vonmises] type = HipComputeVonMisesStress): writes F / stress / disp / sV.<frame>.bin
//   shim-driver case=gradient|gradient_square nx=40 ny=40 nz=40 xmax=2pi ymax=4pi zmax=6pi out=dir
//               (test/tests/gradient/gradient.i, gradient_square.i with HipFFTGradient / HipFFTGradientSquare / HipParsedCompute)
//   shim-driver nranks=P case=cahnhilliard ...    (parallel_mode = FFT_SLAB on P ranks sharing device 0: the launcher starts P copies of
//               itself with rank=r job=<name> BEFORE anything touches the GPU and waits for them; every rank reads the global initial
//               condition, keeps its y slab and writes <name>.<frame>.rank<r>.bin -- cahnhilliard.rank0001.h5 of the reference's
//               2-rank regression is rank 1's file)
// Raw little-endian f64 files, dense row-major, as marlin-hip-run writes them.
#include "moose_stub.h"

#include <sys/wait.h>

#include <cmath>
#include <fstream>

static std::map<std::string, std::string> g_args;
static std::string
arg(const std::string & k, const std::string & dflt = "")
{
  auto it = g_args.find(k);
  return it == g_args.end() ? dflt : it->second;
}
static double
argd(const std::string & k, double dflt)
{
  if (!g_args.count(k))
    return dflt;
  const std::string & s = g_args[k];
  if (s.size() > 2 && s.substr(s.size() - 2) == "pi")
    return std::atof(s.substr(0, s.size() - 2).c_str()) * M_PI;
  return std::atof(s.c_str());
}
static long
argi(const std::string & k, long dflt)
{
  return g_args.count(k) ? std::atol(g_args[k].c_str()) : dflt;
}

static void
dump(const std::string & dir, const std::string & name, int frame, const torch::Tensor & t)
{
  torch::Tensor h = t.is_complex() ? torch::view_as_real(t.resolve_conj()) : t;
  h = h.contiguous().cpu();
  const auto & w = moose_stub::world();
  const std::string path =
      dir + "/" + name + "." + std::to_string(frame) + (w.size > 1 ? ".rank" + std::to_string(w.rank) : std::string()) + ".bin";
  std::ofstream f(path, std::ios::binary);
  if (!f)
    mooseError("cannot write ", path);
  f.write(reinterpret_cast<const char *>(h.data_ptr<double>()), sizeof(double) * h.numel());
}

static torch::Tensor
read_bin(const std::string & path, std::vector<int64_t> shape)
{
  torch::Tensor h = torch::empty(shape, torch::kFloat64);
  std::ifstream f(path, std::ios::binary);
  if (!f || !f.read(reinterpret_cast<char *>(h.data_ptr<double>()), sizeof(double) * h.numel()))
    mooseError("cannot read ", h.numel(), " doubles from ", path);
  return h.to(moose_stub::device());
}

/// this rank's real-space block of an initial condition given on the global grid (FFT_SLAB: the y range of getLocalBounds)
static torch::Tensor
read_ic(const std::string & path, const DomainAction & domain)
{
  const auto & n = domain.getGridSize();
  torch::Tensor g = read_bin(path, std::vector<int64_t>(n.begin(), n.begin() + domain.getDim()));
  if (!domain.isParallelFFT())
    return g;
  std::array<int64_t, 3> b, e;
  domain.getLocalBounds(moose_stub::world().rank, b, e);
  g = g.slice(1, b[1], e[1]);
  if (domain.getDim() == 3)   // (FFT_PENCIL also splits z; FFT_SLAB: the whole axis)
    g = g.slice(2, b[2], e[2]);
  return g.contiguous();
}

/// an [object] block of an input file: type + parameters as text
template <typename T>
static std::shared_ptr<T>
create(TensorProblem & problem, const std::string & type, const std::string & name,
       const std::vector<std::pair<std::string, std::string>> & block)
{
  auto it = MooseStubFactory::registry().find(type);
  if (it == MooseStubFactory::registry().end())
    mooseError("A '", type, "' is not a registered object");
  InputParameters params = it->second.valid_params();
  for (const auto & kv : block)
    params.setFromString(kv.first, kv.second);
  params.set<std::string>("_object_name") = name;
  params.set<TensorProblem *>("_tensor_problem") = &problem;
  params.checkRequired(name);
  auto obj = std::dynamic_pointer_cast<T>(it->second.build(params));
  if (!obj)
    mooseError(name, ": a '", type, "' is not of the requested base class");
  return obj;
}

/// the pre-TensorSolver syntax ([TensorTimeIntegrators], TensorProblem.C: per substep the root compute, then every integrator)
class StubTimeIntegratorSolver : public TensorSolver
{
public:
  static InputParameters validParams() { return TensorSolver::validParams(); }
  StubTimeIntegratorSolver(const InputParameters & p) : TensorSolver(p) {}
  void add(std::shared_ptr<TensorOperatorBase> ti) { _integrators.push_back(std::move(ti)); }

protected:
  virtual void substep() override
  {
    _compute->computeBuffer();
    forwardBuffers();
    for (auto & ti : _integrators)
      ti->computeBuffer();
  }
  std::vector<std::shared_ptr<TensorOperatorBase>> _integrators;
};
registerMooseObject("MarlinApp", StubTimeIntegratorSolver);

/// test/src/tensor_computes/MacroscopicShearTensor.C:31-43
class StubMacroscopicShearTensor : public TensorOperator<>
{
public:
  static InputParameters validParams()
  {
    InputParameters params = TensorOperator<>::validParams();
    params.addParam<TensorInputBufferName>("F", "F", "Deformation gradient tensor.");
    return params;
  }
  StubMacroscopicShearTensor(const InputParameters & p) : TensorOperator<>(p), _tF(getInputBuffer("F")) {}
  virtual void computeBuffer() override
  {
    std::vector<int64_t> grid_dims;
    for (unsigned int d = 0; d < _dim; ++d)
      grid_dims.push_back(d);
    const auto avg = _tF.sum(grid_dims) / double(_domain.getNumberOfCells());
    auto applied = torch::eye(_dim, MooseTensor::floatTensorOptions());
    applied.index_put_({0, 1}, applied.index({0, 1}) + _time);
    _u = applied - avg;
  }

private:
  const torch::Tensor & _tF;
};
registerMooseObject("MarlinApp", StubMacroscopicShearTensor);

/// Transient executioner: TransientBase::incrementStepOrReject (t_step += 1, advanceState), takeStep (dt_old = dt), then
/// TensorProblem::execute(EXEC_TIMESTEP_BEGIN) (TensorProblem.C:176-186: _sub_time = timeOld(), solver->computeBuffer())
template <typename F>
static void
transient(TensorProblem & problem, TensorSolver & solver, const std::vector<double> & dts, F && on_timestep_end)
{
  for (std::size_t s = 0; s < dts.size(); ++s)
  {
    problem.timeOld() = problem.time();
    problem.timeStep() += 1;
    const double dt_prev = problem.dt();
    problem.dt() = dts[s];
    problem.dtOld() = problem.timeStep() > 1 ? dt_prev : dts[s];
    problem.time() = problem.timeOld() + dts[s];
    problem.advanceState();
    problem.subTime() = problem.timeOld();
    solver.computeBuffer();
    on_timestep_end(problem.timeStep());
  }
}

static std::vector<double>
time_steps()
{
  std::vector<double> dts;
  if (g_args.count("dt_sequence"))
  {
    std::istringstream in(g_args["dt_sequence"]);
    std::string tok;
    while (std::getline(in, tok, ','))
      dts.push_back(std::atof(tok.c_str()));
  }
  else
    dts.assign((std::size_t)argi("num_steps", 1), argd("dt", 1.0));
  return dts;
}

static DomainAction
make_domain(unsigned int dim)
{
  return DomainAction(dim, {{argi("nx", 1), argi("ny", 1), argi("nz", 1)}}, {{argd("xmin", 0), argd("ymin", 0), argd("zmin", 0)}},
                      {{argd("xmax", 1), argd("ymax", 1), argd("zmax", 1)}});
}

static int
run_cahnhilliard(const std::string & out)
{
  const unsigned int dim = g_args.count("nz") ? 3 : 2;
  DomainAction domain = make_domain(dim);
  TensorProblem problem(domain);
  problem.getBuffer("c") = read_ic(arg("ic"), domain);
  // the buffers of cahnhilliard.i's [TensorSolver] block; the compute group's work is the solver's (expression = the [mu] block's)
  auto solver = create<TensorSolver>(problem, "HipAdamsBashforthMoulton", "solver",
                                     {{"buffer", "c"},
                                      {"reciprocal_buffer", "cbar"},
                                      {"linear_reciprocal", "kappabarbar"},
                                      {"nonlinear_reciprocal", "Mbarmubar"},
                                      {"substeps", arg("substeps", "10")},
                                      {"predictor_order", arg("predictor_order", "2")},
                                      {"expression", arg("expression", "0.1*c^2*(c-1)^2")},
                                      {"mobility", arg("mobility", "0.2")},
                                      {"kappa_factor", arg("kappa_factor", "-0.001")},
                                      {"chemical_potential", "mu"},
                                      {"fuse_substeps", arg("fuse_substeps", "true")},
                                      {"verbose", "true"}});
  solver->updateDependencies();
  transient(problem, *solver, time_steps(), [&](int step) {
    dump(out, "c", step, problem.getBuffer("c"));
    dump(out, "mu", step, problem.getBuffer("mu"));
    dump(out, "Nhat", step, problem.getBuffer("Mbarmubar")); // (read through the published view: dense values)
    if (moose_stub::world().rank == 0)
      std::cout << "step " << step << " time " << problem.time() << " sub_time " << problem.subTime() << " sub_dt " << problem.subDt()
                << "\n";
  });
  return 0;
}

// cahnhilliard.i in the legacy [TensorTimeIntegrators] form: explicit compute group + FFTSemiImplicit (history_size 1)
static int
run_semi_implicit(const std::string & out)
{
  DomainAction domain = make_domain(2);
  TensorProblem problem(domain);
  problem.getBuffer("c") = read_bin(arg("ic"), std::vector<int64_t>(domain.getShape().begin(), domain.getShape().end()));
  create<TensorOperatorBase>(problem, "HipReciprocalLaplacianFactor", "Mbar", {{"buffer", "Mbar"}, {"factor", arg("mobility", "0.2")}})->computeBuffer();
  create<TensorOperatorBase>(problem, "HipReciprocalLaplacianSquareFactor", "kappabarbar", {{"buffer", "kappabarbar"}, {"factor", arg("kappa_factor", "-0.001")}})
      ->computeBuffer();
  InputParameters gp;
  gp.set<std::string>("_object_name") = "root";
  gp.set<TensorProblem *>("_tensor_problem") = &problem;
  auto root = std::make_shared<ComputeGroup>(gp);
  root->add(create<TensorOperatorBase>(problem, "HipParsedCompute", "mu",
                                       {{"buffer", "mu"}, {"expression", "0.1*c^2*(c-1)^2"}, {"inputs", "c"}, {"derivatives", "c"}}));
  root->add(create<TensorOperatorBase>(problem, "HipForwardFFT", "mubar", {{"buffer", "mubar"}, {"input", "mu"}}));
  root->add(create<TensorOperatorBase>(problem, "HipParsedCompute", "Mbarmubar",
                                       {{"buffer", "Mbarmubar"}, {"expression", "Mbar*mubar"}, {"inputs", "Mbar mubar"}}));
  root->add(create<TensorOperatorBase>(problem, "HipForwardFFT", "cbar", {{"buffer", "cbar"}, {"input", "c"}}));
  problem.computes().push_back(root);
  auto solver = create<StubTimeIntegratorSolver>(problem, "StubTimeIntegratorSolver", "solver",
                                                 {{"root_compute", "root"}, {"substeps", arg("substeps", "10")}});
  solver->add(create<TensorOperatorBase>(problem, "HipFFTSemiImplicit", "c",
                                         {{"buffer", "c"}, {"reciprocal_buffer", "cbar"}, {"linear_reciprocal", "kappabarbar"},
                                          {"nonlinear_reciprocal", "Mbarmubar"}, {"history_size", "1"}}));
  solver->updateDependencies();
  transient(problem, *solver, time_steps(), [&](int step) {
    dump(out, "c", step, problem.getBuffer("c"));
    dump(out, "mu", step, problem.getBuffer("mu"));
  });
  return 0;
}

// test/tests/tensor_compute/coupled_pf_mech.i: Cahn-Hilliard + homogeneous quasistatic elasticity with the eigenstrain e0 * c, through
// the legacy FFTSemiImplicit integrator
static int
run_coupled_pf_mech(const std::string & out)
{
  DomainAction domain = make_domain(3);
  TensorProblem problem(domain);
  const std::vector<int64_t> shape(domain.getShape().begin(), domain.getShape().end());
  problem.getBuffer("c") = read_bin(arg("ic"), shape);
  for (const char * d : {"disp_x", "disp_y", "disp_z"})
    problem.getBuffer(d) = torch::zeros(shape, MooseTensor::floatTensorOptions()); // RandomTensor min = max = 0
  const std::string mu = arg("mu", "50.0"), lambda = arg("lambda", "100.0"), e0 = arg("e0", "0.02");
  create<TensorOperatorBase>(problem, "HipReciprocalLaplacianFactor", "Mbar", {{"buffer", "Mbar"}, {"factor", arg("mobility", "0.2")}})->computeBuffer();
  create<TensorOperatorBase>(problem, "HipReciprocalLaplacianSquareFactor", "kappabarbar", {{"buffer", "kappabarbar"}, {"factor", arg("kappa", "-0.001")}})
      ->computeBuffer();
  InputParameters gp;
  gp.set<std::string>("_object_name") = "Solve";
  gp.set<TensorProblem *>("_tensor_problem") = &problem;
  auto root = std::make_shared<ComputeGroup>(gp);
  root->add(create<TensorOperatorBase>(problem, "HipParsedCompute", "mu",
                                       {{"buffer", "mu"}, {"expression", "0.1*c^2*(c-1)^2"}, {"inputs", "c"}, {"derivatives", "c"}}));
  root->add(create<TensorOperatorBase>(problem, "HipForwardFFT", "mubar", {{"buffer", "mubar"}, {"input", "mu"}}));
  root->add(create<TensorOperatorBase>(problem, "HipForwardFFT", "cbar", {{"buffer", "cbar"}, {"input", "c"}}));
  root->add(create<TensorOperatorBase>(problem, "HipFFTQuasistaticElasticity", "qsmech",
                                       {{"displacements", "disp_x disp_y disp_z"}, {"cbar", "cbar"}, {"mu", mu}, {"lambda", lambda}, {"e0", e0}}));
  root->add(create<TensorOperatorBase>(problem, "HipFFTElasticChemicalPotential", "mumechbar",
                                       {{"buffer", "mumechbar"}, {"displacements", "disp_x disp_y disp_z"}, {"cbar", "cbar"}, {"mu", mu},
                                        {"lambda", lambda}, {"e0", e0}}));
  root->add(create<TensorOperatorBase>(problem, "HipInverseFFT", "mumech", {{"buffer", "mumech"}, {"input", "mumechbar"}}));
  root->add(create<TensorOperatorBase>(problem, "HipParsedCompute", "Mbarmubar",
                                       {{"buffer", "Mbarmubar"}, {"expression", "Mbar*(mubar+mumechbar)"}, {"inputs", "Mbar mubar mumechbar"}}));
  problem.computes().push_back(root);
  auto solver = create<StubTimeIntegratorSolver>(problem, "StubTimeIntegratorSolver", "solver",
                                                 {{"root_compute", "Solve"}, {"substeps", arg("substeps", "10")}});
  solver->add(create<TensorOperatorBase>(problem, "HipFFTSemiImplicit", "c",
                                         {{"buffer", "c"}, {"reciprocal_buffer", "cbar"}, {"linear_reciprocal", "kappabarbar"},
                                          {"nonlinear_reciprocal", "Mbarmubar"}, {"history_size", "1"}}));
  solver->updateDependencies();
  transient(problem, *solver, time_steps(), [&](int step) {
    dump(out, "c", step, problem.getBuffer("c"));
    dump(out, "mumech", step, problem.getBuffer("mumech"));
    for (const char * d : {"disp_x", "disp_y", "disp_z"})
      dump(out, d, step, problem.getBuffer(d));
  });
  return 0;
}

static int
run_brusselator(const std::string & out)
{
  DomainAction domain = make_domain(2);
  TensorProblem problem(domain);
  const std::string cn = "A B", cv = arg("A", "1") + " " + arg("B", "3.5");
  // [Initialize]
  create<TensorOperatorBase>(problem, "HipParsedCompute", "u",
                             {{"buffer", "u"}, {"expression", "sin(x)*sin(y)"}, {"extra_symbols", "true"}, {"expand", "REAL"}})
      ->computeBuffer();
  problem.getBuffer("v") = torch::zeros(domain.getShape(), MooseTensor::floatTensorOptions());
  create<TensorOperatorBase>(problem, "HipReciprocalLaplacianFactor", "Du", {{"buffer", "Du"}, {"factor", arg("Du", "1e-2")}})->computeBuffer();
  create<TensorOperatorBase>(problem, "HipReciprocalLaplacianFactor", "Dv", {{"buffer", "Dv"}, {"factor", arg("Dv", "1e-3")}})->computeBuffer();
  // [Solve], in dependency order
  InputParameters gp;
  gp.set<std::string>("_object_name") = "root";
  gp.set<TensorProblem *>("_tensor_problem") = &problem;
  auto root = std::make_shared<ComputeGroup>(gp);
  auto fft = [&](const std::string & to, const std::string & from)
  { root->add(create<TensorOperatorBase>(problem, "HipForwardFFT", to, {{"buffer", to}, {"input", from}})); };
  auto parsed = [&](const std::string & to, const std::string & expression)
  {
    root->add(create<TensorOperatorBase>(problem, "HipParsedCompute", to,
                                         {{"buffer", to}, {"expression", expression}, {"inputs", "u v"}, {"constant_names", cn},
                                          {"constant_expressions", cv}}));
  };
  fft("u_bar", "u");
  fft("v_bar", "v");
  parsed("source_u", "A - (B+1)*u +u^2*v");
  fft("source_u_bar", "source_u");
  parsed("source_v", "B*u - u^2*v");
  fft("source_v_bar", "source_v");
  problem.computes().push_back(root);
  auto solver = create<TensorSolver>(problem, "HipAdamsBashforthMoulton", "solver",
                                     {{"root_compute", "root"},
                                      {"buffer", "u v"},
                                      {"reciprocal_buffer", "u_bar v_bar"},
                                      {"linear_reciprocal", "Du Dv"},
                                      {"nonlinear_reciprocal", "source_u_bar source_v_bar"},
                                      {"substeps", arg("ss", "10")},
                                      {"corrector_steps", arg("cs", "0")},
                                      {"predictor_order", arg("order", "2")},
                                      {"corrector_order", arg("order", "2")}});
  solver->updateDependencies();
  const double volume = (argd("xmax", 1) - argd("xmin", 0)) * (argd("ymax", 1) - argd("ymin", 0));
  std::ofstream csv(out + "/brusselator.csv");
  csv.precision(17);
  csv << "time,U,V,u_max,u_min,v_max,v_min\n0,0,0,0,0,0,0\n";
  // TensorIntegralPostprocessor.C:29-38 (average * volume), TensorExtremeValuePostprocessor
  // (on several ranks: this rank's part of the integral; the parts add up to the postprocessor's value)
  auto integral = [&](const torch::Tensor & t) { return t.sum().item<double>() / double(domain.getGlobalNumberOfCells()) * volume; };
  transient(problem, *solver, time_steps(), [&](int) {
    const auto & u = problem.getBuffer("u");
    const auto & v = problem.getBuffer("v");
    csv << problem.time() << ',' << integral(u) << ',' << integral(v) << ',' << u.max().item<double>() << ',' << u.min().item<double>()
        << ',' << v.max().item<double>() << ',' << v.min().item<double>() << "\n";
  });
  return 0;
}

// test/tests/solvers/coupled.i: AdamsBashforthMoultonCoupled with the dense operator [[D1, D2], [D2, D1]] and zero nonlinear terms
static int
run_coupled(const std::string & out)
{
  DomainAction domain = make_domain(2);
  TensorProblem problem(domain);
  auto ic = [&](const std::string & buffer, const std::string & expr)
  {
    create<TensorOperatorBase>(problem, "HipParsedCompute", buffer,
                               {{"buffer", buffer}, {"expression", expr}, {"extra_symbols", "true"}, {"expand", "REAL"}})
        ->computeBuffer();
  };
  ic("u", "sin(x)*sin(y)");
  ic("v", "cos(x)*cos(y)");
  problem.getBuffer("zero") = torch::zeros(domain.getReciprocalShape(), MooseTensor::complexFloatTensorOptions()); // ConstantReciprocalTensor
  create<TensorOperatorBase>(problem, "HipReciprocalLaplacianFactor", "D1", {{"buffer", "D1"}, {"factor", arg("D1", "1e-2")}})->computeBuffer();
  create<TensorOperatorBase>(problem, "HipReciprocalLaplacianFactor", "D2", {{"buffer", "D2"}, {"factor", arg("D2", "1e-3")}})->computeBuffer();
  InputParameters gp;
  gp.set<std::string>("_object_name") = "root";
  gp.set<TensorProblem *>("_tensor_problem") = &problem;
  auto root = std::make_shared<ComputeGroup>(gp);
  root->add(create<TensorOperatorBase>(problem, "HipForwardFFT", "u_bar", {{"buffer", "u_bar"}, {"input", "u"}}));
  root->add(create<TensorOperatorBase>(problem, "HipForwardFFT", "v_bar", {{"buffer", "v_bar"}, {"input", "v"}}));
  problem.computes().push_back(root);
  auto solver = create<TensorSolver>(problem, "HipAdamsBashforthMoultonCoupled", "solver",
                                     {{"root_compute", "root"},
                                      {"buffer", "u v"},
                                      {"reciprocal_buffer", "u_bar v_bar"},
                                      {"linear_reciprocal", "D1 D1"},
                                      {"linear_offdiag_cols", "0 1"},
                                      {"linear_offdiag_rows", "1 0"},
                                      {"linear_offdiag", "D2 D2"},
                                      {"nonlinear_reciprocal", "zero zero"},
                                      {"substeps", arg("ss", "10")},
                                      {"corrector_steps", arg("cs", "0")},
                                      {"predictor_order", arg("order", "2")},
                                      {"corrector_order", arg("order", "2")}});
  solver->updateDependencies();
  const double volume = (argd("xmax", 1) - argd("xmin", 0)) * (argd("ymax", 1) - argd("ymin", 0));
  std::ofstream csv(out + "/coupled.csv");
  csv.precision(17);
  csv << "time,U,V,u_max,u_min,v_max,v_min\n0,0,0,0,0,0,0\n";
  // (on several ranks: this rank's part of the integral; the parts add up to the postprocessor's value)
  auto integral = [&](const torch::Tensor & t) { return t.sum().item<double>() / double(domain.getGlobalNumberOfCells()) * volume; };
  transient(problem, *solver, time_steps(), [&](int) {
    const auto & u = problem.getBuffer("u");
    const auto & v = problem.getBuffer("v");
    csv << problem.time() << ',' << integral(u) << ',' << integral(v) << ',' << u.max().item<double>() << ',' << u.min().item<double>()
        << ',' << v.max().item<double>() << ',' << v.min().item<double>() << "\n";
  });
  return 0;
}

// test/tests/solvers/etdrk4_diffusion.i: 1-D periodic diffusion advanced by the ETDRK4 solver against the analytic solution
static int
run_etdrk4(const std::string & out)
{
  DomainAction domain = make_domain(1);
  TensorProblem problem(domain);
  const std::string D = arg("D", "0.05"), k = arg("k", "1.0");
  create<TensorOperatorBase>(problem, "HipParsedCompute", "u0",
                             {{"buffer", "u0"}, {"expression", "sin(kk*x)"}, {"constant_names", "kk"}, {"constant_expressions", k},
                              {"extra_symbols", "true"}, {"expand", "REAL"}})
      ->computeBuffer();
  create<TensorOperatorBase>(problem, "HipParsedCompute", "u", {{"buffer", "u"}, {"expression", "u0"}, {"inputs", "u0"}})->computeBuffer();
  create<TensorOperatorBase>(problem, "HipReciprocalLaplacianFactor", "L", {{"buffer", "L"}, {"factor", D}})->computeBuffer();
  problem.getBuffer("zero") = torch::zeros(domain.getReciprocalShape(), MooseTensor::complexFloatTensorOptions()); // ConstantReciprocalTensor
  InputParameters gp;
  gp.set<std::string>("_object_name") = "root";
  gp.set<TensorProblem *>("_tensor_problem") = &problem;
  auto root = std::make_shared<ComputeGroup>(gp);
  root->add(create<TensorOperatorBase>(problem, "HipForwardFFT", "u_bar", {{"buffer", "u_bar"}, {"input", "u"}}));
  root->add(create<TensorOperatorBase>(problem, "HipParsedCompute", "u_exact",
                                       {{"buffer", "u_exact"}, {"expression", "u0*exp(-DD*kk^2*t)"}, {"inputs", "u0"},
                                        {"constant_names", "DD kk"}, {"constant_expressions", D + " " + k}, {"extra_symbols", "true"},
                                        {"expand", "REAL"}}));
  root->add(create<TensorOperatorBase>(problem, "HipParsedCompute", "u_diff_sq",
                                       {{"buffer", "u_diff_sq"}, {"expression", "(u - u_exact)^2"}, {"inputs", "u u_exact"}}));
  problem.computes().push_back(root);
  auto solver = create<TensorSolver>(problem, "HipETDRK4Solver", "solver",
                                     {{"root_compute", "root"}, {"buffer", "u"}, {"reciprocal_buffer", "u_bar"}, {"linear_reciprocal", "L"},
                                      {"nonlinear_reciprocal", "zero"}, {"substeps", arg("ss", "1")}});
  solver->updateDependencies();
  const double length = argd("xmax", 1) - argd("xmin", 0);
  std::ofstream csv(out + "/etdrk4.csv");
  csv.precision(17);
  csv << "time,mse,rmse\n0,0,0\n";
  transient(problem, *solver, time_steps(), [&](int) {
    const double mse = problem.getBuffer("u_diff_sq").sum().item<double>() / double(domain.getNumberOfCells()) * length;
    csv << problem.time() << ',' << mse << ',' << std::sqrt(mse) << "\n";
  });
  return 0;
}

// test/tests/tensor_compute/rotating_grain_secant.i: SecantSolver + SwiftHohenbergLinear + TensorSolveIterationAdaptiveDT
// (src/timesteppers/TensorSolveIterationAdaptiveDT.C: dt grows / shrinks with the solver's iteration count of the previous step)
static int
run_secant(const std::string & out)
{
  DomainAction domain = make_domain(2);
  TensorProblem problem(domain);
  problem.getBuffer("psi") = read_bin(arg("ic"), std::vector<int64_t>(domain.getShape().begin(), domain.getShape().end()));
  create<TensorOperatorBase>(problem, "HipSwiftHohenbergLinear", "linear", {{"buffer", "linear"}, {"alpha", arg("alpha", "1")}, {"r", arg("r", "0.025")}})
      ->computeBuffer();
  InputParameters gp;
  gp.set<std::string>("_object_name") = "root";
  gp.set<TensorProblem *>("_tensor_problem") = &problem;
  auto root = std::make_shared<ComputeGroup>(gp);
  root->add(create<TensorOperatorBase>(problem, "HipParsedCompute", "psi3",
                                       {{"buffer", "psi3"}, {"expression", "0.20*psi^2-psi^3"}, {"inputs", "psi"}}));
  root->add(create<TensorOperatorBase>(problem, "HipForwardFFT", "psibar", {{"buffer", "psibar"}, {"input", "psi"}}));
  root->add(create<TensorOperatorBase>(problem, "HipForwardFFT", "psi3bar", {{"buffer", "psi3bar"}, {"input", "psi3"}}));
  problem.computes().push_back(root);
  auto solver = create<TensorSolver>(problem, "HipSecantSolver", "solver",
                                     {{"root_compute", "root"}, {"buffer", "psi"}, {"reciprocal_buffer", "psibar"},
                                      {"linear_reciprocal", "linear"}, {"nonlinear_reciprocal", "psi3bar"},
                                      {"substeps", arg("substeps", "3")}, {"max_iterations", arg("max_iterations", "30")}});
  solver->updateDependencies();
  auto * iterative = dynamic_cast<IterativeTensorSolverInterface *>(solver.get());
  if (!iterative)
    mooseError("HipSecantSolver does not implement IterativeTensorSolverInterface");
  const double dt0 = argd("dt", 1.0), growth = argd("growth_factor", 1.4), cutback = argd("cutback_factor", 0.9), dtmax = argd("dtmax", 500.0);
  const unsigned int ts_min = (unsigned int)argi("ts_min_iterations", 100), ts_max = (unsigned int)argi("ts_max_iterations", 400);
  dump(out, "psi", 0, problem.getBuffer("psi"));
  double dt_old = 0.0;
  for (int step = 1; step <= (int)argi("num_steps", 10); ++step)
  {
    double dt = dt0; // computeInitialDT
    if (step > 1)
    {
      dt = dt_old;
      if (iterative->getIterations() < ts_min)
        dt *= growth;
      else if (iterative->getIterations() > ts_max)
        dt *= cutback;
    }
    dt = std::min(dt, dtmax);
    dt_old = dt;
    transient(problem, *solver, {dt}, [&](int) {
      dump(out, "psi", step, problem.getBuffer("psi"));
      std::cout << "step " << step << ": dt=" << dt << " iterations=" << iterative->getIterations() << " converged=" << iterative->isConverged() << "\n";
    });
  }
  return 0;
}

// two coupled reaction-diffusion variables (the Brusselator sources of diagonal.i) integrated implicitly by the Broyden solver: the
// reference ships no input for BroydenSolver; this is the problem of tests/test_broyden_gpu.py
static int
run_broyden(const std::string & out)
{
  DomainAction domain = make_domain(2);
  TensorProblem problem(domain);
  auto ic = [&](const std::string & buffer, const std::string & expr)
  {
    create<TensorOperatorBase>(problem, "HipParsedCompute", buffer,
                               {{"buffer", buffer}, {"expression", expr}, {"extra_symbols", "true"}, {"expand", "REAL"}})
        ->computeBuffer();
  };
  ic("u", "1.0 + 0.1*sin(x)*sin(y)");
  ic("v", "3.0 + 0.1*cos(x)*cos(2*y)");
  create<TensorOperatorBase>(problem, "HipReciprocalLaplacianFactor", "Du", {{"buffer", "Du"}, {"factor", "1e-2"}})->computeBuffer();
  create<TensorOperatorBase>(problem, "HipReciprocalLaplacianFactor", "Dv", {{"buffer", "Dv"}, {"factor", "1e-3"}})->computeBuffer();
  InputParameters gp;
  gp.set<std::string>("_object_name") = "root";
  gp.set<TensorProblem *>("_tensor_problem") = &problem;
  auto root = std::make_shared<ComputeGroup>(gp);
  auto fft = [&](const std::string & to, const std::string & from)
  { root->add(create<TensorOperatorBase>(problem, "HipForwardFFT", to, {{"buffer", to}, {"input", from}})); };
  auto parsed = [&](const std::string & to, const std::string & expression)
  {
    root->add(create<TensorOperatorBase>(problem, "HipParsedCompute", to,
                                         {{"buffer", to}, {"expression", expression}, {"inputs", "u v"}, {"constant_names", "A B"},
                                          {"constant_expressions", "1 3.5"}}));
  };
  fft("u_bar", "u");
  fft("v_bar", "v");
  parsed("su", "A - (B+1)*u +u^2*v");
  fft("su_bar", "su");
  parsed("sv", "B*u - u^2*v");
  fft("sv_bar", "sv");
  problem.computes().push_back(root);
  auto solver = create<TensorSolver>(problem, "HipBroydenSolver", "solver",
                                     {{"root_compute", "root"}, {"buffer", "u v"}, {"reciprocal_buffer", "u_bar v_bar"},
                                      {"linear_reciprocal", "Du Dv"}, {"nonlinear_reciprocal", "su_bar sv_bar"}, {"substeps", "1"},
                                      {"max_iterations", arg("max_iterations", "30")},
                                      {"relative_tolerance", arg("relative_tolerance", "1e-6")},
                                      {"absolute_tolerance", arg("absolute_tolerance", "1e-10")}});
  solver->updateDependencies();
  auto * iterative = dynamic_cast<IterativeTensorSolverInterface *>(solver.get());
  transient(problem, *solver, time_steps(), [&](int step) {
    dump(out, "u", step, problem.getBuffer("u"));
    dump(out, "v", step, problem.getBuffer("v"));
    std::cout << "step " << step << ": iterations=" << iterative->getIterations() << " converged=" << iterative->isConverged() << "\n";
  });
  return 0;
}

// test/tests/cahnhilliard/cahnhilliard_explicit_smooth.i: explicit Euler Cahn-Hilliard with a de-aliasing filter on the rate
static int
run_explicit(const std::string & out)
{
  DomainAction domain = make_domain(2);
  TensorProblem problem(domain);
  problem.getBuffer("c") = read_bin(arg("ic"), std::vector<int64_t>(domain.getShape().begin(), domain.getShape().end()));
  problem.getBuffer("mu") = torch::zeros(domain.getShape(), MooseTensor::floatTensorOptions());                       // ConstantTensor
  problem.getBuffer("dc_dt_bar") = torch::zeros(domain.getReciprocalShape(), MooseTensor::complexFloatTensorOptions()); // ConstantReciprocalTensor
  create<TensorOperatorBase>(problem, "HipReciprocalLaplacianFactor", "Mbar", {{"buffer", "Mbar"}, {"factor", arg("mobility", "0.2")}})->computeBuffer();
  create<TensorOperatorBase>(problem, "HipReciprocalLaplacianSquareFactor", "Mkappabarbar", {{"buffer", "Mkappabarbar"}, {"factor", arg("Mkappa", "2e-5")}})
      ->computeBuffer();
  create<TensorOperatorBase>(problem, "HipDeAliasingTensor", "smooth", {{"buffer", "smooth"}, {"method", arg("method", "SHARP")}})->computeBuffer();
  InputParameters gp;
  gp.set<std::string>("_object_name") = "cahn_hilliard";
  gp.set<TensorProblem *>("_tensor_problem") = &problem;
  auto root = std::make_shared<ComputeGroup>(gp);
  root->add(create<TensorOperatorBase>(problem, "HipParsedCompute", "mu",
                                       {{"buffer", "mu"}, {"expression", "0.1*c^2*(c-1)^2"}, {"inputs", "c"}, {"derivatives", "c"}}));
  root->add(create<TensorOperatorBase>(problem, "HipForwardFFT", "mubar", {{"buffer", "mubar"}, {"input", "mu"}}));
  root->add(create<TensorOperatorBase>(problem, "HipForwardFFT", "cbar", {{"buffer", "cbar"}, {"input", "c"}}));
  root->add(create<TensorOperatorBase>(problem, "HipParsedCompute", "dc_dt_bar",
                                       {{"buffer", "dc_dt_bar"}, {"expression", "smooth * (Mbar*mubar - Mkappabarbar*cbar)"},
                                        {"inputs", "Mbar mubar Mkappabarbar cbar smooth"}}));
  problem.computes().push_back(root);
  auto solver = create<TensorSolver>(problem, "HipForwardEulerSolver", "solver",
                                     {{"root_compute", "cahn_hilliard"}, {"buffer", "c"}, {"reciprocal_buffer", "cbar"},
                                      {"time_derivative_reciprocal", "dc_dt_bar"}, {"substeps", arg("substeps", "50")}});
  solver->updateDependencies();
  transient(problem, *solver, time_steps(), [&](int step) {
    dump(out, "c", step, problem.getBuffer("c"));
    dump(out, "mu", step, problem.getBuffer("mu"));
  });
  return 0;
}

// test/tests/kks/KKS_no_flux_bc.i: two-variable Kim-Kim-Suzuki model with the smooth boundary method; AdamsBashforthMoulton order 3
static int
run_kks(const std::string & out)
{
  DomainAction domain = make_domain(2);
  TensorProblem problem(domain);
  const std::vector<int64_t> shape(domain.getShape().begin(), domain.getShape().end());
  for (const char * b : {"c", "eta", "psi"})
    problem.getBuffer(b) = read_bin(arg(b), shape);
  auto constant = [&](const char * b, double v) { problem.getBuffer(b) = torch::full(shape, v, MooseTensor::floatTensorOptions()); };
  constant("M", argd("M", 5.0));
  constant("L", argd("L", 5.0));
  constant("L_kappa", argd("L", 5.0) * argd("kappa_eta", 5.0));
  // ${F} after MOOSE's textual substitution of h_eta, rho_sq, w, c0_a, c0_b (KKS_no_flux_bc.i:24-25)
  const std::string h = "eta^3*(6*eta^2-15*eta+10)";
  const std::string F = h + "*(2*((c - (1-" + h + ")*(0.7 - 0.3))-0.3)^2) + (1-" + h + ")*(2*((c + (" + h + ")*(0.7 - 0.3))-0.7)^2 ) + 1*(eta^2)*(1-eta)^2";
  InputParameters gp;
  gp.set<std::string>("_object_name") = "root";
  gp.set<TensorProblem *>("_tensor_problem") = &problem;
  auto root = std::make_shared<ComputeGroup>(gp);
  auto add = [&](const std::string & type, const std::string & name, std::vector<std::pair<std::string, std::string>> block)
  {
    block.push_back({"buffer", name});
    root->add(create<TensorOperatorBase>(problem, type, name, block));
  };
  add("HipForwardFFT", "cbar", {{"input", "c"}});
  add("HipForwardFFT", "etabar", {{"input", "eta"}});
  add("HipParsedCompute", "mu", {{"expression", F}, {"inputs", "c eta"}, {"derivatives", "c"}});
  add("HipReciprocalMatDiffusion", "div_J", {{"chemical_potential", "mu"}, {"mobility", "M"}, {"psi", "psi"}});
  add("HipParsedCompute", "domega_chem_deta", {{"expression", F + " - mu*c"}, {"inputs", "mu c eta"}, {"derivatives", "eta"}});
  add("HipReciprocalAllenCahn", "AC_bulk", {{"dF_chem_deta", "domega_chem_deta"}, {"L", "L"}, {"psi", "psi"}});
  add("HipReciprocalMatDiffusion", "kappa_grad_eta", {{"chemical_potential", "eta"}, {"mobility", "L_kappa"}, {"psi", "psi"}});
  add("HipParsedCompute", "AC_bar", {{"expression", "kappa_grad_eta + AC_bulk"}, {"inputs", "AC_bulk kappa_grad_eta"}});
  problem.computes().push_back(root);
  // linear_reciprocal = '0 0': no linear operator (dividing by 1 - dt * 0 is the identity)
  auto solver = create<TensorSolver>(problem, "HipAdamsBashforthMoulton", "solver",
                                     {{"root_compute", "root"}, {"buffer", "c eta"}, {"reciprocal_buffer", "cbar etabar"},
                                      {"linear_reciprocal", "0 0"}, {"nonlinear_reciprocal", "div_J AC_bar"},
                                      {"substeps", arg("substeps", "1000")}, {"predictor_order", arg("predictor_order", "3")},
                                      {"corrector_order", arg("predictor_order", "3")}});
  solver->updateDependencies();
  const double volume = (argd("xmax", 1) - argd("xmin", 0)) * (argd("ymax", 1) - argd("ymin", 0));
  // (on several ranks: this rank's part of the integral; the parts add up to the postprocessor's value)
  auto integral = [&](const torch::Tensor & t) { return t.sum().item<double>() / double(domain.getGlobalNumberOfCells()) * volume; };
  std::ofstream csv(out + "/kks.csv");
  csv.precision(17);
  csv << "time,total_C,total_eta\n";
  auto row = [&](int frame)
  {
    csv << problem.time() << ',' << integral(problem.getBuffer("c")) << ',' << integral(problem.getBuffer("eta")) << "\n";
    for (const char * b : {"c", "eta", "mu"})
      if (problem.getBuffer(b).defined())
        dump(out, b, frame, problem.getBuffer(b));
  };
  row(0);
  transient(problem, *solver, time_steps(), [&](int step) { row(step); });
  return 0;
}

static int
run_mechanics(const std::string & out)
{
  const unsigned int dim = g_args.count("nz") ? 3 : 2;
  DomainAction domain = make_domain(dim);
  TensorProblem problem(domain);
  // [Initialize] of mech3d.i:14-41
  std::string phase = "(cos(x)/2+0.5)^1*(cos(y)/2+0.5)^1";
  if (dim == 3)
    phase += "*(cos(z)/2+0.5)^1";
  create<TensorOperatorBase>(problem, "HipParsedCompute", "phase",
                             {{"buffer", "phase"}, {"expression", phase}, {"extra_symbols", "true"}, {"expand", "REAL"}})
      ->computeBuffer();
  create<TensorOperatorBase>(problem, "HipParsedCompute", "K",
                             {{"buffer", "K"}, {"expression", "(1-phase)*Ka + phase*Kb"}, {"inputs", "phase"}, {"constant_names", "Ka Kb"},
                              {"constant_expressions", arg("Ka", "1") + " " + arg("Kb", "10")}})
      ->computeBuffer();
  create<TensorOperatorBase>(problem, "HipParsedCompute", "mu",
                             {{"buffer", "mu"}, {"expression", "(1-phase)*mua + phase*mub"}, {"inputs", "phase"},
                              {"constant_names", "mua mub"}, {"constant_expressions", arg("mua", "0.5") + " " + arg("mub", "5")}})
      ->computeBuffer();
  {
    // RankTwoIdentity.C:31-32: an expanded view of eye(dim), not a dense array
    std::vector<int64_t> shape(domain.getShape().begin(), domain.getShape().end());
    shape.push_back(dim);
    shape.push_back(dim);
    problem.getBuffer("F") = torch::eye(dim, MooseTensor::floatTensorOptions()).expand(shape);
  }
  // [Solve] root group of mech3d.i:53-71 with type = HipFFTMechanics (which is FFTMechanics + HyperElasticIsotropic)
  InputParameters gp;
  gp.set<std::string>("_object_name") = "root";
  gp.set<TensorProblem *>("_tensor_problem") = &problem;
  auto root = std::make_shared<ComputeGroup>(gp);
  root->add(create<TensorOperatorBase>(problem, "StubMacroscopicShearTensor", "applied_strain", {{"buffer", "applied_strain"}}));
  std::vector<std::pair<std::string, std::string>> mech = {{"buffer", "Fnew"},
                                                           {"F", "F"},
                                                           {"K", "K"},
                                                           {"mu", "mu"},
                                                           {"l_tol", arg("l_tol", "1e-2")},
                                                           {"nl_rel_tol", arg("nl_rel_tol", "1e-5")},
                                                           {"nl_abs_tol", arg("nl_abs_tol", "1e-8")},
                                                           {"stress", "stress"},
                                                           {"applied_macroscopic_strain", "applied_strain"},
                                                           {"verbose", "true"}};
  if (g_args.count("l_max_its"))
    mech.push_back({"l_max_its", arg("l_max_its")});
  root->add(create<TensorOperatorBase>(problem, "HipFFTMechanics", "mech", mech));
  problem.computes().push_back(root);
  auto solver = create<TensorSolver>(problem, "HipForwardEulerSolver", "solver",
                                     {{"root_compute", "root"}, {"forward_buffer", "F"}, {"forward_buffer_new", "Fnew"},
                                      {"substeps", arg("substeps", "1")}});
  solver->updateDependencies();
  // [Postprocess] of mech3d.i:74-85, evaluated before the outputs of a time step
  const bool serial = moose_stub::world().size == 1;   // (the nodal displacement field is a serial postprocess: the class says so itself)
  std::shared_ptr<TensorOperatorBase> displacements;
  if (serial)
    displacements = create<TensorOperatorBase>(problem, "HipComputeDisplacements", "displacements", {{"buffer", "disp"}, {"F", "F"}});
  auto vonmises = create<TensorOperatorBase>(problem, "HipComputeVonMisesStress", "vonmises", {{"buffer", "sV"}});
  transient(problem, *solver, time_steps(), [&](int step) {
    if (serial)
      displacements->computeBuffer();
    vonmises->computeBuffer();
    dump(out, "F", step - 1, problem.getBuffer("F"));
    dump(out, "stress", step - 1, problem.getBuffer("stress"));
    if (serial)
      dump(out, "disp", step - 1, problem.getBuffer("disp"));
    dump(out, "sV", step - 1, problem.getBuffer("sV"));
  });
  return 0;
}

// test/tests/gradient/gradient.i and gradient_square.i: spectral derivatives of sin(x)+sin(y)+sin(z) against the analytic ones,
// the postprocessor value = integral of the absolute difference (TensorIntegralPostprocessor.C:29-38)
static int
run_gradient(const std::string & out, bool square)
{
  DomainAction domain = make_domain(3);
  TensorProblem problem(domain);
  auto parsed = [&](const std::string & buffer, const std::string & expr, const std::string & inputs, bool extra)
  {
    std::vector<std::pair<std::string, std::string>> block = {{"buffer", buffer}, {"expression", expr}};
    if (!inputs.empty())
      block.push_back({"inputs", inputs});
    if (extra)
    {
      block.push_back({"extra_symbols", "true"});
      block.push_back({"expand", "REAL"});
    }
    create<TensorOperatorBase>(problem, "HipParsedCompute", buffer, block)->computeBuffer();
  };
  const double volume = (argd("xmax", 1) - argd("xmin", 0)) * (argd("ymax", 1) - argd("ymin", 0)) * (argd("zmax", 1) - argd("zmin", 0));
  // (on several ranks: this rank's part of the integral; the parts add up to the postprocessor's value)
  auto integral = [&](const torch::Tensor & t) { return t.sum().item<double>() / double(domain.getGlobalNumberOfCells()) * volume; };
  parsed("s", "sin(x)+sin(y)+sin(z)", "", true);
  const auto & wld = moose_stub::world();
  std::ofstream csv(out + (square ? "/gradient_square" : "/gradient") + (wld.size > 1 ? ".rank" + std::to_string(wld.rank) : std::string()) + ".csv");
  csv.precision(17);
  if (square)
  {
    parsed("c2", "cos(x)^2+cos(y)^2+cos(z)^2", "", true);
    create<TensorOperatorBase>(problem, "HipFFTGradientSquare", "grad_sq", {{"buffer", "grad_sq"}, {"input", "s"}})->computeBuffer();
    parsed("diff", "abs(grad_sq - c2)", "grad_sq c2", false);
  }
  else
  {
    parsed("cx", "cos(x)", "", true);
    parsed("cy", "cos(y)", "", true);
    parsed("cz", "cos(z)", "", true);
    const char * dir[] = {"X", "Y", "Z"};
    const char * buf[] = {"gradx_s", "grady_s", "gradz_s"};
    for (int d = 0; d < 3; ++d)
      create<TensorOperatorBase>(problem, "HipFFTGradient", buf[d], {{"buffer", buf[d]}, {"input", "s"}, {"direction", dir[d]}})->computeBuffer();
    parsed("diff", "abs(gradx_s - cx)+abs(grady_s - cy)+abs(gradz_s - cz)", "gradx_s grady_s gradz_s cx cy cz", false);
  }
  csv << "time,diff\n0,0\n1," << integral(problem.getBuffer("diff")) << "\n";
  return 0;
}

int
main(int argc, char ** argv)
{
  for (int i = 1; i < argc; ++i)
  {
    const std::string a = argv[i];
    const auto eq = a.find('=');
    if (eq == std::string::npos)
    {
      std::cerr << "arguments are key=value\n";
      return 2;
    }
    g_args[a.substr(0, eq)] = a.substr(eq + 1);
  }
  try
  {
    if (arg("case") == "types") // (no GPU needed: what registerMooseObject has registered)
    {
      for (const auto & kv : MooseStubFactory::registry())
        std::cout << kv.first << "\n";
      return 0;
    }
    const long nranks = argi("nranks", 1);
    if (nranks > 1 && !g_args.count("rank"))
    {
      // the "mpiexec" of the stub: P copies of this program, started before this process has made any HIP call (it never makes one)
      const std::string job = "shimjob_" + std::to_string(getpid());
      std::vector<pid_t> kids;
      for (long r = 0; r < nranks; ++r)
      {
        const pid_t pid = fork();
        if (pid < 0)
          mooseError("fork failed");
        if (pid == 0)
        {
          std::vector<std::string> av(argv, argv + argc);
          av.push_back("rank=" + std::to_string(r));
          av.push_back("job=" + job);
          std::vector<char *> cav;
          for (auto & a : av)
            cav.push_back(a.data());
          cav.push_back(nullptr);
          execv("/proc/self/exe", cav.data());
          _exit(127);
        }
        kids.push_back(pid);
      }
      int rc = 0;
      for (pid_t k : kids)
      {
        int st = 0;
        waitpid(k, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0)
          rc = 1;
      }
      shm_unlink(("/" + job).c_str());
      return rc;
    }
    if (nranks > 1)
    {
      moose_stub::joinWorld(arg("job"), (unsigned int)nranks, (unsigned int)argi("rank", 0));
      moose_stub::world().pencil = arg("parallel_mode", "FFT_SLAB") == "FFT_PENCIL";
    }
    if (!torch::cuda::is_available())
      mooseError("shim-driver needs a GPU (libTorch sees no HIP device)");
    moose_stub::device() = torch::Device(torch::kCUDA, (c10::DeviceIndex)argi("device", 0));
    const std::string out = arg("out", ".");
    const std::string which = arg("case");
    if (which == "cahnhilliard")
      return run_cahnhilliard(out);
    if (which == "brusselator")
      return run_brusselator(out);
    if (which == "coupled")
      return run_coupled(out);
    if (which == "etdrk4")
      return run_etdrk4(out);
    if (which == "secant")
      return run_secant(out);
    if (which == "broyden")
      return run_broyden(out);
    if (which == "explicit")
      return run_explicit(out);
    if (which == "kks")
      return run_kks(out);
    if (which == "semi_implicit")
      return run_semi_implicit(out);
    if (which == "coupled_pf_mech")
      return run_coupled_pf_mech(out);
    if (which == "mechanics")
      return run_mechanics(out);
    if (which == "gradient" || which == "gradient_square")
      return run_gradient(out, which == "gradient_square");
    mooseError("unknown case '", which, "'");
  }
  catch (const std::exception & e)
  {
    std::cerr << "*** ERROR ***\n" << e.what() << "\n";
    return 1;
  }
}
