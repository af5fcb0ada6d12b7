// marlin-hip-run: drives the host mirror (marlin_host.h) for the reference's regression cases, the way
// `marlin-opt -i case.i` drives the MOOSE objects.  Input is a flat list of key=value arguments that carry
// the same parameters as the reference's input files; output is one raw little-endian f64 file per
// buffer and time step (<out>/<buffer>.<frame>.bin), the data XDMFTensorOutput writes to HDF5.
//
//   marlin-hip-run problem=cahnhilliard dim=2 nx=20 ny=20 xmax=3 ymax=3 ic=c0.bin substeps=10 num_steps=10 dt=1e-3 out=dir
//        (test/tests/cahnhilliard/cahnhilliard.i)
//   marlin-hip-run problem=brusselator dim=2 nx=150 ny=150 xmax=2pi ymax=2pi ss=10 cs=0 order=2 num_steps=25 dt=0.5 out=dir
//        (test/tests/solvers/diagonal.i; writes brusselator.csv with the columns of the reference's CSV output)
//   marlin-hip-run problem=coupled|nl_coupled dim=2 nx=150 ny=150 xmax=2pi ymax=2pi ss=10 cs=0 order=2 num_steps=25 dt=10 out=dir
//        (test/tests/solvers/coupled.i: AdamsBashforthMoultonCoupled; nl_coupled.i: reciprocal-space ParsedComputes)
//   marlin-hip-run problem=rotating_grain_secant dim=2 nx=40 ny=40 xmax=12pi ymax=<..> ic=psi0.bin num_steps=10 out=dir
//        (test/tests/tensor_compute/rotating_grain_secant.i: SecantSolver + SwiftHohenbergLinear + iteration-adaptive dt)
//   marlin-hip-run problem=cahnhilliard_explicit dim=2 nx=50 ny=50 xmax=3 ymax=3 ic=c0.bin method=SHARP substeps=50 num_steps=20 dt=0.5
//        (test/tests/cahnhilliard/cahnhilliard_explicit_smooth.i: ForwardEulerSolver + DeAliasingTensor + reciprocal ParsedCompute)
//   marlin-hip-run problem=kks dim=2 nx=20 ny=20 xmin=-50 xmax=50 ymin=-50 ymax=50 c=c0.bin eta=eta0.bin psi=psi0.bin num_steps=10 dt=0.1
//        (test/tests/kks/KKS_no_flux_bc.i: ReciprocalMatDiffusion, ReciprocalAllenCahn, ParsedCompute derivatives, ABM order 3)
//   marlin-hip-run problem=mechanics dim=3 nx=16 ny=16 nz=16 substeps=10 num_steps=3 dt=0.01 l_tol=1e-2 nl_rel_tol=2e-2
//        nl_abs_tol=2e-2 out=dir          (test/tests/mechanics/mech3d.i)
//
// parallel_mode=FFT_SLAB nranks=P  (test/tests/cahnhilliard/tests:58-70 runs cahnhilliard.i this way under `mpiexec -n 2`):
//   one process per rank / GPU.  Without rank=..., this process is only the launcher: it starts P copies of itself (rank=0..P-1,
//   one job name) BEFORE anything touches the GPU and waits for them, the way mpiexec would.  Every rank reads the initial
//   condition of the GLOBAL grid and keeps its y-slab, and writes <out>/<buffer>.<frame>.rank<r>.bin = its slab (the reference's
//   rankNNNN files).  device=r (default: rank modulo the visible GPUs; device=0 puts all ranks on one GPU, which the library's
//   IPC transport supports), transport=1|2|3 (peer stores, copy engines, RCCL).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

#include <sys/wait.h>
#include <unistd.h>

#include "marlin_host.h"

using namespace marlin_host;

static std::map<std::string, std::string> g_args;
static std::string arg(const std::string & k, const std::string & dflt = "")
{
  auto it = g_args.find(k);
  return it == g_args.end() ? dflt : it->second;
}
static double argd(const std::string & k, double dflt) { return g_args.count(k) ? std::atof(g_args[k].c_str()) : dflt; }
static long argi(const std::string & k, long dflt) { return g_args.count(k) ? std::atol(g_args[k].c_str()) : dflt; }

static std::string g_rank_suffix;  // ".rank<r>" in FFT_SLAB runs

static void dump(const std::string & dir, const std::string & name, int frame, const DeviceTensor & t)
{
  const auto h = t.toHost();
  const std::string path = dir + "/" + name + "." + std::to_string(frame) + g_rank_suffix + ".bin";
  std::ofstream f(path, std::ios::binary);
  if (!f)
    mooseError("cannot write " + path);
  f.write(reinterpret_cast<const char *>(h.data()), sizeof(double) * h.size());
}

static std::vector<double> read_bin(const std::string & path, std::size_t count)
{
  std::vector<double> v(count);
  std::ifstream f(path, std::ios::binary);
  if (!f || !f.read(reinterpret_cast<char *>(v.data()), sizeof(double) * count))
    mooseError("cannot read " + std::to_string(count) + " doubles from " + path);
  return v;
}

// the rank's block of a global row-major field: [nx][y_begin .. y_end)[nz] in FFT_SLAB mode, [nx][y_begin .. y_end)[z_begin .. z_end) in
// FFT_PENCIL mode (parallel_mode NONE: the field itself)
static std::vector<double> local_block(const DomainAction & domain, const std::vector<double> & global, int ncomp = 1)
{
  if (!domain.isSlab() && !domain.isPencil())
    return global;
  const auto & g = domain.getShape();
  const auto & l = domain.getLocalShape();
  const auto & b = domain.getLocalBegin();
  const int dim = domain.getDim();
  const int64_t ny = g[1], nz = dim == 3 ? g[2] : 1;
  const int64_t lx = l[0], ly = l[1], lz = dim == 3 ? l[2] : 1, b0 = b[0], b1 = b[1], b2 = dim == 3 ? b[2] : 0;
  std::vector<double> out((std::size_t)(lx * ly * lz * ncomp));
  for (int64_t i = 0; i < lx; ++i)
    for (int64_t j = 0; j < ly; ++j)
    {
      const auto first = global.begin() + (((b0 + i) * ny + b1 + j) * nz + b2) * ncomp;
      std::copy(first, first + lz * ncomp, out.begin() + ((i * ly + j) * lz) * ncomp);
    }
  return out;
}

// precision=float32: the reference's per-run precision switch (src/utils/MarlinUtils.C:39-44, DomainAction.C:81,201; its published GPU
// numbers are float32 runs, doc/content/installation.md:36-43).  The same time loop -- num_steps steps of `substeps` substeps, AB2 by
// default, first step at first order (TensorProblem.C:455), history ring rotated per substep -- straight over mrl_ch_substeps_f32 with
// float buffers; no output (the wall time of the whole run is what is reported).  Serial contexts, built-in free-energy families.
static int run_cahnhilliard_f32(DomainAction & domain, const std::vector<double> & ic)
{
  if (domain.nranks() > 1)
    mooseError("precision=float32: serial runs only");
  mrl_ctx * ctx = domain.ctx();
  const int64_t nspec = mrl_ch_spec_elems_f32(ctx);
  if (nspec <= 0)
    mooseError("precision=float32: extents must be of {64, 100, 128, 200, 256, 400, 512} in three dimensions");
  const std::size_t n = ic.size();
  std::vector<float> icf(n);
  for (std::size_t i = 0; i < n; ++i)
    icf[i] = (float)ic[i];
  float *c[2] = {nullptr, nullptr}, *ring[3] = {nullptr, nullptr, nullptr};
  const int pred = (int)argi("predictor_order", 2);
  if (pred < 1 || pred > 3)
    paramError("predictor_order", "precision=float32: predictor_order 1 ... 3");
  for (int i = 0; i < 2; ++i)
    if (hipMalloc(reinterpret_cast<void **>(&c[i]), sizeof(float) * n) != hipSuccess)
      mooseError("hipMalloc failed");
  for (int i = 0; i < pred; ++i)
    if (hipMalloc(reinterpret_cast<void **>(&ring[i]), 2 * sizeof(float) * (std::size_t)nspec) != hipSuccess ||
        hipMemset(ring[i], 0, 2 * sizeof(float) * (std::size_t)nspec) != hipSuccess)
      mooseError("hipMalloc failed");
  if (hipMemcpy(c[0], icf.data(), sizeof(float) * n, hipMemcpyHostToDevice) != hipSuccess)
    mooseError("hipMemcpy (host to device) failed");
  mrl_ch_params p{};
  p.family = arg("free_energy", "DOUBLE_WELL") == "PFHUB" ? MRL_FE_PFHUB : MRL_FE_DOUBLE_WELL;
  p.coef[0] = argd("A", 0.1);
  p.coef[1] = argd("c_alpha", 0.3);
  p.coef[2] = argd("c_beta", 0.7);
  p.mobility = argd("mobility", 0.2);
  p.kappa = argd("kappa", -0.001);
  const int substeps = (int)argi("substeps", 1), num_steps = (int)argi("num_steps", 1);
  const double sub_dt = argd("dt", 1e-3) / substeps;
  int head = 0, n_old = 0;
  const auto t0 = std::chrono::steady_clock::now();
  for (int step = 0; step < num_steps; ++step)
  {
    // the history advances between time steps only from the second step on (TensorProblem.C:455: the whole first step is AB1 of the
    // ring's point of view: n_old grows inside the call)
    if (step > 0)
    {
      head = (head + 1) % pred;
      if (n_old < pred - 1)
        n_old += 1;
    }
    domain.check(mrl_ch_substeps_f32(ctx, &p, c[step % 2], c[1 - step % 2], ring, pred, &head, &n_old, pred, substeps, step > 0 ? 1 : 0, sub_dt));
  }
  domain.check(mrl_sync(ctx));
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::vector<float> end(n);
  if (hipMemcpy(end.data(), c[num_steps % 2], sizeof(float) * n, hipMemcpyDeviceToHost) != hipSuccess)
    mooseError("hipMemcpy (device to host) failed");
  double sum_c = 0.0, sum_c2 = 0.0;
  for (const float v : end)
  {
    sum_c += v;
    sum_c2 += (double)v * v;
  }
  const double updates = (double)n * substeps * num_steps;
  std::printf("{\"precision\": \"float32\", \"wall_s\": %.3f, \"grid_point_updates_per_s\": %.6e, \"frames\": 0, \"time\": %.17g, "
              "\"sum_c\": %.17g, \"sum_c2\": %.17g}\n",
              wall, updates / wall, argd("dt", 1e-3) * num_steps, sum_c, sum_c2);
  for (float * q : c)
    (void)hipFree(q);
  for (float * q : ring)
    if (q)
      (void)hipFree(q);
  return 0;
}

static int run_cahnhilliard(DomainAction & domain, const std::string & out)
{
  TensorProblem problem(domain);
  const std::size_t n = domain.getNumberOfCells();
  // RandomTensor IC: a file holding the global field, or ic=splitmix64 (the counter-based generator of bench.py: uniform [0.44, 0.56])
  std::vector<double> ic;
  if (arg("ic") == "splitmix64")
  {
    ic.resize((std::size_t)domain.getGlobalNumberOfCells());
    for (std::size_t i = 0; i < ic.size(); ++i)
    {
      uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + 0x9E3779B97F4A7C15ull;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z = z ^ (z >> 31);
      ic[i] = 0.44 + (0.56 - 0.44) * ((double)(z >> 11) * (1.0 / 9007199254740992.0));
    }
  }
  else
    ic = read_bin(arg("ic"), domain.getGlobalNumberOfCells());
  if (arg("precision", "float64") == "float32")
    return run_cahnhilliard_f32(domain, local_block(domain, ic));
  problem.getBuffer("c") = DeviceTensor::fromHost(local_block(domain, ic));
  problem.getBuffer("mu") = DeviceTensor::zeros(n);                        // ConstantTensor
  AdamsBashforthMoulton::Params p;
  p.substeps = (unsigned int)argi("substeps", 1);
  p.predictor_order = (std::size_t)argi("predictor_order", 2);
  p.ch.family = arg("free_energy", "DOUBLE_WELL") == "PFHUB" ? MRL_FE_PFHUB : MRL_FE_DOUBLE_WELL;
  p.ch.coef[0] = argd("A", 0.1);          // expression = '0.1*c^2*(c-1)^2'
  p.ch.coef[1] = argd("c_alpha", 0.3);
  p.ch.coef[2] = argd("c_beta", 0.7);
  p.ch.mobility = argd("mobility", 0.2);  // ReciprocalLaplacianFactor factor
  p.ch.kappa = argd("kappa", -0.001);     // ReciprocalLaplacianSquareFactor factor
  p.spectral_carry = argi("spectral_carry", 0) != 0;
  p.substep_calls = argi("substep_calls", 0) != 0;
  // expression=... : the [mu] ParsedCompute block of the input file (expression + derivatives = c) instead of a
  // built-in family; the derivative is taken symbolically and compiled into the solver's forward z pass
  mrl_parsed * parsed = nullptr;
  if (!arg("expression").empty())
  {
    const std::string expr = arg("expression");
    const char * in[] = {"c"};
    const char * dv[] = {"c"};
    if (mrl_parsed_create(domain.ctx(), &parsed, expr.c_str(), 1, in, nullptr, 0, nullptr, nullptr, 1, dv, 0, 0) != MRL_OK)
      paramError("expression", mrl_last_error(domain.ctx()));
    p.ch.family = MRL_FE_PARSED;
    p.ch.parsed = parsed;
  }
  std::unique_ptr<TensorSolver> solver;
  if (arg("integrator") == "FFTSemiImplicit")
  {
    // the legacy [TensorTimeIntegrators] form of the same scheme: explicit compute group + FFTSemiImplicit (history_size 1)
    ReciprocalLaplacianFactor(problem, "Mbar", "Mbar", p.ch.mobility).computeBuffer();
    ReciprocalLaplacianFactor(problem, "kappabarbar", "kappabarbar", p.ch.kappa, 2).computeBuffer();
    auto root = std::make_shared<ComputeGroup>(problem, "root");
    ParsedCompute::Params pm;
    pm.buffer = "mu";
    pm.expression = arg("expression", "0.1*c^2*(c-1)^2");
    pm.inputs = {"c"};
    pm.derivatives = {"c"};
    root->add(std::make_shared<ParsedCompute>(problem, "mu", pm));
    root->add(std::make_shared<ForwardFFT>(problem, "mubar", "mubar", "mu"));
    ParsedCompute::Params pn;
    pn.buffer = "Mbarmubar";
    pn.expression = "Mbar*mubar";
    pn.inputs = {"Mbar", "mubar"};
    pn.complex_inputs = {"mubar"};
    pn.reciprocal = true;
    root->add(std::make_shared<ParsedCompute>(problem, "Mbarmubar", pn));
    root->add(std::make_shared<ForwardFFT>(problem, "cbar", "cbar", "c"));
    auto ti = std::make_shared<FFTSemiImplicit>(problem, "c", "c", "cbar", "kappabarbar", "Mbarmubar", 1);
    solver = std::make_unique<TimeIntegratorSolver>(problem, "solver", p.substeps, root,
                                                    std::vector<std::shared_ptr<TensorOperatorBase>>{ti});
  }
  else
    solver = std::make_unique<AdamsBashforthMoulton>(problem, "solver", p);
  Transient ex(problem, *solver, argd("dt", 1e-3));
  // dt_sequence=a,b,c,...: a [TimeStepper] that hands out these step sizes in turn (the last one repeats)
  std::vector<double> dts;
  {
    std::string seq = arg("dt_sequence");
    while (!seq.empty())
    {
      const auto comma = seq.find(',');
      dts.push_back(std::atof(seq.substr(0, comma).c_str()));
      seq = comma == std::string::npos ? "" : seq.substr(comma + 1);
    }
  }
  if (!dts.empty())
    ex.setTimeStepper([dts](int t_step) { return dts[std::min<std::size_t>((std::size_t)t_step - 1, dts.size() - 1)]; });
  if (arg("output") == "xdmf" || arg("output") == "none")
  {
    // [TensorOutputs] XDMFTensorOutput buffer = 'c mu' (examples/cahn_hilliard/cahnhilliard2.i:41-51) in its raw-binary mode: the
    // fields leave the GPU asynchronously and are written by a thread while the next time step computes; `output=none` runs the
    // same steps without any output.  Prints the whole-run wall time (what doc/content/installation.md:36-42 tabulates).
    std::unique_ptr<XDMFTensorOutput> xdmf;
    if (arg("output") == "xdmf")
    {
      XDMFTensorOutput::Params op;
      op.buffer = {"c", "mu"};
      op.file_base = out + "/" + arg("file_base", "cahnhilliard_out");
      op.enable_hdf5 = arg("enable_hdf5", "false") == "true";   // XDMFTensorOutput.C:39
      if (!arg("output_mode").empty())                             // e.g. output_mode=NODE,CELL (one entry per buffer: c, mu)
      {
        std::string m = arg("output_mode");
        while (!m.empty())
        {
          const auto comma = m.find(',');
          op.output_mode.push_back(m.substr(0, comma));
          m = comma == std::string::npos ? "" : m.substr(comma + 1);
        }
      }
      xdmf = std::make_unique<XDMFTensorOutput>(problem, op);
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (xdmf && arg("output_initial", "false") == "true")   // execute_on = 'INITIAL TIMESTEP_END' (cahnhilliard.i): frame 0 = the initial condition
      xdmf->startOutput();
    ex.execute((int)argi("num_steps", 1), [&](int) {
      if (xdmf)
        xdmf->startOutput();
    });
    if (xdmf)
      xdmf->waitForCompletion();
    domain.check(mrl_sync(domain.ctx()));
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double updates = (double)domain.getGlobalNumberOfCells() * (double)p.substeps * (double)argi("num_steps", 1);
    // global checksums of the final field (sum c, sum c^2 over all ranks): lets two runs be compared without writing the field
    double sum_c = 0.0, sum_c2 = 0.0;
    {
      const DeviceTensor c_end = problem.getBuffer("c");
      domain.check(mrl_sum(domain.ctx(), c_end.data(), (int64_t)c_end.numel(), &sum_c));
      domain.check(mrl_dot(domain.ctx(), c_end.data(), c_end.data(), (int64_t)c_end.numel(), &sum_c2));
    }
    if (domain.rank() == 0)
      std::printf("{\"wall_s\": %.3f, \"grid_point_updates_per_s\": %.6e, \"frames\": %d, \"seconds_in_writer_thread\": %.3f, \"time\": %.17g, "
                  "\"sum_c\": %.17g, \"sum_c2\": %.17g}\n",
                  wall, updates / wall, xdmf ? xdmf->frames() : 0, xdmf ? xdmf->secondsWriting() : 0.0, problem.time(), sum_c, sum_c2);
    mrl_parsed_destroy(parsed);
    return 0;
  }
  dump(out, "c", 0, problem.getBuffer("c"));
  ex.execute((int)argi("num_steps", 1), [&](int step) {
    dump(out, "c", step, problem.getBuffer("c"));
    dump(out, "mu", step, problem.getBuffer("mu"));
  });
  mrl_parsed_destroy(parsed);
  return 0;
}

// test/tests/tensor_compute/coupled_pf_mech.i: Cahn-Hilliard + the elastic chemical potential of a homogeneous solid with the
// eigenstrain e0*c (FFTQuasistaticElasticity, FFTElasticChemicalPotential), legacy FFTSemiImplicit integrator.  Operators are
// added in the dependency-resolved order of the input's [Solve] group.
static int run_coupled_pf_mech(DomainAction & domain, const std::string & out)
{
  TensorProblem problem(domain);
  const std::size_t n = domain.getNumberOfCells();
  problem.getBuffer("c") = DeviceTensor::fromHost(read_bin(arg("ic"), n));
  for (const char * d : {"disp_x", "disp_y", "disp_z"})
    problem.getBuffer(d) = DeviceTensor::zeros(n);  // RandomTensor min = max = 0
  const double lame_mu = argd("mu", 50.0), lambda = argd("lambda", 100.0), e0 = argd("e0", 0.02);
  ReciprocalLaplacianFactor(problem, "Mbar", "Mbar", argd("mobility", 0.2)).computeBuffer();
  ReciprocalLaplacianFactor(problem, "kappabarbar", "kappabarbar", argd("kappa", -0.001), 2).computeBuffer();
  const std::vector<std::string> disp = {"disp_x", "disp_y", "disp_z"};
  auto root = std::make_shared<ComputeGroup>(problem, "Solve");
  ParsedCompute::Params pm;
  pm.buffer = "mu";
  pm.expression = arg("expression", "0.1*c^2*(c-1)^2");
  pm.inputs = {"c"};
  pm.derivatives = {"c"};
  root->add(std::make_shared<ParsedCompute>(problem, "mu", pm));
  root->add(std::make_shared<ForwardFFT>(problem, "mubar", "mubar", "mu"));
  root->add(std::make_shared<ForwardFFT>(problem, "cbar", "cbar", "c"));
  root->add(std::make_shared<FFTQuasistaticElasticity>(problem, "qsmech", disp, "cbar", lame_mu, lambda, e0));
  root->add(std::make_shared<FFTElasticChemicalPotential>(problem, "mumechbar", "mumechbar", disp, "cbar", lame_mu, lambda, e0));
  root->add(std::make_shared<InverseFFT>(problem, "mumech", "mumech", "mumechbar"));
  ParsedCompute::Params pn;
  pn.buffer = "Mbarmubar";
  pn.expression = "Mbar*(mubar+mumechbar)";
  pn.inputs = {"Mbar", "mubar", "mumechbar"};
  pn.complex_inputs = {"mubar", "mumechbar"};
  pn.reciprocal = true;
  root->add(std::make_shared<ParsedCompute>(problem, "Mbarmubar", pn));
  auto ti = std::make_shared<FFTSemiImplicit>(problem, "c", "c", "cbar", "kappabarbar", "Mbarmubar", 1);
  TimeIntegratorSolver solver(problem, "solver", (unsigned int)argi("substeps", 10), root,
                              std::vector<std::shared_ptr<TensorOperatorBase>>{ti});
  Transient ex(problem, solver, argd("dt", 0.1));
  dump(out, "c", 0, problem.getBuffer("c"));
  ex.execute((int)argi("num_steps", 2), [&](int step) {
    dump(out, "c", step, problem.getBuffer("c"));
    dump(out, "mumech", step, problem.getBuffer("mumech"));
    for (const auto & d : disp)
      dump(out, d, step, problem.getBuffer(d));
  });
  return 0;
}

static int run_mechanics(DomainAction & domain, const std::string & out)
{
  TensorProblem problem(domain);
  const int dim = domain.getDim();
  const auto & shape = domain.getShape();
  const std::size_t n = domain.getGlobalNumberOfCells();
  // phase = prod_d (cos(x_d)/2 + 0.5); K = (1-phase)*Ka + phase*Kb; mu likewise   (mech3d.i:14-35)
  std::vector<std::vector<double>> ax;
  for (int d = 0; d < dim; ++d)
    ax.push_back(domain.getAxis(d));
  const double Ka = argd("Ka", 1.0), Kb = argd("Kb", 10.0), mua = argd("mua", 0.5), mub = argd("mub", 5.0);
  std::vector<double> K(n), mu(n), F(n * dim * dim, 0.0), phase_field(n);
  for (std::size_t e = 0; e < n; ++e)
  {
    std::size_t r = e;
    std::vector<int64_t> idx(dim);
    for (int d = dim - 1; d >= 0; --d)
    {
      idx[d] = r % shape[d];
      r /= shape[d];
    }
    double phase = 1.0;
    for (int d = 0; d < dim; ++d)
    {
      const double term = std::cos(ax[d][idx[d]]) / 2.0 + 0.5;
      phase = d == 0 ? term : phase * term;
    }
    K[e] = (1.0 - phase) * Ka + phase * Kb;
    mu[e] = (1.0 - phase) * mua + phase * mub;
    phase_field[e] = phase;
    for (int i = 0; i < dim; ++i)
      F[e * dim * dim + i * dim + i] = 1.0;  // RankTwoIdentity
  }
  problem.getBuffer("phase") = DeviceTensor::fromHost(local_block(domain, phase_field));
  problem.getBuffer("K") = DeviceTensor::fromHost(local_block(domain, K));
  problem.getBuffer("mu") = DeviceTensor::fromHost(local_block(domain, mu));
  problem.getBuffer("F") = DeviceTensor::fromHost(local_block(domain, F, dim * dim));

  auto root = std::make_shared<ComputeGroup>(problem, "root");
  root->add(std::make_shared<MacroscopicShearTensor>(problem, "applied_strain", "applied_strain", "F"));
  FFTMechanics::Params mp;
  mp.applied_macroscopic_strain = "applied_strain";
  mp.l_tol = argd("l_tol", 1e-2);
  mp.nl_rel_tol = argd("nl_rel_tol", 1e-5);
  mp.nl_abs_tol = argd("nl_abs_tol", 1e-8);
  mp.l_max_its = argi("l_max_its", 0);
  mp.nl_max_its = (unsigned int)argi("nl_max_its", 100);
  auto mech = std::make_shared<FFTMechanics>(problem, "mech", mp);
  root->add(mech);
  ForwardEulerSolver solver(problem, "solver", (unsigned int)argi("substeps", 1), root);
  solver.addForwardBuffer("F", "Fnew");
  Transient ex(problem, solver, argd("dt", 0.01));
  // [Postprocess] group: evaluated before the outputs of a time step
  ComputeDisplacements displacements(problem, "displacements", "disp", "F");
  ComputeVonMisesStress vonmises(problem, "vonmises", "sV");
  // output=xdmf: the [TensorOutputs] block of mech3d.i:95-103 -- buffer = 'disp sV F phase', output_mode = 'OVERSIZED_NODAL CELL CELL
  // NODE', enable_hdf5 = true, execute_on TIMESTEP_END (serial domains: disp is a global field)
  std::unique_ptr<XDMFTensorOutput> xdmf;
  if (arg("output") == "xdmf")
  {
    XDMFTensorOutput::Params op;
    op.buffer = {"disp", "sV", "F", "phase"};
    op.components = {dim, 1, dim * dim, 1};
    op.output_mode = {"OVERSIZED_NODAL", "CELL", "CELL", "NODE"};
    op.file_base = out + "/" + arg("file_base", "mech_out");
    op.enable_hdf5 = arg("enable_hdf5", "true") == "true";
    // (the displacement buffer has to exist with its final size before the output object sizes its staging area)
    displacements.computeBuffer();
    xdmf = std::make_unique<XDMFTensorOutput>(problem, op);
  }
  ex.execute((int)argi("num_steps", 1), [&](int step) {
    if (!domain.isSlab())  // (ComputeDisplacements interpolates over the global grid: serial domains only)
    {
      displacements.computeBuffer();
      dump(out, "disp", step - 1, problem.getBuffer("disp"));
    }
    vonmises.computeBuffer();
    dump(out, "sV", step - 1, problem.getBuffer("sV"));
    dump(out, "F", step - 1, problem.getBuffer("F"));  // frame 0 = end of step 1 (output on TIMESTEP_END only)
    if (xdmf)
      xdmf->startOutput();
    const auto & st = mech->stats();
    std::printf("step %d: newton_its=%d cg_its_total=%d |R|=%.6e\n", step, st.newton_its, st.cg_its_total, st.last_anorm);
  });
  if (xdmf)
    xdmf->waitForCompletion();
  return 0;
}

// test/tests/solvers/diagonal.i: 2-D Brusselator, ABM orders 1-4 with 0-2 Adams-Moulton corrector steps
static int run_brusselator(DomainAction & domain, const std::string & out)
{
  TensorProblem problem(domain);
  const std::vector<std::pair<std::string, double>> consts = {{"A", argd("A", 1.0)}, {"B", argd("B", 3.5)}};
  // [Initialize]
  ParsedCompute::Params ic;
  ic.buffer = "u";
  ic.expression = arg("u0", "sin(x)*sin(y)");
  ic.extra_symbols = true;
  ParsedCompute(problem, "u", ic).computeBuffer();
  problem.getBuffer("v") = DeviceTensor::zeros(domain.getNumberOfCells());
  ReciprocalLaplacianFactor(problem, "Du", "Du", argd("Du", 1e-2)).computeBuffer();
  ReciprocalLaplacianFactor(problem, "Dv", "Dv", argd("Dv", 1e-3)).computeBuffer();
  // [Solve] in dependency order
  auto root = std::make_shared<ComputeGroup>(problem, "root");
  root->add(std::make_shared<ForwardFFT>(problem, "u_bar", "u_bar", "u"));
  root->add(std::make_shared<ForwardFFT>(problem, "v_bar", "v_bar", "v"));
  ParsedCompute::Params su;
  su.buffer = "source_u";
  su.expression = arg("source_u", "A - (B+1)*u +u^2*v");
  su.inputs = {"u", "v"};
  su.constants = consts;
  root->add(std::make_shared<ParsedCompute>(problem, "source_u", su));
  root->add(std::make_shared<ForwardFFT>(problem, "source_u_bar", "source_u_bar", "source_u"));
  ParsedCompute::Params sv = su;
  sv.buffer = "source_v";
  sv.expression = arg("source_v", "B*u - u^2*v");
  root->add(std::make_shared<ParsedCompute>(problem, "source_v", sv));
  root->add(std::make_shared<ForwardFFT>(problem, "source_v_bar", "source_v_bar", "source_v"));
  const std::size_t order = (std::size_t)argi("order", 2);
  SplitOperatorABM solver(problem, "solver", (unsigned int)argi("ss", 10), root,
                          {{"u", "u_bar", "Du", "source_u_bar"}, {"v", "v_bar", "Dv", "source_v_bar"}}, order, order,
                          (std::size_t)argi("cs", 0));
  Transient ex(problem, solver, argd("dt", 0.5));
  double volume = 1.0;
  for (int d = 0; d < domain.getDim(); ++d)
    volume *= domain.getExtent(d);
  std::ofstream csv(out + "/brusselator.csv");
  csv.precision(17);
  csv << "time,U,V,u_max,u_min,v_max,v_min\n0,0,0,0,0,0,0\n";
  ex.execute((int)argi("num_steps", 25), [&](int) {
    double umin, umax, vmin, vmax;
    TensorPostprocessors::extreme(domain, problem.getBuffer("u"), umin, umax);
    TensorPostprocessors::extreme(domain, problem.getBuffer("v"), vmin, vmax);
    csv << problem.time() << ',' << TensorPostprocessors::integral(domain, problem.getBuffer("u"), volume) << ','
        << TensorPostprocessors::integral(domain, problem.getBuffer("v"), volume) << ',' << umax << ',' << umin << ','
        << vmax << ',' << vmin << "\n";
  });
  return 0;
}

// test/tests/solvers/coupled.i (AdamsBashforthMoultonCoupled, dense operator [[D1, D2], [D2, D1]], zero nonlinear terms) and
// nl_coupled.i (diagonal AdamsBashforthMoulton with the cross terms as reciprocal-space ParsedComputes Du = D2*v_bar, Dv = D2*u_bar)
static int run_coupled(DomainAction & domain, const std::string & out, bool nonlinear)
{
  TensorProblem problem(domain);
  ParsedCompute::Params ic;
  ic.buffer = "u";
  ic.expression = arg("u0", "sin(x)*sin(y)");
  ic.extra_symbols = true;
  ParsedCompute(problem, "u", ic).computeBuffer();
  ic.buffer = "v";
  ic.expression = arg("v0", "cos(x)*cos(y)");
  ParsedCompute(problem, "v", ic).computeBuffer();
  problem.getBuffer("zero") = DeviceTensor::zeros(2 * domain.getReciprocalSize());  // ConstantReciprocalTensor
  ReciprocalLaplacianFactor(problem, "D1", "D1", argd("D1", 1e-2)).computeBuffer();
  ReciprocalLaplacianFactor(problem, "D2", "D2", argd("D2", 1e-3)).computeBuffer();
  auto root = std::make_shared<ComputeGroup>(problem, "root");
  root->add(std::make_shared<ForwardFFT>(problem, "u_bar", "u_bar", "u"));
  root->add(std::make_shared<ForwardFFT>(problem, "v_bar", "v_bar", "v"));
  const std::size_t order = (std::size_t)argi("order", 2);
  const auto ss = (unsigned int)argi("ss", 10);
  const auto cs = (std::size_t)argi("cs", 0);
  std::unique_ptr<TensorSolver> solver;
  if (nonlinear)
  {
    ParsedCompute::Params du;
    du.buffer = "Du";
    du.expression = "D2*v_bar";
    du.inputs = {"D2", "v_bar"};
    du.complex_inputs = {"v_bar"};
    du.reciprocal = true;
    root->add(std::make_shared<ParsedCompute>(problem, "Du", du));
    ParsedCompute::Params dv = du;
    dv.buffer = "Dv";
    dv.expression = "D2*u_bar";
    dv.inputs = {"D2", "u_bar"};
    dv.complex_inputs = {"u_bar"};
    root->add(std::make_shared<ParsedCompute>(problem, "Dv", dv));
    solver = std::make_unique<SplitOperatorABM>(
        problem, "solver", ss, root,
        std::vector<SplitOperatorABM::VariableNames>{{"u", "u_bar", "D1", "Du"}, {"v", "v_bar", "D1", "Dv"}}, order, order, cs);
  }
  else
    solver = std::make_unique<AdamsBashforthMoultonCoupled>(
        problem, "solver", ss, root,
        std::vector<SplitOperatorABM::VariableNames>{{"u", "u_bar", "D1", "zero"}, {"v", "v_bar", "D1", "zero"}},
        std::vector<AdamsBashforthMoultonCoupled::OffDiagonal>{{1, 0, "D2"}, {0, 1, "D2"}}, false, order, order, cs,
        (int)argi("flags", 0));
  Transient ex(problem, *solver, argd("dt", 10.0));
  double volume = 1.0;
  for (int d = 0; d < domain.getDim(); ++d)
    volume *= domain.getExtent(d);
  std::ofstream csv(out + (nonlinear ? "/nl_coupled.csv" : "/coupled.csv"));
  csv.precision(17);
  csv << "time,U,V,u_max,u_min,v_max,v_min\n0,0,0,0,0,0,0\n";
  ex.execute((int)argi("num_steps", 25), [&](int) {
    double umin, umax, vmin, vmax;
    TensorPostprocessors::extreme(domain, problem.getBuffer("u"), umin, umax);
    TensorPostprocessors::extreme(domain, problem.getBuffer("v"), vmin, vmax);
    csv << problem.time() << ',' << TensorPostprocessors::integral(domain, problem.getBuffer("u"), volume) << ','
        << TensorPostprocessors::integral(domain, problem.getBuffer("v"), volume) << ',' << umax << ',' << umin << ','
        << vmax << ',' << vmin << "\n";
  });
  return 0;
}

// test/tests/tensor_compute/rotating_grain_secant.i: 2-D Swift-Hohenberg with the SecantSolver and iteration-adaptive dt;
// ic= holds psi at t = 0 (the reference evaluates a MOOSE ParsedFunction there)
static int run_rotating_grain_secant(DomainAction & domain, const std::string & out)
{
  TensorProblem problem(domain);
  problem.getBuffer("psi") = DeviceTensor::fromHost(read_bin(arg("ic"), (std::size_t)domain.getNumberOfCells()));
  SwiftHohenbergLinear(problem, "linear", "linear", argd("r", 0.025), argd("alpha", 1.0)).computeBuffer();
  auto root = std::make_shared<ComputeGroup>(problem, "root");
  ParsedCompute::Params p3;
  p3.buffer = "psi3";
  p3.expression = arg("expression", "0.20*psi^2-psi^3");
  p3.inputs = {"psi"};
  root->add(std::make_shared<ParsedCompute>(problem, "psi3", p3));
  root->add(std::make_shared<ForwardFFT>(problem, "psibar", "psibar", "psi"));
  root->add(std::make_shared<ForwardFFT>(problem, "psi3bar", "psi3bar", "psi3"));
  SecantSolver::Params sp;
  sp.substeps = (unsigned int)argi("substeps", 3);
  sp.max_iterations = (unsigned int)argi("max_iterations", 30);
  sp.relative_tolerance = argd("relative_tolerance", 1e-9);
  sp.absolute_tolerance = argd("absolute_tolerance", 1e-9);
  sp.damping = argd("damping", 1.0);
  sp.dt_epsilon = argd("dt_epsilon", 1e-4);
  sp.verbose = argi("verbose", 0) != 0;
  SecantSolver solver(problem, "solver", root, {{"psi", "psibar", "linear", "psi3bar"}}, sp);
  if (!arg("predictor_scale").empty())  // [TensorSolver/Predictors] type = LinearTensorPredictor, buffer = psi
    solver.addPredictor(std::make_shared<LinearTensorPredictor>(problem, "psi", argd("predictor_scale", 1.0)));
  TensorSolveIterationAdaptiveDT ts(solver, argd("dt", 1.0), (unsigned int)argi("ts_min_iterations", 100),
                                    (unsigned int)argi("ts_max_iterations", 400), argd("growth_factor", 1.4),
                                    argd("cutback_factor", 0.9), argd("dtmax", 500.0));
  Transient ex(problem, solver, argd("dt", 1.0));
  ex.setTimeStepper([&](int t_step) { return ts.computeDT(t_step); });
  dump(out, "psi", 0, problem.getBuffer("psi"));
  ex.execute((int)argi("num_steps", 10), [&](int step) {
    dump(out, "psi", step, problem.getBuffer("psi"));
    std::printf("step %d: dt=%.17g iterations=%u converged=%d\n", step, problem.dt(), solver.getIterations(), (int)solver.isConverged());
  });
  return 0;
}

// test/tests/postprocessors/postprocessors.i: c = -x+y+0.3 on 40^2, [0,2]x[0,3]; ForwardEulerSolver on u = 0 with du/dt = c_bar,
// 10 substeps; min/max/average/integral/ReciprocalIntegral of c and the root group's execution count
static int run_postprocessors(DomainAction & domain, const std::string & out)
{
  TensorProblem problem(domain);
  ParsedCompute::Params pc;
  pc.buffer = "c";
  pc.expression = arg("expression", "-x+y+0.3");
  pc.extra_symbols = true;
  ParsedCompute(problem, "c", pc).computeBuffer();
  ForwardFFT(problem, "c_bar", "c_bar", "c").computeBuffer();
  problem.getBuffer("u") = DeviceTensor::zeros(domain.getNumberOfCells());  // ConstantTensor real = 0
  auto root = std::make_shared<ComputeGroup>(problem, "root");
  root->add(std::make_shared<ForwardFFT>(problem, "test", "u_bar", "u"));
  ForwardEulerSolver solver(problem, "solver", (unsigned int)argi("substeps", 10), root, {{"u", "u_bar", "c_bar"}});
  Transient ex(problem, solver, argd("dt", 1.0));
  double volume = 1.0;
  for (int d = 0; d < domain.getDim(); ++d)
    volume *= domain.getExtent(d);
  std::ofstream csv(out + "/postprocessors.csv");
  csv.precision(17);
  csv << "time,max_c,min_c,avg_c,int_c,int_c_bar,count,int_u\n";
  auto row = [&]() {
    double mn, mx;
    TensorPostprocessors::extreme(domain, problem.getBuffer("c"), mn, mx);
    csv << problem.time() << ',' << mx << ',' << mn << ',' << TensorPostprocessors::average(domain, problem.getBuffer("c")) << ','
        << TensorPostprocessors::integral(domain, problem.getBuffer("c"), volume) << ','
        << TensorPostprocessors::reciprocalIntegral(domain, problem.getBuffer("c_bar"), volume) << ',' << root->getComputeCount()
        << ',' << TensorPostprocessors::integral(domain, problem.getBuffer("u"), volume) << "\n";
  };
  row();                                                                     // execute_on = INITIAL
  ex.execute((int)argi("num_steps", 0), [&](int) { row(); });
  return 0;
}

// test/tests/cahnhilliard/cahnhilliard_explicit_smooth.i: explicit Euler Cahn-Hilliard, the k-space right-hand side filtered by a
// DeAliasingTensor (method=SHARP|HOULI; method=NONE = cahnhilliard_explicit.i); ic= holds the seed-0 RandomTensor field
static int run_cahnhilliard_explicit(DomainAction & domain, const std::string & out)
{
  TensorProblem problem(domain);
  const std::size_t n = domain.getNumberOfCells();
  problem.getBuffer("c") = DeviceTensor::fromHost(read_bin(arg("ic"), n));
  problem.getBuffer("mu") = DeviceTensor::zeros(n);                                      // ConstantTensor
  problem.getBuffer("dc_dt_bar") = DeviceTensor::zeros(2 * domain.getReciprocalSize());  // ConstantReciprocalTensor
  ReciprocalLaplacianFactor(problem, "Mbar", "Mbar", argd("mobility", 0.2)).computeBuffer();
  ReciprocalLaplacianFactor(problem, "Mkappabarbar", "Mkappabarbar", argd("Mkappa", 0.2 * 1e-4), 2).computeBuffer();
  const std::string method = arg("method", "NONE");
  if (method != "NONE")
    DeAliasingTensor(problem, "smooth", "smooth", method, argd("p", 16.0), argd("alpha", 36.0)).computeBuffer();
  auto root = std::make_shared<ComputeGroup>(problem, "cahn_hilliard");
  ParsedCompute::Params pm;
  pm.buffer = "mu";
  pm.expression = arg("expression", "0.1*c^2*(c-1)^2");
  pm.inputs = {"c"};
  pm.derivatives = {"c"};
  root->add(std::make_shared<ParsedCompute>(problem, "mu", pm));
  root->add(std::make_shared<ForwardFFT>(problem, "mubar", "mubar", "mu"));
  root->add(std::make_shared<ForwardFFT>(problem, "cbar", "cbar", "c"));
  ParsedCompute::Params pd;
  pd.buffer = "dc_dt_bar";
  pd.reciprocal = true;
  pd.complex_inputs = {"mubar", "cbar"};
  if (method != "NONE")
  {
    pd.expression = "smooth * (Mbar*mubar - Mkappabarbar*cbar)";
    pd.inputs = {"Mbar", "mubar", "Mkappabarbar", "cbar", "smooth"};
  }
  else
  {
    pd.expression = "Mbar*mubar - Mkappabarbar*cbar";
    pd.inputs = {"Mbar", "mubar", "Mkappabarbar", "cbar"};
  }
  root->add(std::make_shared<ParsedCompute>(problem, "dc_dt_bar", pd));
  ForwardEulerSolver solver(problem, "solver", (unsigned int)argi("substeps", 50), root, {{"c", "cbar", "dc_dt_bar"}});
  Transient ex(problem, solver, argd("dt", 0.5));
  dump(out, "c", 0, problem.getBuffer("c"));
  ex.execute((int)argi("num_steps", 20), [&](int step) {
    dump(out, "c", step, problem.getBuffer("c"));
    dump(out, "mu", step, problem.getBuffer("mu"));
  });
  return 0;
}

// test/tests/kks/KKS_no_flux_bc.i: two-variable Kim-Kim-Suzuki model with the smooth boundary method: ParsedCompute derivatives of
// the Gibbs energy, ReciprocalMatDiffusion / ReciprocalAllenCahn, AdamsBashforthMoulton order 3 with 1000 substeps per step;
// c=, eta=, psi= hold the initial fields (the reference evaluates MOOSE parsed functions there)
static int run_kks(DomainAction & domain, const std::string & out)
{
  TensorProblem problem(domain);
  const std::size_t n = domain.getNumberOfCells();
  for (const char * b : {"c", "eta", "psi"})
    problem.getBuffer(b) = DeviceTensor::fromHost(read_bin(arg(b), n));
  auto constant = [&](const char * b, double v) { problem.getBuffer(b) = DeviceTensor::fromHost(std::vector<double>(n, v)); };
  constant("M", argd("M", 5.0));
  constant("L", argd("L", 5.0));
  constant("L_kappa", argd("L", 5.0) * argd("kappa_eta", 5.0));
  // ${F} after MOOSE's textual substitution of h_eta, rho_sq, w, c0_a, c0_b (KKS_no_flux_bc.i:24-25)
  const std::string h = "eta^3*(6*eta^2-15*eta+10)";
  const std::string F = arg("F", h + "*(2*((c - (1-" + h + ")*(0.7 - 0.3))-0.3)^2) + (1-" + h + ")*(2*((c + (" + h +
                                     ")*(0.7 - 0.3))-0.7)^2 ) + 1*(eta^2)*(1-eta)^2");
  auto root = std::make_shared<ComputeGroup>(problem, "root");
  root->add(std::make_shared<ForwardFFT>(problem, "cbar", "cbar", "c"));
  root->add(std::make_shared<ForwardFFT>(problem, "etabar", "etabar", "eta"));
  ParsedCompute::Params pm;
  pm.buffer = "mu";
  pm.expression = F;
  pm.inputs = {"c", "eta"};
  pm.derivatives = {"c"};
  root->add(std::make_shared<ParsedCompute>(problem, "mu", pm));
  root->add(std::make_shared<ReciprocalMatDiffusion>(problem, "div_J", "div_J", "mu", "M", "psi"));
  ParsedCompute::Params po;
  po.buffer = "domega_chem_deta";
  po.expression = F + " - mu*c";
  po.inputs = {"mu", "c", "eta"};
  po.derivatives = {"eta"};
  root->add(std::make_shared<ParsedCompute>(problem, "domega_chem_deta", po));
  root->add(std::make_shared<ReciprocalAllenCahn>(problem, "AC_bulk", "AC_bulk", "domega_chem_deta", "L", "psi"));
  root->add(std::make_shared<ReciprocalMatDiffusion>(problem, "kappa_grad_eta", "kappa_grad_eta", "eta", "L_kappa", "psi"));
  ParsedCompute::Params pa;
  pa.buffer = "AC_bar";
  pa.expression = "kappa_grad_eta + AC_bulk";
  pa.inputs = {"AC_bulk", "kappa_grad_eta"};
  pa.complex_inputs = {"AC_bulk", "kappa_grad_eta"};
  pa.reciprocal = true;
  root->add(std::make_shared<ParsedCompute>(problem, "AC_bar", pa));
  const std::size_t order = (std::size_t)argi("predictor_order", 3);
  // linear_reciprocal = 'zero zero': dividing by (1 - dt * 0) is the identity
  SplitOperatorABM solver(problem, "solver", (unsigned int)argi("substeps", 1000), root,
                          {{"c", "cbar", "0", "div_J"}, {"eta", "etabar", "0", "AC_bar"}}, order, order, 0);
  Transient ex(problem, solver, argd("dt", 0.1));
  double volume = 1.0;
  for (int d = 0; d < domain.getDim(); ++d)
    volume *= domain.getExtent(d);
  std::ofstream csv(out + "/kks.csv");
  csv.precision(17);
  csv << "time,total_C,total_eta\n";
  auto row = [&](int frame) {
    csv << problem.time() << ',' << TensorPostprocessors::integral(domain, problem.getBuffer("c"), volume) << ','
        << TensorPostprocessors::integral(domain, problem.getBuffer("eta"), volume) << "\n";
    for (const char * b : {"c", "eta", "mu"})
      if (problem.getBuffer(b).defined())
        dump(out, b, frame, problem.getBuffer(b));
  };
  row(0);
  ex.execute((int)argi("num_steps", 10), [&](int step) { row(step); });
  return 0;
}

// test/tests/postprocessors/interface_velocity.i: c = sin(x + 0.2 t) re-evaluated every step (no solver: the compute group runs at
// timeOld), TensorInterfaceVelocityPostprocessor on c and its previous state
static int run_interface_velocity(DomainAction & domain, const std::string & out)
{
  TensorProblem problem(domain);
  auto root = std::make_shared<ComputeGroup>(problem, "root");
  ParsedCompute::Params pc;
  pc.buffer = "c";
  pc.expression = arg("expression", "sin(x+0.2*t)");
  pc.extra_symbols = true;
  root->add(std::make_shared<ParsedCompute>(problem, "c", pc));
  TensorInterfaceVelocityPostprocessor v(problem, "c");
  ForwardEulerSolver solver(problem, "none", 1, root);   // no [TensorSolver]: TensorProblem::execute runs the computes (TensorProblem.C:176-187)
  Transient ex(problem, solver, argd("dt", 0.01));
  std::ofstream csv(out + "/interface_velocity.csv");
  csv.precision(17);
  csv << "time,v\n0,0\n";
  ex.execute((int)argi("num_steps", 10), [&](int) { csv << problem.time() << ',' << v.getValue() << "\n"; });
  return 0;
}

// test/tests/tensor_compute/smooth_rectangle.i: SmoothRectangleCompute with the sharp, COS and TANH profiles
static int run_smooth_rectangle(DomainAction & domain, const std::string & out)
{
  TensorProblem problem(domain);
  SmoothRectangleCompute::Params p;
  p.x1 = argd("x1", 5);
  p.x2 = argd("x2", 15);
  p.y1 = argd("y1", 5);
  p.y2 = argd("y2", 15);
  p.z1 = argd("z1", 0);
  p.z2 = argd("z2", 0);
  p.inside = argd("inside", -1);
  p.outside = argd("outside", 3);
  const struct { const char * buffer, * profile; double w; } cases[] = {
      {"rectangle_sharp", "COS", 0.0}, {"rectangle_cos", "COS", argd("int_width", 1.0)}, {"rectangle_tanh", "TANH", argd("int_width", 1.0)}};
  for (const auto & c : cases)
  {
    p.buffer = c.buffer;
    p.profile = c.profile;
    p.int_width = c.w;
    SmoothRectangleCompute(problem, c.buffer, p).computeBuffer();
    dump(out, c.buffer, 0, problem.getBuffer(c.buffer));
  }
  return 0;
}

// test/tests/solvers/etdrk4_diffusion.i: 1-D diffusion with ETDRK4 and a zero nonlinear term
static int run_etdrk4_diffusion(DomainAction & domain, const std::string & out)
{
  TensorProblem problem(domain);
  const double D = argd("D", 0.05), k = argd("k", 1.0);
  ParsedCompute::Params p0;
  p0.buffer = "u0";
  p0.expression = "sin(kk*x)";
  p0.constants = {{"kk", k}};
  p0.extra_symbols = true;
  ParsedCompute(problem, "u0", p0).computeBuffer();
  ParsedCompute::Params pu;
  pu.buffer = "u";
  pu.expression = "u0";
  pu.inputs = {"u0"};
  ParsedCompute(problem, "u", pu).computeBuffer();
  ReciprocalLaplacianFactor(problem, "L", "L", D).computeBuffer();
  problem.getBuffer("zero") = DeviceTensor::zeros(2 * domain.getReciprocalSize());  // ConstantReciprocalTensor
  auto root = std::make_shared<ComputeGroup>(problem, "root");
  root->add(std::make_shared<ForwardFFT>(problem, "u_bar", "u_bar", "u"));
  ParsedCompute::Params pe;
  pe.buffer = "u_exact";
  pe.expression = "u0*exp(-DD*kk^2*t)";
  pe.inputs = {"u0"};
  pe.constants = {{"DD", D}, {"kk", k}};
  pe.extra_symbols = true;
  root->add(std::make_shared<ParsedCompute>(problem, "u_exact", pe));
  ParsedCompute::Params pd;
  pd.buffer = "u_diff_sq";
  pd.expression = "(u - u_exact)^2";
  pd.inputs = {"u", "u_exact"};
  root->add(std::make_shared<ParsedCompute>(problem, "u_diff_sq", pd));
  ETDRK4Solver solver(problem, "solver", (unsigned int)argi("ss", 1), root, {{"u", "u_bar", "L", "zero"}});
  Transient ex(problem, solver, argd("dt", 10.0));
  std::ofstream csv(out + "/etdrk4.csv");
  csv.precision(17);
  csv << "time,mse,rmse\n0,0,0\n";
  ex.execute((int)argi("num_steps", 10), [&](int) {
    const double mse = TensorPostprocessors::integral(domain, problem.getBuffer("u_diff_sq"), domain.getExtent(0));
    csv << problem.time() << ',' << mse << ',' << std::sqrt(mse) << "\n";
  });
  return 0;
}

// test/tests/gradient/gradient.i: spectral gradient of sin(x)+sin(y)+sin(z) on an anisotropic box vs the analytic one
static int run_gradient(DomainAction & domain, const std::string & out)
{
  TensorProblem problem(domain);
  auto parsed = [&](const std::string & buffer, const std::string & expr, std::vector<std::string> inputs, bool extra) {
    ParsedCompute::Params p;
    p.buffer = buffer;
    p.expression = expr;
    p.inputs = std::move(inputs);
    p.extra_symbols = extra;
    ParsedCompute(problem, buffer, p).computeBuffer();
  };
  parsed("s", "sin(x)+sin(y)+sin(z)", {}, true);
  if (arg("problem") == "gradient_square")  // test/tests/gradient/gradient_square.i
  {
    parsed("c2", "cos(x)^2+cos(y)^2+cos(z)^2", {}, true);
    FFTGradientSquare(problem, "grad_sq", "grad_sq", "s").computeBuffer();
    parsed("diff", "abs(grad_sq - c2)", {"grad_sq", "c2"}, false);
    double vol = 1.0;
    for (int d = 0; d < domain.getDim(); ++d)
      vol *= domain.getExtent(d);
    const double diff2 = TensorPostprocessors::integral(domain, problem.getBuffer("diff"), vol);   // (global on slab contexts)
    if (domain.rank() == 0)
    {
      std::ofstream csv2(out + "/gradient_square.csv");
      csv2.precision(17);
      csv2 << "time,diff\n0,0\n1," << diff2 << "\n";
    }
    return 0;
  }
  parsed("cx", "cos(x)", {}, true);
  parsed("cy", "cos(y)", {}, true);
  parsed("cz", "cos(z)", {}, true);
  FFTGradient(problem, "gradx_sin", "gradx_s", "s", 0).computeBuffer();
  FFTGradient(problem, "grady_sin", "grady_s", "s", 1).computeBuffer();
  FFTGradient(problem, "gradz_sin", "gradz_s", "s", 2).computeBuffer();
  parsed("diff", "abs(gradx_s - cx)+abs(grady_s - cy)+abs(gradz_s - cz)", {"gradx_s", "grady_s", "gradz_s", "cx", "cy", "cz"}, false);
  double volume = 1.0;
  for (int d = 0; d < domain.getDim(); ++d)
    volume *= domain.getExtent(d);
  const double diff = TensorPostprocessors::integral(domain, problem.getBuffer("diff"), volume);      // (global on slab contexts)
  if (domain.rank() == 0)
  {
    std::ofstream csv(out + "/gradient.csv");
    csv.precision(17);
    csv << "time,diff\n0,0\n1," << diff << "\n";
  }
  return 0;
}

// parallel_mode=FFT_SLAB without rank=: start one child per rank (fork + exec of this binary; the parent has made no HIP call) and
// return the worst exit code.  The job name carries the parent's pid, so concurrent runs do not share a bootstrap segment.
static int launch_ranks(int argc, char ** argv, int nranks)
{
  if (nranks < 1 || nranks > 64)
    paramError("nranks", "1 <= nranks <= 64");
  // a profiler preload (rocprofv3) initialises the GPU before main(): an exec from such a process takes the machine down on this
  // pool.  Profile one `rank=r job=<name>` process directly instead (the other ranks started unprofiled with the same job name).
  for (const char * var : {"LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_REGISTER_FORCE_LOAD"})
  {
    const char * v = std::getenv(var);
    if (v && (std::strstr(v, "rocprof") || std::strstr(v, "roctracer")))
      mooseError(std::string("refusing to fork + exec rank processes under a profiler preload (") + var +
                 " is set): profile one rank=r job=<name> process directly");
  }
  const std::string job = "job=mrlrun_" + std::to_string((long)getpid());
  std::vector<pid_t> kids;
  for (int r = 0; r < nranks; ++r)
  {
    const pid_t pid = fork();
    if (pid < 0)
      mooseError("fork failed");
    if (pid == 0)
    {
      const std::string rk = "rank=" + std::to_string(r);
      std::vector<char *> av(argv, argv + argc);
      av.push_back(const_cast<char *>(rk.c_str()));
      av.push_back(const_cast<char *>(job.c_str()));
      av.push_back(nullptr);
      execv("/proc/self/exe", av.data());
      std::perror("execv");
      _exit(127);
    }
    kids.push_back(pid);
  }
  int rc = 0;
  for (const pid_t k : kids)
  {
    int st = 0;
    if (waitpid(k, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0)
      rc = 1;
  }
  return rc;
}

int main(int argc, char ** argv)
{
  for (int i = 1; i < argc; ++i)
  {
    const std::string a = argv[i];
    const auto eq = a.find('=');
    if (eq == std::string::npos)
    {
      std::cerr << "expected key=value, got '" << a << "'\n";
      return 2;
    }
    g_args[a.substr(0, eq)] = a.substr(eq + 1);
  }
  try
  {
    const int dim = (int)argi("dim", 0);
    if (dim < 1 || dim > 3)
      mooseError("Unsupported mesh dimension");
    const char * nn[3] = {"nx", "ny", "nz"};
    const char * mx[3] = {"xmax", "ymax", "zmax"};
    const char * mn[3] = {"xmin", "ymin", "zmin"};
    std::vector<int64_t> n;
    std::vector<double> lo, hi;
    for (int d = 0; d < dim; ++d)
    {
      n.push_back(argi(nn[d], 1));
      lo.push_back(argd(mn[d], 0.0));
      const std::string m = arg(mx[d]);  // "2pi", "4pi", "6pi" or a number
      hi.push_back(m.size() > 2 && m.substr(m.size() - 2) == "pi" ? std::atof(m.c_str()) * M_PI : argd(mx[d], 1.0));
    }
    DomainAction::Parallel par;
    par.dense_spectra = arg("dense_spectra", "false") == "true";
    if (arg("parallel_mode", "NONE") == "FFT_SLAB")
    {
      par.mode = DomainAction::ParallelMode::FFT_SLAB;
      par.nranks = (int)argi("nranks", 1);
      if (!g_args.count("rank"))
        return launch_ranks(argc, argv, par.nranks);  // nothing has touched the GPU yet
      par.rank = (int)argi("rank", 0);
      par.job = arg("job", "marlin_hip_run");
      par.transport = (int)argi("transport", MRL_TRANSPORT_AUTO);
      int ndev = 0;
      if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        mooseError("no HIP device available");
      par.device = g_args.count("device") ? (int)argi("device", 0) : par.rank % ndev;
      g_rank_suffix = ".rank" + std::to_string(par.rank);
    }
    else if (arg("parallel_mode", "NONE") == "FFT_PENCIL")
    {
      // DomainAction::partitionPencils (DomainAction.C:568-742): nranks = py * pz rank processes, here on the GPUs of one node
      par.mode = DomainAction::ParallelMode::FFT_PENCIL;
      par.nranks = (int)argi("nranks", 4);
      if (!g_args.count("rank"))
        return launch_ranks(argc, argv, par.nranks);  // nothing has touched the GPU yet
      par.rank = (int)argi("rank", 0);
      par.job = arg("job", "marlin_hip_run");
      par.transport = (int)argi("transport", MRL_TRANSPORT_AUTO);
      int ndev = 0;
      if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        mooseError("no HIP device available");
      par.device = g_args.count("device") ? (int)argi("device", 0) : par.rank % ndev;
      g_rank_suffix = ".rank" + std::to_string(par.rank);
    }
    else if (arg("parallel_mode", "NONE") != "NONE")
      paramError("parallel_mode", "NONE, FFT_SLAB or FFT_PENCIL");
    DomainAction domain(dim, n, hi, lo, par);
    if (domain.isSlab() && g_args.count("nsub"))
      domain.check(mrl_ctx_set_option(domain.ctx(), MRL_OPT_SLAB_NSUB, argi("nsub", 1)));
    const std::string out = arg("out", ".");
    const std::string problem = arg("problem");
    if (problem == "cahnhilliard")
      return run_cahnhilliard(domain, out);
    if (problem == "mechanics")
      return run_mechanics(domain, out);
    if (problem == "brusselator")
      return run_brusselator(domain, out);
    if (problem == "coupled" || problem == "nl_coupled")
      return run_coupled(domain, out, problem == "nl_coupled");
    if (problem == "rotating_grain_secant")
      return run_rotating_grain_secant(domain, out);
    if (problem == "postprocessors")
      return run_postprocessors(domain, out);
    if (problem == "cahnhilliard_explicit")
      return run_cahnhilliard_explicit(domain, out);
    if (problem == "coupled_pf_mech")
      return run_coupled_pf_mech(domain, out);
    if (problem == "kks")
      return run_kks(domain, out);
    if (problem == "interface_velocity")
      return run_interface_velocity(domain, out);
    if (problem == "smooth_rectangle")
      return run_smooth_rectangle(domain, out);
    if (problem == "etdrk4_diffusion")
      return run_etdrk4_diffusion(domain, out);
    if (problem == "gradient" || problem == "gradient_square")
      return run_gradient(domain, out);
    mooseError("unknown problem '" + problem + "'");
  }
  catch (const std::exception & e)
  {
    std::cerr << "*** ERROR ***\n" << e.what() << "\n";
    return 1;
  }
}
