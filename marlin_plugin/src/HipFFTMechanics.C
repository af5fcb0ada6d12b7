// HipFFTMechanics: the whole FFTMechanics::computeBuffer (Newton iterations around conjugateGradientSolve with the Gamma projection
// G(A) = ifft(Ghat4 : fft(A)) and the St. Venant-Kirchhoff tangent) as one call of mrl_mech_newton_cg.  Neither Ghat4 (1296 B per
// k-point, FFTMechanics.C:74-84) nor the tangent K4 (648 B per point) is ever formed.
#include "HipFFTMechanics.h"
#include "DomainAction.h"

registerMooseObject("MarlinApp", HipFFTMechanics);

InputParameters
HipFFTMechanics::validParams()
{
  InputParameters params = TensorOperator<>::validParams();
  params.addClassDescription("deGeus variational mechanics solve (hyperelastic isotropic) on libmarlin_hip (MI355X).");
  params.addRequiredParam<TensorInputBufferName>("K", "Bulk modulus");
  params.addRequiredParam<TensorInputBufferName>("mu", "Shear modulus");
  params.addParam<Real>("l_tol", 1e-2, "Linear conjugate gradient solve tolerance");
  params.addParam<unsigned int>("l_max_its", "Maximum number of conjugate gradient iterations");
  params.addParam<Real>("nl_rel_tol", 1e-5, "Nonlinear solve relative tolerance");
  params.addParam<Real>("nl_abs_tol", 1e-8, "Nonlinear solve absolute tolerance");
  params.addParam<unsigned int>("nl_max_its", 100, "Maximum number of nonlinear solve iterations");
  params.addParam<TensorOutputBufferName>("stress", "stress", "Computed first Piola-Kirchhoff stress");
  params.addParam<TensorInputBufferName>("applied_macroscopic_strain", "Applied macroscopic strain");
  params.addParam<TensorInputBufferName>("F", "F", "Deformation gradient tensor.");
  params.addParam<bool>("verbose", false, "Print the iteration counts of every solve.");
  return params;
}

HipFFTMechanics::HipFFTMechanics(const InputParameters & parameters)
  : TensorOperator<>(parameters),
    _hip(HipDomain::get(_domain, comm())),
    _tF(getInputBuffer("F")),
    _tK(getInputBuffer("K")),
    _tmu(getInputBuffer("mu")),
    _tP(getOutputBuffer("stress")),
    _applied_macroscopic_strain(isParamValid("applied_macroscopic_strain") ? &getInputBuffer("applied_macroscopic_strain")
                                                                           : nullptr),
    _verbose(getParam<bool>("verbose"))
{
  _prm = mrl_mech_params{};
  _prm.l_tol = getParam<Real>("l_tol");
  _prm.l_max_its = isParamValid("l_max_its") ? getParam<unsigned int>("l_max_its") : 0; // 0 = number of cells, FFTMechanics.C:64-65
  _prm.nl_rel_tol = getParam<Real>("nl_rel_tol");
  _prm.nl_abs_tol = getParam<Real>("nl_abs_tol");
  _prm.nl_max_its = (int)getParam<unsigned int>("nl_max_its");
}

void
HipFFTMechanics::computeBuffer()
{
  // RankTwoIdentity hands out an expanded view (RankTwoIdentity.C:31-32): the ABI wants dense row-major arrays
  const auto F = _tF.contiguous();
  const auto K = _tK.contiguous(), mu = _tmu.contiguous();
  torch::Tensor Fnew = torch::empty_like(F), P = torch::empty_like(F), app;
  if (_applied_macroscopic_strain)
    app = _applied_macroscopic_strain->contiguous();
  mrl_mech_stats st{};
  const int rc = mrl_mech_newton_cg(_hip->ctx(), &_prm, F.data_ptr<double>(), K.data_ptr<double>(), mu.data_ptr<double>(),
                                    _applied_macroscopic_strain ? app.data_ptr<double>() : nullptr, Fnew.data_ptr<double>(),
                                    P.data_ptr<double>(), &st);
  if (rc == MRL_ERR_NOT_CONVERGED)
    paramError("nl_max_its", mrl_last_error(_hip->ctx())); // FFTMechanics.C:159-161
  _hip->check(rc, name());
  if (_verbose)
    _console << name() << ": " << st.newton_its << " Newton iterations, " << st.cg_its_total << " CG iterations, |dF|/|F| = "
             << st.last_rnorm << '\n';
  _u = Fnew; // FFTMechanics.C:112,138
  _tP = P;   // HyperElasticIsotropic.C:50
}
