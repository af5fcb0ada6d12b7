// Fused Cahn-Hilliard fast path (power-of-two grids). Placeholder until the fast kernels land.
#include "mrl_internal.h"
namespace mrl {
struct ChP;
int ch_substep_fused(mrl_ctx *, const ChP &, const double *, double *, double *, const double *const *, int, double,
                     double *, double *) {
  return MRL_ERR_UNSUPPORTED;
}
}  // namespace mrl
