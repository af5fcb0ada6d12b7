// Placeholders for fast paths that have not landed yet: the callers fall back to the generic HIP stages.
#include "mrl_internal.h"
namespace mrl {
int slab_fast_ok(const mrl_ctx *) { return 0; }
int slab_ch_fwd_local_fast(mrl_ctx *ctx, const ChP &, const double *, double *, double *, int) { return MRL_ERR_UNSUPPORTED; }
int slab_ch_kspace_fast(mrl_ctx *ctx, const ChP &, const double *, double *, double *, const double *const *, int, double, double *) { return MRL_ERR_UNSUPPORTED; }
int slab_inv_finish_fast(mrl_ctx *ctx, const double *, double *) { return MRL_ERR_UNSUPPORTED; }
}
