// Entry points declared in marlin_hip.h whose kernels have not landed yet: fail loudly.
#include "mrl_internal.h"
using namespace mrl;
#define MRL_STUB(ctx, name) return set_error(ctx, MRL_ERR_UNSUPPORTED, name ": not implemented in this build")
extern "C" {
int mrl_slab_counts(const mrl_ctx *ctx, int, int64_t *, int64_t *, int64_t *, int64_t *) { MRL_STUB(ctx, "mrl_slab_counts"); }
int mrl_slab_fwd_local(mrl_ctx *ctx, const double *, double *) { MRL_STUB(ctx, "mrl_slab_fwd_local"); }
int mrl_slab_fwd_finish(mrl_ctx *ctx, const double *, double *) { MRL_STUB(ctx, "mrl_slab_fwd_finish"); }
int mrl_slab_inv_local(mrl_ctx *ctx, const double *, double *) { MRL_STUB(ctx, "mrl_slab_inv_local"); }
int mrl_slab_inv_finish(mrl_ctx *ctx, const double *, double *) { MRL_STUB(ctx, "mrl_slab_inv_finish"); }
int mrl_slab_ch_fwd_local(mrl_ctx *ctx, const mrl_ch_params *, const double *, double *, double *) { MRL_STUB(ctx, "mrl_slab_ch_fwd_local"); }
int mrl_slab_ch_kspace(mrl_ctx *ctx, const mrl_ch_params *, const double *, double *, double *, const double *const *, int, double, double *) { MRL_STUB(ctx, "mrl_slab_ch_kspace"); }
int mrl_gamma_apply(mrl_ctx *ctx, const double *, double *) { MRL_STUB(ctx, "mrl_gamma_apply"); }
int mrl_mech_stress(mrl_ctx *ctx, const double *, const double *, const double *, double *) { MRL_STUB(ctx, "mrl_mech_stress"); }
int mrl_mech_tangent_apply(mrl_ctx *ctx, const double *, const double *, const double *, const double *, double *) { MRL_STUB(ctx, "mrl_mech_tangent_apply"); }
int mrl_mech_newton_cg(mrl_ctx *ctx, const mrl_mech_params *, const double *, const double *, const double *, const double *, double *, double *, mrl_mech_stats *) { MRL_STUB(ctx, "mrl_mech_newton_cg"); }
int mrl_dot(mrl_ctx *ctx, const double *, const double *, int64_t, double *) { MRL_STUB(ctx, "mrl_dot"); }
int mrl_norm2(mrl_ctx *ctx, const double *, int64_t, double *) { MRL_STUB(ctx, "mrl_norm2"); }
int mrl_sum(mrl_ctx *ctx, const double *, int64_t, double *) { MRL_STUB(ctx, "mrl_sum"); }
int mrl_average(mrl_ctx *ctx, const double *, int64_t, double *) { MRL_STUB(ctx, "mrl_average"); }
}
