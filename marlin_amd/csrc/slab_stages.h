// Rank-local stages of the slab pipelines, shared by the staged C-ABI entry points (slab.hip, slab_mech_fused.hip) and by the
// library-owned multi-GPU drivers (slab_driver.hip).
#pragma once
#include "comm_dev.h"
#include "mrl_internal.h"

namespace mrl {

int slab_fast_ok(const mrl_ctx *ctx);      // the fused slab pipeline applies (shift- or table-addressed chunks)
int slab_fast_shift(const mrl_ctx *ctx);   // ... with equal power-of-two partitions: chunk addressing by shifts (the tuned kernels)
long long slab_xplane_of(const mrl_ctx *ctx, long long rows, long long kp);   // x-plane pitch of a chunk with `rows` y rows
int slab_mech_fast_ok(const mrl_ctx *ctx);  // ... and the exchange buffers of the mechanics row pipelines fit 32-bit byte offsets
int slab_mech_table_ok(const mrl_ctx *ctx); // the mechanics pipelines with table-addressed chunks (partitions that are not equal powers of two)
inline int slab_mech_any_ok(const mrl_ctx *ctx) { return slab_mech_fast_ok(ctx) || slab_mech_table_ok(ctx); }
// the slab Newton-CG solve runs on field-major vectors through slab_gamma_fm (the same verdict on every rank)
inline int slab_mech_soa(const mrl_ctx *ctx) { return slab_mech_table_ok(ctx) || (slab_mech_fast_ok(ctx) && ctx->nloc[1] % 2 == 0); }
int slab_gamma_tangent_fusable(const mrl_ctx *ctx);  // fused CG direction + tangent + forward z pass available on this slab context
int slab_tabs_get(mrl_ctx *ctx, long long kp, int nf, const SlabTabs **out);  // device tables of the table-addressed kernels (nf: SlabTabs)
int slab_sub_range(mrl_ctx *ctx, int sub, int nsub, long long *k0, long long *ksub);
long long slab_xplane(const mrl_ctx *ctx, long long kp);    // elements between two x planes of a chunk of those layouts (padded: odd number of 256-byte pieces)
long long slab_kpitch(const mrl_ctx *ctx, long long ksub);  // row pitch of the Cahn-Hilliard exchange layouts (slab_fused.hip)

// Cahn-Hilliard, planned shapes (slab_fused.hip).  otab / utab: destination of chunk p (one entry per rank) in the forward /
// inverse exchange layout; sig: arrival flags raised by the last workgroup of the launch (or none).
int slab_ch_z_fwd_fast(mrl_ctx *ctx, const ChP &cp, const double *c_in, double *mu, int carry);
int slab_ch_x_fwd_fast(mrl_ctx *ctx, int k0, int ksub, cplx *const *otab, const SignalArgs &sig, int carry);
int slab_ch_kspace_fast(mrl_ctx *ctx, const ChP &cp, int k0, int ksub, const double *recv, cplx *const *utab, const SignalArgs &sig,
                        double *Nhat_new, const double *const *Nhat_old, int order, double sub_dt, double *cbar, int carry);
int slab_ch_x_inv_fast(mrl_ctx *ctx, int k0, int ksub, const double *recv);
int slab_ch_z_inv_fast(mrl_ctx *ctx, double *real_out);
int slab_ch_z_inv_fwd_fast(mrl_ctx *ctx, const ChP &cp, double *mu, int carry);

// Cahn-Hilliard, any shape (slab.hip): contiguous send / receive buffers
int gen_z_fwd(mrl_ctx *ctx, const ChP &cp, const double *c_in, double *d_mu, int carry);
int gen_x_fwd(mrl_ctx *ctx, long long k0, long long ksub, double *send, int carry);
int gen_kspace(mrl_ctx *ctx, const ChP &cp, long long k0, long long ksub, const double *recv, double *send, double *Nhat_new,
               const double *const *Nhat_old, int order, double sub_dt, double *d_cbar, int carry);
int gen_x_inv(mrl_ctx *ctx, long long k0, long long ksub, const double *recv);
int gen_z_inv(mrl_ctx *ctx, double *real_out);

// plain slab transforms split at the exchange (slab.hip)
int slab_fwd_local(mrl_ctx *ctx, const double *real_in, double *send);
int slab_fwd_finish(mrl_ctx *ctx, const double *recv, double *spec_out);
int slab_inv_local(mrl_ctx *ctx, const double *spec_in, double *send);
int slab_inv_finish(mrl_ctx *ctx, const double *recv, double *real_out);

// Gamma operator rows on field-major data (slab_mech_fused.hip)
int slab_gamma_row_fwd(mrl_ctx *ctx, int row, const double *A_fm, cplx *const *otab, const SignalArgs &sig);
int slab_gamma_row_mid(mrl_ctx *ctx, const double *recv, cplx *const *otab, const SignalArgs &sig, double scale);
int slab_gamma_row_inv(mrl_ctx *ctx, int row, const double *recv, double *out_fm, const double *dotv_fm);
// all three rows per launch, nine-field exchange layout [p][9][x][y][nzc] (slab_mech_fused.hip)
bool slab_gamma_batched_ok(const mrl_ctx *ctx);
int slab_gamma_rows_fwd(mrl_ctx *ctx, const double *A_fm, cplx *const *otab, const SignalArgs &sig);
int slab_gamma_rows_mid(mrl_ctx *ctx, const double *recv, cplx *const *otab, const SignalArgs &sig, double scale);
int slab_gamma_rows_inv(mrl_ctx *ctx, const double *recv, double *out_fm, const double *dotv_fm);


// library-owned exchanges (slab_driver.hip; need an attached communicator)
int slab_fft_forward(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch);
int slab_fft_inverse(mrl_ctx *ctx, const double *d_in, double *d_out, long long batch);
int slab_gamma_fm(mrl_ctx *ctx, const double *A_fm, double *out_fm, double scale, const double *dotv_fm, double *d_dot);
int slab_gamma_vm(mrl_ctx *ctx, const double *A_vm, double *out_vm, double scale);
int slab_gamma_tangent_z(mrl_ctx *ctx, const double *F, const double *K, const double *mu, double *p, const double *r, const double *S,
                         int i_num, int i_den, double *x, int i_arz, int i_apAp);
// in-place sum over ranks of n device scalars (stream-ordered, no host round trip unless h_out)
int slab_allreduce_scalars(mrl_ctx *ctx, const double *d_local, int n, double *d_global, double *h_out);

}  // namespace mrl
