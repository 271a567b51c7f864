// Shared device helpers for libi2p_ops.so (gfx950 only; wave64 hard-coded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/i2p_ops.h"

#define I2P_WAVE 64

// Launch-error convention of the C ABI: never exit(), return the hipError_t.
#define I2P_RETURN_LAUNCH_STATUS()                      \
    do {                                                \
        hipError_t e__ = hipGetLastError();             \
        return e__ == hipSuccess ? 0 : (int)e__;        \
    } while (0)

// The one squared-norm evaluation order used everywhere (see include/i2p_ops.h).
__device__ __forceinline__ float i2p_sq3(float a, float b, float c) {
    return __fmaf_rn(c, c, __fmaf_rn(b, b, __fmul_rn(a, a)));
}

__device__ __forceinline__ unsigned i2p_f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float i2p_u2f(unsigned u) { return __uint_as_float(u); }

// DPP all-reduce inside a 16-lane row (every lane of the row ends with the row result).
// quad_perm[1,0,3,2] = 0xB1, quad_perm[2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140.
template <int CTRL>
__device__ __forceinline__ unsigned i2p_dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ unsigned i2p_row16_min_u32(unsigned v) {
    v = min(v, i2p_dpp_u32<0xB1>(v));
    v = min(v, i2p_dpp_u32<0x4E>(v));
    v = min(v, i2p_dpp_u32<0x141>(v));
    v = min(v, i2p_dpp_u32<0x140>(v));
    return v;
}
__device__ __forceinline__ unsigned i2p_row16_add_u32(unsigned v) {
    v += i2p_dpp_u32<0xB1>(v);
    v += i2p_dpp_u32<0x4E>(v);
    v += i2p_dpp_u32<0x141>(v);
    v += i2p_dpp_u32<0x140>(v);
    return v;
}

// float sum over a 16-lane row (all lanes end with the row sum), returned as bits
__device__ __forceinline__ unsigned i2p_row16_add_f32(float f) {
    unsigned v = __float_as_uint(f);
    v = __float_as_uint(__uint_as_float(v) + __uint_as_float(i2p_dpp_u32<0xB1>(v)));
    v = __float_as_uint(__uint_as_float(v) + __uint_as_float(i2p_dpp_u32<0x4E>(v)));
    v = __float_as_uint(__uint_as_float(v) + __uint_as_float(i2p_dpp_u32<0x141>(v)));
    v = __float_as_uint(__uint_as_float(v) + __uint_as_float(i2p_dpp_u32<0x140>(v)));
    return v;
}

// Blocks of one batch sample share that sample's range image; the dispatcher places block b
// on XCD b%8 (observed, speed only), so give each XCD a contiguous run of logical blocks and
// its L2 sees one sample's image instead of all of them.
__device__ __forceinline__ unsigned i2p_xcd_swizzle(unsigned bid, unsigned nblocks) {
    if ((nblocks & 7u) != 0u) return bid;
    const unsigned per = nblocks >> 3;
    return (bid & 7u) * per + (bid >> 3);
}

// "The last block to finish finalises" ticket of the layer kernels (mlp.hip, mlp_wreg.hip).  Called by ONE lane of the block after a
// `s_waitcnt vmcnt(0)` of every thread and the block's __syncthreads(); `total` = blocks of the grid.  Ordering (MI355X_MICROARCH.md,
// inter-workgroup visibility; cdna_hip_programming.md Guideline 16): an agent-scope RELEASE before the ticket (buffer_wbl2 sc1 — what
// the block published for the finaliser are agent-scope atomics, already performed at the memory side, but the ordering no longer
// rests on that), the asm wait restating the post-write-back wait where the compiler cannot drop it, the relaxed ticket, and an
// agent-scope ACQUIRE in the block that drew the last one, before it reads the other blocks' sums.  -DI2P_RELAXED_SYNC builds the
// round-3 form (acknowledged atomics + relaxed ticket) for A/B timing (I2P_BUILD_VARIANT, i2pnet_amd/build.py).
__device__ __forceinline__ bool i2p_ticket_is_last(unsigned *ticket, unsigned total) {
#ifndef I2P_RELAXED_SYNC
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    const bool last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1u;
#ifndef I2P_RELAXED_SYNC
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    return last;
}

// deterministic (owner-scans-in-row-order) scatter-add backward kernels, csrc/scatter_det.hip
int i2p_det_gather_rows_grad(int b, int hw, int c, int q, int W, const float *grad_out, const int64_t *h_idx, const int64_t *w_idx,
                             float *grad_feat, void *stream);
int i2p_det_gather_points_grad(int b, int c, int n, int m, const float *grad_out, const int *idx, float *grad_points, void *stream);
int i2p_det_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx, const float *weight,
                                   float *grad_points, void *stream);
// I2P_ATOMIC_SCATTER=1: the first-generation atomicAdd kernels (A/B timing only; not reproducible run to run)
#include <cstdlib>
inline bool i2p_atomic_scatter() { static const char *e = getenv("I2P_ATOMIC_SCATTER"); return e && e[0] == '1'; }

// weights-stationary-in-registers fp32 layer forward for the wide layers (csrc/mlp_wreg.hip); lin_fwd_impl routes to it
bool i2p_wreg_fwd_ok(long long rows, int cin, int cout);
int i2p_wreg_fwd(long long rows, int cin, int cout, const float *x, int x_ld, const float *in_coef, float slope, const float *w,
                 float *y, int y_ld, double *sums, unsigned *fin_counter, const float *fin_gamma, const float *fin_beta,
                 float fin_eps, float *fin_coef, float *fin_mi, void *stream, const float *xb = nullptr,
                 const float *in_coef_b = nullptr, float slope_b = 1.f);
bool i2p_wreg_dgrad_ok(long long rows, int k, int c);
int i2p_wreg_dgrad(long long rows, int k, int c, const float *gz, const float *y2, const double *g_dsums, const float *g_oc,
                   const float *g_omi, long long g_rows, const float *w, float *gz_in, const float *ex, const float *e_coef,
                   const float *e_mi, float e_slope, double *sums, void *stream, float *gz_in_b = nullptr,
                   const float *exb = nullptr, const float *e_coef_b = nullptr, const float *e_mi_b = nullptr, float e_slope_b = 1.f,
                   const float *e_add = nullptr, double *sums_b = nullptr);
bool i2p_wreg_wgrad_ok(long long rows, int cin, int cout);
int i2p_wreg_wgrad(long long rows, int cin, int cout, const float *gz, const float *y2, const double *g_dsums, const float *g_oc,
                   const float *g_omi, long long g_rows, float *bn_out, const float *x, const float *in_coef, float slope_in,
                   const float *xb, const float *in_coef_b, float slope_b, int split, float *dw_partial, unsigned grid, void *stream);
// input gradient + weight gradient of an HBM-bound wide layer from ONE read of gz / y / x (csrc/mlp_wreg_fused.hip)
// csrc/deferred.hip: record a weight-gradient slab reduction instead of launching it (true = recorded).  kind 0: the 16 x 16 float4 shape
// (n = float4 columns), kind 1: reduce_partials_bf16's 32 x 8 shape (n = floats)
bool i2p_defer_reduce(int kind, int nparts, int n, const void *parts, void *out);
bool i2p_defer_conv_fin(int nblk, int NT, const float *partials, float *dW, int s_out, int s_in, int s_kh, int s_kw);
bool i2p_wreg_pair_bwd_fused_ok(void);
int i2p_wreg_pair_bwd_fused(int B, int N, int M, int KT, int NCH, int NL, const float *gz, const float *y2, const double *g_dsums,
                            const float *g_oc, const float *g_omi, const float *f, const float *g, const float *w, float *dw_partial,
                            float *s_df, float *s_dbn, float *s_dg, float *s_dbk, void *stream);
bool i2p_wreg_bwd_fused2_ok(long long rows, int k, int c, int split);
int i2p_wreg_bwd_fused2(long long rows, const float *gz, const float *y2, const double *g_dsums, const float *g_oc, const float *g_omi,
                        long long g_rows, const float *w, float *gz_in_a, const float *xa, const float *coef_a, const float *mi_a,
                        float slope_a, double *sums_a, float *gz_in_b, const float *xb, const float *coef_b, const float *mi_b,
                        float slope_b, double *sums_b, const float *e_add, float *bn_out, float *dw_partial, unsigned grid, void *stream);
bool i2p_wreg_bwd_fused_ok(long long rows, int k, int c);
int i2p_wreg_bwd_fused(long long rows, int k, int c, const float *gz, const float *y2, const double *g_dsums, const float *g_oc,
                       const float *g_omi, long long g_rows, const float *w, float *gz_in, const float *ex, const float *e_coef,
                       const float *e_mi, float e_slope, double *sums, float *bn_out, float *dw_partial, unsigned grid, void *stream);
bool i2p_wreg_pair_bwd_ok(int B, int N, int M, int cin, int cout);
long long i2p_wreg_pair_bwd_scratch(int B, int N, int M, int cin, int cout);
int i2p_wreg_pair_bwd(int B, int N, int M, int cin, int cout, const float *gz, const float *y2, const double *g_dsums,
                      const float *g_oc, const float *g_omi, const float *f, const float *g, const float *w, float *scratch,
                      int *KT_out, int *NCH_out, void *stream);
int i2p_wreg_pair_fwd(int B, int N, int M, int cin, int cout, const float *f, const float *g, const float *bias_n,
                      const float *bias_k, const float *w, float *y, double *sums, unsigned *fin_counter, const float *fin_gamma,
                      const float *fin_beta, float fin_eps, float *fin_coef, float *fin_mi, void *stream);
bool i2p_small_wgrad_ok(long long rows, int cin, int cout);
int i2p_small_wgrad(long long rows, int cin, int cout, const float *gz, const float *y2, const double *g_dsums, const float *g_oc,
                    const float *g_omi, long long g_rows, float g_slope, float *bn_out, const float *x, const float *in_coef,
                    float slope_in, float *dw_partial, unsigned grid, void *stream);
// bf16-storage wgrad with the accumulators in registers (csrc/mlp_wreg_bf16.hip); the caller reduces dw_partial[grid]
// wide layers on few rows (cin > 160 or cout > 128, <= 320 channels): K-tiled fused layer kernels (csrc/mlp_big.hip)
bool i2p_big_layer_ok(long long rows, int cin, int cout);
int i2p_big_fwd(long long rows, int cin, int cout, const float *x, const float *in_coef, float slope_in, const float *w, float *y,
                double *sums, void *stream);
int i2p_big_bwd(long long rows, int cin, int cout, const float *gz, const float *y, float *g_out, const double *out_dsums, const float *out_coef,
                const float *out_mi, float slope_out, const float *x,
                const float *in_coef, const float *in_mi, float slope_in, const float *w, float *gz_in, double *in_dsums,
                float *dw_partial, int max_chunks, float *dw, void *stream);
bool i2p_small_wgrad_bf16_ok(long long rows, int cin, int cout, int x_bf16);
int i2p_small_wgrad_bf16(long long rows, int cin, int cout, const unsigned short *gz, const unsigned short *y, const float *g_coef,
                         float g_slope, const void *x, int x_bf16, const float *in_coef, float slope_in, float *dw_partial, unsigned grid,
                         void *stream);
bool i2p_wreg_wgrad_bf16_ok(long long rows, int cin, int cout);
int i2p_wreg_wgrad_bf16(long long rows, int cin, int cout, const unsigned short *gz, const unsigned short *y, const float *g_coef,
                        float g_slope, const unsigned short *x, const float *in_coef, float slope_in, float *dw_partial, unsigned grid,
                        void *stream, const unsigned short *xb = nullptr, const float *in_coef_b = nullptr, float slope_b = 1.f);
// input gradient + weight gradient of a bf16-storage layer with 64 output channels from ONE read of gz / y / x (csrc/mlp_bwd_fused_bf16.hip)
bool i2p_bwd_fused_bf16_ok(long long rows, int cin, int cout);
int i2p_bwd_fused_bf16(long long rows, int cin, int cout, const unsigned short *gz, const unsigned short *y, const double *out_dsums,
                       const float *out_coef, const float *out_mi, float *coef8, const unsigned short *x, const float *in_coef, const float *in_mi, float slope_in, const float *w,
                       unsigned short *gz_in, double *in_dsums, float *dw_partial, unsigned grid, void *stream);
// second-generation bf16 pair-layer backward (csrc/pair_bwd_bf16.hip): d_f / d_g / d_bn / d_bk accumulate into the caller's zeroed
// outputs, dw_partial[grid][128*128] is reduced by the caller
bool i2p_pair_bwd2_bf16_ok(int B, int N, int M, int cin, int cout);
int i2p_pair_bwd2_bf16_grid(int B, int N, int M);
int i2p_pair_bwd2_bf16(int B, int N, int M, const unsigned short *gz, const unsigned short *y, const double *out_dsums, const float *out_coef,
                       const float *out_mi, float *coef8, const float *f,
                       const float *g, const float *w, float *d_f, float *d_g, float *d_bn, float *d_bk, float *dw_partial, void *stream);
bool i2p_bwd_fused2_bf16_ok(long long rows, int cin_a, int cin_b, int cout);
int i2p_bwd_fused2_bf16(long long rows, const unsigned short *gz, const unsigned short *y, const double *out_dsums, const float *out_coef,
                        const float *out_mi, float *coef8, const unsigned short *xa,
                        const float *coef_a, const float *mi_a, float slope_a, const unsigned short *xb, const float *coef_b,
                        const float *mi_b, float slope_b, const unsigned short *e_add, const float *w, unsigned short *gz_a,
                        double *sums_a, unsigned short *gz_b, double *sums_b, float *dw_partial, unsigned grid, void *stream);

// Kernel-only timing for bench.py's roofline object: when enabled, the launchers of the headline kernels bracket
// THE KERNEL (not the entry's coefficient / reduction launches) with two HIP events on the launch stream; i2p_ktime_last_us()
// synchronises on the closing event and returns the duration of the last bracketed launch.  Off by default (no events recorded);
// not usable inside a stream capture.  Defined in csrc/optim.hip.
void i2p_ktime_begin(hipStream_t st);
void i2p_ktime_end(hipStream_t st);
// first cost-volume layer forward, bf16 output, a strip shared by the four waves of a block (csrc/pair_fwd_bf16.hip)
bool i2p_pair_fwd3_bf16_ok(int B, int N, int M, int cin, int cout);
int i2p_pair_fwd3_bf16(int B, int N, int M, const float *f, const float *g, const float *bias_n, const float *bias_k, const float *w,
                       unsigned short *y, double *sums, void *stream);
