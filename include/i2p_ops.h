/*
 * i2p_ops.h — C ABI of libi2p_ops.so, the MI355X (gfx950) implementation of I2PNet's
 * native point-cloud operators.
 *
 * Every entry point replaces one raw launcher of the reference's two CUDA extensions
 * (cited per function as file:line under the reference tree).  Conventions shared by all:
 *
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers owned by the
 *     caller (the library never allocates, frees or synchronises);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); launches are
 *     asynchronous on it, re-entrant, no global state, safe for hipGraph capture;
 *   - return value: 0 on success, a positive hipError_t if a launch failed, a negative
 *     I2P_ERR_* for a rejected argument.  The library never calls exit() (the reference's
 *     launchers `fprintf(stderr)+exit(-1)`, e.g. fused_conv_go.cu:259-263);
 *   - caller pre-fills are the reference's: outputs of fused_conv_select_k zeroed
 *     (src/projectPN/utils.py:86-94), FPS `temp` = 1e10 (pointnet2/pointnet2_utils.py:56),
 *     ball_query idx = 0 (:249), every *_grad output = 0 (:98,:177,:221);
 *   - integer/index outputs are bit-exact w.r.t. the CPU oracle (oracle/i2p_oracle.c), which
 *     exports the same signatures minus `stream` under the suffix `_cpu`.
 *
 * Squared distances are evaluated in ONE fixed order everywhere (oracle and device):
 *     d = fmaf(dz, dz, fmaf(dy, dy, dx * dx))
 * which is what nvcc's default FMA contraction makes of the reference's
 * `(a)*(a) + (b)*(b) + (c)*(c)` expressions.
 */
#ifndef I2P_OPS_H_
#define I2P_OPS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define I2P_ERR_BAD_ARG   (-1)  /* negative size, NULL pointer with non-empty work          */
#define I2P_ERR_WINDOW    (-2)  /* kernel_size_H*kernel_size_W > I2P_MAX_WINDOW               */
#define I2P_ERR_K         (-3)  /* K > I2P_MAX_WINDOW                                         */

/* The reference keeps `int idx_w[150], idx_h[150]; float Dist[150]` per thread
 * (fused_conv_go.cu:52-53) and silently overflows beyond that; we reject instead. */
#define I2P_MAX_WINDOW 150

#define I2P_FLAG_COPY  1   /* fused_conv_select_k.py:6 */
#define I2P_FLAG_SHIFT 2   /* fused_conv_select_k.py:7 */
/* extension (not in the reference): slots the reference leaves untouched (dead queries, unselected slots
 * without FLAG_COPY) are WRITTEN as index 0 / mask 0, so the caller need not zero-fill the outputs first
 * (the reference relies on torch.zeros outputs, utils.py:77-82). */
#define I2P_FLAG_FILL  4

/* ABI version / build info (not in the reference). */
int i2p_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * Projection-aware neighbour selection.
 * Replaces FusedConvSelectKLauncher (src/projectPN/fused_conv_select/fused_conv_gpu.h:35-58,
 * fused_conv_go.cu:243-264; kernel :11-240).
 *   xyz1 f32 [B,H,W,3], xyz2 f32 [B,small_h,small_w,3], idx_n2 i32 [B,npoints,2],
 *   random_hw i32 [kH*kW]; outputs i64 [B,npoints,K] x3, f32 [B,npoints,K];
 *   valid_idx / valid_in_dis_idx f32 [B,npoints,kH*kW] are accepted and never written
 *   (as in the reference).
 * --------------------------------------------------------------------------------------------- */
int i2p_fused_conv_select_k(int batch_size, int H, int W, int npoints, int kernel_size_H,
                            int kernel_size_W, int K, int flag, float distance, int stride_h,
                            int stride_w, const float *xyz1, const float *xyz2,
                            const int *idx_n2, const int *random_hw, int64_t *selected_b_idx,
                            int64_t *selected_h_idx, int64_t *selected_w_idx, float *valid_idx,
                            float *valid_in_dis_idx, float *selected_mask, int small_h,
                            int small_w, void *stream);

/* ---------------------------------------------------------------------------------------------
 * pointnet2 primitives (pointnet2/src/pointnet2_api.cpp:10-24).
 * --------------------------------------------------------------------------------------------- */

/* furthest_point_sampling_kernel_launcher (pointnet2/src/sampling_gpu.h:26-27, sampling_gpu.cu:211-253).
 * dataset f32 [B,N,3], temp f32 [B,N] (=1e10 on entry, holds min-distances on exit), idxs i32 [B,M]. */
int i2p_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp,
                                int *idxs, void *stream);

/* gather_points_kernel_launcher_fast (sampling_gpu.h:12-13, sampling_gpu.cu:26-44).
 * points f32 [B,C,N], idx i32 [B,M] -> out f32 [B,C,M]. */
int i2p_gather_points(int b, int c, int n, int npoints, const float *points, const int *idx,
                      float *out, void *stream);

/* gather_points_grad_kernel_launcher_fast (sampling_gpu.h:19-20, sampling_gpu.cu:65-83).
 * grad_out f32 [B,C,M], idx i32 [B,M] -> grad_points f32 [B,C,N] += (zeroed by caller). */
int i2p_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                           const int *idx, float *grad_points, void *stream);

/* ball_query_kernel_launcher_fast (ball_query_gpu.h:12-13, ball_query_gpu.cu:48-67).
 * new_xyz f32 [B,M,3], xyz f32 [B,N,3] -> idx i32 [B,M,nsample] (zeroed by caller). */
int i2p_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                   const float *xyz, int *idx, void *stream);

/* group_points_kernel_launcher_fast (group_points_gpu.h:13-14, group_points_gpu.cu:69-87).
 * points f32 [B,C,N], idx i32 [B,npoints,nsample] -> out f32 [B,C,npoints,nsample]. */
int i2p_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                     const int *idx, float *out, void *stream);

/* group_points_grad_kernel_launcher_fast (group_points_gpu.h:19-20, group_points_gpu.cu:27-44).
 * grad_out f32 [B,C,npoints,nsample] -> grad_points f32 [B,C,N] += (zeroed by caller). */
int i2p_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                          const int *idx, float *grad_points, void *stream);

/* three_nn_kernel_launcher_fast (interpolate_gpu.h:13-14, interpolate_gpu.cu:55-74).
 * unknown f32 [B,N,3], known f32 [B,M,3] -> dist2 f32 [B,N,3] (SQUARED), idx i32 [B,N,3]. */
int i2p_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                 int *idx, void *stream);

/* three_interpolate_kernel_launcher_fast (interpolate_gpu.h:20-21, interpolate_gpu.cu:99-117).
 * points f32 [B,C,M], idx i32 [B,N,3], weight f32 [B,N,3] -> out f32 [B,C,N]. */
int i2p_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                          const float *weight, float *out, void *stream);

/* three_interpolate_grad_kernel_launcher_fast (interpolate_gpu.h:27-28, interpolate_gpu.cu:144-160).
 * grad_out f32 [B,C,N] -> grad_points f32 [B,C,M] += (zeroed by caller). */
int i2p_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                               const int *idx, const float *weight, float *grad_points,
                               void *stream);

/* ---------------------------------------------------------------------------------------------
 * Operators the reference implements in eager PyTorch around the two extensions.  They sit on
 * the same hot path and are exported here so the Python layer
 * stays a thin shim.
 * --------------------------------------------------------------------------------------------- */

/* Spherical projection, src/projectPN/utils.py:111-187 (project_seq with use_rank=False).
 *   xyz f32 [B,N,3] (the cloud that defines the cells and fills image 0),
 *   feats: nfeat device pointers f32 [B,N,feat_dims[i]] (nfeat <= 4),
 *   out_xyz f32 [B,H,W,3] and out_feats[i] f32 [B,H,W,feat_dims[i]]: fully written (zero where empty),
 *   cell_winner i32 [B,H*W] scratch, fully rewritten (-1 = empty cell, else the point index
 *   that owns the cell).  Duplicate-cell rule: highest point index wins (= CPU index_put_
 *   last-writer-wins), the same winner for every image.
 *   Rows with non-finite angles (zero points) land on (row 0, col 900*W/1800) like torch-CPU's
 *   NaN.long() -> INT64_MIN -> clamp path.
 */
int i2p_project_seq(int b, int n, int H, int W, float fup_deg, float fdown_deg,
                    const float *xyz, int nfeat, const float *const *feats,
                    const int *feat_dims, float *out_xyz, float *const *out_feats,
                    int *cell_winner, void *stream);

/* Row gather on channel-last images, src/projectPN/utils.py:36-60 (gather_torch):
 *   out[b, q, :] = feat[b, h_idx[b,q]*W + w_idx[b,q], :]    feat f32 [B,HW,C], idx i64 [B,Q]. */
int i2p_gather_rows(int b, int hw, int c, int q, int W, const float *feat,
                    const int64_t *h_idx, const int64_t *w_idx, float *out, void *stream);

/* Backward of i2p_gather_rows: grad_feat[b, h*W+w, :] += grad_out[b, q, :] (zeroed by caller). */
int i2p_gather_rows_grad(int b, int hw, int c, int q, int W, const float *grad_out,
                         const int64_t *h_idx, const int64_t *w_idx, float *grad_feat,
                         void *stream);

/* Brute-force kNN, src/projectPN/utils.py:343-380 (square_distance + topk(largest=False)).
 *   xyz f32 [B,N,3] (searched), new_xyz f32 [B,S,3] (queries) -> idx i32 [B,S,k], ascending by
 *   (distance, index).  The distance is the reference's expanded form
 *   d = -2*(q.p) + |q|^2 + |p|^2 evaluated as in i2p_oracle.c; torch.topk(sorted=False) leaves
 *   the order unspecified, so parity is on neighbour SETS. */
int i2p_knn(int b, int n, int s, int k, const float *xyz, const float *new_xyz, int *idx,
            void *stream);

/* ---------------------------------------------------------------------------------------------
 * Batch-statistics BatchNorm + (Leaky)ReLU on channel-last activations y [rows, c].
 * This is what every point-branch `Conv2d` of the reference runs after its 1x1 conv
 * (src/projectPN/PPBackbone_center.py:28-46: nn.BatchNorm2d(track_running_stats=False) on the
 * [B,C,K,N] view, i.e. biased batch variance over rows, eps 1e-5, affine; then ReLU or
 * LeakyReLU(0.1)), done in eager PyTorch there (permute / conv / BN / act / permute).
 *   out = act((y - mean) * (invstd * gamma) + beta),  act(z) = z > 0 ? z : slope * z
 *   slope: 0.1 LeakyReLU, 0 ReLU, 1 no activation.
 * Statistics are accumulated in fp64 (sum, sum of squares): robust when |mean| >> std.
 * --------------------------------------------------------------------------------------------- */

/* Partial sums are spread over I2P_BN_REPLICAS copies (same-address fp64 atomics serialise):
 * every `sums` / `dsums` buffer is f64 [I2P_BN_REPLICAS][2*c], zeroed by the caller; the true
 * sums are the sums over the replica axis. */
#define I2P_BN_REPLICAS 32

/* sums[rep][0:c] += sum_r y[r,:],  sums[rep][c:2c] += sum_r y[r,:]^2 */
int i2p_bn_stats(long long rows, int c, const float *y, double *sums, void *stream);

/* forward: writes out [rows,c] and mean_invstd f32 [2*c] (saved for backward) */
int i2p_bn_act_fwd(long long rows, int c, const float *y, const double *sums, const float *gamma,
                   const float *beta, float eps, float slope, float *out, float *mean_invstd,
                   void *stream);

/* backward pass 1: dsums (replicated, zeroed by caller) += { sum dz, sum dz*xhat },
 * dz = dout * act'(z) recomputed from y */
int i2p_bn_act_bwd_stats(long long rows, int c, const float *dout, const float *y,
                         const float *mean_invstd, const float *gamma, const float *beta,
                         float slope, double *dsums, void *stream);

/* backward pass 2: dy = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)); dgamma = sum dz*xhat,
 * dbeta = sum dz (f32 [c] each, written) */
int i2p_bn_act_bwd(long long rows, int c, const float *dout, const float *y,
                   const float *mean_invstd, const float *gamma, const float *beta, float slope,
                   const double *dsums, float *dy, float *dgamma, float *dbeta, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Fused per-point linear layer (1x1 conv) — csrc/mlp.hip.  One reference `Conv2d` block is
 * conv1x1 -> BN(batch stats) -> act (PPBackbone_center.py:34-46); a chain of them is executed
 * as one kernel per layer that applies the PREVIOUS layer's BN + activation on load and
 * accumulates its OWN output statistics:
 *     y[r,:] = act_in((x[r,:] - mean_in) * scale_in + beta_in) . W^T ;  sums += {sum y, sum y^2}
 *   x f32 [rows,cin]; in_coef f32 [3][cin] = {mean, invstd*gamma, beta} or NULL (x used as is);
 *   slope_in: activation in front (1 = none); w f32 [cout,cin] (conv weight, bias dropped: it
 *   cancels in the following batch-stat BN); y f32 [rows,cout]; sums replicated f64 (see
 *   I2P_BN_REPLICAS) or NULL.  cin <= 160 and cout <= 128: weights resident in LDS / registers (cout <= 256 in
 *   slices); cin > 160 or cout > 128 (multiples of 4, <= 320): K-tiled kernels (csrc/mlp_big.hip), same semantics.
 * --------------------------------------------------------------------------------------------- */
int i2p_lin_fwd(long long rows, int cin, int cout, const float *x, const float *in_coef,
                float slope_in, const float *w, float *y, double *sums, void *stream);

/* coef f32 [3][c] = {mean, rsqrt(var+eps)*gamma, beta} (and mean_invstd f32 [2][c] if not NULL)
 * from replicated sums of a [rows,c] tensor. */
int i2p_bn_finalize(long long rows, int c, const double *sums, const float *gamma, const float *beta,
                    float eps, float *coef, float *mean_invstd, void *stream);

/* Backward of one fused layer (csrc/mlp.hip).  With z = BN(y), a = act(z) of THIS layer:
 *   g^y   = scale_out*(gz - mean(gz) - xhat_out*mean(gz*xhat_out))   (or gz itself if out_coef NULL)
 *   dw    = g^y^T . act_in(bn_in(x))                                   f32 [cout,cin], written
 *   gz_in = (g^y . w) * act_in'(z_in)                                  f32 [rows,cin], written (skipped if NULL)
 *   in_dsums += { sum gz_in, sum gz_in*xhat_in }                       replicated f64, zeroed by caller
 * gz f32 [rows,cout]; y f32 [rows,cout]; out_coef [3][cout], out_mi [2][cout] (mean, invstd),
 * out_dsums replicated {sum gz, sum gz*xhat_out}; x f32 [rows,cin] the previous pre-BN tensor with
 * in_coef [3][cin], in_mi [2][cin] (or both NULL: x is the raw layer input, gz_in = dL/dx);
 * dw_partial f32 scratch of i2p_lin_bwd_grid(rows)*cout*cin + 8*cout floats; when out_coef is given its LAST 2*cout
 * floats hold, on return, {sum gz [cout], sum gz*xhat [cout]} reduced over the replicas = dbeta, dgamma of the BN
 * behind this layer.  cin, cout multiples of 4, <= 160 / 128 (weight-resident kernels) or <= 320 (K-tiled, csrc/mlp_big.hip).
 * slope_out: 1 when gz is dL/dz (the usual case: the previous call's gz_in).  For the LAST layer of a stack the
 * caller holds dL/da (a = act(z) with this slope, out_coef required): the activation derivative is applied on
 * load, with out_dsums = {sum gz, sum gz*xhat} of the resulting gz (i2p_bn_act_bwd_stats) — the stack's output
 * gradient is then read twice instead of being rewritten as dL/dy first. */
int i2p_lin_bwd_grid(long long rows);
int i2p_lin_bwd(long long rows, int cin, int cout, const float *gz, const float *y,
                const float *out_coef, const float *out_mi, const double *out_dsums, const float *x,
                const float *in_coef, const float *in_mi, float slope_in, const float *w, float *gz_in,
                double *in_dsums, float *dw_partial, float *dw, float slope_out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Hamilton product out = a (x) b with broadcasting over the point axis (src/modules/warp_utils.py:25-55
 * `mul_q`; used by `warp_quat_xyz` :78-94 and the pose composition, modellearn_proj_center.py:388-404).
 *   qa f32 [b,na,4], qb f32 [b,nb,4] as (w,x,y,z); na, nb in {1, n}; out f32 [b,max(na,nb),4].
 *   conj_a / conj_b != 0: that operand is conjugated first (the two gradient products
 *   d/da = g (x) conj(b), d/db = conj(a) (x) g are calls of this same entry).
 * --------------------------------------------------------------------------------------------- */
int i2p_quat_mul(int b, int na, int nb, int conj_a, int conj_b, const float *qa, const float *qb,
                 float *out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Per-quaternion scalings, forward and backward, one launch each (PyTorch runs 5-6 tiny kernels forward and
 * 10-12 backward for each):
 *   mode 0  inverse   : out = conj(q) / (|q|^2 + 1e-10)                      (src/modules/warp_utils.py:10-22 `inv_q`)
 *   mode 1  normalise : out = q / (sqrt(|q|^2 + 1e-10) + 1e-10)              (src/projectPN/PPBackbone_center.py:562)
 * q, out f32 [rows,4] as (w,x,y,z).  Backward: dq f32 [rows,4] from g = dL/dout and the forward INPUT q.
 * --------------------------------------------------------------------------------------------- */
int i2p_quat_unit_fwd(int mode, long long rows, const float *q, float *out, void *stream);
int i2p_quat_unit_bwd(int mode, long long rows, const float *q, const float *g, float *dq, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Row-wise unit variance of the cost-volume operands (src/projectPN/PPBackbone_center.py:388-393):
 *     y[r,:] = (x[r,:] - mean(x[r,:])) / clip(std_unbiased(x[r,:]), 1e-12)
 * x, y f32 [rows,c], 2 <= c <= 256; stat f32 [rows,2] = {1/d, std > 1e-12 ? 1 : 0} (saved for the backward).
 * Backward: gx = (gy - mean(gy) - stat1 * y * sum(gy*y)/(c-1)) * stat0.
 * --------------------------------------------------------------------------------------------- */
int i2p_row_unitvar_fwd(int rows, int c, const float *x, float *y, float *stat, void *stream);
int i2p_row_unitvar_bwd(int rows, int c, const float *gy, const float *y, const float *stat, float *gx,
                        void *stream);

/* ---------------------------------------------------------------------------------------------
 * Image-encoder block tail (src/modules/basicConv.py:13-17: BatchNorm2d -> LeakyReLU(0.1) -> MaxPool2d(3,
 * stride, padding=1)) in training mode, on the NHWC output y f32 [B,H,W,C] of the block's 3x3 convolution.
 *   sums: replicated {sum y, sum y^2} (i2p_bn_stats over rows = B*H*W); C % 4 == 0, (C/4) | 256; stride 1 or 2.
 *   forward: mean_invstd f32 [2C] written; running_mean/var (may be NULL) updated with `momentum` (unbiased
 *     variance; conv_bias, may be NULL, is added to the batch mean first: the bias the caller left out of y
 *     because it cancels in the normalisation); out f32 [B,Ho,Wo,C], Ho = (H-1)/stride+1;
 *     arg u8 [B,Ho,Wo,C] = window position kh*3+kw of the maximum (first maximum in scan order).
 *   backward: gout f32 [B,Ho,Wo,C] -> dy f32 [B,H,W,C] (gradient of the conv output), dgamma, dbeta f32 [C];
 *     dsums replicated f64 scratch, zeroed by the caller.
 * --------------------------------------------------------------------------------------------- */
int i2p_img_bn_pool_fwd(int B, int H, int W, int C, int stride, const float *y, const double *sums,
                        const float *gamma, const float *beta, float eps, float slope, float momentum,
                        const float *conv_bias, float *running_mean, float *running_var, float *out,
                        unsigned char *arg, float *mean_invstd, void *stream);
int i2p_img_bn_pool_bwd(int B, int H, int W, int C, int stride, const float *gout, const unsigned char *arg,
                        const float *y, const float *mean_invstd, const float *gamma, const float *beta,
                        float slope, double *dsums, float *dy, float *dgamma, float *dbeta, void *stream);
/* Device library only — second generation of the same block tail, two launches each way: the statistics pass is part of the call
 * (sums / dsums: zeroed [I2P_BN_REPLICAS][2C] doubles) and every block of the consumer kernel forms mean / invstd (forward) and
 * dbeta / dgamma (backward) from the replica sums in its prologue.  y_bf16: the conv output y and its gradient dy are bf16 bits
 * (MIOpen bf16 convolutions of the bf16 storage mode, BASELINE.json configs[2] / [4]); out_bf16: the pooled output and its gradient
 * are bf16.  Arithmetic fp32, statistics fp64, arg-max decided on the fp32 values; C <= 512. */
int i2p_img_block_fwd(int B, int H, int W, int C, int stride, int y_bf16, int out_bf16, const void *y, double *sums, const float *gamma,
                      const float *beta, float eps, float slope, float momentum, const float *conv_bias, float *running_mean,
                      float *running_var, void *out, unsigned char *arg, float *mean_invstd, void *stream);
int i2p_img_block_bwd(int B, int H, int W, int C, int stride, int y_bf16, int out_bf16, const void *gout, const unsigned char *arg,
                      const void *y, const float *mean_invstd, const float *gamma, const float *beta, float slope, double *dsums, void *dy,
                      float *dgamma, float *dbeta, void *stream);
/* i2p_img_block_fwd without its statistics pass: `sums` already holds sum y / sum y^2 (the producer of y accumulated them:
 * i2p_img_conv16_fwd). */
int i2p_img_block_pool(int B, int H, int W, int C, int stride, int y_bf16, int out_bf16, const void *y, double *sums, const float *gamma,
                       const float *beta, float eps, float slope, float momentum, const float *conv_bias, float *running_mean,
                       float *running_var, void *out, unsigned char *arg, float *mean_invstd, void *stream);
/* Device library only — 3x3 convolution, padding 1, stride 1, between NHWC fp32 tensors, (cin, cout) = (16, 16) or (16, 32): the image
 * encoder's blocks 2-5 (src/modules/basicConv.py:6-20: Conv2d(cin, cout, 3, padding=1) without its bias, which cancels in the
 * BatchNorm behind it), its input gradient and its weight gradient (csrc/image_conv16.hip); replaces F.conv2d / MIOpen for those
 * layers.  w [cout,cin,3,3] addressed by the four element strides ws[4] (host array).  sums (forward, may be NULL): f64
 * [I2P_BN_REPLICAS][2 cout] zeroed by the caller, receives sum y / sum y^2 per channel for i2p_img_block_pool.  H*W*128 < 2^31.
 * wgrad: dW (cout*cin*9 floats) is written in w's layout; partials: f32 [i2p_img_conv_wgrad_rows(B,H,W)][(cout/16)*2304] scratch
 * (block sums, added in fp64 in a fixed order: reproducible run to run, unlike the atomically accumulated split-K kernels it replaces).
 * bf16 must be 0: bf16-storage variants were measured in round 4 and removed in round 5 (outside the bf16 pose contract); the
 * argument stays in the signatures and is rejected with I2P_ERR_BAD_ARG. */
int i2p_img_conv_fwd(int B, int H, int W, int cin, int cout, int bf16, const void *x, const void *w, const int *ws, void *y, double *sums,
                     void *stream);
int i2p_img_conv_bwd_data(int B, int H, int W, int cin, int cout, int bf16, const void *dy, const void *w, const int *ws, void *dx, void *stream);
int i2p_img_block_bwd_dx(int B, int H, int W, int C, int stride, int y_bf16, int out_bf16, const void *gout, const unsigned char *arg,
                         const void *y, const float *mean_invstd, const float *gamma, const float *beta, float slope, double *dsums, void *dy,
                         float *dgamma, float *dbeta, void *stream);
/* i2p_img_block_bwd's statistics pass alone, and the backward of a fp32 16 -> 16 block with a stride-1 MaxPool that consumes it:
 * dy (gradient of the conv output, for i2p_img_conv_wgrad), dx (input gradient of the convolution), dgamma, dbeta from the incoming
 * gradient g [B,H,W,16] in ONE kernel (csrc/image_conv16.hip: the un-pooling and the BatchNorm backward are formed on load, row by row,
 * in front of the MFMAs); replaces i2p_img_block_bwd_dx + i2p_img_conv_bwd_data. */
int i2p_img_block_bwd_stats(int B, int H, int W, int C, int stride, int y_bf16, int out_bf16, const void *gout, const unsigned char *arg,
                            const void *y, const float *mean_invstd, const float *gamma, const float *beta, float slope, double *dsums,
                            void *stream);
int i2p_img_conv_tail_bwd(int B, int H, int W, const float *g, const unsigned char *arg, const float *y, const float *mean_invstd,
                          const float *gamma, const float *beta, float slope, const double *dsums, const float *w, const int *ws, float *dy,
                          float *dx, float *dgamma, float *dbeta, void *stream);
int i2p_img_conv_wgrad_rows(int B, int H, int W);
int i2p_img_conv_wgrad(int B, int H, int W, int cin, int cout, int bf16, const void *x, const void *dy, const int *ws, float *partials, void *dW,
                       void *stream);
/* Device library only — the FIRST block of the image encoder (src/modules/basicConv.py:6-20 with in_channel = 3: Conv2d(3, 16, 3,
 * padding 1) + BatchNorm2d + LeakyReLU + MaxPool2d(3, stride, 1)) without the conv output in memory (csrc/image_first.hip): the
 * convolution is recomputed from the input where it is needed, the batch statistics and the dense parts of the weight gradient come from
 * the 27 x 27 Gram matrix of the input windows.  Replaces F.conv2d (MIOpen) + i2p_img_block_fwd / F.conv2d's weight gradient +
 * i2p_img_block_bwd for that block; the input is not differentiated.
 *   x [B,3,H,W] fp32 addressed by element strides (sb, sc, sh, sw): NCHW or channels_last; wgt [16,3,3,3] addressed by the four element strides ws[4] (host
 *   array; dW comes back in the same layout), no bias in y
 *   (conv_bias only enters running_mean, as in i2p_img_block_fwd); gram: f64 [I2P_BN_REPLICAS][1024] zeroed by the caller;
 *   gram_red f64 [1024] written (rows / columns 0..26 = taps ci*9+kh*3+kw, 27 = the constant 1; kept for the backward);
 *   out [B,Ho,Wo,16] fp32 or bf16 (out_bf16), arg u8 [B,Ho,Wo,16] (window position of the first maximum), mean_invstd f32 [32].
 *   backward: gout [B,Ho,Wo,16] fp32 or bf16 (g_bf16) -> dW [16,3,3,3], dgamma, dbeta [16];
 *   partials: f32 [i2p_img_first_bwd_rows(B,H,W,stride)][16*29] scratch.  B*H*W < 2^31, stride 1 or 2, sw = 1, 0 <= slope <= 1.
 *   parts: 1 = statistics only (gram, gram_red, mean_invstd, running buffers; out / arg / gamma / beta may be NULL), 2 = output only
 *   (reads mean_invstd; gram / gram_red may be NULL), 3 = both. */
int i2p_img_first_fwd(int B, int H, int W, int stride, const float *x, long long sb, long long sc, long long sh, long long sw,
                      const float *wgt, const int *ws, const float *gamma, const float *beta, float eps, float slope, float momentum,
                      const float *conv_bias, float *running_mean, float *running_var, double *gram, double *gram_red, int out_bf16,
                      void *out, unsigned char *arg, float *mean_invstd, int parts, void *stream);
int i2p_img_first_bwd_rows(int B, int H, int W, int stride);
int i2p_img_first_bwd(int B, int H, int W, int stride, const float *x, long long sb, long long sc, long long sh, long long sw,
                      const float *wgt, const int *ws, const float *gamma, const float *beta, float slope, const float *mean_invstd,
                      const double *gram_red, int g_bf16, const void *gout, const unsigned char *arg, float *partials, float *dW,
                      float *dgamma, float *dbeta, void *stream);

/* ---------------------------------------------------------------------------------------------
 * First layer of the all-pixel cost volume (src/projectPN/PPBackbone_center.py:383-418): the
 * reference builds [B,N,M,6+C(+C)] = cat(xyz_n, uv_k, norm(LF_n)*norm(RF_k) (, max-response_k))
 * and runs a 1x1 conv on it.  Factored here as
 *     y[b,n,k,:] = (f[b,n,:] * g[b,k,:]) . w^T + bias_n[b,n,:] + bias_k[b,k,:]
 * (bias_n = w_xyz . xyz_n, bias_k = w_uv . uv_k + w_bv . response_k, computed by the caller on
 * the small [B,N,*] / [B,M,*] tensors) so the [B,N,M,C] product is never materialised.
 *   f f32 [B,N,cin], g f32 [B,M,cin], bias_n f32 [B,N,cout], bias_k f32 [B,M,cout],
 *   w f32 [cout,cin]; y f32 [B*N*M, cout] (pre-BN), sums replicated f64 or NULL.
 * Backward (g^y = dL/dy given directly when out_coef is NULL, else BN-backward on load):
 *   dw [cout,cin] written; d_f, d_g, d_bias_n, d_bias_k accumulated with fp32 atomics (zeroed by caller).
 * --------------------------------------------------------------------------------------------- */
int i2p_pair_lin_fwd(int B, int N, int M, int cin, int cout, const float *f, const float *g,
                     const float *bias_n, const float *bias_k, const float *w, float *y, double *sums,
                     void *stream);
/* dw_partial: f32 scratch [i2p_pair_lin_bwd_grid(B,N,M)][cout*cin] */
int i2p_pair_lin_bwd_grid(int B, int N, int M);
int i2p_pair_lin_bwd(int B, int N, int M, int cin, int cout, const float *gz, const float *y,
                     const float *out_coef, const float *out_mi, const double *out_dsums, const float *f,
                     const float *g, const float *w, float *d_f, float *d_g, float *d_bias_n,
                     float *d_bias_k, float *dw_partial, float *dw, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Cost-volume tail (PPBackbone_center.py:420-433) without concatenated / activated tensors.
 *  - two-source layer: the input of mlp2[0] is cat(pi_xyz_encoding, pi_feat1_new) = two different
 *    pre-BN tensors xa [rows,cin_a], xb [rows,cin_b], each with its own BN coefficients / activation;
 *    the backward writes the two halves of gz_in separately and adds e_add_b (the gradient that
 *    reaches activation b from its other consumer, the weighted sum) before the activation derivative.
 *  - softmax-weighted sum over the M pixels of every point, on the pre-BN logits y5 and values y3:
 *      out[b,n,:] = sum_k softmax_k(act(bn5(y5))) * act(bn3(y3));  msave f32 [B*N,2,C] (max, exp-sum)
 *      backward: gz5 (+ replicated BN-backward sums of bn5) and ga3 = dL/d act(bn3(y3)).
 * --------------------------------------------------------------------------------------------- */
int i2p_lin_fwd_2src(long long rows, int cin_a, int cin_b, int cout, const float *xa, const float *coef_a,
                     float slope_a, const float *xb, const float *coef_b, float slope_b, const float *w,
                     float *y, double *sums, void *stream);
int i2p_lin_bwd_2src(long long rows, int cin_a, int cin_b, int cout, const float *gz, const float *y,
                     const float *out_coef, const float *out_mi, const double *out_dsums, const float *xa,
                     const float *coef_a, const float *mi_a, float slope_a, const float *xb,
                     const float *coef_b, const float *mi_b, float slope_b, const float *e_add_b,
                     const float *w, float *gz_a, double *dsums_a, float *gz_b, double *dsums_b,
                     float *dw_partial, float *dw, void *stream);
int i2p_cv_softmax_wsum_fwd(int B, int N, int M, int C, const float *y5, const float *coef5, float slope5,
                            const float *y3, const float *coef3, float slope3, float *out, float *msave,
                            void *stream);
int i2p_cv_softmax_wsum_bwd(int B, int N, int M, int C, const float *g_out, const float *out,
                            const float *msave, const float *y5, const float *coef5, const float *mi5,
                            float slope5, const float *y3, const float *coef3, float slope3, float *gz5,
                            double *dsums5, float *ga3, void *stream);

/* Set-abstraction tail (PPBackbone_center.py:28-46 + :129 `torch.max(new_points, dim=2)`): BN + activation of the
 * last pre-BN tensor y f32 [groups*K, c] and the max over the K neighbours of each group in one pass:
 *   out f32 [groups, c], arg u8 [groups, c] (first k attaining the maximum); coef f32 [3][c] from i2p_bn_finalize.
 * i2p_unpool_k is its gradient routing: gd f32 [groups*K, c] = g[grp,:] at row grp*K+arg, 0 elsewhere (one write).
 * c % 4 == 0, (c/4) | 256, K <= 255. */
int i2p_bn_act_maxk_fwd(long long groups, int K, int c, const float *y, const float *coef, float slope,
                        float *out, unsigned char *arg, void *stream);
int i2p_unpool_k(long long groups, int K, int c, const float *g, const unsigned char *arg, float *gd, void *stream);

/* Gradient of the factors of the all-pixel position encoding ye[b,n,k,:] = enc_n[b,n,:] + enc_k[b,k,:]
 * (PPBackbone_center.py:416-418: pi_encoding applied to cat(xyz_n, uv_k), a 1x1 conv => an outer sum):
 *   gz f32 [B*N*M, C] = dL/dz_e (z_e = BN(ye)), dsums replicated {sum gz, sum gz*xhat}, coef [3][C], mi [2][C];
 *   d_enc_n f32 [B,N,C] = sum_k dL/dye, d_enc_k f32 [B,M,C] = sum_n dL/dye, dL/dye = scale*(gz - m1 - xhat*m2).
 *   sum_k f32 [B,N,C], sum_n f32 [B,M,C]: scratch, both ZEROED BY THE CALLER (accumulated with atomics).
 * gz is read once; the [B,N,M,C] gradient of ye is never formed.  C % 4 == 0, C | 256. */
int i2p_pair_bias_bn_bwd(int B, int N, int M, int C, const float *gz, const float *enc_n, const float *enc_k,
                         const double *dsums, const float *coef, const float *mi, float *sum_k, float *sum_n,
                         float *d_enc_n, float *d_enc_k, void *stream);

/* Learned-uncertainty pose loss (compute_loss.py:102-133 `Get_loss`) and its gradient in one launch:
 *   out3 (fine), out4 (coarse) f32 [B,7] = (q[4], t[3]); q_gt f32 [B,4]; t_gt f32 [B,3]; w_x, w_q f32 [1] (the learned
 *   log-uncertainties sx, sq); l1_trans: translation term is mean |dt| (cfg.l1_trans_loss) else mean ||dt||.
 *   loss3 f32 [3] = {loss, rotation part, translation part}; d_out3, d_out4 f32 [B,7] = dloss/dout; d_w f32 [2] =
 *   {dloss/dw_x, dloss/dw_q}.  B <= 1024. */
int i2p_pose_loss(int B, int l1_trans, const float *out3, const float *out4, const float *q_gt, const float *t_gt,
                  const float *w_x, const float *w_q, float *loss3, float *d_out3, float *d_out4, float *d_w, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * bf16 mode (BASELINE.json configs[2], configs[4]; the reference itself is fp32-only, train20v2learn_wandb_proj.py:107):
 * the pre-BN [rows, C] tensors of the fused layer chains (PPBackbone_center.py:34-46 blocks) and the gradients between
 * their layers are STORED as bf16 (raw bits, `unsigned short`), the contractions run on v_mfma_f32_32x32x16_bf16 with
 * fp32 accumulation, BN statistics stay fp64, parameters, BN/activation/softmax arithmetic and every other tensor
 * stay fp32.  Every entry below is the bf16-storage twin of the entry of the same name without the suffix; a BN is
 * evaluated as fmaf(y, invstd*gamma, beta - mean*invstd*gamma) on the rounded y in all of them.
 * Channel counts: bf16 tensors 16/32/64/128; an fp32 `x` (a chain's raw input) any multiple of 4 up to 160 (BN on
 * load then needs cin in {16,..,128}).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef unsigned short i2p_bf16;
int i2p_lin_bwd_bf16_grid(long long rows);     /* blocks of the wgrad launch: dw_partial holds grid*cout*cin + 8*cout floats */
int i2p_lin_fwd_bf16(long long rows, int cin, int cout, const void *x, int x_bf16, const float *in_coef, float slope_in,
                     const float *w, i2p_bf16 *y, double *sums, void *stream);
int i2p_lin_fwd_2src_bf16(long long rows, int cin_a, int cin_b, int cout, const i2p_bf16 *xa, const float *coef_a,
                          float slope_a, const i2p_bf16 *xb, const float *coef_b, float slope_b, const float *w,
                          i2p_bf16 *y, double *sums, void *stream);
int i2p_pair_lin_fwd_bf16(int B, int N, int M, int cin, int cout, const float *f, const float *g, const float *bias_n,
                          const float *bias_k, const float *w, i2p_bf16 *y, double *sums, void *stream);
/* gz_in: bf16 [rows,cin] (gz_in_bf16 = 1: x must be bf16; activation derivative + statistics when in_coef is given) or
 * fp32 [rows,cin] (gz_in_bf16 = 0: gradient of a raw fp32 input, no BN in front) or NULL. */
int i2p_lin_bwd_bf16(long long rows, int cin, int cout, const i2p_bf16 *gz, const i2p_bf16 *y, const float *out_coef,
                     const float *out_mi, const double *out_dsums, const void *x, int x_bf16, const float *in_coef,
                     const float *in_mi, float slope_in, const float *w, void *gz_in, int gz_in_bf16, double *in_dsums,
                     float *dw_partial, float *dw, float slope_out, void *stream);
int i2p_lin_bwd_2src_bf16(long long rows, int cin_a, int cin_b, int cout, const i2p_bf16 *gz, const i2p_bf16 *y,
                          const float *out_coef, const float *out_mi, const double *out_dsums, const i2p_bf16 *xa,
                          const float *coef_a, const float *mi_a, float slope_a, const i2p_bf16 *xb, const float *coef_b,
                          const float *mi_b, float slope_b, const i2p_bf16 *e_add_b, const float *w, i2p_bf16 *gz_a,
                          double *dsums_a, i2p_bf16 *gz_b, double *dsums_b, float *dw_partial, float *dw, void *stream);
int i2p_pair_lin_bwd_bf16_grid(int B, int N, int M);
int i2p_pair_lin_bwd_bf16(int B, int N, int M, int cin, int cout, const i2p_bf16 *gz, const i2p_bf16 *y,
                          const float *out_coef, const float *out_mi, const double *out_dsums, const float *f,
                          const float *g, const float *w, float *d_f, float *d_g, float *d_bias_n, float *d_bias_k,
                          float *dw_partial, float *dw, void *stream);
/* streaming kernels on bf16 tensors (csrc/bf16_stream.hip) */
int i2p_outer_sum_bf16(int B, int N, int M, int C, const float *enc_n, const float *enc_k, i2p_bf16 *ye, double *sums,
                       void *stream); /* the same in fp32 storage: ye f32 [B*N*M, C] = enc_n[b,n,:] + enc_k[b,k,:] (PPBackbone_center.py:416-421 position encoding of all
 * point x pixel pairs, pre-BN) and its replicated BN sums in ONE pass (was: broadcast add + i2p_bn_stats) */
int i2p_outer_sum(int B, int N, int M, int C, const float *enc_n, const float *enc_k, float *ye, double *sums, void *stream);
         /* ye[b,n,k,:] = bf16(enc_n[b,n,:] + enc_k[b,k,:]), sums += {sum, sum^2} */
int i2p_to_bf16(long long n, const float *x, i2p_bf16 *y, void *stream);
int i2p_bn_act_fwd_bf16(long long rows, int c, const i2p_bf16 *y, const float *coef, float slope, float *out, void *stream);
int i2p_bn_act_maxk_fwd_bf16(long long groups, int K, int c, const i2p_bf16 *y, const float *coef, float slope, float *out,
                             unsigned char *arg, void *stream);
int i2p_unpool_k_bf16(long long groups, int K, int c, const float *g, const unsigned char *arg, i2p_bf16 *gd, void *stream);
int i2p_bn_act_bwd_stats_bf16(long long rows, int c, const i2p_bf16 *dout, const i2p_bf16 *y, const float *coef,
                              const float *mi, float slope, double *dsums, void *stream);
int i2p_cv_softmax_wsum_fwd_bf16(int B, int N, int M, int C, const i2p_bf16 *y5, const float *coef5, float slope5,
                                 const i2p_bf16 *y3, const float *coef3, float slope3, float *out, float *msave, void *stream);
int i2p_cv_softmax_wsum_bwd_bf16(int B, int N, int M, int C, const float *g_out, const float *out, const float *msave,
                                 const i2p_bf16 *y5, const float *coef5, const float *mi5, float slope5, const i2p_bf16 *y3,
                                 const float *coef3, float slope3, i2p_bf16 *gz5, double *dsums5, i2p_bf16 *ga3, void *stream);
/* Layer forward + BN finalisation in ONE launch: as i2p_lin_fwd / i2p_lin_fwd_2src / i2p_pair_lin_fwd, plus what
 * i2p_bn_finalize(rows, cout, sums, gamma, beta, eps, coef, mean_invstd) would compute, formed by the last block of the
 * layer kernel to finish; `counter` = one uint32 of caller scratch, zero on entry (left zero on exit). */
int i2p_lin_fwd_fin(long long rows, int cin, int cout, const float *x, const float *in_coef, float slope_in, const float *w,
                    float *y, double *sums, const float *gamma, const float *beta, float eps, float *coef,
                    float *mean_invstd, unsigned *counter, void *stream);
int i2p_lin_fwd_2src_fin(long long rows, int cin_a, int cin_b, int cout, const float *xa, const float *coef_a, float slope_a,
                         const float *xb, const float *coef_b, float slope_b, const float *w, float *y, double *sums,
                         const float *gamma, const float *beta, float eps, float *coef, float *mean_invstd,
                         unsigned *counter, void *stream);
int i2p_pair_lin_fwd_fin(int B, int N, int M, int cin, int cout, const float *f, const float *g, const float *bias_n,
                         const float *bias_k, const float *w, float *y, double *sums, const float *gamma, const float *beta,
                         float eps, float *coef, float *mean_invstd, unsigned *counter, void *stream);
/* A whole small MLP chain forward in ONE launch (csrc/mlp_chain.hip): nl <= 4 blocks of 1x1 conv -> batch-statistics BN ->
 * LeakyReLU (Conv2d.forward, PPBackbone_center.py:34-46, stacked as in PPBackbone_center.py:77-131, 241-296, 582-603) on a
 * resident grid with the 64-row activation strips kept in LDS and one grid barrier per BN; replaces nl x i2p_lin_fwd_fin +
 * i2p_bn_act_fwd / i2p_bn_act_maxk_fwd (+ i2p_pad_cols) for row counts whose grid is co-resident.
 *   widths[0] = row length of x (multiple of 4, <= 272), widths[1..nl] = output widths (multiples of 64, <= 256);
 *   w_ld[l] = row length of w[l] = the layer's real input width (<= widths[l]; x columns beyond it are ignored);
 *   w / gamma / beta / y / coef / mean_invstd: HOST arrays of nl device pointers; slopes: host array of nl floats;
 *   y[l] f32 [rows, widths[l+1]] pre-BN outputs, coef[l] [3][c] and mean_invstd[l] [2c] as i2p_bn_finalize writes them;
 *   sums: i2p_chain_sums_len(nl, max output width) zeroed doubles; sync: i2p_chain_sync_words() zeroed uint32 (left zero; word
 *   i2p_chain_sync_words() - 32 is set to 1 if a grid barrier timed out, i.e. the grid was not resident and the results are invalid);
 *   pool_k = 0: out f32 [rows, c_last] = act(bn(y_last));  pool_k > 0 (divides 64 and rows): out [rows / pool_k, c_last] = max over
 *   consecutive groups of pool_k rows and arg u8 (first k attaining it) as i2p_bn_act_maxk_fwd;
 *   w0_pad: NULL or f32 [widths[1]][widths[0]] receiving w[0] with zero columns appended.
 * i2p_chain_fwd_ok: 1 if the shape is taken on the current device (needs a GPU), else 0. */
int i2p_chain_fwd_ok(long long rows, int nl, const int *widths, int pool_k);
long long i2p_chain_sums_len(int nl, int cmax_out);
long long i2p_chain_sync_words(void);
/* Error sinks of the CURRENT device for the chain kernels' grid barriers (a barrier that is not complete after the poll limit —
 * the grid was not co-resident: CU mask, another process, an over-sized grid — is abandoned, the launch's results are invalid):
 *   device_counter: device f32 [>= 1], +1.0 per launch with a timed-out barrier (i2p_clip_adam's `poison` argument reads it);
 *   host_flag: pinned (device-accessible) host uint32 [1], set to 1 at the same moment: the host polls it without synchronising.
 * Both zeroed by the caller and kept alive; NULLs unregister.  The per-launch error word is sync[i2p_chain_sync_words() - 32].
 * i2p_chain_resident_blocks: blocks of chain kernel `kind` (0/1/2: forward with 16/32/64-row strips, 3: backward) the current
 * device holds at once with `lds_bytes` of dynamic LDS = hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs (capped at the
 * blocks per CU the kernel is written for): the number i2p_chain_fwd_ok / i2p_chain_bwd_ok check a grid against. */
int i2p_chain_set_error_words(float *device_counter, unsigned *host_flag);
int i2p_chain_resident_blocks(int kind, long long lds_bytes);
int i2p_chain_fwd(long long rows, int nl, const int *widths, const int *w_ld, const float *x, const float *const *w,
                  const float *const *gamma, const float *const *beta, const float *slopes, float eps, float *const *y,
                  float *const *coef, float *const *mean_invstd, double *sums, int pool_k, float *out, unsigned char *arg,
                  float *w0_pad, unsigned *sync, void *stream);
/* Backward of i2p_chain_fwd in TWO launches (the chain on the same resident grid, then the reduction of the per-block weight-gradient
 * slabs in block order, i.e. deterministic): replaces, per layer, i2p_bn_act_bwd_stats / i2p_lin_bwd (dgrad + wgrad + reduction) and
 * the un-pooling launch (reference: autograd of Conv2d.forward, PPBackbone_center.py:34-46, and of torch.max over K, :129).
 *   x, w, widths, w_ld, slopes, pool_k, arg as in the forward; y / coef / mean_invstd: the forward's outputs (host arrays of nl pointers);
 *   g = dL/dout: f32 [rows, c_last], or [rows / pool_k, c_last] when pool_k > 0;
 *   gx: NULL or f32 [rows, widths[0]] = dL/dx;  dgamma[l], dbeta[l]: f32 [widths[l+1]];
 *   dw: f32 [i2p_chain_bwd_slab(...)] = the dW_l [widths[l+1]][w_ld[l]] back to back; dw_part: ceil(rows / 64) slabs of scratch;
 *   sums: i2p_chain_sums_len(nl, max output width) zeroed doubles; sync: as in the forward. */
int i2p_chain_bwd_ok(long long rows, int nl, const int *widths, int pool_k);
long long i2p_chain_bwd_slab(int nl, const int *widths, const int *w_ld);
int i2p_chain_bwd(long long rows, int nl, const int *widths, const int *w_ld, const float *x, const float *const *w,
                  const float *const *y, const float *const *coef, const float *const *mean_invstd, const float *slopes,
                  const float *g, const unsigned char *arg, int pool_k, float *gx, float *dw_part, float *dw,
                  float *const *dgamma, float *const *dbeta, double *sums, unsigned *sync, void *stream);
/* deterministic variants (no floating-point atomics; fixed summation order => bitwise reproducible gradients):
 * i2p_pair_lin_bwd takes its slabs from dw_partial, which must hold i2p_pair_lin_bwd_scratch(...) floats;
 * i2p_pair_bias_bn_bwd_det is i2p_pair_bias_bn_bwd with caller scratch of i2p_pair_bias_bn_bwd_scratch(...) floats. */
long long i2p_pair_lin_bwd_scratch(int B, int N, int M, int cin, int cout);
long long i2p_pair_bias_bn_bwd_scratch(int B, int N, int M, int C);
int i2p_pair_bias_bn_bwd_det(int B, int N, int M, int C, const float *gz, const float *enc_n, const float *enc_k,
                             const double *dsums, const float *coef, const float *mi, float *scratch, float *d_enc_n,
                             float *d_enc_k, void *stream);
/* Level-1 set-abstraction front end (PPBackbone_center.py:152-187: get_neighbor_copy + gather_torch x2 + feature build)
 * in one launch with the window strip of both range images staged in LDS: selection = i2p_fused_conv_select_k with
 * FLAG_SHIFT|FLAG_COPY, stride 1, random_hw = arange, centres = the cells (qh*stride_h, qw*stride_w);
 * feat f32 [B, out_h*out_w, K, 12] = [nbr_raw - centre_raw (3), centre (3), nbr_raw (3), |.| (1), 0, 0]. */
int i2p_sa_l1_group(int B, int H, int W, int out_h, int out_w, int stride_h, int stride_w, int kH, int kW, int K,
                    float distance, const float *sel_xyz, const float *raw_xyz, float *feat, void *stream);
/* i2p_gather_rows_grad on int64 fixed-point atomics (order-independent, deterministic, 2^-40 of max|grad_out| resolution):
 * scratch = i2p_gather_rows_grad_fx_scratch(b,hw,c) BYTES, zeroed by the caller; grad_feat is accumulated into. */
long long i2p_gather_rows_grad_fx_scratch(int b, int hw, int c);
int i2p_gather_rows_grad_fx(int b, int hw, int c, int q, int W, const float *grad_out, const int64_t *h_idx,
                            const int64_t *w_idx, void *scratch, float *grad_feat, void *stream);
/* the same with a strided source: the c scattered channels are columns [off, off+c) of grad_out rows of pitch ld floats
 * (backward of i2p_sa_rows: no slice copy of the grouped-row gradient) */
int i2p_gather_rows_grad_fx_ld(int b, int hw, int c, int q, int W, const float *grad_out, int ld, int off, const int64_t *h_idx,
                               const int64_t *w_idx, void *scratch, float *grad_feat, void *stream);
/* Grouped input rows of a set-abstraction / up-convolution MLP in one launch (reference: gather_torch of the xyz image,
 * subtraction of the centre, gather_torch of the feature image, cat — PPBackbone_center.py:94-129, :236-262):
 * out f32 [b, n*K, cpad]: columns [xyz_col, +3) = xyz[b,cell,:] - centre[b,n,:], [feat_col, +c) = feat[b,cell,:], zeros elsewhere;
 * cell = h_idx*W + w_idx [b, n*K] i64, xyz f32 [b,hw,3], centre f32 [b,n,3], feat f32 [b,hw,c]; cpad a multiple of 4, >= 3 + c. */
int i2p_sa_rows(int b, int hw, int n, int K, int W, int c, int cpad, int xyz_col, int feat_col, const float *xyz, const float *centre,
                const float *feat, const int64_t *h_idx, const int64_t *w_idx, float *out, void *stream);
/* Weight gradient of a plain linear layer (autograd of F.linear in the reference: dW = g^T x, e.g. basicConv.py:22-58 layers
 * outside the fused kernels' shape limits): out f32 [m, n] = sum_r a[r, :m]^T b[r, :n], a f32 rows of pitch lda, b of pitch ldb;
 * rows are cut over the grid and the chunk results summed in a fixed order (bit-reproducible).
 * scratch = i2p_gemm_tn_scratch(rows, m, n) BYTES (need not be zeroed). */
long long i2p_gemm_tn_scratch(long long rows, int m, int n);
int i2p_gemm_tn(long long rows, int m, int n, const float *a, int lda, const float *b, int ldb, void *scratch, float *out,
                void *stream);
/* Input rows of the kNN pi-stage of the fine cost volume in one launch (reference: knn grouping of the pixel coordinates and features,
 * product with the point feature, cat — PPBackbone_center.py:369-395): out f32 [b, n*K, cpad] =
 * [xyz[b,n,:] (3), pix_xyz[b,idx,:] (3), pts[b,n,:] * pix[b,idx,:] (c), zeros]; xyz [b,n,3], pix_xyz [b,m,3], pts [b,n,c], pix [b,m,c],
 * idx i64 [b, n*K] (pixel index of each neighbour); cpad a multiple of 4, >= 6 + c. */
int i2p_knn_rows_fwd(int b, int n, int m, int K, int c, int cpad, const float *xyz, const float *pix_xyz, const float *pts,
                     const float *pix, const int64_t *idx, float *out, void *stream);
/* its backward for the row gradient g [b, n*K, cpad]: d_pts [b,n,c] = sum_k g_f * pix[idx], d_xyz [b,n,3] = sum_k g[:, 0:3] (NULL:
 * skipped), gq [b, n*K, c] = g_f * pts — the per-neighbour gradient of the gathered pixel features, to be scattered with
 * i2p_gather_rows_grad_fx.  Sums over the K neighbours in index order (bit-reproducible). */
int i2p_knn_rows_bwd(int b, int n, int m, int K, int c, int cpad, const float *g, const float *pts, const float *pix,
                     const int64_t *idx, float *d_xyz, float *d_pts, float *gq, void *stream);
int i2p_pair_bias_bn_finish(int B, int N, int M, int C, const float *sum_k, const float *sum_n, const float *enc_n,
                            const float *enc_k, const double *dsums, const float *coef, const float *mi, float *d_enc_n,
                            float *d_enc_k, void *stream);   /* closed-form half of i2p_pair_bias_bn_bwd on formed sums */
int i2p_pair_bias_bn_bwd_bf16(int B, int N, int M, int C, const i2p_bf16 *gz, const float *enc_n, const float *enc_k,
                              const double *dsums, const float *coef, const float *mi, float *sum_k, float *sum_n,
                              float *d_enc_n, float *d_enc_k, void *stream);

/* Global-norm gradient clip + Adam with L2 weight decay on flat fp32 buffers in two launches (reference:
 * train20v2learn_wandb_proj.py:198-205 torch.optim.Adam(lr 1e-3, betas (0.9, 0.999), eps 1e-8, weight_decay 1e-4) after
 * :472-476 clip_grad_norm_(max_norm 10)).  n % 4 == 0, buffers 16-byte aligned; grad is replaced by gscale*clip_factor*grad
 * (what the optimiser consumed); step / lr are device scalars (step is advanced by one); mask NULL or [n] of 0/1 (parameters
 * autograd leaves without a gradient take no decay and no update, as torch.optim.Adam skips them); partials: scratch of
 * >= 256 doubles; total_out NULL or [1] = the global norm before clipping.  clip <= 0: no clipping.
 * poison: NULL or device f32 [1]; a non-zero value (the chain kernels' error counter, i2p_chain_set_error_words, which the
 * trainer carries at the end of the all-reduced gradient buffer) turns the call into a no-op: step, moments and parameters
 * stay as they are — a step whose gradients came from an abandoned grid barrier is never applied, on any rank. */
int i2p_clip_adam(long long n, float *param, float *grad, float *exp_avg, float *exp_avg_sq, const float *mask, double *partials,
                  float *step, const float *lr, double beta1, double beta2, float eps, float weight_decay, float clip, float gscale,
                  float *total_out, const float *poison, void *stream);

/* Deferred weight-gradient reductions (csrc/deferred.hip).  The reference's step needs no weight gradient before clip + optimiser
 * (train20v2learn_wandb_proj.py:473-481).  Between i2p_defer_begin() and i2p_defer_end() the layer-backward entries (i2p_lin_bwd*,
 * i2p_gemm_tn, i2p_chain_bwd, the bf16 backward entries) RECORD the sum of their per-block weight-gradient slabs instead of
 * launching it; i2p_defer_flush(stream) sums every recorded one in one launch per 24 entries, bit-identical to the immediate form.
 * The caller keeps the `dw_partial` / scratch buffers alive and unread until the flush and does not read a `dw` earlier;
 * i2p_defer_pause(1) ... i2p_defer_pause(0) brackets calls whose `dw` is consumed at once.  i2p_defer_pending(): recorded, not yet
 * flushed; i2p_defer_end() switches deferral off and returns the number of recorded reductions it had to drop (non-zero = a bug in
 * the caller: those weight gradients were never formed).  Process-wide state, guarded by a mutex (autograd's device thread records,
 * the caller's thread flushes). */
int i2p_defer_begin(void);
int i2p_defer_pause(int on);
/* one-shot: the NEXT recorded fp32 reduction, if its slabs are [rows][pitch] floats, keeps the first `cols` columns and writes them
 * densely as [rows][cols] at the start of its `dw` (weight gradient of a layer whose input rows are zero-padded); rows = 0 withdraws the
 * request; returns 1 if the previous request was taken by a reduction, else 0 */
int i2p_defer_compact_next(int rows, int pitch, int cols);
int i2p_defer_pending(void);
int i2p_defer_flush(void *stream);
int i2p_defer_end(void);

/* One-launch replacements for clusters of small elementwise launches (csrc/glue.hip).
 * i2p_row_valid: out[r] = 1.0 if any x[r, 0..c) != 0 else 0.0 — check_valid (src/projectPN/utils.py:106-108).
 * i2p_mask_fill: out[r, :] = valid[r] > 0 ? x[r, :] : fill — the reference's x*valid + (-1e10)*(1-valid) for a 0/1 row mask
 *   (modellearn_proj_center.py:318,376, PPBackbone_center.py:481); its backward is the same call on the gradient with fill = 0.
 * i2p_pad_cols: out [rows, cpad] = [w [rows, c], zeros].
 * i2p_strided_pick2: oa[b,h,w,:] = a[b, h*sh, w*sw, :] (ob from b unless b is NULL), [B,H,W,3] -> [B,oh,ow,3] —
 *   the strided centre picks of a set-abstraction level (PPBackbone_center.py:94-95). */
/* Backward-validation feature of the first cost volume in closed form (PPBackbone_center.py:408-414): respond[b,k,c] = max over
 * the valid points n of pts[b,n,c]*pix[b,k,c] = pix * (pix >= 0 ? max_n pts : min_n pts), -1e10 if the sample has no valid point.
 * pts f32 [B,N,C], pix f32 [B,M,C], valid f32 [B,N] (0/1) -> respond f32 [B,M,C]; for the backward: fmaxmin f32 [B,2,C], imaxmin
 * i32 [B,2,C] (arg-max / arg-min point, lowest index on ties), anyv i32 [B].  Backward: g = dL/drespond -> d_pts [B,N,C] (the
 * gradient reaches the arg-max / arg-min point, as torch.max's does; written completely), d_pix [B,M,C]. */
int i2p_max_response_fwd(int B, int N, int M, int C, const float *pts, const float *pix, const float *valid, float *respond,
                         float *fmaxmin, int *imaxmin, int *anyv, void *stream);
int i2p_max_response_bwd(int B, int N, int M, int C, const float *g, const float *pix, const float *fmaxmin, const int *imaxmin,
                         const int *anyv, float *d_pts, float *d_pix, void *stream);
/* pc-stage front end of the cost volumes in one launch each way (reference: two gather_torch, expand, subtraction, squared norm,
 * sqrt, two cats — PPBackbone_center.py:443-476).  xyz f32 [b,hw,3], pts f32 [b,hw,C], feat f32 [b,hw,c], h_idx/w_idx i64 [b,hw*K]
 * (cell = h*W + w) -> geo f32 [b,hw*K,12] = [xyz[n], xyz[cell], xyz[cell]-xyz[n], sqrt(|.|^2+1e-20), 0, 0],
 * part f32 [b,hw*K,C+c] = [pts[n], feat[cell]], nbf f32 [b,hw*K,c] = feat[cell].
 * Backward: g_geo / g_part / g_nbf -> d_pts [b,hw,C] (sums over the K neighbours in index order), comb f32 [b,hw,c+4] = the own-point
 * part of [d_feat | d_xyz | 0], rows f32 [b,hw*K,c+4] = the per-neighbour parts, to be scattered onto comb with
 * i2p_gather_rows_grad_fx (c+4 channels). */
int i2p_pc_rows_fwd(int b, int hw, int K, int W, int C, int c, const float *xyz, const float *pts, const float *feat,
                    const int64_t *h_idx, const int64_t *w_idx, float *geo, float *part, float *nbf, void *stream);
int i2p_pc_rows_bwd(int b, int hw, int K, int W, int C, int c, const float *xyz, const int64_t *h_idx, const int64_t *w_idx,
                    const float *g_geo, const float *g_part, const float *g_nbf, float *d_pts, float *comb, float *rows, void *stream);
/* Pose-head MLP in one launch each way (PPBackbone_center.py:553-562): hid = (W1 pooled + b1) * mask, qraw = Wq hid + bq,
 * t = Wt hid + bt, q = qraw / (sqrt(|qraw|^2 + 1e-10) + 1e-10).  pooled f32 [B,C], w1 [H,C], mask [B,H] (dropout multiplier, NULL = none). */
int i2p_pose_head_fwd(int B, int C, int H, const float *pooled, const float *w1, const float *b1, const float *mask, const float *wq,
                      const float *bq, const float *wt, const float *bt, float *hid, float *qraw, float *q, float *t, void *stream);
int i2p_pose_head_bwd(int B, int C, int H, const float *gq, const float *gt, const float *qraw, const float *hid, const float *mask,
                      const float *pooled, const float *w1, const float *wq, const float *wt, float *d_pooled, float *dw1, float *db1,
                      float *dwq, float *dbq, float *dwt, float *dbt, void *stream);
/* Composition of the fine pose with the coarse one (modellearn_proj_center.py:388-404): out f32 [B,7] = [q3 (x) qp, (q3 (x) [0,tp] (x)
 * q3^-1)[1:4] + t3] with q^-1 = conj(q) / (|q|^2 + 1e-10), the operation order of i2p_quat_mul / i2p_quat_unit_fwd; backward from
 * g [B,7]: dq3 [B,4], dt3 [B,3], dqp [B,4], dtp [B,4]; tp / dtp quaternion-shaped [B,4] = [0, t_prev] (w of tp read as 0).  One launch
 * each (the unfused chain: 7 forward, ~14 backward). */
int i2p_pose_compose_fwd(int B, const float *q3, const float *t3, const float *qp, const float *tp, float *out, void *stream);
int i2p_pose_compose_bwd(int B, const float *q3, const float *qp, const float *tp, const float *g, float *dq3, float *dt3, float *dqp,
                         float *dtp, void *stream);
/* Warp of a cloud by a pose + empty-cell mask + depth split in one launch each way (warp_utils.py:78-94,
 * modellearn_proj_center.py:345-352, PPBackbone_center.py:377): p f32 [B,N,3], q f32 [B,4] (w,x,y,z), t f32 [B,4] = [0,tx,ty,tz],
 * valid f32 [B,N] (0/1) or NULL -> p' = (q (x) [0,p] (x) conj(q)/(|q|^2+1e-10) + t)[1:4] * valid; z [B,N] = p'_z; uv [B,N,3] = p' / (z + 1e-10);
 * xyz [B,N,3] = uv * z.  Backward: g_uv / g_z / g_xyz (any may be NULL) -> dq [B,4], dt [B,4] (p and valid are data). */
int i2p_warp_split_fwd(int B, int N, const float *p, const float *q, const float *t, const float *valid, float *uv, float *z, float *xyz,
                       void *stream);
int i2p_warp_split_bwd(int B, int N, const float *p, const float *q, const float *t, const float *valid, const float *g_uv, const float *g_z,
                       const float *g_xyz, float *dq, float *dt, void *stream);
/* i2p_unpool_k and i2p_bn_act_bwd_stats of its dense result in one launch (device library): g f32 [groups,c], arg u8 [groups,c],
 * y f32 [groups*K,c] (the pre-BN tensor the maximum was taken over) -> gd f32 [groups*K,c], dsums += {sum gz, sum gz*xhat}. */
int i2p_unpool_k_stats(long long groups, int K, int c, const float *g, const unsigned char *arg, const float *y, const float *mean_invstd,
                       const float *gamma, const float *beta, float slope, float *gd, double *dsums, void *stream);
int i2p_row_valid(long long rows, int c, const float *x, float *out, void *stream);
/* out [B,3,3] = inverse of the intrinsic matrices K [B,3,3] rescaled to a feature map (row 0 entries fx, cx times sx; row 1 entries fy, cy
 * times sy): change_intrinsic (modellearn_proj_center.py:457-463) + torch.inverse (:282) in one launch. */
int i2p_intrinsic_inverse(int B, const float *K, float sx, float sy, float *out, void *stream);
int i2p_mask_fill(long long rows, int c, const float *x, const float *valid, float fill, float *out, void *stream);
int i2p_pad_cols(int rows, int c, int cpad, const float *w, float *out, void *stream);
int i2p_strided_pick2(int B, int H, int W, int oh, int ow, int sh, int sw, const float *a, const float *b, float *oa, float *ob,
                      void *stream);

/* Device-side build of a batch of KITTI odometry samples (csrc/loader_build.hip; host half: i2pnet_amd/data.py) — the per-point and
 * per-pixel work of the reference loader's __getitem__ (src/kitti_odometry_corr_lidarnone_proj.py:524-533 shuffle, :332-343 jitter,
 * :654-656 float64 extrinsic product, :699-711 zero padding, :713-747 top-row drop + cv2.resize x0.5 + crop) in one launch each per
 * BATCH.  `table`: device array of B rows —
 *   points: 128 bytes = { const float *scan [N,4]; const int64 *perm [N]; int64 n = min(N, sample_point); int64 0; double E[12] }
 *   image :  64 bytes = { const uint8 *img [H,W,3] (top rows dropped); int64 H, W, oh, ow (resized size), dx, dy (crop), exact2 }
 * noise: [B, sample_point, 3] standard-normal draws or NULL (no jitter); lidar / raw [B, sample_point, 3], feats [B, sample_point, 1]
 * (rows >= n are written as zeros); rgb [B, 3, out_h, out_w] float in 0..255.  exact2 != 0 (H == 2 oh, W == 2 ow): the 2x2 mean
 * rounded half up (cv2's INTER_AREA fast path for an exact 2x shrink); otherwise cv2's 8-bit bilinear rule (11-bit weights). */
int i2p_kitti_points_build(int B, int sample_point, const void *table, const float *noise, float *lidar, float *raw, float *feats,
                           void *stream);
int i2p_kitti_image_build(int B, int out_h, int out_w, const void *table, float *rgb, void *stream);

/* Measurement hook (bench.py's roofline object): with i2p_ktime_enable(1) the launchers of wreg_bwd_fused_kernel (fp32) and
 * bwd_fused_bf16_kernel bracket the kernel alone with HIP events on the launch stream; i2p_ktime_last_us() waits for the last
 * bracketed launch and returns its duration in microseconds (-1 if none).  Not for use inside a stream capture. */
int i2p_ktime_enable(int on);
float i2p_ktime_last_us(void);

#ifdef __cplusplus
}
#endif
#endif /* I2P_OPS_H_ */
