// First block of the image encoder — Conv2d(3 -> 16, 3x3, pad 1) + BatchNorm2d(batch statistics) + LeakyReLU + MaxPool2d(3, stride, 1)
// (src/modules/basicConv.py:6-20, the first (conv, bn, act, pool) group of RGB_net1) — WITHOUT ever writing the conv output.
//
// At BASELINE.json configs[1] that output is 8 x 375 x 1242 x 16 fp32 = 238 MB, the largest tensor of the step, and the round-3/4
// path moved it six times (MIOpen zero-fill + igemm forward, statistics, pooling; BN-backward statistics, dy written, dy read by
// MIOpen's weight-gradient kernel): 707 us of a 11.9 ms step.  With 3 input channels the convolution is 27 MACs per output value, so
// recomputing it from the 45 MB input is cheaper than one pass over its result, and two facts remove every dense pass of the backward:
//   * the input is not differentiated, so dL/dy is only consumed by the weight gradient dW[co][t] = sum_pos dy[pos][co] * xwin[pos][t]
//     (xwin[pos] = the 27 zero-padded window values at pos, t = ci*9 + kh*3 + kw);
//   * BN backward is dy = A (gy - mean(gy) - xhat * mean(gy * xhat)) with gy = the un-pooled gradient, non-zero only at the arg-max
//     positions.  The sparse part is a sum over POOLED elements; the two dense parts are sum_pos xwin[pos] = s (27 numbers) and
//     sum_pos y[pos][co] * xwin[pos][t] = (W S)[co][t] with S = sum_pos xwin xwin^T, the 27 x 27 Gram matrix of the input windows,
//     which depends on the images only.  S also gives the batch statistics: sum y = W s, sum y^2 = diag(W S W^T).
// Forward = Gram pass over x (fp32 MFMA, fp64 folding) + a 1-block coefficient kernel + conv/BN/act/pool in one kernel (x tile in LDS,
// 16x16x4 MFMAs: weights as the A operand, window values as B, so a lane ends with 4 consecutive channels of one pixel = the NHWC vector).
// Backward = one pass over (gout, arg, x) + a 1-block finalisation.  HBM traffic per step: 2 x 45 MB + 75 MB forward, 45 + 75 MB backward.
#include "common.h"

namespace {

constexpr int THREADS = 256;
constexpr int REP = I2P_BN_REPLICAS;
constexpr int CO = 16, NTAP = 27;           // taps t = ci * 9 + kh * 3 + kw: the weight tensor's own [16][3][3][3] order
constexpr int GS = 32;                      // the Gram matrix is kept as [32][32] doubles: rows / columns 0..26 taps, 27 the constant 1
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

struct XView { const float *p; long long sb, sc, sh, sw; };       // x [B,3,H,W] by element strides (NCHW or channels_last)
struct WView {                                                    // weight [16,3,3,3] by element strides (contiguous or channels_last)
    const float *p; int s0, s1, s2, s3;
    __device__ __forceinline__ int at(int co, int t) const { const int c = t / 9, r = t - c * 9, kh = r / 3; return co * s0 + c * s1 + kh * s2 + (r - kh * 3) * s3; }
    __device__ __forceinline__ float ld(int co, int t) const { return p[at(co, t)]; }
};

__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {            // v_cvt_pk_bf16_f32 (RNE)
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
}

// xs[(c * RH + r) * PITCH + cc] = x[b][c][h_org + r][w_org + cc], zero outside the image (the convolution's padding)
template <int RH, int RW, int PITCH>
__device__ __forceinline__ void stage_x(float *xs, const XView &x, int b, int h_org, int w_org, int H, int W) {
    for (int e = threadIdx.x; e < 3 * RH * RW; e += THREADS) {
        const int c = e / (RH * RW), rem = e - c * (RH * RW), r = rem / RW, cc = rem - r * RW;
        const int h = h_org + r, w = w_org + cc;
        float v = 0.f;
        if (h >= 0 && h < H && w >= 0 && w < W) v = x.p[b * x.sb + c * x.sc + h * x.sh + w * x.sw];
        xs[(c * RH + r) * PITCH + cc] = v;
    }
}
template <int RH, int PITCH> __device__ __forceinline__ int tap_off(int t) {   // LDS offset of tap t relative to the window's corner
    const int c = t / 9, r = t - c * 9, dy = r / 3;
    return (c * RH + dy) * PITCH + (r - dy * 3);
}

// ---- Gram matrix of the input windows ----------------------------------------------------------------------------------------------
// S' = X'^T X' over all B*H*W positions, X'[pos] = (27 window values, 1): a [28 x P] x [P x 28] product with the positions as the MFMA
// contraction index (4 per v_mfma_f32_16x16x4_f32), the three tiles (0,0), (0,1), (1,1) of the symmetric result per step.  A wave owns
// 64 consecutive positions of an image row at a time; the contraction order is free, so step st takes the positions kq*16 + st
// (kq = lane >> 4): a lane's operand values over the 16 steps are 16 CONSECUTIVE floats of one input row — four unaligned 16-byte
// loads per operand straight from global memory, no LDS.  Rows / columns at the image border take the element-wise path (EDGE).
// fp32 accumulation over the 64 positions of a row segment, folded into fp64 registers; block sums -> one of REP fp64 replicas.
constexpr int GROWS = 5, GSEG = 64, GTHREADS = 512;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// One row segment.  MODE 0: every window value of the segment is inside the image (wide loads, no masks); MODE 1: the segment touches
// the left / right image border or is the partial last one (wide loads + per-element masks); MODE 2: first / last image row
// (element-wise).  The wide loads are buffer loads through the image's descriptor.
template <int MODE>
__device__ __forceinline__ void gram_load(__amdgpu_buffer_rsrc_t rsrc, const float *__restrict__ img, long long sc, long long sh, int H, int W,
                                          int r, int w0, int kq, int c0, int dy0, int dx0, bool real1, int c1, int dy1, int dx1, float const1,
                                          float (&v0)[16], float (&v1)[16]) {
    const int colb = w0 + kq * 16, lo0 = colb + dx0 - 1, lo1 = colb + dx1 - 1;
    if constexpr (MODE < 2) {
        // MODE 0: lane offset + scalar row offset.  MODE 1: the row offset goes into the lane offset, so that the descriptor's range
        // check sees the whole address (columns past W in the last rows of the last image are past the tensor), and a lane whose
        // first column is -1 loads from column 0 and shifts (nothing is read in front of the image).
        const int row4 = (r - 1) * (int)sh * 4, soff = MODE == 0 ? row4 : 0, voff = MODE == 0 ? 0 : row4;
        const int adj0 = (MODE == 1 && lo0 < 0) ? 1 : 0, adj1 = (MODE == 1 && lo1 < 0) ? 1 : 0;
        const int b0 = (int)(c0 * sc + dy0 * sh + lo0 + adj0) * 4 + voff, b1 = (int)(c1 * sc + dy1 * sh + lo1 + adj1) * 4 + voff;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32x4 t0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, b0 + 16 * q, soff, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) v0[4 * q + e] = __uint_as_float(t0[e]);
        }
        if (real1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4 t1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, b1 + 16 * q, soff, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) v1[4 * q + e] = __uint_as_float(t1[e]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) v1[e] = const1;
        }
        if constexpr (MODE == 1) {
#pragma unroll
            for (int e = 15; e >= 0; --e) {
                const bool ok = colb + e < W;
                const float s0 = e > 0 ? v0[e - 1] : 0.f, s1 = e > 0 ? v1[e - 1] : 0.f;
                const float t0 = adj0 ? s0 : v0[e], t1 = (adj1 && real1) ? s1 : v1[e];
                v0[e] = (ok && lo0 + e >= 0 && lo0 + e < W) ? t0 : 0.f;
                v1[e] = (ok && (!real1 || (lo1 + e >= 0 && lo1 + e < W))) ? t1 : 0.f;
            }
        }
    } else {
        const int r0 = r + dy0 - 1, r1 = r + dy1 - 1;
        const bool row0 = r0 >= 0 && r0 < H, row1 = r1 >= 0 && r1 < H;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int col = colb + e, q0 = lo0 + e, q1 = lo1 + e;
            const bool ok = col < W;
            float t0 = 0.f, t1 = const1;
            if (ok && row0 && q0 >= 0 && q0 < W) t0 = img[c0 * sc + (long long)r0 * sh + q0];
            if (real1) { t1 = 0.f; if (ok && row1 && q1 >= 0 && q1 < W) t1 = img[c1 * sc + (long long)r1 * sh + q1]; }
            v0[e] = ok ? t0 : 0.f;
            v1[e] = ok ? t1 : 0.f;
        }
    }
}

__global__ __launch_bounds__(GTHREADS) void img1_gram_kernel(XView x, int B, int H, int W, int chunks_h, int segs_w, double *__restrict__ gram) {
    __shared__ double red[GTHREADS / 64][768];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), i = lane & 15, kq = lane >> 4;
    const int t1 = 16 + i;
    const int c0 = i / 9, dy0 = (i - c0 * 9) / 3, dx0 = i - c0 * 9 - dy0 * 3;
    const bool real1 = t1 < NTAP;
    const int tt = real1 ? t1 : 0, c1 = tt / 9, dy1 = (tt - c1 * 9) / 3, dx1 = tt - c1 * 9 - dy1 * 3;
    const float const1 = t1 == NTAP ? 1.f : 0.f;
    double d00[4], d01[4], d11[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { d00[r] = 0.0; d01[r] = 0.0; d11[r] = 0.0; }
    const unsigned nunits = (unsigned)(B * chunks_h * segs_w), nwaves = gridDim.x * (GTHREADS / 64);
    for (unsigned unit = i2p_xcd_swizzle(blockIdx.x, gridDim.x) * (GTHREADS / 64) + wv; unit < nunits; unit += nwaves) {
        const unsigned bh = unit / (unsigned)segs_w;
        const int sg = (int)(unit - bh * (unsigned)segs_w), b = (int)(bh / (unsigned)chunks_h), ch = (int)(bh - (unsigned)b * (unsigned)chunks_h);
        const int h0 = ch * GROWS, w0 = sg * GSEG;
        const float *img = x.p + b * x.sb;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(img), 0,
                                                                              (int)((2 * x.sc + (long long)(H - 1) * x.sh + W) * 4), 0x00020000);
        const bool cols_inside = w0 >= 1 && w0 + GSEG + 1 <= W;
        const int nrows = min(GROWS, H - h0);
        auto load = [&](int r, float (&v0)[16], float (&v1)[16]) {
            if (r < 1 || r + 1 >= H) gram_load<2>(rsrc, img, x.sc, x.sh, H, W, r, w0, kq, c0, dy0, dx0, real1, c1, dy1, dx1, const1, v0, v1);
            else if (cols_inside) gram_load<0>(rsrc, img, x.sc, x.sh, H, W, r, w0, kq, c0, dy0, dx0, real1, c1, dy1, dx1, const1, v0, v1);
            else gram_load<1>(rsrc, img, x.sc, x.sh, H, W, r, w0, kq, c0, dy0, dx0, real1, c1, dy1, dx1, const1, v0, v1);
        };
        // 48 MFMAs of one row segment: fp32 accumulation inside the segment only (16 steps), then fp64 — the batch statistics are read
        // off these sums
        auto mma = [&](const float (&v0)[16], const float (&v1)[16]) {
            f32x4 a00 = {0.f, 0.f, 0.f, 0.f}, a01 = a00, a11 = a00;
            __builtin_amdgcn_s_setprio(1);              // (a wave in its MFMA cluster before one that is issuing loads)
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                a00 = __builtin_amdgcn_mfma_f32_16x16x4f32(v0[st], v0[st], a00, 0, 0, 0);
                a01 = __builtin_amdgcn_mfma_f32_16x16x4f32(v0[st], v1[st], a01, 0, 0, 0);
                a11 = __builtin_amdgcn_mfma_f32_16x16x4f32(v1[st], v1[st], a11, 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) { d00[q] += (double)a00[q]; d01[q] += (double)a01[q]; d11[q] += (double)a11[q]; }
        };
        // two row buffers in turn: the next row's loads are in flight under this row's MFMAs
        float va[16], vb[16], na[16], nb[16];
        load(h0, va, vb);
        int rr = 0;
        for (; rr + 1 < nrows; rr += 2) {
            load(h0 + rr + 1, na, nb);
            mma(va, vb);
            if (rr + 2 < nrows) load(h0 + rr + 2, va, vb);
            mma(na, nb);
        }
        if (rr < nrows) mma(va, vb);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[wv][r * 64 + lane] = d00[r];
        red[wv][256 + r * 64 + lane] = d01[r];
        red[wv][512 + r * 64 + lane] = d11[r];
    }
    __syncthreads();
    double *rep = gram + (size_t)(blockIdx.x % REP) * (GS * GS);
    for (int e = threadIdx.x; e < 768; e += GTHREADS) {
        const int ts = e >> 8, r = (e >> 6) & 3, ln = e & 63;
        const int row = 4 * (ln >> 4) + r + (ts == 2 ? 16 : 0), col = (ln & 15) + (ts >= 1 ? 16 : 0);   // D[4 * (lane >> 4) + r][lane & 15]
        if (row > NTAP || col > NTAP) continue;
        double t = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < GTHREADS / 64; ++w2) t += red[w2][e];
        atomicAdd(rep + row * GS + col, t);
    }
}

// replica sums -> gram_red [32][32] (the lower-left tile by symmetry), and from it the batch statistics of the conv output:
// sum y = W s, sum y^2 = w S w^T per channel (fp64); mean_invstd and the running buffers as img_pool_fwd2_kernel writes them.
__global__ __launch_bounds__(1024) void img1_coef_kernel(const double *__restrict__ gram, WView wgt, long long n_pos, float eps,
                                                         float momentum, const float *__restrict__ conv_bias, float *__restrict__ running_mean,
                                                         float *__restrict__ running_var, double *__restrict__ gram_red,
                                                         float *__restrict__ mean_invstd) {
    __shared__ double S[GS * GS];
    const int e = threadIdx.x, i = e >> 5, j = e & 31;
    const int src = (i >= 16 && j < 16) ? j * GS + i : e;
    double v = 0.0;
    for (int r = 0; r < REP; ++r) v += gram[(size_t)r * (GS * GS) + src];
    S[e] = v;
    gram_red[e] = v;
    __syncthreads();
    // U[co][t] = sum_u w[co][u] S[u][t] on 16 x 27 threads, then sum y = w . s and sum y^2 = w . U[co] per channel
    __shared__ double U[CO * NTAP];
    if (e < CO * NTAP) {
        const int co = e / NTAP, t = e - co * NTAP;
        double a = 0.0;
        for (int u = 0; u < NTAP; ++u) a += (double)wgt.ld(co, u) * S[u * GS + t];
        U[e] = a;
    }
    __syncthreads();
    if (e < CO) {
        double sy = 0.0, sq = 0.0;
        for (int t = 0; t < NTAP; ++t) {
            const double w = (double)wgt.ld(e, t);
            sy += w * S[t * GS + NTAP];
            sq += w * U[e * NTAP + t];
        }
        const double n = (double)n_pos, m = sy / n;
        double var = sq / n - m * m;
        var = var < 0.0 ? 0.0 : var;
        mean_invstd[e] = (float)m;
        mean_invstd[CO + e] = rsqrtf((float)var + eps);
        if (running_mean) {
            const float mb = (float)m + (conv_bias ? conv_bias[e] : 0.f);
            running_mean[e] = (1.f - momentum) * running_mean[e] + momentum * mb;
            const float unbiased = (float)(var * (n / (n > 1.0 ? n - 1.0 : 1.0)));
            running_var[e] = (1.f - momentum) * running_var[e] + momentum * unbiased;
        }
    }
}

// ---- forward: conv -> BN -> LeakyReLU -> max-pool, wave-autonomous -------------------------------------------------------------------
// A wave owns a strip of 16 conv columns (cw = wo_a S - 1 + j, j = lane & 15) and walks down RC pooled rows.  The input enters through
// ONE load per image row: lane L < 54 holds x[c = L / 18][row][cw_0 - 1 + L % 18] (0 outside the image: the descriptor's range check),
// three such row registers cover a conv row's windows and rotate from row to row (two more rows are in flight).  The MFMA B operand
// [tap][position] is gathered from them with ds_bpermute (the LDS crossbar, no memory): lane (j, kq) of step st takes tap (dy, u) with
// dy = st / 2, u = (st & 1) 4 + kq = c 3 + dx for the first six steps (one source register per step), the leftover tap u = 8 of the
// three rows shares step 6 (kq = dy; kq = 3 pads K to 28 with weight 0).  7 MFMAs (weights = A operand [co][tap]) leave channels
// 4 kq .. 4 kq + 3 of position j in the lane; BN + LeakyReLU, positions outside the image = -inf (the pooling's padding); the row's first
// maximum over the three window columns comes from the neighbour lanes (DPP row shifts), the window's over the three row results as in
// img_pool_fwd2_kernel: first maximum in (kh, kw) scan order, NaN propagating to the last NaN.  Lanes j = 1, 3, .., 13 (stride 2;
// 1 .. 14 at stride 1) are window centres and store; no LDS allocation, no barrier.
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// max_pool2d's update rule "a > best || isnan(a)" = !(a <= best), as two selects (no branch)
__device__ __forceinline__ void take_if_greater(float a, unsigned k, float &best, unsigned &bi) {
    const bool up = !(a <= best);
    best = up ? a : best;
    bi = up ? k : bi;
}
__device__ __forceinline__ float bperm(int byte_addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v))); }

template <int S, bool OBF>
__global__ __launch_bounds__(THREADS) void img1_fwd_kernel(XView x, int B, int H, int W, int Ho, int Wo, int chunks_h, int strips_w,
                                                           WView wgt, const float *__restrict__ mean_invstd,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta, float slope,
                                                           void *__restrict__ out, unsigned char *__restrict__ arg) {
    constexpr int NPW = S == 2 ? 7 : 14, RC = 8;
    const int lane = threadIdx.x & 63, j = lane & 15, kq = lane >> 4;
    const unsigned nunits = (unsigned)(B * chunks_h * strips_w);
    // (the wave index is uniform: say so, and the unit's descriptor and row offsets live in SGPRs)
    const unsigned unit = i2p_xcd_swizzle(blockIdx.x, gridDim.x) * (THREADS / 64) + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (unit >= nunits) return;
    const unsigned bh = unit / (unsigned)strips_w;
    const int strip = (int)(unit - bh * (unsigned)strips_w), b = (int)(bh / (unsigned)chunks_h), ch = (int)(bh - (unsigned)b * (unsigned)chunks_h);
    const int wo_a = strip * NPW, ho_a = ch * RC, ho_e = min(Ho, ho_a + RC);
    const int cw = wo_a * S - 1 + j;
    const bool col_in = cw >= 0 && cw < W;
    float wr[7];
    int bp[7];
#pragma unroll
    for (int st = 0; st < 7; ++st) {
        const int dy = st < 6 ? st >> 1 : (kq < 3 ? kq : 2), u = st < 6 ? (st & 1) * 4 + kq : 8, c = u / 3, dx = u - c * 3;
        wr[st] = (st < 6 || kq < 3) ? wgt.ld(j, c * 9 + dy * 3 + dx) : 0.f;
        bp[st] = 4 * (c * 18 + j + dx);
    }
    // z = fma(y, scale, shift), shift = beta - mean scale (the backward kernel forms z the same way); a lane whose conv column is
    // outside the image gets scale 0 and shift -inf: its activation is the pooling's padding without a select
    float scale[4], zshift[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = 4 * kq + r;
        const float sc_ = mean_invstd[CO + c] * gamma[c];
        scale[r] = col_in ? sc_ : 0.f;
        zshift[r] = col_in ? beta[c] - mean_invstd[c] * sc_ : -INFINITY;
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x.p + b * x.sb), 0,
                                                                          (int)((2 * x.sc + (long long)(H - 1) * x.sh + W) * 4), 0x00020000);
    const int lc = lane / 18, xcol = wo_a * S - 2 + (lane - lc * 18);
    const int lvoff = (lane < 54 && xcol >= 0 && xcol < W) ? (int)(lc * x.sc + xcol) * 4 : 0x7fffffff;      // out of range reads 0
    const int sh4 = (int)x.sh * 4;
    auto load_x = [&](int xr) -> float {
        if (xr < 0 || xr >= H) return 0.f;
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, lvoff, xr * sh4, 0));
    };
    const int r_first = ho_a * S - 1;
    float R0 = load_x(r_first - 1), R1 = load_x(r_first), R2 = load_x(r_first + 1), Rn = load_x(r_first + 2), Rnn;
    // one conv row: its activations, then (value, column index) of the first maximum over the three window columns of every lane
    auto row = [&](int r, float (&v)[4], unsigned (&aw)[4]) {
        Rnn = load_x(r + 3);
        if (r < 0 || r >= H) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { v[c] = -INFINITY; aw[c] = 0u; }
        } else {
            float xv[7];
            xv[0] = bperm(bp[0], R0); xv[1] = bperm(bp[1], R0);
            xv[2] = bperm(bp[2], R1); xv[3] = bperm(bp[3], R1);
            xv[4] = bperm(bp[4], R2); xv[5] = bperm(bp[5], R2);
            const float m0 = bperm(bp[6], R0), m1 = bperm(bp[6], R1), m2 = bperm(bp[6], R2);
            xv[6] = kq == 0 ? m0 : (kq == 1 ? m1 : m2);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 7; ++st) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[st], xv[st], acc, 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float z = __fmaf_rn(acc[c], scale[c], zshift[c]);
                const float a = fmaxf(z, z * slope);                                   // LeakyReLU for 0 <= slope <= 1 (first_ok)
                const float left = dpp_f32<0x111>(a), right = dpp_f32<0x101>(a);      // row_shr:1 = lane j - 1, row_shl:1 = lane j + 1
                float best = left;
                unsigned bi = 0u;
                take_if_greater(a, 1u, best, bi);
                take_if_greater(right, 2u, best, bi);
                v[c] = best; aw[c] = bi;
            }
        }
        R0 = R1; R1 = R2; R2 = Rn; Rn = Rnn;
    };
    const bool centre = S == 2 ? ((j & 1) && j < 15) : (j >= 1 && j <= 14);
    const int wo = wo_a + (S == 2 ? (j - 1) >> 1 : j - 1);
    const bool writes = centre && wo < Wo;
    float rv[3][4];
    unsigned ra[3][4];
    auto emit = [&](int ho) {
        float best[4];
        unsigned bi[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { best[c] = rv[0][c]; bi[c] = ra[0][c]; }
#pragma unroll
        for (int kh = 1; kh < 3; ++kh)
#pragma unroll
            for (int c = 0; c < 4; ++c) take_if_greater(rv[kh][c], (unsigned)(kh * 3) + ra[kh][c], best[c], bi[c]);
        if (!writes) return;
        const long long t = (((long long)b * Ho + ho) * Wo + wo) * 4 + kq;
        if constexpr (OBF) {
            const v2u o = {pack_bf2(best[0], best[1]), pack_bf2(best[2], best[3])};
            reinterpret_cast<v2u *>(out)[t] = o;
        } else {
            const f32x4 o = {best[0], best[1], best[2], best[3]};
            reinterpret_cast<f32x4 *>(out)[t] = o;
        }
        reinterpret_cast<unsigned *>(arg)[t] = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    };
    auto shift = [&](int dst, int src) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { rv[dst][c] = rv[src][c]; ra[dst][c] = ra[src][c]; }
    };
    if constexpr (S == 2) {
        row(r_first, rv[0], ra[0]);
        for (int ho = ho_a; ho < ho_e; ++ho) {
            row(2 * ho, rv[1], ra[1]);
            row(2 * ho + 1, rv[2], ra[2]);
            emit(ho);
            shift(0, 2);
        }
    } else {
        row(r_first, rv[0], ra[0]);
        row(r_first + 1, rv[1], ra[1]);
        for (int ho = ho_a; ho < ho_e; ++ho) {
            row(ho + 1, rv[2], ra[2]);
            emit(ho);
            shift(0, 1); shift(1, 2);
        }
    }
}

// ---- backward: the sparse sums over pooled elements -------------------------------------------------------------------------------------
// Per pooled element (p, co): pos = its arg-max conv position, xw = the 27 window values at pos (LDS), y = w[co] . xw (recomputed),
// gz = gout * act'(bn(y)); accumulated per lane (co = lane & 15 fixed): T[t] += gz * xw[t], m1 += gz, m2 += gz * xhat.  Block sums -> one
// row of `partials` [grid][16][29] (fp32; the finalisation adds the rows in fp64, fixed order: reproducible run to run).
template <int S> struct BwdGeom {
    static constexpr int BH = 8, BW = 32;
    // LDS pitch = 9 (mod 64): the 64 lanes of a read are 4 neighbouring pooled pixels x 16 channels, their window corners sit at
    // (ah, 2 slot + aw), ah, aw < 3 -> offsets ah XP + (0 .. 8): three disjoint runs of 9 banks, no conflict between different addresses
    static constexpr int XW = (BW - 1) * S + 5, XH = (BH - 1) * S + 5, XP = ((XW + 54) / 64) * 64 + 9;
    static constexpr int ITEMS = BH * BW / 16;          // pooled pixels per thread (16 threads = the 16 channels of one pixel)
};
constexpr int PROW = NTAP + 2;                           // T[27], m1, m2

template <int S, bool GBF>
__global__ __launch_bounds__(THREADS, 4) void img1_bwd_kernel(XView x, int B, int H, int W, int Ho, int Wo, int tiles_h, int tiles_w,
                                                              WView wgt, const float *__restrict__ mean_invstd,
                                                              const float *__restrict__ gamma, const float *__restrict__ beta, float slope,
                                                              const void *__restrict__ gout, const unsigned char *__restrict__ arg,
                                                              float *__restrict__ partials) {
    using G = BwdGeom<S>;
    __shared__ __attribute__((aligned(16))) float xs[3 * G::XH * G::XP];
    __shared__ float red[4][CO][PROW];
    const int co = threadIdx.x & 15, slot = threadIdx.x >> 4;
    // taps in pairs: v_pk_fma_f32 does two of the element's 2 x 27 multiply-adds per instruction (the kernel is VALU-issue bound)
    constexpr int NP = NTAP / 2;
    f32x2 w2[NP], T2[NP];
    float wl = wgt.ld(co, NTAP - 1), Tl = 0.f, m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int p2 = 0; p2 < NP; ++p2) { w2[p2] = f32x2{wgt.ld(co, 2 * p2), wgt.ld(co, 2 * p2 + 1)}; T2[p2] = f32x2{0.f, 0.f}; }
    const float mean = mean_invstd[co], invstd = mean_invstd[CO + co], scale = invstd * gamma[co], shift = beta[co] - mean * scale;
    const unsigned ntiles = (unsigned)(B * tiles_h * tiles_w);
    for (unsigned tile = i2p_xcd_swizzle(blockIdx.x, gridDim.x); tile < ntiles; tile += gridDim.x) {
        const unsigned bh = tile / (unsigned)tiles_w;
        const int tw = (int)(tile - bh * (unsigned)tiles_w), b = (int)(bh / (unsigned)tiles_h), th = (int)(bh - (unsigned)b * (unsigned)tiles_h);
        const int ho0 = th * G::BH, wo0 = tw * G::BW;
        float gv[G::ITEMS];
        unsigned pos2[G::ITEMS / 2];                 // two 16-bit LDS offsets per register (the element prefetch is what costs registers)
#pragma unroll
        for (int k = 0; k < G::ITEMS / 2; ++k) pos2[k] = 0u;
#pragma unroll
        for (int k = 0; k < G::ITEMS; ++k) {
            const int pp = k * 16 + slot, hl = pp / G::BW, wl = pp - hl * G::BW;
            const int ho = ho0 + hl, wo = wo0 + wl;
            gv[k] = 0.f;
            if (ho < Ho && wo < Wo) {
                const long long e = (((long long)b * Ho + ho) * Wo + wo) * CO + co;
                if constexpr (GBF) gv[k] = __uint_as_float((unsigned)reinterpret_cast<const unsigned short *>(gout)[e] << 16);
                else gv[k] = reinterpret_cast<const float *>(gout)[e];
                const int av = (int)arg[e], ah = av / 3;
                pos2[k >> 1] |= (unsigned)((hl * S + ah) * G::XP + wl * S + (av - ah * 3)) << (16 * (k & 1));
            }
        }
        __syncthreads();
        stage_x<G::XH, G::XW, G::XP>(xs, x, b, ho0 * S - 2, wo0 * S - 2, H, W);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < G::ITEMS; ++k) {
            const float *base = xs + ((pos2[k >> 1] >> (16 * (k & 1))) & 0xffffu);
            f32x2 xw2[NP];
#pragma unroll
            for (int p2 = 0; p2 < NP; ++p2) xw2[p2] = f32x2{base[tap_off<G::XH, G::XP>(2 * p2)], base[tap_off<G::XH, G::XP>(2 * p2 + 1)]};
            const float xl = base[tap_off<G::XH, G::XP>(NTAP - 1)];
            f32x2 ys = {0.f, 0.f};
#pragma unroll
            for (int p2 = 0; p2 < NP; ++p2) ys = __builtin_elementwise_fma(w2[p2], xw2[p2], ys);
            const float y = __fmaf_rn(wl, xl, ys.x + ys.y);
            const float z = __fmaf_rn(y, scale, shift);
            const float gz = z > 0.f ? gv[k] : gv[k] * slope;
            const float xh = (y - mean) * invstd;
            m1 += gz;
            m2 = __fmaf_rn(gz, xh, m2);
            const f32x2 gz2 = {gz, gz};
#pragma unroll
            for (int p2 = 0; p2 < NP; ++p2) T2[p2] = __builtin_elementwise_fma(gz2, xw2[p2], T2[p2]);
            Tl = __fmaf_rn(gz, xl, Tl);
            __builtin_amdgcn_sched_barrier(0);          // one element's 27 window values live at a time (the unrolled loop spilled otherwise)
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    auto wave_sum = [](float v) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); return v; };
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
        const float v = wave_sum(t == NTAP - 1 ? Tl : (t & 1 ? T2[t >> 1].y : T2[t >> 1].x));
        if (lane < 16) red[wv][lane][t] = v;
    }
    m1 = wave_sum(m1); m2 = wave_sum(m2);
    if (lane < 16) { red[wv][lane][NTAP] = m1; red[wv][lane][NTAP + 1] = m2; }
    __syncthreads();
    for (int e = threadIdx.x; e < CO * PROW; e += THREADS) {
        const float *r0 = &red[0][0][0];
        partials[(size_t)blockIdx.x * (CO * PROW) + e] = (r0[e] + r0[CO * PROW + e]) + (r0[2 * CO * PROW + e] + r0[3 * CO * PROW + e]);
    }
}

// dW[co][t] = A (T1 - mean(gy) s[t] - mean(gy xhat) invstd ((W S)[co][t] - mean s[t])), A = gamma invstd; dgamma = sum gy xhat, dbeta = sum gy.
// One block per channel: its 29 columns of `partials` summed over the rows by 32 thread groups (fp64, fixed order), then the 27 + 2 results.
__global__ __launch_bounds__(1024) void img1_bwd_fin_kernel(int nblk, const float *__restrict__ partials, const double *__restrict__ gram_red,
                                                            WView wgt, const float *__restrict__ mean_invstd,
                                                            const float *__restrict__ gamma, long long n_pos, float *__restrict__ dW,
                                                            float *__restrict__ dgamma, float *__restrict__ dbeta) {
    __shared__ double part[32][32], R[PROW];
    const int co = blockIdx.x, k = threadIdx.x & 31, grp = threadIdx.x >> 5;
    double a = 0.0;
    if (k < PROW) {
#pragma unroll 8
        for (int b = grp; b < nblk; b += 32) a += (double)partials[(size_t)b * (CO * PROW) + co * PROW + k];
    }
    part[grp][k] = a;
    __syncthreads();
    if (threadIdx.x < PROW) {
        double t = 0.0;
        for (int g2 = 0; g2 < 32; ++g2) t += part[g2][threadIdx.x];
        R[threadIdx.x] = t;
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < NTAP) {
        const double n = (double)n_pos, mean = (double)mean_invstd[co], invstd = (double)mean_invstd[CO + co];
        const double mg = R[NTAP] / n, mgx = R[NTAP + 1] / n;
        double U = 0.0;
        for (int u = 0; u < NTAP; ++u) U += (double)wgt.ld(co, u) * gram_red[u * GS + t];
        const double st = gram_red[t * GS + NTAP];
        dW[wgt.at(co, t)] = (float)((double)gamma[co] * invstd * (R[t] - mg * st - mgx * invstd * (U - mean * st)));
    } else if (t == NTAP) {
        dbeta[co] = (float)R[NTAP];
        dgamma[co] = (float)R[NTAP + 1];
    }
}

bool first_ok(int B, int H, int W, int s, float slope = 0.f) {
    return B >= 0 && H > 0 && W > 0 && (s == 1 || s == 2) && (long long)B * H * W < (1ll << 31) && slope >= 0.f && slope <= 1.f;
}
unsigned round8(long long v) { return (unsigned)((v + 7) & ~7ll); }
int env_int(const char *name, int dflt) { const char *e = getenv(name); return e && e[0] ? atoi(e) : dflt; }

}  // namespace

// Number of `partials` rows i2p_img_first_bwd writes (rows of 16 * 29 floats): the grid of its pass kernel.
extern "C" int i2p_img_first_bwd_rows(int B, int H, int W, int stride) {
    if (!first_ok(B, H, W, stride) || B == 0) return 0;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long long tiles = (long long)B * ((Ho + 7) / 8) * ((Wo + 31) / 32);
    static const int cap = env_int("I2P_IMG1_BWD_GRID", 1024);
    return (int)round8(tiles < cap ? tiles : cap);
}

// parts: 1 = the statistics (gram, gram_red, mean_invstd, running buffers), 2 = the block's output from them, 3 = both — the statistics
// depend on the images and the weights only, so a caller can issue them early on another stream.
// x [B,3,H,W] fp32 by element strides, wgt [16,3,3,3] by element strides ws[4] (dW is written in the same layout); gram: f64 [I2P_BN_REPLICAS][1024] zeroed by the caller;
// -> gram_red f64 [1024] (kept for the backward), out [B,Ho,Wo,16] fp32 / bf16, arg u8 [B,Ho,Wo,16], mean_invstd f32 [32]
extern "C" int i2p_img_first_fwd(int B, int H, int W, int stride, const float *x, long long sb, long long sc, long long sh, long long sw,
                                 const float *wgt, const int *ws, const float *gamma, const float *beta, float eps, float slope, float momentum,
                                 const float *conv_bias, float *running_mean, float *running_var, double *gram, double *gram_red, int out_bf16,
                                 void *out, unsigned char *arg, float *mean_invstd, int parts, void *stream) {
    if (!first_ok(B, H, W, stride, slope) || !(parts & 3)) return I2P_ERR_BAD_ARG;
    if (B == 0) return 0;
    if (!x || !wgt || !ws || !mean_invstd) return I2P_ERR_BAD_ARG;
    if ((parts & 1) && (!gram || !gram_red)) return I2P_ERR_BAD_ARG;
    if ((parts & 2) && (!gamma || !beta || !out || !arg)) return I2P_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const XView xv{x, sb, sc, sh, sw};
    const WView wv{wgt, ws[0], ws[1], ws[2], ws[3]};
    // rows are read with vector loads and 32-bit BYTE offsets inside an image, formed from the caller's strides: the farthest byte
    // the kernels address is (2 sc + (H + 1) sh + W) * 4 (a channel / row slice of a larger tensor can have large strides)
    if (sw != 1 || sc < 0 || sh < 0 || 3ll * H * W >= (1ll << 31) || (2 * sc + (long long)(H + 1) * sh + W) * 4 >= (1ll << 31)) return I2P_ERR_BAD_ARG;
    if (parts & 1) {        // the statistics: Gram matrix of the input windows, then mean / invstd of the conv output and the running buffers
        const int ch = (H + GROWS - 1) / GROWS, sg = (W + GSEG - 1) / GSEG;
        static const int grid = env_int("I2P_IMG1_GRAM_GRID", 256);
        // (48 KB static + 40 KB dynamic LDS: one block per CU, the 256-block grid cannot double up on a CU)
        static const unsigned pad = (unsigned)env_int("I2P_IMG1_GRAM_LDS_PAD", 40 * 1024);
        hipLaunchKernelGGL(img1_gram_kernel, dim3(round8(grid)), dim3(GTHREADS), pad, st, xv, B, H, W, ch, sg, gram);
        hipLaunchKernelGGL(img1_coef_kernel, dim3(1), dim3(1024), 0, st, (const double *)gram, wv, (long long)B * H * W, eps, momentum, conv_bias,
                           running_mean, running_var, gram_red, mean_invstd);
    }
    if (!(parts & 2)) I2P_RETURN_LAUNCH_STATUS();
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const int th = (Ho + 7) / 8, tw = (Wo + (stride == 2 ? 7 : 14) - 1) / (stride == 2 ? 7 : 14);          // RC = 8 pooled rows x NPW pooled columns per wave
    const dim3 grid(round8(((long long)B * th * tw + 3) / 4));
#define I2P_FWD1(SS, BF) hipLaunchKernelGGL((img1_fwd_kernel<SS, BF>), grid, dim3(THREADS), 0, st, xv, B, H, W, Ho, Wo, th, tw, wv, \
                                            (const float *)mean_invstd, gamma, beta, slope, out, arg)
    if (stride == 2) { if (out_bf16) I2P_FWD1(2, true); else I2P_FWD1(2, false); }
    else { if (out_bf16) I2P_FWD1(1, true); else I2P_FWD1(1, false); }
#undef I2P_FWD1
    I2P_RETURN_LAUNCH_STATUS();
}

// gout [B,Ho,Wo,16] fp32 / bf16 (g_bf16), arg, x, the forward's mean_invstd and gram_red -> dW (27*16 floats in wgt's layout), dgamma, dbeta [16];
// partials: f32 [i2p_img_first_bwd_rows()][16 * 29] scratch
extern "C" int i2p_img_first_bwd(int B, int H, int W, int stride, const float *x, long long sb, long long sc, long long sh, long long sw,
                                 const float *wgt, const int *ws, const float *gamma, const float *beta, float slope, const float *mean_invstd,
                                 const double *gram_red, int g_bf16, const void *gout, const unsigned char *arg, float *partials, float *dW,
                                 float *dgamma, float *dbeta, void *stream) {
    if (!first_ok(B, H, W, stride, slope)) return I2P_ERR_BAD_ARG;
    if (!dW || !dgamma || !dbeta) return I2P_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        (void)hipMemsetAsync(dW, 0, sizeof(float) * CO * NTAP, st);
        (void)hipMemsetAsync(dgamma, 0, sizeof(float) * CO, st);
        (void)hipMemsetAsync(dbeta, 0, sizeof(float) * CO, st);
        I2P_RETURN_LAUNCH_STATUS();
    }
    if (!x || !wgt || !ws || !gamma || !beta || !mean_invstd || !gram_red || !gout || !arg || !partials) return I2P_ERR_BAD_ARG;
    if (sw != 1 || sc < 0 || sh < 0 || (2 * sc + (long long)(H + 1) * sh + W) * 4 >= (1ll << 31)) return I2P_ERR_BAD_ARG;      // (as in the forward entry)
    const XView xv{x, sb, sc, sh, sw};
    const WView wv{wgt, ws[0], ws[1], ws[2], ws[3]};
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const int th = (Ho + 7) / 8, tw = (Wo + 31) / 32;
    const int rows = i2p_img_first_bwd_rows(B, H, W, stride);
    static const unsigned pad = (unsigned)env_int("I2P_IMG1_BWD_LDS_PAD", 16 * 1024);        // 23 + 16 KB: 4 blocks per CU
#define I2P_BWD1(SS, BF) hipLaunchKernelGGL((img1_bwd_kernel<SS, BF>), dim3(rows), dim3(THREADS), pad, st, xv, B, H, W, Ho, Wo, th, tw, wv, \
                                            mean_invstd, gamma, beta, slope, gout, arg, partials)
    if (stride == 2) { if (g_bf16) I2P_BWD1(2, true); else I2P_BWD1(2, false); }
    else { if (g_bf16) I2P_BWD1(1, true); else I2P_BWD1(1, false); }
#undef I2P_BWD1
    hipLaunchKernelGGL(img1_bwd_fin_kernel, dim3(CO), dim3(1024), 0, st, rows, (const float *)partials, gram_red, wv, mean_invstd, gamma,
                       (long long)B * H * W, dW, dgamma, dbeta);
    I2P_RETURN_LAUNCH_STATUS();
}
