// 3x3 convolution (padding 1, stride 1) between NHWC fp32 tensors of 16 or 32 channels — the image encoder's blocks 2-5 at 188 x 621
// (src/modules/basicConv.py:6-20: Conv2d(16, 16, 3, padding=1) x 3 and Conv2d(16, 32, 3, padding=1)): forward, input gradient and
// weight gradient.  MIOpen's implicit-GEMM kernels spend 101 us (forward), 79 us + an 18 us zero-fill (input gradient) and 130 us
// (split-K weight gradient, atomics) on each 16 -> 16 layer at BASELINE configs[1]: 120 MB of traffic and 4.3 GFLOP, 28 % of the fp32
// MFMA rate — 16 channels are a small GEMM.  Forward / input gradient: a wave owns a strip of 16 columns and walks down the rows:
//   * one 16-byte load per lane, input row and 16 channels (1 KB = 16 pixels x 16 channels) into row registers, ALREADY in the layout
//     of the MFMA B operand [k][pixel]: lane (j = lane & 15, kq = lane >> 4) holds channels 4 kq .. 4 kq + 3 of the strip's column j;
//     three rows cover a conv row's windows and rotate, two more are in flight;
//   * step (tap, q, u) of the v_mfma_f32_16x16x4_f32 chain takes channel ci = 16 q + 4 kq + u of column j + kw - 1: component u of the
//     row register itself for the centre column, of the neighbour lane (one DPP row shift, VALU) for the other two — no LDS, no shuffle
//     through memory; the weights (A operand) are ordered to match; two accumulators per output tile alternate;
//   * the MFMAs leave channels 4 kq .. 4 kq + 3 (of each 16-channel tile) of pixel j in the lane = the NHWC vector, stored directly;
//     lanes j = 1 .. 14 own an output column (their windows lie inside the strip's 16 columns);
//   * forward: sum y and sum y^2 of the BatchNorm behind the convolution are accumulated from the registers (fp32 per lane over
//     its ~17 rows, fp64 from there) and added to the replicated fp64 sums the pooling kernel reads — the separate statistics pass over y is gone.
// The input gradient is the same kernel on dL/dy with the weight indices swapped and the taps mirrored.
#include "common.h"

namespace {

constexpr int THREADS = 256;
constexpr int REP = I2P_BN_REPLICAS;
// s_setprio around the MFMA clusters: the waves of these kernels are independent, so at any time some are in their load / epilogue
// phase and some in an MFMA cluster — the scheduler may prefer the latter (-DI2P_NO_MFMA_PRIO: off, for A/B)
#ifndef I2P_NO_MFMA_PRIO
#define I2P_MFMA_PRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define I2P_MFMA_PRIO(p)
#endif
constexpr int C = 16;                       // input channels of the weight-gradient kernel
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {            // v_cvt_pk_bf16_f32 (RNE)
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// W[out][in][kh][kw] by element strides; `flip`: the input-gradient view W'[out = ci][in = co][kh][kw] = W[co][ci][2 - kh][2 - kw]
struct WView16 {
    const float *p; int s_out, s_in, s_kh, s_kw, flip;
    __device__ __forceinline__ float ld(int out, int in, int kh, int kw) const {
        return p[out * s_out + in * s_in + (flip ? 2 - kh : kh) * s_kh + (flip ? 2 - kw : kw) * s_kw];
    }
};

template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// Work = the B x strips x H output rows of all strips in (image, strip, row) order, cut into equal contiguous ranges, one per wave of
// a grid that is resident at once (4 waves per SIMD): every SIMD runs the same number of MFMAs.  A range that crosses a strip
// boundary is walked as two segments.
template <int CIN, int COUT, int MODE>
__global__ __launch_bounds__(THREADS) void conv3x3_kernel(const float *__restrict__ x, int B, int H, int W, int strips_w, WView16 wgt,
                                                          float *__restrict__ y, double *__restrict__ sums) {
    constexpr bool STATS = MODE != 0;
    constexpr int NPW = 14;                                           // output columns per strip
    constexpr int NQ = CIN / 16, NT = COUT / 16, KS = 9 * 4 * NQ;     // 16-channel groups of the input / output, MFMA steps per tile
    struct Row { float q[NQ][4]; };
    const int lane = threadIdx.x & 63, j = lane & 15, kq = lane >> 4;
    const unsigned wave = i2p_xcd_swizzle(blockIdx.x, gridDim.x) * (THREADS / 64) + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned nwaves = gridDim.x * (THREADS / 64);
    const long long total = (long long)B * strips_w * H;
    long long pos = total * wave / nwaves;
    const long long end = total * (wave + 1) / nwaves;
    if (!STATS && pos >= end) return;
    // weights: step (tap, q, u) multiplies channel ci = 16 q + 4 kq + u of the window column kw; A operand [co = 16 nt + j][k = kq]
    float wr[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int qu = 0; qu < 4 * NQ; ++qu) wr[nt][t * 4 * NQ + qu] = wgt.ld(16 * nt + j, 16 * (qu >> 2) + 4 * kq + (qu & 3), t / 3, t % 3);
    const long long img_px = (long long)H * W;
    const int row4 = W * CIN * 4;
    // a lane adds one value per row of its range (~17 rows at the encoder's sizes): fp32 per lane, fp64 from the wave reduction on
    float s[NT][4], q2[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int c = 0; c < 4; ++c) { s[nt][c] = 0.f; q2[nt][c] = 0.f; }
    while (pos < end) {
        const long long bs = pos / H;
        const int r0 = (int)(pos - bs * H), r1 = (int)min((long long)H, r0 + (end - pos));
        const int b = (int)(bs / strips_w), strip = (int)(bs - (long long)b * strips_w);
        pos += r1 - r0;
        const int c0 = strip * NPW;                                   // output columns c0 .. c0 + 13, rows r0 .. r1 - 1
        // the strip's 16 input columns c0 - 1 .. c0 + 14: lane (j, kq) loads the channel quads kq (+ 4 q) of column j
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x + (long long)b * img_px * CIN), 0,
                                                                              (int)(img_px * CIN * 4), 0x00020000);
        const int lcol = c0 - 1 + j;
        const int lvoff = (lcol >= 0 && lcol < W) ? (lcol * CIN + kq * 4) * 4 : 0x7fffffff;         // outside the image: reads 0
        auto load_x = [&](int xr) -> Row {
            Row r;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (xr < 0 || xr >= H) { r.q[q][0] = r.q[q][1] = r.q[q][2] = r.q[q][3] = 0.f; continue; }
                const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lvoff == 0x7fffffff ? lvoff : lvoff + 64 * q, xr * row4, 0);
                r.q[q][0] = __uint_as_float(t[0]); r.q[q][1] = __uint_as_float(t[1]); r.q[q][2] = __uint_as_float(t[2]); r.q[q][3] = __uint_as_float(t[3]);
            }
            return r;
        };
        const int oc = c0 + j - 1;                                    // this lane's output column
        const bool owns = j >= 1 && j <= 14 && oc < W;
        Row R0 = load_x(r0 - 1), R1 = load_x(r0), R2 = load_x(r0 + 1), Rn = load_x(r0 + 2), Rnn;
        float *yb = y + (long long)b * img_px * COUT;
        for (int r = r0; r < r1; ++r) {
            Rnn = load_x(r + 3);
            f32x4 acc[NT], acc2[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[nt] = acc[nt]; }
            I2P_MFMA_PRIO(1);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const Row &R = kh == 0 ? R0 : (kh == 1 ? R1 : R2);
#pragma unroll
                for (int qu = 0; qu < 4 * NQ; ++qu) {
                    // window column kw = strip column j + kw - 1: row_shr:1 reads lane j - 1, row_shl:1 lane j + 1 (lanes 0 / 15 own no output)
                    const float mid = R.q[qu >> 2][qu & 3], left = dpp_f32<0x111>(mid), right = dpp_f32<0x101>(mid);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[nt][(kh * 3 + 0) * 4 * NQ + qu], left, acc[nt], 0, 0, 0);
                        acc2[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[nt][(kh * 3 + 1) * 4 * NQ + qu], mid, acc2[nt], 0, 0, 0);
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[nt][(kh * 3 + 2) * 4 * NQ + qu], right, acc[nt], 0, 0, 0);
                    }
                }
            }
            I2P_MFMA_PRIO(0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[nt][c] += acc2[nt][c];
                if (owns) *reinterpret_cast<f32x4 *>(yb + ((long long)r * W + oc) * COUT + 16 * nt + 4 * kq) = acc[nt];
                if constexpr (MODE == 1) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float v = owns ? acc[nt][c] : 0.f;
                        s[nt][c] += v;
                        q2[nt][c] = __fmaf_rn(v, v, q2[nt][c]);
                    }
                }
            }
            R0 = R1; R1 = R2; R2 = Rn; Rn = Rnn;
        }
    }
    if constexpr (STATS) {
        // the 16 lanes of a row (same kq = same channels) summed, the block's 4 waves through LDS, then one fp64 atomic per channel
        __shared__ double red[THREADS / 64][2 * COUT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                double a = (double)s[nt][c], b2 = (double)q2[nt][c];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m); b2 += __shfl_xor(b2, m); }
                if (j == 0) { red[threadIdx.x >> 6][16 * nt + 4 * kq + c] = a; red[threadIdx.x >> 6][COUT + 16 * nt + 4 * kq + c] = b2; }
            }
        __syncthreads();
        if (threadIdx.x < 2 * COUT) {
            double t = 0.0;
#pragma unroll
            for (int w2 = 0; w2 < THREADS / 64; ++w2) t += red[w2][threadIdx.x];
            atomicAdd(sums + (size_t)(blockIdx.x % REP) * 2 * COUT + threadIdx.x, t);
        }
    }
}

// ---- input gradient with the block tail's backward on load --------------------------------------------------------------------------
// Backward of a 16 -> 16 block whose MaxPool has stride 1: dy = BN-backward(un-pool(g)) of the block's conv output and dx = the input
// gradient of its convolution, in ONE kernel (img_bwd_dx2_kernel + conv3x3_kernel<16,16,0> before: dy written by one, read by the other).
// The strip walk of conv3x3_kernel with one more stage in front: rows of g (float4 per lane) and of the arg-max bytes (one word per
// lane) rotate through registers; dy of a row = the 9 windows that contain the position (three g / arg rows x {lane j - 1, j, j + 1} by
// DPP), the activation derivative and the BN-backward formula from the y row — img_bwd_dx2_kernel's arithmetic — and is at once the
// B operand of the next three conv rows.  dy is still written (the weight gradient reads it), each element by the one wave that owns it.
// A strip yields 12 dx columns (dy needs g of 16 columns for its 14, dx needs dy of 14 for its 12).  dsums: the replica sums of the
// block's BN-backward statistics (taken before, i2p_img_block_bwd_stats); block 0 writes dgamma / dbeta.
struct TailBwd {
    const float *g; const unsigned char *arg; const float *y, *mean_invstd, *gamma, *beta; float slope; const double *dsums;
    float *dy, *dgamma, *dbeta;
};

__global__ __launch_bounds__(THREADS) void conv3x3_tail_bwd_kernel(int B, int H, int W, int strips_w, WView16 wgt, TailBwd tb, float *__restrict__ dx) {
    constexpr int NPW = 12, CC = 16;
    __shared__ double stat[2 * CC], part[THREADS];
    {   // stat[i] = sum over the REP replicas (fixed order): groups of 32 threads share the replicas
        const int idx = threadIdx.x & 31, grp = threadIdx.x >> 5;
        double a = 0.0;
        for (int r = grp; r < REP; r += THREADS / 32) a += tb.dsums[(size_t)r * 2 * CC + idx];
        part[threadIdx.x] = a;
        __syncthreads();
        if (threadIdx.x < 32) {
            double t = 0.0;
            for (int g2 = 0; g2 < THREADS / 32; ++g2) t += part[g2 * 32 + threadIdx.x];
            stat[threadIdx.x] = t;
        }
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x < CC) { tb.dbeta[threadIdx.x] = (float)stat[threadIdx.x]; tb.dgamma[threadIdx.x] = (float)stat[CC + threadIdx.x]; }
    }
    const int lane = threadIdx.x & 63, j = lane & 15, kq = lane >> 4;
    const unsigned wave = i2p_xcd_swizzle(blockIdx.x, gridDim.x) * (THREADS / 64) + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned nwaves = gridDim.x * (THREADS / 64);
    const long long total = (long long)B * strips_w * H;
    long long pos = total * wave / nwaves;
    const long long end = total * (wave + 1) / nwaves;
    if (pos >= end) return;
    float wr[36];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) wr[t * 4 + u] = wgt.ld(j, 4 * kq + u, t / 3, t % 3);
    const float n = (float)((long long)B * H * W);
    float mean[4], invstd[4], scale[4], bet[4], mg[4], mgx[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ch = 4 * kq + c;
        mean[c] = tb.mean_invstd[ch]; invstd[c] = tb.mean_invstd[CC + ch]; scale[c] = invstd[c] * tb.gamma[ch]; bet[c] = tb.beta[ch];
        mg[c] = (float)stat[ch] / n; mgx[c] = (float)stat[CC + ch] / n;
    }
    const long long img_px = (long long)H * W;
    const int row4 = W * CC * 4, rowa = W * CC;
    struct GRow { float g[4]; unsigned a; };
    struct DRow { float q[4]; };
    while (pos < end) {
        const long long bs = pos / H;
        const int r0 = (int)(pos - bs * H), r1 = (int)min((long long)H, r0 + (end - pos));
        const int b = (int)(bs / strips_w), strip = (int)(bs - (long long)b * strips_w);
        pos += r1 - r0;
        const int c0 = strip * NPW, col = c0 - 2 + j;                   // strip column j = image column c0 - 2 + j; dx columns c0 .. c0 + 11
        const bool col_in = col >= 0 && col < W;
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tb.g + (long long)b * img_px * CC), 0, (int)(img_px * CC * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tb.y + (long long)b * img_px * CC), 0, (int)(img_px * CC * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(tb.arg + (long long)b * img_px * CC), 0, (int)(img_px * CC), 0x00020000);
        const int vo4 = col_in ? (col * CC + 4 * kq) * 4 : 0x7fffffff, voa = col_in ? col * CC + 4 * kq : 0x7fffffff;
        auto load_g = [&](int rr) -> GRow {
            GRow o;
            if (rr < 0 || rr >= H) { o.g[0] = o.g[1] = o.g[2] = o.g[3] = 0.f; o.a = 0xffffffffu; return o; }
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rg, vo4, rr * row4, 0);
            o.g[0] = __uint_as_float(t[0]); o.g[1] = __uint_as_float(t[1]); o.g[2] = __uint_as_float(t[2]); o.g[3] = __uint_as_float(t[3]);
            o.a = __builtin_amdgcn_raw_buffer_load_b32(ra, voa, rr * rowa, 0);
            return o;
        };
        auto load_y = [&](int rr) -> DRow {
            DRow o;
            if (rr < 0 || rr >= H) { o.q[0] = o.q[1] = o.q[2] = o.q[3] = 0.f; return o; }
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(ry, vo4, rr * row4, 0);
            o.q[0] = __uint_as_float(t[0]); o.q[1] = __uint_as_float(t[1]); o.q[2] = __uint_as_float(t[2]); o.q[3] = __uint_as_float(t[3]);
            return o;
        };
        float *dyb = tb.dy + (long long)b * img_px * CC;
        // dy of row rho from the g rows rho - 1, rho, rho + 1 (T, M, Bt) and the y row; written when this wave owns it (rows r0 .. r1 - 1,
        // strip columns 2 .. 13)
        auto make_dy = [&](int rho, const GRow &T, const GRow &M, const GRow &Bt, const DRow &Y) -> DRow {
            DRow o;
            if (rho < 0 || rho >= H) { o.q[0] = o.q[1] = o.q[2] = o.q[3] = 0.f; return o; }
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dh = -1; dh <= 1; ++dh) {
                const GRow &G = dh < 0 ? T : (dh == 0 ? M : Bt);
                const unsigned aL = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)G.a, 0x111, 0xF, 0xF, false);      // lane j - 1 (dw = -1)
                const unsigned aR = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)G.a, 0x101, 0xF, 0xF, false);      // lane j + 1 (dw = +1)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float gL = dpp_f32<0x111>(G.g[c]), gR = dpp_f32<0x101>(G.g[c]);
                    const unsigned code = (unsigned)((1 - dh) * 3);                    // window position (1 - dh) * 3 + (1 - dw)
                    acc[c] += ((aL >> (8 * c)) & 0xffu) == code + 2u ? gL : 0.f;
                    acc[c] += ((G.a >> (8 * c)) & 0xffu) == code + 1u ? G.g[c] : 0.f;
                    acc[c] += ((aR >> (8 * c)) & 0xffu) == code ? gR : 0.f;
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float z = (Y.q[c] - mean[c]) * scale[c] + bet[c];
                const float gz = z > 0.f ? acc[c] : acc[c] * tb.slope;
                const float xh = (Y.q[c] - mean[c]) * invstd[c];
                o.q[c] = col_in ? scale[c] * (gz - mg[c] - xh * mgx[c]) : 0.f;
            }
            if (rho >= r0 && rho < r1 && j >= 2 && j <= 13 && col_in)
                *reinterpret_cast<f32x4 *>(dyb + ((long long)rho * W + col) * CC + 4 * kq) = f32x4{o.q[0], o.q[1], o.q[2], o.q[3]};
            return o;
        };
        // g rows r0 - 2 .. and dy rows r0 - 1, r0 to start; each iteration adds dy row r + 1
        GRow G0 = load_g(r0 - 2), G1 = load_g(r0 - 1), G2 = load_g(r0), G3 = load_g(r0 + 1), G4 = load_g(r0 + 2), Gn;
        DRow Y0 = load_y(r0 - 1), Y1 = load_y(r0), Y2 = load_y(r0 + 1), Yn;
        DRow D0 = make_dy(r0 - 1, G0, G1, G2, Y0), D1 = make_dy(r0, G1, G2, G3, Y1), D2;
        G0 = G2; G1 = G3; G2 = G4;                                        // now G0, G1, G2 = rows r0, r0 + 1, r0 + 2
        float *dxb = dx + (long long)b * img_px * CC;
        const bool owns = j >= 2 && j <= 13 && col_in;
        for (int r = r0; r < r1; ++r) {
            Gn = load_g(r + 3);
            Yn = load_y(r + 2);
            D2 = make_dy(r + 1, G0, G1, G2, Y2);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = acc;
            I2P_MFMA_PRIO(1);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const DRow &R = kh == 0 ? D0 : (kh == 1 ? D1 : D2);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float mid = R.q[u], left = dpp_f32<0x111>(mid), right = dpp_f32<0x101>(mid);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[(kh * 3 + 0) * 4 + u], left, acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[(kh * 3 + 1) * 4 + u], mid, acc2, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[(kh * 3 + 2) * 4 + u], right, acc, 0, 0, 0);
                }
            }
            I2P_MFMA_PRIO(0);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] += acc2[c];
            if (owns) *reinterpret_cast<f32x4 *>(dxb + ((long long)r * W + col) * CC + 4 * kq) = acc;
            D0 = D1; D1 = D2; G0 = G1; G1 = G2; G2 = Gn; Y2 = Yn;
        }
    }
}

// (Two more uses of these kernels were built and measured in round 4 and removed in round 5: the forward twin of the one-kernel
//  backward — pooling of block k formed on load by the convolution of block k + 1: 10.97 / 11.02 vs 10.92 ms per step — and bf16-storage
//  variants on v_mfma_f32_16x16x16_bf16: configs[2] 1151 -> 1185 samples/s but outside the 8e-2 pose contract.  DESIGN.md section 4.)

// ---- weight gradient ---------------------------------------------------------------------------------------------------------------
// dW[co][ci][kh][kw] = sum over pixels dy[px][co] x[px + (kh - 1, kw - 1)][ci]: the pixels are the MFMA contraction index, one 16 x 16
// accumulator tile per tap and 16 output channels.  In NHWC storage both operands are lane-linear in memory (A [co = i][k = kq] and
// B [k = kq][ci = j] of a step = 16 consecutive channels of 4 pixels); a wave owns 64 columns as four runs of 16 (kq picks the run, the
// step s the column inside it), so that the three column taps of a row are the SAME 18 loads shifted by one register: 18 + 16 CO/16
// loads feed the 144 CO/16 MFMAs of a row.  Three input rows rotate, the next input / gradient rows are in flight under the MFMAs.
// Waves take equal contiguous ranges of the (image, strip, row) sequence; block sums go to `partials` [blocks][CO/16][9][256], added up
// in fp64 by conv3x3_wgrad_fin_kernel (fixed order).  x has 16 channels.
constexpr int WG_COLS = 64;

template <int CO, bool BF>
__global__ __launch_bounds__(THREADS, 2) void conv3x3_wgrad_kernel(const void *__restrict__ x, const void *__restrict__ dy, int B, int H, int W,
                                                                   int strips_w, float *__restrict__ partials) {
    constexpr int NT = CO / 16, ES = BF ? 2 : 4;                       // bf16 storage: 2-byte loads widened to fp32 (the MFMAs stay fp32)
    __shared__ float red[THREADS / 64][9 * 256];
    const int lane = threadIdx.x & 63, j = lane & 15, kq = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned wave = i2p_xcd_swizzle(blockIdx.x, gridDim.x) * (THREADS / 64) + (unsigned)wv;
    const unsigned nwaves = gridDim.x * (THREADS / 64);
    const long long total = (long long)B * strips_w * H;
    long long pos = total * wave / nwaves;
    const long long end = total * (wave + 1) / nwaves;
    const long long img_px = (long long)H * W;
    const int row4x = W * C * ES, row4g = W * CO * ES;
    f32x4 acc[NT][9];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[nt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    while (pos < end) {
        const long long bs = pos / H;
        const int r0 = (int)(pos - bs * H), r1 = (int)min((long long)H, r0 + (end - pos));
        const int b = (int)(bs / strips_w), strip = (int)(bs - (long long)b * strips_w);
        pos += r1 - r0;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>((const char *)x + (long long)b * img_px * C * ES), 0, (int)(img_px * C * ES), 0x00020000);
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>((const char *)dy + (long long)b * img_px * CO * ES), 0, (int)(img_px * CO * ES), 0x00020000);
        auto ld = [&](__amdgpu_buffer_rsrc_t rs, int vo, int so) -> float {
            if constexpr (BF) return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, vo, so, 0) << 16);
            else return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, 0));
        };
        // slot t = 0 .. 17 of this lane: column strip * 64 + 16 kq + t - 1, channel j; outside the image: an offset that reads 0
        int voff[18];
        const int colb = strip * WG_COLS + 16 * kq - 1;
#pragma unroll
        for (int t = 0; t < 18; ++t) voff[t] = (colb + t >= 0 && colb + t < W) ? ((colb + t) * C + j) * ES : 0x7fffffff;
        auto xrow = [&](int xr, float (&v)[18]) {
            if (xr < 0 || xr >= H) {
#pragma unroll
                for (int t = 0; t < 18; ++t) v[t] = 0.f;
                return;
            }
#pragma unroll
            for (int t = 0; t < 18; ++t) v[t] = ld(rx, voff[t], xr * row4x);
        };
        auto grow = [&](int gr, int nt, float (&v)[16]) {               // gr inside the image; channels 16 nt + j of the gradient row
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                // the gradient's pixel stride is CO floats: (x offset - 4 j) CO / 16 + 4 j
                const int vo = voff[t + 1] == 0x7fffffff ? 0x7fffffff : (voff[t + 1] - ES * j) * NT + ES * j + 16 * ES * nt;
                v[t] = ld(rg, vo, gr * row4g);
            }
        };
        auto tile = [&](int nt, const float (&g)[16], const float (&A0)[18], const float (&A1)[18], const float (&A2)[18]) {
            I2P_MFMA_PRIO(1);
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    acc[nt][0 + kw] = __builtin_amdgcn_mfma_f32_16x16x4f32(g[s2], A0[s2 + kw], acc[nt][0 + kw], 0, 0, 0);
                    acc[nt][3 + kw] = __builtin_amdgcn_mfma_f32_16x16x4f32(g[s2], A1[s2 + kw], acc[nt][3 + kw], 0, 0, 0);
                    acc[nt][6 + kw] = __builtin_amdgcn_mfma_f32_16x16x4f32(g[s2], A2[s2 + kw], acc[nt][6 + kw], 0, 0, 0);
                }
            I2P_MFMA_PRIO(0);
        };
        float X0[18], X1[18], X2[18], Xn[18], G0[16], G1[16];
        xrow(r0 - 1, X0); xrow(r0, X1); xrow(r0 + 1, X2); grow(r0, 0, G0);
        for (int r = r0; r < r1; ++r) {
            xrow(r + 2, Xn);
            if constexpr (NT == 1) {
                if (r + 1 < r1) grow(r + 1, 0, G1);                     // the next row's gradients under this row's 144 MFMAs
                tile(0, G0, X0, X1, X2);
#pragma unroll
                for (int t = 0; t < 16; ++t) G0[t] = G1[t];
            } else {
                grow(r, 1, G1);                                         // the second tile's gradients under the first tile's MFMAs,
                tile(0, G0, X0, X1, X2);
                if (r + 1 < r1) grow(r + 1, 0, G0);                     // the next row's first tile under the second's
                tile(1, G1, X0, X1, X2);
            }
#pragma unroll
            for (int t = 0; t < 18; ++t) { X0[t] = X1[t]; X1[t] = X2[t]; X2[t] = Xn[t]; }
        }
    }
    // D[co = 16 nt + 4 kq + r][ci = j] of tap t -> partials[block][nt][t][(4 kq + r) * 16 + ci], the 4 waves added through LDS
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if (nt) __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wv][t * 256 + (4 * kq + r) * 16 + j] = acc[nt][t][r];
        __syncthreads();
        for (int e = threadIdx.x; e < 9 * 256; e += THREADS)
            partials[((size_t)blockIdx.x * NT + nt) * (9 * 256) + e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    }
}

// dW[co][ci][kh][kw] (weight strides) = sum over the blocks' partial sums, fp64, fixed order.  Block (tile, tap, quarter): 64 outputs x 16
// row groups (every thread a short run of independent loads), the groups added through LDS.
template <bool BF>
__global__ __launch_bounds__(1024) void conv3x3_wgrad_fin_kernel(int nblk, int NT, const float *__restrict__ partials, WView16 wgt, void *__restrict__ dW) {
    __shared__ double part[16][64];
    const int tt = blockIdx.x >> 2, nt = tt / 9, t = tt - nt * 9, e = (blockIdx.x & 3) * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    double a = 0.0;
#pragma unroll 8
    for (int b = grp; b < nblk; b += 16) a += (double)partials[((size_t)b * NT + nt) * (9 * 256) + t * 256 + e];
    part[grp][threadIdx.x & 63] = a;
    __syncthreads();
    if (threadIdx.x < 64) {
        double v = 0.0;
#pragma unroll
        for (int g2 = 0; g2 < 16; ++g2) v += part[g2][threadIdx.x];
        const int co = 16 * nt + (e >> 4), ci = e & 15;
        const int at = co * wgt.s_out + ci * wgt.s_in + (t / 3) * wgt.s_kh + (t % 3) * wgt.s_kw;
        if constexpr (BF) reinterpret_cast<unsigned short *>(dW)[at] = (unsigned short)(pack_bf2((float)v, 0.f) & 0xffffu);
        else reinterpret_cast<float *>(dW)[at] = (float)v;
    }
}

unsigned round8(long long v) { return (unsigned)((v + 7) & ~7ll); }
int num_cus() {
    static const int cus = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
    return cus;
}
bool pair_ok(int cin, int cout) { return cin == 16 && (cout == 16 || cout == 32); }
bool size_ok(int B, int H, int W) { return B >= 0 && H > 0 && W > 0 && (long long)H * W * 32 * 4 < (1ll << 31); }
int wgrad_blocks(int B, int H, int W) {
    const long long total = (long long)B * ((W + WG_COLS - 1) / WG_COLS) * H;       // strip rows
    long long blocks = (long long)num_cus() * 2;                                      // 2 waves per SIMD
    if (blocks * 4 > total) blocks = (total + 3) / 4;
    return (int)round8(blocks < 1 ? 1 : blocks);
}

// forward (flip = 0): x [B,H,W,cin] -> y [B,H,W,cout]; input gradient (flip = 1): x = dL/dy [B,H,W,cout] -> y = dL/dx [B,H,W,cin]
int launch(const void *x, int B, int H, int W, int cin, int cout, const void *w, const int *ws, int flip, void *y, double *sums, int bf16, hipStream_t st) {
    if (!size_ok(B, H, W) || !pair_ok(cin, cout) || bf16) return I2P_ERR_BAD_ARG;      // (bf16 storage: MIOpen runs these layers, see the note above)
    if (B == 0) return 0;
    if (!x || !w || !ws || !y) return I2P_ERR_BAD_ARG;
    const int strips = (W + 13) / 14;
    // 4 waves per SIMD of the whole chip, all resident (<= 128 VGPRs); small tensors: one wave per 4 output rows of a strip
    const long long total = (long long)B * strips * H;
    const int wps = cout == 16 ? 4 : 2;                  // the 32-channel variants hold more weight registers: 2 waves per SIMD fit
    long long blocks = (long long)num_cus() * wps;
    if (blocks * (THREADS / 64) * 4 > total) blocks = (total + 4 * (THREADS / 64) - 1) / (4 * (THREADS / 64));
    const dim3 grid(round8(blocks));
    // forward: W[co][ci][kh][kw] read as [out = co][in = ci]; input gradient: [out = ci][in = co], taps mirrored
    const WView16 wv{(const float *)w, flip ? ws[1] : ws[0], flip ? ws[0] : ws[1], ws[2], ws[3], flip};
#define I2P_CONV(CI_, CO_, MD_) hipLaunchKernelGGL((conv3x3_kernel<CI_, CO_, MD_>), grid, dim3(THREADS), 0, st, (const float *)x, B, H, W, strips, wv, (float *)y, sums)
    if (!flip && cout == 16) { if (sums) I2P_CONV(16, 16, 1); else I2P_CONV(16, 16, 0); }
    else if (!flip) { if (sums) I2P_CONV(16, 32, 1); else I2P_CONV(16, 32, 0); }
    else if (cout == 16) I2P_CONV(16, 16, 0);
    else I2P_CONV(32, 16, 0);
#undef I2P_CONV
    I2P_RETURN_LAUNCH_STATUS();
}

}  // namespace

// y [B,H,W,cout] = conv3x3(x [B,H,W,cin], w [cout,cin,3,3] by element strides ws[4]), padding 1, no bias; (cin, cout) = (16, 16) or
// (16, 32).  bf16 must be 0 (the bf16-storage variants were removed in round 5: I2P_ERR_BAD_ARG).  sums (may be NULL): f64
// [I2P_BN_REPLICAS][2 cout] zeroed by the caller, receives sum y / sum y^2 per output channel (the layout i2p_img_block_pool reads)
extern "C" int i2p_img_conv_fwd(int B, int H, int W, int cin, int cout, int bf16, const void *x, const void *w, const int *ws, void *y, double *sums,
                                void *stream) {
    return launch(x, B, H, W, cin, cout, w, ws, 0, y, sums, bf16, (hipStream_t)stream);
}

// dx [B,H,W,cin] = the input gradient of the same convolution from dy [B,H,W,cout]
extern "C" int i2p_img_conv_bwd_data(int B, int H, int W, int cin, int cout, int bf16, const void *dy, const void *w, const int *ws, void *dx,
                                     void *stream) {
    return launch(dy, B, H, W, cin, cout, w, ws, 1, dx, nullptr, bf16, (hipStream_t)stream);
}

// rows of (cout / 16) * 9 * 256 floats the weight-gradient entry needs in `partials`
extern "C" int i2p_img_conv_wgrad_rows(int B, int H, int W) { return (B <= 0 || H <= 0 || W <= 0) ? 0 : wgrad_blocks(B, H, W); }

// dW (cout * cin * 9 values, written in w's layout: element strides ws[4] of [co][ci][kh][kw]; bf16 must be 0; (formerly: x, dy and dW bf16 bits,
// the products and sums fp32 / fp64) = the weight gradient of the convolution from x [B,H,W,cin] and dy [B,H,W,cout];
// partials: f32 [i2p_img_conv_wgrad_rows()][(cout / 16) * 2304] scratch
extern "C" int i2p_img_conv_wgrad(int B, int H, int W, int cin, int cout, int bf16, const void *x, const void *dy, const int *ws, float *partials,
                                  void *dW, void *stream) {
    if (!size_ok(B, H, W) || !pair_ok(cin, cout) || !ws || !dW || bf16) return I2P_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const WView16 wv{nullptr, ws[0], ws[1], ws[2], ws[3], 0};
    const int nt = cout / 16;
    const int blocks = B == 0 ? 0 : wgrad_blocks(B, H, W), strips = (W + WG_COLS - 1) / WG_COLS;
    if (B > 0) {
        if (!x || !dy || !partials) return I2P_ERR_BAD_ARG;
#define I2P_WG(CO_, BF_) hipLaunchKernelGGL((conv3x3_wgrad_kernel<CO_, BF_>), dim3(blocks), dim3(THREADS), 0, st, x, dy, B, H, W, strips, partials)
        if (cout == 16) I2P_WG(16, false); else I2P_WG(32, false);
#undef I2P_WG
    }
    if (!(B > 0 && i2p_defer_conv_fin(blocks, nt, partials, (float *)dW, wv.s_out, wv.s_in, wv.s_kh, wv.s_kw)))    // (csrc/deferred.hip: with the step's other weight gradients)
    hipLaunchKernelGGL(conv3x3_wgrad_fin_kernel<false>, dim3(36 * nt), dim3(1024), 0, st, blocks, nt, (const float *)partials, wv, dW);
    I2P_RETURN_LAUNCH_STATUS();
}

// Backward of a 16 -> 16 block with a stride-1 MaxPool from its incoming gradient g [B,H,W,16]: dy [B,H,W,16] (gradient of the conv
// output, for i2p_img_conv_wgrad), dx [B,H,W,16], dgamma, dbeta [16].  dsums: the block's BN-backward replica sums, already taken
// (i2p_img_block_bwd_stats).  Replaces i2p_img_block_bwd_dx + i2p_img_conv_bwd_data.
extern "C" int i2p_img_conv_tail_bwd(int B, int H, int W, const float *g, const unsigned char *arg, const float *y, const float *mean_invstd,
                                     const float *gamma, const float *beta, float slope, const double *dsums, const float *w, const int *ws,
                                     float *dy, float *dx, float *dgamma, float *dbeta, void *stream) {
    if (!size_ok(B, H, W)) return I2P_ERR_BAD_ARG;
    if (B == 0) return 0;
    if (!g || !arg || !y || !mean_invstd || !gamma || !beta || !dsums || !w || !ws || !dy || !dx || !dgamma || !dbeta) return I2P_ERR_BAD_ARG;
    const WView16 wv{w, ws[1], ws[0], ws[2], ws[3], 1};               // the input-gradient view of the weights
    const TailBwd tb{g, arg, y, mean_invstd, gamma, beta, slope, dsums, dy, dgamma, dbeta};
    const int strips = (W + 11) / 12;
    const long long total = (long long)B * strips * H;
    long long blocks = (long long)num_cus() * 3;
    if (blocks * (THREADS / 64) * 4 > total) blocks = (total + 4 * (THREADS / 64) - 1) / (4 * (THREADS / 64));
    hipLaunchKernelGGL(conv3x3_tail_bwd_kernel, dim3(round8(blocks)), dim3(THREADS), 0, (hipStream_t)stream, B, H, W, strips, wv, tb, dx);
    I2P_RETURN_LAUNCH_STATUS();
}
