// A whole small MLP chain (1x1 conv -> batch-statistics BN -> LeakyReLU, up to four blocks, optional max over the K
// neighbours of a group at the end) in ONE launch: levels 3-4 of the point pyramid, the cost-volume resampling set conv, the
// up-convolutions, the flow predictors and the pc-stage encodings run on <= 30 000 rows with 64/128-wide layers
// (reference: Conv2d.forward, PPBackbone_center.py:34-46; the stacks of PPBackbone_center.py:77-131, 241-296, 582-603).
//
// Launched layer by layer these are latency chains, not work: a 15 000-row 128 -> 128 layer is 3 us of MFMA time at chip rate
// but 15-25 us as a launch (weights staged, first strip loaded, statistics atomics, ticket, last-block finalisation: five to
// six dependent memory round trips each), plus the stand-alone BN+activation(+max) tail launch.  Here one resident grid keeps
// every block's 64-row strip of activations in LDS through the whole chain:
//
//   per layer:  y = a W^T on v_mfma_f32_16x16x4_f32 (A operand from the LDS strip, W straight from L2: each wave owns a
//               quarter of the output columns for all 64 rows, so a block reads W exactly once) -> strip of y back into LDS
//               -> fp64 column sums {sum y, sum y^2} of the strip, one atomic per block and column on an 8-way replica
//               -> y written to HBM once (saved for the backward, never re-read here) -> GRID BARRIER
//               -> every block forms mean / scale / beta from the replica sums -> BN + activation in place in LDS.
//   last layer: the activated strip (or its max over K consecutive rows + arg-max byte) is the only other thing written.
//
// The grid barrier is a counter in global memory (arrive = one agent-scope atomic per block after its own atomics are
// acknowledged, wait = agent-scope polling); the launcher only accepts row counts whose grid is co-resident
// (i2p_chain_fwd_ok), and a poll limit turns a lost barrier into an error word instead of a hung GPU.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int CH_THREADS = 256, CH_ROWS = 64, CH_MAXL = 4, CH_MAXC = 272, CH_REP = 8;

struct ChainP {
    long long rows;
    int nl;
    int c[CH_MAXL + 1];               // c[0]: row length of x (multiple of 4); c[l+1]: output width of layer l (multiple of 64)
    int w_ld[CH_MAXL];                // row length of W_l = the layer's real input width (<= c[l]; columns beyond it are zero inputs)
    const float *x;
    const float *w[CH_MAXL], *gamma[CH_MAXL], *beta[CH_MAXL];
    float slope[CH_MAXL];
    float eps;
    float *y[CH_MAXL], *coef[CH_MAXL], *mi[CH_MAXL];
    double *sums;                     // [nl][CH_REP][2 * smax], zero on entry; smax = widest output of the chain
    int smax;
    int pool_k;                       // 0: out = act(bn(y_last)) [rows, c]; else out [rows / pool_k, c] + arg
    float *out;
    unsigned char *arg;
    float *w0_pad;                    // optional [c[1]][c[0]]: W_0 with zero columns (what the backward kernels take)
    unsigned *sync;                   // [0] barrier arrivals, [1] exits, [2] error word; [0], [1] zero on entry and on exit
    int lda;
};

__device__ __forceinline__ float act(float z, float slope) { return z > 0.f ? z : z * slope; }

// all blocks of the grid have arrived `target` times in total
__device__ __forceinline__ void grid_barrier(unsigned *sync, unsigned target, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's statistics atomics are acknowledged (performed at L2 / memory side)
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned polls = 0;
        while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++polls > (1u << 19)) {                  // ~ a second: the grid was not co-resident; give up loudly instead of hanging
                __hip_atomic_store(sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

// y strip [64][16*4*NT columns] of one layer: wave `wave` computes columns [wave*16*NT, (wave+1)*16*NT) for the four 16-row tiles
template <int NT>
__device__ __forceinline__ void layer_mma(const float *A, int lda, int kpad, const float *__restrict__ W, int w_ld, bool w_vec,
                                          int wave, int i, int q, f32x4 (&acc)[4][NT]) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *wrow[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wrow[t] = W + (size_t)((wave * NT + t) * 16 + i) * w_ld;
    auto loadb = [&](int k0, f32x4 (&b)[NT]) {
        const int k = k0 + 4 * q;                        // this lane's four contraction indices (k-step e takes element 4q + e)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (w_vec && k + 3 < w_ld) {
                b[t] = *reinterpret_cast<const f32x4 *>(wrow[t] + k);
            } else {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (k + e < w_ld) v[e] = wrow[t][k + e];
                b[t] = v;
            }
        }
    };
    f32x4 cur[NT], nxt[NT];
    loadb(0, cur);
    for (int k0 = 0; k0 < kpad; k0 += 16) {
        if (k0 + 16 < kpad) loadb(k0 + 16, nxt);
        f32x4 a[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) a[rt] = *reinterpret_cast<const f32x4 *>(A + (size_t)(rt * 16 + i) * lda + k0 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][e], cur[t][e], acc[rt][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) cur[t] = nxt[t];
    }
}

template <int NT>
__device__ __forceinline__ void layer_gemm(float *A, int lda, int kpad, const float *W, int w_ld, bool w_vec, int wave, int i, int q) {
    f32x4 acc[4][NT];
    layer_mma<NT>(A, lda, kpad, W, w_ld, w_vec, wave, i, q, acc);
    __syncthreads();                                     // every wave is done reading the input strip: the output strip replaces it
    // D of a tile: lane (column i, q), register e = row 4q + e of the 16
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) A[(size_t)(rt * 16 + 4 * q + e) * lda + (wave * NT + t) * 16 + i] = acc[rt][t][e];
    __syncthreads();
}

__global__ __launch_bounds__(CH_THREADS) void chain_fwd_kernel(ChainP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *A = smem;                                     // [64][lda]
    float *cf = smem + (size_t)CH_ROWS * p.lda;           // [3][CH_MAXC]: mean, scale, beta of the current layer
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
    const long long row0 = (long long)blockIdx.x * CH_ROWS;
    const int nvalid = (int)((p.rows - row0) < CH_ROWS ? (p.rows - row0) : CH_ROWS);
    const unsigned G = gridDim.x;

    {   // input strip: x rows (coalesced 16-byte loads), zero beyond the row / column range up to the next multiple of 16
        const int c0 = p.c[0], kp = (c0 + 15) & ~15, v = kp >> 2;
        for (int idx = tid; idx < CH_ROWS * v; idx += CH_THREADS) {
            const int r = idx / v, c4 = (idx - r * v) * 4;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (r < nvalid && c4 < c0) val = *reinterpret_cast<const f32x4 *>(p.x + (size_t)(row0 + r) * c0 + c4);
            *reinterpret_cast<f32x4 *>(A + (size_t)r * p.lda + c4) = val;
        }
        if (p.w0_pad) {                                  // W_0 with zero columns, rows spread over the grid
            const int c1 = p.c[1], ld = p.w_ld[0];
            for (int r = blockIdx.x; r < c1; r += G)
                for (int c = tid; c < c0; c += CH_THREADS) p.w0_pad[(size_t)r * c0 + c] = c < ld ? p.w[0][(size_t)r * ld + c] : 0.f;
        }
        __syncthreads();
    }

    for (int l = 0; l < p.nl; ++l) {
        const int cin = p.c[l], cout = p.c[l + 1], kpad = (cin + 15) & ~15, w_ld = p.w_ld[l];
        const bool w_vec = (w_ld & 3) == 0 && (reinterpret_cast<uintptr_t>(p.w[l]) & 15) == 0;
        switch (cout >> 6) {
            case 1: layer_gemm<1>(A, p.lda, kpad, p.w[l], w_ld, w_vec, wave, i, q); break;
            case 2: layer_gemm<2>(A, p.lda, kpad, p.w[l], w_ld, w_vec, wave, i, q); break;
            case 3: layer_gemm<3>(A, p.lda, kpad, p.w[l], w_ld, w_vec, wave, i, q); break;
            default: layer_gemm<4>(A, p.lda, kpad, p.w[l], w_ld, w_vec, wave, i, q); break;
        }
        // column sums of the strip (rows beyond the range are exact zeros), one atomic per block, column and moment
        double *sums = p.sums + ((size_t)l * CH_REP + (blockIdx.x % CH_REP)) * 2 * p.smax;
        for (int c = tid; c < cout; c += CH_THREADS) {
            double s = 0.0, s2 = 0.0;
#pragma unroll 8
            for (int r = 0; r < CH_ROWS; ++r) { const double v = (double)A[(size_t)r * p.lda + c]; s += v; s2 += v * v; }
            atomicAdd(sums + c, s);
            atomicAdd(sums + p.smax + c, s2);
        }
        {   // the pre-BN strip goes to HBM once (the backward reads it)
            const int v = cout >> 2;
            float *y = p.y[l];
            for (int idx = tid; idx < nvalid * v; idx += CH_THREADS) {
                const int r = idx / v, c4 = (idx - r * v) * 4;
                *reinterpret_cast<f32x4 *>(y + (size_t)(row0 + r) * cout + c4) = *reinterpret_cast<const f32x4 *>(A + (size_t)r * p.lda + c4);
            }
        }
        grid_barrier(p.sync, (unsigned)(l + 1) * G, tid);
        double *sl = p.sums + (size_t)l * CH_REP * 2 * p.smax;
        for (int c = tid; c < cout; c += CH_THREADS) {
            double s = 0.0, s2 = 0.0;
#pragma unroll
            for (int r = 0; r < CH_REP; ++r) {
                s += __hip_atomic_load(sl + (size_t)r * 2 * p.smax + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s2 += __hip_atomic_load(sl + (size_t)r * 2 * p.smax + p.smax + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const double m = s / (double)p.rows;
            double var = s2 / (double)p.rows - m * m;
            var = var < 0.0 ? 0.0 : var;
            const float invstd = rsqrtf((float)var + p.eps);
            const float mu = (float)m, sc = invstd * p.gamma[l][c], be = p.beta[l][c];
            cf[c] = mu; cf[CH_MAXC + c] = sc; cf[2 * CH_MAXC + c] = be;
            if (blockIdx.x == 0) {
                p.coef[l][c] = mu; p.coef[l][cout + c] = sc; p.coef[l][2 * cout + c] = be;
                p.mi[l][c] = mu; p.mi[l][cout + c] = invstd;
            }
        }
        __syncthreads();
        {   // BN + activation in place; rows beyond the range stay zero (they must not enter the next layer's statistics)
            const int v = cout >> 2;
            const float slope = p.slope[l];
            const bool last = l == p.nl - 1;
            for (int idx = tid; idx < CH_ROWS * v; idx += CH_THREADS) {
                const int r = idx / v, c4 = (idx - r * v) * 4;
                f32x4 val = *reinterpret_cast<const f32x4 *>(A + (size_t)r * p.lda + c4);
                const f32x4 mu = *reinterpret_cast<const f32x4 *>(cf + c4), sc = *reinterpret_cast<const f32x4 *>(cf + CH_MAXC + c4),
                            be = *reinterpret_cast<const f32x4 *>(cf + 2 * CH_MAXC + c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) val[e] = r < nvalid ? act((val[e] - mu[e]) * sc[e] + be[e], slope) : 0.f;
                if (last && !p.pool_k) {
                    if (r < nvalid) *reinterpret_cast<f32x4 *>(p.out + (size_t)(row0 + r) * cout + c4) = val;
                } else {
                    *reinterpret_cast<f32x4 *>(A + (size_t)r * p.lda + c4) = val;
                }
            }
        }
        __syncthreads();
    }

    if (p.pool_k) {      // max over groups of pool_k consecutive rows (pool_k divides 64: a group never leaves the strip); first k wins ties, NaN propagates
        const int cout = p.c[p.nl], v = cout >> 2, K = p.pool_k, ng = CH_ROWS / K;
        const long long g0 = row0 / K;
        for (int idx = tid; idx < ng * v; idx += CH_THREADS) {
            const int g = idx / v, c4 = (idx - g * v) * 4;
            if (g * K >= nvalid) continue;
            f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            uchar4 bi = make_uchar4(0, 0, 0, 0);
            for (int k = 0; k < K; ++k) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(A + (size_t)(g * K + k) * p.lda + c4);
                if (a[0] > best[0] || a[0] != a[0]) { best[0] = a[0]; bi.x = (unsigned char)k; }
                if (a[1] > best[1] || a[1] != a[1]) { best[1] = a[1]; bi.y = (unsigned char)k; }
                if (a[2] > best[2] || a[2] != a[2]) { best[2] = a[2]; bi.z = (unsigned char)k; }
                if (a[3] > best[3] || a[3] != a[3]) { best[3] = a[3]; bi.w = (unsigned char)k; }
            }
            *reinterpret_cast<f32x4 *>(p.out + (size_t)(g0 + g) * cout + c4) = best;
            *reinterpret_cast<uchar4 *>(p.arg + (size_t)(g0 + g) * cout + c4) = bi;
        }
    }
    if (tid == 0) {      // the last block out leaves the barrier words zero for the next launch
        const unsigned t = __hip_atomic_fetch_add(p.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == G - 1) {
            __hip_atomic_store(p.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int chain_cmax(int nl, const int *widths) {
    int m = 0;
    for (int l = 0; l <= nl; ++l) m = widths[l] > m ? widths[l] : m;
    return m;
}

size_t chain_lds_bytes(int cmax) {
    const int lda = ((cmax + 15) & ~15) + 4;
    return ((size_t)CH_ROWS * lda + 3 * CH_MAXC) * sizeof(float);
}

int chain_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cus = prop.multiProcessorCount;
    }
    return cus;
}

}  // namespace

// doubles of zeroed scratch i2p_chain_fwd needs for a chain of `nl` layers whose widest output is `cmax_out`
extern "C" long long i2p_chain_sums_len(int nl, int cmax_out) { return (long long)nl * CH_REP * 2 * cmax_out; }

// widths[0] = row length of x, widths[1..nl] = output widths
extern "C" int i2p_chain_fwd_ok(long long rows, int nl, const int *widths, int pool_k) {
    if (rows <= 0 || nl < 1 || nl > CH_MAXL || !widths) return 0;
    if (widths[0] <= 0 || (widths[0] & 3) || widths[0] > CH_MAXC) return 0;
    for (int l = 1; l <= nl; ++l)
        if (widths[l] <= 0 || (widths[l] & 63) || widths[l] > 256) return 0;
    if (pool_k < 0 || pool_k > CH_ROWS || (pool_k && (CH_ROWS % pool_k || rows % pool_k))) return 0;
    const size_t lds = chain_lds_bytes(chain_cmax(nl, widths));
    const int per_cu = (int)((160 * 1024) / lds) > 2 ? 2 : (int)((160 * 1024) / lds);
    const int cus = chain_cus();
    const long long blocks = (rows + CH_ROWS - 1) / CH_ROWS;
    return per_cu >= 1 && cus > 0 && blocks <= (long long)cus * per_cu ? 1 : 0;
}

extern "C" int i2p_chain_fwd(long long rows, int nl, const int *widths, const int *w_ld, const float *x, const float *const *w,
                             const float *const *gamma, const float *const *beta, const float *slopes, float eps, float *const *y,
                             float *const *coef, float *const *mean_invstd, double *sums, int pool_k, float *out, unsigned char *arg,
                             float *w0_pad, unsigned *sync, void *stream) {
    if (!i2p_chain_fwd_ok(rows, nl, widths, pool_k)) return I2P_ERR_BAD_ARG;
    if (!w_ld || !x || !w || !gamma || !beta || !slopes || !y || !coef || !mean_invstd || !sums || !out || !sync || (pool_k && !arg))
        return I2P_ERR_BAD_ARG;
    ChainP p{};
    p.rows = rows; p.nl = nl; p.x = x; p.eps = eps; p.sums = sums; p.pool_k = pool_k; p.out = out; p.arg = arg; p.w0_pad = w0_pad;
    p.sync = sync;
    p.c[0] = widths[0];
    for (int l = 0; l < nl; ++l) {
        p.c[l + 1] = widths[l + 1]; p.w_ld[l] = w_ld[l];
        if (w_ld[l] <= 0 || w_ld[l] > widths[l] || !w[l] || !gamma[l] || !beta[l] || !y[l] || !coef[l] || !mean_invstd[l]) return I2P_ERR_BAD_ARG;
        p.w[l] = w[l]; p.gamma[l] = gamma[l]; p.beta[l] = beta[l]; p.slope[l] = slopes[l];
        p.y[l] = y[l]; p.coef[l] = coef[l]; p.mi[l] = mean_invstd[l];
    }
    for (int l = 1; l <= nl; ++l) p.smax = widths[l] > p.smax ? widths[l] : p.smax;
    const int cmax = chain_cmax(nl, widths);
    p.lda = ((cmax + 15) & ~15) + 4;
    const size_t bytes = chain_lds_bytes(cmax);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(chain_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const unsigned grid = (unsigned)((rows + CH_ROWS - 1) / CH_ROWS);
    hipLaunchKernelGGL(chain_fwd_kernel, dim3(grid), dim3(CH_THREADS), bytes, (hipStream_t)stream, p);
    I2P_RETURN_LAUNCH_STATUS();
}
