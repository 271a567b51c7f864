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
constexpr int CH_THREADS = 256, CH_ROWS = 64, CH_MAXL = 4, CH_MAXC = 272, CH_REP = 16;

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
    unsigned *sync;                   // i2p_chain_sync_words() words, zero on entry and (but for the error word) on exit
    int lda;
    int abl;                          // diagnostic ablation bits (I2P_CHAIN_ABL; tools/time_chain.py): 0 in production
};

__device__ __forceinline__ float act(float z, float slope) { return z > 0.f ? z : z * slope; }

// Grid barrier number `k` (1-based) of this launch.  Two levels so that no address sees more than G/8 + 8 atomics or pollers
// (one counter polled by all 456 blocks of a 29 000-row chain cost 15 us per barrier: the pollers' loads queue in front of the
// arrivals at the line's home channel): block b arrives at the counter of group b % 8, the last arrival of a group arrives at
// the top counter, the last group publishes k to the eight per-group flags, and a block polls only its group's flag.  Every
// word sits in its own 128-byte line.
constexpr int CH_NG = 8, CH_LINE = 32, CH_TOP = CH_NG * CH_LINE, CH_FLAG = (CH_NG + 1) * CH_LINE, CH_EXIT = (2 * CH_NG + 1) * CH_LINE,
              CH_ERR = (2 * CH_NG + 2) * CH_LINE, CH_SYNC_WORDS = (2 * CH_NG + 3) * CH_LINE;

__device__ __forceinline__ void grid_barrier(unsigned *sync, unsigned k, unsigned G, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's statistics atomics are acknowledged (performed at L2 / memory side)
    __syncthreads();
    if (tid == 0) {
        const unsigned g = blockIdx.x & (CH_NG - 1), ng = G < CH_NG ? G : CH_NG, n_g = (G - g + CH_NG - 1) / CH_NG;
        const unsigned t = __hip_atomic_fetch_add(sync + g * CH_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == k * n_g - 1) {
            const unsigned t2 = __hip_atomic_fetch_add(sync + CH_TOP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t2 == k * ng - 1)
                for (unsigned j = 0; j < ng; ++j) __hip_atomic_store(sync + CH_FLAG + j * CH_LINE, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned polls = 0;
        while (__hip_atomic_load(sync + CH_FLAG + g * CH_LINE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k) {
            __builtin_amdgcn_s_sleep(2);
            if (++polls > (1u << 19)) {                  // ~ a second: the grid was not co-resident; give up loudly instead of hanging
                __hip_atomic_store(sync + CH_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

// Weights of one BATCH of BC 16-element contraction chunks for this wave's NT column tiles (k-step e of a chunk takes elements
// 4q + e, so a lane's operands for four k-steps are one float4 of its W row).  pre[j * NT + t]: chunk j of the batch, tile t.
template <int NT, int BC>
__device__ __forceinline__ void load_wbatch(f32x4 (&pre)[BC * NT], const float *__restrict__ W, int w_ld, bool w_vec, int kb, int kpad, int wave,
                                            int i, int q, int abl) {
#pragma unroll
    for (int j = 0; j < BC; ++j) {
        const int k0 = kb + 16 * j, k = k0 + 4 * q;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float *wrow = W + (size_t)((wave * NT + t) * 16 + i) * w_ld;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k0 < kpad) {
                if (abl & 2) {
                    v = f32x4{0.5f, 0.25f, 0.125f, 1.f};
                } else if (w_vec && k + 3 < w_ld) {
                    v = *reinterpret_cast<const f32x4 *>(wrow + k);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e < w_ld) v[e] = wrow[k + e];
                }
            }
            pre[j * NT + t] = v;
        }
    }
}

struct Strip {              // what a block knows about its 64-row strip
    float *A, *cf;
    int lda, tid, nvalid;
    long long row0;
};

// input strip: x rows (coalesced 16-byte loads, eight in flight per thread), zero beyond the row / column range up to the next multiple
// of 16; thread = (row of a pass, 16-byte column): no division per element
__device__ __forceinline__ void load_x_strip(const ChainP &p, const Strip &s) {
    const int c0 = p.c[0], kp = (c0 + 15) & ~15, v = kp >> 2, rpp = CH_THREADS / v, tr = s.tid / v, c4 = (s.tid - tr * v) * 4;
    if (tr < rpp) {
        for (int rb = tr; rb < CH_ROWS; rb += 8 * rpp) {
            f32x4 buf[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = rb + u * rpp;
                buf[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (r < s.nvalid && c4 < c0) buf[u] = *reinterpret_cast<const f32x4 *>(p.x + (size_t)(s.row0 + r) * c0 + c4);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = rb + u * rpp;
                if (r < CH_ROWS) *reinterpret_cast<f32x4 *>(s.A + (size_t)r * s.lda + c4) = buf[u];
            }
        }
    }
}

// One pass over the strip of layer l once its BN is known (cf): the pre-BN values go to HBM (once; the backward reads them — issued
// AFTER the barrier, so their latency hides under the next layer instead of in front of the barrier), BN + activation in place (or
// straight to `out` for the last layer of an un-pooled chain); rows beyond the range stay zero (they must not enter the next
// layer's statistics).
__device__ __forceinline__ void apply_strip(const ChainP &p, const Strip &s, int l) {
    const int cout = p.c[l + 1], v = cout >> 2, rpp = CH_THREADS / v, tr = s.tid / v, c4 = (s.tid - tr * v) * 4;
    const float slope = p.slope[l];
    const bool to_out = l == p.nl - 1 && !p.pool_k, ywr = !(p.abl & 32);
    float *y = p.y[l];
    if (tr >= rpp) return;
    const f32x4 mu = *reinterpret_cast<const f32x4 *>(s.cf + c4), sc = *reinterpret_cast<const f32x4 *>(s.cf + CH_MAXC + c4),
                be = *reinterpret_cast<const f32x4 *>(s.cf + 2 * CH_MAXC + c4);
    for (int r = tr; r < CH_ROWS; r += rpp) {
        f32x4 val = *reinterpret_cast<const f32x4 *>(s.A + (size_t)r * s.lda + c4);
        const bool r_ok = r < s.nvalid;
        if (r_ok && ywr) *reinterpret_cast<f32x4 *>(y + (size_t)(s.row0 + r) * cout + c4) = val;
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] = r_ok ? act((val[e] - mu[e]) * sc[e] + be[e], slope) : 0.f;
        if (to_out) {
            if (r_ok) *reinterpret_cast<f32x4 *>(p.out + (size_t)(s.row0 + r) * cout + c4) = val;
        } else {
            *reinterpret_cast<f32x4 *>(s.A + (size_t)r * s.lda + c4) = val;
        }
    }
}

// Layer l on the strip: wave `wave` computes columns [wave*16*NT, (wave+1)*16*NT) for the four 16-row tiles, so a block reads W exactly
// once.  The first batch of W is requested FIRST and rides out the input phase (the x strip's loads for layer 0, the BN + activation
// pass over the previous layer's strip otherwise); afterwards the next batch is in flight while a batch's 16 * BC * NT MFMAs run.
template <int NT>
__device__ __forceinline__ void layer_step(const ChainP &p, const Strip &s, int l, int wave, int i, int q) {
    const int kpad = (p.c[l] + 15) & ~15, w_ld = p.w_ld[l], lda = s.lda;
    const float *W = p.w[l];
    const bool w_vec = (w_ld & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;
    float *A = s.A;
    constexpr int BC = NT <= 2 ? 4 : 2;                  // 16-element chunks per weight batch (registers: 2 * BC * NT float4)
    f32x4 cur[BC * NT], nxt[BC * NT];
    load_wbatch<NT, BC>(cur, W, w_ld, w_vec, 0, kpad, wave, i, q, p.abl);
    if (l == 0) load_x_strip(p, s); else apply_strip(p, s, l - 1);
    __syncthreads();
    f32x4 acc[4][NT];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < ((p.abl & 1) ? 0 : kpad); kb += 16 * BC) {
        if (kb + 16 * BC < kpad) load_wbatch<NT, BC>(nxt, W, w_ld, w_vec, kb + 16 * BC, kpad, wave, i, q, p.abl);
#pragma unroll
        for (int j = 0; j < BC; ++j) {
            const int k0 = kb + 16 * j;
            if (k0 < kpad) {
                f32x4 a[4];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) a[rt] = *reinterpret_cast<const f32x4 *>(A + (size_t)(rt * 16 + i) * lda + k0 + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][e], cur[j * NT + t][e], acc[rt][t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < BC * NT; ++u) cur[u] = nxt[u];
    }
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

__global__ __launch_bounds__(CH_THREADS, 2) void chain_fwd_kernel(ChainP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    Strip s;
    s.A = smem;                                          // [64][lda]
    s.cf = smem + (size_t)CH_ROWS * p.lda;                // [3][CH_MAXC]: mean, scale, beta of the current layer
    double *red = reinterpret_cast<double *>(s.cf + 3 * CH_MAXC);     // [row groups][2][cout] partial column sums (<= 512 doubles)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
    s.lda = p.lda; s.tid = tid;
    s.row0 = (long long)blockIdx.x * CH_ROWS;
    s.nvalid = (int)((p.rows - s.row0) < CH_ROWS ? (p.rows - s.row0) : CH_ROWS);
    const unsigned G = gridDim.x;
    float *A = s.A, *cf = s.cf;

    if (p.w0_pad) {                                      // W_0 with zero columns, rows spread over the grid
        const int c0 = p.c[0], c1 = p.c[1], ld = p.w_ld[0];
        for (int r = blockIdx.x; r < c1; r += G)
            for (int c = tid; c < c0; c += CH_THREADS) p.w0_pad[(size_t)r * c0 + c] = c < ld ? p.w[0][(size_t)r * ld + c] : 0.f;
    }

    for (int l = 0; l < p.nl; ++l) {
        const int cout = p.c[l + 1];
        // this layer's gamma / beta (one channel per thread): asked for now, needed after the barrier
        float gam = 0.f, bet = 0.f;
        if (tid < cout) { gam = p.gamma[l][tid]; bet = p.beta[l][tid]; }
        switch (cout >> 6) {
            case 1: layer_step<1>(p, s, l, wave, i, q); break;
            case 2: layer_step<2>(p, s, l, wave, i, q); break;
            case 3: layer_step<3>(p, s, l, wave, i, q); break;
            default: layer_step<4>(p, s, l, wave, i, q); break;
        }
        // column sums of the strip (rows beyond the range are exact zeros): 256 / cout row groups in parallel, combined through LDS,
        // then one atomic per block, column and moment
        {
            const int ngr = CH_THREADS / cout, gr = tid / cout, c = tid - gr * cout, rpg = CH_ROWS / ngr;
            if (gr < ngr) {
                double sm = 0.0, s2 = 0.0;
#pragma unroll 16
                for (int r = gr * rpg; r < (gr + 1) * rpg; ++r) { const double v = (double)A[(size_t)r * p.lda + c]; sm += v; s2 += v * v; }
                red[(gr * 2) * cout + c] = sm; red[(gr * 2 + 1) * cout + c] = s2;
            }
            __syncthreads();
            if (tid < cout && !(p.abl & 4)) {
                double sm = 0.0, s2 = 0.0;
                for (int g2 = 0; g2 < ngr; ++g2) { sm += red[(g2 * 2) * cout + tid]; s2 += red[(g2 * 2 + 1) * cout + tid]; }
                double *sums = p.sums + ((size_t)l * CH_REP + (blockIdx.x % CH_REP)) * 2 * p.smax;
                atomicAdd(sums + tid, sm);
                atomicAdd(sums + p.smax + tid, s2);
            }
        }
        if (!(p.abl & 8)) grid_barrier(p.sync, (unsigned)(l + 1), G, tid);
        double *sl = p.sums + (size_t)l * CH_REP * 2 * p.smax;
        if (tid < cout) {
            const int c = tid;
            double sm = 0.0, s2 = 0.0;
#pragma unroll
            for (int r = 0; r < ((p.abl & 16) ? 0 : CH_REP); ++r) {
                sm += __hip_atomic_load(sl + (size_t)r * 2 * p.smax + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s2 += __hip_atomic_load(sl + (size_t)r * 2 * p.smax + p.smax + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const double m = sm / (double)p.rows;
            double var = s2 / (double)p.rows - m * m;
            var = var < 0.0 ? 0.0 : var;
            const float invstd = rsqrtf((float)var + p.eps);
            const float mu = (float)m, sc = invstd * gam, be = bet;
            cf[c] = mu; cf[CH_MAXC + c] = sc; cf[2 * CH_MAXC + c] = be;
            if (blockIdx.x == 0) {
                p.coef[l][c] = mu; p.coef[l][cout + c] = sc; p.coef[l][2 * cout + c] = be;
                p.mi[l][c] = mu; p.mi[l][cout + c] = invstd;
            }
        }
        __syncthreads();
    }
    apply_strip(p, s, p.nl - 1);
    __syncthreads();

    if (p.pool_k) {      // max over groups of pool_k consecutive rows (pool_k divides 64: a group never leaves the strip); first k wins ties, NaN propagates
        const int cout = p.c[p.nl], v = cout >> 2, K = p.pool_k, ng = CH_ROWS / K, gpp = CH_THREADS / v, tg = tid / v, c4 = (tid - tg * v) * 4;
        const long long g0 = s.row0 / K;
        for (int g = tg; tg < gpp && g < ng && g * K < s.nvalid; g += gpp) {
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
        const unsigned t = __hip_atomic_fetch_add(p.sync + CH_EXIT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == G - 1)
            for (int j = 0; j <= 2 * CH_NG + 1; ++j) __hip_atomic_store(p.sync + j * CH_LINE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int chain_cmax(int nl, const int *widths) {
    int m = 0;
    for (int l = 0; l <= nl; ++l) m = widths[l] > m ? widths[l] : m;
    return m;
}

size_t chain_lds_bytes(int cmax) {
    const int lda = ((cmax + 15) & ~15) + 4;
    return ((size_t)CH_ROWS * lda + 3 * CH_MAXC) * sizeof(float) + 512 * sizeof(double);
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
// uint32 words of zeroed scratch behind `sync`; word i2p_chain_sync_words() - 32 is the error word
extern "C" long long i2p_chain_sync_words(void) { return CH_SYNC_WORDS; }

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
    { const char *e = getenv("I2P_CHAIN_ABL"); p.abl = e ? atoi(e) : 0; }
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
