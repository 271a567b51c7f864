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
//               quarter of the output columns for all 64 rows, so a block reads W exactly once; the first weight batch is requested
//               before the input phase, the next one is in flight under the current one's MFMAs) -> strip of y back into LDS
//               -> fp64 column sums {sum y, sum y^2} of the strip, one atomic per block and column on a 16-way replica
//               -> GRID BARRIER -> every block forms mean / scale / beta from the replica sums
//               -> one pass: y to HBM (once, saved for the backward, never re-read here; issued after the barrier so its latency
//               hides under the next layer) and BN + activation in place in LDS.
//   last layer: the activated strip (or its max over K consecutive rows + arg-max byte) is the only other thing written.
//
// The grid barrier (grid_barrier below) is two levels of counters in global memory with per-group release flags — a single
// polled counter cost 15 us per barrier at 456 blocks; arrive = one agent-scope atomic per block after its own atomics are
// acknowledged, wait = agent-scope polling of the group's flag.  The launcher only accepts row counts whose grid is co-resident
// (i2p_chain_fwd_ok), and a poll limit turns a lost barrier into an error word (+ a process-wide counter the host checks,
// i2p_chain_set_error_counter) instead of a hung GPU.
//
// The second half of the file is the backward of a chain on the same scheme (chain_bwd_kernel + chain_reduce_kernel) and the C ABI.
// Measurements and counters: tools/time_chain.py (ablation bits I2P_CHAIN_ABL), tools/pmc_chain.sh, profiles/r03_pmc_chain.txt.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int CH_THREADS = 256, CH_ROWS = 64, CH_MAXL = 4, CH_MAXC = 272, CH_REP = 16;

// Where a lost grid barrier is reported: the launch's own error word (sync[CH_ERR]), a per-device fp32 counter in device memory
// (the trainer appends it to the flat gradient, so that the optimiser kernel — and, through the all-reduce, every rank — skips the
// update of a poisoned step) and a HOST-MAPPED word the host reads without synchronising (Trainer.step checks it every step).
struct ChainErr {
    float *counter;                   // device fp32 [>= 1] or nullptr
    unsigned *hflag;                  // pinned host uint32 [1], device-accessible, or nullptr
    unsigned poll_limit;              // polls of one barrier wait before it gives up (~1.5 us each)
};

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
    ChainErr err;                     // where a timed-out barrier is reported (i2p_chain_set_error_words) + the poll limit
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

// Memory ordering (VERDICT r3 #1b / MI355X_MICROARCH.md, inter-workgroup visibility): per-XCD L2s are not coherent and a CU's L1 is
// never refreshed by other CUs, so the arrival is preceded by an AGENT-scope RELEASE (buffer_wbl2 sc1: everything this block stored
// before the barrier — the statistics it added with agent-scope atomics are already performed at the memory side, its plain stores
// are written back) and the wait is followed by ONE agent-scope ACQUIRE (buffer_inv sc1) by the polling lane, which covers the block
// through the __syncthreads() behind it; the polls themselves stay relaxed (an acquire per poll would invalidate the L1 every
// iteration).  The asm wait after the release fence restates the post-write-back wait where the compiler cannot drop it (ROCm 7.2
// drops it when its scoreboard is provably empty: guide, Guideline 16 pitfall 12).  abl bit 64 = the round-3 relaxed form (A/B timing).
//
// Returns false when the barrier was abandoned (the grid is not co-resident: CU mask, another process, an over-sized grid): the
// poll limit is reached or another block already gave up (the launch's error word is checked every 64 polls, so one time-out ends
// every waiter).  The caller stops using barriers (its results are garbage anyway) and the three error sinks of ChainErr are set.
__device__ __forceinline__ bool grid_barrier(unsigned *sync, unsigned k, unsigned G, int tid, const ChainErr &err, unsigned *s_ok, int abl) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's statistics atomics are acknowledged (performed at L2 / memory side)
    __syncthreads();
    if (tid == 0) {
#ifdef I2P_RELAXED_SYNC
        const bool fenced = false;
#else
        const bool fenced = !(abl & 64);
#endif
        if (fenced) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned g = blockIdx.x & (CH_NG - 1), ng = G < CH_NG ? G : CH_NG, n_g = (G - g + CH_NG - 1) / CH_NG;
        const unsigned t = __hip_atomic_fetch_add(sync + g * CH_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == k * n_g - 1) {
            const unsigned t2 = __hip_atomic_fetch_add(sync + CH_TOP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t2 == k * ng - 1)
                for (unsigned j = 0; j < ng; ++j) __hip_atomic_store(sync + CH_FLAG + j * CH_LINE, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned polls = 0, ok = 1u;
        while (__hip_atomic_load(sync + CH_FLAG + g * CH_LINE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k) {
            __builtin_amdgcn_s_sleep(2);
            ++polls;
            if ((polls & 63u) == 0u && __hip_atomic_load(sync + CH_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0u; break; }
            if (polls > err.poll_limit) {                // ~ a second by default: the grid was not co-resident; give up loudly instead of hanging
                if (!__hip_atomic_exchange(sync + CH_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {      // first block to give up
                    if (err.counter) __hip_atomic_fetch_add(err.counter, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (err.hflag) __hip_atomic_store(err.hflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                ok = 0u;
                break;
            }
        }
        if (fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *s_ok = ok;
    }
    __syncthreads();
    return *s_ok != 0u;
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

// input strip: x rows (coalesced 16-byte loads, DEPTH in flight per thread), zero beyond the row / column range up to the next multiple
// of 16; thread = (row of a pass, 16-byte column): no division per element
template <int RT, int DEPTH>
__device__ __forceinline__ void load_x_strip(const ChainP &p, const Strip &s) {
    constexpr int FR = 16 * RT;
    const int c0 = p.c[0], kp = (c0 + 15) & ~15, v = kp >> 2, rpp = CH_THREADS / v, tr = s.tid / v, c4 = (s.tid - tr * v) * 4;
    if (tr < rpp) {
        for (int rb = tr; rb < FR; rb += DEPTH * rpp) {
            f32x4 buf[DEPTH];
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const int r = rb + u * rpp;
                buf[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (r < s.nvalid && c4 < c0) buf[u] = *reinterpret_cast<const f32x4 *>(p.x + (size_t)(s.row0 + r) * c0 + c4);
            }
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const int r = rb + u * rpp;
                if (r < FR) *reinterpret_cast<f32x4 *>(s.A + (size_t)r * s.lda + c4) = buf[u];
            }
        }
    }
}

// One pass over the strip of layer l once its BN is known (cf): the pre-BN values go to HBM (once; the backward reads them — issued
// AFTER the barrier, so their latency hides under the next layer instead of in front of the barrier), BN + activation in place (or
// straight to `out` for the last layer of an un-pooled chain); rows beyond the range stay zero (they must not enter the next
// layer's statistics).
template <int RT>
__device__ __forceinline__ void apply_strip(const ChainP &p, const Strip &s, int l) {
    constexpr int FR = 16 * RT;
    const int cout = p.c[l + 1], v = cout >> 2, rpp = CH_THREADS / v, tr = s.tid / v, c4 = (s.tid - tr * v) * 4;
    const float slope = p.slope[l];
    const bool to_out = l == p.nl - 1 && !p.pool_k, ywr = !(p.abl & 32);
    float *y = p.y[l];
    if (tr >= rpp) return;
    const f32x4 mu = *reinterpret_cast<const f32x4 *>(s.cf + c4), sc = *reinterpret_cast<const f32x4 *>(s.cf + CH_MAXC + c4),
                be = *reinterpret_cast<const f32x4 *>(s.cf + 2 * CH_MAXC + c4);
    for (int r = tr; r < FR; r += rpp) {
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
template <int NT, int RT, int XD>
__device__ __forceinline__ void layer_step(const ChainP &p, const Strip &s, int l, int wave, int i, int q) {
    const int kpad = (p.c[l] + 15) & ~15, w_ld = p.w_ld[l], lda = s.lda;
    const float *W = p.w[l];
    const bool w_vec = (w_ld & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;
    float *A = s.A;
    // 16-element chunks per weight batch (registers: 2 * BC * NT float4); 64-row strips with 3-4 column tiles per wave take ONE chunk: with
    // two the kernel spilled 9-36 registers at the 256 of two blocks per CU
    constexpr int BC = NT <= 2 ? 4 : (RT < 4 ? 2 : 1);
    f32x4 cur[BC * NT], nxt[BC * NT];
    load_wbatch<NT, BC>(cur, W, w_ld, w_vec, 0, kpad, wave, i, q, p.abl);
    if (l == 0) load_x_strip<RT, XD>(p, s); else apply_strip<RT>(p, s, l - 1);
    __syncthreads();
    f32x4 acc[RT][NT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < ((p.abl & 1) ? 0 : kpad); kb += 16 * BC) {
        if (kb + 16 * BC < kpad) load_wbatch<NT, BC>(nxt, W, w_ld, w_vec, kb + 16 * BC, kpad, wave, i, q, p.abl);
#pragma unroll
        for (int j = 0; j < BC; ++j) {
            const int k0 = kb + 16 * j;
            if (k0 < kpad) {
                f32x4 a[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) a[rt] = *reinterpret_cast<const f32x4 *>(A + (size_t)(rt * 16 + i) * lda + k0 + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
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
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) A[(size_t)(rt * 16 + 4 * q + e) * lda + (wave * NT + t) * 16 + i] = acc[rt][t][e];
    __syncthreads();
}

// W3: the instantiation that also takes 192-wide layers (three column tiles per wave).  Kept apart because with all four tile counts in
// one function the 64-row kernel sat 7 registers above the 256 of two blocks per CU (three 64-bit address bases spilled in the
// prologue); the W3 one loads the x strip four rows deep instead of eight to stay inside.  No layer of the network is 192 wide.
template <int RT, bool W3>
__global__ __launch_bounds__(CH_THREADS, 2) void chain_fwd_kernel(ChainP p) {
    constexpr int XD = (W3 && RT == 4) ? 4 : 8;
    constexpr int FR = 16 * RT;                          // rows of a strip: 16, 32 or 64 (chain_fwd_rows)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    Strip s;
    s.A = smem;                                          // [64][lda]
    s.cf = smem + (size_t)FR * p.lda;                // [3][CH_MAXC]: mean, scale, beta of the current layer
    double *red = reinterpret_cast<double *>(s.cf + 3 * CH_MAXC);     // [row groups][2][cout] partial column sums (<= 512 doubles)
    unsigned *s_ok = reinterpret_cast<unsigned *>(red + 512);         // the polling lane's verdict on the last barrier
    bool alive = true;                                   // false once a barrier was abandoned: no further barriers (results are invalid)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
    s.lda = p.lda; s.tid = tid;
    s.row0 = (long long)blockIdx.x * FR;
    s.nvalid = (int)((p.rows - s.row0) < FR ? (p.rows - s.row0) : FR);
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
            case 1: layer_step<1, RT, XD>(p, s, l, wave, i, q); break;
            case 2: layer_step<2, RT, XD>(p, s, l, wave, i, q); break;
            case 3: if constexpr (W3) layer_step<3, RT, XD>(p, s, l, wave, i, q); break;
            default: layer_step<4, RT, XD>(p, s, l, wave, i, q); break;
        }
        // column sums of the strip (rows beyond the range are exact zeros): 256 / cout row groups in parallel, combined through LDS,
        // then one atomic per block, column and moment
        {
            const int ngr = CH_THREADS / cout, gr = tid / cout, c = tid - gr * cout, rpg = FR / ngr;
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
        if (!(p.abl & 8) && alive) alive = grid_barrier(p.sync, (unsigned)(l + 1), G, tid, p.err, s_ok, p.abl);
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
    apply_strip<RT>(p, s, p.nl - 1);
    __syncthreads();

    if (p.pool_k) {      // max over groups of pool_k consecutive rows (pool_k divides 64: a group never leaves the strip); first k wins ties, NaN propagates
        const int cout = p.c[p.nl], v = cout >> 2, K = p.pool_k, ng = FR / K, gpp = CH_THREADS / v, tg = tid / v, c4 = (tid - tg * v) * 4;
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

// ---- backward of a whole chain in one launch (+ one reduction launch for the weight-gradient partials) ------------------------------
// Same resident grid and 64-row strips.  Going from the last layer to the first, per layer l (a_0 = x, y_l = a_{l-1} W_l^T,
// a_l = act(bn_l(y_l)); G_l = dL/da_l strip in LDS buffer P):
//   P1  t = G_l * act'(bn_l(y_l)) in place (y_l strip from HBM), column sums {sum t, sum t * xhat_l} -> atomics   -> GRID BARRIER
//   P2  totals: dbeta_l = S1, dgamma_l = S2 (block 0 writes them), m1 = S1 / rows, m2 = S2 / rows
//   P3  g^y = scale_l * (t - m1 - xhat_l * m2) in place (y_l strip re-read: L2-hot)           [BN backward, batch statistics]
//   P4  a_{l-1} strip = act(bn_{l-1}(y_{l-1})) (or the x strip) into LDS buffer Q
//   P5  dW_l partial of the strip = (g^y)^T a_{l-1} on MFMA (64-row contraction) -> this block's slab of dw_part
//   P6  G_{l-1} = g^y W_l on MFMA (W from L2) -> into Q, whose a_{l-1} is consumed; P and Q swap roles (gx for the first layer)
// chain_reduce_kernel then sums the slabs in block order (deterministic), all layers of the chain in one launch.
struct ChainBP {
    long long rows;
    int nl;
    int c[CH_MAXL + 1];
    int w_ld[CH_MAXL];
    int w_off[CH_MAXL];               // offset of layer l's [c_l][w_ld_l] block inside a slab of `tw` floats
    int tw;
    const float *x;
    const float *w[CH_MAXL], *y[CH_MAXL], *coef[CH_MAXL], *mi[CH_MAXL];
    float slope[CH_MAXL];
    const float *g;                   // dL/dout: [rows, c_L], or [rows / pool_k, c_L] with arg
    const unsigned char *arg;
    int pool_k;
    float *gx;                        // [rows, c0] or nullptr
    float *dw_part;                   // [grid][tw]
    float *dgamma[CH_MAXL], *dbeta[CH_MAXL];
    double *sums;                     // [nl][CH_REP][2 * smax], zero on entry
    int smax;
    unsigned *sync;
    ChainErr err;
    int ldp, ldq;
    int abl;
};

struct BStrip {
    float *P, *Q, *tab;
    int ldp, ldq, tid, nvalid;
    long long row0;
};

// RB = rows of a strip (64).  P / Q of the phase functions = the kernel's two strips in their current roles (gradient / other).

// P1: see above.  `first`: G_L comes from p.g (dense, or un-pooled on the fly from the arg-max bytes) instead of P.
template <int RB>
__device__ __forceinline__ void bwd_p1(const ChainBP &p, const BStrip &s, int l, bool first) {
    const int c = p.c[l + 1], v = c >> 2, rpp = CH_THREADS / v, tr = s.tid / v, c4 = (s.tid - tr * v) * 4;
    float *red = s.Q;                                    // [rpp][2][c] floats (Q is free until P4)
    if (tr < rpp) {
        const float *cf = p.coef[l];
        const f32x4 mu = *reinterpret_cast<const f32x4 *>(cf + c4), sc = *reinterpret_cast<const f32x4 *>(cf + c + c4),
                    be = *reinterpret_cast<const f32x4 *>(cf + 2 * c + c4), inv = *reinterpret_cast<const f32x4 *>(p.mi[l] + c + c4);
        const float slope = p.slope[l];
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        for (int rb = tr; rb < RB; rb += 8 * rpp) {
            f32x4 yv[8], gv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = rb + u * rpp;
                yv[u] = f32x4{0.f, 0.f, 0.f, 0.f}; gv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (r < s.nvalid) {
                    yv[u] = *reinterpret_cast<const f32x4 *>(p.y[l] + (size_t)(s.row0 + r) * c + c4);
                    if (first) {
                        if (p.pool_k) {
                            const long long row = s.row0 + r, grp = row / p.pool_k;
                            const unsigned char k = (unsigned char)(row - grp * p.pool_k);
                            const f32x4 t = *reinterpret_cast<const f32x4 *>(p.g + (size_t)grp * c + c4);
                            const uchar4 a = *reinterpret_cast<const uchar4 *>(p.arg + (size_t)grp * c + c4);
                            gv[u] = f32x4{a.x == k ? t[0] : 0.f, a.y == k ? t[1] : 0.f, a.z == k ? t[2] : 0.f, a.w == k ? t[3] : 0.f};
                        } else {
                            gv[u] = *reinterpret_cast<const f32x4 *>(p.g + (size_t)(s.row0 + r) * c + c4);
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = rb + u * rpp;
                if (r < RB) {
                    f32x4 g = first ? gv[u] : *reinterpret_cast<const f32x4 *>(s.P + (size_t)r * s.ldp + c4);
                    f32x4 t;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float z = (yv[u][e] - mu[e]) * sc[e] + be[e];
                        t[e] = r < s.nvalid ? (z > 0.f ? g[e] : g[e] * slope) : 0.f;
                        s1[e] += t[e];
                        s2[e] += t[e] * ((yv[u][e] - mu[e]) * inv[e]);
                    }
                    *reinterpret_cast<f32x4 *>(s.P + (size_t)r * s.ldp + c4) = t;
                }
            }
        }
        *reinterpret_cast<f32x4 *>(red + (size_t)(tr * 2) * c + c4) = s1;
        *reinterpret_cast<f32x4 *>(red + (size_t)(tr * 2 + 1) * c + c4) = s2;
    }
    __syncthreads();
    if (s.tid < c && !(p.abl & 4)) {
        double a = 0.0, b = 0.0;
        for (int g2 = 0; g2 < rpp; ++g2) { a += (double)red[(size_t)(g2 * 2) * c + s.tid]; b += (double)red[(size_t)(g2 * 2 + 1) * c + s.tid]; }
        double *sums = p.sums + ((size_t)l * CH_REP + (blockIdx.x % CH_REP)) * 2 * p.smax;
        atomicAdd(sums + s.tid, a);
        atomicAdd(sums + p.smax + s.tid, b);
    }
}

// P2: totals of the BN-backward sums -> tab[0][c] = m1, tab[1][c] = m2; block 0 writes dbeta / dgamma
__device__ __forceinline__ void bwd_p2(const ChainBP &p, const BStrip &s, int l) {
    const int c = p.c[l + 1];
    if (s.tid < c) {
        double *sl = p.sums + (size_t)l * CH_REP * 2 * p.smax;
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int r = 0; r < CH_REP; ++r) {
            a += __hip_atomic_load(sl + (size_t)r * 2 * p.smax + s.tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            b += __hip_atomic_load(sl + (size_t)r * 2 * p.smax + p.smax + s.tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s.tab[s.tid] = (float)(a / (double)p.rows);
        s.tab[256 + s.tid] = (float)(b / (double)p.rows);
        if (blockIdx.x == 0) { p.dbeta[l][s.tid] = (float)a; p.dgamma[l][s.tid] = (float)b; }
    }
    __syncthreads();
}

// P3: g^y in place in P
template <int RB>
__device__ __forceinline__ void bwd_p3(const ChainBP &p, const BStrip &s, int l) {
    const int c = p.c[l + 1], v = c >> 2, rpp = CH_THREADS / v, tr = s.tid / v, c4 = (s.tid - tr * v) * 4;
    if (tr >= rpp) return;
    const f32x4 mu = *reinterpret_cast<const f32x4 *>(p.coef[l] + c4), sc = *reinterpret_cast<const f32x4 *>(p.coef[l] + c + c4),
                inv = *reinterpret_cast<const f32x4 *>(p.mi[l] + c + c4), m1 = *reinterpret_cast<const f32x4 *>(s.tab + c4),
                m2 = *reinterpret_cast<const f32x4 *>(s.tab + 256 + c4);
    for (int rb = tr; rb < RB; rb += 8 * rpp) {
        f32x4 yv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = rb + u * rpp;
            yv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r < s.nvalid) yv[u] = *reinterpret_cast<const f32x4 *>(p.y[l] + (size_t)(s.row0 + r) * c + c4);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = rb + u * rpp;
            if (r < RB) {
                f32x4 t = *reinterpret_cast<const f32x4 *>(s.P + (size_t)r * s.ldp + c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = r < s.nvalid ? sc[e] * (t[e] - m1[e] - ((yv[u][e] - mu[e]) * inv[e]) * m2[e]) : 0.f;
                *reinterpret_cast<f32x4 *>(s.P + (size_t)r * s.ldp + c4) = t;
            }
        }
    }
}

// P4: the layer's input strip into Q: act(bn_{l-1}(y_{l-1})) for l > 0, the x strip (zero beyond its columns up to a multiple of 16) for l = 0
template <int RB>
__device__ __forceinline__ void bwd_p4(const ChainBP &p, const BStrip &s, int l) {
    const int c = p.c[l], kp = (c + 15) & ~15, v = kp >> 2, rpp = CH_THREADS / v, tr = s.tid / v, c4 = (s.tid - tr * v) * 4;
    if (tr >= rpp) return;
    const float *src = l ? p.y[l - 1] : p.x;
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, sc = mu, be = mu;
    const float slope = l ? p.slope[l - 1] : 1.f;
    if (l && c4 < c) {
        const float *cf = p.coef[l - 1];
        mu = *reinterpret_cast<const f32x4 *>(cf + c4); sc = *reinterpret_cast<const f32x4 *>(cf + c + c4); be = *reinterpret_cast<const f32x4 *>(cf + 2 * c + c4);
    }
    for (int rb = tr; rb < RB; rb += 8 * rpp) {
        f32x4 buf[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = rb + u * rpp;
            buf[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r < s.nvalid && c4 < c) buf[u] = *reinterpret_cast<const f32x4 *>(src + (size_t)(s.row0 + r) * c + c4);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = rb + u * rpp;
            if (r < RB) {
                f32x4 a = buf[u];
                if (l) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = (r < s.nvalid && c4 < c) ? act((a[e] - mu[e]) * sc[e] + be[e], slope) : 0.f;
                }
                *reinterpret_cast<f32x4 *>(s.Q + (size_t)r * s.ldq + c4) = a;
            }
        }
    }
}

// ---- P5 / P6 and the kernel: run-time tile loops, ping-pong LDS strips, two blocks per CU -------------------------------------------
// (Round 3 ran these phases as one fully unrolled function over all (input-gradient tiles x weight-gradient tiles) variants with the
// input gradient's accumulators held across the weight gradient: the whole register file (256 + 256), 158 spilled dwords per lane, one
// block per CU — the 456-block level-3 chain did not fit.  Measured against this form on one box (tools/time_chain.py, forward +
// backward, us): 14 848 x 128-64-64 164.7 vs 162.2, 14 592 x 128-64 125.6 vs 123.9, level 3 not taken vs 183.5 (layer kernels 190.1);
// the 256-wide chains 226.9 vs 260.8 — those go back to the layer kernels (235), see chain_bwd_fits.)
//   P1 .. P4 as above (G = gradient strip, O = the other strip),
//   weight gradient: a wave walks its output-channel tiles; per tile the 16 A fragments of g^y^T (LDS words) stay in registers, per
//     input-channel tile 16 B words of a_in and 16 MFMAs -> the block's slab,
//   sync (O is free now: a_in was only read by the weight gradient),
//   input gradient: a wave walks the input-channel tiles w, w+4, ...; per tile 4 accumulators, g^y as float4 A operands from G, W as
//     dwords from L2 with the next 16-chunk requested under the current one's MFMAs; the finished tile goes STRAIGHT into O (or to
//     gx for the first layer) — no accumulators waiting for the other phase, no copy-back phase,
//   sync, swap (G, O).
// 213 registers, no scratch; LDS = 2 strips of the widest tensor: two blocks per CU for chains up to 128 wide (level 3, the resampling
// set conv, the narrow up-convolution).
template <int RB>
__device__ __forceinline__ void bwd_wgrad(const ChainBP &p, const float *__restrict__ G, const float *__restrict__ O, int ld, int l, int wave,
                                           int i, int q) {
    const int cin_p = (p.c[l] + 15) & ~15, ntiles = cin_p >> 4, cout = p.c[l + 1], w_ld = p.w_ld[l], ntm = cout >> 6;
    float *part = p.dw_part + (size_t)blockIdx.x * p.tw + p.w_off[l];
    for (int tt = 0; tt < ntm; ++tt) {
        const int m0 = (wave * ntm + tt) * 16;
        float a[RB / 16][4];
#pragma unroll
        for (int kc = 0; kc < RB / 16; ++kc)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[kc][e] = G[(size_t)(16 * kc + 4 * q + e) * ld + m0 + i];
        for (int nt = 0; nt < ntiles; ++nt) {
            float b[RB / 16][4];
#pragma unroll
            for (int kc = 0; kc < RB / 16; ++kc)
#pragma unroll
                for (int e = 0; e < 4; ++e) b[kc][e] = O[(size_t)(16 * kc + 4 * q + e) * ld + nt * 16 + i];
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < RB / 16; ++kc)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kc][e], b[kc][e], acc, 0, 0, 0);
            const int n = nt * 16 + i;
            if (n < w_ld) {
#pragma unroll
                for (int e = 0; e < 4; ++e) part[(size_t)(m0 + 4 * q + e) * w_ld + n] = acc[e];
            }
        }
    }
}

template <int RB>
__device__ __forceinline__ void bwd_dgrad(const ChainBP &p, const float *__restrict__ G, float *__restrict__ O, int ld, int l, int wave, int i,
                                           int q, int nvalid, long long row0) {
    constexpr int RT = RB / 16;
    const int cin = p.c[l], cin_p = (cin + 15) & ~15, ntiles = cin_p >> 4, cout = p.c[l + 1], w_ld = p.w_ld[l];
    const float *W = p.w[l];
    for (int nt = wave; nt < ntiles; nt += 4) {
        const int n = nt * 16 + i;
        const bool n_ok = n < w_ld;
        auto loadw = [&](int k0, float (&b)[4]) {
#pragma unroll
            for (int e = 0; e < 4; ++e) b[e] = n_ok ? W[(size_t)(k0 + 4 * q + e) * w_ld + n] : 0.f;
        };
        float cur[4], nxt[4];
        loadw(0, cur);
        f32x4 acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < cout; k0 += 16) {
            if (k0 + 16 < cout) loadw(k0 + 16, nxt);
            f32x4 a[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) a[rt] = *reinterpret_cast<const f32x4 *>(G + (size_t)(rt * 16 + i) * ld + k0 + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][e], cur[e], acc[rt], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) cur[e] = nxt[e];
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = rt * 16 + 4 * q + e;
                if (l) O[(size_t)r * ld + n] = acc[rt][e];
                else if (r < nvalid && n < cin) p.gx[(size_t)(row0 + r) * cin + n] = acc[rt][e];
            }
    }
}

template <int RB>
__global__ __launch_bounds__(CH_THREADS, 2) void chain_bwd_kernel(ChainBP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ld = p.ldp;                                   // both strips: rows of the widest tensor of the chain (+4)
    float *G = smem, *O = smem + (size_t)RB * ld;
    BStrip s;
    s.tab = smem + 2 * (size_t)RB * ld;                     // [2][256]: m1, m2 of the current layer
    unsigned *s_ok = reinterpret_cast<unsigned *>(s.tab + 512);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
    s.ldp = ld; s.ldq = ld; s.tid = tid;
    s.row0 = (long long)blockIdx.x * RB;
    s.nvalid = (int)((p.rows - s.row0) < RB ? (p.rows - s.row0) : RB);
    const unsigned nblocks = gridDim.x;
    bool alive = true;
    for (int l = p.nl - 1; l >= 0; --l) {
        s.P = G; s.Q = O;
        bwd_p1<RB>(p, s, l, l == p.nl - 1);
        if (!(p.abl & 8) && alive) alive = grid_barrier(p.sync, (unsigned)(p.nl - l), nblocks, tid, p.err, s_ok, p.abl); else __syncthreads();
        bwd_p2(p, s, l);
        bwd_p3<RB>(p, s, l);
        bwd_p4<RB>(p, s, l);
        __syncthreads();
        if (!(p.abl & 1)) bwd_wgrad<RB>(p, G, O, ld, l, wave, i, q);
        __syncthreads();                                    // a_in (O) is consumed: the input gradient may overwrite it
        if ((l || p.gx) && !(p.abl & 2)) bwd_dgrad<RB>(p, G, O, ld, l, wave, i, q, s.nvalid, s.row0);
        __syncthreads();
        float *t = G; G = O; O = t;
    }
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(p.sync + CH_EXIT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == nblocks - 1)
            for (int j = 0; j <= 2 * CH_NG + 1; ++j) __hip_atomic_store(p.sync + j * CH_LINE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// dW of every layer of the chain = sum over the blocks' slabs, in block order: 16 float4 columns x 16 slab lanes per block
__global__ __launch_bounds__(256) void chain_reduce_kernel(int nparts, int n4, const float4 *__restrict__ parts, float4 *__restrict__ out) {
    __shared__ float4 red[16][16];
    const int tx = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int o = blockIdx.x * 16 + tx;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (o < n4) {
        int b = pl;
        for (; b + 48 < nparts; b += 64) {
            const float4 v0 = parts[(size_t)b * n4 + o], v1 = parts[(size_t)(b + 16) * n4 + o];
            const float4 v2 = parts[(size_t)(b + 32) * n4 + o], v3 = parts[(size_t)(b + 48) * n4 + o];
            a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
            a.x += v1.x; a.y += v1.y; a.z += v1.z; a.w += v1.w;
            a.x += v2.x; a.y += v2.y; a.z += v2.z; a.w += v2.w;
            a.x += v3.x; a.y += v3.y; a.z += v3.z; a.w += v3.w;
        }
        for (; b < nparts; b += 16) {
            const float4 v = parts[(size_t)b * n4 + o];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    red[pl][tx] = a;
    __syncthreads();
    if (pl == 0 && o < n4) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 16; ++j) { const float4 v = red[j][tx]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        out[o] = t;
    }
}

// LDS rows of the backward: P holds gradients of the layer outputs, Q the layer inputs
void chain_bwd_ld(int nl, const int *widths, int &ldp, int &ldq) {
    int mp = 0, mq = 0;
    for (int l = 1; l <= nl; ++l) mp = widths[l] > mp ? widths[l] : mp;
    for (int l = 0; l < nl; ++l) { const int k = (widths[l] + 15) & ~15; mq = k > mq ? k : mq; }
    ldp = mp + 4; ldq = (mq < 32 ? 32 : mq) + 4;
}

int chain_cus();

// ---- residency: what the launcher checks a grid against (VERDICT r3 #1b) -----------------------------------------------------------
// Blocks of kernel `kind` (0 / 1 / 2: chain_fwd_kernel<1 / 2 / 4>, 3: chain_bwd_kernel<64>) the CURRENT device
// holds at once with `lds` bytes of dynamic LDS: hipOccupancyMaxActiveBlocksPerMultiprocessor (registers, LDS, waves) x CUs of that
// device, capped by the blocks per CU the kernels are written for (launch bounds: two) and by the 256-thread admission rule of MI355X_MICROARCH.md (min(API, 8, 800 / (ceil(sgpr / 16) * 16 + 16)): 6 at the 106
// SGPRs of these kernels — above the caps).  Cached per (device, kind, lds).  A plain launch of a grid within this number has the
// same residency as a cooperative launch of it (same guide: hipLaunchCooperativeKernel adds only the check made here, at +15-19 us
// per launch); what the query cannot see — a CU mask, another process on the GPU, a second chain launch on another stream — is what
// the barrier's poll limit and the error words are for.
constexpr int CH_MAX_DEV = 16;
struct ResKey { int kind; size_t lds; int blocks; };
struct DevState {
    int cus = 0;
    bool attr_set = false;
    ResKey cache[32];
    int ncache = 0;
    ChainErr err{nullptr, nullptr, 0};
};
DevState g_dev[CH_MAX_DEV];

DevState *chain_dev() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CH_MAX_DEV) return nullptr;
    DevState &d = g_dev[dev];
    if (!d.cus) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return nullptr;
        d.cus = prop.multiProcessorCount;
    }
    if (!d.attr_set) {                 // (function attributes are per device)
#define CH_SET_LDS(K) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(K), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
        CH_SET_LDS((chain_fwd_kernel<4, false>)); CH_SET_LDS((chain_fwd_kernel<2, false>)); CH_SET_LDS((chain_fwd_kernel<1, false>));
        CH_SET_LDS((chain_fwd_kernel<4, true>)); CH_SET_LDS((chain_fwd_kernel<2, true>)); CH_SET_LDS((chain_fwd_kernel<1, true>));
#undef CH_SET_LDS
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(chain_bwd_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        d.attr_set = true;
    }
    return &d;
}

int chain_cus() { DevState *d = chain_dev(); return d ? d->cus : 0; }

int chain_resident(int kind, size_t lds) {
    DevState *d = chain_dev();
    if (!d || lds > 160 * 1024) return 0;
    for (int i = 0; i < d->ncache; ++i)
        if (d->cache[i].kind == kind && d->cache[i].lds == lds) return d->cache[i].blocks;
    const void *fn = kind == 0 ? reinterpret_cast<const void *>(chain_fwd_kernel<1, true>) : kind == 1 ? reinterpret_cast<const void *>(chain_fwd_kernel<2, true>)
                   : kind == 2 ? reinterpret_cast<const void *>(chain_fwd_kernel<4, true>) : reinterpret_cast<const void *>(chain_bwd_kernel<64>);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, CH_THREADS, lds) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
    const int cap = 2;
    per_cu = per_cu > cap ? cap : per_cu;
    const int blocks = per_cu * d->cus;
    if (d->ncache < 32) d->cache[d->ncache++] = ResKey{kind, lds, blocks};
    return blocks;
}

// diagnostic switch of tests/test_chain_gpu.py::test_chain_timeout_surfaces: accept grids that are NOT resident (the barrier then times out)
bool chain_force_nonresident() { const char *e = getenv("I2P_CHAIN_FORCE_NONRESIDENT"); return e && e[0] == '1'; }

ChainErr chain_err() {
    DevState *d = chain_dev();
    ChainErr e = d ? d->err : ChainErr{nullptr, nullptr, 0};
    const char *lim = getenv("I2P_CHAIN_POLL_LIMIT");       // polls before a barrier gives up (default 2^19 ~ one second)
    e.poll_limit = lim ? (unsigned)strtoul(lim, nullptr, 10) : (1u << 19);
    return e;
}

// The one-launch backward: both LDS strips as wide as the widest tensor of the chain; taken where TWO blocks per CU fit (chains up to
// 128 wide) and the grid is resident.  Chains under 8192 rows are not taken: the separate slab reduction launch costs them more than
// the layer-by-layer launches they would save (48 vs 38 us forward + backward on 928 .. 1824 rows, tools/time_chain.py); wider chains
// (one block per CU) measured slower than the layer kernels (260.8 vs 235 us, 14 848 x 128-128-256).
size_t chain_bwd_lds_bytes(int ldp, int ldq) { const int ld = ldp > ldq ? ldp : ldq; return ((size_t)2 * 64 * ld + 512 + 4) * sizeof(float); }
bool chain_bwd_fits(long long rows, int ldp, int ldq) {
    if (rows < 8192) return false;
    const size_t lds = chain_bwd_lds_bytes(ldp, ldq);
    if (2 * lds > 160 * 1024) return false;
    return (rows + 63) / 64 <= chain_resident(3, lds) || (chain_force_nonresident() && chain_cus() > 0);
}

int chain_cmax(int nl, const int *widths) {
    int m = 0;
    for (int l = 0; l <= nl; ++l) m = widths[l] > m ? widths[l] : m;
    return m;
}

size_t chain_lds_bytes(int cmax, int fr) {
    const int lda = ((cmax + 15) & ~15) + 4;
    return ((size_t)fr * lda + 3 * CH_MAXC) * sizeof(float) + 512 * sizeof(double) + 16;
}

// rows of a forward strip: the smallest of 16 / 32 / 64 whose grid still has at most one block per CU — few-row chains then spread over
// more CUs (928 rows: 13.0 -> 10.9 us, 1824: 13.1 -> 11.2, 7296 x 128: 14.7 -> 13.8 with 32-row strips, tools/time_chain.py) — while
// the 14 848-row chains stay on 64-row strips (32-row ones read W twice as often and double the barrier's arrivals: 58 -> 63 us);
// 64-row grids may also run two blocks per CU (what the occupancy query admits).  0 = the chain does not fit.
int chain_fwd_rows(long long rows, int cmax, int pool_k) {
    const int cus = chain_cus();
    if (cus <= 0) return 0;
    for (int fr = 16; fr <= 64; fr *= 2) {
        if (pool_k && fr % pool_k) continue;
        const long long blocks = (rows + fr - 1) / fr;
        const size_t lds = chain_lds_bytes(cmax, fr);
        const int resident = chain_resident(fr == 16 ? 0 : fr == 32 ? 1 : 2, lds);
        if (fr < 64) {
            if (blocks > cus || blocks > resident) continue;
            return fr;
        }
        if (blocks <= resident || (chain_force_nonresident() && lds <= 160 * 1024)) return fr;
    }
    return 0;
}

}  // namespace

// doubles of zeroed scratch i2p_chain_fwd needs for a chain of `nl` layers whose widest output is `cmax_out`
extern "C" long long i2p_chain_sums_len(int nl, int cmax_out) { return (long long)nl * CH_REP * 2 * cmax_out; }

// Error sinks of the CURRENT device (see ChainErr): a device fp32 counter (+1 per launch with a timed-out grid barrier — results of that
// launch are invalid) and a host-mapped word set to 1 at the same moment; both zeroed by the caller and kept alive; NULLs unregister.
// The per-launch error word lives in arena scratch that the next step clears.
extern "C" int i2p_chain_set_error_words(float *device_counter, unsigned *host_flag) {
    DevState *d = chain_dev();
    if (!d) return I2P_ERR_BAD_ARG;
    d->err.counter = device_counter; d->err.hflag = host_flag;
    return 0;
}

extern "C" int i2p_chain_resident_blocks(int kind, long long lds_bytes) {
    return (kind < 0 || kind > 3 || lds_bytes < 0) ? 0 : chain_resident(kind, (size_t)lds_bytes);
}

// uint32 words of zeroed scratch behind `sync`; word i2p_chain_sync_words() - 32 is the error word
extern "C" long long i2p_chain_sync_words(void) { return CH_SYNC_WORDS; }

extern "C" int i2p_chain_fwd_ok(long long rows, int nl, const int *widths, int pool_k) {
    if (rows <= 0 || nl < 1 || nl > CH_MAXL || !widths) return 0;
    if (widths[0] <= 0 || (widths[0] & 3) || widths[0] > CH_MAXC) return 0;
    for (int l = 1; l <= nl; ++l)
        if (widths[l] <= 0 || (widths[l] & 63) || widths[l] > 256) return 0;
    if (pool_k < 0 || pool_k > CH_ROWS || (pool_k && (CH_ROWS % pool_k || rows % pool_k))) return 0;
    return chain_fwd_rows(rows, chain_cmax(nl, widths), pool_k) ? 1 : 0;
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
    p.sync = sync; p.err = chain_err();
    { const char *e = getenv("I2P_CHAIN_ABL"); p.abl = e ? atoi(e) : 0; }     /* read per call: tools/time_chain.py switches it */
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
    const int fr = chain_fwd_rows(rows, cmax, pool_k);
    const size_t bytes = chain_lds_bytes(cmax, fr);
    const unsigned grid = (unsigned)((rows + fr - 1) / fr);
    bool w3 = false;
    for (int l = 1; l <= nl; ++l) w3 = w3 || widths[l] == 192;
#define CH_LAUNCH(RT) do { if (w3) hipLaunchKernelGGL((chain_fwd_kernel<RT, true>), dim3(grid), dim3(CH_THREADS), bytes, (hipStream_t)stream, p); \
                           else hipLaunchKernelGGL((chain_fwd_kernel<RT, false>), dim3(grid), dim3(CH_THREADS), bytes, (hipStream_t)stream, p); } while (0)
    if (fr == 16) CH_LAUNCH(1); else if (fr == 32) CH_LAUNCH(2); else CH_LAUNCH(4);
#undef CH_LAUNCH
    I2P_RETURN_LAUNCH_STATUS();
}

// ---- backward ---------------------------------------------------------------------------------------------------------------------
extern "C" int i2p_chain_bwd_ok(long long rows, int nl, const int *widths, int pool_k) {
    if (rows <= 0 || nl < 1 || nl > CH_MAXL || !widths) return 0;
    if (widths[0] <= 0 || (widths[0] & 3) || widths[0] > 256) return 0;
    for (int l = 1; l <= nl; ++l)
        if (widths[l] <= 0 || (widths[l] & 63) || widths[l] > 256) return 0;
    if (pool_k < 0 || pool_k > 255 || (pool_k && rows % pool_k)) return 0;
    int ldp, ldq;
    chain_bwd_ld(nl, widths, ldp, ldq);
    return chain_bwd_fits(rows, ldp, ldq) ? 1 : 0;
}

// floats of one block's weight-gradient slab (= of the reduced `dw` buffer): layer l's [widths[l+1]][w_ld[l]] block starts at the sum
// of the blocks before it; dw_part needs ceil(rows / 64) slabs
extern "C" long long i2p_chain_bwd_slab(int nl, const int *widths, const int *w_ld) {
    long long t = 0;
    for (int l = 0; l < nl; ++l) t += (long long)widths[l + 1] * w_ld[l];
    return t;
}

extern "C" int i2p_chain_bwd(long long rows, int nl, const int *widths, const int *w_ld, const float *x, const float *const *w,
                             const float *const *y, const float *const *coef, const float *const *mean_invstd, const float *slopes,
                             const float *g, const unsigned char *arg, int pool_k, float *gx, float *dw_part, float *dw,
                             float *const *dgamma, float *const *dbeta, double *sums, unsigned *sync, void *stream) {
    if (!i2p_chain_bwd_ok(rows, nl, widths, pool_k)) return I2P_ERR_BAD_ARG;
    if (!w_ld || !x || !w || !y || !coef || !mean_invstd || !slopes || !g || !dw_part || !dw || !dgamma || !dbeta || !sums || !sync ||
        (pool_k && !arg))
        return I2P_ERR_BAD_ARG;
    ChainBP p{};
    p.rows = rows; p.nl = nl; p.x = x; p.g = g; p.arg = arg; p.pool_k = pool_k; p.gx = gx; p.dw_part = dw_part; p.sums = sums; p.sync = sync; p.err = chain_err();
    { const char *e = getenv("I2P_CHAIN_ABL"); p.abl = e ? atoi(e) : 0; }
    p.c[0] = widths[0];
    int off = 0;
    for (int l = 0; l < nl; ++l) {
        if (w_ld[l] <= 0 || w_ld[l] > widths[l] || !w[l] || !y[l] || !coef[l] || !mean_invstd[l] || !dgamma[l] || !dbeta[l]) return I2P_ERR_BAD_ARG;
        p.c[l + 1] = widths[l + 1]; p.w_ld[l] = w_ld[l]; p.w_off[l] = off;
        off += widths[l + 1] * w_ld[l];
        p.w[l] = w[l]; p.y[l] = y[l]; p.coef[l] = coef[l]; p.mi[l] = mean_invstd[l]; p.slope[l] = slopes[l];
        p.dgamma[l] = dgamma[l]; p.dbeta[l] = dbeta[l];
        p.smax = widths[l + 1] > p.smax ? widths[l + 1] : p.smax;
    }
    p.tw = off;
    if ((off & 3) || ((reinterpret_cast<uintptr_t>(dw_part) | reinterpret_cast<uintptr_t>(dw)) & 15)) return I2P_ERR_BAD_ARG;
    chain_bwd_ld(nl, widths, p.ldp, p.ldq);
    const unsigned grid = (unsigned)((rows + 63) / 64);
    const size_t bytes = chain_bwd_lds_bytes(p.ldp, p.ldq);
    p.ldp = p.ldq = p.ldp > p.ldq ? p.ldp : p.ldq;
    hipLaunchKernelGGL(chain_bwd_kernel<64>, dim3(grid), dim3(CH_THREADS), bytes, (hipStream_t)stream, p);
    const int n4 = off >> 2;
    if (!i2p_defer_reduce(0, (int)grid, n4, dw_part, dw))
    hipLaunchKernelGGL(chain_reduce_kernel, dim3((n4 + 15) / 16), dim3(256), 0, (hipStream_t)stream, (int)grid, n4,
                       reinterpret_cast<const float4 *>(dw_part), reinterpret_cast<float4 *>(dw));
    I2P_RETURN_LAUNCH_STATUS();
}
