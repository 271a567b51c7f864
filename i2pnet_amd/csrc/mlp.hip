// Fused per-point linear layers (1x1 conv) on fp32 MFMA for gfx950.
//
// Every point-branch layer of the reference is  conv1x1 -> BatchNorm(batch statistics) ->
// (Leaky)ReLU  on a [B,N,K,C] tensor (src/projectPN/PPBackbone_center.py:10-51), run there as
// separate eager kernels with a full HBM round trip each.  Batch statistics force one global
// reduction per layer, so a layer cannot be fused with its own BN — but it CAN be fused with the
// BN + activation of the layer BEFORE it and with the statistics of its own output:
//
//   lin_fwd :  Y = act_in(bn_in(X)) . W^T          X [rows,Cin] pre-BN output of the previous layer
//              sums += { sum Y, sum Y^2 }           (fp64, replica-spread atomics, as bn_act.hip)
//   lin_bwd :  G  = bn_out_backward(dZ, Y)          formed on load from dZ (grad w.r.t. BN output
//                                                   pre-activation... see below) and Y
//              dW += G^T . X'      (wgrad)          X' = act_in(bn_in(X)) recomputed on load
//              dZin = (G . W) * act_in'(z_in)       (dgrad, continues into the previous BN)
//              dsums_in += { sum dZin, sum dZin * xhat_in }
//
// so a chain of L layers costs L kernels forward (read X, write Y once each) and L backward
// (read dZ, Y, X; write dZin), with no materialised BN outputs, activations or x-hats.
//
// GEMM core: v_mfma_f32_32x32x2_f32 (exact fp32, 256 FLOP/clk/CU), W stationary in LDS
// ([Cout][Cin+1] floats, +1 pad => conflict-free ds_read_b32 fragments), 128-row activation
// tiles staged through LDS with the BN/activation transform applied on the way in, 4 waves x
// (32 rows x Cout) accumulators.
#include "common.h"
#include <cstdlib>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int THREADS = 256;
constexpr int REP = I2P_BN_REPLICAS;

__device__ __forceinline__ float act_apply(float z, float slope) { return z > 0.f ? z : z * slope; }

struct LinFwdParams {
    long long rows;
    int cin, cout;          // logical sizes
    int cin_p, cout_p;      // cin rounded up to even, cout rounded up to 32
    int ldk;                // LDS row stride in floats (cin_p + 1)
    const float *x;         // [rows, cin]
    const float *in_coef;   // [3][cin] mean, scale(=invstd*gamma), beta of the BN in front, or nullptr
    float slope_in;         // activation in front (1 = none)
    const float *w;         // [cout, cin]
    float *y;               // [rows, y_ld], this launch writes columns [ch_off, ch_off+cout)
    double *sums;           // [REP][2*cout_total] (zeroed by caller) or nullptr
    int y_ld, ch_off, cout_total;
    // pair mode (cost volume, all point x pixel pairs): rows = (b, n, k) over B x N x M,
    //   input row = pair_f[b,n,:] * x[b,k,:]   (x is then the [B,M,cin] pixel tensor)
    //   y[r,:]   += bias_n[b,n,:] + bias_k[b,k,:]
    int w_vec;              // w is 16-byte aligned (set by the launcher): float4 weight staging
    int nslice;             // output-channel slices over blockIdx.y (small layers; set by the launcher)
    int ablate;             // diagnostic only (I2P_LIN_ABLATE): 1 no MFMA loop, 2 no stores, 4 no stats, 8 no staging, 32 no streaming hints
    const float *pair_f;    // [B,N,cin] or nullptr (plain mode)
    const float *bias_n;    // [B,N,cout_total] or nullptr
    const float *bias_k;    // [B,M,cout_total] or nullptr
    int pair_N, pair_M;
    // dgrad mode (MODE 2 of lin_fwd2): x = gz [rows,cin] of the layer BEHIND, x2 = its pre-BN output,
    // g_coef [5][cin] = {mean(gz), mean(gz*xhat), scale, mean, invstd}: the tile fed to the MFMAs is
    // g^y = scale*(gz - m1 - xhat*m2).  w_transposed: Ws[o][k] = w[k*cout + o].  Store phase:
    // out = acc * act'(z_prev) with z_prev from ex (pre-BN tensor in front) and e_coef/e_mi; sums
    // receive { sum out, sum out*xhat_prev }.
    const float *x2, *g_coef;   // g_coef: non-null = a BN sits behind (its constants are formed in the prologue from the raw sources below)
    const double *g_dsums;      // [REP][2*cin] replicated {sum gz, sum gz*xhat} of that BN
    const float *g_oc, *g_omi;  // its coef [3][cin] (mean, scale, beta) and mean_invstd [2*cin]
    long long g_rows;
    float g_slope;              // gz arrives as dL/da of the layer behind (a = act(z), slope g_slope): apply act'(z) on load; 1 = gz is dL/dz
    int w_transposed;
    const float *ex, *e_coef, *e_mi;
    float e_slope;
    // two-source mode (input = [x | xb] along channels, no concatenated tensor): the first split_c
    // channels come from x [rows, split_c] with in_coef/slope_in, the rest from xb [rows, cin-split_c]
    // with in_coef_b/slope_b.  In DGRAD mode the OUTPUT is split the same way: columns < split_c go
    // to y/ex/e_coef/sums, the rest to yb/exb/e_coef_b/sums_b, and e_add [rows, cout-split_c] (dL/da
    // arriving from another consumer of that activation) is added before the activation derivative.
    int split_c;
    const float *xb, *in_coef_b;
    float slope_b;
    float *yb; const float *exb, *e_coef_b, *e_mi_b, *e_add; double *sums_b; float e_slope_b;
    // BN finalisation by the LAST block to finish (forward modes): coef [3][cout_total] = mean, invstd*gamma, beta and
    // mean_invstd [2*cout_total] of the output's batch statistics, formed from `sums` once every block has added its
    // share (ticket counter, zero on entry and reset on exit) — a separate 1-block launch per layer otherwise.
    unsigned *fin_counter; const float *fin_gamma, *fin_beta; float fin_eps; float *fin_coef, *fin_mi;
};

// (b,n,k) bookkeeping of pair mode without per-element 64-bit divisions: one division pair per
// TILE (row0 is block-uniform), then rows inside the tile walk forward by wrap-around.
struct PairTile { int bn0, k0, b0, n0; };
__device__ __forceinline__ PairTile pair_tile(long long row0, int N, int M) {
    PairTile t;
    t.bn0 = (int)(row0 / M); t.k0 = (int)(row0 - (long long)t.bn0 * M);
    t.b0 = t.bn0 / N; t.n0 = t.bn0 - t.b0 * N;
    return t;
}
__device__ __forceinline__ void pair_row(const PairTile &t, int d, int N, int M, int &bn, int &bk) {
    int k = t.k0 + d, n = t.n0, b = t.b0;
    bn = t.bn0;
    while (k >= M) { k -= M; ++bn; if (++n == N) { n = 0; ++b; } }
    bk = b * M + k;
}

// one 32x32 C/D fragment: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
__device__ __forceinline__ int frag_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// ---- activation-tile staging: global -> registers (issued early) -> transform -> LDS -------------
// vec4 path (cin % 4 == 0, cin <= 128): 128 rows x cin/4 float4 chunks = <= 16 chunks per thread.
constexpr int MAXCH = 16;

struct StageGeom {
    int c4n;        // float4 chunks per row
    int nchunk;     // TILE_R * c4n
    int shift;      // log2(c4n) if c4n is a power of two, else -1
};

__device__ __forceinline__ void chunk_rc(const StageGeom &g, int i, int &r, int &c4) {
    if (g.shift >= 0) { r = i >> g.shift; c4 = i & (g.c4n - 1); }
    else { r = i / g.c4n; c4 = i - r * g.c4n; }
}

template <int TILE_R>
__device__ __forceinline__ void stage_fetch(const LinFwdParams &p, const StageGeom &g, long long row0, int tid,
                                            float4 (&v)[MAXCH]) {
    PairTile pt; pt.bn0 = pt.k0 = pt.b0 = pt.n0 = 0;
    if (p.pair_f) pt = pair_tile(row0, p.pair_N, p.pair_M);
#pragma unroll
    for (int u = 0; u < MAXCH; ++u) {
        const int i = tid + u * THREADS;
        int r, c4; chunk_rc(g, i, r, c4);
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < g.nchunk && row0 + r < p.rows) {
            long long src = row0 + r;
            if (p.pair_f) { int bn, bk; pair_row(pt, r, p.pair_N, p.pair_M, bn, bk); src = bk; }
            v[u] = *reinterpret_cast<const float4 *>(p.x + (size_t)src * p.cin + c4 * 4);
        }
    }
}

template <int TILE_R>
__device__ __forceinline__ void stage_commit(const LinFwdParams &p, const StageGeom &g, long long row0, int tid,
                                             const float4 (&v)[MAXCH], float *As) {
    PairTile pt; pt.bn0 = pt.k0 = pt.b0 = pt.n0 = 0;
    if (p.pair_f) pt = pair_tile(row0, p.pair_N, p.pair_M);
#pragma unroll
    for (int u = 0; u < MAXCH; ++u) {
        const int i = tid + u * THREADS;
        if (i < g.nchunk) {
            int r, c4; chunk_rc(g, i, r, c4);
            float4 t = v[u];
            if (p.pair_f && row0 + r < p.rows) {
                int bn, bk; pair_row(pt, r, p.pair_N, p.pair_M, bn, bk);
                const float4 f = *reinterpret_cast<const float4 *>(p.pair_f + (size_t)bn * p.cin + c4 * 4);
                t.x *= f.x; t.y *= f.y; t.z *= f.z; t.w *= f.w;
            }
            if (p.in_coef && row0 + r < p.rows) {      // padding rows stay exactly 0 (they feed nothing)
                const float4 m = *reinterpret_cast<const float4 *>(p.in_coef + c4 * 4);
                const float4 s = *reinterpret_cast<const float4 *>(p.in_coef + p.cin + c4 * 4);
                const float4 b = *reinterpret_cast<const float4 *>(p.in_coef + 2 * p.cin + c4 * 4);
                t.x = act_apply((t.x - m.x) * s.x + b.x, p.slope_in);
                t.y = act_apply((t.y - m.y) * s.y + b.y, p.slope_in);
                t.z = act_apply((t.z - m.z) * s.z + b.z, p.slope_in);
                t.w = act_apply((t.w - m.w) * s.w + b.w, p.slope_in);
            }
            float *dst = As + r * p.ldk + c4 * 4;
            dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;
        }
    }
}

template <int TILE_R, int NT>
__global__ __launch_bounds__(THREADS) void lin_fwd_kernel(LinFwdParams p) {
    extern __shared__ float smem[];
    float *Ws = smem;                                   // [cout_p][ldk]
    float *As = smem + (size_t)p.cout_p * p.ldk;        // [TILE_R][ldk]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    static_assert(TILE_R / 4 == 32, "one 32-row MFMA strip per wave");

    // ---- stationary weights ------------------------------------------------------------------
    for (int i = tid; i < p.cout_p * p.cin_p; i += THREADS) {
        const int co = i / p.cin_p, ci = i - co * p.cin_p;
        Ws[co * p.ldk + ci] = (co < p.cout && ci < p.cin) ? p.w[(size_t)co * p.cin + ci] : 0.f;
    }
    if (p.cin_p != p.cin)                               // the zero K-padding column of the activation tile
        for (int r = tid; r < TILE_R; r += THREADS) As[r * p.ldk + p.cin] = 0.f;

    double ssum[NT], ssq[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { ssum[j] = 0.0; ssq[j] = 0.0; }

    const long long ntiles = (p.rows + TILE_R - 1) / TILE_R;
    const bool vec4 = (p.cin & 3) == 0 && (p.cin >> 2) * TILE_R <= MAXCH * THREADS;
    StageGeom g;
    g.c4n = p.cin >> 2; g.nchunk = TILE_R * g.c4n;
    g.shift = (g.c4n > 0 && (g.c4n & (g.c4n - 1)) == 0) ? (31 - __clz(g.c4n)) : -1;

    float4 pf[MAXCH];
    long long tile = blockIdx.x;
    if (vec4 && tile < ntiles) stage_fetch<TILE_R>(p, g, tile * TILE_R, tid, pf);
    for (; tile < ntiles; tile += gridDim.x) {
        const long long row0 = tile * TILE_R;
        __syncthreads();                                // previous tile's fragments are consumed (and Ws is written)
        if (vec4) {
            if (!(p.ablate & 8)) stage_commit<TILE_R>(p, g, row0, tid, pf, As);
        } else {                                        // generic channel counts: scalar, no prefetch
            for (int i = tid; i < TILE_R * p.cin; i += THREADS) {
                const int r = i / p.cin, c = i - r * p.cin;
                float v = 0.f;
                if (row0 + r < p.rows) {
                    v = p.x[(size_t)(row0 + r) * p.cin + c];
                    if (p.in_coef)
                        v = act_apply((v - p.in_coef[c]) * p.in_coef[p.cin + c] + p.in_coef[2 * p.cin + c], p.slope_in);
                }
                As[r * p.ldk + c] = v;
            }
        }
        __syncthreads();
        // next tile's loads fly during this tile's MFMA loop
        if (vec4 && tile + gridDim.x < ntiles && !(p.ablate & 8)) stage_fetch<TILE_R>(p, g, (tile + gridDim.x) * TILE_R, tid, pf);

        // ---- 32 x (32*NT) strip per wave on the matrix cores -----------------------------------
        f32x16 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        const float *arow = As + (wave * 32 + (lane & 31)) * p.ldk + (lane >> 5);
        const float *brow = Ws + (lane & 31) * p.ldk + (lane >> 5);
        // software-pipelined: the fragments of step k+1 are in flight while step k's MFMAs issue
        // (hipcc otherwise emits ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma per MFMA)
        float a_cur = arow[0], b_cur[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) b_cur[j] = brow[(size_t)j * 32 * p.ldk];
#pragma unroll 2
        for (int kk = 0; kk < ((p.ablate & 1) ? 2 : p.cin_p); kk += 2) {
            const int kn = (kk + 2 < p.cin_p) ? kk + 2 : kk;
            const float a_nxt = arow[kn];
            float b_nxt[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) b_nxt[j] = brow[(size_t)j * 32 * p.ldk + kn];
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[j], acc[j], 0, 0, 0);
            // keep the order "issue next step's NT+1 LDS reads, then this step's NT MFMAs"
            __builtin_amdgcn_sched_group_barrier(0x100, NT + 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
            a_cur = a_nxt;
#pragma unroll
            for (int j = 0; j < NT; ++j) b_cur[j] = b_nxt[j];
        }

        // ---- epilogue: store Y, accumulate per-channel statistics -------------------------------
        PairTile ept; ept.bn0 = ept.k0 = ept.b0 = ept.n0 = 0;
        if (p.bias_n) ept = pair_tile(row0, p.pair_N, p.pair_M);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int ch = j * 32 + (lane & 31);
            if (ch < p.cout) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const long long r = row0 + wave * 32 + frag_row(e, lane);
                    if (r < p.rows) {
                        float v = acc[j][e];
                        if (p.bias_n) {
                            int bn, bk; pair_row(ept, wave * 32 + frag_row(e, lane), p.pair_N, p.pair_M, bn, bk);
                            v = (v + p.bias_n[(size_t)bn * p.cout_total + p.ch_off + ch]) +
                                p.bias_k[(size_t)bk * p.cout_total + p.ch_off + ch];
                        }
                        if (!(p.ablate & 2)) p.y[(size_t)r * p.y_ld + p.ch_off + ch] = v;
                        if (!(p.ablate & 4)) { ssum[j] += v; ssq[j] += (double)v * v; }
                    }
                }
            }
        }
    }

    // ---- block reduction of the statistics, one atomic per channel per block --------------------
    if (p.sums) {
        __syncthreads();
        double *red = reinterpret_cast<double *>(smem);          // [THREADS][2*NT] doubles (<= 32 KB)
#pragma unroll
        for (int j = 0; j < NT; ++j) { red[(size_t)tid * 2 * NT + j] = ssum[j]; red[(size_t)tid * 2 * NT + NT + j] = ssq[j]; }
        __syncthreads();
        for (int i = tid; i < 2 * NT * 32; i += THREADS) {       // i -> (which, j, col)
            const int which = i / (NT * 32), rem = i - which * NT * 32, j = rem >> 5, col = rem & 31;
            const int ch = j * 32 + col;
            if (ch < p.cout) {
                double a = 0.0;
                for (int w = 0; w < 4; ++w)                      // lanes col and col+32 of every wave
                    a += red[(size_t)(w * 64 + col) * 2 * NT + which * NT + j] +
                         red[(size_t)(w * 64 + col + 32) * 2 * NT + which * NT + j];
                atomicAdd(p.sums + (size_t)(blockIdx.x % REP) * 2 * p.cout_total + which * p.cout_total + p.ch_off + ch, a);
            }
        }
    }
}

// =================================================================================================
// lin_fwd, second generation (cin % 4 == 0, cin <= 128): no block barrier in the tile loop.
// 8 waves per CU (2 per SIMD); every wave owns 16-row strips end to end — fetch (registers, one
// strip ahead) -> transform -> its private LDS strip -> v_mfma_f32_16x16x4_f32 against the shared
// stationary W -> epilogue through the same LDS strip (full-row float4 stores) — so one wave's
// memory phases run under the other wave's MFMAs on the same SIMD.  The first generation
// (block-synchronous 128-row tiles) measured staging + epilogue + MFMA strictly serialised
// (tools/ablate_lin_fwd.py: 157 + 200 + 300 us on the 853632x128x128 layer).
// =================================================================================================
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
// streaming (non-temporal) 16-byte accesses for tensors that are touched once per launch
__device__ __forceinline__ float4 ld_stream(const float *ptr, bool nt) {
    if (nt) {
        const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt *>(ptr));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const float4 *>(ptr);
}
__device__ __forceinline__ void st_stream(float *ptr, const float4 &v, bool nt) {
    if (nt) {
        const f32x4_nt o = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(o, reinterpret_cast<f32x4_nt *>(ptr));
    } else {
        *reinterpret_cast<float4 *>(ptr) = v;
    }
}

constexpr int F2_THREADS = 512;
constexpr int F2_ROWS = 16;

// the tail of a forward launch: see LinFwdParams::fin_counter
// (`flag`: one word of the kernel's dynamic LDS, dead by now: a static __shared__ on top of the 160 KB carve-out fails to launch)
template <int NTHREADS>
__device__ __forceinline__ void finalize_by_last_block(const LinFwdParams &p, int tid, volatile int *flag) {
    // every thread's statistics atomics are acknowledged, then ONE lane releases, takes the ticket and (last block) acquires:
    // i2p_ticket_is_last (common.h).  (A release fence issued by all 512 threads cost ~1 ms per step over the ~40 layer launches.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) *flag = i2p_ticket_is_last(p.fin_counter, gridDim.x * gridDim.y) ? 1 : 0;
    __syncthreads();
    if (!*flag) return;
    const int c = p.cout_total, n2 = 2 * c;
    // the [REP][2c] replica sums: one round trip for the whole block (REP/ng independent loads per thread, combined through the
    // dead LDS) instead of 2*REP loads per channel thread in register-limited batches — this tail is serial time of every launch
    double *part = reinterpret_cast<double *>(const_cast<int *>(flag)) + 1;
    const bool spread = n2 <= NTHREADS;
    const int ng = spread ? NTHREADS / n2 : 1;
    if (spread) {
        const int idx = tid % n2, grp = tid / n2;
        double a = 0.0;
        if (grp < ng)
            for (int r = grp; r < REP; r += ng) a += __hip_atomic_load(p.sums + (size_t)r * n2 + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        part[tid] = a;
        __syncthreads();
    }
    for (int ch = tid; ch < c; ch += NTHREADS) {
        double s = 0.0, q = 0.0;
        if (spread) {
            for (int g2 = 0; g2 < ng; ++g2) { s += part[g2 * n2 + ch]; q += part[g2 * n2 + c + ch]; }
        } else {
            for (int r = 0; r < REP; ++r) {                 // agent-scope loads: the sums live in L2 (written by atomics only)
                s += __hip_atomic_load(p.sums + (size_t)r * 2 * c + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                q += __hip_atomic_load(p.sums + (size_t)r * 2 * c + c + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        const double m = s / (double)p.rows;
        double var = q / (double)p.rows - m * m;
        var = var < 0.0 ? 0.0 : var;
        const float invstd = rsqrtf((float)var + p.fin_eps);
        p.fin_coef[ch] = (float)m; p.fin_coef[c + ch] = invstd * p.fin_gamma[ch]; p.fin_coef[2 * c + ch] = p.fin_beta[ch];
        p.fin_mi[ch] = (float)m; p.fin_mi[c + ch] = invstd;
    }
    if (tid == 0) *p.fin_counter = 0u;
}



// branch-free (b,n,k) walk for a 16-row strip (M >= 16: at most one wrap inside a strip)
__device__ __forceinline__ void pair_row16(const PairTile &t, int d, int N, int M, int &bn, int &bk) {
    int k = t.k0 + d;
    const int wrap = k >= M ? 1 : 0;
    k -= wrap * M;
    const int n = t.n0 + wrap;
    const int b = t.b0 + (n >= N ? 1 : 0);
    bn = t.bn0 + wrap; bk = b * M + k;
}

// CH = float4 chunks per lane per strip = max(cin, cout)/16 rounded up to 2, 4 or 8: the narrow layers (<= 64
// channels) carry half the prefetch / output registers of the 128-wide ones and run 4 waves per SIMD instead of 2
// (they are HBM-bound: more loads in flight, not more math, is what they need).
template <int NT16, bool PAIR, bool DGRAD, int CH>
__global__ __launch_bounds__(F2_THREADS, CH <= 4 ? 4 : 2) void lin_fwd2_kernel(LinFwdParams p_in) {
    constexpr int F2_CH = CH;
    // Small layers (a few thousand rows) would leave most CUs idle and put a whole 16 x cout strip on one wave
    // (256 dependent-issue MFMAs = 3.4 us for 128 x 128): the launcher then splits the OUTPUT channels over
    // blockIdx.y — each slice is the same kernel on cout/nslice columns (its own weights rows, ch_off, statistics).
    LinFwdParams p = p_in;
    if (p.nslice > 1) {
        const int sl = blockIdx.y;
        p.ch_off += sl * p.cout;
        if (!(DGRAD && p.w_transposed)) p.w += (size_t)sl * p.cout * p.cin;
    }
    extern __shared__ float smem[];
    const int ldk = p.ldk;                                  // max(cin, cout_p) + 2 (8-byte aligned rows), or + 4 (16-byte, wide-K path)
    const bool wide_k = ((p.cin & 15) == 0) && ((ldk & 3) == 0) && !(p.ablate & 16);
    float *Ws = smem;                                       // [cout_p][ldk]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *As = smem + (size_t)p.cout_p * ldk + (size_t)wave * F2_ROWS * ldk;   // this wave's strip

    // (LDS tables of the dgrad instantiation, filled below)
    float *Gt = smem + ((size_t)p.cout_p + 8 * F2_ROWS) * ldk;
    float *Et = Gt + 6 * p.cin;

    // ---- per-lane chunk geometry, fixed for the whole kernel -----------------------------------------
    const int c4n = p.cin >> 2, nchunk = F2_ROWS * c4n;
    // (dgrad: cin/4 is a power of two dividing 64, so the geometry is two shifts of the lane id — keeping the three
    //  arrays in registers is what pushed the dgrad instantiation into spilling)
    int in_r_[DGRAD ? 1 : F2_CH], in_c4_[DGRAD ? 1 : F2_CH];
    bool in_ok_[DGRAD ? 1 : F2_CH];
    const int c4_shift = 31 - __clz(c4n);
    if (!DGRAD) {
#pragma unroll
        for (int u = 0; u < (DGRAD ? 1 : F2_CH); ++u) {
            const int i = lane + u * 64;
            in_ok_[u] = i < nchunk;
            const int ic = in_ok_[u] ? i : nchunk - 1;
            in_r_[u] = ic / c4n; in_c4_[u] = ic - in_r_[u] * c4n;
        }
    } else { in_r_[0] = 0; in_c4_[0] = lane & (c4n - 1); in_ok_[0] = true; }
    auto IR = [&](int u) -> int { if (DGRAD) { const int r = (lane + u * 64) >> c4_shift; return r < F2_ROWS ? r : F2_ROWS - 1; } return in_r_[u]; };
    auto IC = [&](int u) -> int { return DGRAD ? in_c4_[0] : in_c4_[u]; };
    auto IOK = [&](int u) -> bool { return DGRAD ? (lane + u * 64) < nchunk : in_ok_[u]; };
    const int o4n = p.cout >> 2;                            // 4, 8, 16 or 32 (launcher): divides 64
    const int o_shift = 31 - __clz(o4n);
    const int o_c4 = lane & (o4n - 1);                      // this lane's 4 output channels in the store phase
    constexpr int OCH = F2_CH;                              // <= 8 output chunks per lane (16 rows x 32 float4)

    // BN coefficients of the input chunks (loaded once; 3 float4 per chunk slot when they differ per slot)
    // input source of this lane (its channel set is the same in every chunk slot)
    const bool in_b = !DGRAD && p.xb && IC(0) * 4 >= p.split_c;
    const float *src_x = in_b ? p.xb : p.x;
    const int src_ld = (!DGRAD && p.xb) ? (in_b ? p.cin - p.split_c : p.split_c) : p.cin;
    const int src_c0 = in_b ? IC(0) * 4 - p.split_c : IC(0) * 4;
    const float *src_coef = in_b ? p.in_coef_b : p.in_coef;
    const float in_slope = in_b ? p.slope_b : p.slope_in;
    float4 cm = make_float4(0.f, 0.f, 0.f, 0.f), cs = make_float4(1.f, 1.f, 1.f, 1.f), cb = cm;
    if (src_coef) {                                         // (cin/4) divides 64: every slot of a lane has the same channels
        cm = *reinterpret_cast<const float4 *>(src_coef + src_c0);
        cs = *reinterpret_cast<const float4 *>(src_coef + src_ld + src_c0);
        cb = *reinterpret_cast<const float4 *>(src_coef + 2 * src_ld + src_c0);
    }

    // dgrad: BN-backward constants of this lane's input channels, and of its output channels for the store phase
    const bool g_act = DGRAD && p.g_coef && p.g_slope != 1.f;
    const float *g_tab = Gt + IC(0) * 4;                 // this lane's input channels (same in every chunk slot)
    // output destination of this lane's 4 channels (DGRAD two-destination mode splits the columns)
    const bool out_b = DGRAD && p.yb && (p.ch_off + o_c4 * 4) >= p.split_c;
    float *dst_y = out_b ? p.yb : p.y;
    const int dst_ld = (DGRAD && p.yb) ? (out_b ? p.cout_total - p.split_c : p.split_c) : p.y_ld;
    const int dst_c0 = (DGRAD && p.yb) ? (out_b ? p.ch_off + o_c4 * 4 - p.split_c : p.ch_off + o_c4 * 4) : p.ch_off + o_c4 * 4;
    const float *e_x = out_b ? p.exb : p.ex;
    const float *e_cf = out_b ? p.e_coef_b : p.e_coef;
    const float *e_m = out_b ? p.e_mi_b : p.e_mi;
    const float e_sl = out_b ? p.e_slope_b : p.e_slope;
    const float *e_ad = out_b ? p.e_add : nullptr;
    double *dst_sums = out_b ? p.sums_b : p.sums;
    const int sums_c = (DGRAD && p.yb) ? dst_ld : p.cout_total;
    const float *e_tab = Et + p.ch_off + o_c4 * 4;          // this lane's output channels
    (void)e_m;

    double ssum[4] = {0, 0, 0, 0}, ssq[4] = {0, 0, 0, 0};

    // x and y of a layer are touched once per launch: streaming accesses leave L2 to the tensors the neighbouring
    // kernels re-read (+0.65 % on the whole step, A/B on one box; I2P_LIN_ABLATE=32 turns the hint off)
    const bool nt_hint = (p.ablate & 32) == 0;
    const long long nstrips = (p.rows + F2_ROWS - 1) / F2_ROWS;
    const long long sstride = (long long)gridDim.x * 8;
    const long long last_row = p.rows - 1;
    long long strip = (long long)blockIdx.x * 8 + wave;

    // all loads of a strip are issued back to back, no control flow in between (clamped addresses)
    auto fetch = [&](long long st, float4 (&v)[F2_CH], float4 (&v2)[F2_CH]) {
        const long long row0 = st * F2_ROWS;
        PairTile pt; pt.bn0 = pt.k0 = pt.b0 = pt.n0 = 0;
        if (PAIR) pt = pair_tile(row0, p.pair_N, p.pair_M);
#pragma unroll
        for (int u = 0; u < F2_CH; ++u) {
            long long row = row0 + IR(u);
            int d = IR(u);
            if (row > last_row) { d = (int)(last_row - row0); row = last_row; }
            long long src = row;
            if (PAIR) { int bn, bk; pair_row16(pt, d, p.pair_N, p.pair_M, bn, bk); src = bk; }
            // (two-source mode requires one channel set per lane; plain mode takes the slot's own column)
            v[u] = ld_stream(src_x + (size_t)src * src_ld + ((!DGRAD && p.xb) ? src_c0 : IC(u) * 4), nt_hint && !PAIR);
            if (DGRAD && p.g_coef) v2[u] = *reinterpret_cast<const float4 *>(p.x2 + (size_t)src * p.cin + IC(u) * 4);
        }
    };

    // the first strip's loads (and the BN coefficients above) are in flight while the weights are staged: one memory round trip
    // of the prologue instead of two (the small layers' launches are chains of such round trips, ~2 us each)
    float4 pf[F2_CH], pf2[F2_CH];
#pragma unroll
    for (int u = 0; u < F2_CH; ++u) pf2[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (strip < nstrips) fetch(strip, pf, pf2);

    // weights -> LDS with 16-byte loads (cin, cout and ch_off are multiples of 4): 8 independent loads per thread for a
    // 128x128 layer instead of 32 dependent scalar iterations — this prologue is most of a launch on the small layers
    if (!p.w_vec) {                                         // weights not 16-byte aligned (caller's tensor view): scalar loads
        for (int i = tid; i < p.cout_p * p.cin; i += F2_THREADS) {
            const int co = i / p.cin, ci = i - co * p.cin;
            float wv = 0.f;
            if (co < p.cout) wv = (DGRAD && p.w_transposed) ? p.w[(size_t)ci * p.cout_total + p.ch_off + co] : p.w[(size_t)co * p.cin + ci];
            Ws[co * ldk + ci] = wv;
        }
    } else if (DGRAD && p.w_transposed) {                   // Ws[co][ci] = w[ci][ch_off + co]: contiguous along co
        const int o4 = p.cout_p >> 2;
        for (int i = tid; i < p.cin * o4; i += F2_THREADS) {
            const int ci = i / o4, c4 = i - ci * o4;
            float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c4 * 4 < p.cout) wv = *reinterpret_cast<const float4 *>(p.w + (size_t)ci * p.cout_total + p.ch_off + c4 * 4);
            Ws[(c4 * 4 + 0) * ldk + ci] = wv.x; Ws[(c4 * 4 + 1) * ldk + ci] = wv.y;
            Ws[(c4 * 4 + 2) * ldk + ci] = wv.z; Ws[(c4 * 4 + 3) * ldk + ci] = wv.w;
        }
    } else {
        const int c4n_w = p.cin >> 2;
        for (int i = tid; i < p.cout_p * c4n_w; i += F2_THREADS) {
            const int co = i / c4n_w, c4 = i - co * c4n_w;
            float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (co < p.cout) wv = *reinterpret_cast<const float4 *>(p.w + (size_t)co * p.cin + c4 * 4);
            float2 *dst = reinterpret_cast<float2 *>(Ws + co * ldk + c4 * 4);
            dst[0] = make_float2(wv.x, wv.y); dst[1] = make_float2(wv.z, wv.w);
        }
    }
    // dgrad: per-channel constants live in an LDS table behind the strips (30 float4 registers per lane otherwise:
    // the dgrad instantiation spilled).  Gt [6][cin] = BN-backward constants of the layer behind (identity if none),
    // Et [4][cout_total] = mean, scale, beta, invstd of the BN in front (per destination in two-destination mode).
    if (DGRAD) {
        // (formed here from the replica sums: a separate 1-block kernel per layer used to do this — 35 launches per step)
        // (the [REP][2cin] sums in one round trip: REP/ng independent loads per thread, combined through the strips' LDS, which is
        //  not staged yet — 2*REP loads per channel thread in batches of 8 were four dependent round trips of every dgrad launch)
        double *gpart = reinterpret_cast<double *>(smem + (size_t)p.cout_p * ldk);
        const int gn2 = 2 * p.cin;
        const bool gspread = p.g_coef && gn2 <= F2_THREADS && (size_t)F2_THREADS * 2 <= (size_t)8 * F2_ROWS * ldk;
        const int gng = gspread ? F2_THREADS / gn2 : 1;
        if (gspread) {
            const int idx = tid % gn2, grp = tid / gn2;
            double a = 0.0;
            if (grp < gng)
                for (int rp = grp; rp < REP; rp += gng) a += p.g_dsums[(size_t)rp * gn2 + idx];
            gpart[tid] = a;
            __syncthreads();
        }
        for (int ch = tid; ch < p.cin; ch += F2_THREADS) {
            float m1 = 0.f, m2 = 0.f, sc = 1.f, mu = 0.f, is = 1.f, be = 0.f;
            if (p.g_coef) {
                double sd = 0.0, sx = 0.0;
                if (gspread) {
                    for (int g2 = 0; g2 < gng; ++g2) { sd += gpart[g2 * gn2 + ch]; sx += gpart[g2 * gn2 + p.cin + ch]; }
                } else {
#pragma unroll 8
                    for (int rp = 0; rp < REP; ++rp) { sd += p.g_dsums[(size_t)rp * 2 * p.cin + ch]; sx += p.g_dsums[(size_t)rp * 2 * p.cin + p.cin + ch]; }
                }
                m1 = (float)(sd / (double)p.g_rows); m2 = (float)(sx / (double)p.g_rows);
                sc = p.g_oc[p.cin + ch]; mu = p.g_omi[ch]; is = p.g_omi[p.cin + ch]; be = p.g_oc[2 * p.cin + ch];
            }
            Gt[ch] = m1; Gt[p.cin + ch] = m2; Gt[2 * p.cin + ch] = sc; Gt[3 * p.cin + ch] = mu; Gt[4 * p.cin + ch] = is; Gt[5 * p.cin + ch] = be;
        }
        for (int i = tid; i < 4 * p.cout_total; i += F2_THREADS) {
            const int row = i / p.cout_total, ch = i - row * p.cout_total;
            const bool b = p.yb && ch >= p.split_c;
            const float *cf = b ? p.e_coef_b : p.e_coef, *mi = b ? p.e_mi_b : p.e_mi;
            const int ld = p.yb ? (b ? p.cout_total - p.split_c : p.split_c) : p.cout_total, c = b ? ch - p.split_c : ch;
            float v = (row == 1 || row == 3) ? 1.f : 0.f;
            if (cf) v = row < 3 ? cf[row * ld + c] : mi[ld + c];
            Et[i] = v;
        }
    }
    __syncthreads();                                        // the only block barrier

    // Order inside an iteration (vmcnt retires in order and counts stores too):
    //   MFMA(s) -> outputs of s to registers -> commit(s+1) [waits the loads issued one MFMA phase ago]
    //   -> stores(s) -> fetch(s+2).   The stores of a strip are thus always OLDER than the loads the next
    //   commit waits for, and both had a whole MFMA phase to land; no wait ever exposes store latency.
    auto commit = [&](long long st, const float4 (&v)[F2_CH], const float4 (&v2)[F2_CH]) {
        const long long row0 = st * F2_ROWS;
        PairTile pt; pt.bn0 = pt.k0 = pt.b0 = pt.n0 = 0;
        if (PAIR) pt = pair_tile(row0, p.pair_N, p.pair_M);
#pragma unroll
        for (int u = 0; u < F2_CH; ++u) {
            float4 t = v[u];
            if (PAIR) {
                int d = IR(u);
                if (row0 + d > last_row) d = (int)(last_row - row0);
                int bn, bk; pair_row16(pt, d, p.pair_N, p.pair_M, bn, bk);
                const float4 f = *reinterpret_cast<const float4 *>(p.pair_f + (size_t)bn * p.cin + IC(u) * 4);
                t.x *= f.x; t.y *= f.y; t.z *= f.z; t.w *= f.w;
            }
            if (DGRAD && p.g_coef) {                        // BN backward of the layer behind, formed on load
                const float4 g_m1 = *reinterpret_cast<const float4 *>(g_tab), g_m2 = *reinterpret_cast<const float4 *>(g_tab + p.cin);
                const float4 g_sc = *reinterpret_cast<const float4 *>(g_tab + 2 * p.cin), g_mu = *reinterpret_cast<const float4 *>(g_tab + 3 * p.cin);
                const float4 g_is = *reinterpret_cast<const float4 *>(g_tab + 4 * p.cin), g_be = *reinterpret_cast<const float4 *>(g_tab + 5 * p.cin);
                const float4 yv = v2[u];
                if (g_act) {                                // dL/da -> dL/dz of the layer behind
                    t.x = (yv.x - g_mu.x) * g_sc.x + g_be.x > 0.f ? t.x : t.x * p.g_slope;
                    t.y = (yv.y - g_mu.y) * g_sc.y + g_be.y > 0.f ? t.y : t.y * p.g_slope;
                    t.z = (yv.z - g_mu.z) * g_sc.z + g_be.z > 0.f ? t.z : t.z * p.g_slope;
                    t.w = (yv.w - g_mu.w) * g_sc.w + g_be.w > 0.f ? t.w : t.w * p.g_slope;
                }
                t.x = g_sc.x * (t.x - g_m1.x - ((yv.x - g_mu.x) * g_is.x) * g_m2.x);
                t.y = g_sc.y * (t.y - g_m1.y - ((yv.y - g_mu.y) * g_is.y) * g_m2.y);
                t.z = g_sc.z * (t.z - g_m1.z - ((yv.z - g_mu.z) * g_is.z) * g_m2.z);
                t.w = g_sc.w * (t.w - g_m1.w - ((yv.w - g_mu.w) * g_is.w) * g_m2.w);
            }
            if (!DGRAD && src_coef) {                       // launcher guarantees one channel set per lane
                const float4 m = cm, sc = cs, bb = cb;
                t.x = act_apply((t.x - m.x) * sc.x + bb.x, in_slope);
                t.y = act_apply((t.y - m.y) * sc.y + bb.y, in_slope);
                t.z = act_apply((t.z - m.z) * sc.z + bb.z, in_slope);
                t.w = act_apply((t.w - m.w) * sc.w + bb.w, in_slope);
            }
            if (IOK(u)) {
                float2 *dst = reinterpret_cast<float2 *>(As + IR(u) * ldk + IC(u) * 4);
                dst[0] = make_float2(t.x, t.y); dst[1] = make_float2(t.z, t.w);
            }
        }
    };

    if (strip < nstrips) {
        commit(strip, pf, pf2);
        if (strip + sstride < nstrips) fetch(strip + sstride, pf, pf2);
    }
    for (; strip < nstrips; strip += sstride) {
        const long long row0 = strip * F2_ROWS;
        PairTile pt; pt.bn0 = pt.k0 = pt.b0 = pt.n0 = 0;
        if (PAIR) pt = pair_tile(row0, p.pair_N, p.pair_M);

        // ---- 16 x (16*NT16) on the matrix cores ---------------------------------------------------------
        f32x4 acc[NT16];
#pragma unroll
        for (int j = 0; j < NT16; ++j) { acc[j][0] = 0.f; acc[j][1] = 0.f; acc[j][2] = 0.f; acc[j][3] = 0.f; }
        const float *arow = As + (lane & 15) * ldk + (lane >> 4);
        const float *brow = Ws + (lane & 15) * ldk + (lane >> 4);
        // Ping-pong fragment registers, two k-steps per iteration: the operands of a step were requested a
        // full step (8 MFMAs = 256 cycles) earlier, so no s_waitcnt ever sits between two MFMAs and no
        // register copies are needed (a single-buffered version measured 72 % of the MFMA issue rate).
        const int K4 = (p.ablate & 1) ? 0 : (p.cin >> 2);
        if (wide_k) {
            // K permuted across the four k-slots of the fragment layout: slot q = lane>>4 owns the CONTIGUOUS range
            // k in [q*L, (q+1)*L), L = cin/4 (A and B use the same permutation, so the product is unchanged).  One
            // ds_read_b128 then feeds four k-steps: 9 wide LDS reads per 32 MFMAs instead of 36 narrow ones with
            // per-read address arithmetic, and the MFMAs issue back to back.  Rows are 16-byte aligned and
            // ldk = 4 (mod 32), so the eight lanes of a 128-byte LDS pass hit 32 distinct banks.
            const int L = K4;
            const float *aw = As + (lane & 15) * ldk + (lane >> 4) * L;
            const float *bw = Ws + (lane & 15) * ldk + (lane >> 4) * L;
            // B fragments single-buffered in two halves: while the MFMAs of one half of the column tiles run, the
            // other half's registers are refilled for the next 4 k-steps (an accumulator is revisited after
            // NT16/2 >= 4 MFMAs = 128 cycles > the 40-cycle dependent latency); A is double-buffered (4 VGPRs each).
            constexpr int H = NT16 >= 2 ? NT16 / 2 : 1;
            f32x4 av = *reinterpret_cast<const f32x4 *>(aw), an = av;
            f32x4 bv[NT16];
#pragma unroll
            for (int j = 0; j < NT16; ++j) bv[j] = *reinterpret_cast<const f32x4 *>(bw + (size_t)j * 16 * ldk);
            for (int s4 = 0; s4 < L; s4 += 4) {
                const int sn = s4 + 4 < L ? s4 + 4 : s4;          // next chunk (the last one re-reads itself: harmless)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < H; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[j][e], acc[j], 0, 0, 0);
                an = *reinterpret_cast<const f32x4 *>(aw + sn);
#pragma unroll
                for (int j = 0; j < H; ++j) bv[j] = *reinterpret_cast<const f32x4 *>(bw + (size_t)j * 16 * ldk + sn);
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * H, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, H + 1, 0);
                if (NT16 >= 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int j = H; j < NT16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[j][e], acc[j], 0, 0, 0);
#pragma unroll
                    for (int j = H; j < NT16; ++j) bv[j] = *reinterpret_cast<const f32x4 *>(bw + (size_t)j * 16 * ldk + sn);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * (NT16 - H), 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, NT16 - H, 0);
                }
                av = an;
            }
        } else {
        float a0 = arow[0], b0[NT16], a1 = 0.f, b1[NT16];
#pragma unroll
        for (int j = 0; j < NT16; ++j) { b0[j] = brow[(size_t)j * 16 * ldk]; b1[j] = 0.f; }
        int st4 = 0;
        for (; st4 + 1 < K4; st4 += 2) {
            const int k1 = (st4 + 1) * 4;
            a1 = arow[k1];
#pragma unroll
            for (int j = 0; j < NT16; ++j) b1[j] = brow[(size_t)j * 16 * ldk + k1];
#pragma unroll
            for (int j = 0; j < NT16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0[j], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, NT16 + 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NT16, 0);
            const int k2 = (st4 + 2 < K4 ? st4 + 2 : st4 + 1) * 4;
            a0 = arow[k2];
#pragma unroll
            for (int j = 0; j < NT16; ++j) b0[j] = brow[(size_t)j * 16 * ldk + k2];
#pragma unroll
            for (int j = 0; j < NT16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1[j], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, NT16 + 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NT16, 0);
        }
        if (st4 < K4) {
#pragma unroll
            for (int j = 0; j < NT16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0[j], acc[j], 0, 0, 0);
        }

        }

        // ---- epilogue 1: fragments -> this wave's strip (C/D of 16x16: col = lane&15, row = (lane>>4)*4+e) ----
#pragma unroll
        for (int j = 0; j < NT16; ++j) {
            const int ch = j * 16 + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) As[((lane >> 4) * 4 + e) * ldk + ch] = acc[j][e];
        }
        // ---- epilogue 2: row-major float4 chunks to registers: bias (pair mode), statistics -----------------
        float4 ov[OCH];
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < OCH; ++u) {
            const int r = (lane + u * 64) >> o_shift;       // chunk (r, o_c4)
            ov[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < F2_ROWS) {
                const float2 lo = *reinterpret_cast<const float2 *>(As + r * ldk + o_c4 * 4);
                const float2 hi = *reinterpret_cast<const float2 *>(As + r * ldk + o_c4 * 4 + 2);
                float4 v = make_float4(lo.x, lo.y, hi.x, hi.y);
                if (PAIR) {
                    int d = r;
                    if (row0 + d > last_row) d = (int)(last_row - row0);
                    int bn, bk; pair_row16(pt, d, p.pair_N, p.pair_M, bn, bk);
                    const float4 a = *reinterpret_cast<const float4 *>(p.bias_n + (size_t)bn * p.cout_total + p.ch_off + o_c4 * 4);
                    const float4 c = *reinterpret_cast<const float4 *>(p.bias_k + (size_t)bk * p.cout_total + p.ch_off + o_c4 * 4);
                    v.x = (v.x + a.x) + c.x; v.y = (v.y + a.y) + c.y; v.z = (v.z + a.z) + c.z; v.w = (v.w + a.w) + c.w;
                }
                if (DGRAD) {
                    if (e_ad) {
                        long long er = row0 + r; if (er > last_row) er = last_row;
                        const float4 ad = *reinterpret_cast<const float4 *>(e_ad + (size_t)er * dst_ld + dst_c0);
                        v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
                    }
                    if (e_cf) {
                        const float4 e_mu = *reinterpret_cast<const float4 *>(e_tab), e_sc = *reinterpret_cast<const float4 *>(e_tab + p.cout_total);
                        const float4 e_be = *reinterpret_cast<const float4 *>(e_tab + 2 * p.cout_total), e_is = *reinterpret_cast<const float4 *>(e_tab + 3 * p.cout_total);
                        long long er = row0 + r; if (er > last_row) er = last_row;
                        const float4 xr = *reinterpret_cast<const float4 *>(e_x + (size_t)er * dst_ld + dst_c0);
                        const float zx = (xr.x - e_mu.x) * e_sc.x + e_be.x, zy = (xr.y - e_mu.y) * e_sc.y + e_be.y;
                        const float zz = (xr.z - e_mu.z) * e_sc.z + e_be.z, zw = (xr.w - e_mu.w) * e_sc.w + e_be.w;
                        v.x = zx > 0.f ? v.x : v.x * e_sl; v.y = zy > 0.f ? v.y : v.y * e_sl;
                        v.z = zz > 0.f ? v.z : v.z * e_sl; v.w = zw > 0.f ? v.w : v.w * e_sl;
                        if (row0 + r <= last_row) {
                            s1[0] += v.x; s1[1] += v.y; s1[2] += v.z; s1[3] += v.w;
                            s2[0] = fmaf(v.x, (xr.x - e_mu.x) * e_is.x, s2[0]); s2[1] = fmaf(v.y, (xr.y - e_mu.y) * e_is.y, s2[1]);
                            s2[2] = fmaf(v.z, (xr.z - e_mu.z) * e_is.z, s2[2]); s2[3] = fmaf(v.w, (xr.w - e_mu.w) * e_is.w, s2[3]);
                        }
                    }
                } else if (row0 + r <= last_row && !(p.ablate & 4)) {
                    s1[0] += v.x; s1[1] += v.y; s1[2] += v.z; s1[3] += v.w;
                    s2[0] = fmaf(v.x, v.x, s2[0]); s2[1] = fmaf(v.y, v.y, s2[1]);
                    s2[2] = fmaf(v.z, v.z, s2[2]); s2[3] = fmaf(v.w, v.w, s2[3]);
                }
                ov[u] = v;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { ssum[q] += (double)s1[q]; ssq[q] += (double)s2[q]; }
        // ---- the strip is free again: stage the next one, then store this one's outputs ---------------------
        if (strip + sstride < nstrips && !(p.ablate & 8)) commit(strip + sstride, pf, pf2);
#pragma unroll
        for (int u = 0; u < OCH; ++u) {
            const int r = (lane + u * 64) >> o_shift;
            if (r < F2_ROWS && row0 + r <= last_row && !(p.ablate & 2))
                st_stream(dst_y + (size_t)(row0 + r) * dst_ld + dst_c0, ov[u], nt_hint);
        }
        if (strip + 2 * sstride < nstrips && !(p.ablate & 8)) fetch(strip + 2 * sstride, pf, pf2);
    }

    if (p.sums || (DGRAD && p.sums_b)) {
        // lanes with equal (lane & (o4n-1)) own the same 4 channels.  The 8 waves of a block combine through their (now dead) strips
        // and wave 0 issues the block's atomics: 8x fewer same-address fp64 atomics at L2 (a mid-size layer's 1800 waves put ~57
        // serialized atomics on each of the 32 x 2c replica words — microseconds on a 20 us launch); I2P_LIN_ABLATE bit 64 = per wave.
        const bool per_wave = (p.ablate & 64) != 0;
        double *wred = reinterpret_cast<double *>(smem + (size_t)p.cout_p * ldk);       // [8 waves][o4n lanes][8]
        const int wstride = F2_ROWS * ldk / 2;                                            // doubles per wave strip (ldk is even)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double a = ssum[q], b = ssq[q];
            for (int off = 32; off >= o4n; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
            if (per_wave) {
                if (lane < o4n && dst_sums) {
                    double *rep = dst_sums + (size_t)((blockIdx.x * 8 + wave) % REP) * 2 * sums_c;
                    atomicAdd(rep + dst_c0 + q, a); atomicAdd(rep + sums_c + dst_c0 + q, b);
                }
            } else if (lane < o4n) {
                wred[(size_t)wave * wstride + lane * 8 + q] = a; wred[(size_t)wave * wstride + lane * 8 + 4 + q] = b;
            }
        }
        if (!per_wave) {
            __syncthreads();
            if (wave == 0 && lane < o4n && dst_sums) {
                double *rep = dst_sums + (size_t)(blockIdx.x % REP) * 2 * sums_c;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    double a = 0.0, b = 0.0;
#pragma unroll
                    for (int w8 = 0; w8 < 8; ++w8) { a += wred[(size_t)w8 * wstride + lane * 8 + q]; b += wred[(size_t)w8 * wstride + lane * 8 + 4 + q]; }
                    atomicAdd(rep + dst_c0 + q, a); atomicAdd(rep + sums_c + dst_c0 + q, b);
                }
            }
        }
    }
    if (!DGRAD && p.fin_counter) finalize_by_last_block<F2_THREADS>(p, tid, reinterpret_cast<volatile int *>(smem));
}

template <int NT16, bool PAIR, bool DGRAD, int CH>
int launch_fwd2(const LinFwdParams &p, hipStream_t st) {
    const size_t bytes = (((size_t)p.cout_p + 8 * F2_ROWS) * p.ldk + (DGRAD ? 6 * (size_t)p.cin + 4 * (size_t)p.cout_total : 0)) * sizeof(float);
    if (bytes > 160 * 1024) return I2P_ERR_BAD_ARG;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lin_fwd2_kernel<NT16, PAIR, DGRAD, CH>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const long long nstrips = (p.rows + F2_ROWS - 1) / F2_ROWS;
    long long g = (nstrips + 7) / 8;
    // persistent grid: as many blocks as the occupancy of this instantiation keeps resident per CU
    const int per_cu = CH <= 4 ? (bytes <= 52 * 1024 ? 3 : (bytes <= 80 * 1024 ? 2 : 1)) : 1;
    const long long cap = 256LL * (per_cu > 2 ? 2 : per_cu);
    const unsigned grid = (unsigned)(g < cap ? (g < 1 ? 1 : g) : cap);
    LinFwdParams q = p;
    q.w_vec = ((reinterpret_cast<uintptr_t>(p.w) & 15) == 0 && (p.cin & 3) == 0 && (p.cout_total & 3) == 0 && (p.ch_off & 3) == 0 &&
               (p.cout & 3) == 0) ? 1 : 0;
    hipLaunchKernelGGL((lin_fwd2_kernel<NT16, PAIR, DGRAD, CH>), dim3(grid, p.nslice > 1 ? p.nslice : 1), dim3(F2_THREADS), bytes, st, q);
    I2P_RETURN_LAUNCH_STATUS();
}

template <bool PAIR, bool DGRAD>
int dispatch_fwd2(const LinFwdParams &p_in, hipStream_t st) {
    LinFwdParams p = p_in;
    p.nslice = 1;
    {   // small layers: slice the output channels over blockIdx.y until ~256 blocks exist (slices stay >= 16 wide)
        const long long g = ((p.rows + F2_ROWS - 1) / F2_ROWS + 7) / 8;
        int S = 1;
        while (g * S < 192 && p.cout / (2 * S) >= 16 && ((p.cout / (2 * S)) & 15) == 0) S *= 2;
        if (S > 1) { p.nslice = S; p.cout /= S; p.cout_p = p.cout; }
    }
    const int wide = p.cin > p.cout ? p.cin : p.cout;      // channels a strip row holds (input and output phases)
    const bool narrow_ok = !PAIR;
    const bool ch2 = narrow_ok && wide <= 32;               // (a 4-chunk instantiation spilled at its 128 VGPRs and measured no faster: removed)
    switch (p.cout_p / 16) {
        case 1: return ch2 ? launch_fwd2<1, PAIR, DGRAD, 2>(p, st) : launch_fwd2<1, PAIR, DGRAD, 8>(p, st);
        case 2: return ch2 ? launch_fwd2<2, PAIR, DGRAD, 2>(p, st) : launch_fwd2<2, PAIR, DGRAD, 8>(p, st);
        case 4: return launch_fwd2<4, PAIR, DGRAD, 8>(p, st);
        case 8: return launch_fwd2<8, PAIR, DGRAD, 8>(p, st);
        default: return I2P_ERR_BAD_ARG;
    }
}

// mean/scale/beta of a BN from its replica sums:  coef [3][c]
__global__ void bn_finalize_kernel(long long rows, int c, const double *__restrict__ sums,
                                   const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                   float *__restrict__ coef, float *__restrict__ mean_invstd) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    double s = 0.0, q = 0.0;
    for (int r = 0; r < REP; ++r) { s += sums[(size_t)r * 2 * c + ch]; q += sums[(size_t)r * 2 * c + c + ch]; }
    const double m = s / (double)rows;
    double var = q / (double)rows - m * m;
    var = var < 0.0 ? 0.0 : var;
    const float invstd = rsqrtf((float)var + eps);
    coef[ch] = (float)m; coef[c + ch] = invstd * gamma[ch]; coef[2 * c + ch] = beta[ch];
    if (mean_invstd) { mean_invstd[ch] = (float)m; mean_invstd[c + ch] = invstd; }
}

template <int NT>
int launch_fwd(const LinFwdParams &p, hipStream_t st) {
    constexpr int TILE_R = 128;
    const size_t lds = ((size_t)p.cout_p + TILE_R) * p.ldk * sizeof(float);
    const size_t red = (size_t)THREADS * 2 * NT * sizeof(double);
    const size_t bytes = lds > red ? lds : red;
    if (bytes > 160 * 1024) return I2P_ERR_BAD_ARG;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lin_fwd_kernel<TILE_R, NT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const long long ntiles = (p.rows + TILE_R - 1) / TILE_R;
    const unsigned grid = (unsigned)(ntiles < 256 ? ntiles : 256);      // persistent: one block per CU
    hipLaunchKernelGGL((lin_fwd_kernel<TILE_R, NT>), dim3(grid), dim3(THREADS), bytes, st, p);
    I2P_RETURN_LAUNCH_STATUS();
}


// =================================================================================================
// Backward of one fused layer.
//   g^y  = scale_out * (gz - m1 - xhat_out * m2)         BN backward of THIS layer, formed on load
//          (or gz itself when out_coef == nullptr: the caller already holds dL/dy)
//   x'   = act_in(bn_in(x))                               recomputed on load (never materialised)
//   dW  += g^y^T . x'                                     wgrad  (per-block partial, reduced after)
//   gz_in = (g^y . W) * act_in'(z_in)                     dgrad, continues into the previous BN
//   in_dsums += { sum gz_in, sum gz_in * xhat_in }
// Block-synchronous 64-row tiles: LDS holds W [cout_p][cin_p+1], G [64][cout_p+1], X' [64][cin_p+1].
// =================================================================================================
struct LinBwdParams {
    long long rows;
    int cin, cout, cin_p, cout_p;      // *_p rounded up to 32
    int ldw, ldg, ldx;                 // LDS strides (cin_p+1, cout_p+1, cin_p+1)
    const float *gz, *y, *out_coef, *out_mi;
    float slope_out;            // gz is dL/da of this layer's activation (slope_out): act'(z) applied on load; 1 = gz is dL/dz
    const double *out_dsums;
    float *bn_out;              // [8][cout] or nullptr: rows 6, 7 receive dbeta = sum gz, dgamma = sum gz*xhat of the BN behind (block 0)
    const float *x, *in_coef, *in_mi;
    float slope_in;
    const float *w;
    float *gz_in;
    double *in_dsums;
    float *dw_partial;
    // pair mode (first cost-volume layer): x' = pair_f[b,n,:] * pair_g[b,k,:]; instead of gz_in the
    // dgrad result T = g^y . W is reduced on the fly into
    //   d_f[b,n,:] += sum_k T * pair_g ,  d_g[b,k,:] += sum_n T * pair_f      (fp32 atomics, zeroed by caller)
    // and the bias gradients  d_bn[b,n,:] += sum_k g^y ,  d_bk[b,k,:] += sum_n g^y.
    const float *pair_f, *pair_g;
    float *d_f, *d_g, *d_bn, *d_bk;
    int pair_N, pair_M;
};

constexpr int BWD_R = 64;

template <int NTI, int NTO>
__global__ __launch_bounds__(THREADS) void lin_bwd_kernel(LinBwdParams p) {
    extern __shared__ float smem[];
    float *Ws = smem;                                      // [cout_p][ldw]
    float *Gs = Ws + (size_t)p.cout_p * p.ldw;             // [BWD_R][ldg]
    float *Xs = Gs + (size_t)BWD_R * p.ldg;                // [BWD_R][ldx]
    float *Co = Xs + (size_t)BWD_R * p.ldx;                // [6][cout_p]: m1, m2, scale, mean, invstd, beta of the BN behind
    float *Ci = Co + 6 * p.cout_p;                         // [4][cin_p]: mean, scale, beta, invstd of the BN in front
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int DT = 2 * NTI;                            // dgrad tiles per row tile pair
    constexpr int DPW = (DT + 3) / 4;
    constexpr int WT = NTO * NTI;
    constexpr int WPW = (WT + 3) / 4;

    for (int i = tid; i < p.cout_p * p.cin_p; i += THREADS) {
        const int co = i / p.cin_p, ci = i - co * p.cin_p;
        Ws[co * p.ldw + ci] = (co < p.cout && ci < p.cin) ? p.w[(size_t)co * p.cin + ci] : 0.f;
    }
    // per-channel constants once per block (the replica sums are 64 fp64 loads per channel)
    for (int ch = tid; ch < p.cout; ch += THREADS) {
        float m1 = 0.f, m2 = 0.f, sc = 1.f, mu = 0.f, is = 1.f, be = 0.f;
        if (p.out_coef) {
            double sd = 0.0, sx = 0.0;
            for (int rp = 0; rp < REP; ++rp) { sd += p.out_dsums[(size_t)rp * 2 * p.cout + ch]; sx += p.out_dsums[(size_t)rp * 2 * p.cout + p.cout + ch]; }
            m1 = (float)(sd / (double)p.rows); m2 = (float)(sx / (double)p.rows);
            sc = p.out_coef[p.cout + ch]; mu = p.out_mi[ch]; is = p.out_mi[p.cout + ch]; be = p.out_coef[2 * p.cout + ch];
            if (p.bn_out && blockIdx.x == 0) { p.bn_out[6 * p.cout + ch] = (float)sd; p.bn_out[7 * p.cout + ch] = (float)sx; }   // (what bnbwd_coef_kernel returned)
        }
        Co[ch] = m1; Co[p.cout_p + ch] = m2; Co[2 * p.cout_p + ch] = sc; Co[3 * p.cout_p + ch] = mu; Co[4 * p.cout_p + ch] = is;
        Co[5 * p.cout_p + ch] = be;
    }
    for (int ch = tid; ch < p.cin; ch += THREADS) {
        float mu = 0.f, sc = 1.f, be = 0.f, is = 1.f;
        if (p.in_coef) { mu = p.in_coef[ch]; sc = p.in_coef[p.cin + ch]; be = p.in_coef[2 * p.cin + ch]; is = p.in_mi[p.cin + ch]; }
        Ci[ch] = mu; Ci[p.cin_p + ch] = sc; Ci[2 * p.cin_p + ch] = be; Ci[3 * p.cin_p + ch] = is;
    }
    // zero the K-padding columns once (staging only writes logical columns)
    for (int i = tid; i < BWD_R * (p.cout_p - p.cout); i += THREADS) {
        const int r = i / (p.cout_p - p.cout), c = p.cout + i % (p.cout_p - p.cout);
        Gs[r * p.ldg + c] = 0.f;
    }
    for (int i = tid; i < BWD_R * (p.cin_p - p.cin); i += THREADS) {
        const int r = i / (p.cin_p - p.cin), c = p.cin + i % (p.cin_p - p.cin);
        Xs[r * p.ldx + c] = 0.f;
    }

    f32x16 accw[WPW];
#pragma unroll
    for (int t = 0; t < WPW; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) accw[t][e] = 0.f;
    double dsum[DPW], dsxh[DPW];
#pragma unroll
    for (int t = 0; t < DPW; ++t) { dsum[t] = 0.0; dsxh[t] = 0.0; }

    const long long ntiles = (p.rows + BWD_R - 1) / BWD_R;
    const int co4 = p.cout >> 2, ci4 = p.cin >> 2;         // float4 chunks (channel counts are multiples of 4)
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long row0 = tile * BWD_R;
        PairTile bpt; bpt.bn0 = bpt.k0 = bpt.b0 = bpt.n0 = 0;
        if (p.pair_f) bpt = pair_tile(row0, p.pair_N, p.pair_M);
        __syncthreads();
        // ---- stage G = BN-backward(gz, y) --------------------------------------------------------
        for (int i0 = tid; i0 < BWD_R * co4; i0 += THREADS * 4) {
            float4 g[4], yv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * THREADS, r = i / co4, c4 = i - r * co4;
                g[u] = make_float4(0.f, 0.f, 0.f, 0.f); yv[u] = g[u];
                if (i < BWD_R * co4 && row0 + r < p.rows) {
                    g[u] = *reinterpret_cast<const float4 *>(p.gz + (size_t)(row0 + r) * p.cout + c4 * 4);
                    if (p.out_coef) yv[u] = *reinterpret_cast<const float4 *>(p.y + (size_t)(row0 + r) * p.cout + c4 * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * THREADS, r = i / co4, c4 = i - r * co4;
                if (i < BWD_R * co4) {
                    float gv[4] = {g[u].x, g[u].y, g[u].z, g[u].w};
                    if (p.out_coef && row0 + r < p.rows) {
                        const float yy[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int ch = c4 * 4 + q;
                            const float xh = (yy[q] - Co[3 * p.cout_p + ch]) * Co[4 * p.cout_p + ch];
                            if (p.slope_out != 1.f && !((yy[q] - Co[3 * p.cout_p + ch]) * Co[2 * p.cout_p + ch] + Co[5 * p.cout_p + ch] > 0.f))
                                gv[q] *= p.slope_out;                           // dL/da -> dL/dz
                            gv[q] = Co[2 * p.cout_p + ch] * (gv[q] - Co[ch] - xh * Co[p.cout_p + ch]);
                        }
                    }
                    if (p.d_bk && row0 + r < p.rows) {
                        int bn, bk; pair_row(bpt, r, p.pair_N, p.pair_M, bn, bk);
                        float *dk = p.d_bk + (size_t)bk * p.cout + c4 * 4;
                        atomicAdd(dk + 0, gv[0]); atomicAdd(dk + 1, gv[1]); atomicAdd(dk + 2, gv[2]); atomicAdd(dk + 3, gv[3]);
                    }
                    float *dst = Gs + r * p.ldg + c4 * 4;
                    dst[0] = gv[0]; dst[1] = gv[1]; dst[2] = gv[2]; dst[3] = gv[3];
                }
            }
        }
        // ---- stage X' = act(bn(x)) ----------------------------------------------------------------
        for (int i0 = tid; i0 < BWD_R * ci4; i0 += THREADS * 4) {
            float4 xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * THREADS, r = i / ci4, c4 = i - r * ci4;
                xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < BWD_R * ci4 && row0 + r < p.rows) {
                    if (p.pair_f) {
                        int bn, bk; pair_row(bpt, r, p.pair_N, p.pair_M, bn, bk);
                        const float4 f = *reinterpret_cast<const float4 *>(p.pair_f + (size_t)bn * p.cin + c4 * 4);
                        const float4 gg = *reinterpret_cast<const float4 *>(p.pair_g + (size_t)bk * p.cin + c4 * 4);
                        xv[u] = make_float4(f.x * gg.x, f.y * gg.y, f.z * gg.z, f.w * gg.w);
                    } else {
                        xv[u] = *reinterpret_cast<const float4 *>(p.x + (size_t)(row0 + r) * p.cin + c4 * 4);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * THREADS, r = i / ci4, c4 = i - r * ci4;
                if (i < BWD_R * ci4) {
                    float v[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
                    if (p.in_coef && row0 + r < p.rows) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int ch = c4 * 4 + q;
                            v[q] = act_apply((v[q] - Ci[ch]) * Ci[p.cin_p + ch] + Ci[2 * p.cin_p + ch], p.slope_in);
                        }
                    }
                    float *dst = Xs + r * p.ldx + c4 * 4;
                    dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
                }
            }
        }
        __syncthreads();
        if (p.d_bn) {       // per-point bias gradient: column sums of G over the (at most two) points of this tile
            for (int ch = tid; ch < p.cout; ch += THREADS) {
                const long long bn0 = bpt.bn0;
                const long long split = p.pair_M - bpt.k0;                      // rows [0,split) belong to bn0
                float s0 = 0.f, s1 = 0.f;
                for (int r = 0; r < BWD_R; ++r) {
                    const float gvv = Gs[r * p.ldg + ch];
                    if (r < split) s0 += gvv; else s1 += gvv;
                }
                atomicAdd(p.d_bn + (size_t)bn0 * p.cout + ch, s0);
                if (split < BWD_R && row0 + split < p.rows) atomicAdd(p.d_bn + (size_t)(bn0 + 1) * p.cout + ch, s1);
            }
        }

        // ---- wgrad: accw[co tile][ci tile] += G^T . X'   (K = 64 rows) -----------------------------
#pragma unroll
        for (int t = 0; t < WPW; ++t) {
            const int tw = wave + 4 * t;
            if (tw < WT) {
                const int to = tw / NTI, ti = tw - to * NTI;
                const float *ap = Gs + (lane >> 5) * p.ldg + to * 32 + (lane & 31);
                const float *bp = Xs + (lane >> 5) * p.ldx + ti * 32 + (lane & 31);
                // ping-pong fragments, two k-steps per iteration: operands are requested a full MFMA (64 cycles) before use
                float a0 = ap[0], b0 = bp[0], a1, b1;
#pragma unroll 4
                for (int kk = 0; kk < BWD_R; kk += 4) {
                    a1 = ap[(size_t)(kk + 2) * p.ldg]; b1 = bp[(size_t)(kk + 2) * p.ldx];
                    accw[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, accw[t], 0, 0, 0);
                    const int kn = kk + 4 < BWD_R ? kk + 4 : kk + 2;
                    a0 = ap[(size_t)kn * p.ldg]; b0 = bp[(size_t)kn * p.ldx];
                    accw[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, accw[t], 0, 0, 0);
                }
            }
        }
        // ---- dgrad: (row tile, ci tile) = G . W   (K = cout) ---------------------------------------
        if (p.gz_in || p.d_f) {
#pragma unroll
            for (int t = 0; t < DPW; ++t) {
                const int td = wave + 4 * t;
                if (td < DT) {
                    const int rt = td / NTI, ti = td - rt * NTI;
                    f32x16 acc;
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
                    const float *ap = Gs + (rt * 32 + (lane & 31)) * p.ldg + (lane >> 5);
                    const float *bp = Ws + (lane >> 5) * p.ldw + ti * 32 + (lane & 31);
                    float a0 = ap[0], b0 = bp[0], a1, b1;            // ping-pong fragments (see pair_bwd_kernel)
#pragma unroll 4
                    for (int kk = 0; kk < p.cout_p; kk += 4) {
                        a1 = ap[kk + 2]; b1 = bp[(size_t)(kk + 2) * p.ldw];
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
                        const int kn = kk + 4 < p.cout_p ? kk + 4 : kk + 2;
                        a0 = ap[kn]; b0 = bp[(size_t)kn * p.ldw];
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc, 0, 0, 0);
                    }
                    // epilogue: previous layer's activation derivative + BN-backward statistics
                    const int ci = ti * 32 + (lane & 31);
                    if (p.d_f) {
                        if (ci < p.cin) {
                            int cur_bn = -1; float run = 0.f;
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const long long r = row0 + rt * 32 + frag_row(e, lane);
                                if (r < p.rows) {
                                    int bn, bk; pair_row(bpt, rt * 32 + frag_row(e, lane), p.pair_N, p.pair_M, bn, bk);
                                    const float tv = acc[e];
                                    atomicAdd(p.d_g + (size_t)bk * p.cin + ci, tv * p.pair_f[(size_t)bn * p.cin + ci]);
                                    const float contrib = tv * p.pair_g[(size_t)bk * p.cin + ci];
                                    if (bn != cur_bn) {
                                        if (cur_bn >= 0) atomicAdd(p.d_f + (size_t)cur_bn * p.cin + ci, run);
                                        cur_bn = bn; run = 0.f;
                                    }
                                    run += contrib;
                                }
                            }
                            if (cur_bn >= 0) atomicAdd(p.d_f + (size_t)cur_bn * p.cin + ci, run);
                        }
                    } else if (ci < p.cin) {
                        float cm = 0.f, cs = 1.f, cb = 0.f, cinv = 1.f;
                        if (p.in_coef) { cm = Ci[ci]; cs = Ci[p.cin_p + ci]; cb = Ci[2 * p.cin_p + ci]; cinv = Ci[3 * p.cin_p + ci]; }
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const long long r = row0 + rt * 32 + frag_row(e, lane);
                            if (r < p.rows) {
                                float gv = acc[e];
                                if (p.in_coef) {
                                    const float xr = p.x[(size_t)r * p.cin + ci];
                                    const float z = (xr - cm) * cs + cb;
                                    gv = z > 0.f ? gv : gv * p.slope_in;
                                    const float xh = (xr - cm) * cinv;
                                    dsum[t] += gv; dsxh[t] += (double)gv * xh;
                                }
                                p.gz_in[(size_t)r * p.cin + ci] = gv;
                            }
                        }
                    }
                }
            }
        }
    }

    // ---- statistics of gz_in: lanes l and l+32 own the same channel ---------------------------------
    if (p.in_dsums && p.gz_in && p.in_coef) {
#pragma unroll
        for (int t = 0; t < DPW; ++t) {
            const int td = wave + 4 * t;
            double a = dsum[t] + __shfl_xor(dsum[t], 32);
            double b = dsxh[t] + __shfl_xor(dsxh[t], 32);
            if (td < DT && lane < 32) {
                const int ti = td % NTI, ci = ti * 32 + lane;
                if (ci < p.cin) {
                    double *rep = p.in_dsums + (size_t)(blockIdx.x % REP) * 2 * p.cin;
                    atomicAdd(rep + ci, a); atomicAdd(rep + p.cin + ci, b);
                }
            }
        }
    }
    // ---- per-block wgrad partial ----------------------------------------------------------------------
    float *part = p.dw_partial + (size_t)blockIdx.x * p.cout * p.cin;
#pragma unroll
    for (int t = 0; t < WPW; ++t) {
        const int tw = wave + 4 * t;
        if (tw < WT) {
            const int to = tw / NTI, ti = tw - to * NTI;
            const int ci = ti * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = to * 32 + frag_row(e, lane);
                if (co < p.cout && ci < p.cin) part[(size_t)co * p.cin + ci] = accw[t][e];
            }
        }
    }
}

// =================================================================================================
// Backward of the factored first cost-volume layer (pair mode), re-tiled for its reductions.
// A block owns one 64-pixel tile of one sample and a chunk of the points; it walks the points n
// with the pixel tile fixed, so the per-pixel sums (d_g, d_bias_k) accumulate in registers across
// the walk and the per-point sums (d_f, d_bias_n) are column sums of one tile:
//   grid = B x ceil(M/64) x NC,   atomics per block: 2*64*C at the end + 2*C per point.
// =================================================================================================
struct PairBwdParams {
    int B, N, M, NC, NL;               // NC point chunks of NL points
    int cin, cout, cin_p, cout_p, ldw, ldg, ldx;
    long long rows;
    const float *gz, *y, *out_coef, *out_mi;
    float slope_out;            // gz is dL/da of this layer's activation (slope_out): act'(z) applied on load; 1 = gz is dL/dz
    const double *out_dsums;
    const float *f, *g, *w;
    float *d_f, *d_g, *d_bn, *d_bk, *dw_partial;
    // deterministic accumulation: every cross-block sum goes through per-block slabs that one kernel adds up in a fixed
    // order (the first version used fp32 atomics: gradients differed run to run).  Slab s of a quantity is a full copy
    // of its [.., C] array; writers are unique per (slab, element).
    float *s_df;      // [2*KT][B*N*cin]   one slab per (pixel tile, 32-pixel half)
    float *s_dbn;     // [KT][B*N*cout]
    float *s_dg;      // [NC][B*M*cin]
    float *s_dbk;     // [NC][B*M*cout]
};

// out[i] = (((init + s0[i]) + s1[i]) + ...) over nslab slabs of n floats, in slab order
__global__ void slab_reduce_kernel(int nslab, long long n, const float *__restrict__ slabs, float *__restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float a = out[i];
        for (int s = 0; s < nslab; ++s) a += slabs[(size_t)s * n + i];
        out[i] = a;
    }
}

template <int NTI, int NTO>
__global__ __launch_bounds__(THREADS) void pair_bwd_kernel(PairBwdParams p) {
    extern __shared__ float smem[];
    float *Ws = smem;
    float *Gs = Ws + (size_t)p.cout_p * p.ldw;
    float *Xs = Gs + (size_t)BWD_R * p.ldg;
    float *Co = Xs + (size_t)BWD_R * p.ldx;                // [5][cout_p]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int DT = 2 * NTI, DPW = (DT + 3) / 4, WT = NTO * NTI, WPW = (WT + 3) / 4;
    constexpr int GCH = (BWD_R * NTO * 8 + THREADS - 1) / THREADS;    // G chunks (float4) per thread, cout = 32*NTO max
    constexpr int XCH = (BWD_R * NTI * 8 + THREADS - 1) / THREADS;

    const int KT = (p.M + BWD_R - 1) / BWD_R;
    int bid = blockIdx.x;
    const int nc = bid % p.NC; bid /= p.NC;
    const int kt = bid % KT; const int b = bid / KT;
    const int k0 = kt * BWD_R;
    const int n_begin = nc * p.NL, n_end = min(p.N, n_begin + p.NL);
    const int co4 = p.cout >> 2, ci4 = p.cin >> 2;
    const bool col_fix = (THREADS % co4) == 0;              // a thread's G chunks all carry the same 4 channels
    const bool x_fix = (THREADS % ci4) == 0;                // ... and its X' chunks the same 4 input channels

    for (int i = tid; i < p.cout_p * p.cin_p; i += THREADS) {
        const int co = i / p.cin_p, ci = i - co * p.cin_p;
        Ws[co * p.ldw + ci] = (co < p.cout && ci < p.cin) ? p.w[(size_t)co * p.cin + ci] : 0.f;
    }
    for (int ch = tid; ch < p.cout; ch += THREADS) {
        float m1 = 0.f, m2 = 0.f, sc = 1.f, mu = 0.f, is = 1.f;
        if (p.out_coef) {
            double sd = 0.0, sx = 0.0;
            for (int rp = 0; rp < REP; ++rp) { sd += p.out_dsums[(size_t)rp * 2 * p.cout + ch]; sx += p.out_dsums[(size_t)rp * 2 * p.cout + p.cout + ch]; }
            m1 = (float)(sd / (double)p.rows); m2 = (float)(sx / (double)p.rows);
            sc = p.out_coef[p.cout + ch]; mu = p.out_mi[ch]; is = p.out_mi[p.cout + ch];
        }
        Co[ch] = m1; Co[p.cout_p + ch] = m2; Co[2 * p.cout_p + ch] = sc; Co[3 * p.cout_p + ch] = mu; Co[4 * p.cout_p + ch] = is;
    }
    for (int i = tid; i < BWD_R * (p.cout_p - p.cout); i += THREADS) {
        const int r = i / (p.cout_p - p.cout), c = p.cout + i % (p.cout_p - p.cout);
        Gs[r * p.ldg + c] = 0.f;
    }
    for (int i = tid; i < BWD_R * (p.cin_p - p.cin); i += THREADS) {
        const int r = i / (p.cin_p - p.cin), c = p.cin + i % (p.cin_p - p.cin);
        Xs[r * p.ldx + c] = 0.f;
    }

    // pixel-tile operands that do not change along the walk: this thread's g chunks (staging layout)
    // and this lane's g values in the dgrad fragment layout
    float4 gk_chunk[XCH];
#pragma unroll
    for (int u = 0; u < XCH; ++u) {
        const int i = tid + u * THREADS, r = i / ci4, c4 = i - r * ci4;
        gk_chunk[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < BWD_R * ci4 && k0 + r < p.M)
            gk_chunk[u] = *reinterpret_cast<const float4 *>(p.g + ((size_t)b * p.M + k0 + r) * p.cin + c4 * 4);
    }
    float gk_frag[DPW][16];
#pragma unroll
    for (int t = 0; t < DPW; ++t) {
        const int td = wave + 4 * t, rt = td / NTI, ti = td - rt * NTI, ci = ti * 32 + (lane & 31);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = rt * 32 + frag_row(e, lane);
            gk_frag[t][e] = (td < DT && ci < p.cin && k0 + r < p.M) ? p.g[((size_t)b * p.M + k0 + r) * p.cin + ci] : 0.f;
        }
    }

    f32x16 accw[WPW];
#pragma unroll
    for (int t = 0; t < WPW; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) accw[t][e] = 0.f;
    float dg_acc[DPW][16];
#pragma unroll
    for (int t = 0; t < DPW; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) dg_acc[t][e] = 0.f;
    float4 dbk_acc[GCH];
#pragma unroll
    for (int u = 0; u < GCH; ++u) dbk_acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);

    // The G / y chunks and the f row of point n+1 are requested while point n's tiles are on the matrix cores: with one
    // wave per SIMD (132 KB of LDS per block) nothing else hides the load latency (the first version of this kernel
    // spent 3/4 of its time waiting: 1.58 ms for 0.39 ms of MFMA work).
    float *Rs = Co + 5 * p.cout_p;                          // [THREADS/co4][cout] partial column sums (d_bias_n)
    float4 gq[GCH], yq[GCH], f_next = make_float4(0.f, 0.f, 0.f, 0.f);
    float fd_next[DPW];
    auto prefetch = [&](int n) {
        const size_t bn = (size_t)b * p.N + n;
        const size_t rbase = bn * p.M + k0;
#pragma unroll
        for (int u = 0; u < GCH; ++u) {
            const int i = tid + u * THREADS, r = i / co4, c4 = i - r * co4;
            gq[u] = make_float4(0.f, 0.f, 0.f, 0.f); yq[u] = gq[u];
            if (i < BWD_R * co4 && k0 + r < p.M) {
                gq[u] = *reinterpret_cast<const float4 *>(p.gz + (rbase + r) * p.cout + c4 * 4);
                if (p.out_coef) yq[u] = *reinterpret_cast<const float4 *>(p.y + (rbase + r) * p.cout + c4 * 4);
            }
        }
        f_next = *reinterpret_cast<const float4 *>(p.f + bn * p.cin + (tid % ci4) * 4);
#pragma unroll
        for (int t = 0; t < DPW; ++t) {
            const int td = wave + 4 * t, rt = td / NTI, ti = td - rt * NTI, ci = ti * 32 + (lane & 31);
            fd_next[t] = (td < DT && ci < p.cin) ? p.f[bn * p.cin + ci] : 0.f;
        }
    };
    if (n_begin < n_end) prefetch(n_begin);

    for (int n = n_begin; n < n_end; ++n) {
        const size_t bn = (size_t)b * p.N + n;
        __syncthreads();
        // ---- stage G (BN-backward on load), accumulate the per-pixel bias gradient -------------------
        const float4 f_cur = f_next;
        float fd_cur[DPW];
#pragma unroll
        for (int t = 0; t < DPW; ++t) fd_cur[t] = fd_next[t];
        {
            float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);   // this thread's rows, its 4 channels (co4 divides THREADS)
#pragma unroll
            for (int u = 0; u < GCH; ++u) {
                const int i = tid + u * THREADS, r = i / co4, c4 = i - r * co4;
                if (i < BWD_R * co4) {
                    float gv[4] = {gq[u].x, gq[u].y, gq[u].z, gq[u].w};
                    if (p.out_coef && k0 + r < p.M) {
                        const float yy[4] = {yq[u].x, yq[u].y, yq[u].z, yq[u].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int ch = c4 * 4 + q;
                            const float xh = (yy[q] - Co[3 * p.cout_p + ch]) * Co[4 * p.cout_p + ch];
                            gv[q] = Co[2 * p.cout_p + ch] * (gv[q] - Co[ch] - xh * Co[p.cout_p + ch]);
                        }
                    }
                    dbk_acc[u].x += gv[0]; dbk_acc[u].y += gv[1]; dbk_acc[u].z += gv[2]; dbk_acc[u].w += gv[3];
                    csum.x += gv[0]; csum.y += gv[1]; csum.z += gv[2]; csum.w += gv[3];
                    float *dst = Gs + r * p.ldg + c4 * 4;
                    dst[0] = gv[0]; dst[1] = gv[1]; dst[2] = gv[2]; dst[3] = gv[3];
                }
            }
            if (col_fix) *reinterpret_cast<float4 *>(Rs + (tid / co4) * p.cout + (tid % co4) * 4) = csum;
        }
        // ---- stage X' = f[b,n,:] * g[b,k,:] ----------------------------------------------------------
#pragma unroll
        for (int u = 0; u < XCH; ++u) {
            const int i = tid + u * THREADS, r = i / ci4, c4 = i - r * ci4;
            if (i < BWD_R * ci4) {
                const float4 fv = x_fix ? f_cur : *reinterpret_cast<const float4 *>(p.f + bn * p.cin + c4 * 4);
                float *dst = Xs + r * p.ldx + c4 * 4;
                dst[0] = fv.x * gk_chunk[u].x; dst[1] = fv.y * gk_chunk[u].y; dst[2] = fv.z * gk_chunk[u].z; dst[3] = fv.w * gk_chunk[u].w;
            }
        }
        if (n + 1 < n_end) prefetch(n + 1);                 // lands while the MFMAs below run
        __syncthreads();
        // ---- per-point bias gradient: column sums of G ------------------------------------------------
        for (int ch = tid; ch < p.cout; ch += THREADS) {
            float s0 = 0.f;
            if (col_fix) {
                for (int q = 0; q < THREADS / co4; ++q) s0 += Rs[q * p.cout + ch];
            } else {
                for (int r = 0; r < BWD_R; ++r) s0 += Gs[r * p.ldg + ch];
            }
            p.s_dbn[((size_t)kt * p.B * p.N + bn) * p.cout + ch] = s0;
        }
        // ---- wgrad --------------------------------------------------------------------------------------
#pragma unroll
        for (int t = 0; t < WPW; ++t) {
            const int tw = wave + 4 * t;
            if (tw < WT) {
                const int to = tw / NTI, ti = tw - to * NTI;
                const float *ap = Gs + (lane >> 5) * p.ldg + to * 32 + (lane & 31);
                const float *bp = Xs + (lane >> 5) * p.ldx + ti * 32 + (lane & 31);
                // ping-pong fragments, two k-steps per iteration: operands are requested a full MFMA (64 cycles) before use
                float a0 = ap[0], b0 = bp[0], a1, b1;
#pragma unroll 4
                for (int kk = 0; kk < BWD_R; kk += 4) {
                    a1 = ap[(size_t)(kk + 2) * p.ldg]; b1 = bp[(size_t)(kk + 2) * p.ldx];
                    accw[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, accw[t], 0, 0, 0);
                    const int kn = kk + 4 < BWD_R ? kk + 4 : kk + 2;
                    a0 = ap[(size_t)kn * p.ldg]; b0 = bp[(size_t)kn * p.ldx];
                    accw[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, accw[t], 0, 0, 0);
                }
            }
        }
        // ---- dgrad T = G . W, folded into d_g (registers) and d_f (column sums -> atomics) -----------------
#pragma unroll
        for (int t = 0; t < DPW; ++t) {
            const int td = wave + 4 * t;
            if (td < DT) {
                const int rt = td / NTI, ti = td - rt * NTI;
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
                const float *ap = Gs + (rt * 32 + (lane & 31)) * p.ldg + (lane >> 5);
                const float *bp = Ws + (lane >> 5) * p.ldw + ti * 32 + (lane & 31);
                float a0 = ap[0], b0 = bp[0], a1, b1;
#pragma unroll 4
                for (int kk = 0; kk < p.cout_p; kk += 4) {           // cout_p is a multiple of 32
                    a1 = ap[kk + 2]; b1 = bp[(size_t)(kk + 2) * p.ldw];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
                    const int kn = kk + 4 < p.cout_p ? kk + 4 : kk + 2;
                    a0 = ap[kn]; b0 = bp[(size_t)kn * p.ldw];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc, 0, 0, 0);
                }
                const int ci = ti * 32 + (lane & 31);
                if (ci < p.cin) {
                    const float fv = fd_cur[t];
                    float colsum = 0.f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {           // rows beyond M hold G = 0 => T = 0
                        dg_acc[t][e] += acc[e] * fv;
                        colsum += acc[e] * gk_frag[t][e];
                    }
                    // lanes l / l+32 hold pixel rows 4h.. of the same column: fixed order (h = 0 first)
                    const float other = __shfl_xor(colsum, 32);
                    if (lane < 32) p.s_df[((size_t)(kt * 2 + rt) * p.B * p.N + bn) * p.cin + ci] = colsum + other;
                }
            }
        }
    }

    // ---- flush the per-pixel accumulators ------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < DPW; ++t) {
        const int td = wave + 4 * t;
        if (td < DT) {
            const int rt = td / NTI, ti = td - rt * NTI, ci = ti * 32 + (lane & 31);
            if (ci < p.cin)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int r = rt * 32 + frag_row(e, lane);
                    if (k0 + r < p.M) p.s_dg[(((size_t)nc * p.B + b) * p.M + k0 + r) * p.cin + ci] = dg_acc[t][e];
                }
        }
    }
#pragma unroll
    for (int u = 0; u < GCH; ++u) {
        const int i = tid + u * THREADS, r = i / co4, c4 = i - r * co4;
        if (i < BWD_R * co4 && k0 + r < p.M) {
            float *dk = p.s_dbk + (((size_t)nc * p.B + b) * p.M + k0 + r) * p.cout + c4 * 4;
            *reinterpret_cast<float4 *>(dk) = dbk_acc[u];
        }
    }
    float *part = p.dw_partial + (size_t)blockIdx.x * p.cout * p.cin;
#pragma unroll
    for (int t = 0; t < WPW; ++t) {
        const int tw = wave + 4 * t;
        if (tw < WT) {
            const int to = tw / NTI, ti = tw - to * NTI;
            const int ci = ti * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = to * 32 + frag_row(e, lane);
                if (co < p.cout && ci < p.cin) part[(size_t)co * p.cin + ci] = accw[t][e];
            }
        }
    }
}

// =================================================================================================
// Backward, second generation: dgrad runs on the lin_fwd2 template (DGRAD mode); wgrad is this
// kernel.  dW[co][ci] = sum_r g^y[r][co] * x'[r][ci] with both operands formed on load
// (BN-backward of (gz, y); BN + activation of x).  8 waves, double-buffered 64-row G / X' tiles in
// LDS (no weights needed => 2 x 66 KB fits), one barrier per tile; within a barrier interval each
// wave stages the next tile, issues the loads of the one after, and runs its MFMAs, so the two
// waves of a SIMD overlap memory and matrix phases.
// =================================================================================================
struct WgradParams {
    long long rows;
    int cin, cout, cin_p, cout_p, ldg, ldx;
    const float *gz, *y, *g_coef;        // g_coef: non-null = BN behind (constants formed in the prologue), nullptr: gz already is dL/dy
    const double *g_dsums; const float *g_oc, *g_omi; long long g_rows;   // raw sources, as in LinFwdParams
    float *bn_out;                       // [8][cout] scratch tail: rows 6, 7 receive dbeta = sum gz, dgamma = sum gz*xhat (block 0)
    float g_slope;                       // != 1: gz is dL/da of an activation with this slope (act' applied on load)
    const float *x, *in_coef;            // in_coef [3][cin] or nullptr
    float slope_in;
    float *dw_partial;
    int split_c;                         // two-source input: channels >= split_c come from xb / in_coef_b / slope_b
    const float *xb, *in_coef_b;
    float slope_b;
};

constexpr int WG_THREADS = 512;
constexpr int WG_R = 64;

// (bnbwd_coef_kernel — a 64-thread launch that turned the replica sums {sum gz, sum gz*xhat} of a BN into its backward constants
// [8][c] — is gone since round 5: every consumer forms them in its prologue and its block 0 writes rows 6, 7 = dbeta, dgamma)

template <int NTI, int NTO, bool FIXC>
__global__ __launch_bounds__(WG_THREADS, 2) void lin_wgrad_kernel(WgradParams p) {
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bufsz = WG_R * (p.ldg + p.ldx);
    float *Ct = smem + 2 * (size_t)bufsz;                  // [5][cout] then [3][cin] constants
    float *Cg = Ct, *Cx = Ct + 6 * p.cout;
    constexpr int WT = NTO * NTI, WPW = (WT + 7) / 8;
    constexpr int GCH = (WG_R * NTO * 8 + WG_THREADS - 1) / WG_THREADS;   // float4 chunks per thread
    constexpr int XCH = (WG_R * NTI * 8 + WG_THREADS - 1) / WG_THREADS;
    const int co4 = p.cout >> 2, ci4 = p.cin >> 2;

    for (int ch = tid; ch < p.cout; ch += WG_THREADS) {
        float m1 = 0.f, m2 = 0.f, sc = 0.f, mu = 0.f, is = 0.f, be = 0.f;
        if (p.g_coef) {
            double sd = 0.0, sx = 0.0;
#pragma unroll 8
            for (int rp = 0; rp < REP; ++rp) { sd += p.g_dsums[(size_t)rp * 2 * p.cout + ch]; sx += p.g_dsums[(size_t)rp * 2 * p.cout + p.cout + ch]; }
            m1 = (float)(sd / (double)p.g_rows); m2 = (float)(sx / (double)p.g_rows);
            sc = p.g_oc[p.cout + ch]; mu = p.g_omi[ch]; is = p.g_omi[p.cout + ch]; be = p.g_oc[2 * p.cout + ch];
            if (blockIdx.x == 0 && p.bn_out) { p.bn_out[6 * p.cout + ch] = (float)sd; p.bn_out[7 * p.cout + ch] = (float)sx; }
        }
        Cg[ch] = m1; Cg[p.cout + ch] = m2; Cg[2 * p.cout + ch] = sc; Cg[3 * p.cout + ch] = mu; Cg[4 * p.cout + ch] = is; Cg[5 * p.cout + ch] = be;
    }
    // (two-source input: in_coef describes only the first split_c channels; that mode takes its constants from
    //  registers, the table is not used — do not read past the end of the shorter coefficient array)
    for (int i = tid; i < 3 * p.cin; i += WG_THREADS) Cx[i] = (p.in_coef && !p.xb) ? p.in_coef[i] : 0.f;
    // zero the padding columns of both buffers once
    for (int b = 0; b < 2; ++b) {
        float *Gs = smem + (size_t)b * bufsz, *Xs = Gs + WG_R * p.ldg;
        for (int i = tid; i < WG_R * (p.cout_p - p.cout); i += WG_THREADS)
            Gs[(i / (p.cout_p - p.cout)) * p.ldg + p.cout + i % (p.cout_p - p.cout)] = 0.f;
        for (int i = tid; i < WG_R * (p.cin_p - p.cin); i += WG_THREADS)
            Xs[(i / (p.cin_p - p.cin)) * p.ldx + p.cin + i % (p.cin_p - p.cin)] = 0.f;
    }
    __syncthreads();

    const long long ntiles = (p.rows + WG_R - 1) / WG_R;
    const long long last_row = p.rows - 1;

    // chunk geometry and per-channel constants of this thread: fixed for the whole kernel when the float4
    // column count divides the block size (all power-of-two widths); otherwise constants come from the LDS table
    constexpr bool g_fix = FIXC, x_fix = FIXC;
    float4 k_m1 = make_float4(0.f, 0.f, 0.f, 0.f), k_m2 = k_m1, k_sc = make_float4(1.f, 1.f, 1.f, 1.f), k_mu = k_m1, k_is = k_sc, k_be = k_m1;
    const bool g_act = p.g_coef && p.g_slope != 1.f;
    float4 x_mu = k_m1, x_sc = k_sc, x_be = k_m1;
    if (p.g_coef && g_fix) {
        const int c4 = tid % co4;
        k_m1 = *reinterpret_cast<const float4 *>(Cg + c4 * 4); k_m2 = *reinterpret_cast<const float4 *>(Cg + p.cout + c4 * 4);
        k_sc = *reinterpret_cast<const float4 *>(Cg + 2 * p.cout + c4 * 4); k_mu = *reinterpret_cast<const float4 *>(Cg + 3 * p.cout + c4 * 4);
        k_is = *reinterpret_cast<const float4 *>(Cg + 4 * p.cout + c4 * 4);
        k_be = *reinterpret_cast<const float4 *>(Cg + 5 * p.cout + c4 * 4);
    }
    // source of this thread's X' chunks (two-source mode needs the per-thread-constant layout)
    const bool x_b = p.xb && (tid % ci4) * 4 >= p.split_c;
    const float *xs_ptr = x_b ? p.xb : p.x;
    const int xs_ld = p.xb ? (x_b ? p.cin - p.split_c : p.split_c) : p.cin;
    const int xs_c0 = x_b ? (tid % ci4) * 4 - p.split_c : (tid % ci4) * 4;
    const float *xs_coef = x_b ? p.in_coef_b : p.in_coef;
    const float xs_slope = x_b ? p.slope_b : p.slope_in;
    if (xs_coef && x_fix) {
        x_mu = *reinterpret_cast<const float4 *>(xs_coef + xs_c0); x_sc = *reinterpret_cast<const float4 *>(xs_coef + xs_ld + xs_c0);
        x_be = *reinterpret_cast<const float4 *>(xs_coef + 2 * xs_ld + xs_c0);
    }

    float4 rg[GCH], ry[GCH], rx[XCH];
    auto fetch = [&](long long tile) {
        const long long row0 = tile * WG_R;
#pragma unroll
        for (int u = 0; u < GCH; ++u) {
            int i = tid + u * WG_THREADS; if (i >= WG_R * co4) i = WG_R * co4 - 1;
            const int r = i / co4, c4 = i - r * co4;
            long long row = row0 + r; if (row > last_row) row = last_row;
            rg[u] = *reinterpret_cast<const float4 *>(p.gz + (size_t)row * p.cout + c4 * 4);
            if (p.g_coef) ry[u] = *reinterpret_cast<const float4 *>(p.y + (size_t)row * p.cout + c4 * 4);
        }
#pragma unroll
        for (int u = 0; u < XCH; ++u) {
            int i = tid + u * WG_THREADS; if (i >= WG_R * ci4) i = WG_R * ci4 - 1;
            const int r = i / ci4, c4 = i - r * ci4;
            long long row = row0 + r; if (row > last_row) row = last_row;
            rx[u] = x_fix ? *reinterpret_cast<const float4 *>(xs_ptr + (size_t)row * xs_ld + xs_c0)
                          : *reinterpret_cast<const float4 *>(p.x + (size_t)row * p.cin + c4 * 4);
        }
    };
    auto commit = [&](long long tile, int buf) {
        const long long row0 = tile * WG_R;
        float *Gs = smem + (size_t)buf * bufsz, *Xs = Gs + WG_R * p.ldg;
#pragma unroll
        for (int u = 0; u < GCH; ++u) {
            const int i = tid + u * WG_THREADS;
            if (i < WG_R * co4) {
                const int r = i / co4, c4 = i - r * co4;
                float4 g = rg[u];
                if (row0 + r > last_row) g = make_float4(0.f, 0.f, 0.f, 0.f);         // rows past the end contribute nothing
                else if (p.g_coef) {
                    float4 m1 = k_m1, m2 = k_m2, sc = k_sc, mu = k_mu, is = k_is, be = k_be;
                    if constexpr (!g_fix) {
                        be = *reinterpret_cast<const float4 *>(Cg + 5 * p.cout + c4 * 4);
                        m1 = *reinterpret_cast<const float4 *>(Cg + c4 * 4); m2 = *reinterpret_cast<const float4 *>(Cg + p.cout + c4 * 4);
                        sc = *reinterpret_cast<const float4 *>(Cg + 2 * p.cout + c4 * 4); mu = *reinterpret_cast<const float4 *>(Cg + 3 * p.cout + c4 * 4);
                        is = *reinterpret_cast<const float4 *>(Cg + 4 * p.cout + c4 * 4);
                    }
                    const float4 yv = ry[u];
                    if (g_act) {                                                      // dL/da -> dL/dz
                        g.x = (yv.x - mu.x) * sc.x + be.x > 0.f ? g.x : g.x * p.g_slope; g.y = (yv.y - mu.y) * sc.y + be.y > 0.f ? g.y : g.y * p.g_slope;
                        g.z = (yv.z - mu.z) * sc.z + be.z > 0.f ? g.z : g.z * p.g_slope; g.w = (yv.w - mu.w) * sc.w + be.w > 0.f ? g.w : g.w * p.g_slope;
                    }
                    g.x = sc.x * (g.x - m1.x - ((yv.x - mu.x) * is.x) * m2.x); g.y = sc.y * (g.y - m1.y - ((yv.y - mu.y) * is.y) * m2.y);
                    g.z = sc.z * (g.z - m1.z - ((yv.z - mu.z) * is.z) * m2.z); g.w = sc.w * (g.w - m1.w - ((yv.w - mu.w) * is.w) * m2.w);
                }
                *reinterpret_cast<float4 *>(Gs + r * p.ldg + c4 * 4) = g;
            }
        }
#pragma unroll
        for (int u = 0; u < XCH; ++u) {
            const int i = tid + u * WG_THREADS;
            if (i < WG_R * ci4) {
                const int r = i / ci4, c4 = i - r * ci4;
                float4 v = rx[u];
                if (x_fix ? (xs_coef != nullptr) : (p.in_coef != nullptr)) {
                    float4 mu = x_mu, sc = x_sc, be = x_be;
                    if constexpr (!x_fix) {
                        mu = *reinterpret_cast<const float4 *>(Cx + c4 * 4); sc = *reinterpret_cast<const float4 *>(Cx + p.cin + c4 * 4);
                        be = *reinterpret_cast<const float4 *>(Cx + 2 * p.cin + c4 * 4);
                    }
                    const float sl = x_fix ? xs_slope : p.slope_in;
                    v.x = act_apply((v.x - mu.x) * sc.x + be.x, sl); v.y = act_apply((v.y - mu.y) * sc.y + be.y, sl);
                    v.z = act_apply((v.z - mu.z) * sc.z + be.z, sl); v.w = act_apply((v.w - mu.w) * sc.w + be.w, sl);
                }
                *reinterpret_cast<float4 *>(Xs + r * p.ldx + c4 * 4) = v;
            }
        }
    };

    f32x16 acc[WPW];
#pragma unroll
    for (int t = 0; t < WPW; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    long long tile = blockIdx.x;
    int cur = 0;
    if (tile < ntiles) {
        fetch(tile); commit(tile, 0);
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);
    }
    __syncthreads();
    for (; tile < ntiles; tile += gridDim.x, cur ^= 1) {
        if (tile + gridDim.x < ntiles) commit(tile + gridDim.x, cur ^ 1);
        if (tile + 2 * (long long)gridDim.x < ntiles) fetch(tile + 2 * (long long)gridDim.x);
        const float *Gs = smem + (size_t)cur * bufsz, *Xs = Gs + WG_R * p.ldg;
#pragma unroll
        for (int t = 0; t < WPW; ++t) {
            const int tw = wave + 8 * t;
            if (tw < WT) {
                const int to = tw / NTI, ti = tw - to * NTI;
                const float *ap = Gs + (lane >> 5) * p.ldg + to * 32 + (lane & 31);
                const float *bp = Xs + (lane >> 5) * p.ldx + ti * 32 + (lane & 31);
                float a0 = ap[0], b0 = bp[0], a1, b1;
#pragma unroll 4
                for (int kk = 0; kk < WG_R; kk += 4) {
                    a1 = ap[(size_t)(kk + 2) * p.ldg]; b1 = bp[(size_t)(kk + 2) * p.ldx];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[t], 0, 0, 0);
                    const int kn = kk + 4 < WG_R ? kk + 4 : kk + 2;
                    a0 = ap[(size_t)kn * p.ldg]; b0 = bp[(size_t)kn * p.ldx];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    float *part = p.dw_partial + (size_t)blockIdx.x * p.cout * p.cin;
#pragma unroll
    for (int t = 0; t < WPW; ++t) {
        const int tw = wave + 8 * t;
        if (tw < WT) {
            const int to = tw / NTI, ti = tw - to * NTI;
            const int ci = ti * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = to * 32 + frag_row(e, lane);
                if (co < p.cout && ci < p.cin) part[(size_t)co * p.cin + ci] = acc[t][e];
            }
        }
    }
}

// dW = sum over blocks of the per-block partials: 32 outputs x 8 partial-lanes per block
__global__ __launch_bounds__(256) void reduce_partials_kernel(int nparts, int n, const float *__restrict__ parts,
                                                               float *__restrict__ out) {
    __shared__ float red[8][32];
    const int o = blockIdx.x * 32 + (threadIdx.x & 31), pl = threadIdx.x >> 5;
    float a = 0.f;
    if (o < n)
        for (int b = pl; b < nparts; b += 8) a += parts[(size_t)b * n + o];
    red[pl][threadIdx.x & 31] = a;
    __syncthreads();
    if (pl == 0 && o < n) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += red[q][threadIdx.x & 31];
        out[o] = t;
    }
}

// the same sum on 16-byte columns: 16 float4 columns x 16 partial lanes per block (a wave reads four 256-byte runs per step,
// four steps in flight) — the 256 x 64 KB partials of a 128 x 128 layer are read at HBM rate instead of 128-byte touches
__global__ __launch_bounds__(256) void reduce_partials_v4_kernel(int nparts, int n4, const float4 *__restrict__ parts, float4 *__restrict__ out) {
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
        for (int q = 0; q < 16; ++q) { const float4 v = red[q][tx]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        out[o] = t;
    }
}

inline void launch_reduce_partials(int nparts, int n, const float *parts, float *out, hipStream_t st) {
    if ((n & 3) == 0 && ((reinterpret_cast<uintptr_t>(parts) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        const int n4 = n >> 2;
        if (i2p_defer_reduce(0, nparts, n4, parts, out)) return;           // summed by i2p_defer_flush with the step's other weight gradients
        hipLaunchKernelGGL(reduce_partials_v4_kernel, dim3((n4 + 15) / 16), dim3(256), 0, st, nparts, n4, reinterpret_cast<const float4 *>(parts),
                           reinterpret_cast<float4 *>(out));
    } else {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((n + 31) / 32), dim3(256), 0, st, nparts, n, parts, out);
    }
}

template <int NTI, int NTO>
int launch_bwd(LinBwdParams &p, float *dw, hipStream_t st, unsigned grid) {
    const size_t bytes = ((size_t)p.cout_p * p.ldw + (size_t)BWD_R * p.ldg + (size_t)BWD_R * p.ldx + 6 * (size_t)p.cout_p + 4 * (size_t)p.cin_p) * sizeof(float);
    if (bytes > 160 * 1024) return I2P_ERR_BAD_ARG;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lin_bwd_kernel<NTI, NTO>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((lin_bwd_kernel<NTI, NTO>), dim3(grid), dim3(THREADS), bytes, st, p);
    const int n = p.cout * p.cin;
    launch_reduce_partials((int)grid, n, p.dw_partial, dw, st);
    I2P_RETURN_LAUNCH_STATUS();
}

template <int NTI>
int dispatch_bwd_o(LinBwdParams &p, float *dw, hipStream_t st, unsigned grid) {
    switch (p.cout_p / 32) {
        case 1: return launch_bwd<NTI, 1>(p, dw, st, grid);
        case 2: return launch_bwd<NTI, 2>(p, dw, st, grid);
        case 3: return launch_bwd<NTI, 3>(p, dw, st, grid);
        case 4: return launch_bwd<NTI, 4>(p, dw, st, grid);
        default: return I2P_ERR_BAD_ARG;
    }
}

}  // namespace

struct FinArgs { unsigned *counter; const float *gamma, *beta; float eps; float *coef, *mi; };

static int lin_fwd_impl(long long rows, int cin, int cout, const float *x, const float *in_coef, float slope_in,
                        const float *w, float *y, double *sums, const float *pair_f, const float *bias_n,
                        const float *bias_k, int pair_N, int pair_M, void *stream, const float *xb = nullptr,
                        const float *in_coef_b = nullptr, float slope_b = 1.f, int split_c = 0, const FinArgs *fin = nullptr) {
    if (fin && (!sums || !fin->counter || !fin->gamma || !fin->beta || !fin->coef || !fin->mi)) return I2P_ERR_BAD_ARG;
    // the statistics are finalised by the layer kernel's last block when ONE second-generation launch covers the layer;
    // otherwise by the small kernel at the end of this function
    bool fin_done = false;
    auto finish = [&](int rc) -> int {
        if (rc || !fin || fin_done) return rc;
        hipLaunchKernelGGL(bn_finalize_kernel, dim3((cout + 63) / 64), dim3(64), 0, (hipStream_t)stream, rows, cout, sums, fin->gamma,
                           fin->beta, fin->eps, fin->coef, fin->mi);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : (int)e;
    };
    if (rows < 0 || cin <= 0 || cout <= 0 || cout > 320) return I2P_ERR_BAD_ARG;
    if (rows == 0) return 0;
    if (!x || !w || !y) return I2P_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (!pair_f && !xb && i2p_big_layer_ok(rows, cin, cout))        // wide layer on few rows: K-tiled (csrc/mlp_big.hip)
        return finish(i2p_big_fwd(rows, cin, cout, x, in_coef, slope_in, w, y, sums, stream));
    if (cout > 256) return I2P_ERR_BAD_ARG;
    if (pair_f && !xb && !in_coef && cin == 128 && cout == 128 && pair_N > 0 && pair_M > 0 && rows % ((long long)pair_N * pair_M) == 0 &&
        i2p_wreg_pair_bwd_ok((int)(rows / ((long long)pair_N * pair_M)), pair_N, pair_M, cin, cout) && rows * 512 < (1LL << 32)) {
        fin_done = fin != nullptr;                          // first cost-volume layer, 128 x 128 on many rows (csrc/mlp_wreg.hip)
        return i2p_wreg_pair_fwd((int)(rows / ((long long)pair_N * pair_M)), pair_N, pair_M, cin, cout, pair_f, x, bias_n, bias_k, w, y, sums,
                                 fin ? fin->counter : nullptr, fin ? fin->gamma : nullptr, fin ? fin->beta : nullptr, fin ? fin->eps : 0.f,
                                 fin ? fin->coef : nullptr, fin ? fin->mi : nullptr, stream);
    }
    const bool two_ok3 = !xb || (split_c * 2 == cin && in_coef && in_coef_b && slope_b >= 0.f && slope_b <= 1.f);
    if (!pair_f && two_ok3 && i2p_wreg_fwd_ok(rows, cin, cout) && slope_in >= 0.f && slope_in <= 1.f) {  // wide layer on many rows: weights stationary in registers
        fin_done = fin != nullptr;
        return i2p_wreg_fwd(rows, cin, cout, x, xb ? split_c : cin, in_coef, slope_in, w, y, cout, sums, fin ? fin->counter : nullptr,
                            fin ? fin->gamma : nullptr, fin ? fin->beta : nullptr, fin ? fin->eps : 0.f, fin ? fin->coef : nullptr,
                            fin ? fin->mi : nullptr, stream, xb, in_coef_b, slope_b);
    }
    {
        auto pow2_16_128 = [](int c) { return c == 16 || c == 32 || c == 64 || c == 128; };
        const int slice_w = cout > 128 ? 128 : cout;
        const bool gen2_ok = (cin % 4 == 0) && cin <= 128 && cin >= 4 && (cout % slice_w == 0) && pow2_16_128(slice_w) &&
                             (!pair_f || pair_M >= F2_ROWS) &&
                             ((!in_coef && !in_coef_b) || 64 % (cin / 4) == 0) && (!xb || (split_c % 4 == 0 && split_c > 0 && split_c < cin && 64 % (cin / 4) == 0));
        if (gen2_ok) {
            // strip LDS rows hold max(cin, slice) floats (+2): the epilogue transposes the outputs through them
            for (int off = 0; off < cout; off += slice_w) {
                LinFwdParams p;
                p.rows = rows; p.cin = cin; p.cout = slice_w;
                p.cin_p = cin; p.cout_p = slice_w;
                p.ldk = (cin > p.cout_p ? cin : p.cout_p) + ((cin & 15) == 0 ? 4 : 2);   // +4: 16-byte rows, ldk = 4 mod 32 (wide-K fragments)
                p.x = x; p.in_coef = in_coef; p.slope_in = slope_in; p.w = w + (size_t)off * cin; p.y = y; p.sums = sums;
                p.y_ld = cout; p.ch_off = off; p.cout_total = cout;
                p.ablate = 0;
                p.pair_f = pair_f; p.bias_n = bias_n; p.bias_k = bias_k; p.pair_N = pair_N; p.pair_M = pair_M;
                p.x2 = nullptr; p.g_coef = nullptr; p.g_dsums = nullptr; p.g_oc = p.g_omi = nullptr; p.g_rows = 1; p.w_transposed = 0; p.ex = p.e_coef = p.e_mi = nullptr; p.e_slope = 1.f;
                p.split_c = 0; p.xb = xb; p.in_coef_b = in_coef_b; p.slope_b = slope_b; p.yb = nullptr; p.exb = p.e_coef_b = p.e_mi_b = p.e_add = nullptr; p.sums_b = nullptr; p.e_slope_b = 1.f; p.split_c = split_c;
                p.fin_counter = nullptr; p.fin_gamma = p.fin_beta = nullptr; p.fin_eps = 0.f; p.fin_coef = p.fin_mi = nullptr;
                if (fin && slice_w == cout) {
                    p.fin_counter = fin->counter; p.fin_gamma = fin->gamma; p.fin_beta = fin->beta; p.fin_eps = fin->eps;
                    p.fin_coef = fin->coef; p.fin_mi = fin->mi; fin_done = true;
                }
                const int rc = pair_f ? dispatch_fwd2<true, false>(p, st) : dispatch_fwd2<false, false>(p, st);
                if (rc) return rc;
            }
            return finish(0);
        }
    }
    if (xb) return I2P_ERR_BAD_ARG;                        // two-source input exists only in the second generation
    // output channels are processed in slices whose weights fit the LDS next to a 128-row tile
    const int cin_p = (cin + 1) & ~1, ldk = cin_p + 1;
    int slice = (cout + 31) & ~31;
    while (slice > 32 && ((size_t)slice + 128) * ldk * sizeof(float) > 160 * 1024) slice -= 32;
    if (((size_t)slice + 128) * ldk * sizeof(float) > 160 * 1024) return I2P_ERR_BAD_ARG;
    for (int off = 0; off < cout; off += slice) {
        LinFwdParams p;
        p.rows = rows; p.cin = cin; p.cout = (cout - off < slice) ? cout - off : slice;
        p.cin_p = cin_p; p.cout_p = (p.cout + 31) & ~31; p.ldk = ldk;
        p.x = x; p.in_coef = in_coef; p.slope_in = slope_in; p.w = w + (size_t)off * cin; p.y = y; p.sums = sums;
        p.y_ld = cout; p.ch_off = off; p.cout_total = cout;
        p.pair_f = pair_f; p.bias_n = bias_n; p.bias_k = bias_k; p.pair_N = pair_N; p.pair_M = pair_M;
        p.ablate = 0;
        p.x2 = nullptr; p.g_coef = nullptr; p.g_dsums = nullptr; p.g_oc = p.g_omi = nullptr; p.g_rows = 1; p.w_transposed = 0; p.ex = p.e_coef = p.e_mi = nullptr; p.e_slope = 1.f;
                p.split_c = 0; p.xb = xb; p.in_coef_b = in_coef_b; p.slope_b = slope_b; p.yb = nullptr; p.exb = p.e_coef_b = p.e_mi_b = p.e_add = nullptr; p.sums_b = nullptr; p.e_slope_b = 1.f; p.split_c = split_c;
        p.fin_counter = nullptr; p.fin_gamma = p.fin_beta = nullptr; p.fin_eps = 0.f; p.fin_coef = p.fin_mi = nullptr;
        int rc;
        switch (p.cout_p / 32) {
            case 1: rc = launch_fwd<1>(p, st); break;
            case 2: rc = launch_fwd<2>(p, st); break;
            case 3: rc = launch_fwd<3>(p, st); break;
            case 4: rc = launch_fwd<4>(p, st); break;
            case 5: rc = launch_fwd<5>(p, st); break;
            case 6: rc = launch_fwd<6>(p, st); break;
            case 7: rc = launch_fwd<7>(p, st); break;
            default: rc = launch_fwd<8>(p, st); break;
        }
        if (rc) return rc;
    }
    return finish(0);
}

extern "C" int i2p_lin_fwd(long long rows, int cin, int cout, const float *x, const float *in_coef,
                           float slope_in, const float *w, float *y, double *sums, void *stream) {
    return lin_fwd_impl(rows, cin, cout, x, in_coef, slope_in, w, y, sums, nullptr, nullptr, nullptr, 1, 1, stream);
}

extern "C" int i2p_lin_fwd_2src(long long rows, int cin_a, int cin_b, int cout, const float *xa, const float *coef_a,
                                float slope_a, const float *xb, const float *coef_b, float slope_b, const float *w,
                                float *y, double *sums, void *stream) {
    if (!xa || !xb || cin_a <= 0 || cin_b <= 0) return I2P_ERR_BAD_ARG;
    return lin_fwd_impl(rows, cin_a + cin_b, cout, xa, coef_a, slope_a, w, y, sums, nullptr, nullptr, nullptr, 1, 1, stream,
                        xb, coef_b, slope_b, cin_a);
}

extern "C" int i2p_pair_lin_fwd(int B, int N, int M, int cin, int cout, const float *f, const float *g,
                                const float *bias_n, const float *bias_k, const float *w, float *y, double *sums,
                                void *stream) {
    if (B <= 0 || N <= 0 || M <= 0 || (cin & 3) || cin > 128 || !f || !bias_n || !bias_k) return I2P_ERR_BAD_ARG;
    return lin_fwd_impl((long long)B * N * M, cin, cout, g, nullptr, 1.0f, w, y, sums, f, bias_n, bias_k, N, M, stream);
}

// layer + BN finalisation in one call (`_fin`: coef [3][cout], mean_invstd [2*cout] come back with y; counter = one
// zeroed uint32 of caller scratch)
extern "C" int i2p_lin_fwd_fin(long long rows, int cin, int cout, const float *x, const float *in_coef, float slope_in,
                               const float *w, float *y, double *sums, const float *gamma, const float *beta, float eps,
                               float *coef, float *mean_invstd, unsigned *counter, void *stream) {
    const FinArgs f{counter, gamma, beta, eps, coef, mean_invstd};
    return lin_fwd_impl(rows, cin, cout, x, in_coef, slope_in, w, y, sums, nullptr, nullptr, nullptr, 1, 1, stream, nullptr, nullptr, 1.f, 0, &f);
}

extern "C" int i2p_lin_fwd_2src_fin(long long rows, int cin_a, int cin_b, int cout, const float *xa, const float *coef_a,
                                    float slope_a, const float *xb, const float *coef_b, float slope_b, const float *w,
                                    float *y, double *sums, const float *gamma, const float *beta, float eps, float *coef,
                                    float *mean_invstd, unsigned *counter, void *stream) {
    if (!xa || !xb || cin_a <= 0 || cin_b <= 0) return I2P_ERR_BAD_ARG;
    const FinArgs f{counter, gamma, beta, eps, coef, mean_invstd};
    return lin_fwd_impl(rows, cin_a + cin_b, cout, xa, coef_a, slope_a, w, y, sums, nullptr, nullptr, nullptr, 1, 1, stream,
                        xb, coef_b, slope_b, cin_a, &f);
}

extern "C" int i2p_pair_lin_fwd_fin(int B, int N, int M, int cin, int cout, const float *f, const float *g,
                                    const float *bias_n, const float *bias_k, const float *w, float *y, double *sums,
                                    const float *gamma, const float *beta, float eps, float *coef, float *mean_invstd,
                                    unsigned *counter, void *stream) {
    if (B <= 0 || N <= 0 || M <= 0 || (cin & 3) || cin > 128 || !f || !bias_n || !bias_k) return I2P_ERR_BAD_ARG;
    const FinArgs fa{counter, gamma, beta, eps, coef, mean_invstd};
    return lin_fwd_impl((long long)B * N * M, cin, cout, g, nullptr, 1.0f, w, y, sums, f, bias_n, bias_k, N, M, stream, nullptr,
                        nullptr, 1.f, 0, &fa);
}

extern "C" int i2p_bn_finalize(long long rows, int c, const double *sums, const float *gamma,
                               const float *beta, float eps, float *coef, float *mean_invstd, void *stream) {
    if (rows <= 0 || c <= 0 || !sums || !gamma || !beta || !coef) return I2P_ERR_BAD_ARG;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((c + 63) / 64), dim3(64), 0, (hipStream_t)stream, rows, c, sums, gamma,
                       beta, eps, coef, mean_invstd);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_lin_bwd_grid(long long rows) {
    const long long ntiles = (rows + BWD_R - 1) / BWD_R;
    return (int)(ntiles < 256 ? (ntiles < 1 ? 1 : ntiles) : 256);
}

template <int NTI, int NTO, bool FIXC>
static int launch_wgrad_t(const WgradParams &q, float *dw, hipStream_t st, unsigned grid, size_t bytes) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lin_wgrad_kernel<NTI, NTO, FIXC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((lin_wgrad_kernel<NTI, NTO, FIXC>), dim3(grid), dim3(WG_THREADS), bytes, st, q);
    const int n = q.cout * q.cin;
    launch_reduce_partials((int)grid, n, q.dw_partial, dw, st);
    I2P_RETURN_LAUNCH_STATUS();
}

static int launch_wgrad(const WgradParams &q, float *dw, hipStream_t st, unsigned grid, size_t bytes) {
    const int nti = q.cin_p / 32, nto = q.cout_p / 32;
    const bool fixc = (WG_THREADS % (q.cout >> 2)) == 0 && (WG_THREADS % (q.cin >> 2)) == 0;
#define WG_CASE(I, O) if (nti == I && nto == O) return fixc ? launch_wgrad_t<I, O, true>(q, dw, st, grid, bytes) : launch_wgrad_t<I, O, false>(q, dw, st, grid, bytes)
    WG_CASE(1, 1); WG_CASE(1, 2); WG_CASE(1, 4); WG_CASE(2, 1); WG_CASE(2, 2); WG_CASE(2, 4);
    WG_CASE(3, 1); WG_CASE(3, 2); WG_CASE(3, 4); WG_CASE(4, 1); WG_CASE(4, 2); WG_CASE(4, 4);
    WG_CASE(5, 1); WG_CASE(5, 2); WG_CASE(5, 4);
#undef WG_CASE
    return I2P_ERR_BAD_ARG;
}

struct PairBwd { const float *f, *g; float *d_f, *d_g, *d_bn, *d_bk; int N, M; };
struct TwoBwd { int split_c; const float *xb, *in_coef_b, *in_mi_b; float slope_b; float *gz_in_b; double *in_dsums_b; const float *e_add; };

// which family of kernels a backward call lands on: 0 = first generation (one block-synchronous dgrad+wgrad kernel),
// 1 = second / third generation (separate dgrad and wgrad launches), 2 = K-tiled wide layers (csrc/mlp_big.hip)
static int lin_bwd_family(long long rows, int cin, int cout, bool need_gx, bool pair, bool two) {
    auto pow2w = [](int c) { return c == 16 || c == 32 || c == 64 || c == 128; };
    const int cin_p = (cin + 31) & ~31, cout_p = (cout + 31) & ~31;
    const bool dgrad_ok = !need_gx || (pow2w(cin) && pow2w(cout));
    const size_t wg_lds = (2 * (size_t)WG_R * (cout_p + cin_p) + 6 * (size_t)cout + 3 * (size_t)cin) * sizeof(float);
    if (!pair && dgrad_ok && pow2w(cout) && wg_lds <= 160 * 1024 && cin <= 160) return 1;
    if (!pair && !two && i2p_big_layer_ok(rows, cin, cout)) return 2;
    return 0;
}

// `part`: 3 = the whole backward; 1 = input gradient only (dgrad: gz_in, in_dsums; no dw / dw_partial — `coef_scratch` [8*cout]
// floats takes the BN-backward constants of the wide-layer kernels); 2 = weight gradient only (dw, the BN gradients behind the
// partials).  The two halves read the same operands and write disjoint outputs.
static int lin_bwd_impl(long long rows, int cin, int cout, const float *gz, const float *y,
                        const float *out_coef, const float *out_mi, const double *out_dsums,
                        const float *x, const float *in_coef, const float *in_mi, float slope_in,
                        const float *w, float *gz_in, double *in_dsums, float *dw_partial, float *dw,
                        const PairBwd *pair, void *stream, const TwoBwd *two = nullptr, float slope_out = 1.f, int part = 3,
                        float *coef_scratch = nullptr) {
    if (rows <= 0 || cin <= 0 || cout <= 0 || (cin & 3) || (cout & 3)) return I2P_ERR_BAD_ARG;
    if (!gz || (!x && !pair) || !w || part < 1 || part > 3) return I2P_ERR_BAD_ARG;
    if ((part & 2) && (!dw_partial || !dw)) return I2P_ERR_BAD_ARG;
    if (part == 1 && (!gz_in || (out_coef && !coef_scratch))) return I2P_ERR_BAD_ARG;
    if (part == 2) { gz_in = nullptr; in_dsums = nullptr; }
    if (part != 3 && lin_bwd_family(rows, cin, cout, gz_in != nullptr || part == 2, pair != nullptr, two != nullptr) == 0) return I2P_ERR_BAD_ARG;
    if (out_coef && (!y || !out_mi || !out_dsums)) return I2P_ERR_BAD_ARG;
    if (in_coef && !in_mi) return I2P_ERR_BAD_ARG;
    LinBwdParams p;
    p.rows = rows; p.cin = cin; p.cout = cout;
    p.cin_p = (cin + 31) & ~31; p.cout_p = (cout + 31) & ~31;
    p.ldw = p.cin_p + 1; p.ldg = p.cout_p + 1; p.ldx = p.cin_p + 1;
    p.gz = gz; p.y = y; p.out_coef = out_coef; p.out_mi = out_mi; p.out_dsums = out_dsums; p.bn_out = nullptr;
    p.slope_out = out_coef ? slope_out : 1.f;
    p.x = x; p.in_coef = in_coef; p.in_mi = in_mi; p.slope_in = slope_in; p.w = w;
    p.gz_in = gz_in; p.in_dsums = in_dsums; p.dw_partial = dw_partial;
    p.pair_f = p.pair_g = nullptr; p.d_f = p.d_g = p.d_bn = p.d_bk = nullptr; p.pair_N = p.pair_M = 1;
    if (pair) {
        p.pair_f = pair->f; p.pair_g = pair->g; p.d_f = pair->d_f; p.d_g = pair->d_g; p.d_bn = pair->d_bn;
        p.d_bk = pair->d_bk; p.pair_N = pair->N; p.pair_M = pair->M;
    }
    const unsigned grid = (unsigned)i2p_lin_bwd_grid(rows);
    hipStream_t st = (hipStream_t)stream;
    {   // second-generation path: separate dgrad (lin_fwd2 DGRAD) and wgrad kernels
        auto pow2w = [](int c) { return c == 16 || c == 32 || c == 64 || c == 128; };
        // (part 2 = the wgrad half of a split call: the kernel family is the one the WHOLE call — with its input gradient — lands on)
        const bool dgrad_ok = (!gz_in && part != 2) || (pow2w(cin) && pow2w(cout));
        const size_t wg_lds = (2 * (size_t)WG_R * (p.cout_p + p.cin_p) + 6 * (size_t)cout + 3 * (size_t)cin) * sizeof(float);
        const bool gen2 = !pair && dgrad_ok && pow2w(cout) && wg_lds <= 160 * 1024 && cin <= 160;
        if (two && !(gen2 && pow2w(cin))) return I2P_ERR_BAD_ARG;
        // per-channel BN-backward constants behind the partials: [8][cout] = m1, m2, scale, mean, invstd, beta, and
        // the raw sums {sum gz, sum gz*xhat} = dbeta, dgamma of the BN behind (read back by the caller)
        float *g_coef = nullptr;
        if (out_coef && !pair) {
            g_coef = (part & 2) ? dw_partial + (size_t)grid * cout * cin : coef_scratch;   // [8][cout] scratch tail; rows 6, 7 = dbeta, dgamma for the caller
            // (every kernel family forms the BN-backward constants in its prologue and its block 0 returns dbeta / dgamma in rows 6, 7:
            //  no coefficient launch — the first-generation kernel through p.bn_out)
            p.bn_out = g_coef;
        }
        if (gen2) {
            const bool two_d3 = !two || (two->split_c * 2 == cin && two->in_coef_b && two->in_mi_b && two->e_add && two->gz_in_b &&
                                         two->in_dsums_b && in_dsums);
            // (Measured and not adopted, round 4: the backward in n row chunks — dgrad(chunk), wgrad(chunk) — so that the weight gradient
            // re-reads gz / y / x out of the 256 MB Infinity Cache instead of HBM: the cv1 node's backward went 2919 -> 3069 (n = 4) -> 3358
            // (n = 8) -> 3981 us (n = 16); with default-policy instead of non-temporal loads 2879 -> 3027 -> 3332.  The second read is
            // not what bounds these kernels, the extra launches and per-launch prologues cost more.)
            if (part == 3 && !two && gz_in && in_dsums && out_coef && p.slope_out == 1.f && in_coef && slope_in >= 0.f && slope_in <= 1.f && grid == 256 &&
                i2p_wreg_bwd_fused_ok(rows, cout, cin)) {
                // HBM-bound wide layer (64 output channels) on many rows: both gradients from one read of gz / y / x (csrc/mlp_wreg_fused.hip)
                const int rc = i2p_wreg_bwd_fused(rows, cout, cin, gz, y, out_dsums, out_coef, out_mi, rows, w, gz_in, x, in_coef, in_mi, slope_in,
                                                  in_dsums, g_coef, dw_partial, grid, stream);
                if (rc) return rc;
                launch_reduce_partials((int)grid, cout * cin, dw_partial, dw, st);
                I2P_RETURN_LAUNCH_STATUS();
            }
            if (part == 3 && two && two_d3 && gz_in && in_dsums && out_coef && p.slope_out == 1.f && in_coef && slope_in >= 0.f && slope_in <= 1.f &&
                two->slope_b >= 0.f && two->slope_b <= 1.f && grid == 256 && i2p_wreg_bwd_fused2_ok(rows, cout, cin, two->split_c)) {
                // the two-source layer 64 + 64 -> 128 on many rows: both gradients from one read of gz / y / xa / xb, a wave per
                // (strip, source) (csrc/mlp_wreg_fused.hip, TWO instantiation)
                const int rc = i2p_wreg_bwd_fused2(rows, gz, y, out_dsums, out_coef, out_mi, rows, w, gz_in, x, in_coef, in_mi, slope_in, in_dsums,
                                                   two->gz_in_b, two->xb, two->in_coef_b, two->in_mi_b, two->slope_b, two->in_dsums_b, two->e_add,
                                                   g_coef, dw_partial, grid, stream);
                if (rc) return rc;
                launch_reduce_partials((int)grid, cout * cin, dw_partial, dw, st);
                I2P_RETURN_LAUNCH_STATUS();
            }
            if (gz_in && two_d3 && out_coef && p.slope_out == 1.f && in_coef && i2p_wreg_dgrad_ok(rows, cout, cin)) {
                // wide layer on many rows, plain BN on both sides: weights stationary in registers (csrc/mlp_wreg.hip)
                const int rc = two ? i2p_wreg_dgrad(rows, cout, cin, gz, y, out_dsums, out_coef, out_mi, rows, w, gz_in, x, in_coef, in_mi,
                                                    slope_in, in_dsums, stream, two->gz_in_b, two->xb, two->in_coef_b, two->in_mi_b,
                                                    two->slope_b, two->e_add, two->in_dsums_b)
                                   : i2p_wreg_dgrad(rows, cout, cin, gz, y, out_dsums, out_coef, out_mi, rows, w, gz_in, x, in_coef, in_mi,
                                                    slope_in, in_dsums, stream);
                if (rc) return rc;
            } else if (gz_in) {
                LinFwdParams q;
                q.rows = rows; q.cin = cout; q.cout = cin; q.cin_p = cout; q.cout_p = cin;
                q.ldk = (cout > cin ? cout : cin) + ((cout & 15) == 0 ? 4 : 2);      // +4: 16-byte rows for the wide-K fragments
                q.x = gz; q.x2 = y; q.g_coef = g_coef; q.g_slope = p.slope_out; q.in_coef = nullptr; q.slope_in = 1.f;
                q.g_dsums = out_dsums; q.g_oc = out_coef; q.g_omi = out_mi; q.g_rows = rows;
                q.w = w; q.w_transposed = 1; q.y = gz_in; q.sums = in_coef ? in_dsums : nullptr;
                q.y_ld = cin; q.ch_off = 0; q.cout_total = cin; q.ablate = 0;
                q.pair_f = q.bias_n = q.bias_k = nullptr; q.pair_N = q.pair_M = 1;
                q.ex = in_coef ? x : nullptr; q.e_coef = in_coef; q.e_mi = in_mi; q.e_slope = slope_in;
                q.split_c = 0; q.xb = nullptr; q.in_coef_b = nullptr; q.slope_b = 1.f; q.yb = nullptr; q.exb = q.e_coef_b = q.e_mi_b = q.e_add = nullptr; q.sums_b = nullptr; q.e_slope_b = 1.f;
                q.fin_counter = nullptr; q.fin_gamma = q.fin_beta = nullptr; q.fin_eps = 0.f; q.fin_coef = q.fin_mi = nullptr;
                if (two) { q.split_c = two->split_c; q.yb = two->gz_in_b; q.exb = two->xb; q.e_coef_b = two->in_coef_b; q.e_mi_b = two->in_mi_b; q.e_add = two->e_add; q.sums_b = two->in_dsums_b; q.e_slope_b = two->slope_b; q.ex = x; }
                const int rc = dispatch_fwd2<false, true>(q, st);
                if (rc) return rc;
            }
            if (!(part & 2)) I2P_RETURN_LAUNCH_STATUS();            // dgrad half only
            if (!two && out_coef && grid == 256 && i2p_small_wgrad_ok(rows, cin, cout)) {
                // narrow layer on many rows (level 1): HBM streaming, dword columns as MFMA operands (csrc/mlp_wreg.hip)
                const int rc = i2p_small_wgrad(rows, cin, cout, gz, y, out_dsums, out_coef, out_mi, rows, p.slope_out, g_coef, x, in_coef,
                                               slope_in, dw_partial, grid, stream);
                if (rc) return rc;
                const int n = cout * cin;
                launch_reduce_partials((int)grid, n, dw_partial, dw, st);
                I2P_RETURN_LAUNCH_STATUS();
            }
            const bool two_ok = !two || (in_coef && two->in_coef_b && two->split_c * 2 == cin && cin == 128 && two->slope_b >= 0.f && two->slope_b <= 1.f);
            if (two_ok && out_coef && p.slope_out == 1.f && slope_in >= 0.f && slope_in <= 1.f && grid == 256 &&
                (in_coef || !(cin == 128 && cout == 128)) && i2p_wreg_wgrad_ok(rows, cin, cout)) {
                // wide layer on many rows: the [cout][cin] accumulators stationary in registers (csrc/mlp_wreg.hip)
                const int rc = i2p_wreg_wgrad(rows, cin, cout, gz, y, out_dsums, out_coef, out_mi, rows, g_coef, x, in_coef, slope_in,
                                              two ? two->xb : nullptr, two ? two->in_coef_b : nullptr, two ? two->slope_b : 1.f,
                                              two ? two->split_c : 0, dw_partial, grid, stream);
                if (rc) return rc;
                const int n = cout * cin;
                launch_reduce_partials((int)grid, n, dw_partial, dw, st);
                I2P_RETURN_LAUNCH_STATUS();
            }
            WgradParams wq;
            wq.rows = rows; wq.cin = cin; wq.cout = cout; wq.cin_p = p.cin_p; wq.cout_p = p.cout_p;
            wq.ldg = p.cout_p; wq.ldx = p.cin_p;        // 16-B aligned rows; fragments are read along channels
            wq.gz = gz; wq.y = y; wq.g_coef = g_coef; wq.g_slope = p.slope_out; wq.x = x; wq.in_coef = in_coef; wq.slope_in = slope_in;
            wq.g_dsums = out_dsums; wq.g_oc = out_coef; wq.g_omi = out_mi; wq.g_rows = rows; wq.bn_out = g_coef;
            wq.dw_partial = dw_partial;
            wq.split_c = 0; wq.xb = nullptr; wq.in_coef_b = nullptr; wq.slope_b = 1.f;
            if (two) { wq.split_c = two->split_c; wq.xb = two->xb; wq.in_coef_b = two->in_coef_b; wq.slope_b = two->slope_b; }
            return launch_wgrad(wq, dw, st, grid, wg_lds);
        }
    }
    if (!pair && !two && i2p_big_layer_ok(rows, cin, cout))        // wide layer on few rows: K-tiled dgrad + row-split wgrad (csrc/mlp_big.hip)
        return i2p_big_bwd(rows, cin, cout, gz, y, out_coef ? ((part & 2) ? dw_partial + (size_t)grid * cout * cin : coef_scratch) : nullptr,
                           out_dsums, out_coef, out_mi, p.slope_out, x, in_coef, in_mi, slope_in, w, gz_in, in_dsums, (part & 2) ? dw_partial : nullptr, (int)grid,
                           (part & 2) ? dw : nullptr, stream);
    if (part != 3) return I2P_ERR_BAD_ARG;
    switch (p.cin_p / 32) {
        case 1: return dispatch_bwd_o<1>(p, dw, st, grid);
        case 2: return dispatch_bwd_o<2>(p, dw, st, grid);
        case 3: return dispatch_bwd_o<3>(p, dw, st, grid);
        case 4: return dispatch_bwd_o<4>(p, dw, st, grid);
        case 5: return dispatch_bwd_o<5>(p, dw, st, grid);
        default: return I2P_ERR_BAD_ARG;
    }
}

extern "C" int i2p_lin_bwd(long long rows, int cin, int cout, const float *gz, const float *y,
                           const float *out_coef, const float *out_mi, const double *out_dsums,
                           const float *x, const float *in_coef, const float *in_mi, float slope_in,
                           const float *w, float *gz_in, double *in_dsums, float *dw_partial, float *dw,
                           float slope_out, void *stream) {
    return lin_bwd_impl(rows, cin, cout, gz, y, out_coef, out_mi, out_dsums, x, in_coef, in_mi, slope_in, w, gz_in,
                        in_dsums, dw_partial, dw, nullptr, stream, nullptr, slope_out);
}

extern "C" int i2p_pair_lin_bwd_grid(int B, int N, int M) {
    const int KT = (M + BWD_R - 1) / BWD_R;
    int NC = 256 / (B * KT > 0 ? B * KT : 1);
    NC = NC < 1 ? 1 : (NC > N ? N : NC);
    return B * KT * NC;
}

template <int NTI, int NTO>
static int launch_pair_bwd(PairBwdParams &p, float *dw, hipStream_t st, unsigned grid) {
    const size_t bytes = ((size_t)p.cout_p * p.ldw + (size_t)BWD_R * p.ldg + (size_t)BWD_R * p.ldx + 5 * (size_t)p.cout_p +
                          (size_t)THREADS * 4) * sizeof(float);       // + partial column sums [THREADS/co4][cout]
    if (bytes > 160 * 1024) return I2P_ERR_BAD_ARG;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pair_bwd_kernel<NTI, NTO>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((pair_bwd_kernel<NTI, NTO>), dim3(grid), dim3(THREADS), bytes, st, p);
    const int n = p.cout * p.cin;
    launch_reduce_partials((int)grid, n, p.dw_partial, dw, st);
    const int KT = (p.M + BWD_R - 1) / BWD_R;
    auto red = [&](int nslab, long long cnt, const float *slabs, float *out) {
        long long blocks = (cnt + 255) / 256; if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, nslab, cnt, slabs, out);
    };
    red(2 * KT, (long long)p.B * p.N * p.cin, p.s_df, p.d_f);
    red(KT, (long long)p.B * p.N * p.cout, p.s_dbn, p.d_bn);
    red(p.NC, (long long)p.B * p.M * p.cin, p.s_dg, p.d_g);
    red(p.NC, (long long)p.B * p.M * p.cout, p.s_dbk, p.d_bk);
    I2P_RETURN_LAUNCH_STATUS();
}

// floats the caller provides in dw_partial: per-block weight-gradient partials + the slabs of the four pair sums
extern "C" long long i2p_pair_lin_bwd_scratch(int B, int N, int M, int cin, int cout) {
    const long long KT = (M + BWD_R - 1) / BWD_R, grid = i2p_pair_lin_bwd_grid(B, N, M), NC = grid / (B * KT);
    const long long gen2 = grid * cout * cin + 2 * KT * B * N * cin + KT * B * N * cout + NC * B * M * cin + NC * B * M * cout;
    const long long gen3 = i2p_wreg_pair_bwd_ok(B, N, M, cin, cout) ? i2p_wreg_pair_bwd_scratch(B, N, M, cin, cout) : 0;
    return gen3 > gen2 ? gen3 : gen2;
}

extern "C" int i2p_pair_lin_bwd(int B, int N, int M, int cin, int cout, const float *gz, const float *y,
                                const float *out_coef, const float *out_mi, const double *out_dsums,
                                const float *f, const float *g, const float *w, float *d_f, float *d_g,
                                float *d_bias_n, float *d_bias_k, float *dw_partial, float *dw, void *stream) {
    if (B <= 0 || N <= 0 || M <= 0 || (cin & 3) || (cout & 3) || cin > 128 || cout > 128) return I2P_ERR_BAD_ARG;
    if (!gz || !f || !g || !w || !d_f || !d_g || !d_bias_n || !d_bias_k || !dw_partial || !dw) return I2P_ERR_BAD_ARG;
    if (out_coef && (!y || !out_mi || !out_dsums)) return I2P_ERR_BAD_ARG;
    if (out_coef && i2p_wreg_pair_bwd_ok(B, N, M, cin, cout)) {
        // 128 x 128 on many rows: two kernels with the weights / the accumulators stationary in registers (csrc/mlp_wreg.hip)
        int KT3 = 0, NCH3 = 0;
        const int rc = i2p_wreg_pair_bwd(B, N, M, cin, cout, gz, y, out_dsums, out_coef, out_mi, f, g, w, dw_partial, &KT3, &NCH3, stream);
        if (rc) return rc;
        hipStream_t st3 = (hipStream_t)stream;
        const int n = cout * cin;
        launch_reduce_partials(256, n, dw_partial, dw, st3);
        auto red3 = [&](int nslab, long long cnt, const float *slabs, float *out) {
            long long blocks = (cnt + 255) / 256; if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st3, nslab, cnt, slabs, out);
        };
        const float *s_df = dw_partial + (size_t)256 * n;
        const float *s_dbn = s_df + (size_t)KT3 * B * N * cin, *s_dg = s_dbn + (size_t)KT3 * B * N * cout;
        const float *s_dbk = s_dg + (size_t)NCH3 * B * M * cin;
        red3(KT3, (long long)B * N * cin, s_df, d_f);
        red3(KT3, (long long)B * N * cout, s_dbn, d_bias_n);
        red3(NCH3, (long long)B * M * cin, s_dg, d_g);
        red3(NCH3, (long long)B * M * cout, s_dbk, d_bias_k);
        I2P_RETURN_LAUNCH_STATUS();
    }
    PairBwdParams p;
    p.B = B; p.N = N; p.M = M;
    const int KT = (M + BWD_R - 1) / BWD_R;
    const unsigned grid = (unsigned)i2p_pair_lin_bwd_grid(B, N, M);
    p.NC = (int)grid / (B * KT); p.NL = (N + p.NC - 1) / p.NC;
    p.cin = cin; p.cout = cout; p.cin_p = (cin + 31) & ~31; p.cout_p = (cout + 31) & ~31;
    p.ldw = p.cin_p + 1; p.ldg = p.cout_p + 1; p.ldx = p.cin_p + 1;
    p.rows = (long long)B * N * M;
    p.gz = gz; p.y = y; p.out_coef = out_coef; p.out_mi = out_mi; p.out_dsums = out_dsums;
    p.slope_out = 1.f;
    p.f = f; p.g = g; p.w = w; p.d_f = d_f; p.d_g = d_g; p.d_bn = d_bias_n; p.d_bk = d_bias_k; p.dw_partial = dw_partial;
    p.s_df = dw_partial + (size_t)grid * cout * cin;
    p.s_dbn = p.s_df + (size_t)2 * KT * B * N * cin;
    p.s_dg = p.s_dbn + (size_t)KT * B * N * cout;
    p.s_dbk = p.s_dg + (size_t)p.NC * B * M * cin;
    hipStream_t st = (hipStream_t)stream;
    const int nti = p.cin_p / 32, nto = p.cout_p / 32;
    if (nti == 4 && nto == 4) return launch_pair_bwd<4, 4>(p, dw, st, grid);
    if (nti == 2 && nto == 2) return launch_pair_bwd<2, 2>(p, dw, st, grid);
    if (nti == 2 && nto == 1) return launch_pair_bwd<2, 1>(p, dw, st, grid);
    if (nti == 4 && nto == 2) return launch_pair_bwd<4, 2>(p, dw, st, grid);
    if (nti == 1 && nto == 1) return launch_pair_bwd<1, 1>(p, dw, st, grid);
    return I2P_ERR_BAD_ARG;
}

extern "C" int i2p_lin_bwd_2src(long long rows, int cin_a, int cin_b, int cout, const float *gz, const float *y,
                                const float *out_coef, const float *out_mi, const double *out_dsums,
                                const float *xa, const float *coef_a, const float *mi_a, float slope_a,
                                const float *xb, const float *coef_b, const float *mi_b, float slope_b,
                                const float *e_add_b, const float *w, float *gz_a, double *dsums_a, float *gz_b,
                                double *dsums_b, float *dw_partial, float *dw, void *stream) {
    if (!xa || !xb || !coef_a || !coef_b || !gz_a || !gz_b || !dsums_a || !dsums_b) return I2P_ERR_BAD_ARG;
    TwoBwd t; t.split_c = cin_a; t.xb = xb; t.in_coef_b = coef_b; t.in_mi_b = mi_b; t.slope_b = slope_b;
    t.gz_in_b = gz_b; t.in_dsums_b = dsums_b; t.e_add = e_add_b;
    return lin_bwd_impl(rows, cin_a + cin_b, cout, gz, y, out_coef, out_mi, out_dsums, xa, coef_a, mi_a, slope_a, w, gz_a,
                        dsums_a, dw_partial, dw, nullptr, stream, &t);
}
