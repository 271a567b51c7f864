// Third-generation fp32 layer forward for the wide (64 / 128 channel) layers on many rows: WEIGHTS STATIONARY IN
// REGISTERS.  (reference op: PPBackbone_center.py:10-51, 1x1 conv + batch-stat BN + activation, channel-last here.)
//
// The second generation (mlp.hip, lin_fwd2_kernel: W in LDS, two waves per SIMD with wave-private strips, x and y moved
// through LDS to change layouts) stops at ~300 us on the 853632 x 128 x 128 layer: the two waves of a SIMD fall into
// step (both in their MFMA phase, then both in their memory phases).  A variant of THIS kernel with two waves per SIMD
// and half of W each measured the same 340-356 us.  Here
//   * ONE wave per SIMD with the 512-register budget; it keeps the whole [COUT][CIN] weight matrix as MFMA A-operand
//     fragments in registers for the life of the kernel (256 registers at 128x128): no operand traffic but x,
//   * transposed formulation D[cout][row] = W . x^T with the K axis permuted (k-step 4f+e of k-slot q = channel
//     16f + 4q + e): the B operand of lane (row r, slot q) for four k-steps is one aligned float4 of row r, loaded
//     straight from global memory into the registers the MFMAs read (a load instruction covers 64-byte runs of 16 rows;
//     the two halves of a 128-byte line are requested back to back) — no LDS staging,
//   * D leaves each lane with 4 consecutive output channels of one row per 16-channel tile: float4 stores straight from
//     the accumulators,
//   * everything that is not an MFMA — the statistics and stores of the PREVIOUS strip (second accumulator set), the
//     requests of the strip after the next one (into the x registers whose last MFMA has issued), BN + activation of the
//     next strip's input — is cut into slots of a few instructions placed between the MFMAs of the current strip,
//   * per-lane BN statistics live in lane-private LDS rows (read-update-write; LDS float atomics measured 3x the
//     kernel, 64 more registers do not exist), summed across lanes / replicas in fp64 at the end.
// Measured (tools/check_wreg.py, clocks warm): 128->128 262 us (second generation 308), 128->64 154 (187), 64->128 162
// (200), 64->64 98 (130); the MFMA stream alone (back-to-back 16x16x4 from one wave) takes 228 us = 78 % of the nominal
// fp32 rate, and the chip sits at its 1400 W cap at ~2.28 GHz while this kernel runs.
#include "common.h"
#include <type_traits>

// build-time diagnostic: -DWREG_ABL=1 no statistics, 2 no stores, 4 no loads
#ifndef WREG_ABL
#define WREG_ABL 0
#endif

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
constexpr int WR_THREADS = 256;
constexpr int WR_ROWS = 16;
constexpr int REP = I2P_BN_REPLICAS;

struct WregP {
    long long rows;              // multiple of 16
    const float *x; int x_ld;
    const float *in_coef;        // [3][CIN] mean, scale, beta or nullptr
    float slope;
    const float *xb, *in_coef_b; float slope_b;      // second source (two-source instantiation: x_ld = CIN/2 for both)
    const float *w;              // [COUT][CIN]
    float *y; int y_ld;
    double *sums;                // [REP][2*COUT] or nullptr
    unsigned *fin_counter; const float *fin_gamma, *fin_beta; float fin_eps; float *fin_coef, *fin_mi;
};

__device__ __forceinline__ f32x4 ldx(const float *ptr) {
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt *>(ptr));
}
__device__ __forceinline__ void sty(float *ptr, const f32x4 &v) {
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4_nt *>(ptr));
}

// TWO: the input is two tensors of CIN/2 channels each (x, xb; own BN constants and slopes) — the layer after a
// concatenation without the concatenation (mlp.hip i2p_lin_fwd_2src is the general version).
template <int CIN, int COUT, bool BN_IN, bool TWO>
__global__ __launch_bounds__(WR_THREADS, 1) void wreg_fwd_kernel(WregP p) {
    constexpr int NT = COUT / 16;          // output tiles of 16 channels
    constexpr int L = CIN / 4;             // MFMA k-steps = channels per k-slot
    constexpr int NF = L / 4;              // float4 per lane and row
    constexpr int NTHREADS = WR_THREADS;
    __shared__ float tab[2 * CIN];         // input BN as z = fma(x, a, b): a = scale, b = beta - mean * scale
    __shared__ int fin_flag;
    // per-lane statistics accumulators live in LDS (lane-private 16-byte rows, read-update-write between the MFMAs: 64
    // registers otherwise; LDS float atomics measured 3x the whole kernel)
    __shared__ f32x4 st_lds[2 * NT][NTHREADS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave-uniform on purpose: strip counters and steps live in SGPRs
    const int m = lane & 15, q = lane >> 4;
    if (BN_IN) {
        constexpr int CS = TWO ? CIN / 2 : CIN;                 // channels per source
        for (int c = tid; c < CIN; c += NTHREADS) {
            const float *cf = (TWO && c >= CS) ? p.in_coef_b : p.in_coef;
            const int cc = (TWO && c >= CS) ? c - CS : c;
            const float a = cf[CS + cc];
            tab[c] = a; tab[CIN + c] = cf[2 * CS + cc] - cf[cc] * a;
        }
        __syncthreads();
    }

    // weights: wr[j][f][e] = W[16j + m][16f + 4q + e]  (A operand: row = lane & 15, k-slot = lane >> 4; k-step 4f+e of
    // slot q is channel 16f + 4q + e: the same permutation on both operands leaves the product unchanged)
    f32x4 wr[NT][NF];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int f = 0; f < NF; ++f) wr[j][f] = *reinterpret_cast<const f32x4 *>(p.w + (size_t)(16 * j + m) * CIN + 16 * f + 4 * q);

#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) st_lds[i][tid] = f32x4{0.f, 0.f, 0.f, 0.f};

    const long long nstrips = p.rows / WR_ROWS;
    const long long stride = (long long)gridDim.x * 4;
    const long long first = (long long)blockIdx.x * 4 + wave;
    const int n_mine = first < nstrips ? (int)((nstrips - first + stride - 1) / stride) : 0;
    if (n_mine > 0) {
        const size_t x_step = (size_t)stride * WR_ROWS * p.x_ld, y_step = (size_t)stride * WR_ROWS * p.y_ld;
        const float *xsrc = p.x + ((size_t)first * WR_ROWS + m) * p.x_ld + 4 * q;      // strip being LOADED
        const float *xsrc2 = TWO ? p.xb + ((size_t)first * WR_ROWS + m) * p.x_ld + 4 * q : nullptr;
        auto xptr = [&](int f) -> const float * { return (TWO && f >= NF / 2) ? xsrc2 + 16 * (f - NF / 2) : xsrc + 16 * f; };
        float *ydst = p.y + ((size_t)first * WR_ROWS + m) * p.y_ld + 4 * q;            // strip being STORED
        int loaded = 0;                                          // strips of this wave requested so far - 1

        const float *tq = tab + 4 * q;
        f32x4 ca[2], cb[2], zs;                                  // constants (two groups in flight) / scaled values
        ca[0] = ca[1] = cb[0] = cb[1] = zs = f32x4{0.f, 0.f, 0.f, 0.f};
        auto tf_consts = [&](int f) {
            ca[f & 1] = *reinterpret_cast<const f32x4 *>(tq + 16 * f);
            cb[f & 1] = *reinterpret_cast<const f32x4 *>(tq + CIN + 16 * f);
        };
        // act(z) = max(z, slope * z) for 0 <= slope <= 1 (launcher): three short VALU groups, one per slot
        auto tf_part = [&](f32x4 &v, int f, int part) {
            if (part == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(v[e], ca[f & 1][e], cb[f & 1][e]);
            } else if (part == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) zs[e] = v[e] * ((TWO && f >= NF / 2) ? p.slope_b : p.slope);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaxf(v[e], zs[e]);
            }
        };
        // statistics rows in flight (LDS read -> update -> write); an LDS read is used LAT slots (~256 cycles) after its
        // issue, so no s_waitcnt lgkmcnt sits between two MFMAs
        f32x4 r1[2], r2[2];
        r1[0] = r1[1] = r2[0] = r2[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int LAT = NT * L >= 128 ? 8 : 4;

        auto final_epilogue = [&](f32x4 (&prev)[NT]) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                f32x4 a = st_lds[j][tid], b = st_lds[NT + j][tid];
                a += prev[j];
#pragma unroll
                for (int c = 0; c < 4; ++c) b[c] = __builtin_fmaf(prev[j][c], prev[j][c], b[c]);
                st_lds[j][tid] = a; st_lds[NT + j][tid] = b;
                sty(ydst + 16 * j, prev[j]);
            }
        };

        // One strip = NT*L MFMAs issued back to back by the only wave of this SIMD; everything else the wave has to do
        // is cut into slots of a few instructions and placed BETWEEN them (an 8-pass MFMA leaves 7 free issue slots):
        //   slots [0, 5 NT)          statistics (lane-private LDS rows, read-update-write) + store of the PREVIOUS strip's
        //                            accumulators (second accumulator set)
        //   slot 4 NT (f+1) - 1      the registers of input float4 f have fed their last MFMA: request the same float4 of
        //                            the strip after the next one into them (two buffers, a whole strip = 3.4 us at
        //                            128x128 between a request and its first use)
        //   slots TS(f) .. TS(f)+5   BN + activation of float4 f of the NEXT strip's input (requested one strip ago),
        //                            constants two slots ahead of their use; TS spreads the eight groups over the strip
        // sched_barrier(0) after every slot keeps the compiler from regrouping them (it sinks the loads to their first
        // use and hoists the normalisation to the loads otherwise).
        constexpr int NMF = NT * L, E_END = 5 * NT + LAT - 3, SP = (NMF - E_END - LAT - 3) / (NF - 1);
        static_assert(SP >= 3 && 2 * SP >= LAT + 3, "slot plan");      // (constants: two groups in flight)
        auto strip_block = [&](auto epi_tag, f32x4 (&xc)[NF], f32x4 (&xn)[NF], f32x4 (&acc)[NT], f32x4 (&prev)[NT]) {
            constexpr bool EPI = decltype(epi_tag)::value;
            if (loaded + 1 < n_mine) { xsrc += x_step; if (TWO) xsrc2 += x_step; ++loaded; }   // (past the end: the last strip again, never used)
#pragma unroll
            for (int i = 0; i < NMF; ++i) {
                const int t = i / NT, j = i % NT, f = t >> 2, e = t & 3;
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[j][f][e], xc[f][e], t == 0 ? zero : acc[j], 0, 0, 0);
                if (i == 4 * NT * (f + 1) - 1 && (f & 1) && !(WREG_ABL & 4)) {   // both halves of a 128-byte line together
                    xc[f - 1] = ldx(xptr(f - 1)); xc[f] = ldx(xptr(f));
                }
                if (EPI && i < E_END) {
                    const int tj = i / 5, part = i % 5;                  // tile whose rows are REQUESTED / stored in this slot
                    const int uj = (i - LAT) / 5, upart = (i - LAT) % 5; // tile whose rows are UPDATED in this slot
                    if (!(WREG_ABL & 1)) {
                        if (tj < NT && part == 0) { r1[tj & 1] = st_lds[tj][tid]; r2[tj & 1] = st_lds[NT + tj][tid]; }
                        if (i >= LAT && uj < NT && upart == 0) { r1[uj & 1] += prev[uj]; st_lds[uj][tid] = r1[uj & 1]; }
                        if (i >= LAT && uj < NT && upart == 1) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) r2[uj & 1][c] = __builtin_fmaf(prev[uj][c], prev[uj][c], r2[uj & 1][c]);
                            st_lds[NT + uj][tid] = r2[uj & 1];
                        }
                    }
                    // both 64-byte halves of a row's 128-byte line leave in one slot (PMC WRITE_SIZE: 1.06x the tensor; one
                    // tile per slot 1.31x; without the streaming hint 1.02x at the same speed, but y then evicts L2 contents)
                    if (tj < NT && (tj & 1) && part == 1 && !(WREG_ABL & 2)) { sty(ydst + 16 * (tj - 1), prev[tj - 1]); sty(ydst + 16 * tj, prev[tj]); }
                }
                if (EPI && i == E_END) ydst += y_step;
                if (BN_IN && i >= E_END) {
#pragma unroll
                    for (int g = 0; g < NF; ++g) {                   // group g: constants at its base slot, math LAT slots later
                        const int u = i - (E_END + g * SP);
                        if (u == 0) tf_consts(g);
                        if (u >= LAT && u < LAT + 3) tf_part(xn[g], g, u - LAT);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        f32x4 xa[NF], xb[NF], accA[NT], accB[NT];
#pragma unroll
        for (int f = 0; f < NF; ++f) xa[f] = ldx(xptr(f));
        if (1 < n_mine) { xsrc += x_step; if (TWO) xsrc2 += x_step; ++loaded; }
#pragma unroll
        for (int f = 0; f < NF; ++f) xb[f] = ldx(xptr(f));
        if (BN_IN) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                tf_consts(f);
                tf_part(xa[f], f, 0); tf_part(xa[f], f, 1); tf_part(xa[f], f, 2);
            }
        }
        // block k: MFMAs on X[k&1] (and, float4 by float4, the request of strip k+2 into it), normalises X[~k&1]
        strip_block(std::false_type{}, xa, xb, accA, accB);                      // k = 0
        int k = 1;
        for (; k + 1 < n_mine; k += 2) {
            strip_block(std::true_type{}, xb, xa, accB, accA);
            strip_block(std::true_type{}, xa, xb, accA, accB);
        }
        if (k < n_mine) {
            strip_block(std::true_type{}, xb, xa, accB, accA);
            final_epilogue(accB);
        } else {
            final_epilogue(accA);
        }
    }
    __syncthreads();

    if (p.sums) {
        // this lane: channels 16j + 4q + e of its rows; the 16 lanes of a row group (same q) hold the same channels
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double a = (double)st_lds[j][tid][e], b = (double)st_lds[NT + j][tid][e];
#pragma unroll
                for (int off = 8; off >= 1; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
                if (m == 0) {
                    double *rep = p.sums + (size_t)((blockIdx.x * 4 + wave) % REP) * 2 * COUT;
                    atomicAdd(rep + 16 * j + 4 * q + e, a); atomicAdd(rep + COUT + 16 * j + 4 * q + e, b);
                }
            }
    }
    if (p.fin_counter) {
        // (as mlp.hip finalize_by_last_block: acknowledged atomics, then one lane's release -> ticket -> acquire, common.h)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) fin_flag = i2p_ticket_is_last(p.fin_counter, gridDim.x) ? 1 : 0;
        __syncthreads();
        if (!fin_flag) return;
        for (int ch = tid; ch < COUT; ch += NTHREADS) {
            double sa = 0.0, qa = 0.0;
            for (int r = 0; r < REP; ++r) {
                sa += __hip_atomic_load(p.sums + (size_t)r * 2 * COUT + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                qa += __hip_atomic_load(p.sums + (size_t)r * 2 * COUT + COUT + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const double mean = sa / (double)p.rows;
            double var = qa / (double)p.rows - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            const float invstd = rsqrtf((float)var + p.fin_eps);
            p.fin_coef[ch] = (float)mean; p.fin_coef[COUT + ch] = invstd * p.fin_gamma[ch]; p.fin_coef[2 * COUT + ch] = p.fin_beta[ch];
            p.fin_mi[ch] = (float)mean; p.fin_mi[COUT + ch] = invstd;
        }
        if (tid == 0) *p.fin_counter = 0u;
    }
}

template <int CIN, int COUT>
int launch_wreg(const WregP &p, hipStream_t st) {
    const long long nstrips = p.rows / WR_ROWS;
    long long grid = (nstrips + 3) / 4;
    if (grid > 256) grid = 256;
    if (p.xb) hipLaunchKernelGGL((wreg_fwd_kernel<CIN, COUT, true, true>), dim3((unsigned)grid), dim3(WR_THREADS), 0, st, p);
    else if (p.in_coef) hipLaunchKernelGGL((wreg_fwd_kernel<CIN, COUT, true, false>), dim3((unsigned)grid), dim3(WR_THREADS), 0, st, p);
    else hipLaunchKernelGGL((wreg_fwd_kernel<CIN, COUT, false, false>), dim3((unsigned)grid), dim3(WR_THREADS), 0, st, p);
    I2P_RETURN_LAUNCH_STATUS();
}


// =====================================================================================================================
// DGRAD of the same layers on the same machinery: dL/dz_in = act_in'(z_in) .* (g^y . W), g^y = BN-backward(gz, y) of the
// layer BEHIND formed on load, activation derivative of the layer IN FRONT + its BN-backward statistics in the store
// phase (mlp.hip lin_fwd2_kernel<.., DGRAD> is the general version: two destinations, added gradients, any width).
// HBM-bound (4 tensors of rows*C or rows*K floats against 2*rows*K*C flop).  Per lane and strip: gz and y as B-operand
// float4s (ONE register set each: a float4 is re-requested for the next strip right after its last use and normalised
// just before its first MFMA, ~7/8 of a strip later), x of the layer in front as 4-channel rows in the accumulator
// layout, requested a strip ahead.  K = channels of gz / y, C = channels of the result.
//   g^y = sc * (gz - m1 - (y - mu) * is * m2) = fma(sc, gz, fma(y, Bc, Ac)),  Bc = -sc * is * m2,  Ac = -sc * m1 - Bc * mu
//   z_in = fma(x, e_sc, e_zb);  xhat_in = fma(x, e_is, e_nm)      (e_zb = beta - mean * e_sc, e_nm = -mean * e_is)
// =====================================================================================================================
struct WregDgradP {
    long long rows;              // multiple of 16
    const float *gz, *y2;        // [rows, K]
    const double *g_dsums;       // [REP][2*K] sums {gz, gz * xhat} of the BN behind
    const float *g_oc, *g_omi;   // its coef [3][K] (mean, scale, beta) and mean_invstd [2][K]
    long long g_rows;
    const float *w;              // [K][C]
    float *gz_in;                // [rows, C]
    const float *ex;             // [rows, C] pre-BN input of the layer
    const float *e_coef, *e_mi;  // [3][C], [2][C] of the BN in front
    float e_slope;
    double *sums;                // [REP][2*C] {sum g, sum g * xhat_in}
    // two destinations (TWO instantiation): result columns [0, C/2) -> gz_in [rows, C/2] with ex / e_coef / e_mi / sums
    // of C/2 channels, columns [C/2, C) -> gz_in_b with exb / e_coef_b / e_mi_b / sums_b, and e_add [rows, C/2] added to
    // the second half before its activation derivative (the gradient that reaches that tensor on another path)
    float *gz_in_b; const float *exb, *e_coef_b, *e_mi_b, *e_add; float e_slope_b; double *sums_b;
};

template <int K, int C, bool TWO>
__global__ __launch_bounds__(WR_THREADS, 1) void wreg_dgrad_kernel(WregDgradP p) {
    constexpr int CD = TWO ? C / 2 : C;    // channels (= row pitch) of a destination
    constexpr int NTD = CD / 16;           // tiles per destination
    constexpr int NT = C / 16;             // output tiles of 16 channels
    constexpr int L = K / 4;               // MFMA k-steps
    constexpr int NF = L / 4;              // float4 of gz (and of y) per lane and row
    constexpr int G = 4 * NT;              // MFMAs per input float4
    constexpr int NMF = NT * L;
    __shared__ float gt[3 * K];            // sc, Ac, Bc
    __shared__ float et[4 * C];            // e_sc, e_zb, e_is, e_nm
    __shared__ f32x4 st_lds[2 * NT][WR_THREADS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;     // (scalarising `wave` here makes clang spill 150 registers)
    const int m = lane & 15, q = lane >> 4;
    for (int ch = tid; ch < K; ch += WR_THREADS) {
        double sd = 0.0, sx = 0.0;
#pragma unroll 8
        for (int rp = 0; rp < REP; ++rp) { sd += p.g_dsums[(size_t)rp * 2 * K + ch]; sx += p.g_dsums[(size_t)rp * 2 * K + K + ch]; }
        const float m1 = (float)(sd / (double)p.g_rows), m2 = (float)(sx / (double)p.g_rows);
        const float sc = p.g_oc[K + ch], mu = p.g_omi[ch], is = p.g_omi[K + ch];
        const float bc = -sc * is * m2;
        gt[ch] = sc; gt[K + ch] = -sc * m1 - bc * mu; gt[2 * K + ch] = bc;
    }
    for (int ch = tid; ch < C; ch += WR_THREADS) {
        const bool sb = TWO && ch >= CD;
        const float *cf = sb ? p.e_coef_b : p.e_coef, *mi = sb ? p.e_mi_b : p.e_mi;
        const int cc = sb ? ch - CD : ch;
        const float mean = cf[cc], sc = cf[CD + cc], is = mi[CD + cc];
        et[ch] = sc; et[C + ch] = cf[2 * CD + cc] - mean * sc; et[2 * C + ch] = is; et[3 * C + ch] = -mean * is;
    }
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) st_lds[i][tid] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    // weights: wr[j][f][e] = W[k = 16f + 4q + e][c = 16j + m]
    f32x4 wr[NT][NF];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int e = 0; e < 4; ++e) wr[j][f][e] = p.w[(size_t)(16 * f + 4 * q + e) * C + 16 * j + m];

    const long long nstrips = p.rows / WR_ROWS;
    const long long stride = (long long)gridDim.x * 4;
    const long long first = (long long)blockIdx.x * 4 + wave;
    const int n_mine = first < nstrips ? (int)((nstrips - first + stride - 1) / stride) : 0;
    if (n_mine > 0) {
        // BYTE offsets in 32 bits (launcher: every tensor < 4 GB): uniform base pointer + 32-bit lane offset is the scalar-base
        // addressing form of global_load / global_store, one VGPR per address instead of a 64-bit pair and its arithmetic
        const unsigned k_step = (unsigned)(stride * WR_ROWS * K * 4), c_step = (unsigned)(stride * WR_ROWS * CD * 4);
        // element offset of this lane's float4 #0 in the [rows, K] tensors: the strip being REQUESTED (k+1, clamped to the last)
        unsigned koff1 = (unsigned)((((size_t)first * WR_ROWS + m) * K + 4 * q) * 4);
        int loaded = 0;
        auto advance = [&]() { if (loaded + 1 < n_mine) { koff1 += k_step; ++loaded; } };
        // [rows, C] tensors: strip being computed (its x rows are requested) and strip being stored
        unsigned coff_cur = (unsigned)((((size_t)first * WR_ROWS + m) * CD + 4 * q) * 4), coff_prev = coff_cur;
        auto at = [](const float *base, unsigned byte_off) -> const float * { return reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off); };
        auto atw = [](float *base, unsigned byte_off) -> float * { return reinterpret_cast<float *>(reinterpret_cast<char *>(base) + byte_off); };
        // destination of output tile j (compile-time j): pointers, column, slope
        auto dst_of = [&](int j) -> float * { return (TWO && j >= NTD) ? p.gz_in_b + 16 * (j - NTD) : p.gz_in + 16 * j; };
        auto ex_of = [&](int j) -> const float * { return (TWO && j >= NTD) ? p.exb + 16 * (j - NTD) : p.ex + 16 * j; };
        auto slope_of = [&](int j) -> float { return (TWO && j >= NTD) ? p.e_slope_b : p.e_slope; };

        const float *gq = gt + 4 * q, *eq = et + 4 * q;
        f32x4 tsc, tac, tbc, tu;                                 // constants of the input float4 being normalised
        tsc = tac = tbc = tu = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 esc, ezb, eis, enm, r1, r2, ez, ev, vkeep;          // store-phase constants / statistics rows / temporaries
        esc = ezb = eis = enm = r1 = r2 = ez = ev = vkeep = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int LAT = 5;                                   // slots between an LDS read and its use
        constexpr int PE = 10;                                   // store-phase slots per output tile
        static_assert(PE * NT <= NMF && LAT + 2 <= G, "slot plan");

        f32x4 x[NF], yb[NF], exr[NT], ead[TWO ? NTD : 1], accA[NT], accB[NT];
        auto epi_math = [&](const f32x4 &acc, const f32x4 &xr, float slope, f32x4 &v, f32x4 &xh) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float z = __builtin_fmaf(xr[c], esc[c], ezb[c]);
                v[c] = z > 0.f ? acc[c] : acc[c] * slope;
                xh[c] = __builtin_fmaf(xr[c], eis[c], enm[c]);
            }
        };
        auto final_epilogue = [&](f32x4 (&prev)[NT]) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                esc = *reinterpret_cast<const f32x4 *>(eq + 16 * j); ezb = *reinterpret_cast<const f32x4 *>(eq + C + 16 * j);
                eis = *reinterpret_cast<const f32x4 *>(eq + 2 * C + 16 * j); enm = *reinterpret_cast<const f32x4 *>(eq + 3 * C + 16 * j);
                f32x4 v, xh, pj = prev[j];
                if (TWO && j >= NTD) pj += ead[j - NTD];
                epi_math(pj, exr[j], slope_of(j), v, xh);
                f32x4 a = st_lds[j][tid], b = st_lds[NT + j][tid];
                a += v;
#pragma unroll
                for (int c = 0; c < 4; ++c) b[c] = __builtin_fmaf(v[c], xh[c], b[c]);
                st_lds[j][tid] = a; st_lds[NT + j][tid] = b;
                sty(atw(dst_of(j), coff_prev), v);
            }
        };
        auto tf_all = [&](int f) {                               // prologue: g^y of float4 f, not slotted
            tsc = *reinterpret_cast<const f32x4 *>(gq + 16 * f); tac = *reinterpret_cast<const f32x4 *>(gq + K + 16 * f);
            tbc = *reinterpret_cast<const f32x4 *>(gq + 2 * K + 16 * f);
#pragma unroll
            for (int c = 0; c < 4; ++c) x[f][c] = __builtin_fmaf(tsc[c], x[f][c], __builtin_fmaf(yb[f][c], tbc[c], tac[c]));
        };

        // One strip = NMF MFMAs back to back; between them, in slots of a few instructions:
        //   store phase of the PREVIOUS strip, tile j in slots [PE j, PE j + 9]: constants + statistics rows from LDS, LAT slots
        //     later z / act' / xhat, the store (two tiles = one 128-byte line per row together), the statistics update, and
        //     the request of the CURRENT strip's x rows of that tile for the next store phase
        //   input float4 f: g^y formed during the MFMA group in front of its own (f = 0: during the last group, for the
        //     next strip); gz / y registers are re-requested for the next strip in pairs (one 128-byte line per row: PMC
        //     FETCH_SIZE 1.28x the tensors with single float4 requests) once both have been consumed
        auto strip_block = [&](auto epi_tag, f32x4 (&acc)[NT], f32x4 (&prev)[NT]) {
            constexpr bool EPI = decltype(epi_tag)::value;
#pragma unroll
            for (int i = 0; i < NMF; ++i) {
                const int t = i / NT, j = i % NT, f = t >> 2, e = t & 3;
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[j][f][e], x[f][e], t == 0 ? zero : acc[j], 0, 0, 0);
                // ---- input side -----------------------------------------------------------------------------------
                if (i == G * (f + 1) - 1 && (f & 1)) {                                // gz of the next strip, one 128-byte line per row
                    x[f - 1] = ldx(at(p.gz + 16 * (f - 1), koff1)); x[f] = ldx(at(p.gz + 16 * f, koff1));
                }
                {
                    const int fn = (f + 1) % NF, u = i - G * f;                       // float4 normalised during this group
                    if (u == 0) {
                        tsc = *reinterpret_cast<const f32x4 *>(gq + 16 * fn); tac = *reinterpret_cast<const f32x4 *>(gq + K + 16 * fn);
                        tbc = *reinterpret_cast<const f32x4 *>(gq + 2 * K + 16 * fn);
                    }
                    if (u == LAT) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) tu[c] = __builtin_fmaf(yb[fn][c], tbc[c], tac[c]);
                    }
                    if (u == LAT + 1) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) x[fn][c] = __builtin_fmaf(tsc[c], x[fn][c], tu[c]);
                        // y of the next strip, one 128-byte line per row at a time: float4 0 (consumed in the last group of the
                        // previous strip) waits for float4 1 (consumed in this strip's first group)
                        if (fn & 1) { yb[fn - 1] = ldx(at(p.y2 + 16 * (fn - 1), koff1)); yb[fn] = ldx(at(p.y2 + 16 * fn, koff1)); }
                    }
                }
                // ---- store phase of the previous strip --------------------------------------------------------------
                if (EPI && i < PE * NT) {
                    const int tj = i / PE, part = i % PE;
                    if (part == 0) {
                        esc = *reinterpret_cast<const f32x4 *>(eq + 16 * tj); ezb = *reinterpret_cast<const f32x4 *>(eq + C + 16 * tj);
                        r1 = st_lds[tj][tid]; r2 = st_lds[NT + tj][tid];
                    }
                    if (part == 1) { eis = *reinterpret_cast<const f32x4 *>(eq + 2 * C + 16 * tj); enm = *reinterpret_cast<const f32x4 *>(eq + 3 * C + 16 * tj); }
                    if (part == LAT) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) ez[c] = __builtin_fmaf(exr[tj][c], esc[c], ezb[c]);
                        ev = prev[tj];
                        if (TWO && tj >= NTD) ev += ead[tj - NTD];
                    }
                    if (part == LAT + 1) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) ev[c] = ez[c] > 0.f ? ev[c] : ev[c] * slope_of(tj);
                        if (tj & 1) { sty(atw(dst_of(tj - 1), coff_prev), vkeep); sty(atw(dst_of(tj), coff_prev), ev); }
                        else vkeep = ev;
                    }
                    if (part == LAT + 2) { r1 += ev; st_lds[tj][tid] = r1; }
                    if (part == LAT + 3) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) r2[c] = __builtin_fmaf(ev[c], __builtin_fmaf(exr[tj][c], eis[c], enm[c]), r2[c]);
                        st_lds[NT + tj][tid] = r2;
                        if (tj & 1) {                                             // rows of the strip being computed, for its store phase
                            exr[tj - 1] = ldx(at(ex_of(tj - 1), coff_cur)); exr[tj] = ldx(at(ex_of(tj), coff_cur));
                            if (TWO && tj >= NTD) { ead[tj - 1 - NTD] = ldx(at(p.e_add + 16 * (tj - 1 - NTD), coff_cur)); ead[tj - NTD] = ldx(at(p.e_add + 16 * (tj - NTD), coff_cur)); }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!EPI) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    exr[j] = ldx(at(ex_of(j), coff_cur));
                    if (TWO && j >= NTD) ead[j - NTD] = ldx(at(p.e_add + 16 * (j - NTD), coff_cur));
                }
            }
            coff_prev = coff_cur; coff_cur += c_step;                // (past the end: never dereferenced again)
            advance();
        };

        // prologue: strip 0 raw in x / yb, float4 0 normalised
#pragma unroll
        for (int f = 0; f < NF; ++f) { x[f] = ldx(at(p.gz + 16 * f, koff1)); yb[f] = ldx(at(p.y2 + 16 * f, koff1)); }
        advance();                                               // koff1 = strip 1 (clamped)
        tf_all(0);
        strip_block(std::false_type{}, accA, accB);              // k = 0
        int k = 1;
        for (; k + 1 < n_mine; k += 2) {
            strip_block(std::true_type{}, accB, accA);
            strip_block(std::true_type{}, accA, accB);
        }
        if (k < n_mine) {
            strip_block(std::true_type{}, accB, accA);
            final_epilogue(accB);
        } else {
            final_epilogue(accA);
        }
    }
    __syncthreads();
    if (p.sums) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double a = (double)st_lds[j][tid][e], b = (double)st_lds[NT + j][tid][e];
#pragma unroll
                for (int off = 8; off >= 1; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
                if (m == 0) {
                    double *rep = ((TWO && j >= NTD) ? p.sums_b : p.sums) + (size_t)((blockIdx.x * 4 + wave) % REP) * 2 * CD;
                    const int ch = 16 * (TWO && j >= NTD ? j - NTD : j) + 4 * q + e;
                    atomicAdd(rep + ch, a); atomicAdd(rep + CD + ch, b);
                }
            }
    }
}

template <int K, int C>
int launch_wreg_dgrad(const WregDgradP &p, hipStream_t st) {
    const long long nstrips = p.rows / WR_ROWS;
    long long grid = (nstrips + 3) / 4;
    if (grid > 256) grid = 256;
    if (p.gz_in_b) hipLaunchKernelGGL((wreg_dgrad_kernel<K, C, true>), dim3((unsigned)grid), dim3(WR_THREADS), 0, st, p);
    else hipLaunchKernelGGL((wreg_dgrad_kernel<K, C, false>), dim3((unsigned)grid), dim3(WR_THREADS), 0, st, p);
    I2P_RETURN_LAUNCH_STATUS();
}


// =====================================================================================================================
// WGRAD on the same machinery: dW[o][c] = sum_rows g^y[row][o] * a[row][c], a = act(BN(x)) of the layer in front.
// The ACCUMULATORS are what stays in registers here: the whole [CO][CI] result of a wave (256 registers at 128x128),
// rows are the contraction axis: one 16-row strip = 4 k-steps (k-slot q = rows 4q..4q+3, step t its row 4q+t) x (CO/16)(CI/16)
// tiles.  Tile jo of the A operand holds output channels {NO*i + jo} (i = lane & 15, NO = CO/16), so a lane's A values of
// a k-step are NO CONTIGUOUS channels of one row of gz / y: float4 loads straight into the operand registers, no LDS,
// no transposition; same for x (tile jc = channels {NI*n + jc}).  The per-channel constants of a lane never change.
// g^y and a are formed in place during the previous k-step's MFMAs; a k-step's registers are re-requested for the next
// strip right after their last MFMA.  The four waves of a block add their results through LDS in a fixed order; the
// block's partial goes to dw_partial[block] (reduced by the caller's reduce_partials launch, as for lin_wgrad_kernel).
// =====================================================================================================================
struct WregWgradP {
    long long rows;              // multiple of 16
    const float *gz, *y2;        // [rows, CO]
    const double *g_dsums; const float *g_oc, *g_omi; long long g_rows;
    float *bn_out;               // [8][CO]: rows 6, 7 <- dbeta, dgamma of the BN behind (block 0)
    const float *x;              // [rows, CI]
    const float *in_coef;        // [3][CI] or nullptr
    float slope_in;              // 0 <= slope <= 1
    // two-source input (nullptr: one): columns [0, split) of the layer input come from x [rows, split] with in_coef
    // [3][split], columns [split, CI) from xb [rows, CI - split] with in_coef_b; split is a multiple of CI/16
    const float *xb, *in_coef_b; float slope_b; int split;
    float *dw_partial;           // [grid][CO*CI]
};

template <int CO, int CI, bool BN_IN, bool TWO>
__global__ __launch_bounds__(WR_THREADS, 1) void wreg_wgrad_kernel(WregWgradP p) {
    constexpr int NO = CO / 16, NI = CI / 16;      // tiles = channels per lane
    constexpr int HO = NO / 4, HI = NI / 4;        // float4 per lane, row and tensor
    constexpr int SM = NO * NI;                    // MFMAs per k-step
    __shared__ float red[CO * CI];
    __shared__ float gtab[3 * CO];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave-uniform on purpose: strip counters and steps live in SGPRs
    const int n = lane & 15, q = lane >> 4;
    for (int ch = tid; ch < CO; ch += WR_THREADS) {
        double sd = 0.0, sx = 0.0;
#pragma unroll 8
        for (int rp = 0; rp < REP; ++rp) { sd += p.g_dsums[(size_t)rp * 2 * CO + ch]; sx += p.g_dsums[(size_t)rp * 2 * CO + CO + ch]; }
        const float m1 = (float)(sd / (double)p.g_rows), m2 = (float)(sx / (double)p.g_rows);
        // g^y = sc * (gz + y * Bc + Ac), Bc = -is * m2, Ac = -m1 - Bc * mu; the factor sc[o] is applied to row o of the
        // RESULT (8 registers and a multiply per element less in the loop)
        const float sc = p.g_oc[CO + ch], mu = p.g_omi[ch], is = p.g_omi[CO + ch];
        const float bc = -is * m2;
        gtab[ch] = sc; gtab[CO + ch] = -m1 - bc * mu; gtab[2 * CO + ch] = bc;
        if (blockIdx.x == 0 && p.bn_out) { p.bn_out[6 * CO + ch] = (float)sd; p.bn_out[7 * CO + ch] = (float)sx; }
    }
    __syncthreads();
    // this lane's channels: gz / y columns NO*n .. NO*n + NO-1, x columns NI*n .. NI*n + NI-1
    f32x4 cac[HO], cbc[HO], cxa[HI], cxb[HI];
#pragma unroll
    for (int h = 0; h < HO; ++h) {
        cac[h] = *reinterpret_cast<const f32x4 *>(gtab + CO + NO * n + 4 * h);
        cbc[h] = *reinterpret_cast<const f32x4 *>(gtab + 2 * CO + NO * n + 4 * h);
    }
    // source of this lane's x columns (its NI channels never straddle the split)
    // x columns of this lane.  One source: its NI channels NI*n .. NI*n+NI-1 (float4 h = channels NI*n + 4h ..).
    // Two sources of CI/2 channels (HI == 2): float4 0 = channels 4n..4n+3 of x, float4 1 = channels 4n..4n+3 of xb, i.e.
    // tile jc <-> channel (jc < 4 ? 4n + jc : CI/2 + 4n + jc - 4): every load instruction has ONE uniform base pointer
    // (scalar-base addressing, 32-bit lane offsets) and covers whole 256-byte rows.
    static_assert(!TWO || HI == 2, "two-source wgrad: CI = 128");
    constexpr int x_ld = TWO ? CI / 2 : CI;
    auto xchan = [&](int h) -> int { return TWO ? h * (CI / 2) + 4 * n : NI * n + 4 * h; };      // first channel of float4 h
#pragma unroll
    for (int h = 0; h < HI; ++h) {
        cxa[h] = f32x4{1.f, 1.f, 1.f, 1.f}; cxb[h] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (BN_IN) {
            const float *cf = (TWO && h == 1) ? p.in_coef_b : p.in_coef;
            const int cc = TWO ? 4 * n : NI * n + 4 * h;
            const f32x4 mu = *reinterpret_cast<const f32x4 *>(cf + cc);
            cxa[h] = *reinterpret_cast<const f32x4 *>(cf + x_ld + cc);
            cxb[h] = *reinterpret_cast<const f32x4 *>(cf + 2 * x_ld + cc) - mu * cxa[h];
        }
    }
    f32x4 acc[NO][NI];
#pragma unroll
    for (int jo = 0; jo < NO; ++jo)
#pragma unroll
        for (int jc = 0; jc < NI; ++jc) acc[jo][jc] = f32x4{0.f, 0.f, 0.f, 0.f};

    const long long nstrips = p.rows / WR_ROWS;
    const long long stride = (long long)gridDim.x * 4;
    const long long first = (long long)blockIdx.x * 4 + wave;
    // (scalar strip count: scalar loop branch and no spill reload inside the loop of the one-source instantiations; the
    //  two-source one is spill-free in its loop without it and not with it — clang 19 register allocation, checked in the ISA)
    const int n_mine_v = first < nstrips ? (int)((nstrips - first + stride - 1) / stride) : 0;
    const int n_mine = TWO ? n_mine_v : __builtin_amdgcn_readfirstlane(n_mine_v);
    if (n_mine > 0) {
        // element offsets of row 4q of the strip being REQUESTED: k-slot q owns rows 4q..4q+3, k-step t takes row 4q + t (any
        // bijection rows <-> (slot, step) gives the same sum; this one keeps the per-step offsets inside the load immediates)
        // (BYTE offsets in 32 bits against uniform base pointers, launcher: every tensor < 4 GB)
        unsigned goff = (unsigned)((((size_t)first * WR_ROWS + 4 * q) * CO + NO * n) * 4);
        unsigned xoff = (unsigned)((((size_t)first * WR_ROWS + 4 * q) * x_ld + (TWO ? 4 * n : NI * n)) * 4);
        const unsigned g_step = __builtin_amdgcn_readfirstlane((unsigned)(stride * WR_ROWS * CO * 4));
        const unsigned x_step = __builtin_amdgcn_readfirstlane((unsigned)(stride * WR_ROWS * x_ld * 4));
        auto at = [](const float *base, unsigned byte_off) -> const float * { return reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off); };
        int loaded = 0;
        f32x4 gv[4][HO], yv[4][HO], xv[4][HI];
        auto ld_g = [&](int t) {
#pragma unroll
            for (int h = 0; h < HO; ++h) gv[t][h] = ldx(at(p.gz + t * CO + 4 * h, goff));
        };
        auto ld_y = [&](int t) {
#pragma unroll
            for (int h = 0; h < HO; ++h) yv[t][h] = ldx(at(p.y2 + t * CO + 4 * h, goff));
        };
        auto ld_x = [&](int t) {
#pragma unroll
            for (int h = 0; h < HI; ++h) xv[t][h] = ldx(at(((TWO && h == 1) ? p.xb : p.x) + t * x_ld + (TWO ? 0 : 4 * h), xoff));
        };
        auto tf_g = [&](int t, int h) {                          // g^y / sc = gz + fma(y, Bc, Ac)
#pragma unroll
            for (int c = 0; c < 4; ++c) gv[t][h][c] = gv[t][h][c] + __builtin_fmaf(yv[t][h][c], cbc[h][c], cac[h][c]);
        };
        auto tf_x = [&](int t, int h) {                          // a = max(z, slope z), z = fma(x, a, b)
            if (!BN_IN) return;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float z = __builtin_fmaf(xv[t][h][c], cxa[h][c], cxb[h][c]);
                xv[t][h][c] = __builtin_fmaxf(z, z * ((TWO && h == 1) ? p.slope_b : p.slope_in));
            }
        };
#pragma unroll
        for (int t = 0; t < 4; ++t) { ld_g(t); ld_y(t); ld_x(t); }
        if (1 < n_mine) { goff += g_step; xoff += x_step; ++loaded; }
#pragma unroll
        for (int h = 0; h < HO; ++h) tf_g(0, h);
#pragma unroll
        for (int h = 0; h < HI; ++h) tf_x(0, h);
        ld_y(0);                                                 // y of k-step 0 of the next strip
        for (int k = 0; k < n_mine; ++k) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int tn = (t + 1) & 3;                      // k-step whose operands are formed during this one
#pragma unroll
                for (int u = 0; u < SM; ++u) {
                    const int jo = u / NI, jc = u % NI;
                    acc[jo][jc] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[t][jo >> 2][jo & 3], xv[t][jc >> 2][jc & 3], acc[jo][jc], 0, 0, 0);
                    if (u >= 1 && u < 1 + HO) tf_g(tn, u - 1);
                    if (u >= 1 + HO && u < 1 + HO + HI) tf_x(tn, u - 1 - HO);
                    // y of step tn is consumed: request it for the strip after (t = 3: tn = 0 belongs to the NEXT strip, whose
                    // successor is requested after the offsets advance below)
                    if (u == 1 + HO + HI && t < 3) ld_y(tn);
                    if (u == SM - 1) { ld_g(t); ld_x(t); }          // this step's registers: same step of the next strip
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (loaded + 1 < n_mine) { goff += g_step; xoff += x_step; ++loaded; }
            ld_y(0);
        }
    }
    // ---- the four waves add their results through LDS in a fixed order, then the block's partial leaves coalesced ----
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int jo = 0; jo < NO; ++jo)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int h = 0; h < HI; ++h) {
                        float *dst = red + (size_t)(NO * (4 * q + e) + jo) * CI + xchan(h);
                        f32x4 v = {acc[jo][4 * h][e], acc[jo][4 * h + 1][e], acc[jo][4 * h + 2][e], acc[jo][4 * h + 3][e]};
                        v *= gtab[NO * (4 * q + e) + jo];                    // sc of output row o
                        if (w > 0) v += *reinterpret_cast<const f32x4 *>(dst);
                        *reinterpret_cast<f32x4 *>(dst) = v;
                    }
        }
        __syncthreads();
    }
    float *out = p.dw_partial + (size_t)blockIdx.x * CO * CI;
    for (int i = tid; i < CO * CI / 4; i += WR_THREADS)
        *reinterpret_cast<f32x4 *>(out + 4 * i) = *reinterpret_cast<const f32x4 *>(red + 4 * i);
}

template <int CO, int CI>
int launch_wreg_wgrad(const WregWgradP &p, unsigned grid, hipStream_t st) {
    if constexpr (CI == 128) {
        if (p.xb) { hipLaunchKernelGGL((wreg_wgrad_kernel<CO, CI, true, true>), dim3(grid), dim3(WR_THREADS), 0, st, p); I2P_RETURN_LAUNCH_STATUS(); }
    }
    if (p.xb) return I2P_ERR_BAD_ARG;
    if (p.in_coef) hipLaunchKernelGGL((wreg_wgrad_kernel<CO, CI, true, false>), dim3(grid), dim3(WR_THREADS), 0, st, p);
    else {
        // 128 x 128 without an input BN is not instantiated: with the 256 accumulators it kept 13 operand registers in scratch INSIDE
        // the strip loop (profiles/r04_resource_usage_start.txt); no layer of the network has that shape (i2p_wreg_wgrad_ok's caller
        // falls back to lin_wgrad_kernel)
        if constexpr (CO == 128 && CI == 128) return I2P_ERR_BAD_ARG;
        else hipLaunchKernelGGL((wreg_wgrad_kernel<CO, CI, false, false>), dim3(grid), dim3(WR_THREADS), 0, st, p);
    }
    I2P_RETURN_LAUNCH_STATUS();
}


// =====================================================================================================================
// Backward of the PAIR layer (first cost-volume layer: y[b,n,k,:] = W (f[b,n,:] .* g[b,k,:]) + bias_n[b,n,:] + bias_k[b,k,:],
// BN behind; mlp.hip pair_bwd_kernel is the general version) as two kernels on the same machinery.  A wave owns one
// (sample b, 16-pixel tile k0.., chunk of points) and walks the points n of its chunk: strip = rows (b, n, k0..k0+15),
// contiguous in gz / y.  Rows past M in the last pixel tile are loaded from row M-1 and multiplied by 0.
//   kernel W (this one, on the wgrad machinery): dW += g^y^T . (f .* g)  — the product operand is formed from the lane's
//     FIXED pixel rows of g (lane-private LDS) and the strip's row of f —, d_bias_k[b,k,:] = sum_n g^y (lane-private LDS
//     accumulators, one slab per point chunk), d_bias_n[b,n,:] = sum_k g^y (one slab per pixel tile).
//   kernel D (below, on the dgrad machinery): dP = g^y . W; d_f[b,n,:] = sum_k dP .* g, d_g[b,k,:] = sum_n dP .* f.
// Every cross-wave sum goes through slabs the caller reduces in a fixed order (bit-reproducible gradients).
// =====================================================================================================================
struct WregPairP {
    int B, N, M, KT, NCH, NL;    // KT = ceil(M/16) pixel tiles, NCH point chunks of NL points
    const float *gz, *y2;        // [B*N*M, CO]
    const double *g_dsums; const float *g_oc, *g_omi; long long g_rows;
    const float *f, *g, *w;      // [B,N,CI], [B,M,CI], [CO][CI]
    float *dw_partial;           // [grid][CO*CI]                     (kernel W)
    float *s_dbn, *s_dbk;        // [KT][B*N*CO], [NCH][B*M*CO]       (kernel W)
    float *s_df, *s_dg;          // [KT][B*N*CI], [NCH][B*M*CI]       (kernel D)
};

template <int CO, int CI>
__global__ __launch_bounds__(WR_THREADS, 1) void wreg_pair_wgrad_kernel(WregPairP p) {
    constexpr int NO = CO / 16, NI = CI / 16, HO = NO / 4, HI = NI / 4, SM = NO * NI;
    constexpr int NPRIV = 4 * HI + 4 * HO;                      // lane-private float4 rows: g of the lane's pixels, d_bias_k sums
    constexpr int LDS_F = CO * CI > NPRIV * WR_THREADS * 4 ? CO * CI : NPRIV * WR_THREADS * 4;
    __shared__ float lds[LDS_F];
    __shared__ float gtab[3 * CO];
    f32x4 (*priv)[WR_THREADS] = reinterpret_cast<f32x4 (*)[WR_THREADS]>(lds);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave-uniform on purpose: strip counters and steps live in SGPRs
    const int n16 = lane & 15, q = lane >> 4;
    for (int ch = tid; ch < CO; ch += WR_THREADS) {
        double sd = 0.0, sx = 0.0;
#pragma unroll 8
        for (int rp = 0; rp < REP; ++rp) { sd += p.g_dsums[(size_t)rp * 2 * CO + ch]; sx += p.g_dsums[(size_t)rp * 2 * CO + CO + ch]; }
        const float m1 = (float)(sd / (double)p.g_rows), m2 = (float)(sx / (double)p.g_rows);
        const float sc = p.g_oc[CO + ch], mu = p.g_omi[ch], is = p.g_omi[CO + ch];
        const float bc = -sc * is * m2;
        gtab[ch] = sc; gtab[CO + ch] = -sc * m1 - bc * mu; gtab[2 * CO + ch] = bc;
    }
    __syncthreads();
    f32x4 csc[HO], cac[HO], cbc[HO];
#pragma unroll
    for (int h = 0; h < HO; ++h) {
        csc[h] = *reinterpret_cast<const f32x4 *>(gtab + NO * n16 + 4 * h); cac[h] = *reinterpret_cast<const f32x4 *>(gtab + CO + NO * n16 + 4 * h);
        cbc[h] = *reinterpret_cast<const f32x4 *>(gtab + 2 * CO + NO * n16 + 4 * h);
    }
    f32x4 acc[NO][NI];
#pragma unroll
    for (int jo = 0; jo < NO; ++jo)
#pragma unroll
        for (int jc = 0; jc < NI; ++jc) acc[jo][jc] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ntasks = p.B * p.KT * p.NCH;
    for (int task = blockIdx.x * 4 + wave; task < ntasks; task += gridDim.x * 4) {
        const int nc = task % p.NCH, kt = (task / p.NCH) % p.KT, b = task / (p.NCH * p.KT);
        const int k0 = kt * WR_ROWS, n_begin = nc * p.NL, n_end = n_begin + p.NL < p.N ? n_begin + p.NL : p.N;
        const int ns = n_end - n_begin;
        if (ns <= 0) continue;
        // this lane's pixel rows: k-step t -> pixel k0 + 4t + q
        int rowg[4]; float vm[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = k0 + 4 * t + q;
            vm[t] = k < p.M ? 1.f : 0.f;
            const int kc = k < p.M ? k : p.M - 1;
            rowg[t] = kc * CO;
#pragma unroll
            for (int h = 0; h < HI; ++h) priv[t * HI + h][tid] = *reinterpret_cast<const f32x4 *>(p.g + ((size_t)b * p.M + kc) * CI + NI * n16 + 4 * h);
#pragma unroll
            for (int h = 0; h < HO; ++h) priv[4 * HI + t * HO + h][tid] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        size_t gbase = ((size_t)b * p.N + n_begin) * p.M * CO + NO * n16;      // strip being REQUESTED
        size_t fbase = ((size_t)b * p.N + n_begin) * CI + NI * n16;            // f row being REQUESTED
        const size_t g_step = (size_t)p.M * CO;
        int greq = 0, freq = 0;                                               // strips requested so far - 1
        f32x4 gv[4][HO], yv[4][HO], pv[2][HI], fv[HI], fvn[HI], gfix[HI], bkr[HO], bn_sum[HO];
        auto ld_g = [&](int t) {
#pragma unroll
            for (int h = 0; h < HO; ++h) gv[t][h] = ldx(p.gz + gbase + rowg[t] + 4 * h);
        };
        auto ld_y = [&](int t) {
#pragma unroll
            for (int h = 0; h < HO; ++h) yv[t][h] = ldx(p.y2 + gbase + rowg[t] + 4 * h);
        };
        auto ld_f = [&](f32x4 (&dst)[HI]) {
#pragma unroll
            for (int h = 0; h < HI; ++h) dst[h] = *reinterpret_cast<const f32x4 *>(p.f + fbase + 4 * h);
        };
        auto adv_g = [&]() { if (greq + 1 < ns) { gbase += g_step; ++greq; } };
        auto adv_f = [&]() { if (freq + 1 < ns) { fbase += CI; ++freq; } };
        auto tf_g = [&](int t, int h) {                          // g^y, zero on the rows past M; d_bias sums
#pragma unroll
            for (int c = 0; c < 4; ++c)
                gv[t][h][c] = vm[t] * __builtin_fmaf(csc[h][c], gv[t][h][c], __builtin_fmaf(yv[t][h][c], cbc[h][c], cac[h][c]));
            bn_sum[h] += gv[t][h];
        };
#pragma unroll
        for (int h = 0; h < HO; ++h) bn_sum[h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) { ld_g(t); ld_y(t); }
        ld_f(fv); adv_f(); ld_f(fvn); adv_f();
        adv_g();
        // k-step 0 of the first strip: operands formed here
#pragma unroll
        for (int h = 0; h < HO; ++h) { tf_g(0, h); priv[4 * HI + h][tid] += gv[0][h]; }
#pragma unroll
        for (int h = 0; h < HI; ++h) pv[0][h] = priv[h][tid] * fv[h];
        ld_y(0);
        for (int sidx = 0; sidx < ns; ++sidx) {
            const size_t bn_off = (size_t)kt * p.B * p.N * CO + ((size_t)b * p.N + n_begin + sidx) * CO + NO * n16;
            const bool has_next = sidx + 1 < ns;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int tn = (t + 1) & 3;                      // k-step whose operands are formed during this one
#pragma unroll
                for (int u = 0; u < SM; ++u) {
                    const int jo = u / NI, jc = u % NI;
                    acc[jo][jc] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[t][jo >> 2][jo & 3], pv[t & 1][jc >> 2][jc & 3], acc[jo][jc], 0, 0, 0);
                    if (u == 0) {                                // lane-private rows of the step being prepared
#pragma unroll
                        for (int h = 0; h < HI; ++h) gfix[h] = priv[tn * HI + h][tid];
#pragma unroll
                        for (int h = 0; h < HO; ++h) bkr[h] = priv[4 * HI + tn * HO + h][tid];
                    }
                    if (t == 3 && u == 1) {                      // all four k-steps of THIS strip are in bn_sum: sum over the 4 k-slots, store
#pragma unroll
                        for (int h = 0; h < HO; ++h) {
                            f32x4 v = bn_sum[h];
#pragma unroll
                            for (int c = 0; c < 4; ++c) { float a = v[c]; a += __shfl_xor(a, 16); a += __shfl_xor(a, 32); v[c] = a; }
                            if (q == 0) *reinterpret_cast<f32x4 *>(p.s_dbn + bn_off + 4 * h) = v;
                            bn_sum[h] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                    }
                    if (u >= 4 && u < 4 + HO) tf_g(tn, u - 4);
                    if (u == 4 + HO && (t < 3 || has_next)) {        // (t = 3 prepares the NEXT strip's first step: none after the last)
#pragma unroll
                        for (int h = 0; h < HO; ++h) { bkr[h] += gv[tn][h]; priv[4 * HI + tn * HO + h][tid] = bkr[h]; }
                    }
                    if (u == 5 + HO) {
#pragma unroll
                        for (int h = 0; h < HI; ++h) pv[tn & 1][h] = gfix[h] * (t == 3 ? fvn[h] : fv[h]);
                        if (t < 3) ld_y(tn);
                    }
                    if (u == SM - 1) ld_g(t);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            adv_g();
            ld_y(0);
#pragma unroll
            for (int h = 0; h < HI; ++h) fv[h] = fvn[h];
            ld_f(fvn); adv_f();
        }
        // d_bias_k of this lane's pixels
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = k0 + 4 * t + q;
            if (k < p.M) {
#pragma unroll
                for (int h = 0; h < HO; ++h)
                    *reinterpret_cast<f32x4 *>(p.s_dbk + (size_t)nc * p.B * p.M * CO + ((size_t)b * p.M + k) * CO + NO * n16 + 4 * h) = priv[4 * HI + t * HO + h][tid];
            }
        }
    }
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int jo = 0; jo < NO; ++jo)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int h = 0; h < HI; ++h) {
                        float *dst = lds + (size_t)(NO * (4 * q + e) + jo) * CI + NI * n16 + 4 * h;
                        f32x4 v = {acc[jo][4 * h][e], acc[jo][4 * h + 1][e], acc[jo][4 * h + 2][e], acc[jo][4 * h + 3][e]};
                        if (w > 0) v += *reinterpret_cast<const f32x4 *>(dst);
                        *reinterpret_cast<f32x4 *>(dst) = v;
                    }
        }
        __syncthreads();
    }
    float *out = p.dw_partial + (size_t)blockIdx.x * CO * CI;
    for (int i = tid; i < CO * CI / 4; i += WR_THREADS)
        *reinterpret_cast<f32x4 *>(out + 4 * i) = *reinterpret_cast<const f32x4 *>(lds + 4 * i);
}


// sum over the 16 lanes of a DPP row (every lane ends with the row's sum): quad_perm[1,0,3,2], quad_perm[2,3,0,1],
// row_half_mirror, row_mirror
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(b, b, CTRL, 0xF, 0xF, false));
}

// kernel D of the pair-layer backward (see above): dP = g^y . W per strip on the dgrad machinery; d_g[b,k,:] += dP .* f[b,n,:]
// in lane-private LDS rows (the lane's pixel is fixed, one slab per point chunk), d_f[b,n,:] = sum over the strip's 16
// pixels of dP .* g[b,k,:] (DPP row sums, one slab per pixel tile).  K = CO (channels of gz / y), C = CI.
template <int K, int C>
__global__ __launch_bounds__(WR_THREADS, 1) void wreg_pair_dgrad_kernel(WregPairP p) {
    constexpr int NT = C / 16, L = K / 4, NF = L / 4, G = 4 * NT, NMF = NT * L;
    __shared__ float gt[3 * K];
    __shared__ f32x4 priv[2 * NT][WR_THREADS];                  // rows [0, NT): g of the lane's pixel; [NT, 2NT): d_g sums
    __shared__ float fbuf[4][2][C];                             // f[b,n,:] of the strip being computed / stored, per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave-uniform on purpose: strip counters and steps live in SGPRs
    const int m = lane & 15, q = lane >> 4;
    for (int ch = tid; ch < K; ch += WR_THREADS) {
        double sd = 0.0, sx = 0.0;
#pragma unroll 8
        for (int rp = 0; rp < REP; ++rp) { sd += p.g_dsums[(size_t)rp * 2 * K + ch]; sx += p.g_dsums[(size_t)rp * 2 * K + K + ch]; }
        const float m1 = (float)(sd / (double)p.g_rows), m2 = (float)(sx / (double)p.g_rows);
        const float sc = p.g_oc[K + ch], mu = p.g_omi[ch], is = p.g_omi[K + ch];
        const float bc = -sc * is * m2;
        gt[ch] = sc; gt[K + ch] = -sc * m1 - bc * mu; gt[2 * K + ch] = bc;
    }
    __syncthreads();
    f32x4 wr[NT][NF];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int e = 0; e < 4; ++e) wr[j][f][e] = p.w[(size_t)(16 * f + 4 * q + e) * C + 16 * j + m];
    const float *gq = gt + 4 * q;

    const int ntasks = p.B * p.KT * p.NCH;
    for (int task = blockIdx.x * 4 + wave; task < ntasks; task += gridDim.x * 4) {
        const int nc = task % p.NCH, kt = (task / p.NCH) % p.KT, b = task / (p.NCH * p.KT);
        const int k0 = kt * WR_ROWS, n_begin = nc * p.NL, n_end = n_begin + p.NL < p.N ? n_begin + p.NL : p.N;
        const int ns = n_end - n_begin;
        if (ns <= 0) continue;
        const int kpix = k0 + m;
        const float vm = kpix < p.M ? 1.f : 0.f;
        const int kc = kpix < p.M ? kpix : p.M - 1;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            priv[j][tid] = *reinterpret_cast<const f32x4 *>(p.g + ((size_t)b * p.M + kc) * C + 16 * j + 4 * q);
            priv[NT + j][tid] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        size_t koff = (((size_t)b * p.N + n_begin) * p.M + kc) * K + 4 * q;       // strip being REQUESTED (gz / y)
        const size_t k_step = (size_t)p.M * K;
        int kreq = 0;
        auto adv_k = [&]() { if (kreq + 1 < ns) { koff += k_step; ++kreq; } };
        size_t dfoff = (size_t)kt * p.B * p.N * C + ((size_t)b * p.N + n_begin) * C + 4 * q;   // d_f slab row of the strip being STORED

        f32x4 tsc, tac, tbc, tu, gfix, dgr, tprod, frj, ftmp, x[NF], yb[NF], accA[NT], accB[NT];
        tsc = tac = tbc = tu = gfix = dgr = tprod = frj = ftmp = f32x4{0.f, 0.f, 0.f, 0.f};
        const size_t f_row0 = ((size_t)b * p.N + n_begin) * C;
        int fidx = 0;                                            // strip whose f row is requested next
        constexpr int LAT = 5, PE = 10;
        static_assert(PE * NT <= NMF && LAT + 2 <= G, "slot plan");
        auto row_sum = [&](f32x4 &v) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float a = v[c];
                a += dpp_f32<0xB1>(a); a += dpp_f32<0x4E>(a); a += dpp_f32<0x141>(a); a += dpp_f32<0x140>(a);
                v[c] = a;
            }
        };
        auto final_epilogue = [&](f32x4 (&prev)[NT], int par) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                f32x4 d = priv[NT + j][tid], gf = priv[j][tid];
                const f32x4 fj = *reinterpret_cast<const f32x4 *>(&fbuf[wave][par][16 * j + 4 * q]);
#pragma unroll
                for (int c = 0; c < 4; ++c) d[c] = __builtin_fmaf(prev[j][c], fj[c], d[c]);
                priv[NT + j][tid] = d;
                f32x4 t = prev[j] * gf;
                row_sum(t);
                if (m == 0) *reinterpret_cast<f32x4 *>(p.s_df + dfoff + 16 * j) = t;
            }
        };
        auto tf_all = [&](int f) {
            tsc = *reinterpret_cast<const f32x4 *>(gq + 16 * f); tac = *reinterpret_cast<const f32x4 *>(gq + K + 16 * f);
            tbc = *reinterpret_cast<const f32x4 *>(gq + 2 * K + 16 * f);
#pragma unroll
            for (int c = 0; c < 4; ++c) x[f][c] = vm * __builtin_fmaf(tsc[c], x[f][c], __builtin_fmaf(yb[f][c], tbc[c], tac[c]));
        };
        // `par`: parity of the strip being computed = fbuf row its f is written to (the store phase reads the other one)
        auto strip_block = [&](auto epi_tag, auto par_tag, f32x4 (&acc)[NT], f32x4 (&prev)[NT]) {
            constexpr bool EPI = decltype(epi_tag)::value;
            constexpr int PAR = decltype(par_tag)::value;
#pragma unroll
            for (int i = 0; i < NMF; ++i) {
                const int t = i / NT, j = i % NT, f = t >> 2, e = t & 3;
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[j][f][e], x[f][e], t == 0 ? zero : acc[j], 0, 0, 0);
                if (i == G * (f + 1) - 1 && (f & 1)) { x[f - 1] = ldx(p.gz + koff + 16 * (f - 1)); x[f] = ldx(p.gz + koff + 16 * f); }
                // f[b,n,:] of this strip: one float4 per lane of the first C/4 lanes, through a wave-shared LDS row (32
                // registers per lane otherwise: the kernel spilled weights, and every reload waited for vmcnt(0))
                if (i == 1 && lane < C / 4) ftmp = *reinterpret_cast<const f32x4 *>(p.f + f_row0 + (size_t)fidx * C + 4 * lane);
                if (i == NMF / 2 && lane < C / 4) *reinterpret_cast<f32x4 *>(&fbuf[wave][PAR][4 * lane]) = ftmp;
                {
                    const int fn = (f + 1) % NF, u = i - G * f;
                    if (u == 0) {
                        tsc = *reinterpret_cast<const f32x4 *>(gq + 16 * fn); tac = *reinterpret_cast<const f32x4 *>(gq + K + 16 * fn);
                        tbc = *reinterpret_cast<const f32x4 *>(gq + 2 * K + 16 * fn);
                    }
                    if (u == LAT) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) tu[c] = __builtin_fmaf(yb[fn][c], tbc[c], tac[c]);
                    }
                    if (u == LAT + 1) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) x[fn][c] = vm * __builtin_fmaf(tsc[c], x[fn][c], tu[c]);
                        if (fn & 1) { yb[fn - 1] = ldx(p.y2 + koff + 16 * (fn - 1)); yb[fn] = ldx(p.y2 + koff + 16 * fn); }
                    }
                }
                if (EPI && i < PE * NT) {
                    const int tj = i / PE, part = i % PE;
                    if (part == 0) { gfix = priv[tj][tid]; dgr = priv[NT + tj][tid]; frj = *reinterpret_cast<const f32x4 *>(&fbuf[wave][PAR ^ 1][16 * tj + 4 * q]); }
                    if (part == LAT) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) dgr[c] = __builtin_fmaf(prev[tj][c], frj[c], dgr[c]);
                        priv[NT + tj][tid] = dgr;
                    }
                    if (part == LAT + 1) tprod = prev[tj] * gfix;
                    if (part == LAT + 2) row_sum(tprod);
                    if (part == LAT + 3 && m == 0) *reinterpret_cast<f32x4 *>(p.s_df + dfoff + 16 * tj) = tprod;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (EPI) dfoff += C;
            ++fidx;
            adv_k();
        };
#pragma unroll
        for (int f = 0; f < NF; ++f) { x[f] = ldx(p.gz + koff + 16 * f); yb[f] = ldx(p.y2 + koff + 16 * f); }
        adv_k();
        tf_all(0);
        using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
        strip_block(std::false_type{}, P0{}, accA, accB);
        int k = 1;
        for (; k + 1 < ns; k += 2) {
            strip_block(std::true_type{}, P1{}, accB, accA);
            strip_block(std::true_type{}, P0{}, accA, accB);
        }
        if (k < ns) {
            strip_block(std::true_type{}, P1{}, accB, accA);
            final_epilogue(accB, 1);
        } else {
            final_epilogue(accA, 0);
        }
        if (kpix < p.M) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
                *reinterpret_cast<f32x4 *>(p.s_dg + (size_t)nc * p.B * p.M * C + ((size_t)b * p.M + kpix) * C + 16 * j + 4 * q) = priv[NT + j][tid];
        }
    }
}

template <int CO, int CI>
int launch_wreg_pair(const WregPairP &p, unsigned grid, hipStream_t st) {
    hipLaunchKernelGGL((wreg_pair_dgrad_kernel<CO, CI>), dim3(grid), dim3(WR_THREADS), 0, st, p);
    hipLaunchKernelGGL((wreg_pair_wgrad_kernel<CO, CI>), dim3(grid), dim3(WR_THREADS), 0, st, p);
    I2P_RETURN_LAUNCH_STATUS();
}


// =====================================================================================================================
// FORWARD of the pair layer (first cost-volume layer): y[b,n,k,:] = W (f[b,n,:] .* g[b,k,:]) + bias_n[b,n,:] + bias_k[b,k,:]
// on the tasks of the pair backward (a wave = one sample, one 16-pixel tile, a chunk of points; strip = rows (b, n, k0..)).
// Nothing large is READ: the lane's pixel row of g and of bias_k sit in lane-private LDS rows, the strip's rows of f and
// bias_n go through wave-shared LDS rows a strip ahead, the B operand of the next strip is formed in place from them
// right after a float4's last MFMA.  The only HBM traffic is the 437 MB of y.  (mlp.hip lin_fwd2_kernel<.., PAIR> is
// the general version.)
// =====================================================================================================================
struct WregPairFwdP {
    int B, N, M, KT, NCH, NL;
    const float *f, *g, *bias_n, *bias_k, *w;
    float *y; double *sums;
    unsigned *fin_counter; const float *fin_gamma, *fin_beta; float fin_eps; float *fin_coef, *fin_mi;
};

template <int CIN, int COUT>
__global__ __launch_bounds__(WR_THREADS, 1) void wreg_pair_fwd_kernel(WregPairFwdP p) {
    constexpr int NT = COUT / 16, L = CIN / 4, NF = L / 4, G = 4 * NT, NMF = NT * L;
    __shared__ f32x4 st_lds[2 * NT][WR_THREADS];                // per-lane BN statistics
    __shared__ f32x4 priv[NF + NT][WR_THREADS];                 // rows [0, NF): g of the lane's pixel; [NF, NF+NT): bias_k
    __shared__ float fbuf[4][2][CIN];                           // f[b,n,:] per wave, two strips in flight
    __shared__ float nbuf[4][2][COUT];                          // bias_n[b,n,:] per wave
    __shared__ int fin_flag;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, q = lane >> 4;
    f32x4 wr[NT][NF];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int f = 0; f < NF; ++f) wr[j][f] = *reinterpret_cast<const f32x4 *>(p.w + (size_t)(16 * j + m) * CIN + 16 * f + 4 * q);
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) st_lds[i][tid] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ntasks = p.B * p.KT * p.NCH;
    for (int task = blockIdx.x * 4 + wave; task < ntasks; task += gridDim.x * 4) {
        const int nc = task % p.NCH, kt = (task / p.NCH) % p.KT, b = task / (p.NCH * p.KT);
        const int k0 = kt * WR_ROWS, n_begin = nc * p.NL, n_end = n_begin + p.NL < p.N ? n_begin + p.NL : p.N;
        const int ns = n_end - n_begin;
        if (ns <= 0) continue;
        const int kpix = k0 + m;
        const bool live = kpix < p.M;
        const float vm = live ? 1.f : 0.f;
        const int kc = live ? kpix : p.M - 1;
#pragma unroll
        for (int f = 0; f < NF; ++f) priv[f][tid] = *reinterpret_cast<const f32x4 *>(p.g + ((size_t)b * p.M + kc) * CIN + 16 * f + 4 * q);
#pragma unroll
        for (int j = 0; j < NT; ++j) priv[NF + j][tid] = *reinterpret_cast<const f32x4 *>(p.bias_k + ((size_t)b * p.M + kc) * COUT + 16 * j + 4 * q);
        const size_t row0 = (size_t)b * p.N + n_begin;          // first (b, n) row of the chunk in f / bias_n
        auto at = [](float *base, unsigned byte_off) -> float * { return reinterpret_cast<float *>(reinterpret_cast<char *>(base) + byte_off); };
        unsigned yoff = (unsigned)(((row0 * p.M + kc) * COUT + 4 * q) * 4);      // strip being STORED (byte offset, y < 4 GB)
        const unsigned y_step = (unsigned)((size_t)p.M * COUT * 4);

        // rows of f / bias_n of strip s (clamped to the chunk) -> this lane's float4 of the row (first CIN/4 or COUT/4 lanes)
        auto f_row = [&](int s_) { const int c = s_ < ns ? s_ : ns - 1; return *reinterpret_cast<const f32x4 *>(p.f + (row0 + c) * CIN + 4 * (lane & (CIN / 4 - 1))); };
        auto n_row = [&](int s_) { const int c = s_ < ns ? s_ : ns - 1; return *reinterpret_cast<const f32x4 *>(p.bias_n + (row0 + c) * COUT + 4 * (lane & (COUT / 4 - 1))); };
        if (lane < CIN / 4) { *reinterpret_cast<f32x4 *>(&fbuf[wave][0][4 * lane]) = f_row(0); *reinterpret_cast<f32x4 *>(&fbuf[wave][1][4 * lane]) = f_row(1); }

        f32x4 x[NF], accA[NT], accB[NT], gfix, frow, ftmp, btmp, bk, bnr, r1, r2, ev, vkeep;
        gfix = frow = ftmp = btmp = bk = bnr = r1 = r2 = ev = vkeep = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < NF - 1; ++f) x[f] = priv[f][tid] * *reinterpret_cast<const f32x4 *>(&fbuf[wave][0][16 * f + 4 * q]);
        x[NF - 1] = f32x4{0.f, 0.f, 0.f, 0.f};                  // (formed in the first slots of the strip's own block)
        constexpr int LAT = 5, PE = 10;
        static_assert(PE * NT + LAT + 1 <= NMF && LAT + 1 < G, "slot plan");
        int sidx = 0;                                            // strip being computed
        auto epi_tile = [&](const f32x4 &acc, int j, int par) {  // (not slotted: last strip of the chunk)
            f32x4 v = acc + priv[NF + j][tid] + *reinterpret_cast<const f32x4 *>(&nbuf[wave][par][16 * j + 4 * q]);
            f32x4 a = st_lds[j][tid], c2 = st_lds[NT + j][tid];
            const f32x4 vs = v * vm;
            a += vs;
#pragma unroll
            for (int c = 0; c < 4; ++c) c2[c] = __builtin_fmaf(vs[c], v[c], c2[c]);
            st_lds[j][tid] = a; st_lds[NT + j][tid] = c2;
            if (live) sty(at(p.y + 16 * j, yoff), v);
        };
        auto strip_block = [&](auto epi_tag, auto par_tag, f32x4 (&acc)[NT], f32x4 (&prev)[NT]) {
            constexpr bool EPI = decltype(epi_tag)::value;
            constexpr int PAR = decltype(par_tag)::value;        // parity of the strip being computed
#pragma unroll
            for (int i = 0; i < NMF; ++i) {
                const int t = i / NT, j = i % NT, f = t >> 2, e = t & 3;
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                // (the last float4 of THIS strip is formed in slots 0..LAT: its MFMAs start at slot G (NF-1))
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[j][f][e], x[f][e], t == 0 ? zero : acc[j], 0, 0, 0);
                // rows of f for the strip after the next one, of bias_n for this strip: registers now, LDS half a strip later
                if (i == 1) { if (lane < CIN / 4) ftmp = f_row(sidx + 2); if (lane < COUT / 4) btmp = n_row(sidx); }
                if (i == NMF / 2) {
                    if (lane < CIN / 4) *reinterpret_cast<f32x4 *>(&fbuf[wave][PAR][4 * lane]) = ftmp;
                    if (lane < COUT / 4) *reinterpret_cast<f32x4 *>(&nbuf[wave][PAR][4 * lane]) = btmp;
                }
                // ---- B operand: float4 NF-1 of this strip at the start, float4 f of the NEXT strip once its last MFMA issued ----
                if (i == 0) { gfix = priv[NF - 1][tid]; frow = *reinterpret_cast<const f32x4 *>(&fbuf[wave][PAR][16 * (NF - 1) + 4 * q]); }
                if (i == LAT) x[NF - 1] = gfix * frow;
                {
                    const int fp = i / G - 1, u = i - G * (fp + 1);      // float4 whose group ended at slot G (fp + 1) - 1
                    if (fp >= 0 && fp < NF - 1) {
                        if (u == 0 && i >= LAT + 1) { gfix = priv[fp][tid]; frow = *reinterpret_cast<const f32x4 *>(&fbuf[wave][PAR ^ 1][16 * fp + 4 * q]); }
                        if (u == LAT) x[fp] = gfix * frow;
                    }
                }
                // ---- store phase of the previous strip ------------------------------------------------------------------
                if (EPI && i >= LAT + 1 && i < PE * NT + LAT + 1) {
                    const int tj = (i - LAT - 1) / PE, part = (i - LAT - 1) % PE;
                    if (tj < NT) {
                        if (part == 0) { bk = priv[NF + tj][tid]; bnr = *reinterpret_cast<const f32x4 *>(&nbuf[wave][PAR ^ 1][16 * tj + 4 * q]); r1 = st_lds[tj][tid]; r2 = st_lds[NT + tj][tid]; }
                        if (part == LAT) {
                            ev = prev[tj] + bk + bnr;
                            if (tj & 1) { if (live) { sty(at(p.y + 16 * (tj - 1), yoff), vkeep); sty(at(p.y + 16 * tj, yoff), ev); } }
                            else vkeep = ev;
                        }
                        if (part == LAT + 1) { r1 += ev * vm; st_lds[tj][tid] = r1; }
                        if (part == LAT + 2) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) r2[c] = __builtin_fmaf(ev[c] * vm, ev[c], r2[c]);
                            st_lds[NT + tj][tid] = r2;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (EPI) yoff += y_step;
            ++sidx;
        };
        using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
        strip_block(std::false_type{}, P0{}, accA, accB);
        int k = 1;
        for (; k + 1 < ns; k += 2) {
            strip_block(std::true_type{}, P1{}, accB, accA);
            strip_block(std::true_type{}, P0{}, accA, accB);
        }
        if (k < ns) {
            strip_block(std::true_type{}, P1{}, accB, accA);
#pragma unroll
            for (int j = 0; j < NT; ++j) epi_tile(accB[j], j, 1);
        } else {
#pragma unroll
            for (int j = 0; j < NT; ++j) epi_tile(accA[j], j, 0);
        }
    }
    __syncthreads();
    if (p.sums) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double a = (double)st_lds[j][tid][e], b2 = (double)st_lds[NT + j][tid][e];
#pragma unroll
                for (int off = 8; off >= 1; off >>= 1) { a += __shfl_xor(a, off); b2 += __shfl_xor(b2, off); }
                if (m == 0) {
                    double *rep = p.sums + (size_t)((blockIdx.x * 4 + wave) % REP) * 2 * COUT;
                    atomicAdd(rep + 16 * j + 4 * q + e, a); atomicAdd(rep + COUT + 16 * j + 4 * q + e, b2);
                }
            }
    }
    if (p.fin_counter) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) fin_flag = i2p_ticket_is_last(p.fin_counter, gridDim.x) ? 1 : 0;
        __syncthreads();
        if (!fin_flag) return;
        const double rows = (double)p.B * p.N * p.M;
        for (int ch = tid; ch < COUT; ch += WR_THREADS) {
            double sa = 0.0, qa = 0.0;
            for (int r = 0; r < REP; ++r) {
                sa += __hip_atomic_load(p.sums + (size_t)r * 2 * COUT + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                qa += __hip_atomic_load(p.sums + (size_t)r * 2 * COUT + COUT + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const double mean = sa / rows;
            double var = qa / rows - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            const float invstd = rsqrtf((float)var + p.fin_eps);
            p.fin_coef[ch] = (float)mean; p.fin_coef[COUT + ch] = invstd * p.fin_gamma[ch]; p.fin_coef[2 * COUT + ch] = p.fin_beta[ch];
            p.fin_mi[ch] = (float)mean; p.fin_mi[COUT + ch] = invstd;
        }
        if (tid == 0) *p.fin_counter = 0u;
    }
}


// =====================================================================================================================
// wgrad of the NARROW layers on many rows (level-1 set abstraction: 12/16 -> 16 -> 16 -> 32 channels on B*3600*32 rows):
// pure HBM streaming (3 tensors of 48-128 bytes per row), the contraction is 1-2 MFMAs per 4 rows.  Rows are the K axis,
// lane (channel n = lane & 15, k-slot q): one DWORD per tensor, tile and k-step — with 64-byte rows a load instruction of
// the wave covers four whole rows.  No registers to speak of, so 16 waves per CU hide the latency by occupancy; the next
// strip is requested before the current one is consumed.  (mlp.hip lin_wgrad_kernel<1,1> streams these at 2 TB/s.)
// =====================================================================================================================
struct SmallWgradP {
    long long rows;              // multiple of 16
    int cin, cout;               // cin <= 16, cout = 16 * NO
    const float *gz, *y2;        // [rows, cout]
    const double *g_dsums; const float *g_oc, *g_omi; long long g_rows; float g_slope;
    float *bn_out;
    const float *x, *in_coef; float slope_in;
    float *dw_partial;           // [grid][cout*cin]
};
constexpr int SW_THREADS = 1024;

template <int NO>
__global__ __launch_bounds__(SW_THREADS) void small_wgrad_kernel(SmallWgradP p) {
    __shared__ float red[16 * NO * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int CO = 16 * NO, CI = p.cin;
    // per-lane constants: output channels 16 jo + n (g^y = sc (t - m1 - xhat m2), t = act'(z) gz), input channel n
    float m1[NO], m2[NO], sc[NO], mu[NO], is[NO], zb[NO];
#pragma unroll
    for (int jo = 0; jo < NO; ++jo) {
        const int ch = 16 * jo + n;
        double sd = 0.0, sx = 0.0;
#pragma unroll 8
        for (int rp = 0; rp < REP; ++rp) { sd += p.g_dsums[(size_t)rp * 2 * CO + ch]; sx += p.g_dsums[(size_t)rp * 2 * CO + CO + ch]; }
        m1[jo] = (float)(sd / (double)p.g_rows); m2[jo] = (float)(sx / (double)p.g_rows);
        sc[jo] = p.g_oc[CO + ch]; mu[jo] = p.g_omi[ch]; is[jo] = p.g_omi[CO + ch]; zb[jo] = p.g_oc[2 * CO + ch] - mu[jo] * sc[jo];
        if (blockIdx.x == 0 && wave == 0 && q == 0 && p.bn_out) { p.bn_out[6 * CO + ch] = (float)sd; p.bn_out[7 * CO + ch] = (float)sx; }
    }
    const bool xin = n < CI;
    float xa = 1.f, xb = 0.f;
    if (p.in_coef && xin) { xa = p.in_coef[CI + n]; xb = p.in_coef[2 * CI + n] - p.in_coef[n] * xa; }
    const bool gact = p.g_slope != 1.f;

    f32x4 acc[NO];
#pragma unroll
    for (int jo = 0; jo < NO; ++jo) acc[jo] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long nstrips = p.rows / WR_ROWS;
    const long long stride = (long long)gridDim.x * (SW_THREADS / 64);
    float g[4][NO], yy[4][NO], xx[4], gn[4][NO], yn[4][NO], xn[4];
    auto load = [&](long long s, float (&G)[4][NO], float (&Y)[4][NO], float (&X)[4]) {
        const size_t r0 = (size_t)s * WR_ROWS + q;               // k-step t: row r0 + 4t
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int jo = 0; jo < NO; ++jo) {
                G[t][jo] = __builtin_nontemporal_load(p.gz + (r0 + 4 * t) * CO + 16 * jo + n);
                Y[t][jo] = __builtin_nontemporal_load(p.y2 + (r0 + 4 * t) * CO + 16 * jo + n);
            }
            X[t] = xin ? __builtin_nontemporal_load(p.x + (r0 + 4 * t) * CI + n) : 0.f;
        }
    };
    long long s = (long long)blockIdx.x * (SW_THREADS / 64) + wave;
    if (s < nstrips) load(s, g, yy, xx);
    for (; s < nstrips; s += stride) {
        const long long sn = s + stride < nstrips ? s + stride : s;
        load(sn, gn, yn, xn);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float a = xx[t];
            if (p.in_coef) { const float z = __builtin_fmaf(a, xa, xb); a = z > 0.f ? z : z * p.slope_in; }
            if (!xin) a = 0.f;
#pragma unroll
            for (int jo = 0; jo < NO; ++jo) {
                float tg = g[t][jo];
                if (gact) tg = __builtin_fmaf(yy[t][jo], sc[jo], zb[jo]) > 0.f ? tg : tg * p.g_slope;
                const float gy = sc[jo] * (tg - m1[jo] - ((yy[t][jo] - mu[jo]) * is[jo]) * m2[jo]);
                acc[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(gy, a, acc[jo], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            xx[t] = xn[t];
#pragma unroll
            for (int jo = 0; jo < NO; ++jo) { g[t][jo] = gn[t][jo]; yy[t][jo] = yn[t][jo]; }
        }
    }
    // D[o][c]: lane (c = n, q): rows o = 16 jo + 4q + e.  The waves of the block add through LDS in a fixed order.
    for (int w = 0; w < SW_THREADS / 64; ++w) {
        if (wave == w) {
#pragma unroll
            for (int jo = 0; jo < NO; ++jo)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float *dst = red + (16 * jo + 4 * q + e) * 16 + n;
                    *dst = (w > 0 ? *dst : 0.f) + acc[jo][e];
                }
        }
        __syncthreads();
    }
    float *out = p.dw_partial + (size_t)blockIdx.x * CO * CI;
    for (int i = tid; i < CO * CI; i += SW_THREADS) out[i] = red[(i / CI) * 16 + i % CI];
}

}  // namespace

// rows below which the weights-in-registers kernels hand over to the LDS-resident second generation
static long long wreg_min_rows() {
    // 32768: the fine cost volume's layers (8 x 228 x 32 = 58 368 rows; nuScenes 43 776) run 2-3x faster here than on the
    // second generation (A/B on one box: 615 -> 620 samples/s); below that a 1024-wave launch of 16-row strips is mostly pipeline fill
    return 32768;
}

bool i2p_wreg_fwd_ok(long long rows, int cin, int cout) {
    static const char *e = getenv("I2P_NO_WREG");
    if (e && e[0] == '1') return false;
    return rows >= wreg_min_rows() && (rows % WR_ROWS) == 0 && (cin == 64 || cin == 128) && (cout == 64 || cout == 128);
}

int i2p_wreg_fwd(long long rows, int cin, int cout, const float *x, int x_ld, const float *in_coef, float slope, const float *w,
                 float *y, int y_ld, double *sums, unsigned *fin_counter, const float *fin_gamma, const float *fin_beta,
                 float fin_eps, float *fin_coef, float *fin_mi, void *stream, const float *xb, const float *in_coef_b,
                 float slope_b) {
    if (!i2p_wreg_fwd_ok(rows, cin, cout) || (x_ld & 3) || (y_ld & 3) || !(slope >= 0.f && slope <= 1.f)) return I2P_ERR_BAD_ARG;
    WregP p;
    if (xb && (x_ld * 2 != cin || !in_coef || !in_coef_b || !(slope_b >= 0.f && slope_b <= 1.f))) return I2P_ERR_BAD_ARG;
    p.rows = rows; p.x = x; p.x_ld = x_ld; p.in_coef = in_coef; p.slope = slope; p.w = w; p.y = y; p.y_ld = y_ld; p.sums = sums;
    p.xb = xb; p.in_coef_b = in_coef_b; p.slope_b = slope_b;
    p.fin_counter = fin_counter; p.fin_gamma = fin_gamma; p.fin_beta = fin_beta; p.fin_eps = fin_eps; p.fin_coef = fin_coef; p.fin_mi = fin_mi;
    hipStream_t st = (hipStream_t)stream;
    if (cin == 128 && cout == 128) return launch_wreg<128, 128>(p, st);
    if (cin == 128 && cout == 64) return launch_wreg<128, 64>(p, st);
    if (cin == 64 && cout == 128) return launch_wreg<64, 128>(p, st);
    return launch_wreg<64, 64>(p, st);
}

bool i2p_wreg_dgrad_ok(long long rows, int k, int c) {
    static const char *e = getenv("I2P_NO_WREG");
    if (e && e[0] == '1') return false;
    return rows >= wreg_min_rows() && (rows % WR_ROWS) == 0 && (k == 64 || k == 128) && (c == 64 || c == 128);
}

int i2p_wreg_dgrad(long long rows, int k, int c, const float *gz, const float *y2, const double *g_dsums, const float *g_oc,
                   const float *g_omi, long long g_rows, const float *w, float *gz_in, const float *ex, const float *e_coef,
                   const float *e_mi, float e_slope, double *sums, void *stream, float *gz_in_b, const float *exb,
                   const float *e_coef_b, const float *e_mi_b, float e_slope_b, const float *e_add, double *sums_b) {
    if (!i2p_wreg_dgrad_ok(rows, k, c) || !gz || !y2 || !g_dsums || !g_oc || !g_omi || !w || !gz_in || !ex || !e_coef || !e_mi)
        return I2P_ERR_BAD_ARG;
    if (gz_in_b && (!exb || !e_coef_b || !e_mi_b || !e_add || !sums || !sums_b)) return I2P_ERR_BAD_ARG;
    if ((unsigned long long)rows * (unsigned)(k > c ? k : c) * 4ull >= (1ull << 32)) return I2P_ERR_BAD_ARG;     // 32-bit byte offsets
    WregDgradP p;
    p.rows = rows; p.gz = gz; p.y2 = y2; p.g_dsums = g_dsums; p.g_oc = g_oc; p.g_omi = g_omi; p.g_rows = g_rows; p.w = w;
    p.gz_in = gz_in; p.ex = ex; p.e_coef = e_coef; p.e_mi = e_mi; p.e_slope = e_slope; p.sums = sums;
    p.gz_in_b = gz_in_b; p.exb = exb; p.e_coef_b = e_coef_b; p.e_mi_b = e_mi_b; p.e_slope_b = e_slope_b; p.e_add = e_add; p.sums_b = sums_b;
    hipStream_t st = (hipStream_t)stream;
    if (k == 128 && c == 128) return launch_wreg_dgrad<128, 128>(p, st);
    if (k == 128 && c == 64) return launch_wreg_dgrad<128, 64>(p, st);
    if (k == 64 && c == 128) return launch_wreg_dgrad<64, 128>(p, st);
    return launch_wreg_dgrad<64, 64>(p, st);
}

// wgrad: writes dw_partial[grid][cout*cin] (grid = the caller's i2p_lin_bwd_grid(rows), 256 for these row counts) and
// rows 6, 7 of bn_out; the caller reduces the partials.
bool i2p_wreg_wgrad_ok(long long rows, int cin, int cout) {
    static const char *e = getenv("I2P_NO_WREG");
    if (e && e[0] == '1') return false;
    return rows >= wreg_min_rows() && (rows % WR_ROWS) == 0 && (cin == 64 || cin == 128) && (cout == 64 || cout == 128);
}

int i2p_wreg_wgrad(long long rows, int cin, int cout, const float *gz, const float *y2, const double *g_dsums, const float *g_oc,
                   const float *g_omi, long long g_rows, float *bn_out, const float *x, const float *in_coef, float slope_in,
                   const float *xb, const float *in_coef_b, float slope_b, int split, float *dw_partial, unsigned grid, void *stream) {
    if (!i2p_wreg_wgrad_ok(rows, cin, cout) || !gz || !y2 || !g_dsums || !g_oc || !g_omi || !x || !dw_partial || grid == 0 ||
        !(slope_in >= 0.f && slope_in <= 1.f))
        return I2P_ERR_BAD_ARG;
    if ((unsigned long long)rows * (unsigned)(cin > cout ? cin : cout) * 4ull >= (1ull << 32)) return I2P_ERR_BAD_ARG;   // 32-bit byte offsets
    if (xb && (split * 2 != cin || cin != 128 || !in_coef || !in_coef_b ||
               !(slope_b >= 0.f && slope_b <= 1.f)))
        return I2P_ERR_BAD_ARG;
    WregWgradP p;
    p.rows = rows; p.gz = gz; p.y2 = y2; p.g_dsums = g_dsums; p.g_oc = g_oc; p.g_omi = g_omi; p.g_rows = g_rows; p.bn_out = bn_out;
    p.x = x; p.in_coef = in_coef; p.slope_in = slope_in; p.dw_partial = dw_partial;
    p.xb = xb; p.in_coef_b = in_coef_b; p.slope_b = slope_b; p.split = split;
    hipStream_t st = (hipStream_t)stream;
    if (cout == 128 && cin == 128) return launch_wreg_wgrad<128, 128>(p, grid, st);
    if (cout == 128 && cin == 64) return launch_wreg_wgrad<128, 64>(p, grid, st);
    if (cout == 64 && cin == 128) return launch_wreg_wgrad<64, 128>(p, grid, st);
    return launch_wreg_wgrad<64, 64>(p, grid, st);
}

// pair-layer backward (first cost-volume layer) on the two kernels above.  Scratch layout (floats), returned by
// i2p_wreg_pair_bwd_scratch: [256][cout*cin] block partials of dW, then the slabs s_df [KT][B*N*cin], s_dbn [KT][B*N*cout],
// s_dg [NCH][B*M*cin], s_dbk [NCH][B*M*cout]; the caller reduces them (reduce_partials / slab_reduce, fixed order).
static void wreg_pair_geometry(int B, int N, int M, int &KT, int &NCH, int &NL) {
    KT = (M + WR_ROWS - 1) / WR_ROWS;
    NCH = 1024 / (B * KT > 0 ? B * KT : 1);
    NCH = NCH < 1 ? 1 : (NCH > N ? N : NCH);
    NL = (N + NCH - 1) / NCH;
    NCH = (N + NL - 1) / NL;                                     // every chunk non-empty: every slab row gets written
}
bool i2p_wreg_pair_bwd_ok(int B, int N, int M, int cin, int cout) {
    static const char *e = getenv("I2P_NO_WREG");
    if (e && e[0] == '1') return false;
    return (long long)B * N * M >= 65536 && M >= WR_ROWS && cin == 128 && cout == 128;
}
long long i2p_wreg_pair_bwd_scratch(int B, int N, int M, int cin, int cout) {
    int KT, NCH, NL; wreg_pair_geometry(B, N, M, KT, NCH, NL);
    return 256LL * cout * cin + (long long)KT * B * N * (cin + cout) + (long long)NCH * B * M * (cin + cout);
}
int i2p_wreg_pair_bwd(int B, int N, int M, int cin, int cout, const float *gz, const float *y2, const double *g_dsums,
                      const float *g_oc, const float *g_omi, const float *f, const float *g, const float *w, float *scratch,
                      int *KT_out, int *NCH_out, void *stream) {
    if (!i2p_wreg_pair_bwd_ok(B, N, M, cin, cout) || !gz || !y2 || !g_dsums || !g_oc || !g_omi || !f || !g || !w || !scratch)
        return I2P_ERR_BAD_ARG;
    WregPairP p;
    p.B = B; p.N = N; p.M = M; wreg_pair_geometry(B, N, M, p.KT, p.NCH, p.NL);
    p.gz = gz; p.y2 = y2; p.g_dsums = g_dsums; p.g_oc = g_oc; p.g_omi = g_omi; p.g_rows = (long long)B * N * M;
    p.f = f; p.g = g; p.w = w;
    p.dw_partial = scratch;
    p.s_df = scratch + (size_t)256 * cout * cin;
    p.s_dbn = p.s_df + (size_t)p.KT * B * N * cin;
    p.s_dg = p.s_dbn + (size_t)p.KT * B * N * cout;
    p.s_dbk = p.s_dg + (size_t)p.NCH * B * M * cin;
    *KT_out = p.KT; *NCH_out = p.NCH;
    if (i2p_wreg_pair_bwd_fused_ok())      // both halves from ONE read of gz / y (csrc/mlp_wreg_pair_fused.hip): same slabs, same geometry
        return i2p_wreg_pair_bwd_fused(B, N, M, p.KT, p.NCH, p.NL, gz, y2, g_dsums, g_oc, g_omi, f, g, w, p.dw_partial, p.s_df, p.s_dbn, p.s_dg,
                                       p.s_dbk, stream);
    return launch_wreg_pair<128, 128>(p, 256u, (hipStream_t)stream);
}

// pair-layer forward on wreg_pair_fwd_kernel (128 x 128, the shapes of i2p_wreg_pair_bwd_ok)
int i2p_wreg_pair_fwd(int B, int N, int M, int cin, int cout, const float *f, const float *g, const float *bias_n,
                      const float *bias_k, const float *w, float *y, double *sums, unsigned *fin_counter, const float *fin_gamma,
                      const float *fin_beta, float fin_eps, float *fin_coef, float *fin_mi, void *stream) {
    if (!i2p_wreg_pair_bwd_ok(B, N, M, cin, cout) || !f || !g || !bias_n || !bias_k || !w || !y) return I2P_ERR_BAD_ARG;
    if ((unsigned long long)B * N * M * cout * 4ull >= (1ull << 32)) return I2P_ERR_BAD_ARG;       // 32-bit byte offsets into y
    WregPairFwdP p;
    p.B = B; p.N = N; p.M = M; wreg_pair_geometry(B, N, M, p.KT, p.NCH, p.NL);
    p.f = f; p.g = g; p.bias_n = bias_n; p.bias_k = bias_k; p.w = w; p.y = y; p.sums = sums;
    p.fin_counter = fin_counter; p.fin_gamma = fin_gamma; p.fin_beta = fin_beta; p.fin_eps = fin_eps; p.fin_coef = fin_coef; p.fin_mi = fin_mi;
    hipLaunchKernelGGL((wreg_pair_fwd_kernel<128, 128>), dim3(256), dim3(WR_THREADS), 0, (hipStream_t)stream, p);
    I2P_RETURN_LAUNCH_STATUS();
}

// narrow layers on many rows (cin <= 16, cout 16 or 32): streaming wgrad, partials for the caller's reduce_partials launch
bool i2p_small_wgrad_ok(long long rows, int cin, int cout) {
    static const char *e = getenv("I2P_NO_WREG");
    if (e && e[0] == '1') return false;
    return rows >= 262144 && (rows % WR_ROWS) == 0 && cin >= 4 && cin <= 16 && (cout == 16 || cout == 32);
}
int i2p_small_wgrad(long long rows, int cin, int cout, const float *gz, const float *y2, const double *g_dsums, const float *g_oc,
                    const float *g_omi, long long g_rows, float g_slope, float *bn_out, const float *x, const float *in_coef,
                    float slope_in, float *dw_partial, unsigned grid, void *stream) {
    if (!i2p_small_wgrad_ok(rows, cin, cout) || !gz || !y2 || !g_dsums || !g_oc || !g_omi || !x || !dw_partial || grid == 0) return I2P_ERR_BAD_ARG;
    SmallWgradP p;
    p.rows = rows; p.cin = cin; p.cout = cout; p.gz = gz; p.y2 = y2; p.g_dsums = g_dsums; p.g_oc = g_oc; p.g_omi = g_omi; p.g_rows = g_rows;
    p.g_slope = g_slope; p.bn_out = bn_out; p.x = x; p.in_coef = in_coef; p.slope_in = slope_in; p.dw_partial = dw_partial;
    if (cout == 16) hipLaunchKernelGGL((small_wgrad_kernel<1>), dim3(grid), dim3(SW_THREADS), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((small_wgrad_kernel<2>), dim3(grid), dim3(SW_THREADS), 0, (hipStream_t)stream, p);
    I2P_RETURN_LAUNCH_STATUS();
}
