// Backward of a wide fp32 layer on many rows in ONE pass: input gradient AND weight gradient from a single read of gz, y and x
// (reference op: autograd of Conv2d.forward, PPBackbone_center.py:34-46, for the HBM-bound 128->64 and 64->64 layers of the cost
// volumes, PPBackbone_center.py:383-433).  mlp_wreg.hip runs the two halves as two kernels (wreg_dgrad_kernel, wreg_wgrad_kernel):
// each streams gz [rows,K], y [rows,K] and x [rows,C] — 0.87 GB twice per 128->64 layer at batch 8.
//
// One wave per SIMD with the 512-register budget (as mlp_wreg.hip).  Per 16-row strip of a wave:
//   * gz, y (BN backward of the layer behind formed on load: g^y = fma(sc, gz, fma(y, Bc, Ac))) and x arrive in the "row per lane"
//     layout of the input gradient: lane (m = row, q) holds channels 16f + 4q .. + 3 — one float4 per 16-channel block, straight from
//     global memory, the next strip requested before this one is used;
//   * input gradient  D[c][row] = sum_k W[k][c] g^y[row][k]:  W stationary as MFMA A fragments (K*C/64 registers), g^y as B operand
//     from the registers it was formed in;  epilogue as wreg_dgrad_kernel: activation derivative of the layer in front, its
//     BN-backward statistics (lane-private LDS rows), float4 stores straight from the accumulators;
//   * weight gradient dW[k][c] += sum_rows g^y[row][k] a[row][c],  a = act(bn(x)):  the contraction runs over ROWS, so the operands are
//     needed in the "channel per lane" layout (lane (n, q): rows 4q + t, channels NO*n .. / NI*n ..).  g^y and a are written to a
//     wave-private LDS tile in the first layout and read back in the second (12 + 12 16-byte LDS accesses per 256 MFMAs; the wave's
//     own LDS queue orders them, no barrier); the [K][C] accumulators stay in registers for the life of the kernel (128 at 64 x 128)
//     and leave once, through the block's fixed-order LDS sum, as this block's slab of dw_partial.
// Scheduling is the compiler's (no hand-placed slots as in mlp_wreg.hip): the kernel issues half the memory instructions per MFMA of the
// two-kernel form, which is what bounded those (tools/mfma_ceiling.py).
#include "common.h"
#include <type_traits>

// build-time diagnostic (I2P_BUILD_VARIANT / I2P_BUILD_DEFS=-DFUSED_ABL=n): 1 no weight-gradient MFMAs, 2 no LDS tiles, 4 no stores,
// 8 no requests of the next strip, 16 no input-gradient MFMAs
#ifndef FUSED_ABL
#define FUSED_ABL 0
#endif

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
constexpr int WF_THREADS = 256;
constexpr int WF_ROWS = 16;
constexpr int REP = I2P_BN_REPLICAS;

struct WregFusedP {
    long long rows;              // multiple of 16
    const float *gz, *y2;        // [rows, K]
    const double *g_dsums;       // [REP][2K] sums {gz, gz * xhat} of the BN behind
    const float *g_oc, *g_omi;   // its coef [3][K] and mean_invstd [2][K]
    long long g_rows;
    const float *w;              // [K][C]
    float *gz_in;                // [rows, C]
    const float *ex;             // [rows, C] pre-BN input of the layer
    const float *e_coef, *e_mi;  // [3][C], [2][C] of the BN in front
    float e_slope;               // 0 <= slope <= 1
    double *sums;                // [REP][2C] {sum g, sum g * xhat_in}
    float *bn_out;               // [8][K]: rows 6, 7 <- dbeta, dgamma of the BN behind (block 0) or nullptr
    float *dw_partial;           // [grid][K*C]   (TWO: [grid][K*2C])
    // TWO instantiation — the two-source layer 64 + 64 -> 128 (position encoding | mlp1 output, PPBackbone_center.py:418-425): the layer
    // input is two tensors of C channels each; source A = ex / e_coef / e_mi / e_slope / gz_in / sums above (W columns [0, C)), source B
    // below (W columns [C, 2C)); e_add [rows, C] is added to B's input gradient before its activation derivative (the gradient that
    // reaches that tensor on another path), as in wreg_dgrad_kernel<.., true>
    const float *exb, *e_coef_b, *e_mi_b, *e_add; float e_slope_b; float *gz_in_b; double *sums_b;
};

__device__ __forceinline__ f32x4 ldn(const float *ptr) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt *>(ptr)); }
// gz / y of the TWO instantiation: both waves of a strip read the same lines — a non-temporal first read does not stay in L2 and the
// partner fetched the line from HBM again (PMC, profiles/r06_pmc_traffic.json first pass: 2.86 GB per launch = 1.45x the algorithmic
// 1.97 GB); with the default policy the second read is an L2 hit
template <bool SHARED>
__device__ __forceinline__ f32x4 ldg(const float *ptr) {
    if constexpr (SHARED) return *reinterpret_cast<const f32x4 *>(ptr);
    else return ldn(ptr);
}
__device__ __forceinline__ void stn(float *ptr, const f32x4 &v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4_nt *>(ptr)); }

// TWO (round 6): K = 128 output channels cannot keep W (K*C/64 registers) AND dW for C = 128 input channels in one wave (512 registers
// for the two alone).  The two SOURCES of the layer split the work instead: a wave owns (strip, source) — W columns and dW columns of
// its 64 input channels (128 + 128 registers, the <64,128> budget), the source's x rows, its input-gradient destination and statistics
// — and forms g^y for all 128 output channels itself.  The two waves of a strip (neighbours in the block, on two SIMDs of one CU) read
// the same gz / y lines at about the same time: the second read is served by the CU's vector cache / L2, HBM sees each line once.
template <int K, int C, bool TWO>
__global__ __launch_bounds__(WF_THREADS, 1) void wreg_bwd_fused_kernel(WregFusedP p) {
    constexpr int CT = TWO ? 2 * C : C;    // input channels of the layer = row pitch of W and of a weight-gradient slab
    constexpr int NT = C / 16;             // input-gradient tiles (16 result channels each)
    constexpr int NF = K / 16;             // float4 of gz / y per lane and row
    constexpr int NO = K / 16, NI = C / 16;// weight-gradient tiles = channels per lane
    constexpr int HO = NO / 4, HI = NI / 4;
    constexpr int LDG = K + 4, LDA = C + 4;// LDS tile rows (+4: consecutive rows start 4 banks apart)
    static_assert(HO >= 1 && HI >= 1, "K, C multiples of 64");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *gt = smem;                                      // [3][K]  sc, Ac, Bc
    float *et_all = gt + 3 * K;                            // [1 or 2 sources][4][C]  e_sc, e_zb, e_is, e_nm
    f32x4 *st_lds = reinterpret_cast<f32x4 *>(et_all + (TWO ? 8 : 4) * C); // [2 NT][256] lane-private statistics rows
    float *tiles = reinterpret_cast<float *>(st_lds + 2 * NT * WF_THREADS);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int half = TWO ? (wave & 1) : 0;                 // which source this wave owns (wave-uniform)
    float *et = et_all + half * 4 * C;
    const float *src_x = half ? p.exb : p.ex;
    float *dst_g = half ? p.gz_in_b : p.gz_in;
    const float my_slope = half ? p.e_slope_b : p.e_slope;
    const int c0 = half * C;                               // first W / dW column of this wave
    float *tg = tiles + (size_t)wave * WF_ROWS * (LDG + 2 * LDA);   // this wave's g^y tile [16][LDG]
    float *ta = tg + WF_ROWS * LDG;                                 // its a tile [16][LDA]
    float *tx = ta + WF_ROWS * LDA;                                 // and the raw x rows (parked for the epilogue: 32 registers otherwise)

    for (int ch = tid; ch < K; ch += WF_THREADS) {
        double sd = 0.0, sx = 0.0;
#pragma unroll 8
        for (int rp = 0; rp < REP; ++rp) { sd += p.g_dsums[(size_t)rp * 2 * K + ch]; sx += p.g_dsums[(size_t)rp * 2 * K + K + ch]; }
        const float m1 = (float)(sd / (double)p.g_rows), m2 = (float)(sx / (double)p.g_rows);
        const float sc = p.g_oc[K + ch], mu = p.g_omi[ch], is = p.g_omi[K + ch];
        const float bc = -sc * is * m2;
        gt[ch] = sc; gt[K + ch] = -sc * m1 - bc * mu; gt[2 * K + ch] = bc;
        if (blockIdx.x == 0 && p.bn_out) { p.bn_out[6 * K + ch] = (float)sd; p.bn_out[7 * K + ch] = (float)sx; }
    }
    for (int i = tid; i < (TWO ? 2 : 1) * C; i += WF_THREADS) {
        const int sb = i / C, ch = i - sb * C;
        const float *ec = sb ? p.e_coef_b : p.e_coef, *em = sb ? p.e_mi_b : p.e_mi;
        float *e2 = et_all + sb * 4 * C;
        const float mean = ec[ch], sc = ec[C + ch], is = em[C + ch];
        e2[ch] = sc; e2[C + ch] = ec[2 * C + ch] - mean * sc; e2[2 * C + ch] = is; e2[3 * C + ch] = -mean * is;
    }
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) st_lds[i * WF_THREADS + tid] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    // W as A fragments of the input gradient: wr[j][f][e] = W[k = 16f + 4q + e][c = 16j + m]
    f32x4 wr[NT][NF];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int e = 0; e < 4; ++e) wr[j][f][e] = p.w[(size_t)(16 * f + 4 * q + e) * CT + c0 + 16 * j + m];
    // weight-gradient accumulators: dacc[jo][jc][e] = dW[k = NO (4q + e) + jo][c = NI n + jc], n = lane & 15
    f32x4 dacc[NO][NI];
#pragma unroll
    for (int jo = 0; jo < NO; ++jo)
#pragma unroll
        for (int jc = 0; jc < NI; ++jc) dacc[jo][jc] = f32x4{0.f, 0.f, 0.f, 0.f};

    const long long nstrips = p.rows / WF_ROWS;
    constexpr int SPB = TWO ? 2 : 4;                       // strips a block works on at a time
    const long long stride = (long long)gridDim.x * SPB;
    const long long first = (long long)blockIdx.x * SPB + (TWO ? (wave >> 1) : wave);
    const int n_mine = first < nstrips ? (int)((nstrips - first + stride - 1) / stride) : 0;
    auto run = [&](auto add_tag) {
        constexpr bool ADD = decltype(add_tag)::value;     // source B: e_add joins the input gradient
        const float *gq = gt + 4 * q, *eq = et + 4 * q;
        const size_t k_step = (size_t)stride * WF_ROWS * K, c_step = (size_t)stride * WF_ROWS * C;
        size_t koff = ((size_t)first * WF_ROWS + m) * K + 4 * q, coff = ((size_t)first * WF_ROWS + m) * C + 4 * q;
        f32x4 gn[NF], yn[NF], xn[NT];                        // the strip being REQUESTED
        f32x4 ea[ADD ? NT : 1];                              // e_add rows of the strip whose epilogue comes next
#pragma unroll
        for (int f = 0; f < NF; ++f) { gn[f] = ldg<TWO>(p.gz + koff + 16 * f); yn[f] = ldg<TWO>(p.y2 + koff + 16 * f); }
#pragma unroll
        for (int j = 0; j < NT; ++j) xn[j] = ldn(src_x + coff + 16 * j);
        if (ADD) {
#pragma unroll
            for (int j = 0; j < NT; ++j) ea[ADD ? j : 0] = ldn(p.e_add + coff + 16 * j);
        }
        // One strip = two MFMA streams of this wave, back to back: NT*L MFMAs of the input gradient, then NO*NI*4 of the weight gradient.
        // The only wave of its SIMD overlaps nothing by itself, so everything else is cut into slots of a few instructions placed BETWEEN
        // the MFMAs (sched_barrier(0) after every slot keeps the compiler from regrouping them — it otherwise hoists every tile's
        // constants to the top, 200 bytes of scratch, or issues a tile's whole epilogue behind the last MFMA):
        //   stream B (input gradient, SL = 16 slots per 16-channel tile ja of the layer input): constants from LDS, a = act(bn(x)) of the
        //     tile, a and the raw x rows into the wave's LDS tiles, then the same float4 of the NEXT strip's x requested into the
        //     register it leaves; gz / y of the next strip are requested in the first slots (g^y was formed before the stream);
        //   stream C (weight gradient): the channel-per-lane operands of k-step t + 1 read from the LDS tiles during k-step t; the
        //     input gradient's epilogue (act', statistics rows, store) of tile je in slots [SL je, SL je + 9].
        constexpr int L = K / 4, NMF = NT * L, SL = NMF / NT, NWG = NO * NI * 4, SLC = NWG / NT, LAT = 6;
        static_assert(SL >= 13 && SLC >= 10 && NWG / 4 >= 2, "slot plan");
        f32x4 acc[NT];
        // The first strip is PEELED (the body is instantiated in front of the loop and inside it): the compiler's s_waitcnt vmcnt counts at
        // the loop header are the minimum over the entry path and the back edge, and on the entry path (the prologue's 16 requests, no
        // stores behind them) fewer operations are younger than a needed register than in steady state (16 requests + 8 stores) — the
        // steady-state iterations then waited for the x requests issued only half a strip earlier.
        auto strip = [&](const bool has_next) {
            // TWO: g^y is formed IN the registers gz arrived in (K = 128 makes gz / y / g^y 96 registers otherwise; with W and dW at 256 the
            // budget is gone); a gz register pair is requested again for the next strip right behind its last MFMA of stream B
            constexpr bool MERGE = TWO;
            f32x4 xg_own[MERGE ? 1 : NF];
            f32x4 (&xg)[MERGE ? NF : NF] = *reinterpret_cast<f32x4 (*)[NF]>(MERGE ? &gn[0] : &xg_own[0]);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const f32x4 tsc = *reinterpret_cast<const f32x4 *>(gq + 16 * f), tac = *reinterpret_cast<const f32x4 *>(gq + K + 16 * f),
                            tbc = *reinterpret_cast<const f32x4 *>(gq + 2 * K + 16 * f);
#pragma unroll
                for (int c = 0; c < 4; ++c) xg[f][c] = __builtin_fmaf(tsc[c], gn[f][c], __builtin_fmaf(yn[f][c], tbc[c], tac[c]));
                *reinterpret_cast<f32x4 *>(tg + m * LDG + 16 * f + 4 * q) = xg[f];
            }
            const size_t coff_cur = coff;
            // (no next strip: the offsets stay and the requests below fetch the last strip again, never used — UNCONDITIONAL requests, so that
            //  the compiler's s_waitcnt vmcnt counts are exact: with the requests under `if (has_next)` it had to assume the path without
            //  them and waited for every outstanding load, the ones just issued included: 6.9 us per strip instead of 3.4 us of MFMAs)
            if (has_next) { koff += k_step; coff += c_step; }
            __builtin_amdgcn_sched_barrier(0);
            // ---- stream B ------------------------------------------------------------------------------------------------------------
            f32x4 esc, ezb, av_t;
            esc = ezb = av_t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NMF; ++i) {
                const int t = i / NT, j = i % NT, f = t >> 2, e = t & 3;
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                if (!(FUSED_ABL & 16)) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[j][f][e], xg[f][e], t == 0 ? zero : acc[j], 0, 0, 0);
                else if (t == 0) acc[j] = xg[0];
                // gz / y of the next strip: both 64-byte halves of a row's 128-byte line in ONE slot (requests a few slots apart fetch the
                // line twice: PMC FETCH_SIZE 1.28x in mlp_wreg.hip's dgrad before it paired them)
                if (i < NF && !(FUSED_ABL & 8)) {
                    const int f2 = (i >> 1) * 2;
                    if (i & 1) { yn[f2] = ldg<TWO>(p.y2 + koff + 16 * f2); yn[f2 + 1] = ldg<TWO>(p.y2 + koff + 16 * (f2 + 1)); }
                    else if (!MERGE) { gn[f2] = ldg<TWO>(p.gz + koff + 16 * f2); gn[f2 + 1] = ldg<TWO>(p.gz + koff + 16 * (f2 + 1)); }
                }
                if (MERGE && !(FUSED_ABL & 8) && (i + 1) % (8 * NT) == 0) {      // the last MFMA that reads g^y block f2 + 1 has been issued
                    const int f2 = (i + 1) / (8 * NT) * 2 - 2;
                    gn[f2] = ldg<TWO>(p.gz + koff + 16 * f2); gn[f2 + 1] = ldg<TWO>(p.gz + koff + 16 * (f2 + 1));
                }
                const int ja = i / SL, sub = i % SL;
                if (sub == 2) { esc = *reinterpret_cast<const f32x4 *>(eq + 16 * ja); ezb = *reinterpret_cast<const f32x4 *>(eq + C + 16 * ja); }
                if (sub == 2 + LAT) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) av_t[c] = __builtin_fmaf(xn[ja][c], esc[c], ezb[c]);
                }
                if (sub == 3 + LAT) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) av_t[c] = __builtin_fmaxf(av_t[c], av_t[c] * my_slope);
                }
                if (sub == 4 + LAT && !(FUSED_ABL & 2)) {
                    *reinterpret_cast<f32x4 *>(ta + m * LDA + 16 * ja + 4 * q) = av_t;
                    *reinterpret_cast<f32x4 *>(tx + m * LDA + 16 * ja + 4 * q) = xn[ja];
                }
                if (sub == 5 + LAT && (ja & 1) && !(FUSED_ABL & 8)) { xn[ja - 1] = ldn(src_x + coff + 16 * (ja - 1)); xn[ja] = ldn(src_x + coff + 16 * ja); }
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- stream C ------------------------------------------------------------------------------------------------------------
            f32x4 gv[2][HO], av[2][HI];
#pragma unroll
            for (int h = 0; h < HO; ++h) gv[0][h] = *reinterpret_cast<const f32x4 *>(tg + (4 * q) * LDG + NO * m + 4 * h);
#pragma unroll
            for (int h = 0; h < HI; ++h) av[0][h] = *reinterpret_cast<const f32x4 *>(ta + (4 * q) * LDA + NI * m + 4 * h);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 eis, enm, xr, r1, r2, ev, vkeep;
            eis = enm = xr = r1 = r2 = ev = vkeep = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < NWG; ++u) {
                const int t = u / (NO * NI), r = u % (NO * NI), jo = r / NI, jc = r % NI;
                if (!(FUSED_ABL & 1)) dacc[jo][jc] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[t & 1][jo >> 2][jo & 3], av[t & 1][jc >> 2][jc & 3], dacc[jo][jc], 0, 0, 0);
                else if (r == 0) dacc[0][0] += gv[t & 1][0] + av[t & 1][0];
                if (r == 1 && t < 3 && !(FUSED_ABL & 2)) {                                            // operands of k-step t + 1
#pragma unroll
                    for (int h = 0; h < HO; ++h) gv[(t + 1) & 1][h] = *reinterpret_cast<const f32x4 *>(tg + (4 * q + t + 1) * LDG + NO * m + 4 * h);
#pragma unroll
                    for (int h = 0; h < HI; ++h) av[(t + 1) & 1][h] = *reinterpret_cast<const f32x4 *>(ta + (4 * q + t + 1) * LDA + NI * m + 4 * h);
                }
                const int je = u / SLC, sub = u % SLC;
                if (sub == 0) {
                    esc = *reinterpret_cast<const f32x4 *>(eq + 16 * je); ezb = *reinterpret_cast<const f32x4 *>(eq + C + 16 * je);
                    xr = *reinterpret_cast<const f32x4 *>(tx + m * LDA + 16 * je + 4 * q);
                }
                if (sub == 2) {
                    eis = *reinterpret_cast<const f32x4 *>(eq + 2 * C + 16 * je); enm = *reinterpret_cast<const f32x4 *>(eq + 3 * C + 16 * je);
                    r1 = st_lds[je * WF_THREADS + tid]; r2 = st_lds[(NT + je) * WF_THREADS + tid];
                }
                if (sub == LAT) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float z = __builtin_fmaf(xr[c], esc[c], ezb[c]);
                        const float a = ADD ? acc[je][c] + ea[ADD ? je : 0][c] : acc[je][c];
                        ev[c] = z > 0.f ? a : a * my_slope;
                    }
                }
                if (sub == LAT + 1) {                                             // stores too: one 128-byte line per row at a time
                    if ((je & 1) && !(FUSED_ABL & 4)) { stn(dst_g + coff_cur + 16 * (je - 1), vkeep); stn(dst_g + coff_cur + 16 * je, ev); }
                    else vkeep = ev;
                    r1 += ev; st_lds[je * WF_THREADS + tid] = r1;
                    // e_add of the NEXT strip into the registers this tile pair has just left (both halves of a 128-byte line together)
                    if (ADD && (je & 1) && !(FUSED_ABL & 8)) { ea[ADD ? je - 1 : 0] = ldn(p.e_add + coff + 16 * (je - 1)); ea[ADD ? je : 0] = ldn(p.e_add + coff + 16 * je); }
                }
                if (sub == LAT + 3) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) r2[c] = __builtin_fmaf(ev[c], __builtin_fmaf(xr[c], eis[c], enm[c]), r2[c]);
                    st_lds[(NT + je) * WF_THREADS + tid] = r2;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        strip(n_mine > 1);
        for (int k = 1; k < n_mine; ++k) strip(k + 1 < n_mine);
    };
    if (n_mine > 0) {
        if (TWO && half) run(std::true_type{});
        else run(std::false_type{});
    }
    __syncthreads();
    double *my_sums = half ? p.sums_b : p.sums;
    if (my_sums) {      // lanes with the same q hold the same channels: sum over the 16 rows (m), one fp64 atomic per wave, channel and moment
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double a = (double)st_lds[j * WF_THREADS + tid][e], b = (double)st_lds[(NT + j) * WF_THREADS + tid][e];
#pragma unroll
                for (int off = 8; off >= 1; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
                if (m == 0) {
                    double *rep = my_sums + (size_t)((blockIdx.x * 4 + wave) % REP) * 2 * C;
                    atomicAdd(rep + 16 * j + 4 * q + e, a); atomicAdd(rep + C + 16 * j + 4 * q + e, b);
                }
            }
    }
    __syncthreads();
    // ---- the four waves add their weight gradients through LDS in a fixed order (over the statistics rows, which are consumed) ----
    float *red = reinterpret_cast<float *>(st_lds);        // [K][CT]; TWO: runs on into the (now dead) strip tiles behind the statistics rows
    static_assert(K * CT <= 2 * NT * WF_THREADS * 4 + (TWO ? 4 * WF_ROWS * (K + 4 + 2 * (C + 4)) : 0), "reduction buffer fits");
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int jo = 0; jo < NO; ++jo)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int h = 0; h < HI; ++h) {
                        float *dst = red + (size_t)(NO * (4 * q + e) + jo) * CT + c0 + NI * m + 4 * h;
                        f32x4 v = {dacc[jo][4 * h][e], dacc[jo][4 * h + 1][e], dacc[jo][4 * h + 2][e], dacc[jo][4 * h + 3][e]};
                        if (w >= (TWO ? 2 : 1)) v += *reinterpret_cast<const f32x4 *>(dst);      // (TWO: waves 0, 1 are the first writers of their column halves)
                        *reinterpret_cast<f32x4 *>(dst) = v;
                    }
        }
        __syncthreads();
    }
    float *out = p.dw_partial + (size_t)blockIdx.x * K * CT;
    for (int i = tid; i < K * CT / 4; i += WF_THREADS)
        *reinterpret_cast<f32x4 *>(out + 4 * i) = *reinterpret_cast<const f32x4 *>(red + 4 * i);
}

template <int K, int C, bool TWO>
size_t fused_lds_bytes() {
    return ((size_t)3 * K + (TWO ? 8 : 4) * C + (size_t)2 * (C / 16) * WF_THREADS * 4 + (size_t)4 * WF_ROWS * (K + 4 + 2 * (C + 4))) * sizeof(float);
}

template <int K, int C, bool TWO = false>
int launch_fused(const WregFusedP &p, unsigned grid, hipStream_t st) {
    const size_t bytes = fused_lds_bytes<K, C, TWO>();
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(wreg_bwd_fused_kernel<K, C, TWO>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    i2p_ktime_begin(st);
    hipLaunchKernelGGL((wreg_bwd_fused_kernel<K, C, TWO>), dim3(grid), dim3(WF_THREADS), bytes, st, p);
    i2p_ktime_end(st);
    I2P_RETURN_LAUNCH_STATUS();
}

}  // namespace

// k = output channels of the layer (gz / y), c = its input channels (x, gz_in).  Shapes: the HBM-bound layers (k = 64).
bool i2p_wreg_bwd_fused_ok(long long rows, int k, int c) {
    static const char *off = getenv("I2P_NO_FUSED_BWD");
    static const char *nw = getenv("I2P_NO_WREG");
    if ((off && off[0] == '1') || (nw && nw[0] == '1')) return false;
    return rows >= 32768 && (rows % WF_ROWS) == 0 && k == 64 && (c == 64 || c == 128) &&
           (unsigned long long)rows * 128ull * 4ull < (1ull << 40);
}

// the two-source layer 64 + 64 -> 128 in one pass (TWO instantiation); I2P_NO_FUSED_BWD2=1: wreg_dgrad_kernel<128,128,true> + wreg_wgrad_kernel
bool i2p_wreg_bwd_fused2_ok(long long rows, int k, int c, int split) {
    const char *off = getenv("I2P_NO_FUSED_BWD2");
    static const char *off1 = getenv("I2P_NO_FUSED_BWD");
    static const char *nw = getenv("I2P_NO_WREG");
    if ((off && off[0] == '1') || (off1 && off1[0] == '1') || (nw && nw[0] == '1')) return false;
    return rows >= 32768 && (rows % WF_ROWS) == 0 && k == 128 && c == 128 && split == 64 &&
           (unsigned long long)rows * 128ull * 4ull < (1ull << 40);
}

// writes gz_in, adds the BN-backward statistics of the layer in front into `sums`, rows 6, 7 of bn_out and dw_partial[grid][k*c]
// (grid = 256: the caller reduces the slabs)
int i2p_wreg_bwd_fused(long long rows, int k, int c, const float *gz, const float *y2, const double *g_dsums, const float *g_oc,
                       const float *g_omi, long long g_rows, const float *w, float *gz_in, const float *ex, const float *e_coef,
                       const float *e_mi, float e_slope, double *sums, float *bn_out, float *dw_partial, unsigned grid, void *stream) {
    if (!i2p_wreg_bwd_fused_ok(rows, k, c) || !gz || !y2 || !g_dsums || !g_oc || !g_omi || !w || !gz_in || !ex || !e_coef || !e_mi ||
        !dw_partial || grid == 0 || !(e_slope >= 0.f && e_slope <= 1.f))
        return I2P_ERR_BAD_ARG;
    WregFusedP p{};
    p.rows = rows; p.gz = gz; p.y2 = y2; p.g_dsums = g_dsums; p.g_oc = g_oc; p.g_omi = g_omi; p.g_rows = g_rows; p.w = w;
    p.gz_in = gz_in; p.ex = ex; p.e_coef = e_coef; p.e_mi = e_mi; p.e_slope = e_slope; p.sums = sums; p.bn_out = bn_out;
    p.dw_partial = dw_partial;
    const long long nstrips = rows / WF_ROWS;
    long long g = (nstrips + 3) / 4;
    if (g > (long long)grid) g = grid;
    hipStream_t st = (hipStream_t)stream;
    if (c == 128) return launch_fused<64, 128>(p, (unsigned)g, st);
    return launch_fused<64, 64>(p, (unsigned)g, st);
}

// two-source form: x = [xa | xb] with 64 channels each (own BN constants, slopes, destinations, statistics), e_add joins xb's gradient;
// w [128][128], dw_partial [grid][128*128]
int i2p_wreg_bwd_fused2(long long rows, const float *gz, const float *y2, const double *g_dsums, const float *g_oc, const float *g_omi,
                        long long g_rows, const float *w, float *gz_in_a, const float *xa, const float *coef_a, const float *mi_a,
                        float slope_a, double *sums_a, float *gz_in_b, const float *xb, const float *coef_b, const float *mi_b,
                        float slope_b, double *sums_b, const float *e_add, float *bn_out, float *dw_partial, unsigned grid, void *stream) {
    if (!i2p_wreg_bwd_fused2_ok(rows, 128, 128, 64) || !gz || !y2 || !g_dsums || !g_oc || !g_omi || !w || !gz_in_a || !xa || !coef_a ||
        !mi_a || !gz_in_b || !xb || !coef_b || !mi_b || !e_add || !sums_a || !sums_b || !dw_partial || grid == 0 ||
        !(slope_a >= 0.f && slope_a <= 1.f) || !(slope_b >= 0.f && slope_b <= 1.f))
        return I2P_ERR_BAD_ARG;
    WregFusedP p{};
    p.rows = rows; p.gz = gz; p.y2 = y2; p.g_dsums = g_dsums; p.g_oc = g_oc; p.g_omi = g_omi; p.g_rows = g_rows; p.w = w;
    p.gz_in = gz_in_a; p.ex = xa; p.e_coef = coef_a; p.e_mi = mi_a; p.e_slope = slope_a; p.sums = sums_a;
    p.gz_in_b = gz_in_b; p.exb = xb; p.e_coef_b = coef_b; p.e_mi_b = mi_b; p.e_slope_b = slope_b; p.sums_b = sums_b; p.e_add = e_add;
    p.bn_out = bn_out; p.dw_partial = dw_partial;
    const long long nstrips = rows / WF_ROWS;
    long long g = (nstrips + 1) / 2;
    if (g > (long long)grid) g = grid;
    return launch_fused<128, 64, true>(p, (unsigned)g, (hipStream_t)stream);
}
