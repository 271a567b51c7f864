// Backward of the PAIR layer (first cost-volume layer: y[b,n,k,:] = W (f[b,n,:] .* g[b,k,:]) + bias_n[b,n,:] + bias_k[b,k,:], BN behind;
// reference: PPBackbone_center.py:383-433, factored as in DESIGN.md section 1) in ONE pass over gz / y (round 6).
// mlp_wreg.hip runs it as two kernels (wreg_pair_dgrad_kernel: dP = g^y W -> d_f, d_g; wreg_pair_wgrad_kernel: dW, d_bias_n, d_bias_k),
// each streaming gz [rows,128] and y [rows,128]: 0.87 GB twice at batch 8 (384 + 344 us).  K = 128 output channels make W (A fragments of
// dP) and the dW accumulators 256 + 256 registers, so — as in wreg_bwd_fused_kernel<128,64,TWO> (mlp_wreg_fused.hip) — the input
// channels are split over TWO waves: a wave owns (sample b, 16-pixel tile, chunk of points, half of the 128 input channels): W columns
// and dW columns of its 64 channels (128 + 128 registers), its half of f / g, its half of d_f / d_g; it forms g^y for all 128 output
// channels itself, in the registers gz arrived in.  The two waves of a task are neighbours in the block and read the same gz / y lines
// (default cache policy: the second read is an L2 hit); d_bias_n / d_bias_k (functions of g^y alone) are split between them by output
// channel (half h takes channels [64h, 64h + 64)).
// Per strip (fixed point n, 16 pixels): stream B = 128 MFMAs of dP^T[c][pixel] = sum_k W[k][c] g^y[pixel][k]; stream C = 128 MFMAs of
// dW[k][c] += sum_pixel g^y[pixel][k] x'[pixel][c], x' = f[n][c] g[pixel][c], operands transposed through wave-private LDS tiles exactly
// as in mlp_wreg_fused.hip; epilogue of a dP tile: d_g[pixel][c] += dP f[n][c] (lane-private LDS rows over the chunk's points),
// d_f[n][c] = sum over the 16 pixels of dP g[pixel][c] (DPP row sum, one slab per pixel tile).  Every cross-wave sum goes through the
// slabs of i2p_wreg_pair_bwd_scratch, reduced by the caller in a fixed order (bit-reproducible), as for the two-kernel form.
#include "common.h"
#include <type_traits>

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int PF_THREADS = 256, PF_ROWS = 16, REP = I2P_BN_REPLICAS;

struct PairFusedP {
    int B, N, M, KT, NCH, NL;
    const float *gz, *y2;        // [B*N*M, K]
    const double *g_dsums; const float *g_oc, *g_omi; long long g_rows;
    const float *f, *g, *w;      // [B,N,CT], [B,M,CT], [K][CT]
    float *dw_partial;           // [grid][K*CT]
    float *s_df, *s_dbn;         // [KT][B*N*CT], [KT][B*N*K]
    float *s_dg, *s_dbk;         // [NCH][B*M*CT], [NCH][B*M*K]
};

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(b, b, CTRL, 0xF, 0xF, false));
}
// sum over the 16 lanes of a DPP row (every lane ends with the row's sum)
__device__ __forceinline__ void row_sum(f32x4 &v) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float a = v[c];
        a += dpp_f32<0xB1>(a); a += dpp_f32<0x4E>(a); a += dpp_f32<0x141>(a); a += dpp_f32<0x140>(a);
        v[c] = a;
    }
}

template <int K, int C>
__global__ __launch_bounds__(PF_THREADS, 1) void wreg_pair_bwd_fused_kernel(PairFusedP p) {
    constexpr int CT = 2 * C;
    constexpr int NT = C / 16, NF = K / 16, NO = K / 16, NI = C / 16, HO = NO / 4, HI = NI / 4, NFH = NF / 2;
    constexpr int LDG = K + 4, LDA = C + 4;
    constexpr int NPRIV = NT + NFH;                        // lane-private float4 rows: d_g sums of the lane's pixel, its half of d_bias_k
    static_assert(HO >= 1 && HI >= 1 && (NF & 1) == 0, "K multiple of 64 (and of 32 per half), C multiple of 64");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *gt = smem;                                      // [3][K]  sc, Ac, Bc
    f32x4 *priv = reinterpret_cast<f32x4 *>(gt + 3 * K);   // [NPRIV][256]
    float *tiles = reinterpret_cast<float *>(priv + NPRIV * PF_THREADS);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int half = wave & 1, c0 = half * C;
    float *tg = tiles + (size_t)wave * PF_ROWS * (LDG + LDA);
    float *ta = tg + PF_ROWS * LDG;

    for (int ch = tid; ch < K; ch += PF_THREADS) {
        double sd = 0.0, sx = 0.0;
#pragma unroll 8
        for (int rp = 0; rp < REP; ++rp) { sd += p.g_dsums[(size_t)rp * 2 * K + ch]; sx += p.g_dsums[(size_t)rp * 2 * K + K + ch]; }
        const float m1 = (float)(sd / (double)p.g_rows), m2 = (float)(sx / (double)p.g_rows);
        const float sc = p.g_oc[K + ch], mu = p.g_omi[ch], is = p.g_omi[K + ch];
        const float bc = -sc * is * m2;
        gt[ch] = sc; gt[K + ch] = -sc * m1 - bc * mu; gt[2 * K + ch] = bc;
    }
    __syncthreads();

    // W as A fragments of dP: wr[j][f][e] = W[k = 16f + 4q + e][c = c0 + 16j + m]
    f32x4 wr[NT][NF];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int e = 0; e < 4; ++e) wr[j][f][e] = p.w[(size_t)(16 * f + 4 * q + e) * CT + c0 + 16 * j + m];
    // weight-gradient accumulators: dacc[jo][jc][e] = dW[k = NO (4q + e) + jo][c = c0 + NI m + jc]
    f32x4 dacc[NO][NI];
#pragma unroll
    for (int jo = 0; jo < NO; ++jo)
#pragma unroll
        for (int jc = 0; jc < NI; ++jc) dacc[jo][jc] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float *gq = gt + 4 * q;
    const int ntasks = p.B * p.KT * p.NCH;
    // (HALF as a compile-time constant: `half ? gn[NFH + ff] : gn[ff]` with a run-time half became an indexed access and moved the whole
    //  g^y register set into scratch — every access a scratch round trip behind s_waitcnt vmcnt(0): 1065 us)
    auto run = [&](auto half_tag) {
    constexpr int HALF = decltype(half_tag)::value;
    for (int task = blockIdx.x * 2 + (wave >> 1); task < ntasks; task += gridDim.x * 2) {
        const int nc = task % p.NCH, kt = (task / p.NCH) % p.KT, b = task / (p.NCH * p.KT);
        const int k0 = kt * PF_ROWS, n_begin = nc * p.NL, n_end = n_begin + p.NL < p.N ? n_begin + p.NL : p.N;
        const int ns = n_end - n_begin;
        if (ns <= 0) continue;
        const int kpix = k0 + m;
        const float vm = kpix < p.M ? 1.f : 0.f;
        const int kc = kpix < p.M ? kpix : p.M - 1;
        f32x4 gpix[NT];                                    // g of this lane's pixel, its channels c0 + 16j + 4q ..
#pragma unroll
        for (int j = 0; j < NT; ++j) gpix[j] = *reinterpret_cast<const f32x4 *>(p.g + ((size_t)b * p.M + kc) * CT + c0 + 16 * j + 4 * q);
#pragma unroll
        for (int r = 0; r < NPRIV; ++r) priv[r * PF_THREADS + tid] = f32x4{0.f, 0.f, 0.f, 0.f};
        size_t koff = (((size_t)b * p.N + n_begin) * p.M + kc) * K + 4 * q;              // strip being REQUESTED (gz / y)
        const size_t k_step = (size_t)p.M * K;
        size_t foff = ((size_t)b * p.N + n_begin) * CT + c0 + 4 * q;                      // f row being REQUESTED
        int kreq = 0, freq = 0;
        f32x4 gn[NF], yn[NF], fcur[NT];
#pragma unroll
        for (int f = 0; f < NF; ++f) { gn[f] = *reinterpret_cast<const f32x4 *>(p.gz + koff + 16 * f); yn[f] = *reinterpret_cast<const f32x4 *>(p.y2 + koff + 16 * f); }
#pragma unroll
        for (int j = 0; j < NT; ++j) fcur[j] = *reinterpret_cast<const f32x4 *>(p.f + foff + 16 * j);
        if (freq + 1 < ns) { foff += CT; ++freq; }           // foff: the NEXT strip's f row (L2-hot: requested in stream C, used in the next stream B)

        constexpr int L = K / 4, NMF = NT * L, SL = NMF / NT, NWG = NO * NI * 4, SLC = NWG / NT, LAT = 6;
        static_assert(SL >= 13 && SLC >= 12, "slot plan");
        f32x4 acc[NT];
        size_t row_n = (size_t)b * p.N + n_begin;
        // (the first strip of a task is PEELED — the body stands in front of the loop and inside it: the compiler's s_waitcnt vmcnt counts at a
        //  loop header are the minimum over the entry path and the back edge, and the entry path has other operations in flight; see
        //  mlp_wreg_fused.hip)
        auto strip = [&]() {
            // ---- g^y in place (BN backward of the layer behind formed on load; zero on the rows past M) -----------------------------------
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const f32x4 tsc = *reinterpret_cast<const f32x4 *>(gq + 16 * f), tac = *reinterpret_cast<const f32x4 *>(gq + K + 16 * f),
                            tbc = *reinterpret_cast<const f32x4 *>(gq + 2 * K + 16 * f);
#pragma unroll
                for (int c = 0; c < 4; ++c) gn[f][c] = vm * __builtin_fmaf(tsc[c], gn[f][c], __builtin_fmaf(yn[f][c], tbc[c], tac[c]));
                *reinterpret_cast<f32x4 *>(tg + m * LDG + 16 * f + 4 * q) = gn[f];
            }
            // (no next strip: the offsets stay and the requests below fetch the last strip again, never used — UNCONDITIONAL requests
            //  keep the compiler's s_waitcnt vmcnt counts exact, see mlp_wreg_fused.hip)
            if (kreq + 1 < ns) { koff += k_step; ++kreq; }
            __builtin_amdgcn_sched_barrier(0);
            // ---- stream B: dP ----------------------------------------------------------------------------------------------------------
            f32x4 av_t = {0.f, 0.f, 0.f, 0.f}, bsum = av_t, bkr = av_t;
#pragma unroll
            for (int i = 0; i < NMF; ++i) {
                const int t = i / NT, j = i % NT, f = t >> 2, e = t & 3;
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[j][f][e], gn[f][e], t == 0 ? zero : acc[j], 0, 0, 0);
                if (i < NF && (i & 1)) {
                    const int f2 = (i >> 1) * 2;
                    yn[f2] = *reinterpret_cast<const f32x4 *>(p.y2 + koff + 16 * f2); yn[f2 + 1] = *reinterpret_cast<const f32x4 *>(p.y2 + koff + 16 * (f2 + 1));
                }
                if ((i + 1) % (8 * NT) == 0) {               // the last MFMA that reads g^y block f2 + 1 has been issued
                    const int f2 = (i + 1) / (8 * NT) * 2 - 2;
                    gn[f2] = *reinterpret_cast<const f32x4 *>(p.gz + koff + 16 * f2); gn[f2 + 1] = *reinterpret_cast<const f32x4 *>(p.gz + koff + 16 * (f2 + 1));
                }
                const int ja = i / SL, sub = i % SL;
                if (sub == 2) av_t = fcur[ja] * gpix[ja];    // x' of this lane's pixel, channels c0 + 16 ja + 4q ..
                if (sub == 4) *reinterpret_cast<f32x4 *>(ta + m * LDA + 16 * ja + 4 * q) = av_t;
                // this wave's half of d_bias_n (sum over the strip's pixels) and of d_bias_k (sum over the chunk's points): channel blocks
                // f = NFH HALF .. NFH HALF + NFH - 1, each in the MFMA slots of its own g^y block (before that block's registers are
                // requested again): slots 16f + 3 .. 16f + 10
                {
                    const int fb = i / 16, sb = i % 16, ff = fb - HALF * NFH;
                    if (ff >= 0 && ff < NFH) {
                        if (sb == 3) { bsum = gn[fb]; bkr = priv[(NT + ff) * PF_THREADS + tid]; }
                        if (sb == 5) { bkr += bsum; priv[(NT + ff) * PF_THREADS + tid] = bkr; }
                        if (sb >= 6 && sb <= 9) {
                            float a = bsum[sb - 6];
                            a += dpp_f32<0xB1>(a); a += dpp_f32<0x4E>(a); a += dpp_f32<0x141>(a); a += dpp_f32<0x140>(a);
                            bsum[sb - 6] = a;
                        }
                        if (sb == 10 && m == 0) *reinterpret_cast<f32x4 *>(p.s_dbn + (size_t)kt * p.B * p.N * K + row_n * K + 16 * fb + 4 * q) = bsum;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- stream C: dW; epilogue of the dP tiles -----------------------------------------------------------------------------------
            f32x4 gv[2][HO], av[2][HI];
#pragma unroll
            for (int h = 0; h < HO; ++h) gv[0][h] = *reinterpret_cast<const f32x4 *>(tg + (4 * q) * LDG + NO * m + 4 * h);
#pragma unroll
            for (int h = 0; h < HI; ++h) av[0][h] = *reinterpret_cast<const f32x4 *>(ta + (4 * q) * LDA + NI * m + 4 * h);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 dgr = {0.f, 0.f, 0.f, 0.f}, tprod = dgr;
#pragma unroll
            for (int u = 0; u < NWG; ++u) {
                const int t = u / (NO * NI), r = u % (NO * NI), jo = r / NI, jc = r % NI;
                dacc[jo][jc] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[t & 1][jo >> 2][jo & 3], av[t & 1][jc >> 2][jc & 3], dacc[jo][jc], 0, 0, 0);
                if (r == 1 && t < 3) {                                                            // operands of k-step t + 1
#pragma unroll
                    for (int h = 0; h < HO; ++h) gv[(t + 1) & 1][h] = *reinterpret_cast<const f32x4 *>(tg + (4 * q + t + 1) * LDG + NO * m + 4 * h);
#pragma unroll
                    for (int h = 0; h < HI; ++h) av[(t + 1) & 1][h] = *reinterpret_cast<const f32x4 *>(ta + (4 * q + t + 1) * LDA + NI * m + 4 * h);
                }
                const int je = u / SLC, sub = u % SLC;
                if (sub == 0) dgr = priv[je * PF_THREADS + tid];
                if (sub == LAT) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) dgr[c] = __builtin_fmaf(acc[je][c], fcur[je][c], dgr[c]);
                    priv[je * PF_THREADS + tid] = dgr;
                    tprod = acc[je] * gpix[je];
                }
                if (sub == LAT + 1) row_sum(tprod);
                if (sub == LAT + 2 && m == 0)
                    *reinterpret_cast<f32x4 *>(p.s_df + (size_t)kt * p.B * p.N * CT + row_n * CT + c0 + 16 * je + 4 * q) = tprod;
                if (sub == LAT + 3) fcur[je] = *reinterpret_cast<const f32x4 *>(p.f + foff + 16 * je);     // f of the next strip into the registers this tile has left
                __builtin_amdgcn_sched_barrier(0);
            }
            if (freq + 1 < ns) { foff += CT; ++freq; }
            ++row_n;
        };
        strip();
        for (int sidx = 1; sidx < ns; ++sidx) strip();
        // ---- end of the task: this lane's pixel sums ---------------------------------------------------------------------------------
        if (kpix < p.M) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
                *reinterpret_cast<f32x4 *>(p.s_dg + (size_t)nc * p.B * p.M * CT + ((size_t)b * p.M + kpix) * CT + c0 + 16 * j + 4 * q) = priv[j * PF_THREADS + tid];
#pragma unroll
            for (int ff = 0; ff < NFH; ++ff)
                *reinterpret_cast<f32x4 *>(p.s_dbk + (size_t)nc * p.B * p.M * K + ((size_t)b * p.M + kpix) * K + 16 * (HALF * NFH + ff) + 4 * q) =
                    priv[(NT + ff) * PF_THREADS + tid];
        }
    }
    };
    if (half) run(std::integral_constant<int, 1>{}); else run(std::integral_constant<int, 0>{});
    __syncthreads();
    // ---- the four waves add their weight gradients through LDS in a fixed order: waves 0, 1 write their column halves, 2, 3 add ----------
    float *red = reinterpret_cast<float *>(priv);
    static_assert((size_t)K * CT <= (size_t)NPRIV * PF_THREADS * 4 + (size_t)4 * PF_ROWS * (K + 4 + C + 4), "reduction buffer fits");
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
                        if (w >= 2) v += *reinterpret_cast<const f32x4 *>(dst);
                        *reinterpret_cast<f32x4 *>(dst) = v;
                    }
        }
        __syncthreads();
    }
    float *out = p.dw_partial + (size_t)blockIdx.x * K * CT;
    for (int i = tid; i < K * CT / 4; i += PF_THREADS)
        *reinterpret_cast<f32x4 *>(out + 4 * i) = *reinterpret_cast<const f32x4 *>(red + 4 * i);
}

}  // namespace

// I2P_NO_PAIR_FUSED=1: the two-kernel form (wreg_pair_dgrad_kernel + wreg_pair_wgrad_kernel, csrc/mlp_wreg.hip)
bool i2p_wreg_pair_bwd_fused_ok(void) {
    const char *off = getenv("I2P_NO_PAIR_FUSED");
    return !(off && off[0] == '1');
}

// same contract as i2p_wreg_pair_bwd (shapes of i2p_wreg_pair_bwd_ok: cin = cout = 128), same scratch layout and slab geometry
int i2p_wreg_pair_bwd_fused(int B, int N, int M, int KT, int NCH, int NL, const float *gz, const float *y2, const double *g_dsums,
                            const float *g_oc, const float *g_omi, const float *f, const float *g, const float *w, float *dw_partial,
                            float *s_df, float *s_dbn, float *s_dg, float *s_dbk, void *stream) {
    PairFusedP p;
    p.B = B; p.N = N; p.M = M; p.KT = KT; p.NCH = NCH; p.NL = NL;
    p.gz = gz; p.y2 = y2; p.g_dsums = g_dsums; p.g_oc = g_oc; p.g_omi = g_omi; p.g_rows = (long long)B * N * M;
    p.f = f; p.g = g; p.w = w; p.dw_partial = dw_partial; p.s_df = s_df; p.s_dbn = s_dbn; p.s_dg = s_dg; p.s_dbk = s_dbk;
    constexpr int K = 128, C = 64;
    const size_t bytes = ((size_t)3 * K + (size_t)(C / 16 + K / 32) * PF_THREADS * 4 + (size_t)4 * PF_ROWS * (K + 4 + C + 4)) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(wreg_pair_bwd_fused_kernel<K, C>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((wreg_pair_bwd_fused_kernel<K, C>), dim3(256), dim3(PF_THREADS), bytes, (hipStream_t)stream, p);
    I2P_RETURN_LAUNCH_STATUS();
}
