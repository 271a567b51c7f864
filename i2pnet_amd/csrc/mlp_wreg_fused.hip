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
    float *dw_partial;           // [grid][K*C]
};

__device__ __forceinline__ f32x4 ldn(const float *ptr) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt *>(ptr)); }
__device__ __forceinline__ void stn(float *ptr, const f32x4 &v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4_nt *>(ptr)); }

template <int K, int C>
__global__ __launch_bounds__(WF_THREADS, 1) void wreg_bwd_fused_kernel(WregFusedP p) {
    constexpr int NT = C / 16;             // input-gradient tiles (16 result channels each)
    constexpr int NF = K / 16;             // float4 of gz / y per lane and row
    constexpr int NO = K / 16, NI = C / 16;// weight-gradient tiles = channels per lane
    constexpr int HO = NO / 4, HI = NI / 4;
    constexpr int LDG = K + 4, LDA = C + 4;// LDS tile rows (+4: consecutive rows start 4 banks apart)
    static_assert(HO >= 1 && HI >= 1, "K, C multiples of 64");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *gt = smem;                                      // [3][K]  sc, Ac, Bc
    float *et = gt + 3 * K;                                // [4][C]  e_sc, e_zb, e_is, e_nm
    f32x4 *st_lds = reinterpret_cast<f32x4 *>(et + 4 * C); // [2 NT][256] lane-private statistics rows
    float *tiles = reinterpret_cast<float *>(st_lds + 2 * NT * WF_THREADS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, q = lane >> 4;
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
    for (int ch = tid; ch < C; ch += WF_THREADS) {
        const float mean = p.e_coef[ch], sc = p.e_coef[C + ch], is = p.e_mi[C + ch];
        et[ch] = sc; et[C + ch] = p.e_coef[2 * C + ch] - mean * sc; et[2 * C + ch] = is; et[3 * C + ch] = -mean * is;
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
            for (int e = 0; e < 4; ++e) wr[j][f][e] = p.w[(size_t)(16 * f + 4 * q + e) * C + 16 * j + m];
    // weight-gradient accumulators: dacc[jo][jc][e] = dW[k = NO (4q + e) + jo][c = NI n + jc], n = lane & 15
    f32x4 dacc[NO][NI];
#pragma unroll
    for (int jo = 0; jo < NO; ++jo)
#pragma unroll
        for (int jc = 0; jc < NI; ++jc) dacc[jo][jc] = f32x4{0.f, 0.f, 0.f, 0.f};

    const long long nstrips = p.rows / WF_ROWS;
    const long long stride = (long long)gridDim.x * 4;
    const long long first = (long long)blockIdx.x * 4 + wave;
    const int n_mine = first < nstrips ? (int)((nstrips - first + stride - 1) / stride) : 0;
    if (n_mine > 0) {
        const float *gq = gt + 4 * q, *eq = et + 4 * q;
        const size_t k_step = (size_t)stride * WF_ROWS * K, c_step = (size_t)stride * WF_ROWS * C;
        size_t koff = ((size_t)first * WF_ROWS + m) * K + 4 * q, coff = ((size_t)first * WF_ROWS + m) * C + 4 * q;
        f32x4 gn[NF], yn[NF], xn[NT];                        // the strip being REQUESTED
#pragma unroll
        for (int f = 0; f < NF; ++f) { gn[f] = ldn(p.gz + koff + 16 * f); yn[f] = ldn(p.y2 + koff + 16 * f); }
#pragma unroll
        for (int j = 0; j < NT; ++j) xn[j] = ldn(p.ex + coff + 16 * j);
        for (int k = 0; k < n_mine; ++k) {
            f32x4 xg[NF];
            // ---- this strip's operands out of the request registers; g^y formed; the next strip requested ------------------------
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const f32x4 tsc = *reinterpret_cast<const f32x4 *>(gq + 16 * f), tac = *reinterpret_cast<const f32x4 *>(gq + K + 16 * f),
                            tbc = *reinterpret_cast<const f32x4 *>(gq + 2 * K + 16 * f);
#pragma unroll
                for (int c = 0; c < 4; ++c) xg[f][c] = __builtin_fmaf(tsc[c], gn[f][c], __builtin_fmaf(yn[f][c], tbc[c], tac[c]));
            }
            // a = act(bn(x)) and the raw x rows into the wave's LDS tiles BEFORE the request registers are reused
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const f32x4 esc = *reinterpret_cast<const f32x4 *>(eq + 16 * j), ezb = *reinterpret_cast<const f32x4 *>(eq + C + 16 * j);
                f32x4 a;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float z = __builtin_fmaf(xn[j][c], esc[c], ezb[c]);
                    a[c] = __builtin_fmaxf(z, z * p.e_slope);
                }
                *reinterpret_cast<f32x4 *>(ta + m * LDA + 16 * j + 4 * q) = a;
                *reinterpret_cast<f32x4 *>(tx + m * LDA + 16 * j + 4 * q) = xn[j];
                __builtin_amdgcn_sched_barrier(0);           // (keeps the fully unrolled loops from hoisting every tile's constants at once: registers)
            }
            const size_t coff_cur = coff;
            if (k + 1 < n_mine) {
                koff += k_step; coff += c_step;
#pragma unroll
                for (int f = 0; f < NF; ++f) { gn[f] = ldn(p.gz + koff + 16 * f); yn[f] = ldn(p.y2 + koff + 16 * f); }
#pragma unroll
                for (int j = 0; j < NT; ++j) xn[j] = ldn(p.ex + coff + 16 * j);
            }
            // ---- g^y into the wave's LDS tile (row-per-lane layout) ----------------------------------------------------------------------
#pragma unroll
            for (int f = 0; f < NF; ++f) *reinterpret_cast<f32x4 *>(tg + m * LDG + 16 * f + 4 * q) = xg[f];
            // ---- input gradient ----------------------------------------------------------------------------------------------------
            f32x4 acc[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[j][f][e], xg[f][e], acc[j], 0, 0, 0);
            // ---- weight gradient: the tiles back in the channel-per-lane layout (same wave: its LDS queue keeps write -> read order) ----
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 gv[HO], av[HI];
#pragma unroll
                for (int h = 0; h < HO; ++h) gv[h] = *reinterpret_cast<const f32x4 *>(tg + (4 * q + t) * LDG + NO * m + 4 * h);
#pragma unroll
                for (int h = 0; h < HI; ++h) av[h] = *reinterpret_cast<const f32x4 *>(ta + (4 * q + t) * LDA + NI * m + 4 * h);
#pragma unroll
                for (int jo = 0; jo < NO; ++jo)
#pragma unroll
                    for (int jc = 0; jc < NI; ++jc)
                        dacc[jo][jc] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[jo >> 2][jo & 3], av[jc >> 2][jc & 3], dacc[jo][jc], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- epilogue of the input gradient: act', BN-backward statistics of the layer in front, store -------------------------------
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const f32x4 esc = *reinterpret_cast<const f32x4 *>(eq + 16 * j), ezb = *reinterpret_cast<const f32x4 *>(eq + C + 16 * j),
                            eis = *reinterpret_cast<const f32x4 *>(eq + 2 * C + 16 * j), enm = *reinterpret_cast<const f32x4 *>(eq + 3 * C + 16 * j);
                const f32x4 xr = *reinterpret_cast<const f32x4 *>(tx + m * LDA + 16 * j + 4 * q);
                f32x4 v, r1 = st_lds[j * WF_THREADS + tid], r2 = st_lds[(NT + j) * WF_THREADS + tid];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float z = __builtin_fmaf(xr[c], esc[c], ezb[c]);
                    v[c] = z > 0.f ? acc[j][c] : acc[j][c] * p.e_slope;
                    r1[c] += v[c];
                    r2[c] = __builtin_fmaf(v[c], __builtin_fmaf(xr[c], eis[c], enm[c]), r2[c]);
                }
                st_lds[j * WF_THREADS + tid] = r1; st_lds[(NT + j) * WF_THREADS + tid] = r2;
                stn(p.gz_in + coff_cur + 16 * j, v);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();
    if (p.sums) {      // lanes with the same q hold the same channels: sum over the 16 rows (m), one fp64 atomic per wave, channel and moment
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double a = (double)st_lds[j * WF_THREADS + tid][e], b = (double)st_lds[(NT + j) * WF_THREADS + tid][e];
#pragma unroll
                for (int off = 8; off >= 1; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
                if (m == 0) {
                    double *rep = p.sums + (size_t)((blockIdx.x * 4 + wave) % REP) * 2 * C;
                    atomicAdd(rep + 16 * j + 4 * q + e, a); atomicAdd(rep + C + 16 * j + 4 * q + e, b);
                }
            }
    }
    __syncthreads();
    // ---- the four waves add their weight gradients through LDS in a fixed order (over the statistics rows, which are consumed) ----
    float *red = reinterpret_cast<float *>(st_lds);
    static_assert(K * C <= 2 * NT * WF_THREADS * 4, "reduction buffer fits the statistics rows");
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int jo = 0; jo < NO; ++jo)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int h = 0; h < HI; ++h) {
                        float *dst = red + (size_t)(NO * (4 * q + e) + jo) * C + NI * m + 4 * h;
                        f32x4 v = {dacc[jo][4 * h][e], dacc[jo][4 * h + 1][e], dacc[jo][4 * h + 2][e], dacc[jo][4 * h + 3][e]};
                        if (w > 0) v += *reinterpret_cast<const f32x4 *>(dst);
                        *reinterpret_cast<f32x4 *>(dst) = v;
                    }
        }
        __syncthreads();
    }
    float *out = p.dw_partial + (size_t)blockIdx.x * K * C;
    for (int i = tid; i < K * C / 4; i += WF_THREADS)
        *reinterpret_cast<f32x4 *>(out + 4 * i) = *reinterpret_cast<const f32x4 *>(red + 4 * i);
}

template <int K, int C>
size_t fused_lds_bytes() {
    return ((size_t)3 * K + 4 * C + (size_t)2 * (C / 16) * WF_THREADS * 4 + (size_t)4 * WF_ROWS * (K + 4 + 2 * (C + 4))) * sizeof(float);
}

template <int K, int C>
int launch_fused(const WregFusedP &p, unsigned grid, hipStream_t st) {
    const size_t bytes = fused_lds_bytes<K, C>();
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(wreg_bwd_fused_kernel<K, C>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((wreg_bwd_fused_kernel<K, C>), dim3(grid), dim3(WF_THREADS), bytes, st, p);
    I2P_RETURN_LAUNCH_STATUS();
}

}  // namespace

// k = output channels of the layer (gz / y), c = its input channels (x, gz_in).  Shapes: the HBM-bound layers (k = 64).
bool i2p_wreg_bwd_fused_ok(long long rows, int k, int c) {
    static const char *off = getenv("I2P_NO_FUSED_BWD");
    static const char *nw = getenv("I2P_NO_WREG");
    if ((off && off[0] == '1') || (nw && nw[0] == '1')) return false;
    return rows >= 65536 && (rows % WF_ROWS) == 0 && k == 64 && (c == 64 || c == 128) &&
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
    WregFusedP p;
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
