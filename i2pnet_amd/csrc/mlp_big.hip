// Fused layer kernels for the WIDE layers on FEW rows: cin > 160 or cout > 128 (up to 320 channels; levels 3-4, the
// up-convolutions and the flow predictors: 128 -> 256, 256 -> 128, 64 -> 192, 320 -> 128 on <= 15 000 rows).  Their weights do not
// fit the LDS next to a row tile (128 x 321 floats > 160 KB), so the first- and second-generation kernels of mlp.hip refuse
// them and the network used to run them as rocBLAS GEMM + separate BN-statistics / BN+activation / BN-backward kernels
// (10 launches per layer and step).  Same operator semantics as i2p_lin_fwd / i2p_lin_bwd (reference: Conv2d.forward,
// PPBackbone_center.py:34-46 — 1x1 conv, batch-statistics BN, LeakyReLU), K-tiled instead of weight-resident:
//
//   big_nt_kernel<false>  y = act(bn(x)) W^T, BN + activation of the layer in front applied while the operand is loaded,
//                         fp64 {sum y, sum y^2} of the output;
//   big_nt_kernel<true>   gz_in = act'(z_prev) * (g^y W): BN backward of the layer behind applied on load
//                         (g^y = A t + B y + C per channel), activation derivative of the layer in front and its
//                         BN-backward statistics {sum, sum * xhat} in the store phase;
//   big_tn_kernel         dW = (g^y)^T act(bn(x)): rows cut over the grid like gemm_tn.hip, both transforms on load.
//
// All three feed v_mfma_f32_16x16x4_f32 straight from global memory (these tensors are a few MB: L2 resident).  The
// contraction index is permuted inside 16-element chunks (k-step e of a chunk takes elements 4q + e, q = lane >> 4) so that
// a lane's operands for four k-steps are ONE float4 of its row; the weights of the dgrad (contraction along W's rows) are
// read as 64-byte runs per k-slot.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int REP = I2P_BN_REPLICAS;
constexpr int BG_THREADS = 256, BG_ROWS = 64, BG_COLS = 64, BG_MAXC = 320;

struct BigP {
    long long rows;
    int K, C;                     // contraction length, output columns
    const float *a;               // fwd: x [rows,K]; dgrad: gz [rows,K]
    const float *a2;              // dgrad: y [rows,K] (pre-BN output of this layer) or nullptr
    const float *in_coef;         // fwd: [3][K] mean, scale, beta of the BN in front, or nullptr
    float slope_in;
    const float *g_coef;          // dgrad: [8][K] m1, m2, scale, mean, invstd, beta (bnbwd_coef_kernel) or nullptr
    // or (round 5: no coefficient launch) the raw material: dsums [REP][2K] = {sum gz, sum gz*xhat}, coef [3][K], mean_invstd [2][K] of the BN
    // behind, its row count, and g_out [8][K] whose rows 6, 7 block (0,0) fills with dbeta, dgamma for the caller
    const double *g_dsums; const float *g_oc, *g_omi; long long g_rows; float *g_out;
    float g_slope;
    const float *w;               // [cout][cin]: fwd [C][K], dgrad [K][C]
    float *out;                   // [rows, C]
    double *sums;                 // [REP][2C] or nullptr
    const float *ex, *e_coef, *e_mi; float e_slope;      // dgrad store phase: pre-BN tensor in front [rows,C], its coef / mean_invstd
};

__device__ __forceinline__ f32x4 ld4g(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

// the six constants bnbwd_coef_kernel (mlp.hip) wrote per channel, formed here from the same sums in the same order (bit-identical)
__device__ __forceinline__ void bnbwd_consts_of(const double *dsums, const float *oc, const float *omi, long long rows, int c, int ch,
                                                float &m1, float &m2, float &sc, float &mu, float &is, float &be, float *out8) {
    double sd = 0.0, sx = 0.0;
    for (int r = 0; r < I2P_BN_REPLICAS; ++r) { sd += dsums[(size_t)r * 2 * c + ch]; sx += dsums[(size_t)r * 2 * c + c + ch]; }
    m1 = (float)(sd / (double)rows); m2 = (float)(sx / (double)rows);
    sc = oc[c + ch]; mu = omi[ch]; is = omi[c + ch]; be = oc[2 * c + ch];
    if (out8) { out8[6 * c + ch] = (float)sd; out8[7 * c + ch] = (float)sx; }          // dbeta, dgamma of this BN (read back by the caller)
}

template <bool DGRAD>
__global__ __launch_bounds__(BG_THREADS) void big_nt_kernel(BigP p) {
    __shared__ __attribute__((aligned(16))) float tab[5][BG_MAXC];                            // per contraction index: fwd {a, b}; dgrad {gA, gB, gC, za, zb}
    __shared__ double red[2][4][BG_COLS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
    const bool xf = DGRAD ? (p.g_coef != nullptr || p.g_dsums != nullptr) : p.in_coef != nullptr;
    const bool g_act = DGRAD && xf && p.g_slope != 1.f;
    if (xf) {
        for (int k = tid; k < p.K; k += BG_THREADS) {
            if constexpr (DGRAD) {
                float m1, m2, sc, mu, is, be;
                if (p.g_dsums) bnbwd_consts_of(p.g_dsums, p.g_oc, p.g_omi, p.g_rows, p.K, k, m1, m2, sc, mu, is, be,
                                               (blockIdx.x == 0 && blockIdx.y == 0) ? p.g_out : nullptr);
                else { m1 = p.g_coef[k]; m2 = p.g_coef[p.K + k]; sc = p.g_coef[2 * p.K + k]; mu = p.g_coef[3 * p.K + k]; is = p.g_coef[4 * p.K + k]; be = p.g_coef[5 * p.K + k]; }
                const float gB = -(sc * m2) * is;
                tab[0][k] = sc; tab[1][k] = gB; tab[2][k] = -(sc * m1) - gB * mu; tab[3][k] = sc; tab[4][k] = be - mu * sc;
            } else {
                const float sc = p.in_coef[p.K + k];
                tab[0][k] = sc; tab[1][k] = p.in_coef[2 * p.K + k] - p.in_coef[k] * sc;
            }
        }
        __syncthreads();
    }
    const long long r = (long long)blockIdx.x * BG_ROWS + 16 * wave + i;        // this lane's operand row
    const bool r_ok = r < p.rows;
    const int c0 = blockIdx.y * BG_COLS;
    const float *arow = p.a + (size_t)(r_ok ? r : 0) * p.K;
    const float *yrow = (DGRAD && p.a2) ? p.a2 + (size_t)(r_ok ? r : 0) * p.K : nullptr;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    struct Raw { f32x4 a, y, b[4]; };
    auto load = [&](int k0, Raw &R) {
        const int k = k0 + 4 * q;                                // this lane's four contraction indices k .. k+3
        const bool k_ok = k < p.K;                               // K % 4 == 0
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        R.a = (r_ok && k_ok) ? ld4g(arow + k) : z;
        R.y = (DGRAD && yrow && r_ok && k_ok) ? ld4g(yrow + k) : z;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int c = c0 + 16 * t + i;
            if constexpr (DGRAD) {
                f32x4 v = z;
                if (k_ok && c < p.C) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = p.w[(size_t)(k + e) * p.C + c];
                }
                R.b[t] = v;
            } else {
                R.b[t] = (k_ok && c < p.C) ? ld4g(p.w + (size_t)c * p.K + k) : z;
            }
        }
    };
    Raw cur, nxt;
    load(0, cur);
    for (int k0 = 0; k0 < p.K; k0 += 16) {
        if (k0 + 16 < p.K) load(k0 + 16, nxt);
        f32x4 av = cur.a;
        if (xf) {
            const int k = k0 + 4 * q;
            if (k < p.K) {
                if constexpr (DGRAD) {
                    const f32x4 gA = ld4g(&tab[0][k]), gB = ld4g(&tab[1][k]), gC = ld4g(&tab[2][k]);
                    if (g_act) {
                        const f32x4 za = ld4g(&tab[3][k]), zb = ld4g(&tab[4][k]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) av[e] = __builtin_fmaf(cur.y[e], za[e], zb[e]) > 0.f ? av[e] : av[e] * p.g_slope;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[e] = __builtin_fmaf(gA[e], av[e], __builtin_fmaf(gB[e], cur.y[e], gC[e]));
                } else {
                    const f32x4 ta = ld4g(&tab[0][k]), tb = ld4g(&tab[1][k]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float zz = __builtin_fmaf(av[e], ta[e], tb[e]); av[e] = zz > 0.f ? zz : zz * p.slope_in; }
                }
                if (!r_ok) av = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], cur.b[t][e], acc[t], 0, 0, 0);
        cur = nxt;
    }

    // D of tile t: lane (column j = i, q), register e = row 4q + e of the wave's 16
    const long long row0 = (long long)blockIdx.x * BG_ROWS + 16 * wave + 4 * q;
    const bool e_on = DGRAD && p.e_coef != nullptr;
    double s1[4], s2[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int c = c0 + 16 * t + i;
        const bool c_ok = c < p.C;
        float mu = 0.f, sc = 0.f, be = 0.f, is = 0.f;
        if (e_on && c_ok) { mu = p.e_coef[c]; sc = p.e_coef[p.C + c]; be = p.e_coef[2 * p.C + c]; is = p.e_mi[p.C + c]; }
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long long rr = row0 + e;
            if (rr < p.rows && c_ok) {
                float v = acc[t][e];
                if constexpr (DGRAD) {
                    if (e_on) {
                        const float xr = p.ex[(size_t)rr * p.C + c];
                        const float zz = (xr - mu) * sc + be;
                        v = zz > 0.f ? v : v * p.e_slope;
                        a1 += v; a2 = __builtin_fmaf(v, (xr - mu) * is, a2);
                    }
                } else {
                    a1 += v; a2 = __builtin_fmaf(v, v, a2);
                }
                p.out[(size_t)rr * p.C + c] = v;
            }
        }
        s1[t] = (double)a1; s2[t] = (double)a2;
    }
    if (p.sums && (!DGRAD || e_on)) {
        // column sums: over the four row groups of the wave (lanes q), then over the four waves, one fp64 atomic per column and block
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            double a = s1[t], b = s2[t];
            a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
            a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
            if (q == 0) { red[0][wave][16 * t + i] = a; red[1][wave][16 * t + i] = b; }
        }
        __syncthreads();
        if (tid < 2 * BG_COLS) {
            const int which = tid >> 6, col = tid & 63, c = c0 + col;
            if (c < p.C) {
                const double v = (red[which][0][col] + red[which][1][col]) + (red[which][2][col] + red[which][3][col]);
                atomicAdd(p.sums + (size_t)(blockIdx.x % REP) * 2 * p.C + (size_t)which * p.C + c, v);
            }
        }
    }
}

// ---- wgrad: dW[o][c] = sum_r g^y[r][o] * act(bn(x))[r][c], rows cut over the grid (see gemm_tn.hip for the tiling) ----
struct BigTnP {
    long long rows;
    int m, n;                     // cout, cin
    const float *gz, *y;          // [rows, m]
    const float *g_coef; float g_slope;
    const double *g_dsums; const float *g_oc, *g_omi; long long g_rows; float *g_out;      // see BigP
    const float *x;               // [rows, n]
    const float *in_coef; float slope_in;
    int tiles_n, chunk_rows;
    float *partial;               // [chunks][m*n]
};
constexpr int BT_GROUP = 16, BT_UNROLL = 2, BT_ALIGN = BT_GROUP * BT_UNROLL;

__global__ __launch_bounds__(BG_THREADS) void big_tn_kernel(BigTnP p) {
    __shared__ float red[4][4][4][4][64];                        // [wave][tm][tn][e][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, k = lane >> 4;
    const int m0 = (blockIdx.x / p.tiles_n) * 64, n0 = (blockIdx.x % p.tiles_n) * 64;
    const long long r_begin = (long long)blockIdx.y * p.chunk_rows;
    long long r_end = r_begin + p.chunk_rows; if (r_end > p.rows) r_end = p.rows;
    const int ca = m0 + 4 * i, cb = n0 + 4 * i;
    const bool a_ok = ca < p.m, b_ok = cb < p.n;                 // m, n % 4 == 0
    const bool has_g = p.g_coef != nullptr || p.g_dsums != nullptr, g_act = has_g && p.g_slope != 1.f, has_x = p.in_coef != nullptr;
    f32x4 gA = {1.f, 1.f, 1.f, 1.f}, gB = {0.f, 0.f, 0.f, 0.f}, gC = gB, za = gA, zb = gB, xa = gA, xb = gB;
    if (has_g && a_ok) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ch = ca + e;
            float m1, m2, sc, mu, is, be;
            // (rows 6, 7 for the caller: the n-tile 0 blocks of chunk 0 cover every channel once; written only when no dgrad launch did it)
            if (p.g_dsums) bnbwd_consts_of(p.g_dsums, p.g_oc, p.g_omi, p.g_rows, p.m, ch, m1, m2, sc, mu, is, be,
                                           (p.g_out && blockIdx.y == 0 && n0 == 0 && (lane >> 4) == 0 && (threadIdx.x >> 6) == 0) ? p.g_out : nullptr);
            else { m1 = p.g_coef[ch]; m2 = p.g_coef[p.m + ch]; sc = p.g_coef[2 * p.m + ch]; mu = p.g_coef[3 * p.m + ch]; is = p.g_coef[4 * p.m + ch]; be = p.g_coef[5 * p.m + ch]; }
            gA[e] = sc; gB[e] = -(sc * m2) * is; gC[e] = -(sc * m1) - gB[e] * mu; za[e] = sc; zb[e] = be - mu * sc;
        }
    }
    if (has_x && b_ok) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int ch = cb + e; xa[e] = p.in_coef[p.n + ch]; xb[e] = p.in_coef[2 * p.n + ch] - p.in_coef[ch] * xa[e]; }
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
    struct Raw { f32x4 g[BT_UNROLL], y[BT_UNROLL], x[BT_UNROLL]; };
    auto load = [&](long long r, Raw &R) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < BT_UNROLL; ++u) {
            const long long rr = r + u * BT_GROUP;
            const bool ok = rr < r_end;
            R.g[u] = (ok && a_ok) ? ld4g(p.gz + (size_t)rr * p.m + ca) : z;
            R.y[u] = (ok && a_ok && has_g) ? ld4g(p.y + (size_t)rr * p.m + ca) : z;
            R.x[u] = (ok && b_ok) ? ld4g(p.x + (size_t)rr * p.n + cb) : z;
        }
    };
    Raw cur, nxt;
    long long r = r_begin + 4 * wave + k;
    load(r, cur);
    for (long long base = r_begin; base < r_end; base += BT_ALIGN) {
        r += BT_ALIGN;
        load(r, nxt);
#pragma unroll
        for (int u = 0; u < BT_UNROLL; ++u) {
            const bool ok = (r - BT_ALIGN + u * BT_GROUP) < r_end;
            f32x4 av = cur.g[u], bv = cur.x[u];
            if (has_g) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = av[e];
                    if (g_act) t = __builtin_fmaf(cur.y[u][e], za[e], zb[e]) > 0.f ? t : t * p.g_slope;
                    av[e] = __builtin_fmaf(gA[e], t, __builtin_fmaf(gB[e], cur.y[u][e], gC[e]));
                }
            }
            if (has_x) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float zz = __builtin_fmaf(bv[e], xa[e], xb[e]); bv[e] = zz > 0.f ? zz : zz * p.slope_in; }
            }
            if (!ok || !a_ok) av = f32x4{0.f, 0.f, 0.f, 0.f};    // rows past the chunk / channels past m contribute nothing
            if (!ok || !b_ok) bv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tm], bv[tn], acc[tm][tn], 0, 0, 0);
        }
        cur = nxt;
    }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave][tm][tn][e][lane] = acc[tm][tn][e];
    __syncthreads();
    float *dst = p.partial + (size_t)blockIdx.y * p.m * p.n;
    for (int item = threadIdx.x; item < 4 * 4 * 64; item += BG_THREADS) {
        const int l = item & 63, e = (item >> 6) & 3, tm = item >> 8;
        const int mi = m0 + 4 * (4 * (l >> 4) + e) + tm, nj = n0 + 4 * (l & 15);
        if (mi >= p.m || nj >= p.n) continue;
        float v[4];
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) v[tn] = (red[0][tm][tn][e][l] + red[1][tm][tn][e][l]) + (red[2][tm][tn][e][l] + red[3][tm][tn][e][l]);
        *reinterpret_cast<float4 *>(dst + (size_t)mi * p.n + nj) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

__global__ __launch_bounds__(256) void big_reduce_kernel(int nparts, int count4, const float4 *__restrict__ parts, float4 *__restrict__ out) {
    __shared__ float4 red[16][16];
    const int tx = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int o = blockIdx.x * 16 + tx;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (o < count4)
        for (int c = pl; c < nparts; c += 16) { const float4 v = parts[(size_t)c * count4 + o]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    red[pl][tx] = s;
    __syncthreads();
    if (pl == 0 && o < count4) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) { const float4 v = red[qq][tx]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        out[o] = t;
    }
}

}  // namespace

bool i2p_big_layer_ok(long long rows, int cin, int cout) {
    static const char *e = getenv("I2P_NO_BIG");
    if (e && e[0] == '1') return false;
    // cin in (128, 160] (132 = 128 + xyz, 136 = 128 + 6 + padding: level 4, the fine cost volume's first layer) used to run on the
    // first-generation block-synchronous kernels (lin_fwd_kernel<128,4> 91 us, lin_bwd_kernel<5,4> 128 us per launch)
    const int min_cin = 128;
    return rows > 0 && (cin & 3) == 0 && (cout & 3) == 0 && cin <= BG_MAXC && cout <= BG_MAXC && (cin > min_cin || cout > 128);
}

int i2p_big_fwd(long long rows, int cin, int cout, const float *x, const float *in_coef, float slope_in, const float *w, float *y,
                double *sums, void *stream) {
    if (!i2p_big_layer_ok(rows, cin, cout) || !x || !w || !y) return I2P_ERR_BAD_ARG;
    BigP p{};
    p.rows = rows; p.K = cin; p.C = cout; p.a = x; p.in_coef = in_coef; p.slope_in = slope_in; p.w = w; p.out = y; p.sums = sums;
    const dim3 grid((unsigned)((rows + BG_ROWS - 1) / BG_ROWS), (unsigned)((cout + BG_COLS - 1) / BG_COLS));
    hipLaunchKernelGGL(big_nt_kernel<false>, grid, dim3(BG_THREADS), 0, (hipStream_t)stream, p);
    I2P_RETURN_LAUNCH_STATUS();
}

// g_out: [8][cout] scratch of the BN behind (rows 6, 7 receive dbeta, dgamma; the kernels form the constants themselves from out_dsums /
// out_coef / out_mi in their prologues) or nullptr = no BN behind; dw_partial holds max_chunks * cout * cin floats
int i2p_big_bwd(long long rows, int cin, int cout, const float *gz, const float *y, float *g_out, const double *out_dsums, const float *out_coef,
                const float *out_mi, float slope_out, const float *x,
                const float *in_coef, const float *in_mi, float slope_in, const float *w, float *gz_in, double *in_dsums,
                float *dw_partial, int max_chunks, float *dw, void *stream) {
    // dw == nullptr: input gradient only (the weight gradient is issued by another call, possibly on another stream)
    const bool g_coef = g_out != nullptr;
    if (!i2p_big_layer_ok(rows, cin, cout) || !gz || !x || !w || (dw && !dw_partial) || (!dw && !gz_in) || max_chunks < 1 ||
        (g_coef && (!y || !out_dsums || !out_coef || !out_mi))) return I2P_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(dw_partial) | reinterpret_cast<uintptr_t>(dw)) & 15) return I2P_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (gz_in) {
        BigP p{};
        p.rows = rows; p.K = cout; p.C = cin; p.a = gz; p.a2 = y; p.g_coef = nullptr; p.g_slope = g_coef ? slope_out : 1.f; p.w = w; p.out = gz_in;
        p.g_dsums = g_coef ? out_dsums : nullptr; p.g_oc = out_coef; p.g_omi = out_mi; p.g_rows = rows; p.g_out = g_out;
        p.sums = in_coef ? in_dsums : nullptr; p.ex = in_coef ? x : nullptr; p.e_coef = in_coef; p.e_mi = in_mi; p.e_slope = slope_in;
        const dim3 grid((unsigned)((rows + BG_ROWS - 1) / BG_ROWS), (unsigned)((cin + BG_COLS - 1) / BG_COLS));
        hipLaunchKernelGGL(big_nt_kernel<true>, grid, dim3(BG_THREADS), 0, st, p);
    }
    if (!dw) I2P_RETURN_LAUNCH_STATUS();
    BigTnP q{};
    q.rows = rows; q.m = cout; q.n = cin; q.gz = gz; q.y = y; q.g_coef = nullptr; q.g_slope = g_coef ? slope_out : 1.f; q.x = x; q.in_coef = in_coef;
    q.g_dsums = g_coef ? out_dsums : nullptr; q.g_oc = out_coef; q.g_omi = out_mi; q.g_rows = rows; q.g_out = gz_in ? nullptr : g_out;
    q.slope_in = slope_in; q.partial = dw_partial;
    const int tiles_m = (cout + 63) / 64; q.tiles_n = (cin + 63) / 64;
    long long want = 512 / ((long long)tiles_m * q.tiles_n); if (want < 1) want = 1; if (want > max_chunks) want = max_chunks;
    long long cr = (rows + want - 1) / want; cr = (cr + BT_ALIGN - 1) / BT_ALIGN * BT_ALIGN;
    q.chunk_rows = (int)cr;
    const int nchunks = (int)((rows + cr - 1) / cr);
    hipLaunchKernelGGL(big_tn_kernel, dim3(tiles_m * q.tiles_n, nchunks), dim3(BG_THREADS), 0, st, q);
    const int count4 = cout * cin / 4;
    if (!i2p_defer_reduce(0, nchunks, count4, dw_partial, dw))
    hipLaunchKernelGGL(big_reduce_kernel, dim3((count4 + 15) / 16), dim3(256), 0, st, nchunks, count4, reinterpret_cast<const float4 *>(dw_partial),
                       reinterpret_cast<float4 *>(dw));
    I2P_RETURN_LAUNCH_STATUS();
}
