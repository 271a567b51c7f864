// wgrad of the bf16-storage layers (BASELINE configs[2] / [4]) with the ACCUMULATORS stationary in registers — the bf16
// counterpart of mlp_wreg.hip's wreg_wgrad_kernel, replacing mlp_bf16.hip's wgrad_bf16_kernel (128-row tiles staged
// through transposed LDS images, one barrier-synchronous block per CU: 2.7 TB/s) for the wide layers on many rows.
// (reference op: PPBackbone_center.py:10-51, weight gradient of 1x1 conv + batch-stat BN.)
//
// dW[o][c] = sum_rows g^y[row][o] * a[row][c] on v_mfma_f32_32x32x16_bf16: rows are the contraction axis, a 16-row strip is
// ONE k-step of (CO/32)(CI/32) tiles.  The MFMA wants, per lane, 8 consecutive ROWS of one channel packed two per register;
// memory has rows of consecutive channels.  Tile jo of the A operand holds output channels {NO*i + jo} (i = lane & 31,
// NO = CO/32): a lane then reads NO consecutive channels (8 or 4 bytes) of each of its 8 rows (k-half lane >> 5 = rows
// 8kh..8kh+7) — whole 256/128-byte rows per load instruction —, forms g^y / a in fp32 and packs ROW PAIRS with
// v_cvt_pk_bf16_f32 straight into the operand registers: the transposition costs nothing.  No LDS, no barrier in the
// loop; two strips of raw data in flight per wave.  (A forward on the same scheme — weights in registers, x as B operand
// straight from global memory — was measured too: 290 us against 201 us for rg_fwd_kernel at batch 16, whose LDS-staged
// strips with two waves per SIMD already stream at 4.3-4.8 TB/s; not kept.)  The four waves of a block add their 128x128 results through LDS in a
// fixed order; block partials go to dw_partial[block] (reduced by the caller, as before).
#include "bf16_common.h"

namespace {

constexpr int WB_THREADS = 256;
constexpr int WB_ROWS = 16;

struct WregWgradBf16P {
    long long rows;              // multiple of 16
    const bf16_t *gz, *y;        // [rows, CO]
    const float *g_coef;         // [6][CO] m1, m2, scale, mean, invstd, beta of the BN behind, or nullptr (gz is dL/dy)
    float g_slope;
    const bf16_t *x;             // [rows, CI]  (two sources: [rows, CI/2])
    const float *in_coef;        // [3][CI] or nullptr  (two sources: [3][CI/2])
    float slope_in;
    const bf16_t *xb; const float *in_coef_b; float slope_b;     // second source [rows, CI/2] (TWO instantiation, CI = 128)
    float *dw_partial;           // [grid][CO*CI]
};

template <int NC> struct RawRow { unsigned v[(NC + 1) / 2]; };      // NC bf16 channels of one row

template <int NC>
__device__ __forceinline__ RawRow<NC> ld_row(const bf16_t *p) {
    RawRow<NC> r;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    if constexpr (NC == 4) { const u32x2 t = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(p)); r.v[0] = t.x; r.v[1] = t.y; }
    else if constexpr (NC == 2) r.v[0] = __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(p));
    else r.v[0] = __builtin_nontemporal_load(p);             // one channel: 2 bytes (a 32-channel row = the 32 lanes of a half)
    return r;
}

// TWO: the layer input is two tensors of CI/2 channels (x, xb): a lane's four channels are 2i, 2i+1 of x and 2i, 2i+1 of xb
// (tile jc <-> channel jc < 2 ? 2i + jc : CI/2 + 2i + jc - 2), one dword per row and source.
template <int CO, int CI, bool TWO>
__global__ __launch_bounds__(WB_THREADS, 1) void wreg_wgrad_bf16_kernel(WregWgradBf16P p) {
    static_assert(!TWO || CI == 128, "two sources: 64 + 64");
    constexpr int NO = CO / 32, NI = CI / 32;                   // tiles = channels per lane (4 or 2)
    __shared__ float red[CO * CI];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kh = lane >> 5;
    // per-lane constants of its NO output / NI input channels
    float gA[NO], gB[NO], gC[NO], za[NO], zb[NO], xa[NI], xb[NI];
    const bool has_g = p.g_coef != nullptr, g_act = has_g && p.g_slope != 1.f;
#pragma unroll
    for (int u = 0; u < NO; ++u) {
        gA[u] = 1.f; gB[u] = 0.f; gC[u] = 0.f; za[u] = 1.f; zb[u] = 0.f;
        if (has_g) {
            const int ch = NO * i + u;
            const float m1 = p.g_coef[ch], m2 = p.g_coef[CO + ch], sc = p.g_coef[2 * CO + ch], mu = p.g_coef[3 * CO + ch],
                        is = p.g_coef[4 * CO + ch], be = p.g_coef[5 * CO + ch];
            gA[u] = sc; gB[u] = -(sc * m2) * is; gC[u] = -(sc * m1) - gB[u] * mu; za[u] = sc; zb[u] = be - mu * sc;
        }
    }
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        xa[u] = 1.f; xb[u] = 0.f;
        if (p.in_coef) {
            const float *cf = (TWO && u >= 2) ? p.in_coef_b : p.in_coef;
            const int ld = TWO ? CI / 2 : CI, ch = TWO ? 2 * i + (u & 1) : NI * i + u;
            xa[u] = cf[ld + ch]; xb[u] = cf[2 * ld + ch] - cf[ch] * xa[u];
        }
    }
    auto xcol = [&](int u) -> int { return TWO ? (u < 2 ? 2 * i + u : CI / 2 + 2 * i + u - 2) : NI * i + u; };
    i2p_f32x16 acc[NO][NI];
#pragma unroll
    for (int jo = 0; jo < NO; ++jo)
#pragma unroll
        for (int jc = 0; jc < NI; ++jc)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[jo][jc][e] = 0.f;

    const long long nstrips = p.rows / WB_ROWS;
    const long long stride = (long long)gridDim.x * 4;
    const long long first = (long long)blockIdx.x * 4 + wave;
    const int n_mine = __builtin_amdgcn_readfirstlane(first < nstrips ? (int)((nstrips - first + stride - 1) / stride) : 0);
    if (n_mine > 0) {
        // byte offsets (32 bits, launcher: tensors < 4 GB) of this lane's first row (8 kh) of the strip being REQUESTED
        unsigned goff = (unsigned)((((size_t)first * WB_ROWS + 8 * kh) * CO + NO * i) * 2);
        constexpr int XLD = TWO ? CI / 2 : CI;
        unsigned xoff = (unsigned)((((size_t)first * WB_ROWS + 8 * kh) * XLD + (TWO ? 2 : NI) * i) * 2);
        const unsigned g_step = __builtin_amdgcn_readfirstlane((unsigned)(stride * WB_ROWS * CO * 2));
        const unsigned x_step = __builtin_amdgcn_readfirstlane((unsigned)(stride * WB_ROWS * XLD * 2));
        int requested = 0;
        auto at = [](const bf16_t *base, unsigned byte_off) -> const bf16_t * { return reinterpret_cast<const bf16_t *>(reinterpret_cast<const char *>(base) + byte_off); };
        struct Raw { RawRow<NO> g[8], y[8]; RawRow<NI> x[8]; };
        auto load = [&](Raw &R) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                R.g[j] = ld_row<NO>(at(p.gz + j * CO, goff));
                if (has_g) R.y[j] = ld_row<NO>(at(p.y + j * CO, goff));
                if constexpr (TWO) {
                    R.x[j].v[0] = ld_row<2>(at(p.x + j * XLD, xoff)).v[0];
                    R.x[j].v[1] = ld_row<2>(at(p.xb + j * XLD, xoff)).v[0];
                } else R.x[j] = ld_row<NI>(at(p.x + j * CI, xoff));
            }
        };
        auto advance = [&]() { if (requested + 1 < n_mine) { goff += g_step; xoff += x_step; ++requested; } };
        // raw strip -> operands: fp32 math per element, row pairs packed into the registers the MFMAs read
        auto pack = [&](const Raw &R, unsigned (&A)[NO][4], unsigned (&Bm)[NI][4]) {
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {                     // rows 2jp, 2jp+1 of this lane's 8
                float g0[NO], g1[NO];
#pragma unroll
                for (int u = 0; u < NO; ++u) {
                    const unsigned w0 = R.g[2 * jp].v[u >> 1], w1 = R.g[2 * jp + 1].v[u >> 1];
                    float t0 = (u & 1) ? bf_hi(w0) : bf_lo(w0), t1 = (u & 1) ? bf_hi(w1) : bf_lo(w1);
                    if (has_g) {
                        const unsigned y0w = R.y[2 * jp].v[u >> 1], y1w = R.y[2 * jp + 1].v[u >> 1];
                        const float y0 = (u & 1) ? bf_hi(y0w) : bf_lo(y0w), y1 = (u & 1) ? bf_hi(y1w) : bf_lo(y1w);
                        if (g_act) { t0 = bf_bnz(y0, za[u], zb[u]) > 0.f ? t0 : t0 * p.g_slope; t1 = bf_bnz(y1, za[u], zb[u]) > 0.f ? t1 : t1 * p.g_slope; }
                        t0 = __builtin_fmaf(gA[u], t0, __builtin_fmaf(gB[u], y0, gC[u]));
                        t1 = __builtin_fmaf(gA[u], t1, __builtin_fmaf(gB[u], y1, gC[u]));
                    }
                    g0[u] = t0; g1[u] = t1;
                }
#pragma unroll
                for (int u = 0; u < NO; ++u) A[u][jp] = bf_pack2(g0[u], g1[u]);
#pragma unroll
                for (int u = 0; u < NI; ++u) {
                    const unsigned w0 = R.x[2 * jp].v[u >> 1], w1 = R.x[2 * jp + 1].v[u >> 1];
                    float t0 = (u & 1) ? bf_hi(w0) : bf_lo(w0), t1 = (u & 1) ? bf_hi(w1) : bf_lo(w1);
                    const float sl = (TWO && u >= 2) ? p.slope_b : p.slope_in;
                    if (p.in_coef) { t0 = bf_act(bf_bnz(t0, xa[u], xb[u]), sl); t1 = bf_act(bf_bnz(t1, xa[u], xb[u]), sl); }
                    Bm[u][jp] = bf_pack2(t0, t1);
                }
            }
        };
        auto mma = [&](const unsigned (&A)[NO][4], const unsigned (&Bm)[NI][4]) {
#pragma unroll
            for (int jo = 0; jo < NO; ++jo) {
                const uint4 av = make_uint4(A[jo][0], A[jo][1], A[jo][2], A[jo][3]);
#pragma unroll
                for (int jc = 0; jc < NI; ++jc) {
                    const uint4 bv = make_uint4(Bm[jc][0], Bm[jc][1], Bm[jc][2], Bm[jc][3]);
                    acc[jo][jc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(i2p_bf16x8, av), __builtin_bit_cast(i2p_bf16x8, bv), acc[jo][jc], 0, 0, 0);
                }
            }
        };
        auto strip = [&](const Raw &R) { unsigned A[NO][4], Bm[NI][4]; pack(R, A, Bm); mma(A, Bm); };
        // strips of raw data in flight: two where the registers allow it, one at 128 x 128 (256 accumulator registers)
        constexpr bool DEEP = CO * CI < 128 * 128;
        if constexpr (DEEP) {
            Raw R0, R1;
            load(R0); advance();
            load(R1); advance();
            int k = 0;
            for (; k + 1 < n_mine; k += 2) {
                strip(R0);
                load(R0); advance();                             // strip k + 2
                __builtin_amdgcn_sched_barrier(0);
                strip(R1);
                load(R1); advance();                             // strip k + 3
                __builtin_amdgcn_sched_barrier(0);
            }
            if (k < n_mine) strip(R0);
        } else {
            // operands of strip k are packed, then strip k + 1 is requested, then the 16 MFMAs of strip k run under it
            Raw R;
            load(R); advance();
            for (int k = 0; k < n_mine; ++k) {
                unsigned A[NO][4], Bm[NI][4];
                pack(R, A, Bm);
                load(R); advance();
                __builtin_amdgcn_sched_barrier(0);
                mma(A, Bm);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // D: col = lane & 31 -> c = NI * n + jc; row = (e & 3) + 8 (e >> 2) + 4 kh -> o = NO * row + jo
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int jo = 0; jo < NO; ++jo)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int o = NO * ((e & 3) + 8 * (e >> 2) + 4 * kh) + jo;
#pragma unroll
                    for (int jc = 0; jc < NI; ++jc) {
                        float *dst = red + (size_t)o * CI + xcol(jc);
                        *dst = (w > 0 ? *dst : 0.f) + acc[jo][jc][e];
                    }
                }
        }
        __syncthreads();
    }
    float *out = p.dw_partial + (size_t)blockIdx.x * CO * CI;
    for (int t = tid; t < CO * CI / 4; t += WB_THREADS)
        *reinterpret_cast<float4 *>(out + 4 * t) = *reinterpret_cast<const float4 *>(red + 4 * t);
}

template <int CO, int CI>
int launch(const WregWgradBf16P &p, unsigned grid, hipStream_t st) {
    if constexpr (CI == 128) {
        if (p.xb) { hipLaunchKernelGGL((wreg_wgrad_bf16_kernel<CO, CI, true>), dim3(grid), dim3(WB_THREADS), 0, st, p); I2P_RETURN_LAUNCH_STATUS(); }
    }
    if (p.xb) return I2P_ERR_BAD_ARG;
    hipLaunchKernelGGL((wreg_wgrad_bf16_kernel<CO, CI, false>), dim3(grid), dim3(WB_THREADS), 0, st, p);
    I2P_RETURN_LAUNCH_STATUS();
}


// =====================================================================================================================
// wgrad of the NARROW bf16 layers on many rows (level-1 set abstraction at batch 16: 12/16 -> 16 -> 16 -> 32 channels on
// 1.8 M rows): the same streaming scheme as mlp_wreg.hip small_wgrad_kernel — rows are the K axis of v_mfma_f32_16x16x4_f32
// (k-step t of a 16-row strip = rows 4t + q, q = lane >> 4), 16 waves per CU hide the latency by occupancy, the next strip is
// requested before the current one is consumed.  Every load is a DWORD (two bf16 channels):
//   32-channel tensor: lane n reads channels (2n, 2n+1) of row 4t + q — the 16 lanes of a k-slot cover the 64-byte row; the
//     low halves are tile 0 (channel 2n), the high halves tile 1 (channel 2n+1) of the same k-step;
//   16-channel tensor: lanes n < 8 read channels (2n, 2n+1) of row 8u + q, lanes n >= 8 the same channels of row 8u + 4 + q;
//     one DPP row rotation by 8 exchanges halves so that lane n holds channel 2(n & 7) + (n >> 3) of BOTH rows: low half =
//     k-step 2u, high half = k-step 2u + 1.
// g^y and the activated input are formed in fp32 exactly like the bf16 layer kernels form them and rounded to bf16 (the
// operands the bf16 MFMA path sees); accumulation fp32.  (wgrad_bf16_kernel stages these through LDS: 373 us for 16 -> 32.)
// =====================================================================================================================
struct SmallWgradBf16P {
    long long rows;              // multiple of 16
    int cin, cout;               // cin <= 16, cout = 16 * NO
    const bf16_t *gz, *y;        // [rows, cout]
    const float *g_coef; float g_slope;
    const void *x;               // [rows, cin] bf16 (XBF, cin = 16) or f32
    const float *in_coef; float slope_in;
    float *dw_partial;           // [grid][cout*cin]
};
constexpr int SWB_THREADS = 1024;

// 16-channel rows: the strip's two dwords of this lane -> values of k-steps 0..3 for channel 2(n&7) + (n>>3)
__device__ __forceinline__ void swb_unzip16(const unsigned (&raw)[2], bool upper, float (&v)[4]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const unsigned own = raw[u];
        const unsigned par = (unsigned)__builtin_amdgcn_update_dpp(0, (int)own, 0x128 /* row_ror:8 */, 0xf, 0xf, false);
        v[2 * u] = upper ? bf_hi(par) : bf_lo(own);
        v[2 * u + 1] = upper ? bf_hi(own) : bf_lo(par);
    }
}

template <int NO, bool XBF>
__global__ __launch_bounds__(SWB_THREADS) void small_wgrad_bf16_kernel(SmallWgradBf16P p) {
    __shared__ float red[16 * NO * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    const bool upper = n >= 8;
    const int CO = 16 * NO, CI = p.cin;
    const int perm16 = 2 * (n & 7) + (n >> 3);                   // this lane's channel of a 16-channel bf16 tensor
    const bool has_g = p.g_coef != nullptr, g_act = has_g && p.g_slope != 1.f;
    float gA[NO], gB[NO], gC[NO], za[NO], zb[NO];
#pragma unroll
    for (int jo = 0; jo < NO; ++jo) {
        gA[jo] = 1.f; gB[jo] = 0.f; gC[jo] = 0.f; za[jo] = 1.f; zb[jo] = 0.f;
        if (has_g) {
            const int ch = NO == 2 ? 2 * n + jo : perm16;
            const float m1 = p.g_coef[ch], m2 = p.g_coef[CO + ch], sc = p.g_coef[2 * CO + ch], mu = p.g_coef[3 * CO + ch],
                        is = p.g_coef[4 * CO + ch], be = p.g_coef[5 * CO + ch];
            gA[jo] = sc; gB[jo] = -(sc * m2) * is; gC[jo] = -(sc * m1) - gB[jo] * mu; za[jo] = sc; zb[jo] = be - mu * sc;
        }
    }
    const int xch = XBF ? perm16 : n;                            // this lane's input channel
    const bool xin = xch < CI;
    float xa = 1.f, xb = 0.f;
    if (p.in_coef && xin) { xa = p.in_coef[CI + xch]; xb = p.in_coef[2 * CI + xch] - p.in_coef[xch] * xa; }
    const unsigned *gzw = reinterpret_cast<const unsigned *>(p.gz), *yw = reinterpret_cast<const unsigned *>(p.y);
    const unsigned *xw = reinterpret_cast<const unsigned *>(p.x);
    const float *xf = reinterpret_cast<const float *>(p.x);

    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc[NO];
#pragma unroll
    for (int jo = 0; jo < NO; ++jo) acc[jo] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long nstrips = p.rows / 16;
    const long long stride = (long long)gridDim.x * (SWB_THREADS / 64);
    constexpr int NG = NO == 2 ? 4 : 2, NX = XBF ? 2 : 4;
    struct Raw { unsigned g[NG], y[NG]; unsigned x[NX]; };
    auto load = [&](long long s, Raw &R) {
        const size_t r0 = (size_t)s * 16;
        if constexpr (NO == 2) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {                        // row 4t + q, dword n of its 16
                R.g[t] = __builtin_nontemporal_load(gzw + (r0 + 4 * t + q) * 16 + n);
                R.y[t] = has_g ? __builtin_nontemporal_load(yw + (r0 + 4 * t + q) * 16 + n) : 0u;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u) {                        // row 8u + q (+4 for the upper lanes), dword n & 7 of its 8
                const size_t r = r0 + 8 * u + q + (upper ? 4 : 0);
                R.g[u] = __builtin_nontemporal_load(gzw + r * 8 + (n & 7));
                R.y[u] = has_g ? __builtin_nontemporal_load(yw + r * 8 + (n & 7)) : 0u;
            }
        }
        if constexpr (XBF) {
#pragma unroll
            for (int u = 0; u < 2; ++u) R.x[u] = __builtin_nontemporal_load(xw + (r0 + 8 * u + q + (upper ? 4 : 0)) * 8 + (n & 7));
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) R.x[t] = xin ? __float_as_uint(__builtin_nontemporal_load(xf + (r0 + 4 * t + q) * CI + n)) : 0u;
        }
    };
    Raw cur, nxt;
    long long s = (long long)blockIdx.x * (SWB_THREADS / 64) + wave;
    if (s < nstrips) load(s, cur);
    for (; s < nstrips; s += stride) {
        const long long sn = s + stride < nstrips ? s + stride : s;
        load(sn, nxt);
        float gv[NO][4], yv[NO][4], xv[4];
        if constexpr (NO == 2) {
#pragma unroll
            for (int t = 0; t < 4; ++t) { gv[0][t] = bf_lo(cur.g[t]); gv[1][t] = bf_hi(cur.g[t]); yv[0][t] = bf_lo(cur.y[t]); yv[1][t] = bf_hi(cur.y[t]); }
        } else {
            swb_unzip16(cur.g, upper, gv[0]); swb_unzip16(cur.y, upper, yv[0]);
        }
        if constexpr (XBF) swb_unzip16(cur.x, upper, xv);
        else
#pragma unroll
            for (int t = 0; t < 4; ++t) xv[t] = __uint_as_float(cur.x[t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float a = xv[t];
            if (p.in_coef) a = bf_act(bf_bnz(a, xa, xb), p.slope_in);
            a = xin ? bf_round(a) : 0.f;
#pragma unroll
            for (int jo = 0; jo < NO; ++jo) {
                float tg = gv[jo][t];
                if (has_g) {
                    if (g_act) tg = bf_bnz(yv[jo][t], za[jo], zb[jo]) > 0.f ? tg : tg * p.g_slope;
                    tg = __builtin_fmaf(gA[jo], tg, __builtin_fmaf(gB[jo], yv[jo][t], gC[jo]));
                }
                acc[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf_round(tg), a, acc[jo], 0, 0, 0);
            }
        }
        cur = nxt;
    }
    // D of tile jo: lane (column j = n, q), register e = row i = 4q + e; row i <-> output channel 2i + jo (32 outputs) or
    // 2(i & 7) + (i >> 3) (16 outputs), column j <-> input channel xch.  The waves of the block add through LDS in a fixed order.
    for (int w = 0; w < SWB_THREADS / 64; ++w) {
        if (wave == w) {
#pragma unroll
            for (int jo = 0; jo < NO; ++jo)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * q + e;
                    const int o = NO == 2 ? 2 * i + jo : 2 * (i & 7) + (i >> 3);
                    float *dst = red + o * 16 + xch;
                    *dst = (w > 0 ? *dst : 0.f) + acc[jo][e];
                }
        }
        __syncthreads();
    }
    float *out = p.dw_partial + (size_t)blockIdx.x * CO * CI;
    for (int i = tid; i < CO * CI; i += SWB_THREADS) out[i] = red[(i / CI) * 16 + i % CI];
}

}  // namespace

bool i2p_wreg_wgrad_bf16_ok(long long rows, int cin, int cout) {
    static const char *e = getenv("I2P_NO_WREG");
    if (e && e[0] == '1') return false;
    return rows >= 65536 && (rows % WB_ROWS) == 0 && (cin == 32 || cin == 64 || cin == 128) && (cout == 32 || cout == 64 || cout == 128) &&
           (unsigned long long)rows * 128ull * 2ull < (1ull << 32);
}

int i2p_wreg_wgrad_bf16(long long rows, int cin, int cout, const unsigned short *gz, const unsigned short *y, const float *g_coef,
                        float g_slope, const unsigned short *x, const float *in_coef, float slope_in, float *dw_partial, unsigned grid,
                        void *stream, const unsigned short *xb, const float *in_coef_b, float slope_b) {
    if (!i2p_wreg_wgrad_bf16_ok(rows, cin, cout) || !gz || !x || !dw_partial || grid == 0 || (g_coef && !y)) return I2P_ERR_BAD_ARG;
    if (xb && (cin != 128 || !in_coef || !in_coef_b)) return I2P_ERR_BAD_ARG;
    WregWgradBf16P p;
    p.rows = rows; p.gz = gz; p.y = y; p.g_coef = g_coef; p.g_slope = g_slope; p.x = x; p.in_coef = in_coef; p.slope_in = slope_in;
    p.dw_partial = dw_partial; p.xb = xb; p.in_coef_b = in_coef_b; p.slope_b = slope_b;
    hipStream_t st = (hipStream_t)stream;
#define WB_CASE(O, I) if (cout == O && cin == I) return launch<O, I>(p, grid, st)
    WB_CASE(128, 128); WB_CASE(128, 64); WB_CASE(128, 32); WB_CASE(64, 128); WB_CASE(64, 64); WB_CASE(64, 32);
    WB_CASE(32, 128); WB_CASE(32, 64); WB_CASE(32, 32);
#undef WB_CASE
    return I2P_ERR_BAD_ARG;
}

bool i2p_small_wgrad_bf16_ok(long long rows, int cin, int cout, int x_bf16) {
    static const char *e = getenv("I2P_NO_WREG");
    if (e && e[0] == '1') return false;
    return rows >= 262144 && (rows % 16) == 0 && (cout == 16 || cout == 32) && (x_bf16 ? cin == 16 : (cin >= 4 && cin <= 16 && (cin & 3) == 0));
}

int i2p_small_wgrad_bf16(long long rows, int cin, int cout, const unsigned short *gz, const unsigned short *y, const float *g_coef,
                         float g_slope, const void *x, int x_bf16, const float *in_coef, float slope_in, float *dw_partial, unsigned grid,
                         void *stream) {
    if (!i2p_small_wgrad_bf16_ok(rows, cin, cout, x_bf16) || !gz || !x || !dw_partial || grid == 0 || (g_coef && !y)) return I2P_ERR_BAD_ARG;
    SmallWgradBf16P p;
    p.rows = rows; p.cin = cin; p.cout = cout; p.gz = gz; p.y = y; p.g_coef = g_coef; p.g_slope = g_slope; p.x = x; p.in_coef = in_coef;
    p.slope_in = slope_in; p.dw_partial = dw_partial;
    hipStream_t st = (hipStream_t)stream;
    if (cout == 16) {
        if (x_bf16) hipLaunchKernelGGL((small_wgrad_bf16_kernel<1, true>), dim3(grid), dim3(SWB_THREADS), 0, st, p);
        else hipLaunchKernelGGL((small_wgrad_bf16_kernel<1, false>), dim3(grid), dim3(SWB_THREADS), 0, st, p);
    } else {
        if (x_bf16) hipLaunchKernelGGL((small_wgrad_bf16_kernel<2, true>), dim3(grid), dim3(SWB_THREADS), 0, st, p);
        else hipLaunchKernelGGL((small_wgrad_bf16_kernel<2, false>), dim3(grid), dim3(SWB_THREADS), 0, st, p);
    }
    I2P_RETURN_LAUNCH_STATUS();
}
