// One-pass backward of the bf16-storage layers with 64 output channels (128 -> 64 and 64 -> 64 of both cost volumes,
// BASELINE configs[2] / [4]): input gradient AND weight gradient from ONE read of gz, y, x — the bf16 twin of
// mlp_wreg_fused.hip.  (reference op: PPBackbone_center.py:10-51 inside CostVolume.forward :383-433; backward of
// 1x1 conv + batch-stat BN + LeakyReLU.)
//
// The two-kernel form (rg_dgrad_kernel + wreg_wgrad_bf16_kernel) streamed gz, y twice and x twice (as the dgrad's `ex` and as
// the wgrad's input): rows*(4K + 3C)*2 B against rows*(2K + 2C)*2 B here (gz, y, x read once, dL/dz_in written once).
//
//   g^y = BN-backward(gz, y)                      (per-channel constants, formed on load, rounded to bf16)
//   a   = act(bn(x))                              (rounded to bf16)
//   dW[k][c] = sum_r g^y[r][k] a[r][c]            v_mfma_f32_32x32x16_bf16, rows = contraction index
//   D[r][c]  = sum_k g^y[r][k] W[k][c]            same instruction, transposed: D^T = W^T g^y^T
//   dz_in = bf16(D) * act'(bn(x)), statistics sum dz_in, sum dz_in * xhat
//
// The two contractions want g^y in two layouts (8 consecutive CHANNELS of a row per lane for D, 8 ROWS of a channel per lane for
// dW): a lane loads blocks of 8 rows x 4 or 8 channels (eight loads; every load instruction of the wave covers 4 whole
// consecutive rows), forms the values in fp32 and writes them to wave-private LDS images twice — row-major and transposed (the
// row-pair packing of v_cvt_pk_bf16_f32 IS the transposition).  One wave per SIMD, 32-row strips, no block barrier in the loop;
// a wave requests its next strip right after its staging phase, so the requests stay in flight under the MFMA phases and the
// epilogue.  The epilogue needs the raw x again (sign of bn(x), xhat): it is parked row-major in LDS by the staging phase —
// held in registers across the strip it cost, with the next strip's registers, more than the 256 architectural VGPRs (the
// first version of this kernel, 64-row strips, spilled 170 registers to scratch and with every reload waited for the prefetch).
// LDS per wave at C = 128: Gr 4 KB + Gt 4 KB + At 8 KB (re-used for the packed D strip) + Xr 8 KB.
// Everything here is far below its pipe's limit (per 32-row strip: 32 MFMAs, ~1.3 k VALU, ~110 LDS instructions against ~12 k
// cycles of HBM time at 5 TB/s); the kernel is a memory skeleton with arithmetic hung into it.
#include "bf16_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int REP = I2P_BN_REPLICAS;
constexpr int FB_THREADS = 256;
constexpr int FB_ROWS = 32;
constexpr int FB_K = 64;                 // output channels of the layer (= channels of gz / y)

struct FusedBwdBf16P {
    long long rows;
    const bf16_t *gz, *y;        // [rows, 64]
    BnBwdSrc g;                  // the BN behind: replica sums + finalised coefficients (constants formed in the prologue)
    const bf16_t *x;             // [rows, C] pre-BN tensor in front
    const float *e_coef, *e_mi;  // [3][C] mean, scale, beta; [2][C] mean, invstd
    float e_slope;
    const float *w;              // [64][C] fp32
    bf16_t *gz_in;               // [rows, C]
    double *sums;                // [REP][2C]
    float *dw_partial;           // [grid][64*C]
};

__device__ __forceinline__ int w_perm(int m) { return 16 * ((m >> 2) & 1) + (m & 3) + 4 * (m >> 3); }
// LDS images in 16-byte chunks.  8 chunks per row (row-major images of 64-channel rows, W^T): an MFMA operand read takes one
// logical chunk of 16 consecutive rows -> XOR with (row >> 1) & 7.  4 chunks per row (transposed images: 32 strip rows of one
// channel): reads take 16 consecutive channels, writes come from lanes whose channels are 4 or 8 apart -> XOR with bits 2-3 and
// 4-5 of the channel (reads conflict-free, writes 4-way: the minimum, all of a write's rows share row & 3).  16 chunks per row:
// XOR with row & 15 where fragments (16 consecutive rows, one chunk) are written, none where whole rows go in and out.
__device__ __forceinline__ int ch8(int row, int c) { return (row << 3) + (c ^ ((row >> 1) & 7)); }
__device__ __forceinline__ int ch4(int row, int c) { return (row << 2) + (c ^ (((row >> 2) ^ (row >> 4)) & 3)); }
__device__ __forceinline__ int ch16(int row, int c) { return (row << 4) + (c ^ (row & 15)); }

__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// NC bf16 channels of one row (NC = 4: 8 bytes, NC = 8: 16 bytes), streaming
template <int NC> struct Raw { unsigned v[NC / 2]; };
template <int NC> __device__ __forceinline__ Raw<NC> ld_raw(const void *ptr) {
    Raw<NC> r;
    if constexpr (NC == 8) { const uint4 t = ld_u4_stream(ptr); r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; }
    else { typedef unsigned u32x2 __attribute__((ext_vector_type(2))); const u32x2 t = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(ptr)); r.v[0] = t.x; r.v[1] = t.y; }
    return r;
}
template <int NC> __device__ __forceinline__ void st_raw(void *ptr, const Raw<NC> &r) {
    if constexpr (NC == 8) st_u4_stream(ptr, make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]));
    else { typedef unsigned u32x2 __attribute__((ext_vector_type(2))); const u32x2 t = {r.v[0], r.v[1]}; __builtin_nontemporal_store(t, reinterpret_cast<u32x2 *>(ptr)); }
}
template <int NC> __device__ __forceinline__ void unpack(const Raw<NC> &r, float (&f)[NC]) {
#pragma unroll
    for (int i = 0; i < NC / 2; ++i) { f[2 * i] = bf_lo(r.v[i]); f[2 * i + 1] = bf_hi(r.v[i]); }
}
template <int NC> __device__ __forceinline__ Raw<NC> pack(const float (&f)[NC]) {
    Raw<NC> r;
#pragma unroll
    for (int i = 0; i < NC / 2; ++i) r.v[i] = bf_pack2(f[2 * i], f[2 * i + 1]);
    return r;
}
// NC fp32 per-channel constants from the LDS table
template <int NC> __device__ __forceinline__ void ldc(const float *p, float (&v)[NC]) {
#pragma unroll
    for (int i = 0; i < NC / 4; ++i) { const float4 a = *reinterpret_cast<const float4 *>(p + 4 * i); v[4 * i] = a.x; v[4 * i + 1] = a.y; v[4 * i + 2] = a.z; v[4 * i + 3] = a.w; }
}
// NC channels (2 NC bytes) of a row-major LDS image row whose chunks sit at `base` (16-byte units), logical chunk -> physical by `phys`
template <int NC> __device__ __forceinline__ void lds_put(uint4 *img, int chunk_phys, int half, const Raw<NC> &r) {
    if constexpr (NC == 8) img[chunk_phys] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    else reinterpret_cast<uint2 *>(img + chunk_phys)[half] = make_uint2(r.v[0], r.v[1]);
}
template <int NC> __device__ __forceinline__ Raw<NC> lds_get(const uint4 *img, int chunk_phys, int half) {
    Raw<NC> r;
    if constexpr (NC == 8) { const uint4 t = img[chunk_phys]; r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; }
    else { const uint2 t = reinterpret_cast<const uint2 *>(img + chunk_phys)[half]; r.v[0] = t.x; r.v[1] = t.y; }
    return r;
}

template <int C>
__global__ __launch_bounds__(FB_THREADS, 1) void bwd_fused_bf16_kernel(FusedBwdBf16P p) {
    extern __shared__ uint4 smem[];
    constexpr int NCX = C / 8;                 // 16-byte chunks per x row (16 or 8)
    constexpr int NT = C / 32;                 // 32-channel tiles of the layer input
    constexpr int XC = C / 16;                 // x channels per lane and row: 8 (16-byte accesses) or 4 (8-byte)
    constexpr int GC = 4;                      // g channels per lane and row
    constexpr int WAVE_SZ = 256 + 256 + 4 * C + 32 * NCX;   // uint4 per wave: Gr [32][8], Gt [64][4], At [C][4], Xr [32][NCX]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint4 *Ws = smem;                          // [C][8]: row 32ct + m = W^T row of channel 32ct + w_perm(m), chunks along k
    uint4 *Gr = smem + 8 * C + wave * WAVE_SZ; // g^y row-major          (dgrad B operand)
    uint4 *Gt = Gr + 256;                      // g^y transposed         (wgrad A operand)
    uint4 *At = Gt + 256;                      // a transposed           (wgrad B operand)
    uint4 *Dr = At;                            // [32][NCX] packed D, after the wgrad MFMAs have consumed At (4C = 32 NCX chunks)
    uint4 *Xr = At + 4 * C;                    // [32][NCX] raw x, row-major, for the epilogue
    float *tab = reinterpret_cast<float *>(smem + 8 * C + 4 * WAVE_SZ);     // gA, gB, gC [64]; sa, sb, xp, xq [C]

    for (int i = tid; i < 8 * C; i += FB_THREADS) {
        const int prow = i >> 3, kc = i & 7;
        const int c = (prow & ~31) + w_perm(prow & 31);
        float f[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] = p.w[(size_t)(8 * kc + q) * C + c];
        Ws[ch8(prow, kc)] = bf_pack8(f);
    }
    if (tid < FB_K) {
        const int ch = tid;
        float gA, gB, gC;
        bn_bwd_consts(p.g, ch, gA, gB, gC);
        tab[ch] = gA; tab[64 + ch] = gB; tab[128 + ch] = gC;
    }
    if (tid < C) {
        const int ch = tid;
        const float mu = p.e_coef[ch], sc = p.e_coef[C + ch], be = p.e_coef[2 * C + ch], is = p.e_mi[C + ch];
        float *tx = tab + 192;
        tx[ch] = sc; tx[C + ch] = be - mu * sc; tx[2 * C + ch] = is; tx[3 * C + ch] = -mu * is;
    }
    __syncthreads();                           // the only block barrier before the final reduction

    // lane roles.  blocks: lane (cq, rq) owns channels [NC*cq, NC*cq + NC) of rows rq + 4j, j = 0..7 (16 lanes = one whole row)
    const int cq_ = lane & 15, rq_ = lane >> 4;
    const int n_ = lane & 31, h_ = lane >> 5;                      // MFMA roles
    const float slope = p.e_slope;

    i2p_f32x16 accw[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) accw[mt][nt][e] = 0.f;
    float s1[XC], s2[XC];
#pragma unroll
    for (int q = 0; q < XC; ++q) { s1[q] = 0.f; s2[q] = 0.f; }

    Raw<GC> G[8], Y[8];
    Raw<XC> X[8];
    const long long last_row = p.rows - 1;
    const char *gzb = reinterpret_cast<const char *>(p.gz), *yb = reinterpret_cast<const char *>(p.y), *xb = reinterpret_cast<const char *>(p.x);

    // ---- requests of a strip: rows rq + 4j; every instruction covers 4 consecutive rows of its tensor --------------------------
    auto load_strip = [&](unsigned goff, unsigned xoff) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            G[j] = ld_raw<GC>(gzb + goff + j * (4 * FB_K * 2));
            Y[j] = ld_raw<GC>(yb + goff + j * (4 * FB_K * 2));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) X[j] = ld_raw<XC>(xb + xoff + j * (4 * C * 2));
    };
    // partial last strip: every row clamped to the tensor's last row (its values are masked out in the staging phase)
    auto load_tail = [&](long long row0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            long long r = row0 + rq_ + 4 * j; if (r > last_row) r = last_row;
            G[j] = ld_raw<GC>(p.gz + (size_t)r * FB_K + GC * cq_);
            Y[j] = ld_raw<GC>(p.y + (size_t)r * FB_K + GC * cq_);
            X[j] = ld_raw<XC>(p.x + (size_t)r * C + XC * cq_);
        }
    };

    // ---- one strip: staging -> (request the next strip) -> dW MFMAs -> D MFMAs -> epilogue ----------------------------------
    // nvalid < 32 only in the TAIL instantiation
    auto strip = [&](auto prefetch_tag, auto tail_tag, long long row0, int nvalid, unsigned goff_next, unsigned xoff_next) {
        constexpr bool PREFETCH = decltype(prefetch_tag)::value, TAIL = decltype(tail_tag)::value;
        // the LDS addresses and per-channel constants of a strip are re-derived from the lane id per strip: hoisted out of the
        // loop (~100 address registers + 70 constants) they cost more registers than they save instructions
        int cq = cq_, rq = rq_, n = n_, h = h_;
        asm volatile("" : "+v"(cq), "+v"(rq), "+v"(n), "+v"(h));
        // -- staging: g^y -> Gr (row-major, 8 bytes per lane and row) + Gt (transposed: chunk rq of channel k = rows rq + 4j) ----
        {
            float cA[GC], cB[GC], cC[GC];
            ldc<GC>(tab + GC * cq, cA); ldc<GC>(tab + 64 + GC * cq, cB); ldc<GC>(tab + 128 + GC * cq, cC);
            unsigned tp[GC][4];
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                float g0[GC], g1[GC], y0[GC], y1[GC];
                unpack<GC>(G[2 * jp], g0); unpack<GC>(Y[2 * jp], y0); unpack<GC>(G[2 * jp + 1], g1); unpack<GC>(Y[2 * jp + 1], y1);
                const int r0 = rq + 4 * (2 * jp), r1 = r0 + 4;
#pragma unroll
                for (int q = 0; q < GC; ++q) {
                    g0[q] = __builtin_fmaf(cA[q], g0[q], __builtin_fmaf(cB[q], y0[q], cC[q]));
                    g1[q] = __builtin_fmaf(cA[q], g1[q], __builtin_fmaf(cB[q], y1[q], cC[q]));
                    if (TAIL) { g0[q] = r0 < nvalid ? g0[q] : 0.f; g1[q] = r1 < nvalid ? g1[q] : 0.f; }
                }
                lds_put<GC>(Gr, ch8(r0, cq >> 1), cq & 1, pack<GC>(g0));
                lds_put<GC>(Gr, ch8(r1, cq >> 1), cq & 1, pack<GC>(g1));
#pragma unroll
                for (int q = 0; q < GC; ++q) tp[q][jp] = bf_pack2(g0[q], g1[q]);
                __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise unpacks every raw register up front: 2x the registers)
            }
#pragma unroll
            for (int q = 0; q < GC; ++q) Gt[ch4(GC * cq + q, rq)] = make_uint4(tp[q][0], tp[q][1], tp[q][2], tp[q][3]);
        }
        // -- staging: a = act(bn(x)) -> At (transposed), raw x -> Xr ------------------------------------------------------------
        {
            float sa[XC], sb[XC];
            ldc<XC>(tab + 192 + XC * cq, sa); ldc<XC>(tab + 192 + C + XC * cq, sb);
            unsigned tp[XC][4];
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                float a0[XC], a1[XC];
                unpack<XC>(X[2 * jp], a0); unpack<XC>(X[2 * jp + 1], a1);
                const int r0 = rq + 4 * (2 * jp), r1 = r0 + 4;
                if constexpr (XC == 8) { lds_put<XC>(Xr, (r0 << 4) + cq, 0, X[2 * jp]); lds_put<XC>(Xr, (r1 << 4) + cq, 0, X[2 * jp + 1]); }
                else { lds_put<XC>(Xr, (r0 << 3) + (cq >> 1), cq & 1, X[2 * jp]); lds_put<XC>(Xr, (r1 << 3) + (cq >> 1), cq & 1, X[2 * jp + 1]); }
#pragma unroll
                for (int q = 0; q < XC; ++q) {
                    a0[q] = bf_act(bf_bnz(a0[q], sa[q], sb[q]), slope);
                    a1[q] = bf_act(bf_bnz(a1[q], sa[q], sb[q]), slope);
                    if (TAIL) { a0[q] = r0 < nvalid ? a0[q] : 0.f; a1[q] = r1 < nvalid ? a1[q] : 0.f; }
                    tp[q][jp] = bf_pack2(a0[q], a1[q]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < XC; ++q) At[ch4(XC * cq + q, rq)] = make_uint4(tp[q][0], tp[q][1], tp[q][2], tp[q][3]);
        }
        // -- the next strip's 16 KB leave now and land under everything below ---------------------------------------------------
        if constexpr (PREFETCH) load_strip(goff_next, xoff_next);
        __builtin_amdgcn_sched_barrier(0);
        wave_sync_lds();
        // -- dW += g^y^T a: contraction over the strip's rows (chunk r4 of a transposed image = rows r4 + 4j) -------------------
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            i2p_bf16x8 av[2], bv[NT];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) av[mt] = __builtin_bit_cast(i2p_bf16x8, Gt[ch4(32 * mt + n, 2 * s + h)]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = __builtin_bit_cast(i2p_bf16x8, At[ch4(32 * nt + n, 2 * s + h)]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) accw[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[mt], bv[nt], accw[mt][nt], 0, 0, 0);
        }
        wave_sync_lds();                       // At is dead: the packed D strip takes its place
        // -- D^T = W^T g^y^T, all input-channel tiles at once (independent MFMA chains); fragments -> Dr as bf16 rows ------------
        {
            i2p_f32x16 d[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) d[t][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const i2p_bf16x8 bb = __builtin_bit_cast(i2p_bf16x8, Gr[ch8(n, 2 * ks + h)]);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const i2p_bf16x8 wa = __builtin_bit_cast(i2p_bf16x8, Ws[ch8(32 * t + n, 2 * ks + h)]);
                    d[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, bb, d[t], 0, 0, 0);
                }
            }
            // lane (n, h): channels 32t + 16h + [0, 16) of row n
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float lo[8], hi[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { lo[q] = d[t][q]; hi[q] = d[t][8 + q]; }
                if constexpr (NCX == 16) { Dr[ch16(n, 4 * t + 2 * h)] = bf_pack8(lo); Dr[ch16(n, 4 * t + 2 * h + 1)] = bf_pack8(hi); }
                else { Dr[ch8(n, 4 * t + 2 * h)] = bf_pack8(lo); Dr[ch8(n, 4 * t + 2 * h + 1)] = bf_pack8(hi); }
            }
        }
        wave_sync_lds();
        // -- epilogue in the block layout: activation derivative, rounding, statistics, stores (4 whole rows per instruction) -----
        {
            float sa[XC], sb[XC], xp[XC], xq[XC];
            ldc<XC>(tab + 192 + XC * cq, sa); ldc<XC>(tab + 192 + C + XC * cq, sb); ldc<XC>(tab + 192 + 2 * C + XC * cq, xp); ldc<XC>(tab + 192 + 3 * C + XC * cq, xq);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = rq + 4 * j;
                float f[XC], xv[XC];
                if constexpr (XC == 8) { unpack<XC>(lds_get<XC>(Dr, ch16(r, cq), 0), f); unpack<XC>(lds_get<XC>(Xr, (r << 4) + cq, 0), xv); }
                else { unpack<XC>(lds_get<XC>(Dr, ch8(r, cq >> 1), cq & 1), f); unpack<XC>(lds_get<XC>(Xr, (r << 3) + (cq >> 1), cq & 1), xv); }
                if (!TAIL || r < nvalid) {
#pragma unroll
                    for (int q = 0; q < XC; ++q) {
                        f[q] = bf_bnz(xv[q], sa[q], sb[q]) > 0.f ? f[q] : f[q] * slope;
                        f[q] = bf_round(f[q]);
                        s1[q] += f[q]; s2[q] = __builtin_fmaf(f[q], __builtin_fmaf(xv[q], xp[q], xq[q]), s2[q]);
                    }
                    st_raw<XC>(p.gz_in + (size_t)(row0 + r) * C + XC * cq, pack<XC>(f));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        wave_sync_lds();                       // Dr (= At), Xr, Gr, Gt are free for the next staging phase
    };

    const long long nfull = p.rows / FB_ROWS;
    const long long stride = (long long)gridDim.x * 4;
    const long long first = (long long)blockIdx.x * 4 + wave;
    const int n_mine = __builtin_amdgcn_readfirstlane(first < nfull ? (int)((nfull - first + stride - 1) / stride) : 0);
    std::true_type yes; std::false_type no;
    if (n_mine > 0) {
        // byte offsets (32 bits, launcher: tensors < 4 GB) of this lane's first row of the strip being requested
        unsigned goff = (unsigned)((((size_t)first * FB_ROWS + rq_) * FB_K + GC * cq_) * 2);
        unsigned xoff = (unsigned)((((size_t)first * FB_ROWS + rq_) * C + XC * cq_) * 2);
        const unsigned g_step = __builtin_amdgcn_readfirstlane((unsigned)(stride * FB_ROWS * FB_K * 2));
        const unsigned x_step = __builtin_amdgcn_readfirstlane((unsigned)(stride * FB_ROWS * C * 2));
        load_strip(goff, xoff);
        long long row0 = first * FB_ROWS;
        for (int k = 0; k + 1 < n_mine; ++k) {
            goff += g_step; xoff += x_step;
            strip(yes, no, row0, FB_ROWS, goff, xoff);
            row0 += stride * FB_ROWS;
        }
        strip(no, no, row0, FB_ROWS, 0u, 0u);
    }
    // the rows behind the last full strip (at most one partial strip in the whole grid): the last wave of the last block
    if (nfull * FB_ROWS < p.rows && blockIdx.x == gridDim.x - 1 && wave == 3) {
        load_tail(nfull * FB_ROWS);
        strip(no, yes, nfull * FB_ROWS, (int)(p.rows - nfull * FB_ROWS), 0u, 0u);
    }

    // ---- statistics: lanes with equal cq own the same XC channels ----------------------------------------------------------------
    if (p.sums) {
#pragma unroll
        for (int q = 0; q < XC; ++q) {
            double a = (double)s1[q], b = (double)s2[q];
            a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
            a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
            if (lane < 16) {
                double *rep = p.sums + (size_t)((blockIdx.x * 4 + wave) % REP) * 2 * C;
                atomicAdd(rep + XC * cq_ + q, a); atomicAdd(rep + C + XC * cq_ + q, b);
            }
        }
    }
    // ---- dW: the four waves add through LDS in a fixed order; D tile (mt, nt): col = lane & 31 -> c, reg e -> k ----------------
    __syncthreads();
    float *red = reinterpret_cast<float *>(smem);
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = 32 * mt + (e & 3) + 8 * (e >> 2) + 4 * h_;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float *dst = red + (size_t)k * C + 32 * nt + n_;
                        *dst = (w > 0 ? *dst : 0.f) + accw[mt][nt][e];
                    }
                }
        }
        __syncthreads();
    }
    float *out = p.dw_partial + (size_t)blockIdx.x * FB_K * C;
    for (int t = tid; t < FB_K * C / 4; t += FB_THREADS)
        *reinterpret_cast<float4 *>(out + 4 * t) = *reinterpret_cast<const float4 *>(red + 4 * t);
}

// =====================================================================================================================
// The two-source layer 64 + 64 -> 128 (position encoding | mlp1 output -> first layer of mlp2, PPBackbone_center.py:418-426) in
// one pass: gz, y [rows,128], xa, xb, e_add [rows,64] read once, dL/dz of both sources written once — rows*1152 B against
// rows*1728 B for rg_dgrad_kernel<4> + wreg_wgrad_bf16_kernel<128,128,two>.  K = 128 output channels make the weight gradient
// 128 x 128 = ALL 256 accumulator registers of a wave, so everything else has to live in the architectural half: 16-row strips
// (56 registers in flight per wave), the input gradient on v_mfma_f32_16x16x32_bf16 (D^T tiles of 16 channels x 16 rows: 32
// accumulator registers instead of 64 half-empty ones), raw xa | xb and e_add parked in LDS for the epilogue.  LDS images use
// padded pitches (272 B for 256-byte rows, 48 B for the 32-byte rows of the transposed images: conflict-free operand reads, one
// address register per image).  Contraction slot (kh, pos) of the weight gradient's single 16-row step <-> strip row
// 2 kh + (pos >> 2) + 4 (pos & 3): a lane's four rows rq + 4j are one 8-byte half of a transposed chunk.
// =====================================================================================================================
using i2p_f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int F2_ROWS = 16, F2_K = 128, F2_C = 128, F2_H = 64;
constexpr int F2_PW = 272, F2_PT = 48, F2_PE = 144;
constexpr int F2_WS = 128 * F2_PW;                       // W^T image
constexpr int F2_GR = F2_ROWS * F2_PW, F2_GT = F2_K * F2_PT, F2_AT = F2_C * F2_PT, F2_XR = F2_ROWS * F2_PW, F2_ER = F2_ROWS * F2_PE;
constexpr int F2_WAVE = F2_GR + F2_GT + F2_AT + F2_XR + F2_ER;
constexpr int F2_TAB = (3 * F2_K + 4 * F2_C) * 4;
static_assert(F2_AT >= F2_ROWS * F2_PW, "the packed D strip re-uses the transposed input image");

struct Fused2P {
    long long rows;              // multiple of 16
    const bf16_t *gz, *y;        // [rows, 128]
    BnBwdSrc g;                  // the BN behind (128 channels)
    const bf16_t *xa, *xb;       // [rows, 64] pre-BN tensors of the two sources
    const float *coef_a, *mi_a, *coef_b, *mi_b;
    float slope_a, slope_b;
    const bf16_t *e_add;         // [rows, 64] added to source b's gradient before its activation derivative
    const float *w;              // [128][128]: columns [0,64) source a, [64,128) source b
    bf16_t *gz_a, *gz_b;         // [rows, 64]
    double *sums_a, *sums_b;     // [REP][2*64]
    float *dw_partial;           // [grid][128*128]
};

struct Raw2 { uint4 g[4], y[4]; uint2 xa[4], xb[4], e[4]; };

__device__ __forceinline__ uint2 ld_u2_stream(const void *ptr) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 t = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(ptr));
    return make_uint2(t.x, t.y);
}
__device__ __forceinline__ void st_u2_stream(void *ptr, const uint2 &v) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 t = {v.x, v.y};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x2 *>(ptr));
}

__global__ __launch_bounds__(FB_THREADS, 1) void bwd_fused2_bf16_kernel(Fused2P p) {
    extern __shared__ uint4 smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char *Ws = reinterpret_cast<char *>(smem);           // row 16t + m = W^T row of channel 32 (m >> 2) + 4t + (m & 3), 16 chunks along k
    char *Wv = Ws + F2_WS + wave * F2_WAVE;
    char *Gr = Wv, *Gt = Gr + F2_GR, *At = Gt + F2_GT, *Dr = At, *Xr = At + F2_AT, *Er = Xr + F2_XR;
    float *tab = reinterpret_cast<float *>(Ws + F2_WS + 4 * F2_WAVE);        // gA, gB, gC [128]; sa, sb, xp, xq [128] (a | b)

    for (int i = tid; i < 128 * 16; i += FB_THREADS) {
        const int prow = i >> 4, kc = i & 15;
        const int t = prow >> 4, m = prow & 15;
        const int c = 32 * (m >> 2) + 4 * t + (m & 3);
        float f[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] = p.w[(size_t)(8 * kc + q) * F2_C + c];
        *reinterpret_cast<uint4 *>(Ws + prow * F2_PW + kc * 16) = bf_pack8(f);
    }
    if (tid < F2_K) {
        const int ch = tid, K = F2_K;
        float gA, gB, gC;
        bn_bwd_consts(p.g, ch, gA, gB, gC);
        tab[ch] = gA; tab[K + ch] = gB; tab[2 * K + ch] = gC;
        const bool b = ch >= F2_H;
        const float *cf = b ? p.coef_b : p.coef_a, *mi = b ? p.mi_b : p.mi_a;
        const int c = b ? ch - F2_H : ch;
        const float emu = cf[c], esc = cf[F2_H + c], ebe = cf[2 * F2_H + c], eis = mi[F2_H + c];
        float *tx = tab + 3 * K;
        tx[ch] = esc; tx[F2_C + ch] = ebe - emu * esc; tx[2 * F2_C + ch] = eis; tx[3 * F2_C + ch] = -emu * eis;
    }
    __syncthreads();

    const int cq_ = lane & 15, rq_ = lane >> 4;
    const int n_ = lane & 31, h_ = lane >> 5;

    i2p_f32x16 accw[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) accw[mt][nt][e] = 0.f;
    float s1a[4], s2a[4], s1b[4], s2b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { s1a[q] = 0.f; s2a[q] = 0.f; s1b[q] = 0.f; s2b[q] = 0.f; }

    Raw2 R;
    const char *gzb = reinterpret_cast<const char *>(p.gz), *yb = reinterpret_cast<const char *>(p.y);
    const char *xab = reinterpret_cast<const char *>(p.xa), *xbb = reinterpret_cast<const char *>(p.xb), *eb = reinterpret_cast<const char *>(p.e_add);
    auto load_strip = [&](unsigned goff, unsigned xoff) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            R.g[j] = ld_u4_stream(gzb + goff + j * (4 * F2_K * 2));
            R.y[j] = ld_u4_stream(yb + goff + j * (4 * F2_K * 2));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            R.xa[j] = ld_u2_stream(xab + xoff + j * (4 * F2_H * 2));
            R.xb[j] = ld_u2_stream(xbb + xoff + j * (4 * F2_H * 2));
            R.e[j] = ld_u2_stream(eb + xoff + j * (4 * F2_H * 2));
        }
    };

    auto strip = [&](auto prefetch_tag, long long row0, unsigned goff_next, unsigned xoff_next) {
        constexpr bool PREFETCH = decltype(prefetch_tag)::value;
        int cq = cq_, rq = rq_, n = n_, h = h_;
        asm volatile("" : "+v"(cq), "+v"(rq), "+v"(n), "+v"(h));
        char *t_half = reinterpret_cast<char *>(0) + (rq >> 1) * 16 + (rq & 1) * 8;      // this lane's half chunk in a transposed row
        const int toff = (int)(t_half - reinterpret_cast<char *>(0));
        // -- staging: g^y -> Gr (row-major) + Gt (transposed) ---------------------------------------------------------------------
        {
            float cA[8], cB[8], cC[8];
            ldc<8>(tab + 8 * cq, cA); ldc<8>(tab + F2_K + 8 * cq, cB); ldc<8>(tab + 2 * F2_K + 8 * cq, cC);
            unsigned tp[8][2];
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                float g0[8], g1[8], y0[8], y1[8];
                bf_unpack8(R.g[2 * jp], g0); bf_unpack8(R.y[2 * jp], y0); bf_unpack8(R.g[2 * jp + 1], g1); bf_unpack8(R.y[2 * jp + 1], y1);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    g0[q] = __builtin_fmaf(cA[q], g0[q], __builtin_fmaf(cB[q], y0[q], cC[q]));
                    g1[q] = __builtin_fmaf(cA[q], g1[q], __builtin_fmaf(cB[q], y1[q], cC[q]));
                }
                *reinterpret_cast<uint4 *>(Gr + (rq + 4 * (2 * jp)) * F2_PW + cq * 16) = bf_pack8(g0);
                *reinterpret_cast<uint4 *>(Gr + (rq + 4 * (2 * jp + 1)) * F2_PW + cq * 16) = bf_pack8(g1);
#pragma unroll
                for (int q = 0; q < 8; ++q) tp[q][jp] = bf_pack2(g0[q], g1[q]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<uint2 *>(Gt + (8 * cq + q) * F2_PT + toff) = make_uint2(tp[q][0], tp[q][1]);
        }
        // -- staging: a = act(bn(x)) of both sources -> At (transposed); raw xa | xb -> Xr, raw e_add -> Er -----------------------
#pragma unroll
        for (int src = 0; src < 2; ++src) {
            float sa[4], sb[4];
            ldc<4>(tab + 3 * F2_K + 64 * src + 4 * cq, sa); ldc<4>(tab + 3 * F2_K + F2_C + 64 * src + 4 * cq, sb);
            const float slope = src ? p.slope_b : p.slope_a;
            float a[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint2 raw = src ? R.xb[j] : R.xa[j];
                *reinterpret_cast<uint2 *>(Xr + (rq + 4 * j) * F2_PW + 128 * src + cq * 8) = raw;
                if (src) *reinterpret_cast<uint2 *>(Er + (rq + 4 * j) * F2_PE + cq * 8) = R.e[j];
                const float x0 = bf_lo(raw.x), x1 = bf_hi(raw.x), x2 = bf_lo(raw.y), x3 = bf_hi(raw.y);
                a[j][0] = bf_act(bf_bnz(x0, sa[0], sb[0]), slope); a[j][1] = bf_act(bf_bnz(x1, sa[1], sb[1]), slope);
                a[j][2] = bf_act(bf_bnz(x2, sa[2], sb[2]), slope); a[j][3] = bf_act(bf_bnz(x3, sa[3], sb[3]), slope);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<uint2 *>(At + (64 * src + 4 * cq + q) * F2_PT + toff) = make_uint2(bf_pack2(a[0][q], a[1][q]), bf_pack2(a[2][q], a[3][q]));
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (PREFETCH) load_strip(goff_next, xoff_next);
        __builtin_amdgcn_sched_barrier(0);
        wave_sync_lds();
        // -- dW += g^y^T a: ONE 16-row contraction step ---------------------------------------------------------------------------
        {
            i2p_bf16x8 av[4], bv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                av[t] = __builtin_bit_cast(i2p_bf16x8, *reinterpret_cast<const uint4 *>(Gt + (32 * t + n) * F2_PT + h * 16));
                bv[t] = __builtin_bit_cast(i2p_bf16x8, *reinterpret_cast<const uint4 *>(At + (32 * t + n) * F2_PT + h * 16));
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) accw[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[mt], bv[nt], accw[mt][nt], 0, 0, 0);
        }
        wave_sync_lds();                       // At is dead: the packed D strip takes its place
        // -- D^T = W^T g^y^T on 16x16x32: lane (row n16, kq): tile t, register e = channel 32 kq + 4t + e ------------------------
        {
            const int n16 = lane & 15, kq = lane >> 4;
            int n16v = n16, kqv = kq;
            asm volatile("" : "+v"(n16v), "+v"(kqv));
            i2p_f32x4 d[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) d[t] = i2p_f32x4{0.f, 0.f, 0.f, 0.f};
            const char *gr_r = Gr + n16v * F2_PW + kqv * 16, *ws_r = Ws + n16v * F2_PW + kqv * 16;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const i2p_bf16x8 bb = __builtin_bit_cast(i2p_bf16x8, *reinterpret_cast<const uint4 *>(gr_r + ks * 64));
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const i2p_bf16x8 wa = __builtin_bit_cast(i2p_bf16x8, *reinterpret_cast<const uint4 *>(ws_r + t * (16 * F2_PW) + ks * 64));
                    d[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, bb, d[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float v8[8] = {d[2 * u][0], d[2 * u][1], d[2 * u][2], d[2 * u][3], d[2 * u + 1][0], d[2 * u + 1][1], d[2 * u + 1][2], d[2 * u + 1][3]};
                *reinterpret_cast<uint4 *>(Dr + n16v * F2_PW + (4 * kqv + u) * 16) = bf_pack8(v8);
            }
        }
        wave_sync_lds();
        // -- epilogue: lane (cq, rq): channels 4cq..4cq+3 of each source, rows rq + 4j ----------------------------------------------
#pragma unroll
        for (int src = 0; src < 2; ++src) {
            float sa[4], sb[4], xp[4], xq[4];
            const float *tx = tab + 3 * F2_K + 64 * src + 4 * cq;
            ldc<4>(tx, sa); ldc<4>(tx + F2_C, sb); ldc<4>(tx + 2 * F2_C, xp); ldc<4>(tx + 3 * F2_C, xq);
            const float slope = src ? p.slope_b : p.slope_a;
            bf16_t *dst = src ? p.gz_b : p.gz_a;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = rq + 4 * j;
                const uint2 dv = *reinterpret_cast<const uint2 *>(Dr + r * F2_PW + 128 * src + cq * 8);
                const uint2 xr = *reinterpret_cast<const uint2 *>(Xr + r * F2_PW + 128 * src + cq * 8);
                float f[4] = {bf_lo(dv.x), bf_hi(dv.x), bf_lo(dv.y), bf_hi(dv.y)};
                const float xv[4] = {bf_lo(xr.x), bf_hi(xr.x), bf_lo(xr.y), bf_hi(xr.y)};
                if (src) {
                    const uint2 ev = *reinterpret_cast<const uint2 *>(Er + r * F2_PE + cq * 8);
                    f[0] += bf_lo(ev.x); f[1] += bf_hi(ev.x); f[2] += bf_lo(ev.y); f[3] += bf_hi(ev.y);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f[q] = bf_bnz(xv[q], sa[q], sb[q]) > 0.f ? f[q] : f[q] * slope;
                    f[q] = bf_round(f[q]);
                    const float xh = __builtin_fmaf(xv[q], xp[q], xq[q]);
                    if (src) { s1b[q] += f[q]; s2b[q] = __builtin_fmaf(f[q], xh, s2b[q]); }
                    else { s1a[q] += f[q]; s2a[q] = __builtin_fmaf(f[q], xh, s2a[q]); }
                }
                st_u2_stream(dst + (size_t)(row0 + r) * F2_H + 4 * cq, make_uint2(bf_pack2(f[0], f[1]), bf_pack2(f[2], f[3])));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        wave_sync_lds();
    };

    const long long nstrips = p.rows / F2_ROWS;
    const long long stride = (long long)gridDim.x * 4;
    const long long first = (long long)blockIdx.x * 4 + wave;
    const int n_mine = __builtin_amdgcn_readfirstlane(first < nstrips ? (int)((nstrips - first + stride - 1) / stride) : 0);
    std::true_type yes; std::false_type no;
    if (n_mine > 0) {
        unsigned goff = (unsigned)((((size_t)first * F2_ROWS + rq_) * F2_K + 8 * cq_) * 2);
        unsigned xoff = (unsigned)((((size_t)first * F2_ROWS + rq_) * F2_H + 4 * cq_) * 2);
        const unsigned g_step = __builtin_amdgcn_readfirstlane((unsigned)(stride * F2_ROWS * F2_K * 2));
        const unsigned x_step = __builtin_amdgcn_readfirstlane((unsigned)(stride * F2_ROWS * F2_H * 2));
        load_strip(goff, xoff);
        long long row0 = first * F2_ROWS;
        for (int k = 0; k + 1 < n_mine; ++k) {
            goff += g_step; xoff += x_step;
            strip(yes, row0, goff, xoff);
            row0 += stride * F2_ROWS;
        }
        strip(no, row0, 0u, 0u);
    }
    // ---- statistics: lanes with equal cq own the same 4 channels of each source -----------------------------------------------
#pragma unroll
    for (int src = 0; src < 2; ++src) {
        double *sums = src ? p.sums_b : p.sums_a;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double a = (double)(src ? s1b[q] : s1a[q]), b = (double)(src ? s2b[q] : s2a[q]);
            a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
            a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
            if (lane < 16) {
                double *rep = sums + (size_t)((blockIdx.x * 4 + wave) % REP) * 2 * F2_H;
                atomicAdd(rep + 4 * cq_ + q, a); atomicAdd(rep + F2_H + 4 * cq_ + q, b);
            }
        }
    }
    // ---- dW: the four waves add through LDS in a fixed order ---------------------------------------------------------------------
    __syncthreads();
    float *red = reinterpret_cast<float *>(smem);
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = 32 * mt + (e & 3) + 8 * (e >> 2) + 4 * h_;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        float *dst = red + (size_t)k * F2_C + 32 * nt + n_;
                        *dst = (w > 0 ? *dst : 0.f) + accw[mt][nt][e];
                    }
                }
        }
        __syncthreads();
    }
    float *out = p.dw_partial + (size_t)blockIdx.x * F2_K * F2_C;
    for (int t = tid; t < F2_K * F2_C / 4; t += FB_THREADS)
        *reinterpret_cast<float4 *>(out + 4 * t) = *reinterpret_cast<const float4 *>(red + 4 * t);
}

template <int C>
int launch(const FusedBwdBf16P &p, unsigned grid, hipStream_t st) {
    constexpr size_t bytes = ((size_t)8 * C + 4 * (512 + 4 * C + 32 * (C / 8))) * sizeof(uint4) + (192 + 4 * C) * sizeof(float);
    static_assert(bytes <= 160 * 1024, "LDS budget");
    static_assert(bytes >= (size_t)FB_K * C * sizeof(float), "dW reduction buffer");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(bwd_fused_bf16_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    i2p_ktime_begin(st);
    hipLaunchKernelGGL(bwd_fused_bf16_kernel<C>, dim3(grid), dim3(FB_THREADS), bytes, st, p);
    i2p_ktime_end(st);
    I2P_RETURN_LAUNCH_STATUS();
}

}  // namespace

bool i2p_bwd_fused_bf16_ok(long long rows, int cin, int cout) {
    const char *e = getenv("I2P_NO_FUSED_BF16");                   // (read per call: the tests compare the two forms in one process)
    if (e && e[0] == '1') return false;
    return rows >= 65536 && cout == FB_K && (cin == 64 || cin == 128) && (unsigned long long)rows * 128ull * 2ull < (1ull << 32);
}

int i2p_bwd_fused_bf16(long long rows, int cin, int cout, const unsigned short *gz, const unsigned short *y, const double *out_dsums,
                       const float *out_coef, const float *out_mi, float *coef8,
                       const unsigned short *x, const float *in_coef, const float *in_mi, float slope_in, const float *w,
                       unsigned short *gz_in, double *in_dsums, float *dw_partial, unsigned grid, void *stream) {
    if (!i2p_bwd_fused_bf16_ok(rows, cin, cout) || !gz || !y || !out_dsums || !out_coef || !out_mi || !x || !in_coef || !in_mi || !w || !gz_in || !dw_partial || grid == 0)
        return I2P_ERR_BAD_ARG;
    FusedBwdBf16P p;
    p.rows = rows; p.gz = gz; p.y = y; p.g = BnBwdSrc{out_dsums, out_coef, out_mi, rows, coef8, FB_K}; p.x = x; p.e_coef = in_coef; p.e_mi = in_mi; p.e_slope = slope_in; p.w = w;
    p.gz_in = gz_in; p.sums = in_dsums; p.dw_partial = dw_partial;
    hipStream_t st = (hipStream_t)stream;
    return cin == 128 ? launch<128>(p, grid, st) : launch<64>(p, grid, st);
}

bool i2p_bwd_fused2_bf16_ok(long long rows, int cin_a, int cin_b, int cout) {
    const char *e = getenv("I2P_NO_FUSED_BF16");
    if (e && e[0] == '1') return false;
    return rows >= 65536 && (rows % F2_ROWS) == 0 && cout == F2_K && cin_a == F2_H && cin_b == F2_H && (unsigned long long)rows * 128ull * 2ull < (1ull << 32);
}

int i2p_bwd_fused2_bf16(long long rows, const unsigned short *gz, const unsigned short *y, const double *out_dsums, const float *out_coef,
                        const float *out_mi, float *coef8, const unsigned short *xa,
                        const float *coef_a, const float *mi_a, float slope_a, const unsigned short *xb, const float *coef_b,
                        const float *mi_b, float slope_b, const unsigned short *e_add, const float *w, unsigned short *gz_a,
                        double *sums_a, unsigned short *gz_b, double *sums_b, float *dw_partial, unsigned grid, void *stream) {
    if (!gz || !y || !out_dsums || !out_coef || !out_mi || !xa || !xb || !coef_a || !mi_a || !coef_b || !mi_b || !e_add || !w || !gz_a || !gz_b || !sums_a || !sums_b ||
        !dw_partial || grid == 0 || (rows % F2_ROWS))
        return I2P_ERR_BAD_ARG;
    Fused2P p;
    p.rows = rows; p.gz = gz; p.y = y; p.g = BnBwdSrc{out_dsums, out_coef, out_mi, rows, coef8, F2_K}; p.xa = xa; p.xb = xb; p.coef_a = coef_a; p.mi_a = mi_a; p.coef_b = coef_b; p.mi_b = mi_b;
    p.slope_a = slope_a; p.slope_b = slope_b; p.e_add = e_add; p.w = w; p.gz_a = gz_a; p.gz_b = gz_b; p.sums_a = sums_a; p.sums_b = sums_b;
    p.dw_partial = dw_partial;
    constexpr size_t bytes = (size_t)F2_WS + 4 * (size_t)F2_WAVE + F2_TAB;
    static_assert(bytes <= 160 * 1024, "LDS budget");
    static_assert(bytes >= (size_t)F2_K * F2_C * sizeof(float), "dW reduction buffer");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(bwd_fused2_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(bwd_fused2_bf16_kernel, dim3(grid), dim3(FB_THREADS), bytes, (hipStream_t)stream, p);
    I2P_RETURN_LAUNCH_STATUS();
}
