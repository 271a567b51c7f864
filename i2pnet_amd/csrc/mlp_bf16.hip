// Fused per-point linear layers with bf16 activation storage and bf16 MFMA (fp32 accumulate) for gfx950 —
// the bf16 counterparts of mlp.hip's lin_fwd / lin_bwd / pair kernels (BASELINE.json configs[2], configs[4]:
// "bf16 ... cost-volume HIP kernels with MFMA point-MLP").  Same layer algebra as mlp.hip:
//
//   forward :  Y = act_in(bn_in(X)) . W^T   (+ pair-mode factors / biases),  sums += {sum Y, sum Y^2} (fp64)
//   dgrad   :  dZin = (g^y . W) * act_in'(z_in),  g^y = BN-backward(dZ, Y) formed on load
//   wgrad   :  dW = g^y^T . X'
//
// but X / Y / dZ live in HBM as bf16 and the contraction runs on v_mfma_f32_32x32x16_bf16, at 16x the fp32 MFMA
// rate: these kernels are HBM-bound (a 128->128 layer moves rows*512 B against 32.8 kFLOP per row: 64 FLOP/B, the
// bf16 ridge is ~310 FLOP/B), so the design goal is bytes in flight and fully coalesced 16-byte accesses, not MFMA
// issue slots.
//
// Row-GEMM kernels (forward, dgrad): 256 threads, W stationary in LDS as bf16, every wave owns 32-row strips end to
// end (no block barrier in the loop): coalesced 16-byte loads (one strip ahead, in registers) -> BN/activation in
// fp32 -> bf16 -> the wave's LDS strip -> MFMA in the TRANSPOSED formulation D[cout][row] = W . X^T with the
// output-channel order permuted so that a lane ends up with 16 consecutive channels of one row per tile -> packed
// back through the same LDS strip -> coalesced 16-byte stores, statistics taken from the rounded values.
#include "bf16_common.h"
#include <cstdlib>

namespace {

constexpr int REP = I2P_BN_REPLICAS;
constexpr int RG_THREADS = 256;
constexpr int RG_ROWS = 32;           // rows per strip (one 32x32 MFMA column block)
constexpr int UMAX_BF = 8;            // staging chunks per lane: bf16 input, cin <= 128
constexpr int UMAX_F32 = 10;          // fp32 input, cin <= 160

// D fragment of v_mfma_f32_32x32x16: col = lane&31, row m = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// A-operand row m of a tile is fed with weight row  perm(m) = 16*((m>>2)&1) + (m&3) + 4*(m>>3), so that the 16
// registers of lane-half h hold output channels  tile*32 + 16*h + reg  (16 consecutive channels).
__device__ __forceinline__ int w_perm(int m) { return 16 * ((m >> 2) & 1) + (m & 3) + 4 * (m >> 3); }

__device__ __forceinline__ int ilog2(int v) { return 31 - __clz(v); }

struct FwdP {
    long long rows;
    int cin, cout;               // logical sizes (cout in {16,32,64,128})
    int ncx;                     // input chunks of 8 channels per row, cin rounded up to 16 channels => even
    int cpi_s, cpo_s;            // log2 of the LDS chunk pitch of the input image (>= ncx) / output image (= cout/8)
    const void *x; int x_ld;     // bf16 (XBF16) or fp32 source of channels [0, split)
    const bf16_t *xb; int xb_ld; // second bf16 source: channels [split, cin)   (two-source layer, else nullptr)
    int split;
    const float *coef_a, *coef_b;     // [3][c_src] mean, scale, beta of the BN in front of each source, or nullptr
    float slope_a, slope_b;
    const float *w;              // [cout][cin] fp32
    bf16_t *y;                   // [rows, cout]
    double *sums;                // [REP][2*cout] or nullptr
    // pair mode: input row (b,n,k) = pair_f[b,n,:] * x[b,k,:], y += bias_n[b,n,:] + bias_k[b,k,:]
    const float *pair_f, *bias_n, *bias_k;
    int pN, pM;
};

// weights -> LDS image [32*NT][CP] of bf16 chunks along k (zero padded); transposed: Ws[o][k] = w[k*ld + o]
__device__ __forceinline__ void stage_weights(uint4 *Ws, const float *w, int nout_pad, int nout, int kdim, int ncx, int cps,
                                              bool transposed, int ld, int tid) {
    for (int i = tid; i < nout_pad * ncx; i += RG_THREADS) {
        const int o = i / ncx, c = i - o * ncx;
        float f[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = c * 8 + q;
            f[q] = (o < nout && k < kdim) ? (transposed ? w[(size_t)k * ld + o] : w[(size_t)o * ld + k]) : 0.f;
        }
        Ws[bf_chunk(o, c, cps)] = bf_pack8(f);
    }
}

// lane-task geometry of a 32-row strip image with `nc` chunks per row: task id = lane + 64*u -> (row, chunk)
__device__ __forceinline__ void task_rc(int id, int nc, int nc_shift, int &row, int &c) {
    if (nc_shift >= 0) { row = id >> nc_shift; c = id & (nc - 1); }
    else { row = id / nc; c = id - row * nc; }
}

// (fp32-input instantiations carry twice the prefetch registers: one block per CU, up to 512 VGPRs)
// (b,n,k) of the rows of one strip without per-row 64-bit divisions: one 32-bit division pair per strip (wave-uniform),
// then a row d of the strip wraps at most once (M >= 32)
struct PairStrip { int bn0, k0, b0, n0; };
__device__ __forceinline__ PairStrip pair_strip(long long row0, int N, int M) {
    PairStrip t;
    const unsigned r = (unsigned)row0;
    t.bn0 = (int)(r / (unsigned)M); t.k0 = (int)(r - (unsigned)t.bn0 * (unsigned)M);
    t.b0 = t.bn0 / N; t.n0 = t.bn0 - t.b0 * N;
    return t;
}
__device__ __forceinline__ void pair_row_of(const PairStrip &t, int d, int N, int M, int &bn, int &bk) {
    int k = t.k0 + d;
    const int wrap = k >= M ? 1 : 0;
    k -= wrap * M;
    const int b = t.b0 + ((t.n0 + wrap) >= N ? 1 : 0);
    bn = t.bn0 + wrap; bk = b * M + k;
}

template <int NT, bool XBF16, bool PAIR>
__global__ __launch_bounds__(RG_THREADS, XBF16 ? 2 : 1) void rg_fwd_kernel(FwdP p) {
    extern __shared__ uint4 smem[];
    constexpr int UMAX = (XBF16 || PAIR) ? UMAX_BF : UMAX_F32;
    constexpr int PFN = XBF16 ? UMAX : 2 * UMAX;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cpi_s = p.cpi_s, cpo_s = p.cpo_s;
    const int strip_chunks = RG_ROWS << (cpi_s > cpo_s ? cpi_s : cpo_s);
    uint4 *Ws = smem;                                            // [32*NT][1 << cpi_s]
    uint4 *Ss = smem + ((32 * NT) << cpi_s) + wave * strip_chunks; // this wave's strip

    stage_weights(Ws, p.w, 32 * NT, p.cout, p.cin, p.ncx, cpi_s, false, p.cin, tid);
    __syncthreads();                                             // the only block barrier

    const int ncx = p.ncx, U = ncx >> 1;                         // 32*ncx tasks / 64 lanes
    const int nshift = (ncx & (ncx - 1)) == 0 ? ilog2(ncx) : -1;
    const int nco = p.cout >> 3, OU = nco >= 2 ? nco >> 1 : 1;   // output chunks per row / tasks per lane
    const int oshift = ilog2(nco);
    const int oc = lane & (nco - 1);                             // this lane's output chunk (8 channels) in the store phase

    // BN + activation of the input: this lane's 8 channels are the same in every task when ncx is a power of two
    // (the launcher guarantees that whenever a coefficient array is given)
    float ca[8], cb[8];
    float in_slope = 1.f;
    bool has_coef = false;
    {
        int r0, c0; task_rc(lane, ncx, nshift, r0, c0);
        const int ch = c0 * 8;
        const bool second = XBF16 && p.xb && ch >= p.split;
        const float *cf = second ? p.coef_b : p.coef_a;
        const int ldc = (XBF16 && p.xb) ? (second ? p.cin - p.split : p.split) : p.cin;
        const int cc = second ? ch - p.split : ch;
        in_slope = second ? p.slope_b : p.slope_a;
        has_coef = cf != nullptr;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float a = 1.f, b = 0.f;
            if (cf && cc + q < ldc) { a = cf[ldc + cc + q]; b = cf[2 * ldc + cc + q] - cf[cc + q] * a; }
            ca[q] = a; cb[q] = b;
        }
    }

    const long long last_row = p.rows - 1;
    const long long nstrips = (p.rows + RG_ROWS - 1) / RG_ROWS;
    const long long sstride = (long long)gridDim.x * 4;
    long long strip = (long long)blockIdx.x * 4 + wave;

    uint4 pfb[XBF16 ? UMAX : 1];
    float4 pff[XBF16 ? 1 : PFN];
    (void)pfb; (void)pff;
    // pair mode: a 32-row strip spans at most two points (M >= 32), so the point factor f[b,n,:] of this lane's 8 input
    // channels is fetched ONCE per strip (two candidates) with the pixel rows, instead of one L2 round trip per row
    float4 pf_f[PAIR ? 4 : 1];
    long long pf_bn0 = 0;
    int pf_k0 = 0;
    const long long total_bn = PAIR ? p.rows / p.pM : 0;        // points of the whole batch
    (void)pf_f; (void)pf_bn0; (void)pf_k0; (void)total_bn;
    const int lane_c = (lane & (ncx - 1)) * 8;                   // (pair / bf16 inputs: ncx is a power of two)

    auto fetch = [&](long long st) {
        const long long row0 = st * RG_ROWS;
        PairStrip ps; ps.bn0 = ps.k0 = ps.b0 = ps.n0 = 0;
        if constexpr (PAIR) ps = pair_strip(row0, p.pN, p.pM);
#pragma unroll
        for (int u = 0; u < UMAX; ++u) {
            if (u < U) {
                int r, c; task_rc(lane + 64 * u, ncx, nshift, r, c);
                long long row = row0 + r; if (row > last_row) { r = (int)(last_row - row0); row = last_row; }
                if constexpr (XBF16) {
                    const int ch = c * 8;
                    const bf16_t *src = (p.xb && ch >= p.split) ? p.xb + (size_t)row * p.xb_ld + (ch - p.split)
                                                               : reinterpret_cast<const bf16_t *>(p.x) + (size_t)row * p.x_ld + ch;
                    pfb[u] = ld_u4_stream(src);
                } else {
                    long long src = row;
                    if constexpr (PAIR) { int bn, bk; pair_row_of(ps, r, p.pN, p.pM, bn, bk); src = bk; }
                    const float *xr = reinterpret_cast<const float *>(p.x) + (size_t)src * p.x_ld;
                    const int k0 = c * 8;
                    pff[2 * u] = k0 < p.cin ? *reinterpret_cast<const float4 *>(xr + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
                    pff[2 * u + 1] = k0 + 4 < p.cin ? *reinterpret_cast<const float4 *>(xr + k0 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
        if constexpr (PAIR) {
            pf_bn0 = ps.bn0; pf_k0 = ps.k0;
            const long long bn1 = ps.bn0 + 1 < total_bn ? ps.bn0 + 1 : ps.bn0;
            const float *fa = p.pair_f + (size_t)pf_bn0 * p.cin + lane_c, *fb = p.pair_f + (size_t)bn1 * p.cin + lane_c;
            pf_f[0] = *reinterpret_cast<const float4 *>(fa); pf_f[1] = *reinterpret_cast<const float4 *>(fa + 4);
            pf_f[2] = *reinterpret_cast<const float4 *>(fb); pf_f[3] = *reinterpret_cast<const float4 *>(fb + 4);
        }
    };
    auto commit = [&](long long) {
#pragma unroll
        for (int u = 0; u < UMAX; ++u) {
            if (u < U) {
                int r, c; task_rc(lane + 64 * u, ncx, nshift, r, c);
                float f[8];
                if constexpr (XBF16) bf_unpack8(pfb[u], f);
                else {
                    f[0] = pff[2 * u].x; f[1] = pff[2 * u].y; f[2] = pff[2 * u].z; f[3] = pff[2 * u].w;
                    f[4] = pff[2 * u + 1].x; f[5] = pff[2 * u + 1].y; f[6] = pff[2 * u + 1].z; f[7] = pff[2 * u + 1].w;
                }
                if constexpr (PAIR) {
                    const bool second = pf_k0 + r >= p.pM;           // (rows past the end: values never stored)
                    const float4 f0 = second ? pf_f[2] : pf_f[0], f1 = second ? pf_f[3] : pf_f[1];
                    f[0] *= f0.x; f[1] *= f0.y; f[2] *= f0.z; f[3] *= f0.w; f[4] *= f1.x; f[5] *= f1.y; f[6] *= f1.z; f[7] *= f1.w;
                }
                if (has_coef) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) f[q] = bf_act(bf_bnz(f[q], ca[q], cb[q]), in_slope);
                }
                Ss[bf_chunk(r, c, cpi_s)] = bf_pack8(f);
            }
        }
    };

    double ssum[8], ssq[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { ssum[q] = 0.0; ssq[q] = 0.0; }

    const int n = lane & 31, h = lane >> 5;
    const int wrow0 = w_perm(n);
    const int KS = ncx >> 1;                                     // k-steps of 16

    if (strip < nstrips) fetch(strip);
    for (; strip < nstrips; strip += sstride) {
        const long long row0 = strip * RG_ROWS;
        // pair mode: y = product + bias_n[b,n,:] + bias_k[b,k,:].  The two bias rows of this lane's output row are
        // requested FIRST (they land under the staging work) and become the initial value of the accumulators.
        i2p_f32x16 acc[NT];
        float4 bnv[PAIR ? NT * 4 : 1];
        (void)bnv;
        if constexpr (PAIR) {
            const PairStrip es = pair_strip(row0, p.pN, p.pM);
            int dn = n; if (row0 + dn > last_row) dn = (int)(last_row - row0);
            int pbn, pbk; pair_row_of(es, dn, p.pN, p.pM, pbn, pbk);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int co = t * 32 + 16 * h;
                const float *bnp = p.bias_n + (size_t)pbn * p.cout + co, *bkp = p.bias_k + (size_t)pbk * p.cout + co;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float4 kk = make_float4(0.f, 0.f, 0.f, 0.f); bnv[t * 4 + j] = kk;
                    if (co < p.cout) { kk = *reinterpret_cast<const float4 *>(bkp + 4 * j); bnv[t * 4 + j] = *reinterpret_cast<const float4 *>(bnp + 4 * j); }
                    acc[t][4 * j] = kk.x; acc[t][4 * j + 1] = kk.y; acc[t][4 * j + 2] = kk.z; acc[t][4 * j + 3] = kk.w;
                }
            }
        }
        commit(strip);
        if (strip + sstride < nstrips) fetch(strip + sstride);   // in flight during the MFMA + store phases
        __builtin_amdgcn_wave_barrier();

        if constexpr (PAIR) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[t][4 * j] += bnv[t * 4 + j].x; acc[t][4 * j + 1] += bnv[t * 4 + j].y;
                    acc[t][4 * j + 2] += bnv[t * 4 + j].z; acc[t][4 * j + 3] += bnv[t * 4 + j].w;
                }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        }
        for (int ks = 0; ks < KS; ++ks) {
            const int kc = 2 * ks + h;
            const i2p_bf16x8 xb = __builtin_bit_cast(i2p_bf16x8, Ss[bf_chunk(n, kc, cpi_s)]);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const i2p_bf16x8 wa = __builtin_bit_cast(i2p_bf16x8, Ws[bf_chunk(t * 32 + wrow0, kc, cpi_s)]);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, xb, acc[t], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- fragments -> the wave's strip as bf16 rows: lane (n,h) holds channels t*32 + 16h + [0,16) of row n ----
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int co = t * 32 + 16 * h;
            if (co < p.cout) {
                float v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = acc[t][e];
                float lo8[8], hi8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { lo8[q] = v[q]; hi8[q] = v[8 + q]; }
                Ss[bf_chunk(n, (co >> 3), cpo_s)] = bf_pack8(lo8);
                Ss[bf_chunk(n, (co >> 3) + 1, cpo_s)] = bf_pack8(hi8);
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- coalesced read-back: statistics of the ROUNDED values, 16-byte stores ----------------------------
        float s1[8], s2[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { s1[q] = 0.f; s2[q] = 0.f; }
#pragma unroll
        for (int u = 0; u < 2 * NT; ++u) {
            if (u < OU) {
                const int r = (lane + 64 * u) >> oshift;
                if (r < RG_ROWS && row0 + r <= last_row) {
                    const uint4 v = Ss[bf_chunk(r, oc, cpo_s)];
                    float f[8]; bf_unpack8(v, f);
#pragma unroll
                    for (int q = 0; q < 8; ++q) { s1[q] += f[q]; s2[q] = __builtin_fmaf(f[q], f[q], s2[q]); }
                    st_u4_stream(p.y + (size_t)(row0 + r) * p.cout + oc * 8, v);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) { ssum[q] += (double)s1[q]; ssq[q] += (double)s2[q]; }
        __builtin_amdgcn_wave_barrier();
    }

    if (p.sums) {       // lanes with equal (lane & (nco-1)) own the same 8 channels
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            double a = ssum[q], b = ssq[q];
            for (int off = 32; off >= nco; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
            if (lane < nco) {
                double *rep = p.sums + (size_t)((blockIdx.x * 4 + wave) % REP) * 2 * p.cout;
                atomicAdd(rep + oc * 8 + q, a); atomicAdd(rep + p.cout + oc * 8 + q, b);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// dgrad on the same strip machinery: input image = g^y = BN-backward(gz, y2) of the layer BEHIND formed on load
// (K = that layer's output channels), weights transposed, output = dL/dz of the layer in front (bf16, activation
// derivative + BN-backward statistics in the store phase, optionally split over two destination tensors with a
// second incoming gradient added) or plain fp32 dL/dx for a chain's raw input.
// ---------------------------------------------------------------------------------------------------------------
struct DgradP {
    long long rows;
    int kdim, cout;              // kdim = channels of gz / y2 (layer output), cout = channels of the result (layer input)
    int ncx, cpi_s, cpo_s;
    const bf16_t *gz, *y2;       // [rows, kdim]
    const float *g_coef;         // [6][kdim] m1, m2, scale, mean, invstd, beta of the BN behind, or nullptr (gz is dL/dy)
    float g_slope;               // != 1: gz is dL/da of that layer's activation
    const float *w;              // layer weights [kdim][cout_total] fp32 (this launch uses all cout columns)
    // bf16 destination(s): columns [0, split) -> ya (ld split), [split, cout) -> yb (ld cout-split); split == cout: one
    bf16_t *ya, *yb;
    int split;
    const bf16_t *exa, *exb;     // pre-BN tensors in front of each destination (nullptr: no activation / statistics)
    const float *e_coef_a, *e_mi_a, *e_coef_b, *e_mi_b;
    float e_slope_a, e_slope_b;
    const bf16_t *e_add;         // [rows, cout-split] bf16 added to destination b before its activation derivative
    double *sums_a, *sums_b;
    float *y32;                  // fp32 destination [rows, cout] (OUT32 instantiation; no activation, no statistics)
};

template <int NT, bool OUT32>
__global__ __launch_bounds__(RG_THREADS, 1) void rg_dgrad_kernel(DgradP p) {
    extern __shared__ uint4 smem[];
    constexpr int UMAX = UMAX_BF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cpi_s = p.cpi_s, cpo_s = p.cpo_s;
    const int strip_chunks = RG_ROWS << (cpi_s > cpo_s ? cpi_s : cpo_s);
    uint4 *Ws = smem;
    uint4 *Ss = smem + ((32 * NT) << cpi_s) + wave * strip_chunks;

    stage_weights(Ws, p.w, 32 * NT, p.cout, p.kdim, p.ncx, cpi_s, true, p.cout, tid);
    __syncthreads();

    const int ncx = p.ncx, U = ncx >> 1;                         // ncx = kdim/8: 2, 4, 8 or 16
    const int nshift = ilog2(ncx);
    const int nco = p.cout >> 3, OU = nco >= 2 ? nco >> 1 : 1;
    const int oshift = ilog2(nco);
    const int oc = lane & (nco - 1);
    const int ic = lane & (ncx - 1);                             // this lane's input chunk: the same in every task

    // load side: g^y = A*gz' + (B*y + C), gz' = gz * act'(za*y + zb)
    float gA[8], gB[8], gC[8], za[8], zb[8];
    const bool has_g = p.g_coef != nullptr;
    const bool g_act = has_g && p.g_slope != 1.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        gA[q] = 1.f; gB[q] = 0.f; gC[q] = 0.f; za[q] = 1.f; zb[q] = 0.f;
        if (has_g) {
            const int ch = ic * 8 + q, K = p.kdim;
            const float m1 = p.g_coef[ch], m2 = p.g_coef[K + ch], sc = p.g_coef[2 * K + ch], mu = p.g_coef[3 * K + ch],
                        is = p.g_coef[4 * K + ch], be = p.g_coef[5 * K + ch];
            gA[q] = sc; gB[q] = -(sc * m2) * is; gC[q] = -(sc * m1) - gB[q] * mu;
            za[q] = sc; zb[q] = be - mu * sc;
        }
    }
    // store side (bf16 destinations): z = ea*ex + eb (sign), xhat = ex*xp + xq
    const bool dst_b = !OUT32 && p.yb && oc * 8 >= p.split;
    bf16_t *dst = dst_b ? p.yb : p.ya;
    const int dst_ld = dst_b ? p.cout - p.split : p.split;
    const int dst_c0 = dst_b ? oc * 8 - p.split : oc * 8;
    const bf16_t *ex = dst_b ? p.exb : p.exa;
    const float *ecf = dst_b ? p.e_coef_b : p.e_coef_a, *emi = dst_b ? p.e_mi_b : p.e_mi_a;
    const float e_slope = dst_b ? p.e_slope_b : p.e_slope_a;
    const bf16_t *eadd = dst_b ? p.e_add : nullptr;
    double *dsums = dst_b ? p.sums_b : p.sums_a;
    float ea[8], eb[8], xp[8], xq[8];
    const bool has_e = !OUT32 && ecf != nullptr;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        ea[q] = 1.f; eb[q] = 1.f; xp[q] = 0.f; xq[q] = 0.f;
        if (has_e) {
            const int ch = dst_c0 + q;
            const float mu = ecf[ch], sc = ecf[dst_ld + ch], be = ecf[2 * dst_ld + ch], is = emi[dst_ld + ch];
            ea[q] = sc; eb[q] = be - mu * sc; xp[q] = is; xq[q] = -mu * is;
        }
    }

    const long long last_row = p.rows - 1;
    const long long nstrips = (p.rows + RG_ROWS - 1) / RG_ROWS;
    const long long sstride = (long long)gridDim.x * 4;
    long long strip = (long long)blockIdx.x * 4 + wave;

    uint4 pg[UMAX], py[UMAX];
    auto fetch = [&](long long st) {
        const long long row0 = st * RG_ROWS;
#pragma unroll
        for (int u = 0; u < UMAX; ++u) {
            if (u < U) {
                const int r = (lane + 64 * u) >> nshift;
                long long row = row0 + r; if (row > last_row) row = last_row;
                pg[u] = ld_u4_stream(p.gz + (size_t)row * p.kdim + ic * 8);
                if (has_g) py[u] = ld_u4_stream(p.y2 + (size_t)row * p.kdim + ic * 8);
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < UMAX; ++u) {
            if (u < U) {
                const int r = (lane + 64 * u) >> nshift;
                float g[8]; bf_unpack8(pg[u], g);
                if (has_g) {
                    float yv[8]; bf_unpack8(py[u], yv);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        float t = g[q];
                        if (g_act) t = bf_bnz(yv[q], za[q], zb[q]) > 0.f ? t : t * p.g_slope;
                        g[q] = __builtin_fmaf(gA[q], t, __builtin_fmaf(gB[q], yv[q], gC[q]));
                    }
                }
                Ss[bf_chunk(r, ic, cpi_s)] = bf_pack8(g);
            }
        }
    };

    double ssum[8], ssq[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { ssum[q] = 0.0; ssq[q] = 0.0; }
    const int n = lane & 31, h = lane >> 5;
    const int wrow0 = w_perm(n);
    const int KS = ncx >> 1;

    if (strip < nstrips) fetch(strip);
    for (; strip < nstrips; strip += sstride) {
        const long long row0 = strip * RG_ROWS;
        commit();
        if (strip + sstride < nstrips) fetch(strip + sstride);
        // the store phase's operands (pre-BN tensor in front, second incoming gradient) land under the MFMA loop
        uint4 ex_pf[OUT32 ? 1 : 2 * NT], ea_pf[OUT32 ? 1 : 2 * NT];
        (void)ex_pf; (void)ea_pf;
        if constexpr (!OUT32) {
#pragma unroll
            for (int u = 0; u < 2 * NT; ++u) {
                ex_pf[u] = make_uint4(0, 0, 0, 0); ea_pf[u] = ex_pf[u];
                if (u < OU) {
                    int r = (lane + 64 * u) >> oshift; if (r >= RG_ROWS) r = RG_ROWS - 1;
                    long long row = row0 + r; if (row > last_row) row = last_row;
                    const size_t off = (size_t)row * dst_ld + dst_c0;
                    if (has_e) ex_pf[u] = *reinterpret_cast<const uint4 *>(ex + off);
                    if (eadd) ea_pf[u] = ld_u4_stream(eadd + off);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();

        i2p_f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        for (int ks = 0; ks < KS; ++ks) {
            const int kc = 2 * ks + h;
            const i2p_bf16x8 xb = __builtin_bit_cast(i2p_bf16x8, Ss[bf_chunk(n, kc, cpi_s)]);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const i2p_bf16x8 wa = __builtin_bit_cast(i2p_bf16x8, Ws[bf_chunk(t * 32 + wrow0, kc, cpi_s)]);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, xb, acc[t], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();

        if constexpr (OUT32) {          // raw-input gradient: 64 contiguous bytes per lane and tile, straight from the fragments
            const long long row = row0 + n;
            if (row <= last_row) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int co = t * 32 + 16 * h;
                    if (co < p.cout) {
                        float *o = p.y32 + (size_t)row * p.cout + co;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            *reinterpret_cast<float4 *>(o + 4 * j) = make_float4(acc[t][4 * j], acc[t][4 * j + 1], acc[t][4 * j + 2], acc[t][4 * j + 3]);
                    }
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int co = t * 32 + 16 * h;
                if (co < p.cout) {
                    float lo8[8], hi8[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) { lo8[q] = acc[t][q]; hi8[q] = acc[t][8 + q]; }
                    Ss[bf_chunk(n, (co >> 3), cpo_s)] = bf_pack8(lo8);
                    Ss[bf_chunk(n, (co >> 3) + 1, cpo_s)] = bf_pack8(hi8);
                }
            }
            __builtin_amdgcn_wave_barrier();
            float s1[8], s2[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { s1[q] = 0.f; s2[q] = 0.f; }
#pragma unroll
            for (int u = 0; u < 2 * NT; ++u) {
                if (u < OU) {
                    const int r = (lane + 64 * u) >> oshift;
                    if (r < RG_ROWS && row0 + r <= last_row) {
                        const size_t off = (size_t)(row0 + r) * dst_ld + dst_c0;
                        float f[8]; bf_unpack8(Ss[bf_chunk(r, oc, cpo_s)], f);
                        if (eadd) {
                            float a[8]; bf_unpack8(ea_pf[u], a);
#pragma unroll
                            for (int q = 0; q < 8; ++q) f[q] += a[q];
                        }
                        if (has_e) {
                            float xv[8]; bf_unpack8(ex_pf[u], xv);
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                f[q] = bf_bnz(xv[q], ea[q], eb[q]) > 0.f ? f[q] : f[q] * e_slope;
                                f[q] = bf_round(f[q]);
                                s1[q] += f[q]; s2[q] = __builtin_fmaf(f[q], __builtin_fmaf(xv[q], xp[q], xq[q]), s2[q]);
                            }
                        }
                        st_u4_stream(dst + off, bf_pack8(f));
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) { ssum[q] += (double)s1[q]; ssq[q] += (double)s2[q]; }
            __builtin_amdgcn_wave_barrier();
        }
    }

    if (!OUT32 && has_e && dsums) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            double a = ssum[q], b = ssq[q];
            for (int off = 32; off >= nco; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
            if (lane < nco) {
                double *rep = dsums + (size_t)((blockIdx.x * 4 + wave) % REP) * 2 * dst_ld;
                atomicAdd(rep + dst_c0 + q, a); atomicAdd(rep + dst_ld + dst_c0 + q, b);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// wgrad: dW[co][ci] = sum_r g^y[r][co] * x'[r][ci].  The contraction runs over ROWS, so both MFMA operands need
// 8 consecutive rows of one channel per lane: a staging task loads 8 rows x 8 channels (eight 16-byte loads of one
// chunk column), applies BN-backward / BN+activation in fp32 and converts with the pairing (row 2j, row 2j+1), so
// that the TRANSPOSED LDS images Gt[co][row], Xt[ci][row] are written with 16-byte stores at no extra cost.
// 256 threads = one wave per SIMD (up to 512 VGPRs each), 128-row tiles: every thread stages one G task and one X
// task per tile (uniform code), the next tile's loads issued in two halves as soon as their registers are free,
// double-buffered images, one barrier per tile, 4 output tiles of 32 x 32 per wave.
// ---------------------------------------------------------------------------------------------------------------
struct WgradP {
    long long rows;
    int cin, cout;               // cout in {16,..,128}; cin <= 128: bf16 source(s) pow2, or one fp32 source with cin % 4 == 0
    int ncx, nco;                // chunks of 8 channels: inputs (cin rounded up to 8), outputs
    const bf16_t *gz, *y;        // [rows, cout]
    const float *g_coef; float g_slope;
    const void *x; int x_ld;
    const bf16_t *xb; int xb_ld; int split;
    const float *coef_a, *coef_b; float slope_a, slope_b;
    float *dw_partial;           // [grid][cout][cin]
};

constexpr int WG_THREADS = 256;
constexpr int WG_R = 128;
constexpr int WG_CPS = 4;         // 128 rows = 16 chunks per image row

// transposed images: the 8 lanes of a 16-byte store hold channels 8 apart (same row group), the 16 lanes of a fragment
// read hold 16 consecutive channels: XOR with the low four channel bits AND bits 4..6 keeps both conflict-free
__device__ __forceinline__ int wg_chunk(int ch, int c) { return (ch << WG_CPS) + (c ^ ((ch & 15) ^ (((ch >> 4) & 7) << 1))); }

template <bool XBF16>
struct WgRegs { uint4 g[8], y[8], x[XBF16 ? 8 : 16]; };

template <bool XBF16>
__global__ __launch_bounds__(WG_THREADS, 1) void wgrad_bf16_kernel(WgradP p) {
    extern __shared__ uint4 smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nco = p.nco, ncx = p.ncx;
    const int g_rows = nco * 8, x_rows = ((ncx * 8 + 31) / 32) * 32;          // image rows (channels), X padded to 32
    const int buf_chunks = (g_rows + x_rows) << WG_CPS;
    // tasks: (chunk column, 8-row group) with the chunk column fastest over the lanes (a row's chunks are adjacent lanes)
    const bool do_g = tid < 16 * nco, do_x = tid < 16 * ncx;
    const int gtg = do_g ? tid / nco : 0, gtc = do_g ? tid - gtg * nco : 0;
    const int xtg = do_x ? tid / ncx : 0, xtc = do_x ? tid - xtg * ncx : 0;
    for (int b = 0; b < 2; ++b) {                                // zero the X image's padding channels once
        uint4 *Xt = smem + (size_t)b * buf_chunks + (g_rows << WG_CPS);
        for (int i = tid; i < ((x_rows - ncx * 8) << WG_CPS); i += WG_THREADS) Xt[((ncx * 8) << WG_CPS) + i] = make_uint4(0, 0, 0, 0);
    }

    // constants of this thread's 8 output channels (G task) and 8 input channels (X task)
    float gA[8], gB[8], gC[8], za[8], zb[8], xa_[8], xb_[8];
    const bool has_g = p.g_coef != nullptr, g_act = has_g && p.g_slope != 1.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        gA[q] = 1.f; gB[q] = 0.f; gC[q] = 0.f; za[q] = 1.f; zb[q] = 0.f;
        if (has_g) {
            const int ch = gtc * 8 + q, K = p.cout;
            const float m1 = p.g_coef[ch], m2 = p.g_coef[K + ch], sc = p.g_coef[2 * K + ch], mu = p.g_coef[3 * K + ch],
                        is = p.g_coef[4 * K + ch], be = p.g_coef[5 * K + ch];
            gA[q] = sc; gB[q] = -(sc * m2) * is; gC[q] = -(sc * m1) - gB[q] * mu; za[q] = sc; zb[q] = be - mu * sc;
        }
    }
    const int xch = xtc * 8;
    const bool second = XBF16 && p.xb && xch >= p.split;
    const float *xcf = second ? p.coef_b : p.coef_a;
    const int xldc = (XBF16 && p.xb) ? (second ? p.cin - p.split : p.split) : p.cin;
    const int xs_c0 = second ? xch - p.split : xch;
    const int xs_ld = second ? p.xb_ld : p.x_ld;
    const bf16_t *xs_bf = second ? p.xb : reinterpret_cast<const bf16_t *>(p.x);
    const float *xs_f = reinterpret_cast<const float *>(p.x);
    const float x_slope = second ? p.slope_b : p.slope_a;
    const bool x_coef = xcf != nullptr;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float a = 1.f, b = 0.f;
        if (xcf && xs_c0 + q < xldc) { a = xcf[xldc + xs_c0 + q]; b = xcf[2 * xldc + xs_c0 + q] - xcf[xs_c0 + q] * a; }
        xa_[q] = a; xb_[q] = b;
    }

    const long long last_row = p.rows - 1;
    const long long ntiles = (p.rows + WG_R - 1) / WG_R;
    auto fetch_g = [&](long long tile, WgRegs<XBF16> &R) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (do_g) {
                long long row = tile * WG_R + gtg * 8 + j; if (row > last_row) row = last_row;
                R.g[j] = ld_u4_stream(p.gz + (size_t)row * p.cout + gtc * 8);
                if (has_g) R.y[j] = ld_u4_stream(p.y + (size_t)row * p.cout + gtc * 8);
            }
        }
    };
    auto fetch_x = [&](long long tile, WgRegs<XBF16> &R) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (do_x) {
                long long row = tile * WG_R + xtg * 8 + j; if (row > last_row) row = last_row;
                if constexpr (XBF16) R.x[j] = ld_u4_stream(xs_bf + (size_t)row * xs_ld + xs_c0);
                else {
                    const float *xr = xs_f + (size_t)row * xs_ld;
                    const float4 a = xs_c0 < p.cin ? *reinterpret_cast<const float4 *>(xr + xs_c0) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 b = xs_c0 + 4 < p.cin ? *reinterpret_cast<const float4 *>(xr + xs_c0 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    R.x[2 * j] = make_uint4(__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), __float_as_uint(a.w));
                    R.x[2 * j + 1] = make_uint4(__float_as_uint(b.x), __float_as_uint(b.y), __float_as_uint(b.z), __float_as_uint(b.w));
                }
            }
        }
    };
    auto commit_g = [&](long long tile, int buf, const WgRegs<XBF16> &R) {
        uint4 *Gt = smem + (size_t)buf * buf_chunks;
        if (do_g) {
            float v[8][8];                                       // [row j][channel q]
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool live = tile * WG_R + gtg * 8 + j <= last_row;
                float g[8]; bf_unpack8(R.g[j], g);
                if (has_g) {
                    float yv[8]; bf_unpack8(R.y[j], yv);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        float t = g[q];
                        if (g_act) t = bf_bnz(yv[q], za[q], zb[q]) > 0.f ? t : t * p.g_slope;
                        g[q] = __builtin_fmaf(gA[q], t, __builtin_fmaf(gB[q], yv[q], gC[q]));
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) v[j][q] = live ? g[q] : 0.f;     // rows past the end contribute nothing
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)                          // image row = channel gtc*8+q, chunk = row group
                Gt[wg_chunk(gtc * 8 + q, gtg)] = make_uint4(bf_pack2(v[0][q], v[1][q]), bf_pack2(v[2][q], v[3][q]),
                                                            bf_pack2(v[4][q], v[5][q]), bf_pack2(v[6][q], v[7][q]));
        }
    };
    auto commit_x = [&](long long tile, int buf, const WgRegs<XBF16> &R) {
        uint4 *Xt = smem + (size_t)buf * buf_chunks + (g_rows << WG_CPS);
        if (do_x) {
            float v[8][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool live = tile * WG_R + xtg * 8 + j <= last_row;
                float f[8];
                if constexpr (XBF16) bf_unpack8(R.x[j], f);
                else {
                    f[0] = __uint_as_float(R.x[2 * j].x); f[1] = __uint_as_float(R.x[2 * j].y); f[2] = __uint_as_float(R.x[2 * j].z); f[3] = __uint_as_float(R.x[2 * j].w);
                    f[4] = __uint_as_float(R.x[2 * j + 1].x); f[5] = __uint_as_float(R.x[2 * j + 1].y); f[6] = __uint_as_float(R.x[2 * j + 1].z); f[7] = __uint_as_float(R.x[2 * j + 1].w);
                }
                if (x_coef) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) f[q] = bf_act(bf_bnz(f[q], xa_[q], xb_[q]), x_slope);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) v[j][q] = live ? f[q] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                Xt[wg_chunk(xtc * 8 + q, xtg)] = make_uint4(bf_pack2(v[0][q], v[1][q]), bf_pack2(v[2][q], v[3][q]),
                                                            bf_pack2(v[4][q], v[5][q]), bf_pack2(v[6][q], v[7][q]));
        }
    };

    // output tiles: MT x NTI blocks of 32 x 32, round-robin over the 4 waves (<= 4 per wave)
    const int MT = (g_rows + 31) / 32, NTI = x_rows / 32, WT = MT * NTI;
    i2p_f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const int n = lane & 31, h = lane >> 5;
    auto mfma_tile = [&](int buf) {
        const uint4 *Gt = smem + (size_t)buf * buf_chunks, *Xt = Gt + (g_rows << WG_CPS);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int tw = wave + 4 * t;
            if (tw < WT) {
                const int mt = tw / NTI, nt = tw - mt * NTI;
                int grow = mt * 32 + n; if (grow >= g_rows) grow = g_rows - 1;    // cout = 16: rows 16..31 of the tile duplicate (never stored)
#pragma unroll
                for (int ks = 0; ks < WG_R / 16; ++ks) {
                    const i2p_bf16x8 a = __builtin_bit_cast(i2p_bf16x8, Gt[wg_chunk(grow, 2 * ks + h)]);
                    const i2p_bf16x8 b = __builtin_bit_cast(i2p_bf16x8, Xt[wg_chunk(nt * 32 + n, 2 * ks + h)]);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
                }
            }
        }
    };

    // tiles of this block: tile0, tile0 + G, ...  One register set, refilled in two halves: the next tile's gz / y loads
    // are issued as soon as this tile's G image is staged (they fly under the X staging and the MFMA phase), its x loads
    // right after the X image is staged (they fly under the MFMA phase and the next G staging): the memory pipe never
    // drains, at half the registers of a two-tile prefetch (which spilled at 512 VGPRs).
    const long long G = gridDim.x;
    WgRegs<XBF16> R;
    long long tile = blockIdx.x;
    if (tile < ntiles) { fetch_g(tile, R); fetch_x(tile, R); }
    __syncthreads();                                             // padding rows of both X images are written
    int cur = 0;
    for (; tile < ntiles; tile += G, cur ^= 1) {
        const bool more = tile + G < ntiles;
        commit_g(tile, cur, R);
        if (more) fetch_g(tile + G, R);
        commit_x(tile, cur, R);
        if (more) fetch_x(tile + G, R);
        __syncthreads();
        mfma_tile(cur);
    }

    float *part = p.dw_partial + (size_t)blockIdx.x * p.cout * p.cin;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int tw = wave + 4 * t;
        if (tw < WT) {
            const int mt = tw / NTI, nt = tw - mt * NTI;
            const int ci = nt * 32 + n;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (co < p.cout && ci < p.cin) part[(size_t)co * p.cin + ci] = acc[t][e];
            }
        }
    }
}

// Wt[ci][co] = w[co][ci] as a bf16 image with rows = input channels (dgrad B operand of the pair kernel)
__device__ __forceinline__ void stage_weights_t(uint4 *Wt, const float *w, int cin, int cout, int nco, int cps, int tid) {
    for (int i = tid; i < cin * nco; i += 256) {
        const int ci = i / nco, c = i - ci * nco;
        float f[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] = w[(size_t)(c * 8 + q) * cin + ci];
        Wt[bf_chunk(ci, c, cps)] = bf_pack8(f);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of the factored first cost-volume layer (pair mode) on bf16 gz / y, the counterpart of mlp.hip's
// pair_bwd_kernel: a block owns one 64-pixel tile of one sample and a chunk of its points and walks the points with
// the pixel tile fixed (per-pixel sums d_g / d_bias_k stay in registers across the walk, per-point sums d_f / d_bias_n
// are column sums of one tile).  Per point: G = BN-backward(gz, y) (64 px x cout) is staged as a row-major image
// (dgrad A operand) AND a transposed image (wgrad A operand), X' = f[n] * g[k] as a transposed image; dW += G^T X',
// T = G . W on v_mfma_f32_32x32x16_bf16.  512 threads: 256 stage G (4 px x 8 ch each), 128 stage X' (8 px x 8 ch of
// the block-constant pixel factors in registers), all 8 waves share the MFMA tiles; images double-buffered, one
// barrier per point, the next point's gz / y in flight during the MFMA phase.  (256 threads: every thread stages
// 4 px x 8 ch of G; the pixel factors are a block-constant image, see the kernel.)
// ---------------------------------------------------------------------------------------------------------------
struct PairBwdP {
    int B, N, M, NC, NL;
    int cin, cout;
    const bf16_t *gz, *y;
    const float *g_coef;         // [6][cout] or nullptr
    const float *f, *g, *w;
    float *d_f, *d_g, *d_bn, *d_bk, *dw_partial;
};
constexpr int PB_THREADS = 256;          // 4 waves, one per SIMD: up to 512 VGPRs each (the tile accumulators + the staging block)
constexpr int PB_PX = 64;

__global__ __launch_bounds__(PB_THREADS, 1) void pair_bwd_bf16_kernel(PairBwdP p) {
    extern __shared__ uint4 smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nco = p.cout >> 3, ncx = p.cin >> 3;
    const int cps_o = ilog2(nco);                                // chunk pitch of the images whose rows hold `cout` channels
    // LDS carve-up (16-byte units)
    uint4 *Wt = smem;                                            // [cin][nco]: Wt[ci][co] = w[co][ci]            (dgrad B)
    uint4 *Xg = Wt + (p.cin << cps_o);                           // [cin][8]:  Xg[ci][px] = g[b, k0+px, ci]       (wgrad B, constant)
    const int gt_sz = p.cout << 3, gr_sz = PB_PX << cps_o;
    const int buf_sz = gt_sz + gr_sz;
    uint4 *bufs = Xg + (p.cin << 3);
    float *Rs = reinterpret_cast<float *>(bufs + 2 * buf_sz);    // [2][16][cout] column sums of G per 4-pixel row group

    const int KT = (p.M + PB_PX - 1) / PB_PX;
    int bid = blockIdx.x;
    const int nc = bid % p.NC; bid /= p.NC;
    const int kt = bid % KT; const int b = bid / KT;
    const int k0 = kt * PB_PX;
    const int n_begin = nc * p.NL, n_end = min(p.N, n_begin + p.NL);

    stage_weights_t(Wt, p.w, p.cin, p.cout, nco, cps_o, tid);
    // The wgrad operand X'[px][ci] = f[n][ci] * g[px][ci] factors: dW[co][ci] += f[n][ci] * sum_px G[px][co] * g[px][ci],
    // so the pixel factors are staged ONCE as a transposed bf16 image and f[n] scales the per-point product in fp32.
    for (int t = tid; t < 8 * ncx; t += PB_THREADS) {
        const int tc = t % ncx, tg = t / ncx;
        float gk[8][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + tg * 8 + j;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
            if (k < p.M) {
                const float *gp = p.g + ((size_t)b * p.M + k) * p.cin + tc * 8;
                a = *reinterpret_cast<const float4 *>(gp); c = *reinterpret_cast<const float4 *>(gp + 4);
            }
            gk[j][0] = a.x; gk[j][1] = a.y; gk[j][2] = a.z; gk[j][3] = a.w; gk[j][4] = c.x; gk[j][5] = c.y; gk[j][6] = c.z; gk[j][7] = c.w;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            Xg[bf_chunk(tc * 8 + q, tg, 3)] = make_uint4(bf_pack2(gk[0][q], gk[1][q]), bf_pack2(gk[2][q], gk[3][q]),
                                                         bf_pack2(gk[4][q], gk[5][q]), bf_pack2(gk[6][q], gk[7][q]));
    }

    // ---- staging role: 16*nco threads stage G, 4 pixels x 8 channels each ----------------------------------------
    const bool is_g = tid < 16 * nco;
    const int gtc = is_g ? tid % nco : 0, gtg = is_g ? tid / nco : 0;           // chunk column, 4-pixel row group
    float gA[8], gB[8], gC[8];
    const bool has_g = p.g_coef != nullptr;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        gA[q] = 1.f; gB[q] = 0.f; gC[q] = 0.f;
        if (is_g && has_g) {
            const int ch = gtc * 8 + q, K = p.cout;
            const float m1 = p.g_coef[ch], m2 = p.g_coef[K + ch], sc = p.g_coef[2 * K + ch], mu = p.g_coef[3 * K + ch], is = p.g_coef[4 * K + ch];
            gA[q] = sc; gB[q] = -(sc * m2) * is; gC[q] = -(sc * m1) - gB[q] * mu;
        }
    }
    // MFMA roles
    const int NTI = p.cin >> 5, MT = p.cout >> 5, WT = MT * NTI, DT = 2 * NTI;
    const int n = lane & 31, h = lane >> 5;
    constexpr int WPW = 4, DPW = 2;                              // wgrad / dgrad tiles per wave (16 / 8 tiles at 128 x 128)
    int d_pt[DPW], d_ci[DPW]; bool has_d[DPW];
    float gk_frag[DPW][16];                                      // g[b, k0 + px(r), ci] for this lane's dgrad fragments
#pragma unroll
    for (int t = 0; t < DPW; ++t) {
        const int td = wave + 4 * t;
        has_d[t] = td < DT;
        d_pt[t] = has_d[t] ? td / NTI : 0; d_ci[t] = (has_d[t] ? td % NTI : 0) * 32 + n;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int k = k0 + d_pt[t] * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            gk_frag[t][e] = (has_d[t] && k < p.M) ? p.g[((size_t)b * p.M + k) * p.cin + d_ci[t]] : 0.f;
        }
    }
    int w_ci[WPW];                                               // input channel of this lane in its wgrad tiles
#pragma unroll
    for (int t = 0; t < WPW; ++t) w_ci[t] = ((wave + 4 * t) % NTI) * 32 + n;
    float accw[WPW][16], dg_acc[DPW][16], dbk_acc[4][8];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
#pragma unroll
        for (int t = 0; t < WPW; ++t) accw[t][e] = 0.f;
#pragma unroll
        for (int t = 0; t < DPW; ++t) dg_acc[t][e] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 8; ++q) dbk_acc[j][q] = 0.f;

    uint4 rg[4], ry[4];
    float fd_next[DPW], fw_next[WPW];
#pragma unroll
    for (int t = 0; t < DPW; ++t) fd_next[t] = 0.f;
#pragma unroll
    for (int t = 0; t < WPW; ++t) fw_next[t] = 0.f;
    auto fetch = [&](int nn) {
        const size_t bn = (size_t)b * p.N + nn;
        if (is_g) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int k = k0 + gtg * 4 + j; if (k >= p.M) k = p.M - 1;
                const size_t off = (bn * p.M + k) * p.cout + gtc * 8;
                rg[j] = ld_u4_stream(p.gz + off);
                if (has_g) ry[j] = ld_u4_stream(p.y + off);
            }
        }
#pragma unroll
        for (int t = 0; t < DPW; ++t) fd_next[t] = p.f[bn * p.cin + d_ci[t]];
#pragma unroll
        for (int t = 0; t < WPW; ++t) fw_next[t] = p.f[bn * p.cin + w_ci[t]];
    };
    auto commit = [&](int buf) {
        uint4 *Gt = bufs + (size_t)buf * buf_sz, *Gr = Gt + gt_sz;
        if (is_g) {
            float v[4][8], csum[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) csum[q] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float gq[8]; bf_unpack8(rg[j], gq);
                if (has_g) {
                    float yv[8]; bf_unpack8(ry[j], yv);
#pragma unroll
                    for (int q = 0; q < 8; ++q) gq[q] = __builtin_fmaf(gA[q], gq[q], __builtin_fmaf(gB[q], yv[q], gC[q]));
                }
                const bool live = k0 + gtg * 4 + j < p.M;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    // (the MFMAs see the bf16-rounded G; the bias sums take the same rounded values so that all four
                    //  gradients of this layer describe one and the same G)
                    const float t = live ? bf_round(gq[q]) : 0.f;
                    v[j][q] = t; dbk_acc[j][q] += t; csum[q] += t;
                }
                Gr[bf_chunk(gtg * 4 + j, gtc, cps_o)] = bf_pack8(v[j]);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {                        // transposed image: row = channel, 8-byte half chunk of 4 pixels
                uint2 *dst = reinterpret_cast<uint2 *>(Gt + bf_chunk(gtc * 8 + q, gtg >> 1, 3)) + (gtg & 1);
                *dst = make_uint2(bf_pack2(v[0][q], v[1][q]), bf_pack2(v[2][q], v[3][q]));
            }
            float *rs = Rs + ((size_t)buf * 16 + gtg) * p.cout + gtc * 8;
            *reinterpret_cast<float4 *>(rs) = make_float4(csum[0], csum[1], csum[2], csum[3]);
            *reinterpret_cast<float4 *>(rs + 4) = make_float4(csum[4], csum[5], csum[6], csum[7]);
        }
    };

    int cur = 0;
    float fd_cur[DPW], fw[WPW];
#pragma unroll
    for (int t = 0; t < DPW; ++t) fd_cur[t] = 0.f;
#pragma unroll
    for (int t = 0; t < WPW; ++t) fw[t] = 0.f;
    if (n_begin < n_end) {
        fetch(n_begin);
#pragma unroll
        for (int t = 0; t < DPW; ++t) fd_cur[t] = fd_next[t];
#pragma unroll
        for (int t = 0; t < WPW; ++t) fw[t] = fw_next[t];
        commit(0);
        if (n_begin + 1 < n_end) fetch(n_begin + 1);
    }
    __syncthreads();
    for (int nn = n_begin; nn < n_end; ++nn, cur ^= 1) {
        const size_t bn = (size_t)b * p.N + nn;
        if (tid < p.cout) {                                      // per-point bias gradient: column sums of G
            float s0 = 0.f;
            const float *rs = Rs + (size_t)cur * 16 * p.cout + tid;
#pragma unroll
            for (int r = 0; r < 16; ++r) s0 += rs[(size_t)r * p.cout];
            atomicAdd(p.d_bn + bn * p.cout + tid, s0);
        }
        float fd_s[DPW], fw_s[WPW];                              // values of point nn+1 (fetched one phase ago)
#pragma unroll
        for (int t = 0; t < DPW; ++t) fd_s[t] = fd_next[t];
#pragma unroll
        for (int t = 0; t < WPW; ++t) fw_s[t] = fw_next[t];
        if (nn + 1 < n_end) commit(cur ^ 1);
        if (nn + 2 < n_end) fetch(nn + 2);
        const uint4 *Gt = bufs + (size_t)cur * buf_sz, *Gr = Gt + gt_sz;
        // ---- wgrad: dW tile (mt, nt) += f[n][ci] * (Gt[mt] . Xg[nt]^T) over the 64 pixels ---------------------
#pragma unroll
        for (int t = 0; t < WPW; ++t) {
            const int tw = wave + 4 * t;
            if (tw < WT) {
                const int mt = tw / NTI, nt = tw - mt * NTI;
                i2p_f32x16 pr;
#pragma unroll
                for (int e = 0; e < 16; ++e) pr[e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < PB_PX / 16; ++ks) {
                    const i2p_bf16x8 a = __builtin_bit_cast(i2p_bf16x8, Gt[bf_chunk(mt * 32 + n, 2 * ks + h, 3)]);
                    const i2p_bf16x8 bb = __builtin_bit_cast(i2p_bf16x8, Xg[bf_chunk(nt * 32 + n, 2 * ks + h, 3)]);
                    pr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, pr, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) accw[t][e] = __builtin_fmaf(pr[e], fw[t], accw[t][e]);
            }
        }
        // ---- dgrad T = G . W, folded into d_g (registers) and d_f (column sums -> atomics) -----------------------
#pragma unroll
        for (int t = 0; t < DPW; ++t) {
            if (has_d[t]) {
                i2p_f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
                for (int ks = 0; ks < (nco >> 1); ++ks) {
                    const i2p_bf16x8 a = __builtin_bit_cast(i2p_bf16x8, Gr[bf_chunk(d_pt[t] * 32 + n, 2 * ks + h, cps_o)]);
                    const i2p_bf16x8 bb = __builtin_bit_cast(i2p_bf16x8, Wt[bf_chunk(d_ci[t], 2 * ks + h, cps_o)]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, acc, 0, 0, 0);
                }
                float colsum = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    dg_acc[t][e] = __builtin_fmaf(acc[e], fd_cur[t], dg_acc[t][e]);
                    colsum = __builtin_fmaf(acc[e], gk_frag[t][e], colsum);
                }
                colsum += __shfl_xor(colsum, 32);
                if (lane < 32) atomicAdd(p.d_f + bn * p.cin + d_ci[t], colsum);
            }
        }
#pragma unroll
        for (int t = 0; t < DPW; ++t) fd_cur[t] = fd_s[t];
#pragma unroll
        for (int t = 0; t < WPW; ++t) fw[t] = fw_s[t];
        __syncthreads();
    }

    // ---- flush the per-pixel accumulators and the weight-gradient partial ---------------------------------------
#pragma unroll
    for (int t = 0; t < DPW; ++t) {
        if (has_d[t]) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = k0 + d_pt[t] * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (k < p.M) atomicAdd(p.d_g + ((size_t)b * p.M + k) * p.cin + d_ci[t], dg_acc[t][e]);
            }
        }
    }
    if (is_g) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + gtg * 4 + j;
            if (k < p.M) {
                float *dk = p.d_bk + ((size_t)b * p.M + k) * p.cout + gtc * 8;
#pragma unroll
                for (int q = 0; q < 8; ++q) atomicAdd(dk + q, dbk_acc[j][q]);
            }
        }
    }
    float *part = p.dw_partial + (size_t)blockIdx.x * p.cout * p.cin;
#pragma unroll
    for (int t = 0; t < WPW; ++t) {
        const int tw = wave + 4 * t;
        if (tw < WT) {
            const int mt = tw / NTI, nt = tw - mt * NTI;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                part[(size_t)co * p.cin + nt * 32 + n] = accw[t][e];
            }
        }
    }
}

// dW = sum of the per-block partials
__global__ __launch_bounds__(256) void reduce_partials_bf16(int nparts, int n, const float *__restrict__ parts, float *__restrict__ out) {
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

// per-channel BN-backward constants [8][c]: m1, m2, scale, mean, invstd, beta, dbeta (= sum gz), dgamma (= sum gz*xhat)
__global__ void bnbwd_coef_bf16(long long rows, int c, const double *__restrict__ dsums, const float *__restrict__ coef,
                                const float *__restrict__ mi, float *__restrict__ out) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    double sd = 0.0, sx = 0.0;
    for (int r = 0; r < REP; ++r) { sd += dsums[(size_t)r * 2 * c + ch]; sx += dsums[(size_t)r * 2 * c + c + ch]; }
    out[ch] = (float)(sd / (double)rows); out[c + ch] = (float)(sx / (double)rows);
    out[2 * c + ch] = coef[c + ch]; out[3 * c + ch] = mi[ch]; out[4 * c + ch] = mi[c + ch]; out[5 * c + ch] = coef[2 * c + ch];
    out[6 * c + ch] = (float)sd; out[7 * c + ch] = (float)sx;
}

inline bool pow2_16_128(int c) { return c == 16 || c == 32 || c == 64 || c == 128; }
inline int log2i(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }

template <int NT, bool XBF16, bool PAIR>
int launch_fwd(const FwdP &p, hipStream_t st) {
    const int strip = RG_ROWS << (p.cpi_s > p.cpo_s ? p.cpi_s : p.cpo_s);
    const size_t bytes = ((size_t)((32 * NT) << p.cpi_s) + 4 * (size_t)strip) * sizeof(uint4);
    if (bytes > 160 * 1024) return I2P_ERR_BAD_ARG;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(rg_fwd_kernel<NT, XBF16, PAIR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const long long nstrips = (p.rows + RG_ROWS - 1) / RG_ROWS;
    long long g = (nstrips + 3) / 4;
    const long long cap = (XBF16 && bytes <= 80 * 1024) ? 512 : 256;
    const unsigned grid = (unsigned)(g < cap ? (g < 1 ? 1 : g) : cap);
    hipLaunchKernelGGL((rg_fwd_kernel<NT, XBF16, PAIR>), dim3(grid), dim3(RG_THREADS), bytes, st, p);
    I2P_RETURN_LAUNCH_STATUS();
}

template <bool XBF16, bool PAIR>
int dispatch_fwd(const FwdP &p, hipStream_t st) {
    switch ((p.cout + 31) / 32) {
        case 1: return launch_fwd<1, XBF16, PAIR>(p, st);
        case 2: return launch_fwd<2, XBF16, PAIR>(p, st);
        case 4: return launch_fwd<4, XBF16, PAIR>(p, st);
        default: return I2P_ERR_BAD_ARG;
    }
}

int fwd_impl(long long rows, int cin, int cout, const void *x, int x_bf16, int x_ld, const bf16_t *xb, int xb_ld, int split,
             const float *coef_a, float slope_a, const float *coef_b, float slope_b, const float *w, bf16_t *y, double *sums,
             const float *pair_f, const float *bias_n, const float *bias_k, int pN, int pM, void *stream) {
    if (rows < 0 || cin <= 0 || !pow2_16_128(cout)) return I2P_ERR_BAD_ARG;
    if (rows == 0) return 0;
    if (!x || !w || !y) return I2P_ERR_BAD_ARG;
    if (x_bf16 ? (cin > 128 || (cin & 7)) : (cin > 160 || (cin & 3))) return I2P_ERR_BAD_ARG;
    if (xb && (!x_bf16 || (split & 7) || split <= 0 || split >= cin)) return I2P_ERR_BAD_ARG;
    FwdP p{};
    p.rows = rows; p.cin = cin; p.cout = cout;
    p.ncx = ((cin + 15) / 16) * 2;
    if ((coef_a || coef_b) && (p.ncx & (p.ncx - 1))) return I2P_ERR_BAD_ARG;      // BN on load needs one channel block per lane
    if ((x_bf16 || pair_f) && (p.ncx & (p.ncx - 1))) return I2P_ERR_BAD_ARG;
    p.cpi_s = log2i(p.ncx); p.cpo_s = log2i(cout / 8);
    p.x = x; p.x_ld = x_ld; p.xb = xb; p.xb_ld = xb_ld; p.split = xb ? split : cin;
    p.coef_a = coef_a; p.coef_b = coef_b; p.slope_a = slope_a; p.slope_b = slope_b;
    p.w = w; p.y = y; p.sums = sums;
    p.pair_f = pair_f; p.bias_n = bias_n; p.bias_k = bias_k; p.pN = pN; p.pM = pM;
    hipStream_t st = (hipStream_t)stream;
    if (pair_f) return dispatch_fwd<false, true>(p, st);
    return x_bf16 ? dispatch_fwd<true, false>(p, st) : dispatch_fwd<false, false>(p, st);
}

template <int NT, bool OUT32>
int launch_dgrad(const DgradP &p, hipStream_t st) {
    const int strip = RG_ROWS << (p.cpi_s > p.cpo_s ? p.cpi_s : p.cpo_s);
    const size_t bytes = ((size_t)((32 * NT) << p.cpi_s) + 4 * (size_t)strip) * sizeof(uint4);
    if (bytes > 160 * 1024) return I2P_ERR_BAD_ARG;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(rg_dgrad_kernel<NT, OUT32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const long long nstrips = (p.rows + RG_ROWS - 1) / RG_ROWS;
    long long g = (nstrips + 3) / 4;
    const long long cap = 256;           // (register budget: one block per CU)
    const unsigned grid = (unsigned)(g < cap ? (g < 1 ? 1 : g) : cap);
    hipLaunchKernelGGL((rg_dgrad_kernel<NT, OUT32>), dim3(grid), dim3(RG_THREADS), bytes, st, p);
    I2P_RETURN_LAUNCH_STATUS();
}

template <bool OUT32>
int dispatch_dgrad(const DgradP &p, hipStream_t st) {
    switch ((p.cout + 31) / 32) {
        case 1: return launch_dgrad<1, OUT32>(p, st);
        case 2: return launch_dgrad<2, OUT32>(p, st);
        case 4: return launch_dgrad<4, OUT32>(p, st);
        default: return I2P_ERR_BAD_ARG;
    }
}

template <bool XBF16>
int wgrad_launch_t(WgradP &q, float *dw, hipStream_t st, unsigned grid, size_t bytes) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(wgrad_bf16_kernel<XBF16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(wgrad_bf16_kernel<XBF16>, dim3(grid), dim3(WG_THREADS), bytes, st, q);
    const int n = q.cout * q.cin;
    if (!i2p_defer_reduce(1, (int)grid, n, q.dw_partial, dw))
        hipLaunchKernelGGL(reduce_partials_bf16, dim3((n + 31) / 32), dim3(256), 0, st, (int)grid, n, q.dw_partial, dw);
    I2P_RETURN_LAUNCH_STATUS();
}

int wgrad_launch(WgradP &q, float *dw, hipStream_t st, unsigned grid, bool x_bf16) {
    const int g_rows = q.nco * 8, x_rows = ((q.ncx * 8 + 31) / 32) * 32;
    if (q.nco > 16 || q.ncx > 16) return I2P_ERR_BAD_ARG;
    const size_t bytes = 2 * ((size_t)(g_rows + x_rows) << WG_CPS) * sizeof(uint4);
    return x_bf16 ? wgrad_launch_t<true>(q, dw, st, grid, bytes) : wgrad_launch_t<false>(q, dw, st, grid, bytes);
}

struct TwoSrc { int split; const bf16_t *xb; const float *coef_b, *mi_b; float slope_b; bf16_t *gz_b; double *dsums_b; const bf16_t *e_add; };

int bwd_impl(long long rows, int cin, int cout, const bf16_t *gz, const bf16_t *y, const float *out_coef, const float *out_mi,
             const double *out_dsums, const void *x, int x_bf16, const float *in_coef, const float *in_mi, float slope_in,
             const float *w, void *gz_in, int gz_in_bf16, double *in_dsums, float *dw_partial, float *dw, float slope_out,
             const TwoSrc *two, void *stream) {
    if (rows <= 0 || cin <= 0 || !pow2_16_128(cout)) return I2P_ERR_BAD_ARG;
    if (!gz || !x || !w || !dw_partial || !dw) return I2P_ERR_BAD_ARG;
    if (out_coef && (!y || !out_mi || !out_dsums)) return I2P_ERR_BAD_ARG;
    if (in_coef && !in_mi) return I2P_ERR_BAD_ARG;
    if (x_bf16 ? !pow2_16_128(cin) : (cin > 128 || (cin & 3))) return I2P_ERR_BAD_ARG;
    if (two && (!x_bf16 || (two->split & 7))) return I2P_ERR_BAD_ARG;
    if (gz_in && !pow2_16_128(cin)) return I2P_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)i2p_lin_bwd_bf16_grid(rows);
    float *g_coef = out_coef ? dw_partial + (size_t)grid * cout * cin : nullptr;    // [8][cout] scratch tail; rows 6, 7 = dbeta, dgamma for the caller
    if (gz_in && gz_in_bf16 && x_bf16 && !two && out_coef && in_coef && slope_out == 1.f && grid == 256 && i2p_bwd_fused_bf16_ok(rows, cin, cout)) {
        // 64-output-channel layer on many rows: dgrad + wgrad from one read of gz / y / x (csrc/mlp_bwd_fused_bf16.hip)
        const int rc = i2p_bwd_fused_bf16(rows, cin, cout, gz, y, out_dsums, out_coef, out_mi, g_coef, reinterpret_cast<const bf16_t *>(x), in_coef, in_mi, slope_in, w,
                                          reinterpret_cast<bf16_t *>(gz_in), in_dsums, dw_partial, grid, stream);
        if (rc) return rc;
        const int n = cout * cin;
        if (!i2p_defer_reduce(1, (int)grid, n, dw_partial, dw))
            hipLaunchKernelGGL(reduce_partials_bf16, dim3((n + 31) / 32), dim3(256), 0, st, (int)grid, n, dw_partial, dw);
        I2P_RETURN_LAUNCH_STATUS();
    }
    if (gz_in && gz_in_bf16 && x_bf16 && two && out_coef && in_coef && two->coef_b && two->mi_b && two->e_add && two->gz_b && in_dsums && two->dsums_b &&
        slope_out == 1.f && grid == 256 && i2p_bwd_fused2_bf16_ok(rows, two->split, cin - two->split, cout)) {
        // the two-source 64 + 64 -> 128 layer on many rows: one pass over gz / y / xa / xb / e_add (csrc/mlp_bwd_fused_bf16.hip)
        const int rc = i2p_bwd_fused2_bf16(rows, gz, y, out_dsums, out_coef, out_mi, g_coef, reinterpret_cast<const bf16_t *>(x), in_coef, in_mi, slope_in, two->xb, two->coef_b,
                                           two->mi_b, two->slope_b, two->e_add, w, reinterpret_cast<bf16_t *>(gz_in), in_dsums, two->gz_b,
                                           two->dsums_b, dw_partial, grid, stream);
        if (rc) return rc;
        const int n = cout * cin;
        if (!i2p_defer_reduce(1, (int)grid, n, dw_partial, dw))
            hipLaunchKernelGGL(reduce_partials_bf16, dim3((n + 31) / 32), dim3(256), 0, st, (int)grid, n, dw_partial, dw);
        I2P_RETURN_LAUNCH_STATUS();
    }
    // (the one-pass kernels above form the BN-backward constants in their prologues; the two-kernel forms take them from this launch)
    if (out_coef) hipLaunchKernelGGL(bnbwd_coef_bf16, dim3((cout + 63) / 64), dim3(64), 0, st, rows, cout, out_dsums, out_coef, out_mi, g_coef);
    if (gz_in) {
        DgradP q{};
        q.rows = rows; q.kdim = cout; q.cout = cin; q.ncx = cout / 8; q.cpi_s = log2i(q.ncx); q.cpo_s = log2i(cin / 8);
        q.gz = gz; q.y2 = y; q.g_coef = g_coef; q.g_slope = out_coef ? slope_out : 1.f; q.w = w;
        if (gz_in_bf16) {
            q.ya = reinterpret_cast<bf16_t *>(gz_in); q.split = cin; q.exa = in_coef ? reinterpret_cast<const bf16_t *>(x) : nullptr;
            q.e_coef_a = in_coef; q.e_mi_a = in_mi; q.e_slope_a = slope_in; q.sums_a = in_coef ? in_dsums : nullptr;
            if (in_coef && !x_bf16) return I2P_ERR_BAD_ARG;
            if (two) {
                q.split = two->split; q.yb = two->gz_b; q.exb = two->xb; q.e_coef_b = two->coef_b; q.e_mi_b = two->mi_b;
                q.e_slope_b = two->slope_b; q.e_add = two->e_add; q.sums_b = two->dsums_b;
            }
            const int rc = dispatch_dgrad<false>(q, st);
            if (rc) return rc;
        } else {
            if (in_coef || two) return I2P_ERR_BAD_ARG;
            q.y32 = reinterpret_cast<float *>(gz_in); q.split = cin;
            const int rc = dispatch_dgrad<true>(q, st);
            if (rc) return rc;
        }
    }
    const bool two_w3 = !two || (cin == 128 && two->split == 64 && in_coef && two->coef_b);
    if (two_w3 && x_bf16 && grid == 256 && i2p_wreg_wgrad_bf16_ok(rows, cin, cout)) {
        // wide layer on many rows: accumulators stationary in registers, no LDS staging (csrc/mlp_wreg_bf16.hip)
        const int rc = i2p_wreg_wgrad_bf16(rows, cin, cout, gz, y, g_coef, out_coef ? slope_out : 1.f, reinterpret_cast<const bf16_t *>(x),
                                           in_coef, slope_in, dw_partial, grid, stream, two ? two->xb : nullptr, two ? two->coef_b : nullptr,
                                           two ? two->slope_b : 1.f);
        if (rc) return rc;
        const int n = cout * cin;
        if (!i2p_defer_reduce(1, (int)grid, n, dw_partial, dw))
            hipLaunchKernelGGL(reduce_partials_bf16, dim3((n + 31) / 32), dim3(256), 0, st, (int)grid, n, dw_partial, dw);
        I2P_RETURN_LAUNCH_STATUS();
    }
    if (!two && grid == 256 && i2p_small_wgrad_bf16_ok(rows, cin, cout, x_bf16)) {
        // narrow level-1 layer on many rows: HBM streaming, one element per lane and k-step (csrc/mlp_wreg_bf16.hip)
        const int rc = i2p_small_wgrad_bf16(rows, cin, cout, gz, y, g_coef, out_coef ? slope_out : 1.f, x, x_bf16, in_coef, slope_in, dw_partial,
                                            grid, stream);
        if (rc) return rc;
        const int n = cout * cin;
        if (!i2p_defer_reduce(1, (int)grid, n, dw_partial, dw))
            hipLaunchKernelGGL(reduce_partials_bf16, dim3((n + 31) / 32), dim3(256), 0, st, (int)grid, n, dw_partial, dw);
        I2P_RETURN_LAUNCH_STATUS();
    }
    WgradP wq{};
    wq.rows = rows; wq.cin = cin; wq.cout = cout; wq.nco = cout / 8; wq.ncx = (cin + 7) / 8;
    wq.gz = gz; wq.y = y; wq.g_coef = g_coef; wq.g_slope = out_coef ? slope_out : 1.f;
    wq.x = x; wq.x_ld = two ? two->split : cin; wq.xb = two ? two->xb : nullptr; wq.xb_ld = two ? cin - two->split : 0;
    wq.split = two ? two->split : cin;
    wq.coef_a = in_coef; wq.coef_b = two ? two->coef_b : nullptr; wq.slope_a = slope_in; wq.slope_b = two ? two->slope_b : 1.f;
    wq.dw_partial = dw_partial;
    return wgrad_launch(wq, dw, st, grid, x_bf16 != 0);
}

}  // namespace

extern "C" int i2p_lin_bwd_bf16_grid(long long rows) {
    const long long ntiles = (rows + WG_R - 1) / WG_R;
    return (int)(ntiles < 256 ? (ntiles < 1 ? 1 : ntiles) : 256);
}

extern "C" int i2p_lin_fwd_bf16(long long rows, int cin, int cout, const void *x, int x_bf16, const float *in_coef,
                                float slope_in, const float *w, bf16_t *y, double *sums, void *stream) {
    return fwd_impl(rows, cin, cout, x, x_bf16, cin, nullptr, 0, 0, in_coef, slope_in, nullptr, 1.f, w, y, sums, nullptr, nullptr,
                    nullptr, 1, 1, stream);
}

extern "C" int i2p_lin_fwd_2src_bf16(long long rows, int cin_a, int cin_b, int cout, const bf16_t *xa, const float *coef_a,
                                     float slope_a, const bf16_t *xb, const float *coef_b, float slope_b, const float *w,
                                     bf16_t *y, double *sums, void *stream) {
    if (!xa || !xb || cin_a <= 0 || cin_b <= 0) return I2P_ERR_BAD_ARG;
    return fwd_impl(rows, cin_a + cin_b, cout, xa, 1, cin_a, xb, cin_b, cin_a, coef_a, slope_a, coef_b, slope_b, w, y, sums,
                    nullptr, nullptr, nullptr, 1, 1, stream);
}

extern "C" int i2p_pair_lin_fwd_bf16(int B, int N, int M, int cin, int cout, const float *f, const float *g,
                                     const float *bias_n, const float *bias_k, const float *w, bf16_t *y, double *sums,
                                     void *stream) {
    if (B <= 0 || N <= 0 || M <= 0 || (cin & 7) || cin > 128 || !f || !g || !bias_n || !bias_k) return I2P_ERR_BAD_ARG;
    if (w && y && i2p_pair_fwd3_bf16_ok(B, N, M, cin, cout))                 // strip shared by the block's four waves (csrc/pair_fwd_bf16.hip)
        return i2p_pair_fwd3_bf16(B, N, M, f, g, bias_n, bias_k, w, y, sums, stream);
    // (few rows: the row-order kernel; a pixel-tile-stationary single-wave form, 0.27 of 8 TB/s, was superseded by pair_fwd3 in round 5)
    return fwd_impl((long long)B * N * M, cin, cout, g, 0, cin, nullptr, 0, 0, nullptr, 1.f, nullptr, 1.f, w, y, sums, f, bias_n,
                    bias_k, N, M, stream);
}

extern "C" int i2p_lin_bwd_bf16(long long rows, int cin, int cout, const bf16_t *gz, const bf16_t *y, const float *out_coef,
                                const float *out_mi, const double *out_dsums, const void *x, int x_bf16, const float *in_coef,
                                const float *in_mi, float slope_in, const float *w, void *gz_in, int gz_in_bf16,
                                double *in_dsums, float *dw_partial, float *dw, float slope_out, void *stream) {
    return bwd_impl(rows, cin, cout, gz, y, out_coef, out_mi, out_dsums, x, x_bf16, in_coef, in_mi, slope_in, w, gz_in, gz_in_bf16,
                    in_dsums, dw_partial, dw, slope_out, nullptr, stream);
}

extern "C" int i2p_lin_bwd_2src_bf16(long long rows, int cin_a, int cin_b, int cout, const bf16_t *gz, const bf16_t *y,
                                     const float *out_coef, const float *out_mi, const double *out_dsums, const bf16_t *xa,
                                     const float *coef_a, const float *mi_a, float slope_a, const bf16_t *xb, const float *coef_b,
                                     const float *mi_b, float slope_b, const bf16_t *e_add_b, const float *w, bf16_t *gz_a,
                                     double *dsums_a, bf16_t *gz_b, double *dsums_b, float *dw_partial, float *dw, void *stream) {
    if (!xa || !xb || !coef_a || !coef_b || !gz_a || !gz_b || !dsums_a || !dsums_b) return I2P_ERR_BAD_ARG;
    TwoSrc t; t.split = cin_a; t.xb = xb; t.coef_b = coef_b; t.mi_b = mi_b; t.slope_b = slope_b; t.gz_b = gz_b; t.dsums_b = dsums_b;
    t.e_add = e_add_b;
    return bwd_impl(rows, cin_a + cin_b, cout, gz, y, out_coef, out_mi, out_dsums, xa, 1, coef_a, mi_a, slope_a, w, gz_a, 1, dsums_a,
                    dw_partial, dw, 1.f, &t, stream);
}

static int pair_bwd_gen1_grid(int B, int N, int M) {
    const int KT = (M + PB_PX - 1) / PB_PX;
    int NC = 256 / (B * KT > 0 ? B * KT : 1);
    NC = NC < 1 ? 1 : (NC > N ? N : NC);
    return B * KT * NC;
}

// blocks whose weight-gradient partials the caller's scratch must hold (whichever of the two kernels runs)
extern "C" int i2p_pair_lin_bwd_bf16_grid(int B, int N, int M) {
    const int g1 = pair_bwd_gen1_grid(B, N, M), g2 = i2p_pair_bwd2_bf16_grid(B, N, M);
    return g1 > g2 ? g1 : g2;
}

extern "C" int i2p_pair_lin_bwd_bf16(int B, int N, int M, int cin, int cout, const bf16_t *gz, const bf16_t *y,
                                     const float *out_coef, const float *out_mi, const double *out_dsums, const float *f,
                                     const float *g, const float *w, float *d_f, float *d_g, float *d_bias_n, float *d_bias_k,
                                     float *dw_partial, float *dw, void *stream) {
    auto ok = [](int c) { return c == 32 || c == 64 || c == 128; };
    if (B <= 0 || N <= 0 || M <= 0 || !ok(cin) || !ok(cout)) return I2P_ERR_BAD_ARG;
    if (!gz || !f || !g || !w || !d_f || !d_g || !d_bias_n || !d_bias_k || !dw_partial || !dw) return I2P_ERR_BAD_ARG;
    if (out_coef && (!y || !out_mi || !out_dsums)) return I2P_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (out_coef && i2p_pair_bwd2_bf16_ok(B, N, M, cin, cout)) {
        // 128 x 128 on many rows: persistent blocks over 32-pixel strips, three strips in flight per wave (csrc/pair_bwd_bf16.hip)
        const unsigned galloc = (unsigned)i2p_pair_lin_bwd_bf16_grid(B, N, M), g2 = (unsigned)i2p_pair_bwd2_bf16_grid(B, N, M);
        float *g_coef = dw_partial + (size_t)galloc * cout * cin;
        const int rc = i2p_pair_bwd2_bf16(B, N, M, gz, y, out_dsums, out_coef, out_mi, g_coef, f, g, w, d_f, d_g, d_bias_n, d_bias_k, dw_partial, stream);
        if (rc) return rc;
        const int nel = cout * cin;
        if (!i2p_defer_reduce(1, (int)g2, nel, dw_partial, dw))
            hipLaunchKernelGGL(reduce_partials_bf16, dim3((nel + 31) / 32), dim3(256), 0, st, (int)g2, nel, dw_partial, dw);
        I2P_RETURN_LAUNCH_STATUS();
    }
    PairBwdP p{};
    p.B = B; p.N = N; p.M = M; p.cin = cin; p.cout = cout;
    const int KT = (M + PB_PX - 1) / PB_PX;
    const unsigned grid = (unsigned)pair_bwd_gen1_grid(B, N, M);
    p.NC = (int)grid / (B * KT); p.NL = (N + p.NC - 1) / p.NC;
    p.gz = gz; p.y = y; p.f = f; p.g = g; p.w = w; p.d_f = d_f; p.d_g = d_g; p.d_bn = d_bias_n; p.d_bk = d_bias_k; p.dw_partial = dw_partial;
    if (out_coef) {
        float *g_coef = dw_partial + (size_t)grid * cout * cin;
        hipLaunchKernelGGL(bnbwd_coef_bf16, dim3((cout + 63) / 64), dim3(64), 0, st, (long long)B * N * M, cout, out_dsums, out_coef, out_mi, g_coef);
        p.g_coef = g_coef;
    }
    const int nco = cout / 8;
    int cps = 0; while ((1 << cps) < nco) ++cps;
    const size_t bytes = ((size_t)(cin << cps) + (size_t)(cin << 3) + 2 * ((size_t)(cout << 3) + ((size_t)PB_PX << cps))) * sizeof(uint4) +
                         2 * 16 * (size_t)cout * sizeof(float);
    if (bytes > 160 * 1024) return I2P_ERR_BAD_ARG;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pair_bwd_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(pair_bwd_bf16_kernel, dim3(grid), dim3(PB_THREADS), bytes, st, p);
    const int nel = cout * cin;
    if (!i2p_defer_reduce(1, (int)grid, nel, dw_partial, dw))
        hipLaunchKernelGGL(reduce_partials_bf16, dim3((nel + 31) / 32), dim3(256), 0, st, (int)grid, nel, dw_partial, dw);
    I2P_RETURN_LAUNCH_STATUS();
}
