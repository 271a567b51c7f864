// Level-1 set abstraction front end as ONE kernel: window K-NN selection + neighbour gather + feature build
// (reference: fused_conv_go.cu:49-238 for the selection; src/projectPN/utils.py:36-60 gathers;
// PPBackbone_center.py:157-187 feature build  [dxyz_raw(3), centre xyz(3), neighbour xyz_raw(3), |dxyz|(1)]).
//
// The reference (and this repository's first round) runs the selection, materialises three int64 [B,N,K] index
// tensors + a mask, gathers the neighbour coordinates row by row, subtracts, takes norms and concatenates: ~10
// launches and 34 MB of intermediates per sample at level 1.  Here a block owns 16 consecutive query cells of one
// output row; their 9 x 15 windows overlap almost completely, so the block stages the union strip
// (kH rows x (kW + 15*stride_w) columns) of the selection image in LDS once (round 3: the raw coordinates are gathered at the end)
// — "LDS staging of per-group neighbourhoods" — selects exactly like fcsk_kernel (16-lane DPP rows, sorted per-lane
// runs, equal distances redone serially in the reference's order), and writes the 10-channel feature rows (padded to
// 12 floats = three 16-byte stores per neighbour) straight from the staged strip.  No index tensor exists.
// Semantics of empty slots / empty centres are the reference's with FLAG_SHIFT|FLAG_COPY: empty slots repeat the
// nearest hit, a query without any hit (or with an empty centre) gathers cell (0,0).
#include "common.h"

namespace {

constexpr int GROUP = 16, QPB = 16, SA_THREADS = GROUP * QPB;
constexpr unsigned SENT_BITS = 0x501502F9u, PAD_BITS = 0x7FFFFFFFu;
constexpr unsigned CODE_STORED = 0x100u, CODE_VALID = 0x200u;
constexpr int STAGE_IT = 4;               // 16-byte staging loads per thread and image, all in flight before the first LDS store

struct SaParams {
    int B, H, W, out_h, out_w, stride_h, stride_w, kH, kW, K;
    float dist2;
    const float *sel_xyz, *raw_xyz;       // [B,H,W,3]: selection coordinates (centres + candidates), raw coordinates (features)
    float *feat;                          // [B, out_h*out_w, K, 12]
    int sw;                               // strip width in cells = kW + 15*stride_w
    int pad_l;                            // cells staged to the left of the first window column (16-byte alignment of the strip rows)
    int swf;                              // strip row pitch in floats
    int region_f;                         // floats of the shared strip / parked-runs region (outc follows it)
    int force_serial;
};

// LDS accesses of one 16-lane query row are private to its wave: ordering them needs the compiler to keep program order
// (the hardware executes a wave's LDS operations in order), not a block barrier.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// VEC: the strip rows start on a 16-byte boundary of the image row and are whole float4s (launcher checks), so a block stages
// 2 x kH x swf/4 aligned 16-byte loads — all issued before the first LDS store: one global round trip per block instead of five.
template <int SLOTS, bool VEC>
__global__ __launch_bounds__(SA_THREADS) void sa_l1_kernel(SaParams p) {
    // Dynamic LDS: region R = max(strip sel [kH][swf] floats, parked runs), then outc u16 [QPB][K + 2 rounded up to 4].
    // Round 6: the parked sorted runs (distance bits [QPB][SLOTS*16] u32 + (position | stored) codes [QPB][SLOTS*16] u16, 13.8 KB at
    // level 1) live IN the strip's 14.7 KB once every wave of the block has evaluated its distances — the strip is only read while the
    // candidates are evaluated; the output rows gather the raw coordinates from global memory.  30.2 -> 16.5 KB per block = 8
    // blocks per CU instead of 5: the 1920 blocks of a batch-8 launch are resident at once (2048 slots) instead of running in two
    // phase-locked rounds whose stores did not overlap anything (profiles/r05_sa_l1_ablation.txt).  Price: one more block barrier,
    // and the (rare) tie redo re-reads its candidates from global memory.
    extern __shared__ float strip[];
    __shared__ int tab[SLOTS * GROUP];
    unsigned (*lst_d)[SLOTS * GROUP] = reinterpret_cast<unsigned (*)[SLOTS * GROUP]>(strip);
    unsigned short (*lst_c)[SLOTS * GROUP] = reinterpret_cast<unsigned short (*)[SLOTS * GROUP]>(strip + QPB * SLOTS * GROUP);

    const int kt = p.kH * p.kW;
    const int tid = threadIdx.x, g = tid >> 4, l16 = tid & 15;
    const int wblocks = (p.out_w + QPB - 1) / QPB;
    int bid = (int)i2p_xcd_swizzle(blockIdx.x, gridDim.x);    // a sample's blocks on one XCD: its two images stay in that L2
    const int wb = bid % wblocks; bid /= wblocks;
    const int qh = bid % p.out_h; const int b = bid / p.out_h;
    const int qw = wb * QPB + g;
    const bool in_range = qw < p.out_w;
    const int ch = qh * p.stride_h, cw0 = wb * QPB * p.stride_w;
    const int h_lo = ch - p.kH / 2, w_lo = cw0 - p.kW / 2 - p.pad_l;          // image cell of strip cell (0, 0)
    // (round 3: only the SELECTION image is staged; the raw coordinates of the <= K selected cells and of the centre are gathered
    // from global memory / L2 at the end — 15 KB less LDS per block = 5 instead of 3 blocks per CU hiding each other's round trips)
    float *ssel = strip;
    const int ocp = (p.K + 2 + 3) & ~3;
    unsigned short *outc_g = reinterpret_cast<unsigned short *>(strip + p.region_f) + g * ocp;     // this query's selected codes
    const size_t img = (size_t)b * p.H * p.W * 3;

    // what an unset slot (and every slot of an empty centre) gathers: cell (0,0) of the raw image — requested now, so that the
    // output phase does not start with a dependent global round trip
    const float c00x = p.raw_xyz[img], c00y = p.raw_xyz[img + 1], c00z = p.raw_xyz[img + 2];
    float crx = 0.f, cry = 0.f, crz = 0.f;                 // this query's centre in the raw image
    if (in_range) {
        const size_t o = img + ((size_t)ch * p.W + (size_t)qw * p.stride_w) * 3;
        crx = p.raw_xyz[o]; cry = p.raw_xyz[o + 1]; crz = p.raw_xyz[o + 2];
    }
    // ---- stage the strip of both images (zeros outside the image rows; columns wrap: FLAG_SHIFT) ----------------
    if (VEC) {
        const int n4row = p.swf >> 2, n4 = p.kH * n4row, w3 = p.W * 3;
        float4 va[STAGE_IT];
#pragma unroll
        for (int it = 0; it < STAGE_IT; ++it) {
            const int i = tid + it * SA_THREADS;
            va[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n4) {
                const int r = i / n4row, j = i - r * n4row;
                const int h = h_lo + r;
                int fo = w_lo * 3 + 4 * j;                 // float offset inside the image row; wraps on a 16-byte boundary
                if (fo < 0) fo += w3;
                if (fo >= w3) fo -= w3;
                if (h >= 0 && h < p.H) {
                    const size_t o = img + (size_t)h * w3 + fo;
                    va[it] = *reinterpret_cast<const float4 *>(p.sel_xyz + o);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < STAGE_IT; ++it) {
            const int i = tid + it * SA_THREADS;
            if (i < n4) reinterpret_cast<float4 *>(ssel)[i] = va[it];
        }
    } else {
        const int cells = p.swf / 3, ncell = p.kH * cells;
        for (int i = tid; i < ncell; i += SA_THREADS) {
            const int r = i / cells, cidx = i - r * cells;
            const int h = h_lo + r;
            int w = w_lo + cidx;
            if (w < 0) w += p.W;
            if (w >= p.W) w -= p.W;
            float sx = 0.f, sy = 0.f, sz = 0.f;
            if (h >= 0 && h < p.H && w >= 0 && w < p.W) {
                const size_t o = img + ((size_t)h * p.W + w) * 3;
                sx = p.sel_xyz[o]; sy = p.sel_xyz[o + 1]; sz = p.sel_xyz[o + 2];
            }
            ssel[i * 3] = sx; ssel[i * 3 + 1] = sy; ssel[i * 3 + 2] = sz;
        }
    }
    for (int i = tid; i < SLOTS * GROUP; i += SA_THREADS) {
        int v = 0;
        if (i < kt) { const int dh = i / p.kW - p.kH / 2, dw = i % p.kW - p.kW / 2; v = (dh << 16) | (dw & 0xffff); }   // random_hw = arange (utils.py:84)
        tab[i] = v;
    }
    __syncthreads();                                       // strip and table are block-shared
    const bool wave_in = __any(in_range ? 1 : 0);          // false: a wave of queries beyond out_w (last block of a row) — it still meets the barrier below

    // centre of this query: strip cell (kH/2, pad_l + kW/2 + g*stride_w)
    const int ccol = p.pad_l + p.kW / 2 + g * p.stride_w;
    const int cc = (p.kH / 2) * p.swf + ccol * 3;
    const float cx = ssel[cc], cy = ssel[cc + 1], cz = ssel[cc + 2];
    const bool live = wave_in && in_range && !(fmaxf(i2p_sq3(cx, cy, cz), 1e-10f) <= 1e-10f);          // go.cu:72-74
    // window cell (dh, dw) of this query -> strip cell; rows outside the image were staged as empty cells, which the
    // reference skips just like a zero point (go.cu:99-103 vs :143: both leave the slot unset)
    auto eval = [&](int tabv, unsigned &dbits, unsigned &stored) {
        dbits = SENT_BITS; stored = 0;
        const int r = p.kH / 2 + (tabv >> 16), cidx = ccol + (int)(short)(tabv & 0xffff);
        const float *q = ssel + r * p.swf + cidx * 3;
        const float xq = q[0], yq = q[1], zq = q[2];
        if (i2p_sq3(xq, yq, zq) <= 1e-10f) return;                                            // go.cu:141-143
        const float dq = fmaxf(i2p_sq3(cx - xq, cy - yq, cz - zq), 1e-10f);                   // go.cu:154
        if (dq > p.dist2) return;                                                             // go.cu:157
        dbits = i2p_f2u(dq); stored = 1;
    };

    // the tie redo's candidate evaluation: the same arithmetic on the cell read from global memory (the strip is gone by then)
    auto eval_global = [&](int tabv, unsigned &dbits, unsigned &stored) {
        dbits = SENT_BITS; stored = 0;
        const int h = ch + (tabv >> 16);
        if (h < 0 || h >= p.H) return;                                                        // staged as an empty cell
        int w = qw * p.stride_w + (int)(short)(tabv & 0xffff);
        if (w < 0) w += p.W;
        if (w >= p.W) w -= p.W;
        const float *q = p.sel_xyz + img + ((size_t)h * p.W + w) * 3;
        const float xq = q[0], yq = q[1], zq = q[2];
        if (i2p_sq3(xq, yq, zq) <= 1e-10f) return;
        const float dq = fmaxf(i2p_sq3(cx - xq, cy - yq, cz - zq), 1e-10f);
        if (dq > p.dist2) return;
        dbits = i2p_f2u(dq); stored = 1;
    };

    // A wave whose four queries all have empty centres (93 % of the centres of a 8192-point scan) selects nothing: it goes
    // straight to the output rows.  Everything from the second barrier to the output is private to the wave (lst[g], outc[g]).
    const bool wave_live = __any(live ? 1 : 0);
    // ---- A: evaluate and sort own keys (fcsk_kernel step A); the keys stay in registers across the barrier ---------------
    unsigned long long key[SLOTS];
    if (wave_live) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int pos = s * GROUP + l16;
            unsigned dbits = PAD_BITS, stored = 0;
            if (pos < kt) { dbits = SENT_BITS; if (live) eval(tab[pos], dbits, stored); }
            key[s] = ((unsigned long long)dbits << 32) | (unsigned)(pos | (stored << 8));
        }
#pragma unroll
        for (int i = 0; i < SLOTS - 1; ++i)
#pragma unroll
            for (int j = 0; j < SLOTS - 1 - i; ++j) {
                const unsigned long long a = key[j], b2 = key[j + 1];
                const bool sw_ = a > b2;
                key[j] = sw_ ? b2 : a; key[j + 1] = sw_ ? a : b2;
            }
    }
    __syncthreads();                                       // every wave has read the strip for the last time: its LDS becomes the parked runs
    if (wave_live) {
        for (int i = l16; i < p.K; i += GROUP) outc_g[i] = 0;
        // ---- B: park the runs (fcsk_kernel step B) ---------------------------------------------------------------
#pragma unroll
        for (int s = 1; s < SLOTS; ++s) { lst_d[g][s * GROUP + l16] = (unsigned)(key[s] >> 32); lst_c[g][s * GROUP + l16] = (unsigned short)key[s]; }
        wave_lds_sync();

        // ---- C: K extraction steps (ends as soon as every query of the wave is done) ----------------------------
        // head = the lane's smallest remaining key, nxt = the one behind it (already in registers): a pop promotes nxt and
        // requests the following key from LDS, whose latency is off the critical path unless the same lane pops twice in a row
        unsigned head_hi = (unsigned)(key[0] >> 32), head_lo = (unsigned)key[0];
        unsigned nxt_hi = SLOTS > 1 ? (unsigned)(key[SLOTS > 1 ? 1 : 0] >> 32) : PAD_BITS, nxt_lo = SLOTS > 1 ? (unsigned)key[SLOTS > 1 ? 1 : 0] : 0u;
        int ptr = 2;
        unsigned pops = 0, steps = 0, prev = 0;
        bool tie = false, done = !live;
        for (int s = 0; s < p.K; ++s) {
            if (!__any(done ? 0 : 1)) break;
            const unsigned gmin = i2p_row16_min_u32(head_hi);
            if (!done) {
                if (gmin >= SENT_BITS) done = true;
                else {
                    tie |= (s > 0 && gmin == prev);
                    prev = gmin; ++steps;
                    if (head_hi == gmin) {
                        outc_g[s] = (unsigned short)(head_lo | CODE_VALID);
                        ++pops;
                        head_hi = nxt_hi; head_lo = nxt_lo;
                        nxt_hi = PAD_BITS; nxt_lo = 0;
                        if (ptr < SLOTS) { nxt_hi = lst_d[g][ptr * GROUP + l16]; nxt_lo = lst_c[g][ptr * GROUP + l16]; }
                        ++ptr;
                    }
                }
            }
        }
        {
            const unsigned gnext = i2p_row16_min_u32(head_hi);
            const unsigned total = i2p_row16_add_u32(pops);
            if (!done && steps > 0 && gnext == prev) tie = true;
            if (total != steps) tie = true;
        }
        const bool need_serial = live && (tie || p.force_serial);
        if (__any(need_serial ? 1 : 0)) {                                  // ---- F: reference's serial order on ties
            wave_lds_sync();
            if (need_serial) {
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
                    const int pos = s * GROUP + l16;
                    unsigned dbits = SENT_BITS, stored = 0;
                    if (pos < kt) eval_global(tab[pos], dbits, stored);
                    lst_d[g][pos] = dbits; lst_c[g][pos] = (unsigned short)(pos | (stored << 8));
                }
            }
            wave_lds_sync();
            if (need_serial && l16 == 0) {
                unsigned *ad = lst_d[g]; unsigned short *ac = lst_c[g];
                for (int s = 0; s < p.K; ++s) {                                                    // go.cu:183-236
                    int mi = s;
                    if (s < kt) {
                        float dm = i2p_u2f(ad[s]);
                        for (int t = s + 1; t < kt; ++t) {
                            const float dt = i2p_u2f(ad[t]);
                            if (dt < dm) { dm = dt; mi = t; }
                        }
                        if (mi != s) {
                            const unsigned td = ad[mi]; ad[mi] = ad[s]; ad[s] = td;
                            const unsigned short tc = ac[mi]; ac[mi] = ac[s]; ac[s] = tc;
                        }
                        const float ds = i2p_u2f(ad[s]);
                        outc_g[s] = (unsigned short)((ac[s] & 0x1ffu) | (ds < 1e10f ? CODE_VALID : 0u));
                    } else outc_g[s] = 0;
                }
            }
        }
        wave_lds_sync();
    }
    if (!in_range) return;

    // ---- feature rows: [nbr_raw - centre_raw, centre, nbr_raw, |d|, 0, 0] ----------------------------------------
    const size_t obase = (((size_t)b * p.out_h + qh) * p.out_w + qw) * p.K;
    const unsigned copy_code = live ? outc_g[0] : 0u;
    for (int s = l16; s < p.K; s += GROUP) {
        unsigned code = live ? outc_g[s] : 0u;
        if (!(code & CODE_VALID)) code = copy_code;                    // FLAG_COPY (go.cu:211-222); empty centre: (0,0)
        float nx, ny, nz;
        if (code & CODE_STORED) {
            const int tv = tab[code & 0xff];
            const int h = ch + (tv >> 16);                             // (a stored cell lies inside the image rows)
            int w = qw * p.stride_w + (int)(short)(tv & 0xffff);
            if (w < 0) w += p.W;
            if (w >= p.W) w -= p.W;
            const float *q = p.raw_xyz + img + ((size_t)h * p.W + w) * 3;
            nx = q[0]; ny = q[1]; nz = q[2];
        } else { nx = c00x; ny = c00y; nz = c00z; }
        const float dx = nx - crx, dy = ny - cry, dz = nz - crz;
        const float dn = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
        float4 *o = reinterpret_cast<float4 *>(p.feat + (obase + s) * 12);
#if defined(SA_ABL) && (SA_ABL & 1)
        if (dn == 12345.678f)                                           // ablation build (-DSA_ABL=1): no feature stores — profiles/r05_sa_l1_ablation.txt
#endif
        { o[0] = make_float4(dx, dy, dz, cx); o[1] = make_float4(cy, cz, nx, ny); o[2] = make_float4(nz, dn, 0.f, 0.f); }
    }
}

template <int SLOTS, bool VEC>
int launch(const SaParams &p0, hipStream_t st) {
    SaParams p = p0;
    const int lists_f = QPB * SLOTS * GROUP + (QPB * SLOTS * GROUP + 1) / 2;           // u32 distances + u16 codes, in floats
    p.region_f = ((p.kH * p.swf > lists_f ? p.kH * p.swf : lists_f) + 3) & ~3;
    const size_t bytes = (size_t)p.region_f * sizeof(float) + (size_t)QPB * ((p.K + 2 + 3) & ~3) * sizeof(unsigned short);
    if (bytes > 96 * 1024) return I2P_ERR_BAD_ARG;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(sa_l1_kernel<SLOTS, VEC>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_set = true;
    }
    const unsigned grid = (unsigned)((long long)p.B * p.out_h * ((p.out_w + QPB - 1) / QPB));
    hipLaunchKernelGGL((sa_l1_kernel<SLOTS, VEC>), dim3(grid), dim3(SA_THREADS), bytes, st, p);
    I2P_RETURN_LAUNCH_STATUS();
}

template <int SLOTS>
int launch_any(const SaParams &p, bool vec, hipStream_t st) {
    return vec ? launch<SLOTS, true>(p, st) : launch<SLOTS, false>(p, st);
}

}  // namespace

// Level-1 grouping features: feat [B, out_h*out_w, K, 12] (10 channels + 2 zeros), queries = the strided cells
// (qh*stride_h, qw*stride_w) of the H x W images, window kH x kW around them with column wrap, distance limit.
extern "C" int i2p_sa_l1_group(int B, int H, int W, int out_h, int out_w, int stride_h, int stride_w, int kH, int kW, int K,
                               float distance, const float *sel_xyz, const float *raw_xyz, float *feat, void *stream) {
    if (B < 0 || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0 || stride_h <= 0 || stride_w <= 0 || kH <= 0 || kW <= 0 || K <= 0)
        return I2P_ERR_BAD_ARG;
    const int kt = kH * kW;
    if (kt > I2P_MAX_WINDOW) return I2P_ERR_WINDOW;
    if (K > I2P_MAX_WINDOW) return I2P_ERR_K;
    if ((out_h - 1) * stride_h >= H || (out_w - 1) * stride_w >= W) return I2P_ERR_BAD_ARG;
    if (B == 0) return 0;
    if (!sel_xyz || !raw_xyz || !feat) return I2P_ERR_BAD_ARG;
    SaParams p;
    p.B = B; p.H = H; p.W = W; p.out_h = out_h; p.out_w = out_w; p.stride_h = stride_h; p.stride_w = stride_w;
    p.kH = kH; p.kW = kW; p.K = K; p.dist2 = distance * distance; p.sel_xyz = sel_xyz; p.raw_xyz = raw_xyz; p.feat = feat;
    p.sw = kW + (QPB - 1) * stride_w;
    p.force_serial = !(p.dist2 < 1e10f);
    if (p.sw > W) return I2P_ERR_BAD_ARG;                              // (a strip never wraps onto itself)
    // 16-byte staging: every strip row must start on a 16-byte boundary of its image row and hold whole float4s, and the
    // column wrap must fall between float4s: (kW/2 + pad_l) % 4 == 0, cells per row % 4 == 0, (16*stride_w) % 4 == 0, W % 4 == 0
    const int pad_l = (4 - (kW / 2) % 4) % 4;
    const int cells = (p.sw + pad_l + 3) / 4 * 4;
    const bool vec = (W % 4 == 0) && ((QPB * stride_w) % 4 == 0) && cells <= W && kH * (cells * 3 / 4) <= SA_THREADS * STAGE_IT &&
                     ((reinterpret_cast<uintptr_t>(sel_xyz) | reinterpret_cast<uintptr_t>(raw_xyz)) & 15) == 0 &&
                     !getenv("I2P_SA_SCALAR_STAGE");
    p.pad_l = vec ? pad_l : 0;
    p.swf = (vec ? cells : p.sw) * 3;
    hipStream_t st = (hipStream_t)stream;
    if (kt <= 16) return launch_any<1>(p, vec, st);
    if (kt <= 48) return launch_any<3>(p, vec, st);
    if (kt <= 144) return launch_any<9>(p, vec, st);
    return launch_any<10>(p, vec, st);
}


// ---------------------------------------------------------------------------------------------------------------------
// Set-abstraction levels 2-4, up-convolutions: the grouped input rows of the point MLP in ONE launch from the selected
// cells (reference: gather_torch x2 + subtraction + cat, PPBackbone_center.py:94-129):
//   out[b, n*K + k, xo:xo+3] = xyz[b, cell, 0:3] - centre[b, n, 0:3],  out[.., fo:fo+C] = feat[b, cell, 0:C],  0 elsewhere
// (set abstraction: xo = 0, fo = 3; up-convolution: fo = 0, xo = C)
// cell = h_idx * W + w_idx of the row.  One thread per (row, float4 of the output).
// ---------------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void sa_rows_kernel(int hw, int q, int K, int W, int c, int cpad, int xo, int fo, const float *__restrict__ xyz,
                                                      const float *__restrict__ centre, const float *__restrict__ feat,
                                                      const int64_t *__restrict__ h_idx, const int64_t *__restrict__ w_idx,
                                                      float *__restrict__ out) {
    const int bi = blockIdx.y, c4 = cpad >> 2;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)q * c4) return;
    const int row = (int)(t / c4), j = (int)(t % c4);
    const long long cell = h_idx[(size_t)bi * q + row] * W + w_idx[(size_t)bi * q + row];
    const float *fr = feat + ((size_t)bi * hw + cell) * c;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ch = 4 * j + e;
        float x = 0.f;
        if (ch >= xo && ch < xo + 3) x = xyz[((size_t)bi * hw + cell) * 3 + ch - xo] - centre[((size_t)bi * (q / K) + row / K) * 3 + ch - xo];
        else if (ch >= fo && ch < fo + c) x = fr[ch - fo];
        v[e] = x;
    }
    *reinterpret_cast<float4 *>(out + ((size_t)bi * q + row) * cpad + 4 * j) = make_float4(v[0], v[1], v[2], v[3]);
}

// Input rows of the kNN pi-stage of the fine cost volume (PPBackbone_center.py:369-395: knn grouping of the pixel features,
// product with the point feature, concat with the two coordinate triples) in one launch from the indices:
// out[b, n*K + k, :] = [xyz[b,n,:] (3), pix_xyz[b,idx,:] (3), pts[b,n,:] * pix[b,idx,:] (C), zero padding]; idx i64 [b, n*K].
__global__ __launch_bounds__(256) void knn_rows_fwd_kernel(int n, int m, int K, int c, int cpad, const float *__restrict__ xyz,
                                                           const float *__restrict__ pix_xyz, const float *__restrict__ pts,
                                                           const float *__restrict__ pix, const int64_t *__restrict__ idx,
                                                           float *__restrict__ out) {
    const int bi = blockIdx.y, c4 = cpad >> 2;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)n * K * c4) return;
    const int row = (int)(t / c4), j = (int)(t % c4), pn = row / K;
    const long long cell = idx[(size_t)bi * n * K + row];
    const float *pr = pix + ((size_t)bi * m + cell) * c, *fr = pts + ((size_t)bi * n + pn) * c;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ch = 4 * j + e;
        float x = 0.f;
        if (ch < 3) x = xyz[((size_t)bi * n + pn) * 3 + ch];
        else if (ch < 6) x = pix_xyz[((size_t)bi * m + cell) * 3 + ch - 3];
        else if (ch < 6 + c) x = fr[ch - 6] * pr[ch - 6];
        v[e] = x;
    }
    *reinterpret_cast<float4 *>(out + ((size_t)bi * n * K + row) * cpad + 4 * j) = make_float4(v[0], v[1], v[2], v[3]);
}

// backward of the rows: block = one point (b, n), thread = one feature channel; the K neighbours in order (fixed summation order).
//   d_pts[b,n,c] = sum_k g[row, 6+c] * pix[b,idx,c];  gq[row, c] = g[row, 6+c] * pts[b,n,c] (scattered to the pixels by
//   i2p_gather_rows_grad_fx afterwards);  d_xyz[b,n,:] = sum_k g[row, 0:3]
__global__ __launch_bounds__(128) void knn_rows_bwd_kernel(int n, int m, int K, int c, int cpad, const float *__restrict__ g,
                                                           const float *__restrict__ pts, const float *__restrict__ pix,
                                                           const int64_t *__restrict__ idx, float *__restrict__ d_xyz,
                                                           float *__restrict__ d_pts, float *__restrict__ gq) {
    const int pn = blockIdx.x, bi = blockIdx.y;
    const size_t row0 = ((size_t)bi * n + pn) * K;
    for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
        const float f = pts[((size_t)bi * n + pn) * c + ch];
        float acc = 0.f;
        int k = 0;
        for (; k + 3 < K; k += 4) {          // four neighbours' loads in flight (index -> pixel row is a dependent pair); same summation order
            long long cell[4]; float gv[4], pv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) cell[u] = idx[row0 + k + u];
#pragma unroll
            for (int u = 0; u < 4; ++u) { gv[u] = g[(row0 + k + u) * cpad + 6 + ch]; pv[u] = pix[((size_t)bi * m + cell[u]) * c + ch]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc = fmaf(gv[u], pv[u], acc); gq[(row0 + k + u) * c + ch] = gv[u] * f; }
        }
        for (; k < K; ++k) {
            const long long cell = idx[row0 + k];
            const float gv = g[(row0 + k) * cpad + 6 + ch];
            acc = fmaf(gv, pix[((size_t)bi * m + cell) * c + ch], acc);
            gq[(row0 + k) * c + ch] = gv * f;
        }
        d_pts[((size_t)bi * n + pn) * c + ch] = acc;
    }
    if (threadIdx.x < 3 && d_xyz) {
        float a = 0.f;
        for (int k = 0; k < K; ++k) a += g[(row0 + k) * cpad + threadIdx.x];
        d_xyz[((size_t)bi * n + pn) * 3 + threadIdx.x] = a;
    }
}
// ---------------------------------------------------------------------------------------------------------------------
// pc-stage front end of the cost volumes (PPBackbone_center.py:443-476: two gather_torch, expand, subtraction, squared norm,
// sqrt, two cats — ~10 launches forward, ~16 backward per cost volume) in one launch each way from the selected cells:
//   geo  [b, n*K+k, 0:12] = [xyz[b,n] (3), xyz[b,cell] (3), xyz[b,cell] - xyz[b,n] (3), sqrt(|diff|^2 + 1e-20), 0, 0]
//   part [b, n*K+k, :]    = [pts[b,n,:] (C), feat[b,cell,:] (c)]        (columns 64.. of the mask MLP's 256-channel input)
//   nbf  [b, n*K+k, :]    = feat[b,cell,:]                               (the values of the softmax-weighted sum)
// cell = h_idx*W + w_idx.  One thread per float4 of a row's outputs.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pc_rows_fwd_kernel(int hw, int K, int W, int C, int c, const float *__restrict__ xyz,
                                                          const float *__restrict__ pts, const float *__restrict__ feat,
                                                          const int64_t *__restrict__ h_idx, const int64_t *__restrict__ w_idx,
                                                          float *__restrict__ geo, float *__restrict__ part, float *__restrict__ nbf) {
    const int bi = blockIdx.y;
    const int per = 3 + ((C + c) >> 2);                       // float4 tasks per row: 3 of geo, (C+c)/4 of part (the c/4 last also fill nbf)
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long q = (long long)hw * K;
    if (t >= q * per) return;
    const int row = (int)(t / per), j = (int)(t % per), n = row / K;
    const long long cell = h_idx[(size_t)bi * q + row] * W + w_idx[(size_t)bi * q + row];
    const size_t orow = (size_t)bi * q + row;
    if (j < 3) {
        const float *o = xyz + ((size_t)bi * hw + n) * 3, *nb = xyz + ((size_t)bi * hw + cell) * 3;
        const float ox = o[0], oy = o[1], oz = o[2], nx = nb[0], ny = nb[1], nz = nb[2];
        const float dx = nx - ox, dy = ny - oy, dz = nz - oz;
        float4 v;
        if (j == 0) v = make_float4(ox, oy, oz, nx);
        else if (j == 1) v = make_float4(ny, nz, dx, dy);
        else v = make_float4(dz, sqrtf(__fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)), 1e-20f)), 0.f, 0.f);
        reinterpret_cast<float4 *>(geo + orow * 12)[j] = v;
        return;
    }
    const int ch = (j - 3) * 4;
    float4 v;
    if (ch < C) v = *reinterpret_cast<const float4 *>(pts + ((size_t)bi * hw + n) * C + ch);
    else {
        v = *reinterpret_cast<const float4 *>(feat + ((size_t)bi * hw + cell) * c + (ch - C));
        *reinterpret_cast<float4 *>(nbf + orow * c + (ch - C)) = v;
    }
    *reinterpret_cast<float4 *>(part + orow * (C + c) + ch) = v;
}

// backward: block = one point (b, n); its K neighbours in index order (fixed summation order).
//   d_pts[b,n,:]              = sum_k g_part[row, 0:C]
//   comb[b,n, c:c+3]          = sum_k (g_own - g_diff - g_euc * diff/euc)      (own-point part of d_xyz; comb[b,n,0:c] = 0, [c+3] = 0)
//   rows[row, 0:c]            = g_part[row, C:] + g_nbf[row, :]                (per-neighbour gradient of feat[cell])
//   rows[row, c:c+3]          = g_nb + g_diff + g_euc * diff/euc               (per-neighbour gradient of xyz[cell]); rows[row, c+3] = 0
// `rows` is then scattered onto comb by the fixed-point row scatter (i2p_gather_rows_grad_fx): comb = [d_feat | d_xyz | 0].
__global__ __launch_bounds__(128) void pc_rows_bwd_kernel(int hw, int K, int W, int C, int c, const float *__restrict__ xyz,
                                                          const int64_t *__restrict__ h_idx, const int64_t *__restrict__ w_idx,
                                                          const float *__restrict__ g_geo, const float *__restrict__ g_part,
                                                          const float *__restrict__ g_nbf, float *__restrict__ d_pts,
                                                          float *__restrict__ comb, float *__restrict__ rows) {
    const int n = blockIdx.x, bi = blockIdx.y, tid = threadIdx.x;
    const size_t row0 = ((size_t)bi * hw + n) * K;
    const int ld = c + 4;
    for (int ch = tid; ch < C; ch += 128) {
        float a = 0.f;
        for (int k = 0; k < K; ++k) a += g_part[(row0 + k) * (C + c) + ch];
        d_pts[((size_t)bi * hw + n) * C + ch] = a;
    }
    for (int ch = tid; ch < c; ch += 128) {
        for (int k = 0; k < K; ++k)
            rows[(row0 + k) * ld + ch] = g_part[(row0 + k) * (C + c) + C + ch] + (g_nbf ? g_nbf[(row0 + k) * c + ch] : 0.f);
        comb[((size_t)bi * hw + n) * ld + ch] = 0.f;
    }
    if (tid < 4) {
        float own = 0.f;
        const float o = tid < 3 ? xyz[((size_t)bi * hw + n) * 3 + tid] : 0.f;
        for (int k = 0; k < K; ++k) {
            float v = 0.f;
            if (tid < 3 && g_geo) {
                const long long cell = h_idx[row0 + k] * W + w_idx[row0 + k];
                const float *nb = xyz + ((size_t)bi * hw + cell) * 3, *oo = xyz + ((size_t)bi * hw + n) * 3;
                const float dx = nb[0] - oo[0], dy = nb[1] - oo[1], dz = nb[2] - oo[2];
                const float euc = sqrtf(__fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)), 1e-20f));
                const float d = tid == 0 ? dx : (tid == 1 ? dy : dz);
                const float *gg = g_geo + (row0 + k) * 12;
                const float ge = gg[9] * (d / euc);                              // d euc / d diff
                v = gg[3 + tid] + gg[6 + tid] + ge;
                own += gg[tid] - gg[6 + tid] - ge;
                (void)o;
            }
            rows[(row0 + k) * ld + c + tid] = v;
        }
        comb[((size_t)bi * hw + n) * ld + c + tid] = own;
    }
}

}  // namespace

extern "C" int i2p_pc_rows_fwd(int b, int hw, int K, int W, int C, int c, const float *xyz, const float *pts, const float *feat,
                               const int64_t *h_idx, const int64_t *w_idx, float *geo, float *part, float *nbf, void *stream) {
    if (b <= 0 || hw <= 0 || K <= 0 || W <= 0 || C < 0 || c <= 0 || (C & 3) || (c & 3)) return I2P_ERR_BAD_ARG;
    if (!xyz || (C && !pts) || !feat || !h_idx || !w_idx || !geo || !part || !nbf) return I2P_ERR_BAD_ARG;
    const long long tot = (long long)hw * K * (3 + ((C + c) >> 2));
    hipLaunchKernelGGL(pc_rows_fwd_kernel, dim3((unsigned)((tot + 255) / 256), b), dim3(256), 0, (hipStream_t)stream, hw, K, W, C, c, xyz, pts, feat,
                       h_idx, w_idx, geo, part, nbf);
    I2P_RETURN_LAUNCH_STATUS();
}

// g_geo / g_nbf may be NULL (no gradient arrived on that output); comb [b, hw, c+4], rows [b, hw*K, c+4] are written completely
extern "C" int i2p_pc_rows_bwd(int b, int hw, int K, int W, int C, int c, const float *xyz, const int64_t *h_idx, const int64_t *w_idx,
                               const float *g_geo, const float *g_part, const float *g_nbf, float *d_pts, float *comb, float *rows, void *stream) {
    if (b <= 0 || hw <= 0 || K <= 0 || W <= 0 || C < 0 || c <= 0 || (C & 3) || (c & 3)) return I2P_ERR_BAD_ARG;
    if (!xyz || !h_idx || !w_idx || !g_part || (C && !d_pts) || !comb || !rows) return I2P_ERR_BAD_ARG;
    hipLaunchKernelGGL(pc_rows_bwd_kernel, dim3(hw, b), dim3(128), 0, (hipStream_t)stream, hw, K, W, C, c, xyz, h_idx, w_idx, g_geo, g_part, g_nbf,
                       d_pts, comb, rows);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_sa_rows(int b, int hw, int n, int K, int W, int c, int cpad, int xyz_col, int feat_col, const float *xyz,
                           const float *centre, const float *feat, const int64_t *h_idx, const int64_t *w_idx, float *out, void *stream) {
    if (b <= 0 || hw <= 0 || n <= 0 || K <= 0 || W <= 0 || c < 0 || (cpad & 3) || cpad < 3 + c) return I2P_ERR_BAD_ARG;
    if (xyz_col < 0 || feat_col < 0 || xyz_col + 3 > cpad || feat_col + c > cpad || (xyz_col < feat_col + c && feat_col < xyz_col + 3 && c > 0))
        return I2P_ERR_BAD_ARG;
    if (!xyz || !centre || (c && !feat) || !h_idx || !w_idx || !out) return I2P_ERR_BAD_ARG;
    const long long tot = (long long)n * K * (cpad >> 2);
    hipLaunchKernelGGL(sa_rows_kernel, dim3((unsigned)((tot + 255) / 256), b), dim3(256), 0, (hipStream_t)stream, hw, n * K, K, W, c, cpad,
                       xyz_col, feat_col, xyz, centre, feat, h_idx, w_idx, out);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_knn_rows_fwd(int b, int n, int m, int K, int c, int cpad, const float *xyz, const float *pix_xyz, const float *pts,
                                const float *pix, const int64_t *idx, float *out, void *stream) {
    if (b <= 0 || n <= 0 || m <= 0 || K <= 0 || c <= 0 || (cpad & 3) || cpad < 6 + c) return I2P_ERR_BAD_ARG;
    if (!xyz || !pix_xyz || !pts || !pix || !idx || !out) return I2P_ERR_BAD_ARG;
    const long long tot = (long long)n * K * (cpad >> 2);
    hipLaunchKernelGGL(knn_rows_fwd_kernel, dim3((unsigned)((tot + 255) / 256), b), dim3(256), 0, (hipStream_t)stream, n, m, K, c, cpad, xyz,
                       pix_xyz, pts, pix, idx, out);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_knn_rows_bwd(int b, int n, int m, int K, int c, int cpad, const float *g, const float *pts, const float *pix,
                                const int64_t *idx, float *d_xyz, float *d_pts, float *gq, void *stream) {
    if (b <= 0 || n <= 0 || m <= 0 || K <= 0 || c <= 0 || (cpad & 3) || cpad < 6 + c) return I2P_ERR_BAD_ARG;
    if (!g || !pts || !pix || !idx || !d_pts || !gq) return I2P_ERR_BAD_ARG;
    hipLaunchKernelGGL(knn_rows_bwd_kernel, dim3(n, b), dim3(128), 0, (hipStream_t)stream, n, m, K, c, cpad, g, pts, pix, idx, d_xyz, d_pts, gq);
    I2P_RETURN_LAUNCH_STATUS();
}
