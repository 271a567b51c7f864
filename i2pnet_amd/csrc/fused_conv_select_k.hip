// fused_conv_select_k for gfx950 — replaces fused_conv_select_k_gpu / FusedConvSelectKLauncher
// (reference: src/projectPN/fused_conv_select/fused_conv_go.cu:11-264).
//
// The reference runs ONE 512-thread block per batch sample (<= B of 256 CUs busy), each thread
// owning a query with three 150-entry private arrays and an O(K*kt) serial selection sort.
// Here a query is owned by a 16-lane DPP row (4 queries per wave64, 16 per 256-thread block,
// grid = ceil(B*npoints/16) blocks):
//
//   A. each lane evaluates ceil(kt/16) window cells (consecutive lanes -> consecutive cells of
//      a window row -> contiguous 12-B reads) and packs them as 64-bit keys (d2 bits : position);
//   B. sorts its own keys in registers, parks the sorted run in LDS and keeps the head;
//   C. K extraction steps: DPP row all-reduce (min) over the 16 heads, the owning lane records
//      the winner and pops its next key from LDS.  K*(4 DPP + ~8) instructions per 4 queries.
//
// Bit-exactness.  The reference's result order under EQUAL distances depends on the swap
// history of its selection sort, which no parallel order reproduces.  Steps A-C
// give the unique answer whenever the K+1 smallest valid distances are distinct; equal
// distances are detected (duplicate pops / equal consecutive minima) and that query is redone
// by one lane running the reference's serial algorithm verbatim out of LDS (step F).  Queries
// with distance^2 >= 1e10 (where valid entries can collide with the 1e10 sentinel) always take
// step F.
#include "common.h"

namespace {

constexpr int GROUP = 16;                 // lanes per query
constexpr int QPB = 16;                   // queries per 256-thread block
constexpr unsigned SENT_BITS = 0x501502F9u;   // bits of 1e10f, the reference's "empty slot" distance
constexpr unsigned PAD_BITS = 0x7FFFFFFFu;    // padding keys (position >= kt), above everything

// out-code per selected slot: pos (8 bits) | stored << 8 | valid << 9
constexpr unsigned CODE_STORED = 0x100u, CODE_VALID = 0x200u;

struct FcskParams {
    int B, H, W, npoints, kH, kW, K, flag;
    float dist2;
    int stride_h, stride_w, small_h, small_w;
    const float *xyz1, *xyz2;
    const int *idx_n2, *random_hw;
    int64_t *ob, *oh, *ow;
    float *om;
    int force_serial;
};

struct Center {
    float x, y, z;
    int base_h, base_w;
    int b;
};

// One window cell: returns the reference's Dist[pos] bits and whether it was "stored"
// (go.cu:82-180).  tab = packed (dh << 16 | dw & 0xffff) for pos < kt.
__device__ __forceinline__ void eval_cell(const FcskParams &p, const Center &c, int tabv,
                                          unsigned &dbits, unsigned &stored) {
    dbits = SENT_BITS; stored = 0;
    int h = c.base_h + (tabv >> 16);
    int w = c.base_w + (int)(short)(tabv & 0xffff);
    if (h < 0 || h >= p.small_h) return;                                  // go.cu:99-103 / :115
    if (p.flag & I2P_FLAG_SHIFT) {                                        // go.cu:106-112
        if (w < 0) w = p.small_w + w;
        if (w >= p.small_w) w = w - p.small_w;
    } else if (w < 0 || w >= p.small_w) {
        return;
    }
    if (w < 0 || w >= p.small_w) return;      // (window wider than the image: the reference reads out of bounds; skipped like an empty cell)
    const float *q = p.xyz2 + (((size_t)c.b * p.small_h + h) * p.small_w + w) * 3;
    const float xq = q[0], yq = q[1], zq = q[2];
    const float d0 = i2p_sq3(xq, yq, zq);                                 // go.cu:141
    if (d0 <= 1e-10f) return;                                             // go.cu:143
    const float dq = fmaxf(i2p_sq3(c.x - xq, c.y - yq, c.z - zq), 1e-10f); // go.cu:154
    if (dq > p.dist2) return;                                             // go.cu:157
    dbits = i2p_f2u(dq); stored = 1;
}

__device__ __forceinline__ void cell_hw(const FcskParams &p, const Center &c, int tabv, int &h,
                                        int &w) {
    h = c.base_h + (tabv >> 16);
    w = c.base_w + (int)(short)(tabv & 0xffff);
    if (p.flag & I2P_FLAG_SHIFT) {
        if (w < 0) w = p.small_w + w;
        if (w >= p.small_w) w = w - p.small_w;
    }
}

// LDS accesses of one 16-lane query row are private to its wave (g = tid >> 4: four queries per wave): ordering them needs the
// compiler to keep program order (the hardware executes a wave's LDS operations in order), not a block barrier.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Round 6 (what the level-1 grouping kernel, csrc/sa_group.hip, learned in rounds 3-6, applied to the boundary operator): the parked
// runs as 32-bit distance + 16-bit code (6 instead of 8 bytes per candidate) and the per-query result codes sized by K in dynamic
// LDS — 24 KB -> 15.6 KB per block at level 1, so that the 1800 blocks of a batch-8 call are resident at once (8 per CU) instead of
// 1.2 rounds of 6; head and next key of a lane in registers (a pop's LDS read is off the critical path unless the same lane pops
// twice in a row); the K loop ends when every query of the wave is done; everything after the window table is wave-private, so
// the four block barriers became one.
template <int SLOTS>
__global__ __launch_bounds__(256) void fcsk_kernel(FcskParams p) {
    __shared__ int tab[SLOTS * GROUP];
    __shared__ unsigned lst_d[QPB][SLOTS * GROUP];
    __shared__ unsigned short lst_c[QPB][SLOTS * GROUP];
    extern __shared__ unsigned short outc_all[];           // [QPB][(K + 2 rounded up to 4)]
    const int ocp = (p.K + 2 + 3) & ~3;

    const int kt = p.kH * p.kW;
    const int tid = threadIdx.x;
    const int g = tid >> 4, l16 = tid & 15;
    unsigned short *outc_g = outc_all + g * ocp;
    const unsigned nblocks = gridDim.x;
    const unsigned lb = i2p_xcd_swizzle(blockIdx.x, nblocks);

    // window offset table (one integer division per entry per block instead of per candidate)
    for (int i = tid; i < SLOTS * GROUP; i += 256) {
        int v = 0;
        if (i < kt) {
            const int k = p.random_hw[i];                                 // go.cu:86
            const int dh = k / p.kW - p.kH / 2, dw = k % p.kW - p.kW / 2; // go.cu:89-92
            v = (dh << 16) | (dw & 0xffff);
        }
        tab[i] = v;
    }
    for (int i = l16; i < p.K; i += GROUP) outc_g[i] = 0;

    const long long q = (long long)lb * QPB + g;
    const bool in_range = q < (long long)p.B * p.npoints;
    Center c; c.b = 0; c.x = c.y = c.z = 0.f; c.base_h = c.base_w = 0;
    int n = 0;
    bool live = false;
    if (in_range) {
        c.b = (int)(q / p.npoints); n = (int)(q % p.npoints);
        const int ch = p.idx_n2[((size_t)c.b * p.npoints + n) * 2 + 0];   // go.cu:65-66
        const int cw = p.idx_n2[((size_t)c.b * p.npoints + n) * 2 + 1];
        const float *cp = p.xyz1 + (((size_t)c.b * p.H + ch) * p.W + cw) * 3;
        c.x = cp[0]; c.y = cp[1]; c.z = cp[2];
        const float dc = fmaxf(i2p_sq3(c.x, c.y, c.z), 1e-10f);           // go.cu:72
        live = !(dc <= 1e-10f);                                           // go.cu:74
        c.base_h = ch / p.stride_h; c.base_w = cw / p.stride_w;
    }
    __syncthreads();

    // ---- A: evaluate this lane's window cells --------------------------------------------
    // (measured and not kept, round 6: all candidate loads issued before the centre is looked at — idx_n2 -> {centre, candidates} as two
    //  dependent round trips instead of three: 16.5 -> 18.5 / 19.2 us on the live-centre layouts, 27 more registers and the loads of
    //  queries that turn out empty)
    unsigned long long key[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int pos = s * GROUP + l16;
        unsigned dbits = PAD_BITS, stored = 0;
        if (pos < kt) {
            dbits = SENT_BITS;
            if (live) eval_cell(p, c, tab[pos], dbits, stored);
        }
        key[s] = ((unsigned long long)dbits << 32) | (unsigned)(pos | (stored << 8));
    }

    // ---- B: sort own keys, park the run in LDS -------------------------------------------
#pragma unroll
    for (int i = 0; i < SLOTS - 1; ++i)
#pragma unroll
        for (int j = 0; j < SLOTS - 1 - i; ++j) {
            const unsigned long long a = key[j], b2 = key[j + 1];
            const bool sw = a > b2;
            key[j] = sw ? b2 : a; key[j + 1] = sw ? a : b2;
        }
    const bool wave_live = __any(live ? 1 : 0);            // a wave whose four centres are all empty selects nothing
    bool need_serial = false;
    if (wave_live) {
#pragma unroll
        for (int s = 1; s < SLOTS; ++s) { lst_d[g][s * GROUP + l16] = (unsigned)(key[s] >> 32); lst_c[g][s * GROUP + l16] = (unsigned short)key[s]; }
        wave_lds_sync();

        // ---- C: K extraction steps (ends as soon as every query of the wave is done) ---------
        unsigned head_hi = (unsigned)(key[0] >> 32), head_lo = (unsigned)key[0];
        unsigned nxt_hi = SLOTS > 1 ? (unsigned)(key[SLOTS > 1 ? 1 : 0] >> 32) : PAD_BITS, nxt_lo = SLOTS > 1 ? (unsigned)key[SLOTS > 1 ? 1 : 0] : 0u;
        int ptr = 2;
        unsigned pops = 0, steps = 0, prev = 0;
        bool tie = false, done = !live;
        for (int s = 0; s < p.K; ++s) {
            if (!__any(done ? 0 : 1)) break;
            const unsigned gmin = i2p_row16_min_u32(head_hi);
            if (!done) {
                if (gmin >= SENT_BITS) {
                    done = true;                        // only empty slots remain: nothing more to write
                } else {
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
        {   // boundary: the (K+1)-th smallest must differ from the K-th, every pop must be unique
            const unsigned gnext = i2p_row16_min_u32(head_hi);
            const unsigned total = i2p_row16_add_u32(pops);
            if (!done && steps > 0 && gnext == prev) tie = true;
            if (total != steps) tie = true;
        }
        need_serial = live && (tie || p.force_serial);

        // ---- F: exact serial redo for queries with equal distances (wave-uniform branch) --------
        if (__any(need_serial ? 1 : 0)) {
            wave_lds_sync();
            if (need_serial) {
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
                    const int pos = s * GROUP + l16;
                    unsigned dbits = SENT_BITS, stored = 0;
                    if (pos < kt) eval_cell(p, c, tab[pos], dbits, stored);
                    lst_d[g][pos] = dbits; lst_c[g][pos] = (unsigned short)(pos | (stored << 8));
                }
            }
            wave_lds_sync();
            if (need_serial && l16 == 0) {
                unsigned *ad = lst_d[g]; unsigned short *ac = lst_c[g];   // ad[t] = Dist[t] bits, ac[t] = pos | stored
                for (int s = 0; s < p.K; ++s) {                                 // go.cu:183-236
                    // positions >= kt are never searched (go.cu:188); slots s >= kt stay sentinel
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
                    } else {
                        outc_g[s] = 0;
                    }
                }
            }
        }
        wave_lds_sync();
    }

    // ---- output: coalesced along K ---------------------------------------------------------
    const bool fill = (p.flag & I2P_FLAG_FILL) != 0;
    if (!live && !(fill && in_range)) return;                              // go.cu:74-78: untouched
    const size_t obase = ((size_t)c.b * p.npoints + n) * p.K;
    if (!live) {                                                           // FILL: what torch.zeros would hold
        for (int s = l16; s < p.K; s += GROUP) {
            p.ob[obase + s] = 0; p.oh[obase + s] = 0; p.ow[obase + s] = 0; p.om[obase + s] = 0.0f;
        }
        return;
    }
    const unsigned copy_code = outc_g[0];
    for (int s = l16; s < p.K; s += GROUP) {
        unsigned code = outc_g[s];
        bool wr = (code & CODE_VALID) != 0;                                // go.cu:225
        if (!wr && (p.flag & I2P_FLAG_COPY)) { code = copy_code; wr = true; } // go.cu:211-222
        if (!wr) {
            if (fill) { p.ob[obase + s] = 0; p.oh[obase + s] = 0; p.ow[obase + s] = 0; p.om[obase + s] = 0.0f; }
            continue;
        }
        int h = 0, w = 0;                                                  // non-stored slot: idx_h = idx_w = 0
        if (code & CODE_STORED) cell_hw(p, c, tab[code & 0xff], h, w);
        p.ob[obase + s] = c.b; p.oh[obase + s] = h; p.ow[obase + s] = w; p.om[obase + s] = 1.0f;
    }
}

template <int SLOTS>
int launch(const FcskParams &p, hipStream_t st) {
    const long long nq = (long long)p.B * p.npoints;
    const unsigned grid = (unsigned)((nq + QPB - 1) / QPB);
    const size_t bytes = (size_t)QPB * ((p.K + 2 + 3) & ~3) * sizeof(unsigned short);
    hipLaunchKernelGGL(fcsk_kernel<SLOTS>, dim3(grid), dim3(256), bytes, st, p);
    I2P_RETURN_LAUNCH_STATUS();
}

}  // namespace

extern "C" int i2p_fused_conv_select_k(int batch_size, int H, int W, int npoints,
                                       int kernel_size_H, int kernel_size_W, int K, int flag,
                                       float distance, int stride_h, int stride_w,
                                       const float *xyz1, const float *xyz2, const int *idx_n2,
                                       const int *random_hw, int64_t *selected_b_idx,
                                       int64_t *selected_h_idx, int64_t *selected_w_idx,
                                       float *valid_idx, float *valid_in_dis_idx,
                                       float *selected_mask, int small_h, int small_w,
                                       void *stream) {
    (void)valid_idx; (void)valid_in_dis_idx;   // accepted, never written (as the reference)
    if (batch_size < 0 || npoints < 0 || K < 0 || kernel_size_H <= 0 || kernel_size_W <= 0 ||
        stride_h <= 0 || stride_w <= 0 || H <= 0 || W <= 0 || small_h <= 0 || small_w <= 0)
        return I2P_ERR_BAD_ARG;
    const int kt = kernel_size_H * kernel_size_W;
    if (kt > I2P_MAX_WINDOW) return I2P_ERR_WINDOW;
    if (K > I2P_MAX_WINDOW) return I2P_ERR_K;
    if ((long long)batch_size * npoints == 0 || K == 0) return 0;
    if (!xyz1 || !xyz2 || !idx_n2 || !random_hw || !selected_b_idx || !selected_h_idx ||
        !selected_w_idx || !selected_mask)
        return I2P_ERR_BAD_ARG;

    FcskParams p;
    p.B = batch_size; p.H = H; p.W = W; p.npoints = npoints; p.kH = kernel_size_H;
    p.kW = kernel_size_W; p.K = K; p.flag = flag; p.dist2 = distance * distance;  // go.cu:26
    p.stride_h = stride_h; p.stride_w = stride_w; p.small_h = small_h; p.small_w = small_w;
    p.xyz1 = xyz1; p.xyz2 = xyz2; p.idx_n2 = idx_n2; p.random_hw = random_hw;
    p.ob = selected_b_idx; p.oh = selected_h_idx; p.ow = selected_w_idx; p.om = selected_mask;
    // valid distances >= 1e10 would collide with the sentinel: take the verbatim serial path
    p.force_serial = !(p.dist2 < 1e10f);
    hipStream_t st = (hipStream_t)stream;
    if (kt <= 16) return launch<1>(p, st);
    if (kt <= 48) return launch<3>(p, st);
    if (kt <= 144) return launch<9>(p, st);
    return launch<10>(p, st);
}
