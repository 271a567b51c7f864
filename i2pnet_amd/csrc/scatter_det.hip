// Deterministic scatter-add: the backward of every gather of the path (gather_torch utils.py:36-60 -> torch.gather's
// backward; GroupingOperation / GatherOperation / ThreeInterpolate backward, group_points_gpu.cu:8-25,
// sampling_gpu.cu:46-63, interpolate_gpu.cu:120-142) without floating-point atomics.
//
// The reference accumulates with atomicAdd in whatever order the hardware serialises them, so its gradients are not
// reproducible run to run; here every destination cell is OWNED by one wave, which scans the index list in ascending
// row order and adds the matching source rows in that order:
//     dst[cell] = (((dst[cell] + src[r1]) + src[r2]) + ...)      r1 < r2 < ... the rows that point at `cell`
// — bit for bit the result of the serial CPU loop of the oracle (oracle/i2p_oracle.c: *_grad_cpu), independent of
// scheduling.  The index lists of this network are short (<= 65 k rows per sample) and L2-resident, so the scan costs a
// few microseconds; a destination needs no zero-fill beyond what the caller's contract already gives (the existing
// content of `dst` is the start value of the sum, as with the reference's `+=`).
//
// One wave owns CPW consecutive destination cells; its accumulators live in LDS ([CPW][C], dynamically indexed by the
// matching cell), lane = channel (+64, +128, ...).  Per 64-row chunk of the index list: one coalesced index load, a
// ballot of the rows that fall into the wave's cells, then the matching rows in ascending order, four source rows
// in flight at a time.
#include "common.h"

namespace {

constexpr int CPW = 16;              // destination cells per wave
constexpr int WAVES = 4;             // waves per block
constexpr int MAXCU = 4;             // channels per lane: C <= 256

struct ScatterP {
    int ncell, c, q;                 // destination cells per sample, channels, source rows per sample
    // index of source row r: I64PAIR: h_idx[r]*W + w_idx[r]; else idx32[r]
    const int64_t *h_idx, *w_idx; int W;
    const int *idx32;
    // source element (row r, channel ch):  src[b*src_b + (r / src_div)*src_row + ch*src_ch]  (* weight[b*q + r] if weighted)
    const float *src; long long src_b; int src_row, src_ch, src_div;
    const float *weight;
    // destination element (cell, ch): dst[b*dst_b + cell*dst_row + ch*dst_ch]
    float *dst; long long dst_b; int dst_row, dst_ch;
};

template <bool I64PAIR, bool WEIGHTED>
__global__ __launch_bounds__(64 * WAVES) void scatter_det_kernel(ScatterP p) {
    extern __shared__ float acc_all[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int cell0 = (blockIdx.x * WAVES + wave) * CPW;
    if (cell0 >= p.ncell) return;                      // (no block-level barrier below: waves are independent)
    float *acc = acc_all + (size_t)wave * CPW * p.c;
    const int cu = (p.c + 63) >> 6;
    float *dst = p.dst + (size_t)b * p.dst_b;
    const float *src = p.src + (size_t)b * p.src_b;
    for (int i = 0; i < CPW; ++i)
        for (int u = 0; u < cu; ++u) {
            const int ch = lane + 64 * u;
            if (ch < p.c) acc[i * p.c + ch] = cell0 + i < p.ncell ? dst[(size_t)(cell0 + i) * p.dst_row + (size_t)ch * p.dst_ch] : 0.f;
        }
    const size_t ib = (size_t)b * p.q;
    for (int r0 = 0; r0 < p.q; r0 += 64) {
        const int r = r0 + lane;
        int rel = -1;
        if (r < p.q) {
            long long cell;
            if constexpr (I64PAIR) cell = p.h_idx[ib + r] * p.W + p.w_idx[ib + r];
            else cell = p.idx32[ib + r];
            const long long d = cell - cell0;
            rel = (d >= 0 && d < CPW) ? (int)d : -1;
        }
        unsigned long long mask = __ballot(rel >= 0);
        while (mask) {
            int li[4], ci[4], nb = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                li[t] = 0; ci[t] = 0;
                if (mask) {
                    li[t] = __builtin_ctzll(mask); mask &= mask - 1;
                    ci[t] = __builtin_amdgcn_readlane(rel, li[t]);
                    nb = t + 1;
                }
            }
            float v[4][MAXCU];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < nb) {
                    const int row = r0 + li[t];
                    float wgt = 1.f;
                    if constexpr (WEIGHTED) wgt = p.weight[ib + row];
                    const float *s = src + (size_t)(row / p.src_div) * p.src_row;
#pragma unroll
                    for (int u = 0; u < MAXCU; ++u) {
                        const int ch = lane + 64 * u;
                        v[t][u] = (u < cu && ch < p.c) ? s[(size_t)ch * p.src_ch] : 0.f;
                        if constexpr (WEIGHTED) v[t][u] = __fmul_rn(v[t][u], wgt);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < nb) {
#pragma unroll
                    for (int u = 0; u < MAXCU; ++u) {
                        const int ch = lane + 64 * u;
                        if (u < cu && ch < p.c) acc[ci[t] * p.c + ch] += v[t][u];
                    }
                }
            }
        }
    }
    for (int i = 0; i < CPW; ++i)
        for (int u = 0; u < cu; ++u) {
            const int ch = lane + 64 * u;
            if (ch < p.c && cell0 + i < p.ncell) dst[(size_t)(cell0 + i) * p.dst_row + (size_t)ch * p.dst_ch] = acc[i * p.c + ch];
        }
}

template <bool I64PAIR, bool WEIGHTED>
int launch(const ScatterP &p, int b, hipStream_t st) {
    if (p.c > 64 * MAXCU) return I2P_ERR_BAD_ARG;
    const size_t bytes = (size_t)WAVES * CPW * p.c * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(scatter_det_kernel<I64PAIR, WEIGHTED>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const unsigned gx = (unsigned)((p.ncell + WAVES * CPW - 1) / (WAVES * CPW));
    hipLaunchKernelGGL((scatter_det_kernel<I64PAIR, WEIGHTED>), dim3(gx, b), dim3(64 * WAVES), bytes, st, p);
    I2P_RETURN_LAUNCH_STATUS();
}

}  // namespace

// grad_feat[b, h*W+w, :] += grad_out[b, row, :]  (channel-last rows)
int i2p_det_gather_rows_grad(int b, int hw, int c, int q, int W, const float *grad_out, const int64_t *h_idx, const int64_t *w_idx,
                             float *grad_feat, void *stream) {
    ScatterP p{};
    p.ncell = hw; p.c = c; p.q = q; p.h_idx = h_idx; p.w_idx = w_idx; p.W = W;
    p.src = grad_out; p.src_b = (long long)q * c; p.src_row = c; p.src_ch = 1; p.src_div = 1;
    p.dst = grad_feat; p.dst_b = (long long)hw * c; p.dst_row = c; p.dst_ch = 1;
    if (c > 64 * MAXCU) {                                          // wide rows: channel slices of 256
        for (int c0 = 0; c0 < c; c0 += 64 * MAXCU) {
            ScatterP s = p; s.c = c - c0 < 64 * MAXCU ? c - c0 : 64 * MAXCU; s.src = grad_out + c0; s.dst = grad_feat + c0;
            const int rc = launch<true, false>(s, b, (hipStream_t)stream);
            if (rc) return rc;
        }
        return 0;
    }
    return launch<true, false>(p, b, (hipStream_t)stream);
}

// grad_points[b, ch, idx[b,j]] += grad_out[b, ch, j]  (channel-major, pointnet2 layout)
int i2p_det_gather_points_grad(int b, int c, int n, int m, const float *grad_out, const int *idx, float *grad_points, void *stream) {
    for (int c0 = 0; c0 < c; c0 += 64 * MAXCU) {
        ScatterP p{};
        p.ncell = n; p.c = c - c0 < 64 * MAXCU ? c - c0 : 64 * MAXCU; p.q = m; p.idx32 = idx;
        p.src = grad_out + (size_t)c0 * m; p.src_b = (long long)c * m; p.src_row = 1; p.src_ch = m; p.src_div = 1;
        p.dst = grad_points + (size_t)c0 * n; p.dst_b = (long long)c * n; p.dst_row = 1; p.dst_ch = n;
        const int rc = launch<false, false>(p, b, (hipStream_t)stream);
        if (rc) return rc;
    }
    return 0;
}

// grad_points[b, ch, idx[b,p,k]] += grad_out[b, ch, p] * weight[b,p,k], k = 0..2 in this order (interpolate_gpu.cu:139-141)
int i2p_det_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx, const float *weight,
                                   float *grad_points, void *stream) {
    for (int c0 = 0; c0 < c; c0 += 64 * MAXCU) {
        ScatterP p{};
        p.ncell = m; p.c = c - c0 < 64 * MAXCU ? c - c0 : 64 * MAXCU; p.q = 3 * n; p.idx32 = idx; p.weight = weight;
        p.src = grad_out + (size_t)c0 * n; p.src_b = (long long)c * n; p.src_row = 1; p.src_ch = n; p.src_div = 3;
        p.dst = grad_points + (size_t)c0 * m; p.dst_b = (long long)c * m; p.dst_row = 1; p.dst_ch = m;
        const int rc = launch<false, true>(p, b, (hipStream_t)stream);
        if (rc) return rc;
    }
    return 0;
}
