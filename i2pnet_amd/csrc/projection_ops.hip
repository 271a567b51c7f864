// Spherical projection, channel-last row gather (+grad) and brute-force kNN for gfx950.
// These are eager-PyTorch code in the reference (src/projectPN/utils.py:36-60, :111-187,
// :343-380); they sit between the two native extensions on the same hot path.
#include "common.h"
#include <limits.h>

namespace {

// ------------------------------------------------------------------------------------------
// project_seq (use_rank=False).  Three passes, all HBM/L2-bound and tiny:
//   1. cell_winner[:] = -1
//   2. per point: cell = f(xyz) ; atomicMax(cell_winner[cell], point index)   (deterministic:
//      highest index wins == torch-CPU index_put_ last-writer-wins, one winner for all images)
//   3. per cell: copy the winner's xyz / feature rows, or write zeros (no separate memset).
// ------------------------------------------------------------------------------------------
struct ProjConst {
    float az_res, vres, voff;
};

__device__ __forceinline__ long long trunc_to_i64(float v) {
    // torch-CPU `.long()` on x86: cvttss2si returns INT64_MIN for NaN / out of range
    if (!(v >= -9.2233720368547758e18f && v < 9.2233720368547758e18f)) return LLONG_MIN;
    return (long long)v;
}

__device__ __forceinline__ int project_cell(float x, float y, float z, int H, int W, ProjConst pc) {
    const float PIf = 3.14159274101257324f;                                // float(np.pi)
    const float r = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
    const float fcol = __fdiv_rn(__fsub_rn(PIf, atan2f(y, x)), pc.az_res); // utils.py:147
    const float beta = asinf(__fdiv_rn(z, r));                             // :150
    const float frow = __fadd_rn(__fdiv_rn(beta, pc.vres), pc.voff);       // :152
    long long icol = trunc_to_i64(fcol);
    const long long irow_t = trunc_to_i64(frow);
    long long irow = (long long)((unsigned long long)H - (unsigned long long)irow_t);
    irow = irow < 0 ? 0 : (irow > H - 1 ? H - 1 : irow);                   // :154
    icol = icol < 0 ? 0 : (icol > W - 1 ? W - 1 : icol);                   // :155
    return (int)irow * W + (int)icol;
}

__global__ void proj_init_kernel(int total, int *__restrict__ cell_winner) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) cell_winner[i] = -1;
}

__global__ void proj_assign_kernel(int n, int H, int W, ProjConst pc, const float *__restrict__ xyz,
                                   int *__restrict__ cell_winner) {
    const int bi = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float *pt = xyz + ((size_t)bi * n + p) * 3;
    const int cell = project_cell(pt[0], pt[1], pt[2], H, W, pc);
    atomicMax(cell_winner + (size_t)bi * H * W + cell, p);
}

struct ProjFeats {
    const float *src[5];
    float *dst[5];
    int dim[5];
    int off[6];     // prefix sums of dim
    int n_img;
};

// one thread per (cell, channel) over the concatenated channel list of all images
__global__ void proj_fill_kernel(int n, int hw, int ctot, ProjFeats f,
                                 const int *__restrict__ cell_winner) {
    const int bi = blockIdx.y;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)hw * ctot) return;
    const int cell = (int)(t / ctot), ch = (int)(t % ctot);
    int img = 0;
#pragma unroll
    for (int i = 1; i < 5; ++i) if (i < f.n_img && ch >= f.off[i]) img = i;
    const int lc = ch - f.off[img], d = f.dim[img];
    const int w = cell_winner[(size_t)bi * hw + cell];
    float v = 0.f;
    if (w >= 0) v = f.src[img][((size_t)bi * n + w) * d + lc];
    f.dst[img][((size_t)bi * hw + cell) * d + lc] = v;
}

// ------------------------------------------------------------------------------------------
// gather_rows (+grad): out[b,q,:] = feat[b, h*W+w, :]  — gather_torch, utils.py:36-60
// ------------------------------------------------------------------------------------------
__global__ void gather_rows_kernel(int hw, int c, int q, int W, const float *__restrict__ feat,
                                   const int64_t *__restrict__ h_idx,
                                   const int64_t *__restrict__ w_idx, float *__restrict__ out) {
    const int bi = blockIdx.y;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)q * c) return;
    const int row = (int)(t / c), ch = (int)(t % c);
    const long long cell = h_idx[(size_t)bi * q + row] * W + w_idx[(size_t)bi * q + row];
    out[((size_t)bi * q + row) * c + ch] = feat[((size_t)bi * hw + cell) * c + ch];
}

// Scatter-add backward.  Neighbour lists repeat the nearest hit in their empty slots
// (FLAG_COPY), so consecutive rows very often target the same cell: a thread walks RUN consecutive
// rows for one channel, sums equal-cell runs in a register and issues one atomic per run.
constexpr int GRAD_RUN = 8;
__global__ void gather_rows_grad_kernel(int hw, int c, int q, int W,
                                        const float *__restrict__ grad_out,
                                        const int64_t *__restrict__ h_idx,
                                        const int64_t *__restrict__ w_idx,
                                        float *__restrict__ grad_feat) {
    const int bi = blockIdx.y;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nrun = (q + GRAD_RUN - 1) / GRAD_RUN;
    if (t >= (long long)nrun * c) return;
    const int run = (int)(t / c), ch = (int)(t % c);
    const int r0 = run * GRAD_RUN, r1 = min(q, r0 + GRAD_RUN);
    long long cur = -1; float acc = 0.f;
    for (int row = r0; row < r1; ++row) {
        const long long cell = h_idx[(size_t)bi * q + row] * W + w_idx[(size_t)bi * q + row];
        const float gv = grad_out[((size_t)bi * q + row) * c + ch];
        if (cell != cur) {
            if (cur >= 0) atomicAdd(grad_feat + ((size_t)bi * hw + cur) * c + ch, acc);
            cur = cell; acc = 0.f;
        }
        acc += gv;
    }
    if (cur >= 0) atomicAdd(grad_feat + ((size_t)bi * hw + cur) * c + ch, acc);
}

// ------------------------------------------------------------------------------------------
// kNN — knn_point, utils.py:366-380.  One wave per query; k rounds, each extracting the
// smallest (distance, index) key greater than the previous one.  Distances are recomputed
// per round (6 flops) instead of stored; fine for the 228x468 / k=32 call of cost_volume2.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float knn_dist(float qx, float qy, float qz, float qq, const float *p) {
    const float px = p[0], py = p[1], pz = p[2];
    const float pp = __fadd_rn(__fadd_rn(__fmul_rn(px, px), __fmul_rn(py, py)), __fmul_rn(pz, pz));
    const float dot = __fmaf_rn(qz, pz, __fmaf_rn(qy, py, __fmul_rn(qx, px)));
    return __fadd_rn(__fadd_rn(__fmul_rn(-2.0f, dot), qq), pp);           // utils.py:362-364
}

// total order on (float d, int j) including negative d (cancellation can give d < 0)
__device__ __forceinline__ unsigned long long knn_key(float d, int j) {
    unsigned u = __float_as_uint(d);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned)j;
}

// Second-generation kNN: ONE scan of the cloud per query instead of k.  Every lane keeps the KT smallest keys of its
// own points in a sorted register list; the k winners are then popped by k wave-wide minimum reductions (DPP, no LDS).
// A lane whose list runs dry although it had to drop candidates re-scans its points above its last popped key
// (rare: k/64 winners per lane on average).  Same keys, same total order, same output as the first-generation kernel (k scans per query, removed in round 5) and as the oracle.
constexpr int KNN_KT = 4;

template <int CTRL>
__device__ __forceinline__ void knn_dpp_min(unsigned &hi, unsigned &lo) {
    const unsigned ohi = i2p_dpp_u32<CTRL>(hi), olo = i2p_dpp_u32<CTRL>(lo);
    const bool take = (ohi < hi) || (ohi == hi && olo < lo);
    hi = take ? ohi : hi; lo = take ? olo : lo;
}

__device__ __forceinline__ unsigned long long knn_wave_min(unsigned long long key) {
    unsigned hi = (unsigned)(key >> 32), lo = (unsigned)key;
    knn_dpp_min<0xB1>(hi, lo); knn_dpp_min<0x4E>(hi, lo); knn_dpp_min<0x141>(hi, lo); knn_dpp_min<0x140>(hi, lo);
    unsigned long long r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        r[q] = ((unsigned long long)__builtin_amdgcn_readlane(hi, 16 * q) << 32) | __builtin_amdgcn_readlane(lo, 16 * q);
    const unsigned long long a = r[0] < r[1] ? r[0] : r[1], b = r[2] < r[3] ? r[2] : r[3];
    return a < b ? a : b;
}

__global__ __launch_bounds__(256) void knn_kernel2(int n, int s, int k, const float *__restrict__ xyz,
                                                   const float *__restrict__ new_xyz, int *__restrict__ idx) {
    const int bi = blockIdx.y;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (qi >= s) return;
    const float *q = new_xyz + ((size_t)bi * s + qi) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    const float qq = __fadd_rn(__fadd_rn(__fmul_rn(qx, qx), __fmul_rn(qy, qy)), __fmul_rn(qz, qz));
    const float *src = xyz + (size_t)bi * n * 3;
    int *o = idx + ((size_t)bi * s + qi) * k;

    unsigned long long L[KNN_KT];
    bool dropped = false;
    unsigned long long mine = 0ull;                    // last key popped from THIS lane (valid once popped_any)
    bool popped_any = false;
    auto scan = [&]() {
#pragma unroll
        for (int i = 0; i < KNN_KT; ++i) L[i] = ~0ull;
        dropped = false;
        for (int j = lane; j < n; j += 64) {
            const float d = knn_dist(qx, qy, qz, qq, src + (size_t)j * 3);
            if (d != d) continue;                       // NaN never selected (as `<` in the oracle)
            unsigned long long key = knn_key(d, j);
            if (popped_any && key <= mine) continue;    // already delivered
            if (key >= L[KNN_KT - 1]) { dropped = true; continue; }
            if (L[KNN_KT - 1] != ~0ull) dropped = true; // the current tail falls off
#pragma unroll
            for (int i = 0; i < KNN_KT; ++i) {          // sorted insertion by a bubble of swaps
                const unsigned long long cur = L[i];
                const bool sw = key < cur;
                L[i] = sw ? key : cur; key = sw ? cur : key;
            }
        }
    };
    scan();
    for (int t = 0; t < k; ++t) {
        const unsigned long long best = knn_wave_min(L[0]);
        if (lane == 0) o[t] = (int)(unsigned)(best & 0xffffffffull);
        if (L[0] == best && best != ~0ull) {            // the owning lane pops its head (keys are unique)
            mine = best; popped_any = true;
#pragma unroll
            for (int i = 0; i + 1 < KNN_KT; ++i) L[i] = L[i + 1];
            L[KNN_KT - 1] = ~0ull;
        }
        const bool need = (L[0] == ~0ull) && dropped;
        if (__ballot(need) != 0ull) {
            if (need) scan();
        }
    }
}

}  // namespace

extern "C" int i2p_project_seq(int b, int n, int H, int W, float fup_deg, float fdown_deg,
                               const float *xyz, int nfeat, const float *const *feats,
                               const int *feat_dims, float *out_xyz, float *const *out_feats,
                               int *cell_winner, void *stream) {
    if (b < 0 || n < 0 || H <= 1 || W <= 0 || nfeat < 0 || nfeat > 4) return I2P_ERR_BAD_ARG;
    if (b == 0) return 0;
    if (!out_xyz || !cell_winner || (n > 0 && !xyz) || (nfeat > 0 && (!feats || !feat_dims || !out_feats)))
        return I2P_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    // constants exactly as the oracle derives them (double math, rounded to fp32 once)
    const double deg2rad = 3.14159265358979323846 / 180.0;
    const double vdown = (double)fdown_deg * deg2rad, vup = (double)fup_deg * deg2rad;
    const double vres_d = (vup - vdown) / (H - 1);
    ProjConst pc;
    pc.az_res = (float)((360.0 / W) * deg2rad);
    pc.vres = (float)vres_d;
    pc.voff = (float)(-vdown / vres_d);

    ProjFeats f;
    f.n_img = nfeat + 1;
    f.src[0] = xyz; f.dst[0] = out_xyz; f.dim[0] = 3; f.off[0] = 0;
    for (int i = 0; i < nfeat; ++i) {
        if (feat_dims[i] <= 0 || !feats[i] || !out_feats[i]) return I2P_ERR_BAD_ARG;
        f.src[i + 1] = feats[i]; f.dst[i + 1] = out_feats[i]; f.dim[i + 1] = feat_dims[i];
    }
    for (int i = nfeat + 1; i < 5; ++i) { f.src[i] = nullptr; f.dst[i] = nullptr; f.dim[i] = 0; }
    for (int i = 0; i < 5; ++i) f.off[i + 1] = f.off[i] + f.dim[i];
    const int ctot = f.off[5];
    const int hw = H * W;

    hipLaunchKernelGGL(proj_init_kernel, dim3((b * hw + 255) / 256), dim3(256), 0, st, b * hw, cell_winner);
    if (n > 0)
        hipLaunchKernelGGL(proj_assign_kernel, dim3((n + 255) / 256, b), dim3(256), 0, st, n, H, W, pc, xyz,
                           cell_winner);
    const long long tot = (long long)hw * ctot;
    hipLaunchKernelGGL(proj_fill_kernel, dim3((unsigned)((tot + 255) / 256), b), dim3(256), 0, st, n, hw, ctot,
                       f, cell_winner);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_gather_rows(int b, int hw, int c, int q, int W, const float *feat,
                               const int64_t *h_idx, const int64_t *w_idx, float *out,
                               void *stream) {
    if (b < 0 || hw < 0 || c < 0 || q < 0 || W <= 0) return I2P_ERR_BAD_ARG;
    if ((long long)b * q * c == 0) return 0;
    if (!feat || !h_idx || !w_idx || !out) return I2P_ERR_BAD_ARG;
    const long long tot = (long long)q * c;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((tot + 255) / 256), b), dim3(256), 0,
                       (hipStream_t)stream, hw, c, q, W, feat, h_idx, w_idx, out);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_gather_rows_grad(int b, int hw, int c, int q, int W, const float *grad_out,
                                    const int64_t *h_idx, const int64_t *w_idx,
                                    float *grad_feat, void *stream) {
    if (b < 0 || hw < 0 || c < 0 || q < 0 || W <= 0) return I2P_ERR_BAD_ARG;
    if ((long long)b * q * c == 0) return 0;
    if (!grad_out || !h_idx || !w_idx || !grad_feat) return I2P_ERR_BAD_ARG;
    if (!i2p_atomic_scatter()) return i2p_det_gather_rows_grad(b, hw, c, q, W, grad_out, h_idx, w_idx, grad_feat, stream);
    const long long tot = (long long)((q + GRAD_RUN - 1) / GRAD_RUN) * c;
    hipLaunchKernelGGL(gather_rows_grad_kernel, dim3((unsigned)((tot + 255) / 256), b), dim3(256), 0,
                       (hipStream_t)stream, hw, c, q, W, grad_out, h_idx, w_idx, grad_feat);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_knn(int b, int n, int s, int k, const float *xyz, const float *new_xyz,
                       int *idx, void *stream) {
    if (b < 0 || n < 0 || s < 0 || k < 0 || k > n) return I2P_ERR_BAD_ARG;
    if ((long long)b * s == 0 || k == 0) return 0;
    if (!xyz || !new_xyz || !idx) return I2P_ERR_BAD_ARG;
    hipLaunchKernelGGL(knn_kernel2, dim3((s + 3) / 4, b), dim3(256), 0, (hipStream_t)stream, n, s, k, xyz, new_xyz, idx);
    I2P_RETURN_LAUNCH_STATUS();
}


/* -------------------------------------------------------------------------------------------
 * Hamilton product with broadcasting over the point axis — src/modules/warp_utils.py:25-55
 * (the reference evaluates it with 16 elementwise multiplies, 12 adds and a stack; its autograd
 * backward is ~80 tiny launches per product).  One thread per output quaternion; the expression
 * order of the reference is kept term by term (compiled with -ffp-contract=off).
 * conj_a / conj_b: use the conjugate of that operand (what the backward needs:
 * d/da = g (x) conj(b), d/db = conj(a) (x) g).
 * ------------------------------------------------------------------------------------------- */
__global__ void quat_mul_kernel(int total, int n, int na, int nb, int conj_a, int conj_b,
                                const float4 *__restrict__ a, const float4 *__restrict__ b,
                                float4 *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int bi = i / n, pi = i - bi * n;
    float4 qa = a[(size_t)bi * na + (na == 1 ? 0 : pi)];
    float4 qb = b[(size_t)bi * nb + (nb == 1 ? 0 : pi)];
    if (conj_a) { qa.y = -qa.y; qa.z = -qa.z; qa.w = -qa.w; }
    if (conj_b) { qb.y = -qb.y; qb.z = -qb.z; qb.w = -qb.w; }
    const float aw = qa.x, ax = qa.y, ay = qa.z, az = qa.w;
    const float bw = qb.x, bx = qb.y, by = qb.z, bz = qb.w;
    float4 o;
    o.x = ((aw * bw - ax * bx) - ay * by) - az * bz;
    o.y = ((aw * bx + ax * bw) + ay * bz) - az * by;
    o.z = ((aw * by - ax * bz) + ay * bw) + az * bx;
    o.w = ((aw * bz + ax * by) - ay * bx) + az * bw;
    out[i] = o;
}

// per-quaternion inverse / normalisation and their gradients (include/i2p_ops.h: i2p_quat_unit_*)
__global__ void quat_unit_fwd_kernel(int mode, long long rows, const float4 *__restrict__ q, float4 *__restrict__ out) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float4 v = q[r];
    const float n2 = (v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w) + 1e-10f;
    if (mode == 0) {
        out[r] = make_float4(v.x / n2, -v.y / n2, -v.z / n2, -v.w / n2);
    } else {
        const float d = sqrtf(n2) + 1e-10f;
        out[r] = make_float4(v.x / d, v.y / d, v.z / d, v.w / d);
    }
}

__global__ void quat_unit_bwd_kernel(int mode, long long rows, const float4 *__restrict__ q, const float4 *__restrict__ g,
                                     float4 *__restrict__ dq) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float4 v = q[r], go = g[r];
    const float n2 = (v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w) + 1e-10f;
    if (mode == 0) {
        // out_j = s_j q_j / n2  ->  dq = (s o g - 2 q <g, out>) / n2
        const float dot = (go.x * v.x - go.y * v.y - go.z * v.z - go.w * v.w) / n2;
        dq[r] = make_float4((go.x - 2.f * v.x * dot) / n2, (-go.y - 2.f * v.y * dot) / n2,
                            (-go.z - 2.f * v.z * dot) / n2, (-go.w - 2.f * v.w * dot) / n2);
    } else {
        // out = q / d, d = r + 1e-10, r = sqrt(n2)  ->  dq = g / d - q <g, q> / (d^2 r)
        const float rr = sqrtf(n2), d = rr + 1e-10f;
        const float k = (go.x * v.x + go.y * v.y + go.z * v.z + go.w * v.w) / (d * d * rr);
        dq[r] = make_float4(go.x / d - v.x * k, go.y / d - v.y * k, go.z / d - v.z * k, go.w / d - v.w * k);
    }
}

extern "C" int i2p_quat_unit_fwd(int mode, long long rows, const float *q, float *out, void *stream) {
    if (rows < 0 || (mode != 0 && mode != 1)) return I2P_ERR_BAD_ARG;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(quat_unit_fwd_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, (hipStream_t)stream, mode, rows,
                       (const float4 *)q, (float4 *)out);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_quat_unit_bwd(int mode, long long rows, const float *q, const float *g, float *dq, void *stream) {
    if (rows < 0 || (mode != 0 && mode != 1)) return I2P_ERR_BAD_ARG;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(quat_unit_bwd_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, (hipStream_t)stream, mode, rows,
                       (const float4 *)q, (const float4 *)g, (float4 *)dq);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_quat_mul(int b, int na, int nb, int conj_a, int conj_b, const float *qa, const float *qb,
                            float *out, void *stream) {
    const int n = na > nb ? na : nb;
    if (b < 0 || na < 1 || nb < 1 || (na != n && na != 1) || (nb != n && nb != 1)) return I2P_ERR_BAD_ARG;
    const long long total = (long long)b * n;
    if (total == 0) return 0;
    if (total > 0x7fffffffLL) return I2P_ERR_BAD_ARG;
    hipLaunchKernelGGL(quat_mul_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (int)total, n, na, nb, conj_a, conj_b, (const float4 *)qa, (const float4 *)qb, (float4 *)out);
    I2P_RETURN_LAUNCH_STATUS();
}


/* -------------------------------------------------------------------------------------------
 * Warp of the level-3 cloud by the coarse pose + masking of empty cells + depth split, one launch each way
 * (reference: warp_utils.py:78-94 warp_quat_xyz = cat, two Hamilton products, the quaternion inverse, an add, a slice;
 * modellearn_proj_center.py:345-352 `* valid`, `z = P[:, :, 2:]`, `uv = P / (z + 1e-10)`; PPBackbone_center.py:377 `xyz = uv * z`:
 * ~10 launches forward and ~25 backward on [B,228,3] tensors).
 *   p' = (q (x) [0,p] (x) r + t)[1:4] * valid,  r = conj(q) / (|q|^2 + 1e-10);  z = p'_z;  uv = p' / (z + 1e-10);  xyz = uv * z
 * Same fp32 operation order as quat_mul_kernel / quat_unit_fwd_kernel, so the values are those of the unfused chain.
 * Backward (p and valid are data): dq [B,4], dt [B,4] (w component 0) from g_uv, g_z, g_xyz; one block per sample, the
 * per-point contributions summed in a fixed tree order.
 * ------------------------------------------------------------------------------------------- */
namespace {
__device__ __forceinline__ float4 qmul(const float4 &a, const float4 &b) {
    const float aw = a.x, ax = a.y, ay = a.z, az = a.w, bw = b.x, bx = b.y, by = b.z, bz = b.w;
    float4 o;
    o.x = ((aw * bw - ax * bx) - ay * by) - az * bz;
    o.y = ((aw * bx + ax * bw) + ay * bz) - az * by;
    o.z = ((aw * by - ax * bz) + ay * bw) + az * bx;
    o.w = ((aw * bz + ax * by) - ay * bx) + az * bw;
    return o;
}
__device__ __forceinline__ float4 qconj(const float4 &a) { return make_float4(a.x, -a.y, -a.z, -a.w); }
__device__ __forceinline__ float4 qinv(const float4 &v) {
    const float n2 = (v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w) + 1e-10f;
    return make_float4(v.x / n2, -v.y / n2, -v.z / n2, -v.w / n2);
}

__global__ __launch_bounds__(256) void warp_split_fwd_kernel(int N, const float *__restrict__ p, const float4 *__restrict__ q,
                                                             const float4 *__restrict__ t, const float *__restrict__ valid,
                                                             float *__restrict__ uv, float *__restrict__ z, float *__restrict__ xyz) {
    const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const size_t i = (size_t)b * N + n;
    const float4 qq = q[b], r = qinv(qq), tt = t[b];
    const float4 P = make_float4(0.f, p[i * 3], p[i * 3 + 1], p[i * 3 + 2]);
    const float4 h = qmul(qmul(qq, P), r);
    const float v = valid ? valid[i] : 1.f;
    const float px = (h.y + tt.y) * v, py = (h.z + tt.z) * v, pz = (h.w + tt.w) * v;
    const float d = pz + 1e-10f;
    const float ux = px / d, uy = py / d, uz = pz / d;
    uv[i * 3] = ux; uv[i * 3 + 1] = uy; uv[i * 3 + 2] = uz;
    z[i] = pz;
    xyz[i * 3] = ux * pz; xyz[i * 3 + 1] = uy * pz; xyz[i * 3 + 2] = uz * pz;
}

__global__ __launch_bounds__(256) void warp_split_bwd_kernel(int N, const float *__restrict__ p, const float4 *__restrict__ q,
                                                             const float4 *__restrict__ t, const float *__restrict__ valid,
                                                             const float *__restrict__ g_uv, const float *__restrict__ g_z,
                                                             const float *__restrict__ g_xyz, float4 *__restrict__ dq, float4 *__restrict__ dt) {
    __shared__ float red[256][12];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float4 qq = q[b], r = qinv(qq), tt = t[b];
    float acc[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // dq_direct (4), dr (4), dt (4)
    for (int n = tid; n < N; n += 256) {
        const size_t i = (size_t)b * N + n;
        const float4 P = make_float4(0.f, p[i * 3], p[i * 3 + 1], p[i * 3 + 2]);
        const float4 A = qmul(qq, P), h = qmul(A, r);
        const float v = valid ? valid[i] : 1.f;
        const float pp[3] = {(h.y + tt.y) * v, (h.z + tt.z) * v, (h.w + tt.w) * v};
        const float zz = pp[2], d = zz + 1e-10f;
        // p' -> (uv = p'/d, z = p'_z, xyz = uv * z):  d xyz_i / d p'_i = z/d, d xyz_i / d p'_z += p'_i * 1e-10 / d^2
        float gp[3] = {0.f, 0.f, 0.f};
        float gz = g_z ? g_z[i] : 0.f;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const float gu = (g_uv ? g_uv[i * 3 + e] : 0.f) + (g_xyz ? g_xyz[i * 3 + e] * zz : 0.f);   // gradient on uv_e (direct + through xyz = uv*z)
            gp[e] += gu / d;
            gz += (g_xyz ? g_xyz[i * 3 + e] * (pp[e] / d) : 0.f) - gu * pp[e] / (d * d);
        }
        gp[2] += gz;
        const float4 G = make_float4(0.f, gp[0] * v, gp[1] * v, gp[2] * v);                            // through `* valid`; the w component was sliced away
        // h = A (x) r:  dA = G (x) conj(r),  dr = conj(A) (x) G;   A = q (x) P:  dq = dA (x) conj(P)
        const float4 dA = qmul(G, qconj(r)), dr = qmul(qconj(A), G), dqd = qmul(dA, qconj(P));
        acc[0] += dqd.x; acc[1] += dqd.y; acc[2] += dqd.z; acc[3] += dqd.w;
        acc[4] += dr.x; acc[5] += dr.y; acc[6] += dr.z; acc[7] += dr.w;
        acc[9] += G.y; acc[10] += G.z; acc[11] += G.w;
    }
#pragma unroll
    for (int e = 0; e < 12; ++e) red[tid][e] = acc[e];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s)
#pragma unroll
            for (int e = 0; e < 12; ++e) red[tid][e] += red[tid + s][e];
        __syncthreads();
    }
    if (tid == 0) {
        // r = conj(q) / n2 (quat_unit_bwd_kernel, mode 0):  dq += (s o dr - 2 q <dr, r>) / n2
        const float n2 = (qq.x * qq.x + qq.y * qq.y + qq.z * qq.z + qq.w * qq.w) + 1e-10f;
        const float gx = red[0][4], gy = red[0][5], gzz = red[0][6], gw = red[0][7];
        const float dot = (gx * qq.x - gy * qq.y - gzz * qq.z - gw * qq.w) / n2;
        dq[b] = make_float4(red[0][0] + (gx - 2.f * qq.x * dot) / n2, red[0][1] + (-gy - 2.f * qq.y * dot) / n2,
                            red[0][2] + (-gzz - 2.f * qq.z * dot) / n2, red[0][3] + (-gw - 2.f * qq.w * dot) / n2);
        dt[b] = make_float4(0.f, red[0][9], red[0][10], red[0][11]);
    }
}
}  // namespace

extern "C" int i2p_warp_split_fwd(int B, int N, const float *p, const float *q, const float *t, const float *valid, float *uv, float *z,
                                  float *xyz, void *stream) {
    if (B <= 0 || N <= 0 || !p || !q || !t || !uv || !z || !xyz) return I2P_ERR_BAD_ARG;
    hipLaunchKernelGGL(warp_split_fwd_kernel, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, N, p, (const float4 *)q, (const float4 *)t,
                       valid, uv, z, xyz);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_warp_split_bwd(int B, int N, const float *p, const float *q, const float *t, const float *valid, const float *g_uv,
                                  const float *g_z, const float *g_xyz, float *dq, float *dt, void *stream) {
    if (B <= 0 || N <= 0 || !p || !q || !t || !dq || !dt) return I2P_ERR_BAD_ARG;
    hipLaunchKernelGGL(warp_split_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, N, p, (const float4 *)q, (const float4 *)t, valid, g_uv,
                       g_z, g_xyz, (float4 *)dq, (float4 *)dt);
    I2P_RETURN_LAUNCH_STATUS();
}

/* -------------------------------------------------------------------------------------------
 * Composition of the fine pose with the coarse one, one launch each way (reference: modellearn_proj_center.py:388-404,
 * q = q3 (x) q_prev, t = (q3 (x) [0, t_prev] (x) q3^-1)[1:4] + t3 — as torch / the unfused device chain: three Hamilton products, the
 * quaternion inverse, two cats, an add forward and ~14 launches backward on [B,4] tensors).
 *   out f32 [B,7] = [q (4), t (3)];  same fp32 operation order as quat_mul_kernel / quat_unit_fwd_kernel.
 * tp / dtp are quaternion-shaped [B,4] = [0, t_prev] as the model carries the coarse translation (the w component of tp is read as 0).
 * Backward from g [B,7]: dq3 [B,4], dt3 [B,3], dqp [B,4], dtp [B,4].
 * ------------------------------------------------------------------------------------------- */
namespace {
__global__ void pose_compose_fwd_kernel(int B, const float4 *__restrict__ q3, const float *__restrict__ t3, const float4 *__restrict__ qp,
                                        const float4 *__restrict__ tp, float *__restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float4 q = q3[b], r = qinv(q);
    const float4 oq = qmul(q, qp[b]);
    const float4 T = make_float4(0.f, tp[b].y, tp[b].z, tp[b].w);
    const float4 h = qmul(qmul(q, T), r);
    float *o = out + (size_t)b * 7;
    o[0] = oq.x; o[1] = oq.y; o[2] = oq.z; o[3] = oq.w;
    o[4] = h.y + t3[b * 3]; o[5] = h.z + t3[b * 3 + 1]; o[6] = h.w + t3[b * 3 + 2];
}

__global__ void pose_compose_bwd_kernel(int B, const float4 *__restrict__ q3, const float4 *__restrict__ qp, const float4 *__restrict__ tp,
                                        const float *__restrict__ g, float4 *__restrict__ dq3, float *__restrict__ dt3,
                                        float4 *__restrict__ dqp, float4 *__restrict__ dtp) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float4 q = q3[b], r = qinv(q), p = qp[b];
    const float *gb = g + (size_t)b * 7;
    const float4 gq = make_float4(gb[0], gb[1], gb[2], gb[3]);
    const float4 G = make_float4(0.f, gb[4], gb[5], gb[6]);             // the w component of the rotated translation was sliced away
    const float4 T = make_float4(0.f, tp[b].y, tp[b].z, tp[b].w);
    // out_q = q (x) p:  dq = gq (x) conj(p),  dp = conj(q) (x) gq
    const float4 dq_a = qmul(gq, qconj(p));
    dqp[b] = qmul(qconj(q), gq);
    // h = A (x) r, A = q (x) T:  dA = G (x) conj(r),  dr = conj(A) (x) G,  dq += dA (x) conj(T),  dT = conj(q) (x) dA
    const float4 A = qmul(q, T);
    const float4 dA = qmul(G, qconj(r)), dr = qmul(qconj(A), G), dq_b = qmul(dA, qconj(T)), dT = qmul(qconj(q), dA);
    // r = conj(q) / n2 (quat_unit_bwd_kernel, mode 0):  dq += (s o dr - 2 q <dr, r>) / n2
    const float n2 = (q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w) + 1e-10f;
    const float dot = (dr.x * q.x - dr.y * q.y - dr.z * q.z - dr.w * q.w) / n2;
    dq3[b] = make_float4(dq_a.x + dq_b.x + (dr.x - 2.f * q.x * dot) / n2, dq_a.y + dq_b.y + (-dr.y - 2.f * q.y * dot) / n2,
                         dq_a.z + dq_b.z + (-dr.z - 2.f * q.z * dot) / n2, dq_a.w + dq_b.w + (-dr.w - 2.f * q.w * dot) / n2);
    dtp[b] = dT;
    dt3[b * 3] = gb[4]; dt3[b * 3 + 1] = gb[5]; dt3[b * 3 + 2] = gb[6];
}
}  // namespace

extern "C" int i2p_pose_compose_fwd(int B, const float *q3, const float *t3, const float *qp, const float *tp, float *out, void *stream) {
    if (B <= 0 || !q3 || !t3 || !qp || !tp || !out) return I2P_ERR_BAD_ARG;
    hipLaunchKernelGGL(pose_compose_fwd_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, B, (const float4 *)q3, t3, (const float4 *)qp,
                       (const float4 *)tp, out);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_pose_compose_bwd(int B, const float *q3, const float *qp, const float *tp, const float *g, float *dq3, float *dt3,
                                    float *dqp, float *dtp, void *stream) {
    if (B <= 0 || !q3 || !qp || !tp || !g || !dq3 || !dt3 || !dqp || !dtp) return I2P_ERR_BAD_ARG;
    hipLaunchKernelGGL(pose_compose_bwd_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, B, (const float4 *)q3, (const float4 *)qp, (const float4 *)tp, g,
                       (float4 *)dq3, dt3, (float4 *)dqp, (float4 *)dtp);
    I2P_RETURN_LAUNCH_STATUS();
}

/* -------------------------------------------------------------------------------------------
 * Row-wise "unit variance" of the cost-volume inputs — src/projectPN/PPBackbone_center.py:388-393:
 *     y = (x - mean_c(x)) / clip(std_c(x) (unbiased), min=1e-12)
 * (torch: mean, sub, std, clip, div forward and ~14 autograd launches backward; here one launch each way).
 * One wave per row, c <= 256 (4 values per lane), two-pass variance.
 * stat f32 [rows,2] = {1/d, s > 1e-12 ? 1 : 0} for the backward:
 *     gx = (gy - mean(gy) - [s>1e-12] * y * sum(gy*y)/(c-1)) / d
 * ------------------------------------------------------------------------------------------- */
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ void row_unitvar_fwd_kernel(int rows, int c, const float *__restrict__ x, float *__restrict__ y,
                                       float *__restrict__ stat) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *xr = x + (size_t)row * c;
    float v[4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = lane + 64 * j;
        v[j] = k < c ? xr[k] : 0.f;
        s += v[j];
    }
    const float mean = wave_sum(s) / (float)c;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = lane + 64 * j;
        v[j] = k < c ? v[j] - mean : 0.f;
        q += v[j] * v[j];
    }
    const float sd = sqrtf(wave_sum(q) / (float)(c - 1));
    const float d = fmaxf(sd, 1e-12f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = lane + 64 * j;
        if (k < c) y[(size_t)row * c + k] = v[j] / d;
    }
    if (lane == 0) { stat[2 * row] = 1.0f / d; stat[2 * row + 1] = sd > 1e-12f ? 1.f : 0.f; }
}

__global__ void row_unitvar_bwd_kernel(int rows, int c, const float *__restrict__ gy, const float *__restrict__ y,
                                       const float *__restrict__ stat, float *__restrict__ gx) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float g[4], yy[4];
    float sg = 0.f, sgy = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = lane + 64 * j;
        g[j] = k < c ? gy[(size_t)row * c + k] : 0.f;
        yy[j] = k < c ? y[(size_t)row * c + k] : 0.f;
        sg += g[j]; sgy += g[j] * yy[j];
    }
    const float mg = wave_sum(sg) / (float)c;
    const float proj = stat[2 * row + 1] * wave_sum(sgy) / (float)(c - 1);
    const float inv_d = stat[2 * row];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = lane + 64 * j;
        if (k < c) gx[(size_t)row * c + k] = (g[j] - mg - yy[j] * proj) * inv_d;
    }
}

extern "C" int i2p_row_unitvar_fwd(int rows, int c, const float *x, float *y, float *stat, void *stream) {
    if (rows < 0 || c < 2 || c > 256) return I2P_ERR_BAD_ARG;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(row_unitvar_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, rows, c, x, y, stat);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_row_unitvar_bwd(int rows, int c, const float *gy, const float *y, const float *stat, float *gx,
                                   void *stream) {
    if (rows < 0 || c < 2 || c > 256) return I2P_ERR_BAD_ARG;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(row_unitvar_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, rows, c, gy, y, stat, gx);
    I2P_RETURN_LAUNCH_STATUS();
}

/* -------------------------------------------------------------------------------------------
 * Learned-uncertainty pose loss and its gradient in one launch — compute_loss.py:102-133 (`Get_loss`):
 *   per level L (coarse = out4, weight 1.6; fine = out3, weight 0.8):
 *     lq = mean_b sqrt(sum_4 (q_gt - q)^2 + 1e-10),  lx = mean |t - t_gt| (l1) or mean_b sqrt(sum_3 (t - t_gt)^2 + 1e-10)
 *     l  = lx*exp(-w_x) + w_x + lq*exp(-w_q) + w_q
 *   loss3 = {1.6 l_c + 0.8 l_f, 1.6 lq_c + 0.8 lq_f, 1.6 lx_c + 0.8 lx_f}
 * (eager PyTorch: 28 launches forward, 36 backward, all on [B,7] tensors).  One block; B <= 1024.
 * ------------------------------------------------------------------------------------------- */
__global__ __launch_bounds__(64) void pose_loss_kernel(int B, int l1, const float *__restrict__ out3,
                                                        const float *__restrict__ out4, const float *__restrict__ qgt,
                                                        const float *__restrict__ tgt, const float *__restrict__ wx,
                                                        const float *__restrict__ wq, float *__restrict__ loss3,
                                                        float *__restrict__ d_out3, float *__restrict__ d_out4,
                                                        float *__restrict__ d_w) {
    const int lane = threadIdx.x;
    const float ex = expf(-wx[0]), eq = expf(-wq[0]);
    float lq[2] = {0.f, 0.f}, lx[2] = {0.f, 0.f};               // 0 = coarse (out4), 1 = fine (out3)
    for (int b = lane; b < B; b += 64) {
#pragma unroll
        for (int L = 0; L < 2; ++L) {
            const float *o = (L == 0 ? out4 : out3) + (size_t)b * 7;
            float *d = (L == 0 ? d_out4 : d_out3) + (size_t)b * 7;
            const float wL = L == 0 ? 1.6f : 0.8f;
            float dq[4], s = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { dq[i] = qgt[b * 4 + i] - o[i]; s += dq[i] * dq[i]; }
            const float nq = sqrtf(s + 1e-10f);
            lq[L] += nq;
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = wL * eq * (-dq[i] / nq) / (float)B;
            float dt[3], st = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) { dt[i] = o[4 + i] - tgt[b * 3 + i]; st += dt[i] * dt[i]; }
            if (l1) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    lx[L] += fabsf(dt[i]);
                    d[4 + i] = wL * ex * (dt[i] > 0.f ? 1.f : (dt[i] < 0.f ? -1.f : 0.f)) / (float)(3 * B);
                }
            } else {
                const float nt = sqrtf(st + 1e-10f);
                lx[L] += nt;
#pragma unroll
                for (int i = 0; i < 3; ++i) d[4 + i] = wL * ex * (dt[i] / nt) / (float)B;
            }
        }
    }
#pragma unroll
    for (int L = 0; L < 2; ++L) { lq[L] = wave_sum(lq[L]) / (float)B; lx[L] = wave_sum(lx[L]) / (float)(l1 ? 3 * B : B); }
    if (lane == 0) {
        const float lc = lx[0] * ex + wx[0] + lq[0] * eq + wq[0], lf = lx[1] * ex + wx[0] + lq[1] * eq + wq[0];
        loss3[0] = 1.6f * lc + 0.8f * lf;
        loss3[1] = 1.6f * lq[0] + 0.8f * lq[1];
        loss3[2] = 1.6f * lx[0] + 0.8f * lx[1];
        d_w[0] = 1.6f * (1.f - lx[0] * ex) + 0.8f * (1.f - lx[1] * ex);       // d loss / d w_x
        d_w[1] = 1.6f * (1.f - lq[0] * eq) + 0.8f * (1.f - lq[1] * eq);       // d loss / d w_q
    }
}

extern "C" int i2p_pose_loss(int B, int l1_trans, const float *out3, const float *out4, const float *q_gt, const float *t_gt,
                             const float *w_x, const float *w_q, float *loss3, float *d_out3, float *d_out4, float *d_w,
                             void *stream) {
    if (B <= 0 || B > 1024) return I2P_ERR_BAD_ARG;
    if (!out3 || !out4 || !q_gt || !t_gt || !w_x || !w_q || !loss3 || !d_out3 || !d_out4 || !d_w) return I2P_ERR_BAD_ARG;
    hipLaunchKernelGGL(pose_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, B, l1_trans, out3, out4, q_gt, t_gt, w_x, w_q,
                       loss3, d_out3, d_out4, d_w);
    I2P_RETURN_LAUNCH_STATUS();
}

