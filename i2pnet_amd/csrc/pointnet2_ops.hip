// PointNet++ primitives for gfx950 — replaces pointnet2/src/{sampling,ball_query,group_points,
// interpolate}_gpu.cu of the reference (launchers cited in include/i2p_ops.h).
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------
// Furthest point sampling (reference: sampling_gpu.cu:93-209).
//
// The reference's result under equal distances depends on its launch geometry: BS =
// opt_n_threads(N) threads scan k = tid, tid+BS, ... keeping their FIRST maximum, then an LDS
// tree keeps the LOWER slot on ties.  That tree is equivalent to the total order
//     (d2 descending, bitrev_{log2 BS}(k mod BS) ascending, k ascending)
// (two tids meet at the level of their lowest differing bit and the one with a 0 there wins),
// so ANY reduction under this order returns the reference's pick.  Here: one 1024-thread
// block per sample, coordinates and running min-distances register-resident for N <= 8192,
// DPP/shuffle wave reduction, one LDS exchange + one barrier per iteration.
// ------------------------------------------------------------------------------------------
struct FpsBest {
    float d;
    unsigned rk;   // (bitrev(k mod BS) << 16 | ...) is too narrow for k: keep rank and k apart
    int k;
};

__device__ __forceinline__ bool fps_better(float d2, unsigned r2, int k2, float d1, unsigned r1,
                                           int k1) {
    // true if candidate 2 beats candidate 1 under the total order above
    return (d2 > d1) || (d2 == d1 && (r2 < r1 || (r2 == r1 && k2 < k1)));
}

__device__ __forceinline__ unsigned fps_rank(int k, int bs_mask, int bs_shift) {
    // bit-reverse the low log2(BS) bits of k
    return bs_shift >= 32 ? 0u : (__brev((unsigned)(k & bs_mask)) >> bs_shift);
}

constexpr int FPS_THREADS = 1024;
constexpr int FPS_REG_PTS = 8;     // register-resident points per thread (N <= 8192)

template <bool IN_REGS>
__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(int n, int m, int bs,
                                                           const float *__restrict__ dataset,
                                                           float *__restrict__ temp,
                                                           int *__restrict__ idxs) {
    __shared__ float s_d[2][FPS_THREADS / I2P_WAVE];
    __shared__ unsigned s_r[2][FPS_THREADS / I2P_WAVE];
    __shared__ int s_k[2][FPS_THREADS / I2P_WAVE];

    const int bi = blockIdx.x;
    dataset += (size_t)bi * n * 3; temp += (size_t)bi * n; idxs += (size_t)bi * m;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int bs_mask = bs - 1;
    const int bs_shift = 32 - (31 - __clz(bs));       // 32 - log2(bs); bs == 1 -> 32

    float px[FPS_REG_PTS], py[FPS_REG_PTS], pz[FPS_REG_PTS], td[FPS_REG_PTS];
    if (IN_REGS) {
#pragma unroll
        for (int i = 0; i < FPS_REG_PTS; ++i) {
            const int k = tid + i * FPS_THREADS;
            if (k < n) {
                px[i] = dataset[k * 3 + 0]; py[i] = dataset[k * 3 + 1]; pz[i] = dataset[k * 3 + 2];
                td[i] = temp[k];
            } else { px[i] = py[i] = pz[i] = 0.f; td[i] = 0.f; }
        }
    }

    int old = 0;
    if (tid == 0) idxs[0] = 0;                                            // sampling_gpu.cu:112-114
    for (int j = 1; j < m; ++j) {
        const float x1 = dataset[old * 3 + 0], y1 = dataset[old * 3 + 1], z1 = dataset[old * 3 + 2];
        float bd = -1.f; unsigned br = 0xffffffffu; int bk = 0x7fffffff;
        // a thread with no point keeps (-1, worst rank): the reference's idle thread reports
        // (best=-1, besti=0), which can only win when every d2 is < -1, i.e. never.
        if (IN_REGS) {
#pragma unroll
            for (int i = 0; i < FPS_REG_PTS; ++i) {
                const int k = tid + i * FPS_THREADS;
                if (k < n) {
                    const float d = i2p_sq3(px[i] - x1, py[i] - y1, pz[i] - z1); // :132
                    const float d2 = fminf(d, td[i]);                             // :133
                    td[i] = d2;
                    const unsigned r = fps_rank(k, bs_mask, bs_shift);
                    if (fps_better(d2, r, k, bd, br, bk)) { bd = d2; br = r; bk = k; }
                }
            }
        } else {
            for (int k = tid; k < n; k += FPS_THREADS) {
                const float d = i2p_sq3(dataset[k * 3 + 0] - x1, dataset[k * 3 + 1] - y1,
                                        dataset[k * 3 + 2] - z1);
                const float d2 = fminf(d, temp[k]);
                temp[k] = d2;
                const unsigned r = fps_rank(k, bs_mask, bs_shift);
                if (fps_better(d2, r, k, bd, br, bk)) { bd = d2; br = r; bk = k; }
            }
        }
        // wave reduction (total order => any tree is fine)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float od = __shfl_xor(bd, off); const unsigned orr = __shfl_xor(br, off);
            const int ok = __shfl_xor(bk, off);
            if (fps_better(od, orr, ok, bd, br, bk)) { bd = od; br = orr; bk = ok; }
        }
        const int buf = j & 1;
        if (lane == 0) { s_d[buf][wv] = bd; s_r[buf][wv] = br; s_k[buf][wv] = bk; }
        __syncthreads();
        // every wave reduces the 16 partials redundantly: no second barrier
        float cd = -1.f; unsigned cr = 0xffffffffu; int ck = 0x7fffffff;
        if (lane < FPS_THREADS / I2P_WAVE) { cd = s_d[buf][lane]; cr = s_r[buf][lane]; ck = s_k[buf][lane]; }
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            const float od = __shfl_xor(cd, off); const unsigned orr = __shfl_xor(cr, off);
            const int ok = __shfl_xor(ck, off);
            if (fps_better(od, orr, ok, cd, cr, ck)) { cd = od; cr = orr; ck = ok; }
        }
        old = __shfl(ck, 0);
        // all-idle corner (n == 0 is rejected by the launcher); reference default besti = 0
        if (old == 0x7fffffff) old = 0;
        if (tid == 0) idxs[j] = old;                                      // :205-207
    }
    if (IN_REGS) {                                                        // temp holds the final min-distances
#pragma unroll
        for (int i = 0; i < FPS_REG_PTS; ++i) {
            const int k = tid + i * FPS_THREADS;
            if (k < n) temp[k] = td[i];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Second-generation FPS for clouds that fit a thread block's registers (n <= 8192): what bounds the kernel is
// the chain of m-1 dependent block-wide arg-max reductions, so every iteration is trimmed to
//   * candidates as ONE 64-bit key  (distance bits : ~(rank << 13 | k))  — the distance is a non-negative float, so
//     its bit pattern orders like the value, and the low word reproduces the reference's tie rule (smaller
//     bit-reversed rank, then smaller index) under an unsigned max;
//   * a DPP row reduction (4 steps, no LDS traffic) + 4 scalar lane reads per wave instead of 18 ds_bpermutes;
//   * one barrier; every wave reduces the 16 wave keys redundantly;
//   * the winner's coordinates from an LDS copy of the cloud (96 KB) instead of a dependent global load.
// First generation (kept for n > 8192): 2.4 us per iteration at n = 8192.
// ------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ void fps_dpp_max(unsigned &hi, unsigned &lo) {
    const unsigned ohi = i2p_dpp_u32<CTRL>(hi), olo = i2p_dpp_u32<CTRL>(lo);
    const bool take = (ohi > hi) || (ohi == hi && olo > lo);
    hi = take ? ohi : hi; lo = take ? olo : lo;
}

__device__ __forceinline__ unsigned long long fps_row_max(unsigned hi, unsigned lo) {
    fps_dpp_max<0xB1>(hi, lo);      // quad_perm [1,0,3,2]
    fps_dpp_max<0x4E>(hi, lo);      // quad_perm [2,3,0,1]
    fps_dpp_max<0x141>(hi, lo);     // row_half_mirror
    fps_dpp_max<0x140>(hi, lo);     // row_mirror: every lane of a 16-lane row now holds the row maximum
    return ((unsigned long long)hi << 32) | lo;
}

// T threads own the cloud (8 points each): fewer waves for the small pyramid levels = a cheaper barrier; the thread
// count only changes which thread holds a point, not the result (the tie rule lives in the key).
template <int T>
__global__ __launch_bounds__(T) void fps_kernel_fast(int n, int m, int bs, const float *__restrict__ dataset,
                                                                float *__restrict__ temp, int *__restrict__ idxs) {
    extern __shared__ float s_pts[];                                      // [n][3] copy of this sample's cloud
    __shared__ unsigned long long s_key[2][(T / I2P_WAVE)];
    const int bi = blockIdx.x;
    dataset += (size_t)bi * n * 3; temp += (size_t)bi * n; idxs += (size_t)bi * m;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int bs_mask = bs - 1;
    const int bs_shift = 32 - (31 - __clz(bs));
    for (int i = tid; i < n * 3; i += T) s_pts[i] = dataset[i];

    float px[FPS_REG_PTS], py[FPS_REG_PTS], pz[FPS_REG_PTS], td[FPS_REG_PTS];
    unsigned lo_k[FPS_REG_PTS];                                           // ~(rank << 13 | k): constant per point
#pragma unroll
    for (int i = 0; i < FPS_REG_PTS; ++i) {
        const int k = tid + i * T;
        if (k < n) {
            px[i] = dataset[k * 3 + 0]; py[i] = dataset[k * 3 + 1]; pz[i] = dataset[k * 3 + 2];
            td[i] = temp[k];
            lo_k[i] = ~((fps_rank(k, bs_mask, bs_shift) << 13) | (unsigned)k);
        } else { px[i] = py[i] = pz[i] = 0.f; td[i] = 0.f; lo_k[i] = 0u; }
    }
    __syncthreads();

    int old = 0;
    if (tid == 0) idxs[0] = 0;                                            // sampling_gpu.cu:112-114
    for (int j = 1; j < m; ++j) {
        const float x1 = s_pts[old * 3 + 0], y1 = s_pts[old * 3 + 1], z1 = s_pts[old * 3 + 2];
        unsigned bhi = 0u, blo = 0u;                                      // an idle thread never wins (real keys have lo != 0)
#pragma unroll
        for (int i = 0; i < FPS_REG_PTS; ++i) {
            const int k = tid + i * T;
            if (k < n) {
                const float d = i2p_sq3(px[i] - x1, py[i] - y1, pz[i] - z1);      // :132
                const float d2 = fminf(d, td[i]);                                  // :133
                td[i] = d2;
                const unsigned hi = i2p_f2u(d2);
                const bool take = (hi > bhi) || (hi == bhi && lo_k[i] > blo);
                bhi = take ? hi : bhi; blo = take ? lo_k[i] : blo;
            }
        }
        unsigned long long key = fps_row_max(bhi, blo);
        {   // the four 16-lane rows of the wave
            const unsigned h1 = __builtin_amdgcn_readlane((unsigned)(key >> 32), 16), l1 = __builtin_amdgcn_readlane((unsigned)key, 16);
            const unsigned h2 = __builtin_amdgcn_readlane((unsigned)(key >> 32), 32), l2 = __builtin_amdgcn_readlane((unsigned)key, 32);
            const unsigned h3 = __builtin_amdgcn_readlane((unsigned)(key >> 32), 48), l3 = __builtin_amdgcn_readlane((unsigned)key, 48);
            const unsigned h0 = __builtin_amdgcn_readlane((unsigned)(key >> 32), 0), l0 = __builtin_amdgcn_readlane((unsigned)key, 0);
            unsigned long long k0 = ((unsigned long long)h0 << 32) | l0, k1 = ((unsigned long long)h1 << 32) | l1;
            unsigned long long k2 = ((unsigned long long)h2 << 32) | l2, k3 = ((unsigned long long)h3 << 32) | l3;
            k0 = k0 > k1 ? k0 : k1; k2 = k2 > k3 ? k2 : k3;
            key = k0 > k2 ? k0 : k2;                                       // wave-uniform
        }
        unsigned wlo = (unsigned)key;
        if (T > I2P_WAVE) {
            const int buf = j & 1;
            if (lane == 0) s_key[buf][wv] = key;
            __syncthreads();
            unsigned long long ck = lane < (T / I2P_WAVE) ? s_key[buf][lane] : 0ull;   // lanes 0..15 = row 0
            ck = fps_row_max((unsigned)(ck >> 32), (unsigned)ck);
            wlo = __builtin_amdgcn_readlane((unsigned)ck, 0);
        }
        old = (int)((~wlo) & 0x1fffu);
        if (tid == 0) idxs[j] = old;                                      // :205-207
    }
#pragma unroll
    for (int i = 0; i < FPS_REG_PTS; ++i) {
        const int k = tid + i * T;
        if (k < n) temp[k] = td[i];
    }
}

// ------------------------------------------------------------------------------------------
// gather_points (+grad) — sampling_gpu.cu:8-24 / :46-63.  One thread per output point walks
// the channel chunk so idx is read once, stores are coalesced along M for every channel.
// ------------------------------------------------------------------------------------------
constexpr int CH_CHUNK = 16;

__global__ void gather_points_kernel(int c, int n, int m, const float *__restrict__ points,
                                     const int *__restrict__ idx, float *__restrict__ out) {
    const int bi = blockIdx.z, c0 = blockIdx.y * CH_CHUNK;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    const int src = idx[(size_t)bi * m + p];
    const int c1 = min(c0 + CH_CHUNK, c);
    for (int ci = c0; ci < c1; ++ci)
        out[((size_t)bi * c + ci) * m + p] = points[((size_t)bi * c + ci) * n + src];
}

__global__ void gather_points_grad_kernel(int c, int n, int m, const float *__restrict__ grad_out,
                                          const int *__restrict__ idx,
                                          float *__restrict__ grad_points) {
    const int bi = blockIdx.z, c0 = blockIdx.y * CH_CHUNK;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    const int dst = idx[(size_t)bi * m + p];
    const int c1 = min(c0 + CH_CHUNK, c);
    for (int ci = c0; ci < c1; ++ci)
        atomicAdd(grad_points + ((size_t)bi * c + ci) * n + dst, grad_out[((size_t)bi * c + ci) * m + p]);
}

// ------------------------------------------------------------------------------------------
// group_points (+grad) — group_points_gpu.cu:47-66 / :8-25.  Same shape as gather with
// M = npoints*nsample.
// ------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------
// ball_query — ball_query_gpu.cu:9-45.  One wave per query: 64 points tested per step,
// ballot + prefix popcount keep the reference's ascending-k fill order and early exit.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ball_query_kernel(int n, int m, float radius2, int nsample,
                                                         const float *__restrict__ new_xyz,
                                                         const float *__restrict__ xyz,
                                                         int *__restrict__ idx) {
    const int bi = blockIdx.y;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= m) return;
    const float *qp = new_xyz + ((size_t)bi * m + q) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const float *src = xyz + (size_t)bi * n * 3;
    int *o = idx + ((size_t)bi * m + q) * nsample;
    int cnt = 0;
    for (int base = 0; base < n && cnt < nsample; base += 64) {
        const int k = base + lane;
        bool hit = false;
        if (k < n) {
            const float d2 = i2p_sq3(qx - src[k * 3 + 0], qy - src[k * 3 + 1], qz - src[k * 3 + 2]); // :33
            hit = d2 < radius2;                                                                     // :34
        }
        const unsigned long long mask = __ballot(hit);
        if (mask == 0ull) continue;
        if (cnt == 0) {                                                   // :35-39
            const int first = base + __builtin_ctzll(mask);
            for (int l = lane; l < nsample; l += 64) o[l] = first;
        }
        const int rank = cnt + __popcll(mask & ((1ull << lane) - 1ull));
        if (hit && rank < nsample) o[rank] = k;                           // :40-42
        cnt += __popcll(mask);
    }
}

// ------------------------------------------------------------------------------------------
// three_nn — interpolate_gpu.cu:9-52.  Thread per unknown point, `known` streamed through LDS.
// ------------------------------------------------------------------------------------------
constexpr int NN_TILE = 1024;

__global__ __launch_bounds__(256) void three_nn_kernel(int n, int m, const float *__restrict__ unknown,
                                                       const float *__restrict__ known,
                                                       float *__restrict__ dist2,
                                                       int *__restrict__ idx) {
    __shared__ float tile[NN_TILE * 3];
    const int bi = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = p < n;
    float ux = 0, uy = 0, uz = 0;
    if (act) {
        const float *u = unknown + ((size_t)bi * n + p) * 3;
        ux = u[0]; uy = u[1]; uz = u[2];
    }
    const float *kn = known + (size_t)bi * m * 3;
    double best1 = 1e40, best2 = 1e40, best3 = 1e40;                      // :28
    int besti1 = 0, besti2 = 0, besti3 = 0;
    for (int base = 0; base < m; base += NN_TILE) {
        const int cntp = min(NN_TILE, m - base);
        __syncthreads();
        for (int i = threadIdx.x; i < cntp * 3; i += blockDim.x) tile[i] = kn[(size_t)base * 3 + i];
        __syncthreads();
        if (act)
            for (int k = 0; k < cntp; ++k) {
                const float d = i2p_sq3(ux - tile[k * 3 + 0], uy - tile[k * 3 + 1], uz - tile[k * 3 + 2]);
                const double dd = (double)d;
                if (dd < best1) {                                         // :35-48
                    best3 = best2; besti3 = besti2; best2 = best1; besti2 = besti1;
                    best1 = dd; besti1 = base + k;
                } else if (dd < best2) {
                    best3 = best2; besti3 = besti2; best2 = dd; besti2 = base + k;
                } else if (dd < best3) {
                    best3 = dd; besti3 = base + k;
                }
            }
    }
    if (act) {
        float *od = dist2 + ((size_t)bi * n + p) * 3;
        int *oi = idx + ((size_t)bi * n + p) * 3;
        od[0] = (float)best1; od[1] = (float)best2; od[2] = (float)best3; // :50
        oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;                   // :51
    }
}

// three_interpolate (+grad) — interpolate_gpu.cu:77-97 / :120-142
__global__ void three_interpolate_kernel(int c, int m, int n, const float *__restrict__ points,
                                         const int *__restrict__ idx,
                                         const float *__restrict__ weight,
                                         float *__restrict__ out) {
    const int bi = blockIdx.z, c0 = blockIdx.y * CH_CHUNK;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float *w = weight + ((size_t)bi * n + p) * 3;
    const int *id = idx + ((size_t)bi * n + p) * 3;
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    const int i0 = id[0], i1 = id[1], i2 = id[2];
    const int c1 = min(c0 + CH_CHUNK, c);
    for (int ci = c0; ci < c1; ++ci) {
        const float *pt = points + ((size_t)bi * c + ci) * m;
        out[((size_t)bi * c + ci) * n + p] =
            __fmaf_rn(w2, pt[i2], __fmaf_rn(w1, pt[i1], __fmul_rn(w0, pt[i0]))); // :96
    }
}

__global__ void three_interpolate_grad_kernel(int c, int n, int m, const float *__restrict__ grad_out,
                                              const int *__restrict__ idx,
                                              const float *__restrict__ weight,
                                              float *__restrict__ grad_points) {
    const int bi = blockIdx.z, c0 = blockIdx.y * CH_CHUNK;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float *w = weight + ((size_t)bi * n + p) * 3;
    const int *id = idx + ((size_t)bi * n + p) * 3;
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    const int i0 = id[0], i1 = id[1], i2 = id[2];
    const int c1 = min(c0 + CH_CHUNK, c);
    for (int ci = c0; ci < c1; ++ci) {
        const float g = grad_out[((size_t)bi * c + ci) * n + p];
        float *gp = grad_points + ((size_t)bi * c + ci) * m;
        atomicAdd(gp + i0, __fmul_rn(g, w0));                             // :139-141
        atomicAdd(gp + i1, __fmul_rn(g, w1));
        atomicAdd(gp + i2, __fmul_rn(g, w2));
    }
}

inline dim3 chunk_grid(int pts, int c, int b) {
    return dim3((pts + 255) / 256, (c + CH_CHUNK - 1) / CH_CHUNK, b);
}

}  // namespace

extern "C" int i2p_abi_version(void) { return 1; }

extern "C" int i2p_furthest_point_sampling(int b, int n, int m, const float *dataset,
                                           float *temp, int *idxs, void *stream) {
    if (b < 0 || n < 0 || m < 0) return I2P_ERR_BAD_ARG;
    if (b == 0 || m <= 0) return 0;                                       // sampling_gpu.cu:100
    if (n <= 0 || !dataset || !temp || !idxs) return I2P_ERR_BAD_ARG;
    // opt_n_threads, pointnet2/src/cuda_utils.h:10-14 (same double expression as the reference)
    const int pow_2 = (int)(std::log((double)n) / std::log(2.0));
    int bs = 1 << pow_2;
    bs = bs > 1024 ? 1024 : bs; bs = bs < 1 ? 1 : bs;
    hipStream_t st = (hipStream_t)stream;
    if (n <= FPS_THREADS * FPS_REG_PTS) {
        const size_t lds = (size_t)n * 3 * sizeof(float);
        if (n <= 64 * FPS_REG_PTS) {
            hipLaunchKernelGGL(fps_kernel_fast<64>, dim3(b), dim3(64), lds, st, n, m, bs, dataset, temp, idxs);
        } else if (n <= 256 * FPS_REG_PTS) {
            hipLaunchKernelGGL(fps_kernel_fast<256>, dim3(b), dim3(256), lds, st, n, m, bs, dataset, temp, idxs);
        } else {
            static bool attr_set = false;
            if (!attr_set) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fps_kernel_fast<FPS_THREADS>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, FPS_THREADS * FPS_REG_PTS * 3 * (int)sizeof(float));
                attr_set = true;
            }
            hipLaunchKernelGGL(fps_kernel_fast<FPS_THREADS>, dim3(b), dim3(FPS_THREADS), lds, st, n, m, bs, dataset, temp, idxs);
        }
    } else if (n <= FPS_THREADS * FPS_REG_PTS)
        hipLaunchKernelGGL(fps_kernel<true>, dim3(b), dim3(FPS_THREADS), 0, st, n, m, bs, dataset, temp, idxs);
    else
        hipLaunchKernelGGL(fps_kernel<false>, dim3(b), dim3(FPS_THREADS), 0, st, n, m, bs, dataset, temp, idxs);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_gather_points(int b, int c, int n, int npoints, const float *points,
                                 const int *idx, float *out, void *stream) {
    if (b < 0 || c < 0 || n < 0 || npoints < 0) return I2P_ERR_BAD_ARG;
    if ((long long)b * c * npoints == 0) return 0;
    if (!points || !idx || !out) return I2P_ERR_BAD_ARG;
    hipLaunchKernelGGL(gather_points_kernel, chunk_grid(npoints, c, b), dim3(256), 0,
                       (hipStream_t)stream, c, n, npoints, points, idx, out);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                                      const int *idx, float *grad_points, void *stream) {
    if (b < 0 || c < 0 || n < 0 || npoints < 0) return I2P_ERR_BAD_ARG;
    if ((long long)b * c * npoints == 0) return 0;
    if (!grad_out || !idx || !grad_points) return I2P_ERR_BAD_ARG;
    if (!i2p_atomic_scatter()) return i2p_det_gather_points_grad(b, c, n, npoints, grad_out, idx, grad_points, stream);
    hipLaunchKernelGGL(gather_points_grad_kernel, chunk_grid(npoints, c, b), dim3(256), 0,
                       (hipStream_t)stream, c, n, npoints, grad_out, idx, grad_points);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_group_points(int b, int c, int n, int npoints, int nsample,
                                const float *points, const int *idx, float *out, void *stream) {
    if (nsample < 0) return I2P_ERR_BAD_ARG;
    return i2p_gather_points(b, c, n, npoints * nsample, points, idx, out, stream);
}

extern "C" int i2p_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                     const float *grad_out, const int *idx, float *grad_points,
                                     void *stream) {
    if (nsample < 0) return I2P_ERR_BAD_ARG;
    return i2p_gather_points_grad(b, c, n, npoints * nsample, grad_out, idx, grad_points, stream);
}

extern "C" int i2p_ball_query(int b, int n, int m, float radius, int nsample,
                              const float *new_xyz, const float *xyz, int *idx, void *stream) {
    if (b < 0 || n < 0 || m < 0 || nsample < 0) return I2P_ERR_BAD_ARG;
    if ((long long)b * m == 0 || nsample == 0) return 0;
    if (!new_xyz || !idx || (n > 0 && !xyz)) return I2P_ERR_BAD_ARG;
    hipLaunchKernelGGL(ball_query_kernel, dim3((m + 3) / 4, b), dim3(256), 0, (hipStream_t)stream,
                       n, m, radius * radius, nsample, new_xyz, xyz, idx);   // ball_query_gpu.cu:23
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_three_nn(int b, int n, int m, const float *unknown, const float *known,
                            float *dist2, int *idx, void *stream) {
    if (b < 0 || n < 0 || m < 0) return I2P_ERR_BAD_ARG;
    if ((long long)b * n == 0) return 0;
    if (!unknown || !dist2 || !idx || (m > 0 && !known)) return I2P_ERR_BAD_ARG;
    hipLaunchKernelGGL(three_nn_kernel, dim3((n + 255) / 256, b), dim3(256), 0, (hipStream_t)stream,
                       n, m, unknown, known, dist2, idx);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_three_interpolate(int b, int c, int m, int n, const float *points,
                                     const int *idx, const float *weight, float *out,
                                     void *stream) {
    if (b < 0 || c < 0 || n < 0 || m < 0) return I2P_ERR_BAD_ARG;
    if ((long long)b * c * n == 0) return 0;
    if (!points || !idx || !weight || !out) return I2P_ERR_BAD_ARG;
    hipLaunchKernelGGL(three_interpolate_kernel, chunk_grid(n, c, b), dim3(256), 0,
                       (hipStream_t)stream, c, m, n, points, idx, weight, out);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                          const int *idx, const float *weight,
                                          float *grad_points, void *stream) {
    if (b < 0 || c < 0 || n < 0 || m < 0) return I2P_ERR_BAD_ARG;
    if ((long long)b * c * n == 0) return 0;
    if (!grad_out || !idx || !weight || !grad_points) return I2P_ERR_BAD_ARG;
    if (!i2p_atomic_scatter()) return i2p_det_three_interpolate_grad(b, c, n, m, grad_out, idx, weight, grad_points, stream);
    hipLaunchKernelGGL(three_interpolate_grad_kernel, chunk_grid(n, c, b), dim3(256), 0,
                       (hipStream_t)stream, c, n, m, grad_out, idx, weight, grad_points);
    I2P_RETURN_LAUNCH_STATUS();
}
