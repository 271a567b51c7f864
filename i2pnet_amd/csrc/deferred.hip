// Deferred weight-gradient reductions (round 6).  Every layer backward leaves its weight gradient as per-block partial slabs and
// sums them with a launch of its own (reduce_partials_v4_kernel, big_reduce_kernel, gemm_tn_reduce_kernel, chain_reduce_kernel in fp32,
// reduce_partials_bf16 in bf16 storage): 43 launches of a captured fp32 step, each a dependent launch of >= 4.5 us that moves a few
// KB .. 16 MB — and nothing needs a weight gradient before the optimiser (reference: train20v2learn_wandb_proj.py:473-481, backward,
// clip, step).  Between i2p_defer_begin() and i2p_defer_flush() the launchers RECORD their reduction (partials, destination, counts)
// instead of launching it; the flush sums all of them in one launch per 24 recorded entries.  A block of the flush kernel does exactly
// what a block of the kernel it replaces did — the fp32 kernels all share one shape: 16 float4 columns x 16 partial lanes, a lane
// adding its partials in ascending order, the 16 lanes added in order; the bf16 one 32 columns x 8 lanes — so the sums are
// bit-identical to the immediate form (tests/test_train_gpu.py).  The caller keeps the partial buffers alive until the flush
// (i2pnet_amd/ops.py) and must not read a weight gradient earlier; i2p_defer_pause() brackets a call whose result is consumed at
// once (the column slice of a padded first-layer weight gradient).
#include "common.h"
#include <mutex>
#include <vector>

namespace {

struct DefEntry {
    const void *parts; void *out;
    int nparts, n, block0, kind;      // kind 0: n = float4 columns (16 x 16 shape); 1: n = floats (32 x 8 shape);
    int pitch, cols;                  // kind 2: kind 0's sums of the first `cols` columns of [rows][pitch] slabs, written as [rows][cols] (n = rows * cols)
    int s_out, s_in, s_kh, s_kw;      // kind 3: conv3x3_wgrad_fin_kernel (image_conv16.hip): nparts blocks x [NT = n][9][256] partials, fp64 sums,
};                                    //         written at the weight's element strides
constexpr int DEF_MAX = 24;
struct DefTable { DefEntry e[DEF_MAX]; int count; };

__global__ __launch_bounds__(256) void deferred_reduce_kernel(DefTable t) {
    __shared__ __attribute__((aligned(16))) float4 red4[16][16];        // 4 KB, re-viewed per kind
    int i = 0;
#pragma unroll 1
    for (int j = 1; j < t.count; ++j) if ((int)blockIdx.x >= t.e[j].block0) i = j;
    const DefEntry e = t.e[i];
    const int blk = (int)blockIdx.x - e.block0;
    if (e.kind == 0) {
        const float4 *__restrict__ parts = reinterpret_cast<const float4 *>(e.parts);
        const int n4 = e.n, nparts = e.nparts;
        const int tx = threadIdx.x & 15, pl = threadIdx.x >> 4;
        const int o = blk * 16 + tx;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (o < n4) {
            int b = pl;
            for (; b + 48 < nparts; b += 64) {
                const float4 v0 = parts[(size_t)b * n4 + o], v1 = parts[(size_t)(b + 16) * n4 + o];
                const float4 v2 = parts[(size_t)(b + 32) * n4 + o], v3 = parts[(size_t)(b + 48) * n4 + o];
                a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
                a.x += v1.x; a.y += v1.y; a.z += v1.z; a.w += v1.w;
                a.x += v2.x; a.y += v2.y; a.z += v2.z; a.w += v2.w;
                a.x += v3.x; a.y += v3.y; a.z += v3.z; a.w += v3.w;
            }
            for (; b < nparts; b += 16) {
                const float4 v = parts[(size_t)b * n4 + o];
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
        }
        red4[pl][tx] = a;
        __syncthreads();
        if (pl == 0 && o < n4) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < 16; ++q) { const float4 v = red4[q][tx]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
            reinterpret_cast<float4 *>(e.out)[o] = s;
        }
    } else if (e.kind == 2) {
        // column-compacting form of kind 0 (the weight gradient of a layer whose input rows are zero-padded: only the real columns are
        // kept): per element the same additions in the same order as kind 0 followed by a column slice
        float (*red)[16] = reinterpret_cast<float (*)[16]>(&red4[0][0]);
        const float *__restrict__ parts = reinterpret_cast<const float *>(e.parts);
        const int nparts = e.nparts, cols = e.cols;
        const size_t slab = (size_t)(e.n / cols) * e.pitch;
        const int tx = threadIdx.x & 15, pl = threadIdx.x >> 4;
        const int o = blk * 16 + tx;
        float a = 0.f;
        if (o < e.n) {
            const int r = o / cols, c = o - r * cols;
            const float *src = parts + (size_t)r * e.pitch + c;
            for (int b = pl; b < nparts; b += 16) a += src[(size_t)b * slab];
        }
        red[pl][tx] = a;
        __syncthreads();
        if (pl == 0 && o < e.n) {
            float s2 = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) s2 += red[q][tx];
            reinterpret_cast<float *>(e.out)[o] = s2;
        }
    } else if (e.kind == 3) {
        // a quarter of a block of conv3x3_wgrad_fin_kernel: 16 outputs x 16 row groups, group g adding blocks g, g + 16, ... in fp64, the
        // groups added in order — the same additions per element as the 1024-thread kernel it replaces
        double (*part)[16] = reinterpret_cast<double (*)[16]>(&red4[0][0]);
        const float *__restrict__ partials = reinterpret_cast<const float *>(e.parts);
        const int NT = e.n, nblk = e.nparts;
        const int tt = blk >> 4, nt = tt / 9, t = tt - nt * 9, el = (blk & 15) * 16 + (threadIdx.x & 15), grp = threadIdx.x >> 4;
        double a = 0.0;
        for (int b = grp; b < nblk; b += 16) a += (double)partials[((size_t)b * NT + nt) * (9 * 256) + t * 256 + el];
        part[grp][threadIdx.x & 15] = a;
        __syncthreads();
        if (threadIdx.x < 16) {
            double v = 0.0;
#pragma unroll
            for (int g2 = 0; g2 < 16; ++g2) v += part[g2][threadIdx.x];
            const int co = 16 * nt + (el >> 4), ci = el & 15;
            reinterpret_cast<float *>(e.out)[co * e.s_out + ci * e.s_in + (t / 3) * e.s_kh + (t % 3) * e.s_kw] = (float)v;
        }
    } else {
        float (*red)[32] = reinterpret_cast<float (*)[32]>(&red4[0][0]);
        const float *__restrict__ parts = reinterpret_cast<const float *>(e.parts);
        const int n = e.n, nparts = e.nparts;
        const int o = blk * 32 + (threadIdx.x & 31), pl = threadIdx.x >> 5;
        float a = 0.f;
        if (o < n)
            for (int b = pl; b < nparts; b += 8) a += parts[(size_t)b * n + o];
        red[pl][threadIdx.x & 31] = a;
        __syncthreads();
        if (pl == 0 && o < n) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += red[q][threadIdx.x & 31];
            reinterpret_cast<float *>(e.out)[o] = s;
        }
    }
}

std::mutex g_mu;                       // (backward nodes run on the autograd engine's device thread, begin / flush on the caller's)
bool g_on = false;
int g_pause = 0;
std::vector<DefEntry> g_list;
int g_c_rows = 0, g_c_pitch = 0, g_c_cols = 0;   // one-shot: the next recorded kind-0 reduction of rows * pitch floats keeps `cols` columns
bool g_c_used = false;

}  // namespace

// launchers: true = recorded (do not launch), false = deferral is off, launch as before
bool i2p_defer_reduce(int kind, int nparts, int n, const void *parts, void *out) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_on || g_pause > 0 || nparts <= 0 || n <= 0) return false;
    if (g_c_rows > 0) {                // a compaction request is pending: this reduction is recorded in the compact form, or it runs NOW
        const bool fits = kind == 0 && (long long)n * 4 == (long long)g_c_rows * g_c_pitch;
        const int rows = g_c_rows;
        g_c_rows = 0;
        if (!fits) return false;       // (bf16 slabs, another shape: the caller slices the immediate result)
        g_list.push_back(DefEntry{parts, out, nparts, rows * g_c_cols, 0, 2, g_c_pitch, g_c_cols, 0, 0, 0, 0});
        g_c_used = true;
        return true;
    }
    g_list.push_back(DefEntry{parts, out, nparts, n, 0, kind, 0, 0, 0, 0, 0, 0});
    return true;
}

// the finalisation of an image-encoder 3x3 weight gradient (csrc/image_conv16.hip): nblk blocks of [NT][9][256] fp32 partials, summed
// in fp64, written at the weight's element strides
bool i2p_defer_conv_fin(int nblk, int NT, const float *partials, float *dW, int s_out, int s_in, int s_kh, int s_kw) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_on || g_pause > 0 || g_c_rows > 0 || nblk <= 0 || NT <= 0) return false;
    g_list.push_back(DefEntry{partials, dW, nblk, NT, 0, 3, 0, 0, s_out, s_in, s_kh, s_kw});
    return true;
}

extern "C" int i2p_defer_begin(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = true; g_pause = 0; g_list.clear(); g_c_rows = 0; g_c_used = false;
    return 0;
}

extern "C" int i2p_defer_pause(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_pause += on ? 1 : -1;
    if (g_pause < 0) g_pause = 0;
    return 0;
}

// One-shot request for the NEXT recorded fp32 reduction, if it sums [rows][pitch] slabs: keep the first `cols` columns and write them
// densely as [rows][cols] at the start of `out` (the weight gradient of a layer whose input rows carry zero padding).  rows = 0 withdraws
// the request; the return value says whether the previous request was taken (1) — if not, the caller slices the full result itself.
extern "C" int i2p_defer_compact_next(int rows, int pitch, int cols) {
    std::lock_guard<std::mutex> lk(g_mu);
    const int used = g_c_used ? 1 : 0;
    g_c_used = false;
    g_c_rows = (rows > 0 && pitch >= cols && cols > 0) ? rows : 0; g_c_pitch = pitch; g_c_cols = cols;
    return used;
}

extern "C" int i2p_defer_pending(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return (int)g_list.size();
}

// sums every recorded reduction on `stream` (one launch per 24 entries) and forgets them; deferral stays on until i2p_defer_end()
extern "C" int i2p_defer_flush(void *stream) {
    std::vector<DefEntry> todo;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        todo.swap(g_list);
    }
    for (size_t s0 = 0; s0 < todo.size(); s0 += DEF_MAX) {
        DefTable t;
        t.count = (int)(todo.size() - s0 < (size_t)DEF_MAX ? todo.size() - s0 : (size_t)DEF_MAX);
        int blocks = 0;
        for (int j = 0; j < t.count; ++j) {
            t.e[j] = todo[s0 + j];
            t.e[j].block0 = blocks;
            blocks += t.e[j].kind == 1 ? (t.e[j].n + 31) / 32 : t.e[j].kind == 3 ? 9 * t.e[j].n * 16 : (t.e[j].n + 15) / 16;
        }
        hipLaunchKernelGGL(deferred_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t);
    }
    I2P_RETURN_LAUNCH_STATUS();
}

// -> number of recorded reductions that were never flushed (their weight gradients are garbage: the caller raises)
extern "C" int i2p_defer_end(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    const int n = (int)g_list.size();
    g_on = false; g_pause = 0; g_list.clear();
    return n;
}
