// Deterministic scatter-add: the backward of every gather of the path (gather_torch utils.py:36-60 -> torch.gather's
// backward; GroupingOperation / GatherOperation / ThreeInterpolate backward, group_points_gpu.cu:8-25,
// sampling_gpu.cu:46-63, interpolate_gpu.cu:120-142) without floating-point atomics.
//
// The reference accumulates with atomicAdd in whatever order the hardware serialises them, so its gradients are not
// reproducible run to run.  Here every group of destination cells is OWNED by one block whose summation order is a
// fixed function of the index list alone:
//   * the block stages the index list in LDS (4096 rows per pass) — a scan straight from global memory was
//     latency-bound: one dependent L2 round trip per 64 rows and wave;
//   * its 16 waves split the ROWS of every pass into 16 contiguous ranges; a wave walks its range in ascending order,
//     64 rows at a time: ballot of the rows that fall into the block's cells, then the matching rows one after the
//     other (lane = channel, sixteen source rows in flight) into its private LDS accumulators;
//   * consecutive matches of ONE cell are summed in registers and reach LDS once per run (the empty slots of the
//     neighbour lists repeat one cell thousands of times: 93 % of the level-2 rows of a sparse scan point at cell 0);
//   * at the end the 16 per-wave partials of a cell are added in wave order onto the existing content of `dst` (the
//     reference's `+=` contract) by a single writer.
// Bitwise reproducible; the association order differs from a serial loop's (the oracle's), the sum of course not.
#include "common.h"

namespace {

constexpr int WAVES = 16;            // waves per block (1024 threads)
constexpr int MAXCU = 4;             // channels per lane: C <= 256
constexpr int SEG = 4096;            // index-list rows staged per pass (16 KB of LDS; with the accumulators 80 KB: two blocks per CU)
constexpr int NB = 16;               // source rows in flight per wave
constexpr int ACC_FLOATS = 1024;     // accumulator floats per wave: cells per block = ACC_FLOATS / C (1..32)

struct ScatterP {
    int ncell, c, q, cpb;            // destination cells per sample, channels, source rows per sample, cells per block
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
    extern __shared__ float smem_f[];
    int *cells = reinterpret_cast<int *>(smem_f);                      // [SEG] destination cell of every staged row
    float *acc_all = smem_f + SEG;                                     // [WAVES][cpb][c]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int cpb = p.cpb, C = p.c;
    const int cell0 = blockIdx.x * cpb;
    float *acc = acc_all + (size_t)wave * cpb * C;
    const int cu = (C + 63) >> 6;
    float *dst = p.dst + (size_t)b * p.dst_b;
    const float *src = p.src + (size_t)b * p.src_b;
    for (int i = tid; i < WAVES * cpb * C; i += 64 * WAVES) acc_all[i] = 0.f;
    const size_t ib = (size_t)b * p.q;
    int run_cell = -1;
    float run[MAXCU] = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < p.q; s0 += SEG) {
        const int slen = min(SEG, p.q - s0);
        __syncthreads();                                               // previous pass fully scanned (and acc zeroed)
        {   // the whole block stages the index segment once: all loads of a thread in flight together
            constexpr int PER = SEG / (64 * WAVES);
            long long hv[PER], wv[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int r = tid + u * 64 * WAVES;
                hv[u] = 0; wv[u] = -1;
                if (r < slen) {
                    if constexpr (I64PAIR) { hv[u] = p.h_idx[ib + s0 + r]; wv[u] = p.w_idx[ib + s0 + r]; }
                    else wv[u] = p.idx32[ib + s0 + r];
                }
            }
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int r = tid + u * 64 * WAVES;
                const long long cell = I64PAIR ? hv[u] * p.W + wv[u] : wv[u];
                if (r < slen) cells[r] = (cell >= 0 && cell < p.ncell) ? (int)cell : -1;
            }
        }
        __syncthreads();
        const int nchunk = (slen + 63) >> 6;
        const int c_lo = (int)((long long)nchunk * wave / WAVES), c_hi = (int)((long long)nchunk * (wave + 1) / WAVES);
        for (int ck = c_lo; ck < c_hi; ++ck) {
            const int r0 = ck << 6;
            const int r = r0 + lane;
            int rel = -1;
            if (r < slen) {
                const int d = cells[r] - cell0;
                rel = (cells[r] >= 0 && d >= 0 && d < cpb) ? d : -1;
            }
            unsigned long long mask = __ballot(rel >= 0);
            while (mask) {
                int li[NB], ci[NB], nb = 0;
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    li[t] = 0; ci[t] = 0;
                    if (mask) {
                        li[t] = __builtin_ctzll(mask); mask &= mask - 1;
                        ci[t] = __builtin_amdgcn_readlane(rel, li[t]);
                        nb = t + 1;
                    }
                }
                float v[NB][MAXCU];
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    if (t < nb) {
                        const int row = s0 + r0 + li[t];
                        float wgt = 1.f;
                        if constexpr (WEIGHTED) wgt = p.weight[ib + row];
                        const float *sp = src + (size_t)(row / p.src_div) * p.src_row;
#pragma unroll
                        for (int u = 0; u < MAXCU; ++u) {
                            const int ch = lane + 64 * u;
                            v[t][u] = (u < cu && ch < C) ? sp[(size_t)ch * p.src_ch] : 0.f;
                            if constexpr (WEIGHTED) v[t][u] = __fmul_rn(v[t][u], wgt);
                        }
                    }
                }
                // consecutive matches of one cell are summed in registers (row order) and reach LDS once per run: the
                // empty slots of the neighbour lists repeat one cell thousands of times
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    if (t < nb) {
                        if (ci[t] != run_cell) {
                            if (run_cell >= 0) {
#pragma unroll
                                for (int u = 0; u < MAXCU; ++u) {
                                    const int ch = lane + 64 * u;
                                    if (u < cu && ch < C) acc[run_cell * C + ch] += run[u];
                                }
                            }
                            run_cell = ci[t];
#pragma unroll
                            for (int u = 0; u < MAXCU; ++u) run[u] = v[t][u];
                        } else {
#pragma unroll
                            for (int u = 0; u < MAXCU; ++u) run[u] += v[t][u];
                        }
                    }
                }
            }
        }
        if (run_cell >= 0) {                                           // flush before the LDS index copy is replaced
#pragma unroll
            for (int u = 0; u < MAXCU; ++u) {
                const int ch = lane + 64 * u;
                if (u < cu && ch < C) acc[run_cell * C + ch] += run[u];
            }
            run_cell = -1;
        }
    }
    __syncthreads();
    // single writer per (cell, channel): existing content + the 16 partials in wave order
    for (int i = tid; i < cpb * C; i += 64 * WAVES) {
        const int ci = i / C, ch = i - ci * C;
        if (cell0 + ci < p.ncell) {
            float *d = dst + (size_t)(cell0 + ci) * p.dst_row + (size_t)ch * p.dst_ch;
            float a = *d;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) a += acc_all[((size_t)w * cpb + ci) * C + ch];
            *d = a;
        }
    }
}

template <bool I64PAIR, bool WEIGHTED>
int launch(ScatterP p, int b, hipStream_t st) {
    if (p.c > 64 * MAXCU || p.c <= 0) return I2P_ERR_BAD_ARG;
    int cpb = ACC_FLOATS / p.c; cpb = cpb < 1 ? 1 : (cpb > 32 ? 32 : cpb);
    // small destinations: fewer cells per block so that the chip still sees a few hundred blocks
    const long long want = ((long long)p.ncell * b + 511) / 512;
    if (want < cpb) cpb = want < 1 ? 1 : (int)want;
    p.cpb = cpb;
    const size_t bytes = ((size_t)SEG + (size_t)WAVES * cpb * p.c) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(scatter_det_kernel<I64PAIR, WEIGHTED>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const unsigned gx = (unsigned)((p.ncell + cpb - 1) / cpb);
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

// ---------------------------------------------------------------------------------------------------------------
// Fixed-point variant of the channel-last row scatter-add (what the network's step uses): integer addition is
// associative, so hardware atomics on int64 accumulators give one result whatever order they are served in —
// deterministic at the speed of the atomic kernel.  Three launches on caller scratch:
//   1. m = max |grad_out|                                   (unsigned max of the float bits: order-independent)
//   2. acc[b,cell,ch] += llrint(g * 2^(40 - exponent(m)))    (int64 atomics; equal-cell runs of 8 consecutive rows are
//                                                            merged in registers first, as in the float kernel)
//   3. grad_feat += acc * 2^-(40 - exponent(m))              (one rounding per element)
// Every addend is quantised to 2^-40 of the largest one: the result is within 1e-12 (relative to max|g|) of the exact
// sum, i.e. more accurate than ANY order of float additions.  Non-finite gradients propagate as NaN.
// scratch: acc int64 [B,HW,C] and one uint32 (max bits), BOTH ZEROED BY THE CALLER.
// ---------------------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void absmax_bits_kernel(long long n, const float *__restrict__ x, unsigned *__restrict__ out) {
    unsigned m = 0;
    const long long n4 = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? n >> 2 : 0;       // 16-byte loads, 4 in flight per lane
    const uint4 *x4 = reinterpret_cast<const uint4 *>(x);
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i + u * stride < n4 ? x4[i + u * stride] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            m = max(max(m, v[u].x & 0x7fffffffu), max(max(v[u].y & 0x7fffffffu, v[u].z & 0x7fffffffu), v[u].w & 0x7fffffffu));
    }
    for (long long i = n4 * 4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    // ONE atomic per block, and only if it can raise the maximum: same-address atomics serialise in L2 (~10 ns each —
    // one per wave of 2048 blocks was 80 us of a 90 us launch)
    __shared__ unsigned wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
        if (m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, m);
    }
}

// 2^(shift - exponent(max)); 1 for max == 0.  Every addend is then < 2^(shift+1) in magnitude; `shift` = 40 leaves room for 2^21
// addends per cell in the int64 accumulator, the launcher lowers it for longer scatters (fx_shift)
__device__ __forceinline__ double fx_scale(unsigned maxbits, int shift) {
    const int e = (int)((maxbits >> 23) & 0xff) - 127;               // |max| in [2^e, 2^(e+1))
    return maxbits ? ldexp(1.0, shift - e) : 1.0;
}

constexpr int FX_RUN = 8;
// grad_out rows have pitch `ld` floats, the c scattered channels start at column `off` (ld = c, off = 0: a dense tensor)
__global__ void scatter_fx_kernel(int hw, int c, int q, int W, const float *__restrict__ grad_out, int ld, int off,
                                  const int64_t *__restrict__ h_idx, const int64_t *__restrict__ w_idx,
                                  const unsigned *__restrict__ maxbits, unsigned long long *__restrict__ acc, int shift) {
    const int bi = blockIdx.y;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nrun = (q + FX_RUN - 1) / FX_RUN;
    if (t >= (long long)nrun * c) return;
    const double scale = fx_scale(*maxbits, shift);
    const int run = (int)(t / c), ch = (int)(t % c);
    const int r0 = run * FX_RUN, r1 = min(q, r0 + FX_RUN);
    long long cur = -1, sum = 0;
    for (int row = r0; row < r1; ++row) {
        const long long cell = h_idx[(size_t)bi * q + row] * W + w_idx[(size_t)bi * q + row];
        const long long v = __double2ll_rn((double)grad_out[((size_t)bi * q + row) * ld + off + ch] * scale);
        if (cell != cur) {
            if (cur >= 0 && cur < hw) atomicAdd(acc + ((size_t)bi * hw + cur) * c + ch, (unsigned long long)sum);
            cur = cell; sum = 0;
        }
        sum += v;
    }
    if (cur >= 0 && cur < hw) atomicAdd(acc + ((size_t)bi * hw + cur) * c + ch, (unsigned long long)sum);
}

// Round 6: the hot cell.  Every empty slot of a neighbour list and every slot of an empty query points at cell (0,0) (FLAG_COPY,
// go.cu:211-222): 93 % of the level-2 rows of a sparse scan.  scatter_fx_kernel merges equal-cell runs of 8 consecutive rows per
// thread and then issues one global atomic per (run, channel): 1808 same-address atomics per channel and sample at level 2, which
// serialise in L2 (102 us for a 3.7 M-element scatter; 40 us when the same rows point at random cells).  Here the block keeps ONE
// accumulator row for cell 0 in LDS (ds_add_u64), walks ITER x 256 (run, channel) pairs, and issues one global atomic per channel
// for it at the end: 1808 -> 57 per channel and sample at level 2.  Everything else is scatter_fx_kernel: integer sums, so the
// result is bit-identical.  Two more general forms were built and measured first (profiles/r06_scatter.txt): a block-shared
// open-addressing table with atomicCAS probing — 47 us hot, but 62 us on random cells (a CAS round trip per run, a wave walking rows
// serially) — and a wave-private direct-mapped cache — 64 / 54 us.  The hot cell is the only repeat that matters.
template <int ITER>
__global__ __launch_bounds__(256) void scatter_fx_hot_kernel(int hw, int c, int q, int W, const float *__restrict__ grad_out, int ld, int off,
                                                             const int64_t *__restrict__ h_idx, const int64_t *__restrict__ w_idx,
                                                             const unsigned *__restrict__ maxbits, unsigned long long *__restrict__ acc, int shift) {
    extern __shared__ unsigned long long hot[];            // [c] sums for cell 0 of this sample
    const int bi = blockIdx.y;
    for (int i = threadIdx.x; i < c; i += 256) hot[i] = 0ull;
    __syncthreads();
    const int nrun = (q + FX_RUN - 1) / FX_RUN;
    const double scale = fx_scale(*maxbits, shift);
    unsigned long long *accb = acc + (size_t)bi * hw * c;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const long long t = ((long long)blockIdx.x * ITER + it) * 256 + threadIdx.x;
        if (t >= (long long)nrun * c) break;
        const int run = (int)(t / c), ch = (int)(t % c);
        const int r0 = run * FX_RUN, r1 = min(q, r0 + FX_RUN);
        long long cur = -1, sum = 0;
        auto flush = [&]() {
            if (cur == 0) { if (sum != 0) atomicAdd(hot + ch, (unsigned long long)sum); }
            else if (cur > 0 && cur < hw) atomicAdd(accb + (size_t)cur * c + ch, (unsigned long long)sum);
        };
        for (int row = r0; row < r1; ++row) {
            const long long cell = h_idx[(size_t)bi * q + row] * W + w_idx[(size_t)bi * q + row];
            const long long v = __double2ll_rn((double)grad_out[((size_t)bi * q + row) * ld + off + ch] * scale);
            if (cell != cur) { flush(); cur = cell; sum = 0; }
            sum += v;
        }
        flush();
    }
    __syncthreads();
    if (hw > 0)
        for (int i = threadIdx.x; i < c; i += 256) { const unsigned long long v = hot[i]; if (v != 0ull) atomicAdd(accb + i, v); }
}

__global__ __launch_bounds__(256) void fx_finalize_kernel(long long n, const long long *__restrict__ acc, const unsigned *__restrict__ maxbits,
                                                           float *__restrict__ dst, int shift) {
    const unsigned mb = *maxbits;
    const double inv = 1.0 / fx_scale(mb, shift);
    const bool bad = mb >= 0x7f800000u;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        dst[i] = bad ? __uint_as_float(0x7fc00000u) : dst[i] + (float)((double)acc[i] * inv);
}

}  // namespace

extern "C" long long i2p_gather_rows_grad_fx_scratch(int b, int hw, int c) { return ((long long)b * hw * c + 1) * 8; }   // bytes

static int gather_rows_grad_fx_impl(int b, int hw, int c, int q, int W, const float *grad_out, int ld, int off, const int64_t *h_idx,
                                    const int64_t *w_idx, void *scratch, float *grad_feat, void *stream) {
    if (b < 0 || hw < 0 || c < 0 || q < 0 || W <= 0 || off < 0 || ld < off + c) return I2P_ERR_BAD_ARG;
    if ((long long)b * q * c == 0) return 0;
    if (!grad_out || !h_idx || !w_idx || !grad_feat || !scratch) return I2P_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(scratch);
    unsigned *mx = reinterpret_cast<unsigned *>(acc + (size_t)b * hw * c);
    // (strided source: the maximum is taken over the WHOLE rows — a superset of the scattered columns, still a valid scale)
    const long long nsrc = (long long)b * q * ld, ndst = (long long)b * hw * c;
    long long g1 = (nsrc + 256 * 16 - 1) / (256 * 16); if (g1 > 1024) g1 = 1024; if (g1 < 1) g1 = 1;
    hipLaunchKernelGGL(absmax_bits_kernel, dim3((unsigned)g1), dim3(256), 0, st, nsrc, grad_out, mx);
    const long long tot = (long long)((q + FX_RUN - 1) / FX_RUN) * c;
    // all q rows of a sample may land on ONE cell (every empty-centre query gathers cell (0,0)): |sum| < q * 2^(shift+1) must stay
    // below 2^62 -> shift = min(40, 61 - ceil(log2 q)); q <= 2^21 keeps the full 40 bits
    int lg = 0;
    while ((1LL << lg) < (long long)q) ++lg;
    const int shift = lg > 21 ? 61 - lg : 40;
    const char *v1 = getenv("I2P_SCATTER_V1");            // A/B: the kernel of rounds 2-5 without the hot-cell accumulator (read per call: tests toggle it)
    if (c <= 4096 && !(v1 && v1[0] == '1')) {
        constexpr int ITER = 4;
        hipLaunchKernelGGL((scatter_fx_hot_kernel<ITER>), dim3((unsigned)((tot + 256 * ITER - 1) / (256 * ITER)), b), dim3(256), (size_t)c * 8, st, hw, c, q, W,
                           grad_out, ld, off, h_idx, w_idx, mx, acc, shift);
    } else
        hipLaunchKernelGGL(scatter_fx_kernel, dim3((unsigned)((tot + 255) / 256), b), dim3(256), 0, st, hw, c, q, W, grad_out, ld, off, h_idx, w_idx, mx, acc,
                           shift);
    long long g3 = (ndst + 255) / 256; if (g3 > 2048) g3 = 2048;
    hipLaunchKernelGGL(fx_finalize_kernel, dim3((unsigned)g3), dim3(256), 0, st, ndst, reinterpret_cast<const long long *>(acc), mx, grad_feat, shift);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_gather_rows_grad_fx(int b, int hw, int c, int q, int W, const float *grad_out, const int64_t *h_idx,
                                       const int64_t *w_idx, void *scratch, float *grad_feat, void *stream) {
    return gather_rows_grad_fx_impl(b, hw, c, q, W, grad_out, c, 0, h_idx, w_idx, scratch, grad_feat, stream);
}

extern "C" int i2p_gather_rows_grad_fx_ld(int b, int hw, int c, int q, int W, const float *grad_out, int ld, int off,
                                          const int64_t *h_idx, const int64_t *w_idx, void *scratch, float *grad_feat, void *stream) {
    return gather_rows_grad_fx_impl(b, hw, c, q, W, grad_out, ld, off, h_idx, w_idx, scratch, grad_feat, stream);
}
