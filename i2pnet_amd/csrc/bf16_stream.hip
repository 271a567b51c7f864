// HBM-streaming kernels on bf16-stored [rows, C] tensors: the bf16 twins of bn_act.hip / cv_softmax.hip
// (see bf16_common.h for the storage convention).  16 bytes (8 channels) per lane along C, grid-stride over rows.
#include "bf16_common.h"

namespace {

constexpr int THREADS = 256;
constexpr int REP = I2P_BN_REPLICAS;
constexpr int MAX_BLOCKS = 2048;

struct G8 { int cv, rpb; };                       // 8-channel chunks per row, rows per block iteration
inline bool ok8(int c) { return c > 0 && (c & 7) == 0 && (c >> 3) <= THREADS && THREADS % (c >> 3) == 0; }
inline G8 geom8(int c) { G8 g; g.cv = c >> 3; g.rpb = THREADS / g.cv; return g; }
inline unsigned grid_rows(long long rows, int rpb, int cap) {
    long long b = (rows + rpb - 1) / rpb;
    return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

__device__ __forceinline__ void load_ab(const float *coef, int c, int ch0, float (&a)[8], float (&b)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) { a[q] = coef[c + ch0 + q]; b[q] = coef[2 * c + ch0 + q] - coef[ch0 + q] * a[q]; }
}

// block reduction of 16 per-thread doubles (8 channels x {s, q}) over the threads that own the same chunk column,
// then one replica-spread atomic per channel and block
__device__ __forceinline__ void reduce16(const double (&s)[8], const double (&q)[8], int cv, int c, double *sums) {
    __shared__ double red[THREADS][16];
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[threadIdx.x][i] = s[i]; red[threadIdx.x][8 + i] = q[i]; }
    __syncthreads();
    if (threadIdx.x < cv) {
        double a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = 0.0;
        for (int t = threadIdx.x; t < THREADS; t += cv)
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] += red[t][i];
        double *rep = sums + (size_t)(blockIdx.x % REP) * 2 * c;
#pragma unroll
        for (int i = 0; i < 8; ++i) { atomicAdd(rep + threadIdx.x * 8 + i, a[i]); atomicAdd(rep + c + threadIdx.x * 8 + i, a[8 + i]); }
    }
}

// reduce16 for a thread that owns channels [4 v, 4 v + 4) and [c/2 + 4 v, c/2 + 4 v + 4) (the fp32 outer sum: each of its two 16-byte
// stores then covers a contiguous 128-byte line per 8 lanes instead of every other 16 bytes)
__device__ __forceinline__ void reduce16_split(const double (&s)[8], const double (&q)[8], int cv, int c, double *sums) {
    __shared__ double red[THREADS][16];
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[threadIdx.x][i] = s[i]; red[threadIdx.x][8 + i] = q[i]; }
    __syncthreads();
    if (threadIdx.x < cv) {
        double a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = 0.0;
        for (int t = threadIdx.x; t < THREADS; t += cv)
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] += red[t][i];
        double *rep = sums + (size_t)(blockIdx.x % REP) * 2 * c;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ch = (i < 4 ? 0 : c / 2 - 4) + threadIdx.x * 4 + i;
            atomicAdd(rep + ch, a[i]); atomicAdd(rep + c + ch, a[8 + i]);
        }
    }
}

// ye[b,n,k,:] = enc_n[b,n,:] + enc_k[b,k,:] (stored as bf16, or as f32 with F32) and the BN statistics of the stored values
template <bool F32>
__global__ __launch_bounds__(THREADS) void outer_sum_kernel(long long rows, int N, int M, int c, G8 g, const float *__restrict__ en,
                                                             const float *__restrict__ ek, void *__restrict__ ye_,
                                                             double *__restrict__ sums) {
    const int vcol = threadIdx.x % g.cv, rsub = threadIdx.x / g.cv;
    double s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] = 0.0; q[i] = 0.0; }
    for (long long r = (long long)blockIdx.x * g.rpb + rsub; r < rows; r += (long long)gridDim.x * g.rpb) {
        // (b, n, k) of the row with 32-bit divisions (launcher: rows < 2^31); two 64-bit ones per row were a third of this kernel
        const unsigned r32 = (unsigned)r, bn = r32 / (unsigned)M, k = r32 - bn * (unsigned)M, b = bn / (unsigned)N;
        // F32: the lane's second four channels sit half a row away (see reduce16_split)
        const int c_lo = F32 ? vcol * 4 : vcol * 8, c_hi = F32 ? c / 2 + vcol * 4 : vcol * 8 + 4;
        const float *pn = en + (size_t)bn * c, *pk = ek + ((size_t)b * M + k) * c;
        const float4 a0 = *reinterpret_cast<const float4 *>(pn + c_lo), a1 = *reinterpret_cast<const float4 *>(pn + c_hi);
        const float4 b0 = *reinterpret_cast<const float4 *>(pk + c_lo), b1 = *reinterpret_cast<const float4 *>(pk + c_hi);
        const float f[8] = {a0.x + b0.x, a0.y + b0.y, a0.z + b0.z, a0.w + b0.w, a1.x + b1.x, a1.y + b1.y, a1.z + b1.z, a1.w + b1.w};
        if constexpr (F32) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { s[i] += f[i]; q[i] += (double)f[i] * f[i]; }
            // two 16-byte streaming stores per lane, each a contiguous 128-byte line per 8 lanes (eight 4-byte stores: 118 us for the
            // 218 MB tensor = 1.85 TB/s; two 16-byte stores 32 bytes apart: 102 us)
            typedef float f32x4_nt __attribute__((ext_vector_type(4)));
            float *o = reinterpret_cast<float *>(ye_) + (size_t)r * c;
            __builtin_nontemporal_store(f32x4_nt{f[0], f[1], f[2], f[3]}, reinterpret_cast<f32x4_nt *>(o + c_lo));
            __builtin_nontemporal_store(f32x4_nt{f[4], f[5], f[6], f[7]}, reinterpret_cast<f32x4_nt *>(o + c_hi));
        } else {
            const uint4 o = bf_pack8(f);
            float fr[8]; bf_unpack8(o, fr);
#pragma unroll
            for (int i = 0; i < 8; ++i) { s[i] += fr[i]; q[i] += (double)fr[i] * fr[i]; }
            st_u4_stream(reinterpret_cast<bf16_t *>(ye_) + (size_t)r * c + vcol * 8, o);
        }
    }
    if constexpr (F32) reduce16_split(s, q, g.cv, c, sums); else reduce16(s, q, g.cv, c, sums);
}

__global__ __launch_bounds__(THREADS) void to_bf16_kernel(long long n8, const float4 *__restrict__ x, uint4 *__restrict__ y) {
    for (long long i = (long long)blockIdx.x * THREADS + threadIdx.x; i < n8; i += (long long)gridDim.x * THREADS) {
        const float4 a = x[2 * i], b = x[2 * i + 1];
        const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        y[i] = bf_pack8(f);
    }
}

__global__ __launch_bounds__(THREADS) void bn_act_fwd_bf16_kernel(long long rows, int c, G8 g, const bf16_t *__restrict__ y,
                                                                   const float *__restrict__ coef, float slope, float *__restrict__ out) {
    const int vcol = threadIdx.x % g.cv, rsub = threadIdx.x / g.cv;
    float a[8], b[8]; load_ab(coef, c, vcol * 8, a, b);
    for (long long r = (long long)blockIdx.x * g.rpb + rsub; r < rows; r += (long long)gridDim.x * g.rpb) {
        float f[8]; bf_unpack8(ld_u4_stream(y + (size_t)r * c + vcol * 8), f);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = bf_act(bf_bnz(f[i], a[i], b[i]), slope);
        float *o = out + (size_t)r * c + vcol * 8;
        *reinterpret_cast<float4 *>(o) = make_float4(f[0], f[1], f[2], f[3]);
        *reinterpret_cast<float4 *>(o + 4) = make_float4(f[4], f[5], f[6], f[7]);
    }
}

// out[g,:] = max_k act(bn(y[g*K+k,:])), arg = first k attaining it (PPBackbone_center.py:129)
__global__ __launch_bounds__(THREADS) void maxk_bf16_kernel(long long groups, int K, int c, const bf16_t *__restrict__ y,
                                                             const float *__restrict__ coef, float slope, float *__restrict__ out,
                                                             unsigned char *__restrict__ arg) {
    const int cv = c >> 3;
    const long long total = groups * cv;
    const int vcol = threadIdx.x % cv;
    float a[8], b[8]; load_ab(coef, c, vcol * 8, a, b);
    for (long long t = (long long)blockIdx.x * THREADS + threadIdx.x; t < total; t += (long long)gridDim.x * THREADS) {
        const long long grp = t / cv;
        float best[8]; unsigned char bi[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { best[i] = -INFINITY; bi[i] = 0; }
        const bf16_t *src = y + (size_t)grp * K * c + vcol * 8;
        for (int k = 0; k < K; ++k) {
            float f[8]; bf_unpack8(*reinterpret_cast<const uint4 *>(src + (size_t)k * c), f);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float v = bf_act(bf_bnz(f[i], a[i], b[i]), slope);
                if (v > best[i] || v != v) { best[i] = v; bi[i] = (unsigned char)k; }
            }
        }
        float *o = out + (size_t)grp * c + vcol * 8;
        *reinterpret_cast<float4 *>(o) = make_float4(best[0], best[1], best[2], best[3]);
        *reinterpret_cast<float4 *>(o + 4) = make_float4(best[4], best[5], best[6], best[7]);
        const unsigned lo = bi[0] | (bi[1] << 8) | (bi[2] << 16) | ((unsigned)bi[3] << 24);
        const unsigned hi = bi[4] | (bi[5] << 8) | (bi[6] << 16) | ((unsigned)bi[7] << 24);
        *reinterpret_cast<uint2 *>(arg + (size_t)grp * c + vcol * 8) = make_uint2(lo, hi);
    }
}

__global__ __launch_bounds__(THREADS) void unpool_bf16_kernel(long long groups, int K, int c, const float *__restrict__ g,
                                                               const unsigned char *__restrict__ arg, bf16_t *__restrict__ gd) {
    const int cv = c >> 3;
    const long long total = groups * K * cv;
    for (long long t = (long long)blockIdx.x * THREADS + threadIdx.x; t < total; t += (long long)gridDim.x * THREADS) {
        const long long r = t / cv;
        const int vcol = (int)(t - r * cv);
        const long long grp = r / K;
        const unsigned k = (unsigned)(r - grp * K);
        const float *gp = g + (size_t)grp * c + vcol * 8;
        const float4 g0 = *reinterpret_cast<const float4 *>(gp), g1 = *reinterpret_cast<const float4 *>(gp + 4);
        const uint2 a = *reinterpret_cast<const uint2 *>(arg + (size_t)grp * c + vcol * 8);
        float f[8];
        f[0] = (a.x & 255u) == k ? g0.x : 0.f; f[1] = ((a.x >> 8) & 255u) == k ? g0.y : 0.f;
        f[2] = ((a.x >> 16) & 255u) == k ? g0.z : 0.f; f[3] = (a.x >> 24) == k ? g0.w : 0.f;
        f[4] = (a.y & 255u) == k ? g1.x : 0.f; f[5] = ((a.y >> 8) & 255u) == k ? g1.y : 0.f;
        f[6] = ((a.y >> 16) & 255u) == k ? g1.z : 0.f; f[7] = (a.y >> 24) == k ? g1.w : 0.f;
        st_u4_stream(gd + (size_t)t * 8, bf_pack8(f));
    }
}

// replicated {sum dz, sum dz*xhat}, dz = dout * act'(bn(y))
__global__ __launch_bounds__(THREADS) void bwd_stats_bf16_kernel(long long rows, int c, G8 g, const bf16_t *__restrict__ dout,
                                                                  const bf16_t *__restrict__ y, const float *__restrict__ coef,
                                                                  const float *__restrict__ mi, float slope, double *__restrict__ dsums) {
    const int vcol = threadIdx.x % g.cv, rsub = threadIdx.x / g.cv;
    float a[8], b[8], xp[8], xq[8];
    load_ab(coef, c, vcol * 8, a, b);
#pragma unroll
    for (int i = 0; i < 8; ++i) { xp[i] = mi[c + vcol * 8 + i]; xq[i] = -mi[vcol * 8 + i] * xp[i]; }
    double s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] = 0.0; q[i] = 0.0; }
    const long long stride = (long long)gridDim.x * g.rpb;
    for (long long r0 = (long long)blockIdx.x * g.rpb + rsub; r0 < rows; r0 += 2 * stride) {
        uint4 dv[2], yv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long r = r0 + u * stride;
            const bool ok = r < rows;
            dv[u] = ok ? ld_u4_stream(dout + (size_t)r * c + vcol * 8) : make_uint4(0, 0, 0, 0);
            yv[u] = ok ? ld_u4_stream(y + (size_t)r * c + vcol * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float d[8], f[8]; bf_unpack8(dv[u], d); bf_unpack8(yv[u], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float dz = bf_bnz(f[i], a[i], b[i]) > 0.f ? d[i] : d[i] * slope;
                s[i] += dz; q[i] += (double)dz * __builtin_fmaf(f[i], xp[i], xq[i]);
            }
        }
    }
    reduce16(s, q, g.cv, c, dsums);
}

// ---- softmax-over-pixels weighted sum (cv_softmax.hip) on bf16 y5 / y3: thread = (8-channel chunk, k phase) ----------
struct SmP {
    int B, N, M, C;
    const bf16_t *y5, *y3; const float *coef5, *mi5, *coef3; float slope5, slope3;
    const float *g_out; float *out, *msave; bf16_t *gz5, *ga3; double *dsums5;
};

__global__ __launch_bounds__(THREADS) void sm_fwd_bf16_kernel(SmP p) {
    __shared__ float sm[THREADS][8], ss[THREADS][8], sa[THREADS][8];
    const int bn = blockIdx.x, tid = threadIdx.x, cv = p.C >> 3;
    const int vc = tid % cv, kg = tid / cv, KG = THREADS / cv;
    float a5[8], b5[8], a3[8], b3[8];
    load_ab(p.coef5, p.C, vc * 8, a5, b5); load_ab(p.coef3, p.C, vc * 8, a3, b3);
    const size_t base = (size_t)bn * p.M * p.C + vc * 8;
    float mx[8], S[8], A[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { mx[i] = -INFINITY; S[i] = 0.f; A[i] = 0.f; }
    for (int k = kg; k < p.M; k += KG) {
        float f5[8], f3[8];
        bf_unpack8(*reinterpret_cast<const uint4 *>(p.y5 + base + (size_t)k * p.C), f5);
        bf_unpack8(*reinterpret_cast<const uint4 *>(p.y3 + base + (size_t)k * p.C), f3);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float h5 = bf_act(bf_bnz(f5[i], a5[i], b5[i]), p.slope5), h3 = bf_act(bf_bnz(f3[i], a3[i], b3[i]), p.slope3);
            const float mn = fmaxf(mx[i], h5);
            const float sc = __expf(mx[i] - mn), e = __expf(h5 - mn);
            S[i] = S[i] * sc + e; A[i] = A[i] * sc + e * h3; mx[i] = mn;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { sm[tid][i] = mx[i]; ss[tid][i] = S[i]; sa[tid][i] = A[i]; }
    __syncthreads();
    if (tid < p.C) {                     // thread = channel
        const int c = tid, v = c >> 3, i = c & 7;
        float M_ = -INFINITY;
        for (int g = 0; g < KG; ++g) M_ = fmaxf(M_, sm[g * cv + v][i]);
        float St = 0.f, At = 0.f;
        for (int g = 0; g < KG; ++g) {
            const float w = __expf(sm[g * cv + v][i] - M_);
            St += ss[g * cv + v][i] * w; At += sa[g * cv + v][i] * w;
        }
        p.out[(size_t)bn * p.C + c] = At / St;
        p.msave[(size_t)bn * 2 * p.C + c] = M_; p.msave[(size_t)bn * 2 * p.C + p.C + c] = St;
    }
}

__global__ __launch_bounds__(THREADS) void sm_bwd_bf16_kernel(SmP p) {
    const int bn = blockIdx.x, tid = threadIdx.x, cv = p.C >> 3;
    const int vc = tid % cv, kg = tid / cv, KG = THREADS / cv;
    float a5[8], b5[8], a3[8], b3[8], xp[8], xq[8], M_[8], inv[8], go[8], o[8];
    load_ab(p.coef5, p.C, vc * 8, a5, b5); load_ab(p.coef3, p.C, vc * 8, a3, b3);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = vc * 8 + i;
        xp[i] = p.mi5[p.C + c]; xq[i] = -p.mi5[c] * xp[i];
        M_[i] = p.msave[(size_t)bn * 2 * p.C + c]; inv[i] = 1.f / p.msave[(size_t)bn * 2 * p.C + p.C + c];
        go[i] = p.g_out[(size_t)bn * p.C + c]; o[i] = p.out[(size_t)bn * p.C + c];
    }
    const size_t base = (size_t)bn * p.M * p.C + vc * 8;
    double ds[8], dq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { ds[i] = 0.0; dq[i] = 0.0; }
    for (int k = kg; k < p.M; k += KG) {
        float f5[8], f3[8], g5[8], g3[8];
        bf_unpack8(*reinterpret_cast<const uint4 *>(p.y5 + base + (size_t)k * p.C), f5);
        bf_unpack8(*reinterpret_cast<const uint4 *>(p.y3 + base + (size_t)k * p.C), f3);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float z5 = bf_bnz(f5[i], a5[i], b5[i]);
            const float h5 = bf_act(z5, p.slope5), h3 = bf_act(bf_bnz(f3[i], a3[i], b3[i]), p.slope3);
            const float s = __expf(h5 - M_[i]) * inv[i];
            float t = s * go[i] * (h3 - o[i]);
            t = z5 > 0.f ? t : t * p.slope5;
            g5[i] = t; g3[i] = go[i] * s;
        }
        const uint4 o5 = bf_pack8(g5);
        st_u4_stream(p.gz5 + base + (size_t)k * p.C, o5);
        st_u4_stream(p.ga3 + base + (size_t)k * p.C, bf_pack8(g3));
        float r5[8]; bf_unpack8(o5, r5);                 // statistics of the stored (rounded) gradient
#pragma unroll
        for (int i = 0; i < 8; ++i) { ds[i] += r5[i]; dq[i] += (double)r5[i] * __builtin_fmaf(f5[i], xp[i], xq[i]); }
    }
    reduce16(ds, dq, cv, p.C, p.dsums5);
}

// k- and n-sums of dL/dz_e (bf16) for the position-encoding factors (cv_softmax.hip: pair_sum_kernel).  Two streaming
// passes, each with a single writer per output element (no atomics, fixed order): the tensor is 218 MB at batch 16 and
// the second pass largely hits the 256 MB Infinity Cache.  (One pass with fp32 atomics for both sums: 423 us.)
//   sum_n[b,k,:] = sum over points n:  thread = (b, k, 8-channel chunk), walks n with 8 loads in flight
__global__ __launch_bounds__(THREADS) void pair_sum_n_bf16_kernel(int B, int N, int M, int C, const bf16_t *__restrict__ g,
                                                                   float *__restrict__ sum_n) {
    const int cv = C >> 3;
    const long long t = (long long)blockIdx.x * THREADS + threadIdx.x;
    if (t >= (long long)B * M * cv) return;
    const int c8 = (int)(t % cv); const long long bk = t / cv;
    const int k = (int)(bk % M), b = (int)(bk / M);
    const bf16_t *src = g + (((size_t)b * N) * M + k) * C + c8 * 8;
    const size_t nstride = (size_t)M * C;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int n = 0;
    for (; n + 8 <= N; n += 8) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ld_u4_stream(src + (size_t)(n + u) * nstride);
#pragma unroll
        for (int u = 0; u < 8; ++u) { float f[8]; bf_unpack8(v[u], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += f[i]; }
    }
    for (; n < N; ++n) { float f[8]; bf_unpack8(ld_u4_stream(src + (size_t)n * nstride), f);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += f[i]; }
    float *o = sum_n + ((size_t)b * M + k) * C + c8 * 8;
    *reinterpret_cast<float4 *>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4 *>(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
}

//   sum_k[b,n,:] = sum over pixels k:  block = one point (b,n), thread = (k phase, 8-channel chunk), block reduction
__global__ __launch_bounds__(THREADS) void pair_sum_k_bf16_kernel(int M, int C, const bf16_t *__restrict__ g, float *__restrict__ sum_k) {
    __shared__ float red[THREADS][8];
    const int cv = C >> 3, ks = THREADS / cv;
    const int c8 = threadIdx.x % cv, kslot = threadIdx.x / cv;
    const bf16_t *src = g + (size_t)blockIdx.x * M * C + c8 * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int k = kslot;
    for (; k + 3 * ks < M; k += 4 * ks) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4 *>(src + (size_t)(k + u * ks) * C);
#pragma unroll
        for (int u = 0; u < 4; ++u) { float f[8]; bf_unpack8(v[u], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += f[i]; }
    }
    for (; k < M; k += ks) { float f[8]; bf_unpack8(*reinterpret_cast<const uint4 *>(src + (size_t)k * C), f);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += f[i]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[threadIdx.x][i] = acc[i];
    __syncthreads();
    if (threadIdx.x < C) {
        const int c = threadIdx.x;
        float a = 0.f;
        for (int q = 0; q < ks; ++q) a += red[q * cv + (c >> 3)][c & 7];
        sum_k[(size_t)blockIdx.x * C + c] = a;
    }
}

}  // namespace

extern "C" int i2p_outer_sum_bf16(int B, int N, int M, int C, const float *enc_n, const float *enc_k, bf16_t *ye, double *sums,
                                  void *stream) {
    if (B <= 0 || N <= 0 || M <= 0 || !ok8(C) || !enc_n || !enc_k || !ye || !sums) return I2P_ERR_BAD_ARG;
    const long long rows = (long long)B * N * M;
    if (rows >= (1LL << 31)) return I2P_ERR_BAD_ARG;
    const G8 g = geom8(C);
    hipLaunchKernelGGL(outer_sum_kernel<false>, dim3(grid_rows(rows, g.rpb, 1024)), dim3(THREADS), 0, (hipStream_t)stream, rows, N, M, C, g,
                       enc_n, enc_k, (void *)ye, sums);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_outer_sum(int B, int N, int M, int C, const float *enc_n, const float *enc_k, float *ye, double *sums, void *stream) {
    if (B <= 0 || N <= 0 || M <= 0 || !ok8(C) || !enc_n || !enc_k || !ye || !sums) return I2P_ERR_BAD_ARG;
    const long long rows = (long long)B * N * M;
    if (rows >= (1LL << 31)) return I2P_ERR_BAD_ARG;
    const G8 g = geom8(C);
    hipLaunchKernelGGL(outer_sum_kernel<true>, dim3(grid_rows(rows, g.rpb, 1024)), dim3(THREADS), 0, (hipStream_t)stream, rows, N, M, C, g,
                       enc_n, enc_k, (void *)ye, sums);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_to_bf16(long long n, const float *x, bf16_t *y, void *stream) {
    if (n < 0 || (n & 7)) return I2P_ERR_BAD_ARG;
    if (n == 0) return 0;
    if (!x || !y) return I2P_ERR_BAD_ARG;
    const long long n8 = n >> 3;
    long long blocks = (n8 + THREADS - 1) / THREADS;
    if (blocks > MAX_BLOCKS) blocks = MAX_BLOCKS;
    hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)blocks), dim3(THREADS), 0, (hipStream_t)stream, n8, (const float4 *)x, (uint4 *)y);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_bn_act_fwd_bf16(long long rows, int c, const bf16_t *y, const float *coef, float slope, float *out, void *stream) {
    if (rows < 0 || !ok8(c)) return I2P_ERR_BAD_ARG;
    if (rows == 0) return 0;
    if (!y || !coef || !out) return I2P_ERR_BAD_ARG;
    const G8 g = geom8(c);
    hipLaunchKernelGGL(bn_act_fwd_bf16_kernel, dim3(grid_rows(rows, g.rpb, MAX_BLOCKS)), dim3(THREADS), 0, (hipStream_t)stream, rows, c, g,
                       y, coef, slope, out);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_bn_act_maxk_fwd_bf16(long long groups, int K, int c, const bf16_t *y, const float *coef, float slope, float *out,
                                        unsigned char *arg, void *stream) {
    if (groups < 0 || K <= 0 || K > 255 || !ok8(c)) return I2P_ERR_BAD_ARG;
    if (groups == 0) return 0;
    if (!y || !coef || !out || !arg) return I2P_ERR_BAD_ARG;
    const long long total = groups * (c >> 3);
    long long blocks = (total + THREADS - 1) / THREADS;
    if (blocks > (1 << 20)) blocks = 1 << 20;
    hipLaunchKernelGGL(maxk_bf16_kernel, dim3((unsigned)blocks), dim3(THREADS), 0, (hipStream_t)stream, groups, K, c, y, coef, slope, out, arg);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_unpool_k_bf16(long long groups, int K, int c, const float *g, const unsigned char *arg, bf16_t *gd, void *stream) {
    if (groups < 0 || K <= 0 || K > 255 || c <= 0 || (c & 7)) return I2P_ERR_BAD_ARG;
    if (groups == 0) return 0;
    if (!g || !arg || !gd) return I2P_ERR_BAD_ARG;
    const long long total = groups * K * (c >> 3);
    long long blocks = (total + THREADS - 1) / THREADS;
    if (blocks > (1 << 20)) blocks = 1 << 20;
    hipLaunchKernelGGL(unpool_bf16_kernel, dim3((unsigned)blocks), dim3(THREADS), 0, (hipStream_t)stream, groups, K, c, g, arg, gd);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_bn_act_bwd_stats_bf16(long long rows, int c, const bf16_t *dout, const bf16_t *y, const float *coef, const float *mi,
                                         float slope, double *dsums, void *stream) {
    if (rows < 0 || !ok8(c)) return I2P_ERR_BAD_ARG;
    if (rows == 0) return 0;
    if (!dout || !y || !coef || !mi || !dsums) return I2P_ERR_BAD_ARG;
    const G8 g = geom8(c);
    hipLaunchKernelGGL(bwd_stats_bf16_kernel, dim3(grid_rows(rows, 2 * g.rpb, 1024)), dim3(THREADS), 0, (hipStream_t)stream, rows, c, g,
                       dout, y, coef, mi, slope, dsums);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_cv_softmax_wsum_fwd_bf16(int B, int N, int M, int C, const bf16_t *y5, const float *coef5, float slope5,
                                            const bf16_t *y3, const float *coef3, float slope3, float *out, float *msave, void *stream) {
    if (B <= 0 || N <= 0 || M <= 0 || !ok8(C) || C > THREADS) return I2P_ERR_BAD_ARG;
    if (!y5 || !coef5 || !y3 || !coef3 || !out || !msave) return I2P_ERR_BAD_ARG;
    SmP p{}; p.B = B; p.N = N; p.M = M; p.C = C; p.y5 = y5; p.coef5 = coef5; p.slope5 = slope5; p.y3 = y3; p.coef3 = coef3;
    p.slope3 = slope3; p.out = out; p.msave = msave;
    hipLaunchKernelGGL(sm_fwd_bf16_kernel, dim3(B * N), dim3(THREADS), 0, (hipStream_t)stream, p);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_cv_softmax_wsum_bwd_bf16(int B, int N, int M, int C, const float *g_out, const float *out, const float *msave,
                                            const bf16_t *y5, const float *coef5, const float *mi5, float slope5, const bf16_t *y3,
                                            const float *coef3, float slope3, bf16_t *gz5, double *dsums5, bf16_t *ga3, void *stream) {
    if (B <= 0 || N <= 0 || M <= 0 || !ok8(C) || C > THREADS) return I2P_ERR_BAD_ARG;
    if (!g_out || !out || !msave || !y5 || !coef5 || !mi5 || !y3 || !coef3 || !gz5 || !dsums5 || !ga3) return I2P_ERR_BAD_ARG;
    SmP p{}; p.B = B; p.N = N; p.M = M; p.C = C; p.y5 = y5; p.coef5 = coef5; p.mi5 = mi5; p.slope5 = slope5; p.y3 = y3;
    p.coef3 = coef3; p.slope3 = slope3; p.g_out = g_out; p.out = const_cast<float *>(out); p.msave = const_cast<float *>(msave);
    p.gz5 = gz5; p.ga3 = ga3; p.dsums5 = dsums5;
    hipLaunchKernelGGL(sm_bwd_bf16_kernel, dim3(B * N), dim3(THREADS), 0, (hipStream_t)stream, p);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_pair_bias_bn_bwd_bf16(int B, int N, int M, int C, const bf16_t *gz, const float *enc_n, const float *enc_k,
                                         const double *dsums, const float *coef, const float *mi, float *sum_k, float *sum_n,
                                         float *d_enc_n, float *d_enc_k, void *stream) {
    if (B <= 0 || N <= 0 || M <= 0 || !ok8(C) || C > 256 || THREADS % C) return I2P_ERR_BAD_ARG;
    if (!gz || !enc_n || !enc_k || !dsums || !coef || !mi || !sum_k || !sum_n || !d_enc_n || !d_enc_k) return I2P_ERR_BAD_ARG;
    const long long tn = (long long)B * M * (C >> 3);
    hipLaunchKernelGGL(pair_sum_n_bf16_kernel, dim3((unsigned)((tn + THREADS - 1) / THREADS)), dim3(THREADS), 0, (hipStream_t)stream, B, N, M, C, gz, sum_n);
    hipLaunchKernelGGL(pair_sum_k_bf16_kernel, dim3(B * N), dim3(THREADS), 0, (hipStream_t)stream, M, C, gz, sum_k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    return i2p_pair_bias_bn_finish(B, N, M, C, sum_k, sum_n, enc_n, enc_k, dsums, coef, mi, d_enc_n, d_enc_k, stream);
}
