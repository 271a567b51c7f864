// Image-encoder block tail for gfx950: BatchNorm2d (batch statistics, running buffers updated) +
// LeakyReLU(0.1) + MaxPool2d(3, stride, padding 1) on the NHWC output of a 3x3 convolution
// (reference: src/modules/basicConv.py:6-20, 15 such blocks in front of the cost volumes).
//
// PyTorch/MIOpen runs this as 3 BN kernels + LeakyReLU + pooling forward (4 full-tensor reads, 2 full
// writes + the pooled write) and 3 BN-backward kernels + LeakyReLU backward + pooling backward
// (≈ 9 full-tensor passes); the largest tensor is 238 MB at batch 8.  Here:
//   forward : bn_stats (1 read; csrc/bn_act.hip)  +  img_pool_fwd (1 read, pooled write + 1-byte arg-max)
//   backward: img_bwd_stats (pooled-size reads + arg-max gathers)  +  img_bwd_dx (1 read, 1 write)
// The conv output is never normalised/activated in memory; the backward recomputes z = bn(y) on load.
// HBM-streaming kernels: float4 per lane along C, consecutive lanes along C then W (coalesced).
#include "common.h"

namespace {

constexpr int THREADS = 256;
constexpr int REP = I2P_BN_REPLICAS;
constexpr int MAX_STAT_BLOCKS = 1024;

typedef float v4f __attribute__((ext_vector_type(4)));

// exact unsigned 32-bit division by a run-time constant without the ~25-instruction software divide (round-up multiplier, Granlund &
// Montgomery; the branch-free form of libdivide): q = mulhi(m, n); result = (((n - q) >> 1) + q) >> (L - 1), L = ceil(log2 d)
struct FastDiv { unsigned d, m, sh; };
inline FastDiv fast_div(unsigned d) {
    FastDiv f; f.d = d; f.m = 0; f.sh = 0;
    if (d > 1) {
        unsigned L = 0;
        while ((1ull << L) < d) ++L;
        f.m = (unsigned)((((1ull << 32) * ((1ull << L) - d)) / d) + 1);
        f.sh = L - 1;
    }
    return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv &f) {
    if (f.d == 1) return n;
    const unsigned q = __umulhi(f.m, n);
    return (((n - q) >> 1) + q) >> f.sh;
}

// Logical block id of the pooling kernels.  A pooled element reads a 3x3 window of the conv output, so vertically adjacent rows share
// input rows; rows are ~10 blocks apart and the dispatcher places consecutive blocks on consecutive XCDs (b % 8, observed), each with
// its own L2: every XCD fetched the shared rows again — PMC FETCH_SIZE (profiles/r04_pmc_img_*): 1.53x the conv output in
// img_pool_fwd at stride 2, 3.1x at stride 1, 1.5-2.2x in img_bwd_dx / img_bwd_stats.  With the swizzle the blocks an XCD runs at a
// time cover a contiguous band of rows (grids are multiples of 8: grid_for), the shared rows hit in that XCD's L2.
__device__ __forceinline__ unsigned img_block_id() { return i2p_xcd_swizzle(blockIdx.x, gridDim.x); }

struct PoolGeom {
    int B, H, W, C, s, Ho, Wo, cv, cvs;      // cvs = log2(cv) (cv divides THREADS, so it is a power of two)
    FastDiv fH, fW, fHo, fWo;
};

// flat float4 index -> (b, row, col): 32-bit divisions unless the tensor has 2^31 float4 or more (a 64-bit
// division by a run-time divisor is ~100 VALU instructions; three of them per element made these streaming
// kernels instruction-bound at 2.5 TB/s)
template <bool WIDE> struct Idx { using type = unsigned; };
template <> struct Idx<true> { using type = unsigned long long; };
template <bool WIDE>
__device__ __forceinline__ void decode(typename Idx<WIDE>::type t, int cvs, const FastDiv &rows, const FastDiv &cols, int &b, int &r, int &c) {
    typename Idx<WIDE>::type q = t >> cvs;
    if constexpr (WIDE) {
        const unsigned long long q2 = q / cols.d;
        c = (int)(q - q2 * cols.d);
        b = (int)(q2 / rows.d);
        r = (int)(q2 - (unsigned long long)b * rows.d);
    } else {     // two multiply-high divisions (the software 32-bit divide: ~25 VALU instructions each, in kernels that are VALU-bound)
        const unsigned q2 = fdiv(q, cols);
        c = (int)(q - q2 * cols.d);
        const unsigned bb = fdiv(q2, rows);
        b = (int)bb;
        r = (int)(q2 - bb * rows.d);
    }
}

__device__ __forceinline__ double rep_sum(const double *sums, int c, int idx) {
    double a = 0.0;
#pragma unroll 8
    for (int r = 0; r < REP; ++r) a += sums[(size_t)r * 2 * c + idx];
    return a;
}

// per-channel batch statistics -> mean_invstd [2C]; running buffers (momentum update, unbiased variance,
// running mean of conv+bias when the caller skipped the cancelling conv bias)
__global__ void img_finalize_kernel(long long n, int c, const double *__restrict__ sums, float eps, float momentum,
                                    const float *__restrict__ conv_bias, float *__restrict__ running_mean,
                                    float *__restrict__ running_var, float *__restrict__ mean_invstd) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    const double m = rep_sum(sums, c, ch) / (double)n;
    double var = rep_sum(sums, c, c + ch) / (double)n - m * m;
    var = var < 0.0 ? 0.0 : var;
    mean_invstd[ch] = (float)m;
    mean_invstd[c + ch] = rsqrtf((float)var + eps);
    if (running_mean) {
        const float mb = (float)m + (conv_bias ? conv_bias[ch] : 0.f);
        running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * mb;
        const float unbiased = (float)(var * ((double)n / (double)(n > 1 ? n - 1 : 1)));
        running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * unbiased;
    }
}

// (Statistics kernels that finalise in their last block — returning fp64 atomics + a ticket + the last block's re-read of the replica
//  sums — were measured in round 3 against the 64-thread finalize launches they replace: 583 vs 598 samples/s.  Removed in round 5.)

struct Coef4 { float mean[4], invstd[4], scale[4], beta[4]; };

__device__ __forceinline__ Coef4 load_coef(const float *mean_invstd, const float *gamma, const float *beta, int c,
                                           int vcol) {
    Coef4 k;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = vcol * 4 + i;
        k.mean[i] = mean_invstd[ch]; k.invstd[i] = mean_invstd[c + ch];
        k.scale[i] = k.invstd[i] * gamma[ch]; k.beta[i] = beta[ch];
    }
    return k;
}

__device__ __forceinline__ float bn_z(float y, const Coef4 &k, int i) { return (y - k.mean[i]) * k.scale[i] + k.beta[i]; }

template <bool WIDE>
__global__ __launch_bounds__(THREADS) void img_pool_fwd_kernel(PoolGeom g, const float4 *__restrict__ y,
                                                                const float *__restrict__ mean_invstd,
                                                                const float *__restrict__ gamma,
                                                                const float *__restrict__ beta, float slope,
                                                                float4 *__restrict__ out, uchar4 *__restrict__ arg) {
    const long long total = (long long)g.B * g.Ho * g.Wo * g.cv;
    const int vcol = threadIdx.x & (g.cv - 1);              // THREADS % cv == 0 and the grid stride is a multiple of THREADS
    const Coef4 k = load_coef(mean_invstd, gamma, beta, g.C, vcol);
    for (long long t = (long long)img_block_id() * THREADS + threadIdx.x; t < total; t += (long long)gridDim.x * THREADS) {
        int b, ho, wo;
        decode<WIDE>((typename Idx<WIDE>::type)t, g.cvs, g.fHo, g.fWo, b, ho, wo);
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned char bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int h = ho * g.s - 1 + kh;
            if (h < 0 || h >= g.H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int w = wo * g.s - 1 + kw;
                if (w < 0 || w >= g.W) continue;
                const float4 v = y[(((long long)b * g.H + h) * g.W + w) * g.cv + vcol];
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float z = bn_z(vv[i], k, i);
                    const float a = z > 0.f ? z : z * slope;
                    if (a > best[i] || a != a) { best[i] = a; bi[i] = (unsigned char)(kh * 3 + kw); }   // first max wins
                }
            }
        }
        out[t] = make_float4(best[0], best[1], best[2], best[3]);
        arg[t] = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
    }
}

// sums over all conv-output positions of gz = dL/dz (z = BN output) and gz*xhat, visited through the pooled
// outputs: each pooled element routes its gradient to its arg-max position.
template <bool WIDE>
__global__ __launch_bounds__(THREADS) void img_bwd_stats_kernel(PoolGeom g, const float4 *__restrict__ gout,
                                                                 const uchar4 *__restrict__ arg,
                                                                 const float *__restrict__ y,
                                                                 const float *__restrict__ mean_invstd,
                                                                 const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta, float slope,
                                                                 double *__restrict__ dsums) {
    __shared__ double red[THREADS][8];
    const long long total = (long long)g.B * g.Ho * g.Wo * g.cv;
    const int vcol = threadIdx.x & (g.cv - 1);
    const Coef4 k = load_coef(mean_invstd, gamma, beta, g.C, vcol);
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    for (long long t = (long long)img_block_id() * THREADS + threadIdx.x; t < total; t += (long long)gridDim.x * THREADS) {
        int b, ho, wo;
        decode<WIDE>((typename Idx<WIDE>::type)t, g.cvs, g.fHo, g.fWo, b, ho, wo);
        const float4 go = gout[t];
        const uchar4 a = arg[t];
        const float gv[4] = {go.x, go.y, go.z, go.w};
        const unsigned char av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int h = ho * g.s - 1 + av[i] / 3, w = wo * g.s - 1 + av[i] % 3;
            const float yv = y[((((long long)b * g.H + h) * g.W + w) * g.cv + vcol) * 4 + i];
            const float z = bn_z(yv, k, i);
            const float gz = z > 0.f ? gv[i] : gv[i] * slope;
            const float xh = (yv - k.mean[i]) * k.invstd[i];
            s[i] += gz; q[i] += (double)gz * xh;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[threadIdx.x][i] = s[i]; red[threadIdx.x][4 + i] = q[i]; }
    __syncthreads();
    if (threadIdx.x < g.cv) {
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = threadIdx.x; t < THREADS; t += g.cv)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += red[t][i];
        double *rep = dsums + (size_t)(blockIdx.x % REP) * 2 * g.C;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            atomicAdd(rep + vcol * 4 + i, acc[i]);
            atomicAdd(rep + g.C + vcol * 4 + i, acc[4 + i]);
        }
    }
}

// dbeta = sum gz, dgamma = sum gz*xhat from the replicated fp64 sums (one thread per channel, so that the
// streaming kernel below does not re-reduce 2*REP doubles per channel in every thread)
__global__ void img_bwd_coef_kernel(int c, const double *__restrict__ dsums, float *__restrict__ dgamma,
                                    float *__restrict__ dbeta) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    dbeta[ch] = (float)rep_sum(dsums, c, ch);
    dgamma[ch] = (float)rep_sum(dsums, c, c + ch);
}

// dL/dy of the conv output: gather the pooled gradients whose arg-max is this position, LeakyReLU', BN backward.
template <bool WIDE>
__global__ __launch_bounds__(THREADS) void img_bwd_dx_kernel(PoolGeom g, const float4 *__restrict__ gout,
                                                              const uchar4 *__restrict__ arg,
                                                              const float4 *__restrict__ y,
                                                              const float *__restrict__ mean_invstd,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, float slope,
                                                              float4 *__restrict__ dy, const float *__restrict__ dgamma,
                                                              const float *__restrict__ dbeta) {
    const long long total = (long long)g.B * g.H * g.W * g.cv;
    const int vcol = threadIdx.x & (g.cv - 1);
    const Coef4 k = load_coef(mean_invstd, gamma, beta, g.C, vcol);
    const float n = (float)((long long)g.B * g.H * g.W);
    float mg[4], mgx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { mg[i] = dbeta[vcol * 4 + i] / n; mgx[i] = dgamma[vcol * 4 + i] / n; }
    for (long long t = (long long)img_block_id() * THREADS + threadIdx.x; t < total; t += (long long)gridDim.x * THREADS) {
        int b, h, w;
        decode<WIDE>((typename Idx<WIDE>::type)t, g.cvs, g.fH, g.fW, b, h, w);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        // outputs ho with ho*s-1 <= h <= ho*s+1
        // stride 1 or 2 (geom_ok): the four signed divisions by it are shifts (s = 1: the numerator may be -1 and stays -1; s = 2: it is >= 0)
        const int ssh = g.s - 1;
        const int ho0 = max(0, (h - 1 + g.s - 1) >> ssh), ho1 = min(g.Ho - 1, (h + 1) >> ssh);
        const int wo0 = max(0, (w - 1 + g.s - 1) >> ssh), wo1 = min(g.Wo - 1, (w + 1) >> ssh);
        for (int ho = ho0; ho <= ho1; ++ho)
            for (int wo = wo0; wo <= wo1; ++wo) {
                const unsigned char p = (unsigned char)((h - (ho * g.s - 1)) * 3 + (w - (wo * g.s - 1)));
                const long long o = (((long long)b * g.Ho + ho) * g.Wo + wo) * g.cv + vcol;
                const uchar4 a = arg[o];
                if (a.x != p && a.y != p && a.z != p && a.w != p) continue;
                const float4 go = gout[o];
                acc[0] += a.x == p ? go.x : 0.f; acc[1] += a.y == p ? go.y : 0.f;
                acc[2] += a.z == p ? go.z : 0.f; acc[3] += a.w == p ? go.w : 0.f;
            }
        const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(y) + t);   // streamed once: keep arg / gout in L2
        const float vv[4] = {v.x, v.y, v.z, v.w};
        float o4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float z = bn_z(vv[i], k, i);
            const float gz = z > 0.f ? acc[i] : acc[i] * slope;
            const float xh = (vv[i] - k.mean[i]) * k.invstd[i];
            o4[i] = k.scale[i] * (gz - mg[i] - xh * mgx[i]);
        }
        const v4f o = {o4[0], o4[1], o4[2], o4[3]};
        __builtin_nontemporal_store(o, reinterpret_cast<v4f *>(dy) + t);
    }
}


// ---- second generation: storage type per tensor + coefficients formed in the consumer's prologue -------------------------------
// YBF: the conv output y / its gradient dy are bf16 (MIOpen bf16 convolutions, ops.set_precision("bf16")); OBF: the pooled output /
// its gradient are bf16 (every block but the last of the bf16 part, whose output feeds fp32 consumers).  Arithmetic is fp32,
// statistics fp64.  A lane owns CPL consecutive channels of a pixel: 4 with fp32 y (16-byte loads), 8 with bf16 y (again 16-byte
// loads: with 4 the bf16 kernels issued twice the load instructions per byte and ran no faster than fp32).
// The per-channel mean / invstd (forward) and dbeta / dgamma (backward) are not written by a 64-thread launch between the
// statistics kernel and its consumer: every consumer block reduces the [REP][2C] fp64 replica sums itself (2C*REP loads spread over
// the block, through LDS); block 0 also writes them out and updates the running buffers.
constexpr int MAX_C2 = 512;

__device__ __forceinline__ float bfl(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bfh(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {            // v_cvt_pk_bf16_f32 (RNE)
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
}
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

// N channels at vector index i (units of N channels); NT: streamed once (non-temporal)
template <bool BF, int N, bool NT = false> __device__ __forceinline__ void ldv(const void *p, long long i, float (&f)[N]) {
    if constexpr (BF && N == 8) {
        v4u v;
        if constexpr (NT) v = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p) + i);
        else v = reinterpret_cast<const v4u *>(p)[i];
        f[0] = bfl(v.x); f[1] = bfh(v.x); f[2] = bfl(v.y); f[3] = bfh(v.y); f[4] = bfl(v.z); f[5] = bfh(v.z); f[6] = bfl(v.w); f[7] = bfh(v.w);
    } else if constexpr (BF) {
        v2u v;
        if constexpr (NT) v = __builtin_nontemporal_load(reinterpret_cast<const v2u *>(p) + i);
        else v = reinterpret_cast<const v2u *>(p)[i];
        f[0] = bfl(v.x); f[1] = bfh(v.x); f[2] = bfl(v.y); f[3] = bfh(v.y);
    } else {
#pragma unroll
        for (int h = 0; h < N / 4; ++h) {
            v4f v;
            if constexpr (NT) v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p) + i * (N / 4) + h);
            else v = reinterpret_cast<const v4f *>(p)[i * (N / 4) + h];
            f[4 * h] = v.x; f[4 * h + 1] = v.y; f[4 * h + 2] = v.z; f[4 * h + 3] = v.w;
        }
    }
}
template <bool BF, int N, bool NT = false> __device__ __forceinline__ void stv(void *p, long long i, const float (&f)[N]) {
    if constexpr (BF && N == 8) {
        const v4u o = {pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7])};
        if constexpr (NT) __builtin_nontemporal_store(o, reinterpret_cast<v4u *>(p) + i);
        else reinterpret_cast<v4u *>(p)[i] = o;
    } else if constexpr (BF) {
        const v2u o = {pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3])};
        if constexpr (NT) __builtin_nontemporal_store(o, reinterpret_cast<v2u *>(p) + i);
        else reinterpret_cast<v2u *>(p)[i] = o;
    } else {
#pragma unroll
        for (int h = 0; h < N / 4; ++h) {
            const v4f o = {f[4 * h], f[4 * h + 1], f[4 * h + 2], f[4 * h + 3]};
            if constexpr (NT) __builtin_nontemporal_store(o, reinterpret_cast<v4f *>(p) + i * (N / 4) + h);
            else reinterpret_cast<v4f *>(p)[i * (N / 4) + h] = o;
        }
    }
}
template <bool BF> __device__ __forceinline__ float ld1(const void *p, long long e) {
    if constexpr (BF) return __uint_as_float((unsigned)reinterpret_cast<const unsigned short *>(p)[e] << 16);
    else return reinterpret_cast<const float *>(p)[e];
}
// the N arg-max bytes of a lane (window position 0..8 per channel)
template <int N> struct ArgW { unsigned w[N / 4]; };
template <int N> __device__ __forceinline__ ArgW<N> ld_arg(const unsigned char *arg, long long i) {
    ArgW<N> a;
    if constexpr (N == 8) { const uint2 v = reinterpret_cast<const uint2 *>(arg)[i]; a.w[0] = v.x; a.w[1] = v.y; }
    else a.w[0] = reinterpret_cast<const unsigned *>(arg)[i];
    return a;
}
template <int N> __device__ __forceinline__ unsigned arg_byte(const ArgW<N> &a, int k) { return (a.w[k >> 2] >> (8 * (k & 3))) & 0xffu; }

// stat[i] = sum over the REP replicas of sums[r][i], i < 2C, for the whole block (fixed summation order); ends with a barrier
__device__ __forceinline__ void block_rep_sums(const double *__restrict__ sums, int c, double *stat, double *part) {
    const int n2 = 2 * c;
    if (n2 <= THREADS) {                                   // n2 is a power of two: groups of n2 threads share the replicas
        const int idx = threadIdx.x & (n2 - 1), grp = threadIdx.x / n2, ng = THREADS / n2;
        double a = 0.0;
        for (int r = grp; r < REP; r += ng) a += sums[(size_t)r * n2 + idx];
        part[threadIdx.x] = a;
        __syncthreads();
        if (threadIdx.x < n2) {
            double t = 0.0;
            for (int g2 = 0; g2 < ng; ++g2) t += part[g2 * n2 + threadIdx.x];
            stat[threadIdx.x] = t;
        }
    } else {
        for (int i = threadIdx.x; i < n2; i += THREADS) stat[i] = rep_sum(sums, c, i);
    }
    __syncthreads();
}

template <int N> struct CoefN { float mean[N], invstd[N], scale[N], beta[N]; };
template <int N> __device__ __forceinline__ CoefN<N> load_coef_n(const float *mean_invstd, const float *gamma, const float *beta, int c, int vcol) {
    CoefN<N> k;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int ch = vcol * N + i;
        k.mean[i] = mean_invstd[ch]; k.invstd[i] = mean_invstd[c + ch];
        k.scale[i] = k.invstd[i] * gamma[ch]; k.beta[i] = beta[ch];
    }
    return k;
}
template <int N> __device__ __forceinline__ float bn_zn(float y, const CoefN<N> &k, int i) { return (y - k.mean[i]) * k.scale[i] + k.beta[i]; }

template <bool BF>
__global__ __launch_bounds__(THREADS) void img_stats2_kernel(long long n, int c, int cv, int rpb, const void *__restrict__ y,
                                                             double *__restrict__ sums) {
    constexpr int N = BF ? 8 : 4;
    __shared__ double red[THREADS][2 * N];
    const int vcol = threadIdx.x % cv, rsub = threadIdx.x / cv;
    double s[N], q[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { s[i] = 0.0; q[i] = 0.0; }
    const long long stride = (long long)gridDim.x * rpb;
    for (long long r0 = (long long)blockIdx.x * rpb + rsub; r0 < n; r0 += stride * 4) {
        float v[4][N];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long r = r0 + u * stride;
            if (r < n) ldv<BF, N>(y, r * cv + vcol, v[u]);
            else {
#pragma unroll
                for (int i = 0; i < N; ++i) v[u][i] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < N; ++i) { s[i] += v[u][i]; q[i] += (double)v[u][i] * v[u][i]; }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) { red[threadIdx.x][i] = s[i]; red[threadIdx.x][N + i] = q[i]; }
    __syncthreads();
    if (threadIdx.x < cv) {
        double a[2 * N];
#pragma unroll
        for (int i = 0; i < 2 * N; ++i) a[i] = 0.0;
        for (int t = threadIdx.x; t < THREADS; t += cv)
#pragma unroll
            for (int i = 0; i < 2 * N; ++i) a[i] += red[t][i];
        double *rep = sums + (size_t)(blockIdx.x % REP) * 2 * c;
#pragma unroll
        for (int i = 0; i < N; ++i) { atomicAdd(rep + vcol * N + i, a[i]); atomicAdd(rep + c + vcol * N + i, a[N + i]); }
    }
}

template <bool WIDE, bool YBF, bool OBF>
__global__ __launch_bounds__(THREADS) void img_pool_fwd2_kernel(PoolGeom g, const void *__restrict__ y, const double *__restrict__ sums,
                                                                const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                                float slope, float momentum, const float *__restrict__ conv_bias,
                                                                float *__restrict__ running_mean, float *__restrict__ running_var,
                                                                void *__restrict__ out, unsigned char *__restrict__ arg,
                                                                float *__restrict__ mean_invstd, int POOL_RC) {
    constexpr int N = YBF ? 8 : 4;
    __shared__ double stat[2 * MAX_C2], part[THREADS];
    block_rep_sums(sums, g.C, stat, part);
    const double n = (double)((long long)g.B * g.H * g.W);
    const int vcol = threadIdx.x & (g.cv - 1);
    CoefN<N> k;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int ch = vcol * N + i;
        const double m = stat[ch] / n;
        double var = stat[g.C + ch] / n - m * m;
        var = var < 0.0 ? 0.0 : var;
        k.mean[i] = (float)m; k.invstd[i] = rsqrtf((float)var + eps);
        k.scale[i] = k.invstd[i] * gamma[ch]; k.beta[i] = beta[ch];
    }
    if (blockIdx.x == 0)
        for (int ch = threadIdx.x; ch < g.C; ch += THREADS) {
            const double m = stat[ch] / n;
            double var = stat[g.C + ch] / n - m * m;
            var = var < 0.0 ? 0.0 : var;
            mean_invstd[ch] = (float)m;
            mean_invstd[g.C + ch] = rsqrtf((float)var + eps);
            if (running_mean) {
                const float mb = (float)m + (conv_bias ? conv_bias[ch] : 0.f);
                running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * mb;
                const float unbiased = (float)(var * (n / (n > 1.0 ? n - 1.0 : 1.0)));
                running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * unbiased;
            }
        }
    // Sliding window down a column of outputs (round 4; the round-3 form visited the 3x3 window of every pooled element from scratch:
    // 9 loads and 9 x N BN + activation evaluations per element — SQ counters, profiles/r04_pmc_img_SQ*.txt: ~500 VALU instructions per
    // lane and element, the stride-1 blocks were VALU-bound at 2.8 TB/s).  A block owns a tile of POOL_RC output rows x (THREADS / cv)
    // output columns of one image; a lane walks its column downwards and keeps, per INPUT row, the first maximum over the row's three
    // window columns (value + column index): stride 1 forms one new row result per output (3 loads, 3 x N evaluations), stride 2 two.
    // The window maximum = first maximum over the three row results = first maximum in (kh, kw) scan order, NaN propagating to the LAST
    // NaN, exactly as the flat scan (max_pool2d's rule; bit-exact arg-max against the oracle, tests/test_ops_gpu.py).
    const int twp = THREADS >> g.cvs;                              // output columns of a tile
    const int tiles_w = (g.Wo + twp - 1) / twp, chunks_h = (g.Ho + POOL_RC - 1) / POOL_RC;
    const unsigned ntiles = (unsigned)(g.B * chunks_h * tiles_w);
    const int wo_l = threadIdx.x >> g.cvs;
    for (unsigned tile = img_block_id(); tile < ntiles; tile += gridDim.x) {
        const unsigned bh = tile / (unsigned)tiles_w;
        const int wt = (int)(tile - bh * (unsigned)tiles_w), b = (int)(bh / (unsigned)chunks_h), hc = (int)(bh - (unsigned)b * (unsigned)chunks_h);
        const int wo = wt * twp + wo_l;
        if (wo >= g.Wo) continue;
        const int ho0 = hc * POOL_RC, ho1 = min(g.Ho, ho0 + POOL_RC);
        const long long img = (long long)b * g.H;
        float rv[3][N];
        unsigned ra[3][N];
        auto row = [&](int h, float (&v)[N], unsigned (&aw)[N]) {          // first maximum over the window columns of input row h
#pragma unroll
            for (int i = 0; i < N; ++i) { v[i] = -INFINITY; aw[i] = 0u; }
            if (h < 0 || h >= g.H) return;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int w = wo * g.s - 1 + kw;
                if (w < 0 || w >= g.W) continue;
                float vv[N];
                ldv<YBF, N>(y, ((img + h) * g.W + w) * g.cv + vcol, vv);
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const float z = bn_zn<N>(vv[i], k, i);
                    const float a = z > 0.f ? z : z * slope;
                    if (a > v[i] || a != a) { v[i] = a; aw[i] = (unsigned)kw; }
                }
            }
        };
        auto emit = [&](int ho) {
            float best[N];
            unsigned bi[N];
#pragma unroll
            for (int i = 0; i < N; ++i) { best[i] = -INFINITY; bi[i] = 0u; }
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int i = 0; i < N; ++i)
                    if (rv[kh][i] > best[i] || rv[kh][i] != rv[kh][i]) { best[i] = rv[kh][i]; bi[i] = (unsigned)(kh * 3) + ra[kh][i]; }
            const long long t = (((long long)b * g.Ho + ho) * g.Wo + wo) * g.cv + vcol;
            stv<OBF, N>(out, t, best);
            if constexpr (N == 8)
                reinterpret_cast<uint2 *>(arg)[t] = make_uint2(bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24),
                                                               bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24));
            else
                reinterpret_cast<unsigned *>(arg)[t] = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
        };
        auto shift = [&](int dst, int src) {
#pragma unroll
            for (int i = 0; i < N; ++i) { rv[dst][i] = rv[src][i]; ra[dst][i] = ra[src][i]; }
        };
        if (g.s == 1) {
            row(ho0 - 1, rv[0], ra[0]);
            row(ho0, rv[1], ra[1]);
            for (int ho = ho0; ho < ho1; ++ho) {
                row(ho + 1, rv[2], ra[2]);
                emit(ho);
                shift(0, 1); shift(1, 2);
            }
        } else {
            row(2 * ho0 - 1, rv[0], ra[0]);
            for (int ho = ho0; ho < ho1; ++ho) {
                row(2 * ho, rv[1], ra[1]);
                row(2 * ho + 1, rv[2], ra[2]);
                emit(ho);
                shift(0, 2);
            }
        }
    }
}

template <bool WIDE, bool YBF, bool OBF>
__global__ __launch_bounds__(THREADS) void img_bwd_stats2_kernel(PoolGeom g, const void *__restrict__ gout, const unsigned char *__restrict__ arg,
                                                                 const void *__restrict__ y, const float *__restrict__ mean_invstd,
                                                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                 float slope, double *__restrict__ dsums) {
    constexpr int N = YBF ? 8 : 4;
    __shared__ double red[THREADS][2 * N];
    const long long total = (long long)g.B * g.Ho * g.Wo * g.cv;
    const int vcol = threadIdx.x & (g.cv - 1);
    const CoefN<N> k = load_coef_n<N>(mean_invstd, gamma, beta, g.C, vcol);
    double s[N], q[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { s[i] = 0.0; q[i] = 0.0; }
    for (long long t = (long long)img_block_id() * THREADS + threadIdx.x; t < total; t += (long long)gridDim.x * THREADS) {
        int b, ho, wo;
        decode<WIDE>((typename Idx<WIDE>::type)t, g.cvs, g.fHo, g.fWo, b, ho, wo);
        float gv[N];
        ldv<OBF, N>(gout, t, gv);
        const ArgW<N> a = ld_arg<N>(arg, t);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int av = (int)arg_byte<N>(a, i);
            const int h = ho * g.s - 1 + av / 3, w = wo * g.s - 1 + av % 3;
            const float yv = ld1<YBF>(y, ((((long long)b * g.H + h) * g.W + w) * g.cv + vcol) * N + i);
            const float z = bn_zn<N>(yv, k, i);
            const float gz = z > 0.f ? gv[i] : gv[i] * slope;
            const float xh = (yv - k.mean[i]) * k.invstd[i];
            s[i] += gz; q[i] += (double)gz * xh;
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) { red[threadIdx.x][i] = s[i]; red[threadIdx.x][N + i] = q[i]; }
    __syncthreads();
    if (threadIdx.x < g.cv) {
        double acc[2 * N];
#pragma unroll
        for (int i = 0; i < 2 * N; ++i) acc[i] = 0.0;
        for (int t = threadIdx.x; t < THREADS; t += g.cv)
#pragma unroll
            for (int i = 0; i < 2 * N; ++i) acc[i] += red[t][i];
        double *rep = dsums + (size_t)(blockIdx.x % REP) * 2 * g.C;
#pragma unroll
        for (int i = 0; i < N; ++i) { atomicAdd(rep + vcol * N + i, acc[i]); atomicAdd(rep + g.C + vcol * N + i, acc[N + i]); }
    }
}

template <bool WIDE, bool YBF, bool OBF>
__global__ __launch_bounds__(THREADS) void img_bwd_dx2_kernel(PoolGeom g, const void *__restrict__ gout, const unsigned char *__restrict__ arg,
                                                              const void *__restrict__ y, const float *__restrict__ mean_invstd,
                                                              const float *__restrict__ gamma, const float *__restrict__ beta, float slope,
                                                              const double *__restrict__ dsums, void *__restrict__ dy,
                                                              float *__restrict__ dgamma, float *__restrict__ dbeta) {
    constexpr int N = YBF ? 8 : 4;
    __shared__ double stat[2 * MAX_C2], part[THREADS];
    block_rep_sums(dsums, g.C, stat, part);
    if (blockIdx.x == 0)
        for (int ch = threadIdx.x; ch < g.C; ch += THREADS) { dbeta[ch] = (float)stat[ch]; dgamma[ch] = (float)stat[g.C + ch]; }
    const long long total = (long long)g.B * g.H * g.W * g.cv;
    const int vcol = threadIdx.x & (g.cv - 1);
    const CoefN<N> k = load_coef_n<N>(mean_invstd, gamma, beta, g.C, vcol);
    const float n = (float)((long long)g.B * g.H * g.W);
    float mg[N], mgx[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { mg[i] = (float)stat[vcol * N + i] / n; mgx[i] = (float)stat[g.C + vcol * N + i] / n; }
    for (long long t = (long long)img_block_id() * THREADS + threadIdx.x; t < total; t += (long long)gridDim.x * THREADS) {
        int b, h, w;
        decode<WIDE>((typename Idx<WIDE>::type)t, g.cvs, g.fH, g.fW, b, h, w);
        float acc[N];
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] = 0.f;
        // (a version that requested the arg-max words of all 9 candidate outputs in one batch and their gradients in a second one
        //  measured 1.7x SLOWER: these kernels are bound by VALU work per element — index math, byte compares, unpacking — not by
        //  the latency of the dependent loads)
        // stride 1 or 2 (geom_ok): the four signed divisions by it are shifts (s = 1: the numerator may be -1 and stays -1; s = 2: it is >= 0)
        const int ssh = g.s - 1;
        const int ho0 = max(0, (h - 1 + g.s - 1) >> ssh), ho1 = min(g.Ho - 1, (h + 1) >> ssh);
        const int wo0 = max(0, (w - 1 + g.s - 1) >> ssh), wo1 = min(g.Wo - 1, (w + 1) >> ssh);
        for (int ho = ho0; ho <= ho1; ++ho)
            for (int wo = wo0; wo <= wo1; ++wo) {
                const unsigned p = (unsigned)((h - (ho * g.s - 1)) * 3 + (w - (wo * g.s - 1)));
                const long long o = (((long long)b * g.Ho + ho) * g.Wo + wo) * g.cv + vcol;
                const ArgW<N> a = ld_arg<N>(arg, o);
                bool any = false;
#pragma unroll
                for (int i = 0; i < N; ++i) any = any || arg_byte<N>(a, i) == p;
                if (!any) continue;
                float go[N];
                ldv<OBF, N>(gout, o, go);
#pragma unroll
                for (int i = 0; i < N; ++i) acc[i] += arg_byte<N>(a, i) == p ? go[i] : 0.f;
            }
        float vv[N], o4[N];
        ldv<YBF, N, true>(y, t, vv);                        // streamed once: keep arg / gout in L2
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float z = bn_zn<N>(vv[i], k, i);
            const float gz = z > 0.f ? acc[i] : acc[i] * slope;
            const float xh = (vv[i] - k.mean[i]) * k.invstd[i];
            o4[i] = k.scale[i] * (gz - mg[i] - xh * mgx[i]);
        }
        stv<YBF, N, true>(dy, t, o4);
    }
}

bool geom_ok(int B, int H, int W, int C, int s) {
    return B >= 0 && H > 0 && W > 0 && C >= 4 && C % 4 == 0 && (THREADS % (C / 4)) == 0 && ((C / 4) & (C / 4 - 1)) == 0 && (s == 1 || s == 2);
}

PoolGeom make_geom(int B, int H, int W, int C, int s) {
    PoolGeom g;
    g.B = B; g.H = H; g.W = W; g.C = C; g.s = s; g.cv = C / 4;
    g.cvs = 0;
    while ((1 << g.cvs) < g.cv) ++g.cvs;
    g.Ho = (H - 1) / s + 1; g.Wo = (W - 1) / s + 1;        // floor((H + 2*1 - 3)/s) + 1
    g.fH = fast_div((unsigned)H); g.fW = fast_div((unsigned)W); g.fHo = fast_div((unsigned)g.Ho); g.fWo = fast_div((unsigned)g.Wo);
    return g;
}

unsigned grid_for(long long total, int cap) {      // a multiple of 8 (img_block_id): surplus blocks find t >= total and leave
    long long b = (total + THREADS - 1) / THREADS;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)((b + 7) & ~7LL);
}

}  // namespace

extern "C" int i2p_img_bn_pool_fwd(int B, int H, int W, int C, int stride, const float *y, const double *sums,
                                   const float *gamma, const float *beta, float eps, float slope, float momentum,
                                   const float *conv_bias, float *running_mean, float *running_var, float *out,
                                   unsigned char *arg, float *mean_invstd, void *stream) {
    if (!geom_ok(B, H, W, C, stride)) return I2P_ERR_BAD_ARG;
    if (B == 0) return 0;
    const PoolGeom g = make_geom(B, H, W, C, stride);
    hipStream_t st = (hipStream_t)stream;
    if (!sums) return I2P_ERR_BAD_ARG;
    hipLaunchKernelGGL(img_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, st, (long long)B * H * W, C, sums, eps,
                       momentum, conv_bias, running_mean, running_var, mean_invstd);
    const long long total = (long long)B * g.Ho * g.Wo * g.cv;
    if (total < (1ll << 31))
        hipLaunchKernelGGL(img_pool_fwd_kernel<false>, dim3(grid_for(total, 1 << 14)), dim3(THREADS), 0, st, g,
                           (const float4 *)y, mean_invstd, gamma, beta, slope, (float4 *)out, (uchar4 *)arg);
    else
        hipLaunchKernelGGL(img_pool_fwd_kernel<true>, dim3(grid_for(total, 1 << 14)), dim3(THREADS), 0, st, g,
                           (const float4 *)y, mean_invstd, gamma, beta, slope, (float4 *)out, (uchar4 *)arg);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_img_bn_pool_bwd(int B, int H, int W, int C, int stride, const float *gout, const unsigned char *arg,
                                   const float *y, const float *mean_invstd, const float *gamma, const float *beta,
                                   float slope, double *dsums, float *dy, float *dgamma, float *dbeta, void *stream) {
    if (!geom_ok(B, H, W, C, stride)) return I2P_ERR_BAD_ARG;
    if (B == 0) return 0;
    const PoolGeom g = make_geom(B, H, W, C, stride);
    hipStream_t st = (hipStream_t)stream;
    const long long tot_o = (long long)B * g.Ho * g.Wo * g.cv, tot_i = (long long)B * H * W * g.cv;
    const bool wide = tot_i >= (1ll << 31);
    if (!wide)
        hipLaunchKernelGGL(img_bwd_stats_kernel<false>, dim3(grid_for(tot_o, MAX_STAT_BLOCKS)), dim3(THREADS), 0, st, g,
                           (const float4 *)gout, (const uchar4 *)arg, y, mean_invstd, gamma, beta, slope, dsums);
    else
        hipLaunchKernelGGL(img_bwd_stats_kernel<true>, dim3(grid_for(tot_o, MAX_STAT_BLOCKS)), dim3(THREADS), 0, st, g,
                           (const float4 *)gout, (const uchar4 *)arg, y, mean_invstd, gamma, beta, slope, dsums);
    hipLaunchKernelGGL(img_bwd_coef_kernel, dim3((C + 63) / 64), dim3(64), 0, st, C, dsums, dgamma, dbeta);
    if (!wide)
        hipLaunchKernelGGL(img_bwd_dx_kernel<false>, dim3(grid_for(tot_i, 1 << 12)), dim3(THREADS), 0, st, g,
                           (const float4 *)gout, (const uchar4 *)arg, (const float4 *)y, mean_invstd, gamma, beta, slope,
                           (float4 *)dy, dgamma, dbeta);
    else
        hipLaunchKernelGGL(img_bwd_dx_kernel<true>, dim3(grid_for(tot_i, 1 << 12)), dim3(THREADS), 0, st, g,
                           (const float4 *)gout, (const uchar4 *)arg, (const float4 *)y, mean_invstd, gamma, beta, slope,
                           (float4 *)dy, dgamma, dbeta);
    I2P_RETURN_LAUNCH_STATUS();
}

// gen-2 geometry: a lane owns 8 channels when y is bf16 (16-byte loads), 4 otherwise
static bool geom2_ok(int B, int H, int W, int C, int s, int y_bf16) {
    const int cpl = y_bf16 ? 8 : 4;
    return B >= 0 && H > 0 && W > 0 && C >= cpl && C % cpl == 0 && C <= MAX_C2 && (THREADS % (C / cpl)) == 0 && ((C / cpl) & (C / cpl - 1)) == 0 &&
           (s == 1 || s == 2);
}
static PoolGeom make_geom2(int B, int H, int W, int C, int s, int y_bf16) {
    PoolGeom g = make_geom(B, H, W, C, s);
    g.cv = C / (y_bf16 ? 8 : 4);
    g.cvs = 0;
    while ((1 << g.cvs) < g.cv) ++g.cvs;
    return g;
}

static int gen2_grid(int dflt) { return dflt; }

// ---- second-generation entry points (two launches each way) ----------------------------------------------------------------------
#define IMG_DISPATCH(KERNEL, GRID, ...)                                                                                              \
    do {                                                                                                                             \
        const int sel = (wide ? 4 : 0) | (y_bf16 ? 2 : 0) | (out_bf16 ? 1 : 0);                                                      \
        switch (sel) {                                                                                                               \
        case 0: hipLaunchKernelGGL((KERNEL<false, false, false>), GRID, dim3(THREADS), 0, st, __VA_ARGS__); break;                   \
        case 1: hipLaunchKernelGGL((KERNEL<false, false, true>), GRID, dim3(THREADS), 0, st, __VA_ARGS__); break;                    \
        case 2: hipLaunchKernelGGL((KERNEL<false, true, false>), GRID, dim3(THREADS), 0, st, __VA_ARGS__); break;                    \
        case 3: hipLaunchKernelGGL((KERNEL<false, true, true>), GRID, dim3(THREADS), 0, st, __VA_ARGS__); break;                     \
        case 4: hipLaunchKernelGGL((KERNEL<true, false, false>), GRID, dim3(THREADS), 0, st, __VA_ARGS__); break;                    \
        case 5: hipLaunchKernelGGL((KERNEL<true, false, true>), GRID, dim3(THREADS), 0, st, __VA_ARGS__); break;                     \
        case 6: hipLaunchKernelGGL((KERNEL<true, true, false>), GRID, dim3(THREADS), 0, st, __VA_ARGS__); break;                     \
        default: hipLaunchKernelGGL((KERNEL<true, true, true>), GRID, dim3(THREADS), 0, st, __VA_ARGS__); break;                     \
        }                                                                                                                            \
    } while (0)

// BN(batch statistics) + LeakyReLU + MaxPool(3, stride, 1) of the conv output y [B,H,W,C] (fp32, or bf16 bits when y_bf16):
// statistics into `sums` (f64 [I2P_BN_REPLICAS][2C], zeroed by the caller), then the pooling kernel, which forms mean / invstd from
// the sums in every block's prologue; out [B,Ho,Wo,C] fp32 or bf16 (out_bf16), arg u8, mean_invstd [2C] and the running buffers written.
static int img_block_fwd_impl(bool with_stats, int B, int H, int W, int C, int stride, int y_bf16, int out_bf16, const void *y, double *sums,
                              const float *gamma, const float *beta, float eps, float slope, float momentum, const float *conv_bias,
                              float *running_mean, float *running_var, void *out, unsigned char *arg, float *mean_invstd, void *stream) {
    if (!geom2_ok(B, H, W, C, stride, y_bf16)) return I2P_ERR_BAD_ARG;
    if (B == 0) return 0;
    if (!y || !sums || !gamma || !beta || !out || !arg || !mean_invstd) return I2P_ERR_BAD_ARG;
    const PoolGeom g = make_geom2(B, H, W, C, stride, y_bf16);
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)B * H * W;
    const int rpb = THREADS / g.cv;
    long long blocks = (n + (long long)rpb * 4 - 1) / ((long long)rpb * 4);
    if (blocks > MAX_STAT_BLOCKS) blocks = MAX_STAT_BLOCKS;
    if (!with_stats) {}
    else if (y_bf16)
        hipLaunchKernelGGL(img_stats2_kernel<true>, dim3((unsigned)blocks), dim3(THREADS), 0, st, n, C, g.cv, rpb, y, sums);
    else
        hipLaunchKernelGGL(img_stats2_kernel<false>, dim3((unsigned)blocks), dim3(THREADS), 0, st, n, C, g.cv, rpb, y, sums);
    const long long total = (long long)B * g.Ho * g.Wo * g.cv;
    const bool wide = n * g.cv >= (1ll << 31);
    // tiles of rc output rows x (THREADS / cv) output columns; rc = 8 while that still gives >= 2048 tiles (the small late blocks: fewer rows)
    const int twp = THREADS / g.cv, tiles_w = (g.Wo + twp - 1) / twp;
    int rc = 8;
    while (rc > 1 && (long long)B * ((g.Ho + rc - 1) / rc) * tiles_w < 2048) rc >>= 1;
    const long long tiles = (long long)B * ((g.Ho + rc - 1) / rc) * tiles_w;
    (void)total;
    IMG_DISPATCH(img_pool_fwd2_kernel, dim3(grid_for(tiles * THREADS, gen2_grid(1 << 14))), g, y, (const double *)sums, gamma, beta, eps, slope, momentum,
                 conv_bias, running_mean, running_var, out, arg, mean_invstd, rc);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_img_block_fwd(int B, int H, int W, int C, int stride, int y_bf16, int out_bf16, const void *y, double *sums,
                                 const float *gamma, const float *beta, float eps, float slope, float momentum, const float *conv_bias,
                                 float *running_mean, float *running_var, void *out, unsigned char *arg, float *mean_invstd, void *stream) {
    return img_block_fwd_impl(true, B, H, W, C, stride, y_bf16, out_bf16, y, sums, gamma, beta, eps, slope, momentum, conv_bias, running_mean,
                              running_var, out, arg, mean_invstd, stream);
}
// the same with `sums` already holding sum y / sum y^2 (written by the producer of y: i2p_img_conv16_fwd): only the pooling launch
extern "C" int i2p_img_block_pool(int B, int H, int W, int C, int stride, int y_bf16, int out_bf16, const void *y, double *sums,
                                  const float *gamma, const float *beta, float eps, float slope, float momentum, const float *conv_bias,
                                  float *running_mean, float *running_var, void *out, unsigned char *arg, float *mean_invstd, void *stream) {
    return img_block_fwd_impl(false, B, H, W, C, stride, y_bf16, out_bf16, y, sums, gamma, beta, eps, slope, momentum, conv_bias, running_mean,
                              running_var, out, arg, mean_invstd, stream);
}

// backward of the same: (gout [B,Ho,Wo,C] fp32/bf16, arg, y, mean_invstd) -> dy [B,H,W,C] (storage of y), dgamma, dbeta [C] fp32;
// dsums: f64 [I2P_BN_REPLICAS][2C], zeroed by the caller
static int img_block_bwd_impl(int parts, int B, int H, int W, int C, int stride, int y_bf16, int out_bf16, const void *gout,
                              const unsigned char *arg, const void *y, const float *mean_invstd, const float *gamma, const float *beta, float slope,
                              double *dsums, void *dy, float *dgamma, float *dbeta, void *stream) {
    if (!geom2_ok(B, H, W, C, stride, y_bf16)) return I2P_ERR_BAD_ARG;
    if (B == 0) return 0;
    if (!gout || !arg || !y || !mean_invstd || !gamma || !beta || !dsums) return I2P_ERR_BAD_ARG;
    if ((parts & 2) && (!dy || !dgamma || !dbeta)) return I2P_ERR_BAD_ARG;
    const PoolGeom g = make_geom2(B, H, W, C, stride, y_bf16);
    hipStream_t st = (hipStream_t)stream;
    const long long tot_o = (long long)B * g.Ho * g.Wo * g.cv, tot_i = (long long)B * H * W * g.cv;
    const bool wide = tot_i >= (1ll << 31);
    if (parts & 1)
        IMG_DISPATCH(img_bwd_stats2_kernel, dim3(grid_for(tot_o, MAX_STAT_BLOCKS)), g, gout, arg, y, mean_invstd, gamma, beta, slope, dsums);
    if (parts & 2)
        IMG_DISPATCH(img_bwd_dx2_kernel, dim3(grid_for(tot_i, gen2_grid(1 << 12))), g, gout, arg, y, mean_invstd, gamma, beta, slope,
                     (const double *)dsums, dy, dgamma, dbeta);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_img_block_bwd(int B, int H, int W, int C, int stride, int y_bf16, int out_bf16, const void *gout, const unsigned char *arg,
                                 const void *y, const float *mean_invstd, const float *gamma, const float *beta, float slope, double *dsums,
                                 void *dy, float *dgamma, float *dbeta, void *stream) {
    return img_block_bwd_impl(3, B, H, W, C, stride, y_bf16, out_bf16, gout, arg, y, mean_invstd, gamma, beta, slope, dsums, dy, dgamma, dbeta,
                              stream);
}
// the same with `dsums` already holding sum gz / sum gz xhat (written by an earlier i2p_img_block_bwd_stats call): only the
// dy launch
extern "C" int i2p_img_block_bwd_dx(int B, int H, int W, int C, int stride, int y_bf16, int out_bf16, const void *gout, const unsigned char *arg,
                                    const void *y, const float *mean_invstd, const float *gamma, const float *beta, float slope, double *dsums,
                                    void *dy, float *dgamma, float *dbeta, void *stream) {
    return img_block_bwd_impl(2, B, H, W, C, stride, y_bf16, out_bf16, gout, arg, y, mean_invstd, gamma, beta, slope, dsums, dy, dgamma, dbeta,
                              stream);
}
// only the statistics pass of i2p_img_block_bwd (dsums: zeroed [I2P_BN_REPLICAS][2C] doubles): what i2p_img_conv_tail_bwd consumes
extern "C" int i2p_img_block_bwd_stats(int B, int H, int W, int C, int stride, int y_bf16, int out_bf16, const void *gout, const unsigned char *arg,
                                       const void *y, const float *mean_invstd, const float *gamma, const float *beta, float slope, double *dsums,
                                       void *stream) {
    return img_block_bwd_impl(1, B, H, W, C, stride, y_bf16, out_bf16, gout, arg, y, mean_invstd, gamma, beta, slope, dsums, nullptr, nullptr, nullptr,
                              stream);
}
