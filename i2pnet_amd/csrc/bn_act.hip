// Batch-statistics BatchNorm + (Leaky)ReLU on channel-last activations [rows, C] for gfx950.
// Replaces the eager permute / BatchNorm2d / activation / permute of every point-branch Conv2d
// of the reference (src/projectPN/PPBackbone_center.py:34-46) — see include/i2p_ops.h.
//
// All four kernels are HBM-streaming: float4 per lane along C (coalesced 16 B/lane), grid-stride
// over rows, fp64 per-lane partial sums, one LDS reduction and C fp64 atomics per block.
//   forward : bn_stats (1 read)            + bn_act_fwd (1 read, 1 write)
//   backward: bn_act_bwd_stats (2 reads)   + bn_act_bwd (2 reads, 1 write)
#include "common.h"

namespace {

constexpr int THREADS = 256;
constexpr int MAX_BLOCKS = 2048;      // streaming (apply) kernels
constexpr int MAX_STAT_BLOCKS = 1024; // reduction kernels: fewer, fatter blocks
constexpr int UNROLL = 4;             // independent 16-B loads in flight per lane in the reductions
constexpr int REP = I2P_BN_REPLICAS;  // same-address fp64 atomics cost ~90 ns each: spread them

struct BnGeom {
    int cv;        // float4 columns = C/4
    int rpb;       // rows per block iteration = THREADS / cv
};

__device__ __forceinline__ float act_fwd(float z, float slope) { return z > 0.f ? z : z * slope; }

// ---- vector path: C % 4 == 0 and (C/4) divides 256 -------------------------------------------
__global__ __launch_bounds__(THREADS) void bn_stats_v4(long long rows, int c, BnGeom g,
                                                        const float4 *__restrict__ y,
                                                        double *__restrict__ sums) {
    __shared__ double red[THREADS][8];
    const int vcol = threadIdx.x % g.cv, rsub = threadIdx.x / g.cv;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    const long long stride = (long long)gridDim.x * g.rpb;
    for (long long r0 = (long long)blockIdx.x * g.rpb + rsub; r0 < rows; r0 += stride * UNROLL) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long long r = r0 + u * stride;
            v[u] = r < rows ? y[r * g.cv + vcol] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            s[0] += v[u].x; s[1] += v[u].y; s[2] += v[u].z; s[3] += v[u].w;
            q[0] += (double)v[u].x * v[u].x; q[1] += (double)v[u].y * v[u].y;
            q[2] += (double)v[u].z * v[u].z; q[3] += (double)v[u].w * v[u].w;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[threadIdx.x][i] = s[i]; red[threadIdx.x][4 + i] = q[i]; }
    __syncthreads();
    if (threadIdx.x < g.cv) {
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = threadIdx.x; t < THREADS; t += g.cv)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] += red[t][i];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double *rep = sums + (size_t)(blockIdx.x % REP) * 2 * c;
            atomicAdd(rep + vcol * 4 + i, a[i]);
            atomicAdd(rep + c + vcol * 4 + i, a[4 + i]);
        }
    }
}

struct ChanCoef { float mean, invstd, scale, beta; };

__device__ __forceinline__ double rep_sum(const double *sums, int c, int idx) {
    double a = 0.0;
#pragma unroll 8
    for (int r = 0; r < REP; ++r) a += sums[(size_t)r * 2 * c + idx];
    return a;
}

__device__ __forceinline__ ChanCoef coef_from_sums(const double *sums, int c, int ch, long long rows,
                                                   const float *gamma, const float *beta, float eps) {
    const double m = rep_sum(sums, c, ch) / (double)rows;
    double var = rep_sum(sums, c, c + ch) / (double)rows - m * m;
    var = var < 0.0 ? 0.0 : var;
    ChanCoef k;
    k.mean = (float)m;
    k.invstd = rsqrtf((float)var + eps);
    k.scale = k.invstd * gamma[ch];
    k.beta = beta[ch];
    return k;
}

__global__ __launch_bounds__(THREADS) void bn_act_fwd_v4(long long rows, int c, BnGeom g,
                                                          const float4 *__restrict__ y,
                                                          const double *__restrict__ sums,
                                                          const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float eps, float slope,
                                                          float4 *__restrict__ out,
                                                          float *__restrict__ mean_invstd) {
    const int vcol = threadIdx.x % g.cv, rsub = threadIdx.x / g.cv;
    // per-channel constants once per BLOCK (the replica reduction is 64 fp64 loads per channel: done per thread it
    // cost ~90 us of L2 traffic on every launch with >= 2048 blocks, whatever the tensor size)
    __shared__ ChanCoef s_k[1024];            // c <= 4 * THREADS (vec_ok)
    for (int ch = threadIdx.x; ch < c; ch += THREADS) s_k[ch] = coef_from_sums(sums, c, ch, rows, gamma, beta, eps);
    __syncthreads();
    ChanCoef k[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) k[i] = s_k[vcol * 4 + i];
    if (blockIdx.x == 0 && rsub == 0)
#pragma unroll
        for (int i = 0; i < 4; ++i) { mean_invstd[vcol * 4 + i] = k[i].mean; mean_invstd[c + vcol * 4 + i] = k[i].invstd; }
    for (long long r = (long long)blockIdx.x * g.rpb + rsub; r < rows; r += (long long)gridDim.x * g.rpb) {
        const float4 v = y[r * g.cv + vcol];
        float4 o;
        o.x = act_fwd((v.x - k[0].mean) * k[0].scale + k[0].beta, slope);
        o.y = act_fwd((v.y - k[1].mean) * k[1].scale + k[1].beta, slope);
        o.z = act_fwd((v.z - k[2].mean) * k[2].scale + k[2].beta, slope);
        o.w = act_fwd((v.w - k[3].mean) * k[3].scale + k[3].beta, slope);
        out[r * g.cv + vcol] = o;
    }
}

struct BwdCoef { float mean, invstd, scale, beta; };

__device__ __forceinline__ void dz_xhat(float yv, float go, const BwdCoef &k, float slope, float &dz, float &xh) {
    xh = (yv - k.mean) * k.invstd;
    const float z = (yv - k.mean) * k.scale + k.beta;      // identical expression to the forward
    dz = z > 0.f ? go : go * slope;
}

__global__ __launch_bounds__(THREADS) void bn_act_bwd_stats_v4(long long rows, int c, BnGeom g,
                                                                const float4 *__restrict__ dout,
                                                                const float4 *__restrict__ y,
                                                                const float *__restrict__ mean_invstd,
                                                                const float *__restrict__ gamma,
                                                                const float *__restrict__ beta, float slope,
                                                                double *__restrict__ dsums) {
    __shared__ double red[THREADS][8];
    const int vcol = threadIdx.x % g.cv, rsub = threadIdx.x / g.cv;
    BwdCoef k[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = vcol * 4 + i;
        k[i].mean = mean_invstd[ch]; k[i].invstd = mean_invstd[c + ch];
        k[i].scale = k[i].invstd * gamma[ch]; k[i].beta = beta[ch];
    }
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    const long long stride = (long long)gridDim.x * g.rpb;
    for (long long r0 = (long long)blockIdx.x * g.rpb + rsub; r0 < rows; r0 += stride * 2) {
        float4 v[2], go[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long r = r0 + u * stride;
            const bool ok = r < rows;
            v[u] = ok ? y[r * g.cv + vcol] : make_float4(0.f, 0.f, 0.f, 0.f);
            go[u] = ok ? dout[r * g.cv + vcol] : make_float4(0.f, 0.f, 0.f, 0.f);   // dz = 0: no contribution
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float dz, xh;
            dz_xhat(v[u].x, go[u].x, k[0], slope, dz, xh); s[0] += dz; q[0] += (double)dz * xh;
            dz_xhat(v[u].y, go[u].y, k[1], slope, dz, xh); s[1] += dz; q[1] += (double)dz * xh;
            dz_xhat(v[u].z, go[u].z, k[2], slope, dz, xh); s[2] += dz; q[2] += (double)dz * xh;
            dz_xhat(v[u].w, go[u].w, k[3], slope, dz, xh); s[3] += dz; q[3] += (double)dz * xh;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[threadIdx.x][i] = s[i]; red[threadIdx.x][4 + i] = q[i]; }
    __syncthreads();
    if (threadIdx.x < g.cv) {
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = threadIdx.x; t < THREADS; t += g.cv)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] += red[t][i];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double *rep = dsums + (size_t)(blockIdx.x % REP) * 2 * c;
            atomicAdd(rep + vcol * 4 + i, a[i]);
            atomicAdd(rep + c + vcol * 4 + i, a[4 + i]);
        }
    }
}

__global__ __launch_bounds__(THREADS) void bn_act_bwd_v4(long long rows, int c, BnGeom g,
                                                          const float4 *__restrict__ dout,
                                                          const float4 *__restrict__ y,
                                                          const float *__restrict__ mean_invstd,
                                                          const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float slope,
                                                          const double *__restrict__ dsums,
                                                          float4 *__restrict__ dy, float *__restrict__ dgamma,
                                                          float *__restrict__ dbeta) {
    const int vcol = threadIdx.x % g.cv, rsub = threadIdx.x / g.cv;
    __shared__ float s_m1[1024], s_m2[1024];               // replica sums reduced once per block (see bn_act_fwd_v4)
    for (int ch = threadIdx.x; ch < c; ch += THREADS) {
        const double sd = rep_sum(dsums, c, ch), sx = rep_sum(dsums, c, c + ch);
        s_m1[ch] = (float)(sd / (double)rows); s_m2[ch] = (float)(sx / (double)rows);
        if (blockIdx.x == 0) { dbeta[ch] = (float)sd; dgamma[ch] = (float)sx; }
    }
    __syncthreads();
    BwdCoef k[4];
    float m1[4], m2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = vcol * 4 + i;
        k[i].mean = mean_invstd[ch]; k[i].invstd = mean_invstd[c + ch];
        k[i].scale = k[i].invstd * gamma[ch]; k[i].beta = beta[ch];
        m1[i] = s_m1[ch]; m2[i] = s_m2[ch];
    }
    for (long long r = (long long)blockIdx.x * g.rpb + rsub; r < rows; r += (long long)gridDim.x * g.rpb) {
        const float4 v = y[r * g.cv + vcol], go = dout[r * g.cv + vcol];
        float4 o; float dz, xh;
        dz_xhat(v.x, go.x, k[0], slope, dz, xh); o.x = k[0].scale * (dz - m1[0] - xh * m2[0]);
        dz_xhat(v.y, go.y, k[1], slope, dz, xh); o.y = k[1].scale * (dz - m1[1] - xh * m2[1]);
        dz_xhat(v.z, go.z, k[2], slope, dz, xh); o.z = k[2].scale * (dz - m1[2] - xh * m2[2]);
        dz_xhat(v.w, go.w, k[3], slope, dz, xh); o.w = k[3].scale * (dz - m1[3] - xh * m2[3]);
        dy[r * g.cv + vcol] = o;
    }
}

// ---- BN + activation + max over the K neighbours of a group (set-abstraction tail) ---------------------------
// out[g,:] = max_k act(bn(y[g*K+k,:])), arg[g,:] = first k attaining it (PPBackbone_center.py:129,
// torch.max(new_points, dim=2)): the activated [groups*K, C] tensor is never written.  Thread per (group, float4).
__global__ __launch_bounds__(THREADS) void bn_act_maxk_fwd_v4(long long groups, int K, int c, const float4 *__restrict__ y,
                                                               const float *__restrict__ coef, float slope,
                                                               float4 *__restrict__ out, uchar4 *__restrict__ arg) {
    const int cv = c >> 2;
    const long long total = groups * cv;
    const int vcol = threadIdx.x % cv;
    const float4 mu = *reinterpret_cast<const float4 *>(coef + vcol * 4), sc = *reinterpret_cast<const float4 *>(coef + c + vcol * 4),
                 be = *reinterpret_cast<const float4 *>(coef + 2 * c + vcol * 4);
    for (long long t = (long long)blockIdx.x * THREADS + threadIdx.x; t < total; t += (long long)gridDim.x * THREADS) {
        const long long g = t / cv;
        float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        uchar4 bi = make_uchar4(0, 0, 0, 0);
        const float4 *src = y + g * K * cv + vcol;
        for (int k = 0; k < K; ++k) {
            const float4 v = src[(size_t)k * cv];
            const float ax = act_fwd((v.x - mu.x) * sc.x + be.x, slope), ay = act_fwd((v.y - mu.y) * sc.y + be.y, slope);
            const float az = act_fwd((v.z - mu.z) * sc.z + be.z, slope), aw = act_fwd((v.w - mu.w) * sc.w + be.w, slope);
            if (ax > best.x || ax != ax) { best.x = ax; bi.x = (unsigned char)k; }
            if (ay > best.y || ay != ay) { best.y = ay; bi.y = (unsigned char)k; }
            if (az > best.z || az != az) { best.z = az; bi.z = (unsigned char)k; }
            if (aw > best.w || aw != aw) { best.w = aw; bi.w = (unsigned char)k; }
        }
        out[t] = best; arg[t] = bi;
    }
}

// backward of the max: dense dL/da [groups*K, C] written in ONE pass (torch: zeros + scatter)
__global__ __launch_bounds__(THREADS) void unpool_k_v4(long long groups, int K, int c, const float4 *__restrict__ g,
                                                        const uchar4 *__restrict__ arg, float4 *__restrict__ gd) {
    const int cv = c >> 2;
    const long long total = groups * K * cv;
    for (long long t = (long long)blockIdx.x * THREADS + threadIdx.x; t < total; t += (long long)gridDim.x * THREADS) {
        const long long r = t / cv;
        const int vcol = (int)(t - r * cv);
        const long long grp = r / K;
        const unsigned char k = (unsigned char)(r - grp * K);
        const float4 gv = g[grp * cv + vcol];
        const uchar4 a = arg[grp * cv + vcol];
        gd[t] = make_float4(a.x == k ? gv.x : 0.f, a.y == k ? gv.y : 0.f, a.z == k ? gv.z : 0.f, a.w == k ? gv.w : 0.f);
    }
}

// unpool_k_v4 + bn_act_bwd_stats_v4 in one pass: only the arg-max row of a group carries gradient, so the statistics
// {sum gz, sum gz*xhat} (gz = dL/da * act'(bn(y))) need y at that row only — the separate statistics kernel re-read the dense
// [groups*K, C] gradient and y (two full passes at level 1: 2 x 118 MB).  One thread per (group, float4 column); THREADS % cv == 0.
__global__ __launch_bounds__(THREADS) void unpool_k_stats_v4(long long groups, int K, int c, const float4 *__restrict__ g,
                                                              const uchar4 *__restrict__ arg, const float *__restrict__ y,
                                                              const float *__restrict__ mean_invstd, const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, float slope, float4 *__restrict__ gd,
                                                              double *__restrict__ dsums) {
    __shared__ double red[THREADS][8];
    const int cv = c >> 2, vcol = threadIdx.x % cv;
    BwdCoef kc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = vcol * 4 + i;
        kc[i].mean = mean_invstd[ch]; kc[i].invstd = mean_invstd[c + ch];
        kc[i].scale = kc[i].invstd * gamma[ch]; kc[i].beta = beta[ch];
    }
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    const long long total = groups * cv;
    for (long long t = (long long)blockIdx.x * THREADS + threadIdx.x; t < total; t += (long long)gridDim.x * THREADS) {
        const long long grp = t / cv;                          // (t % cv == vcol: the grid stride is a multiple of cv)
        const float4 gv = g[t];
        const uchar4 a = arg[t];
        const float gq[4] = {gv.x, gv.y, gv.z, gv.w};
        const unsigned char aq[4] = {a.x, a.y, a.z, a.w};
        for (int k = 0; k < K; ++k)
            gd[(grp * K + k) * cv + vcol] = make_float4(a.x == k ? gv.x : 0.f, a.y == k ? gv.y : 0.f, a.z == k ? gv.z : 0.f, a.w == k ? gv.w : 0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float yv = y[((grp * K + aq[i]) * cv + vcol) * 4 + i];
            float dz, xh;
            dz_xhat(yv, gq[i], kc[i], slope, dz, xh);
            s[i] += dz; q[i] += (double)dz * xh;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[threadIdx.x][i] = s[i]; red[threadIdx.x][4 + i] = q[i]; }
    __syncthreads();
    if (threadIdx.x < cv) {
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = threadIdx.x; t < THREADS; t += cv)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += red[t][i];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double *rep = dsums + (size_t)(blockIdx.x % REP) * 2 * c;
            atomicAdd(rep + vcol * 4 + i, acc[i]);
            atomicAdd(rep + c + vcol * 4 + i, acc[4 + i]);
        }
    }
}

// ---- generic path (any C): one thread per element, channel = index % C ------------------------
__global__ void bn_stats_gen(long long total, int c, const float *__restrict__ y, double *__restrict__ sums) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const float v = y[i]; const int ch = (int)(i % c);
        double *rep = sums + (size_t)(blockIdx.x % REP) * 2 * c;
        atomicAdd(rep + ch, (double)v); atomicAdd(rep + c + ch, (double)v * v);
    }
}
__global__ void bn_act_fwd_gen(long long total, long long rows, int c, const float *__restrict__ y,
                               const double *__restrict__ sums, const float *__restrict__ gamma,
                               const float *__restrict__ beta, float eps, float slope, float *__restrict__ out,
                               float *__restrict__ mean_invstd) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % c);
        const ChanCoef k = coef_from_sums(sums, c, ch, rows, gamma, beta, eps);
        if (i < c) { mean_invstd[ch] = k.mean; mean_invstd[c + ch] = k.invstd; }
        out[i] = act_fwd((y[i] - k.mean) * k.scale + k.beta, slope);
    }
}
__global__ void bn_act_bwd_stats_gen(long long total, int c, const float *__restrict__ dout,
                                     const float *__restrict__ y, const float *__restrict__ mean_invstd,
                                     const float *__restrict__ gamma, const float *__restrict__ beta, float slope,
                                     double *__restrict__ dsums) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % c);
        BwdCoef k; k.mean = mean_invstd[ch]; k.invstd = mean_invstd[c + ch]; k.scale = k.invstd * gamma[ch]; k.beta = beta[ch];
        float dz, xh; dz_xhat(y[i], dout[i], k, slope, dz, xh);
        double *rep = dsums + (size_t)(blockIdx.x % REP) * 2 * c;
        atomicAdd(rep + ch, (double)dz); atomicAdd(rep + c + ch, (double)dz * xh);
    }
}
__global__ void bn_act_bwd_gen(long long total, long long rows, int c, const float *__restrict__ dout,
                               const float *__restrict__ y, const float *__restrict__ mean_invstd,
                               const float *__restrict__ gamma, const float *__restrict__ beta, float slope,
                               const double *__restrict__ dsums, float *__restrict__ dy, float *__restrict__ dgamma,
                               float *__restrict__ dbeta) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % c);
        BwdCoef k; k.mean = mean_invstd[ch]; k.invstd = mean_invstd[c + ch]; k.scale = k.invstd * gamma[ch]; k.beta = beta[ch];
        const double sd = rep_sum(dsums, c, ch), sx = rep_sum(dsums, c, c + ch);
        if (i < c) { dbeta[ch] = (float)sd; dgamma[ch] = (float)sx; }
        float dz, xh; dz_xhat(y[i], dout[i], k, slope, dz, xh);
        dy[i] = k.scale * (dz - (float)(sd / (double)rows) - xh * (float)(sx / (double)rows));
    }
}

inline bool vec_ok(int c, const void *a, const void *b) {
    if (c % 4 != 0) return false;
    const int cv = c / 4;
    if (cv > THREADS || THREADS % cv != 0) return false;
    return ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0);
}
inline BnGeom geom(int c) { BnGeom g; g.cv = c / 4; g.rpb = THREADS / g.cv; return g; }
inline unsigned grid_for(long long rows, int rpb) {
    long long b = (rows + rpb - 1) / rpb;
    return (unsigned)(b < 1 ? 1 : (b > MAX_BLOCKS ? MAX_BLOCKS : b));
}
inline unsigned grid_stat(long long rows, int rpb, int unroll) {
    long long b = (rows + (long long)rpb * unroll - 1) / ((long long)rpb * unroll);
    return (unsigned)(b < 1 ? 1 : (b > MAX_STAT_BLOCKS ? MAX_STAT_BLOCKS : b));
}
inline unsigned grid_gen(long long total) {
    long long b = (total + THREADS - 1) / THREADS;
    return (unsigned)(b < 1 ? 1 : (b > MAX_BLOCKS ? MAX_BLOCKS : b));
}

}  // namespace

extern "C" int i2p_bn_stats(long long rows, int c, const float *y, double *sums, void *stream) {
    if (rows < 0 || c <= 0) return I2P_ERR_BAD_ARG;
    if (rows == 0) return 0;
    if (!y || !sums) return I2P_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (vec_ok(c, y, y)) {
        const BnGeom g = geom(c);
        hipLaunchKernelGGL(bn_stats_v4, dim3(grid_stat(rows, g.rpb, UNROLL)), dim3(THREADS), 0, st, rows, c, g,
                           (const float4 *)y, sums);
    } else {
        hipLaunchKernelGGL(bn_stats_gen, dim3(grid_gen(rows * c)), dim3(THREADS), 0, st, rows * c, c, y, sums);
    }
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_bn_act_fwd(long long rows, int c, const float *y, const double *sums,
                              const float *gamma, const float *beta, float eps, float slope,
                              float *out, float *mean_invstd, void *stream) {
    if (rows < 0 || c <= 0) return I2P_ERR_BAD_ARG;
    if (rows == 0) return 0;
    if (!y || !sums || !gamma || !beta || !out || !mean_invstd) return I2P_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (vec_ok(c, y, out)) {
        const BnGeom g = geom(c);
        hipLaunchKernelGGL(bn_act_fwd_v4, dim3(grid_for(rows, g.rpb)), dim3(THREADS), 0, st, rows, c, g,
                           (const float4 *)y, sums, gamma, beta, eps, slope, (float4 *)out, mean_invstd);
    } else {
        hipLaunchKernelGGL(bn_act_fwd_gen, dim3(grid_gen(rows * c)), dim3(THREADS), 0, st, rows * c, rows, c, y,
                           sums, gamma, beta, eps, slope, out, mean_invstd);
    }
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_bn_act_bwd_stats(long long rows, int c, const float *dout, const float *y,
                                    const float *mean_invstd, const float *gamma, const float *beta,
                                    float slope, double *dsums, void *stream) {
    if (rows < 0 || c <= 0) return I2P_ERR_BAD_ARG;
    if (rows == 0) return 0;
    if (!dout || !y || !mean_invstd || !gamma || !beta || !dsums) return I2P_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (vec_ok(c, y, dout)) {
        const BnGeom g = geom(c);
        hipLaunchKernelGGL(bn_act_bwd_stats_v4, dim3(grid_stat(rows, g.rpb, 2)), dim3(THREADS), 0, st, rows, c, g,
                           (const float4 *)dout, (const float4 *)y, mean_invstd, gamma, beta, slope, dsums);
    } else {
        hipLaunchKernelGGL(bn_act_bwd_stats_gen, dim3(grid_gen(rows * c)), dim3(THREADS), 0, st, rows * c, c, dout, y,
                           mean_invstd, gamma, beta, slope, dsums);
    }
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_bn_act_bwd(long long rows, int c, const float *dout, const float *y,
                              const float *mean_invstd, const float *gamma, const float *beta,
                              float slope, const double *dsums, float *dy, float *dgamma, float *dbeta,
                              void *stream) {
    if (rows < 0 || c <= 0) return I2P_ERR_BAD_ARG;
    if (rows == 0) return 0;
    if (!dout || !y || !mean_invstd || !gamma || !beta || !dsums || !dy || !dgamma || !dbeta) return I2P_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (vec_ok(c, y, dout) && ((uintptr_t)dy % 16 == 0)) {
        const BnGeom g = geom(c);
        hipLaunchKernelGGL(bn_act_bwd_v4, dim3(grid_for(rows, g.rpb)), dim3(THREADS), 0, st, rows, c, g,
                           (const float4 *)dout, (const float4 *)y, mean_invstd, gamma, beta, slope, dsums,
                           (float4 *)dy, dgamma, dbeta);
    } else {
        hipLaunchKernelGGL(bn_act_bwd_gen, dim3(grid_gen(rows * c)), dim3(THREADS), 0, st, rows * c, rows, c, dout, y,
                           mean_invstd, gamma, beta, slope, dsums, dy, dgamma, dbeta);
    }
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_bn_act_maxk_fwd(long long groups, int K, int c, const float *y, const float *coef, float slope,
                                   float *out, unsigned char *arg, void *stream) {
    if (groups < 0 || K <= 0 || K > 255 || c <= 0 || (c & 3) || THREADS % (c >> 2)) return I2P_ERR_BAD_ARG;
    if (groups == 0) return 0;
    if (!y || !coef || !out || !arg) return I2P_ERR_BAD_ARG;
    const long long total = groups * (c >> 2);
    long long blocks = (total + THREADS - 1) / THREADS;
    if (blocks > (1 << 20)) blocks = 1 << 20;
    hipLaunchKernelGGL(bn_act_maxk_fwd_v4, dim3((unsigned)blocks), dim3(THREADS), 0, (hipStream_t)stream, groups, K, c,
                       (const float4 *)y, coef, slope, (float4 *)out, (uchar4 *)arg);
    I2P_RETURN_LAUNCH_STATUS();
}

// i2p_unpool_k + i2p_bn_act_bwd_stats of the dense result in one launch (dsums zeroed by the caller)
extern "C" int i2p_unpool_k_stats(long long groups, int K, int c, const float *g, const unsigned char *arg, const float *y,
                                  const float *mean_invstd, const float *gamma, const float *beta, float slope, float *gd, double *dsums,
                                  void *stream) {
    if (groups < 0 || K <= 0 || K > 255 || c <= 0 || (c & 3) || THREADS % (c >> 2)) return I2P_ERR_BAD_ARG;
    if (groups == 0) return 0;
    if (!g || !arg || !y || !mean_invstd || !gamma || !beta || !gd || !dsums) return I2P_ERR_BAD_ARG;
    const long long total = groups * (c >> 2);
    long long blocks = (total + THREADS - 1) / THREADS;
    if (blocks > MAX_STAT_BLOCKS) blocks = MAX_STAT_BLOCKS;
    hipLaunchKernelGGL(unpool_k_stats_v4, dim3((unsigned)blocks), dim3(THREADS), 0, (hipStream_t)stream, groups, K, c, (const float4 *)g,
                       (const uchar4 *)arg, y, mean_invstd, gamma, beta, slope, (float4 *)gd, dsums);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_unpool_k(long long groups, int K, int c, const float *g, const unsigned char *arg, float *gd,
                            void *stream) {
    if (groups < 0 || K <= 0 || K > 255 || c <= 0 || (c & 3)) return I2P_ERR_BAD_ARG;
    if (groups == 0) return 0;
    if (!g || !arg || !gd) return I2P_ERR_BAD_ARG;
    const long long total = groups * K * (c >> 2);
    long long blocks = (total + THREADS - 1) / THREADS;
    if (blocks > (1 << 20)) blocks = 1 << 20;
    hipLaunchKernelGGL(unpool_k_v4, dim3((unsigned)blocks), dim3(THREADS), 0, (hipStream_t)stream, groups, K, c,
                       (const float4 *)g, (const uchar4 *)arg, (float4 *)gd);
    I2P_RETURN_LAUNCH_STATUS();
}
