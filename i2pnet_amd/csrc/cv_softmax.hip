// Softmax-over-pixels weighted sum of the cost volume's pi-stage
// (src/projectPN/PPBackbone_center.py:430-433:  WQ = softmax(pi_concat, dim=2); sum(WQ * pi_feat1_new, dim=2))
// fused with the BN + LeakyReLU of both operands: it consumes the PRE-BN tensors y5 (logits branch) and
// y3 (value branch) plus their BN coefficients, so neither activation tensor nor the softmax is materialised.
//   forward : out[b,n,c] = sum_k softmax_k(h5[b,n,k,c]) * h3[b,n,k,c],   h = act(bn(y))
//   backward: gz5 = s * g_out * (h3 - out) * act'(z5)   (+ BN-backward statistics of the logits BN)
//             ga3 = g_out * s                            (dL/dh3 of this consumer; the layer-4 dgrad adds its own)
// One block per point (b,n): thread = (channel, k-phase); a point's 468 x 64 logits stay L2-resident between the
// two sweeps of the backward.
#include "common.h"

namespace {

constexpr int THREADS = 256;
constexpr int REP = I2P_BN_REPLICAS;

__device__ __forceinline__ float act_f(float z, float slope) { return z > 0.f ? z : z * slope; }

struct SmParams {
    int B, N, M, C;
    const float *y5, *coef5, *mi5, *y3, *coef3;
    float slope5, slope3;
    const float *g_out;
    float *out, *msave;        // msave [B*N, 2, C]: running max and exp-sum of the logits
    float *gz5, *ga3;
    double *dsums5;
};

__global__ __launch_bounds__(THREADS) void sm_fwd_kernel(SmParams p) {
    __shared__ float sm[THREADS], ss[THREADS], sa[THREADS];
    const int bn = blockIdx.x, tid = threadIdx.x;
    const int c = tid % p.C, kg = tid / p.C, KG = THREADS / p.C;
    const float m5 = p.coef5[c], s5 = p.coef5[p.C + c], b5 = p.coef5[2 * p.C + c];
    const float m3 = p.coef3[c], s3 = p.coef3[p.C + c], b3 = p.coef3[2 * p.C + c];
    const size_t base = (size_t)bn * p.M * p.C + c;
    float mx = -INFINITY, S = 0.f, A = 0.f;
    for (int k = kg; k < p.M; k += KG) {
        const float h5 = act_f((p.y5[base + (size_t)k * p.C] - m5) * s5 + b5, p.slope5);
        const float h3 = act_f((p.y3[base + (size_t)k * p.C] - m3) * s3 + b3, p.slope3);
        const float mn = fmaxf(mx, h5);
        const float sc = __expf(mx - mn), e = __expf(h5 - mn);
        S = S * sc + e; A = A * sc + e * h3; mx = mn;
    }
    sm[tid] = mx; ss[tid] = S; sa[tid] = A;
    __syncthreads();
    if (kg == 0) {
        float M_ = mx;
        for (int g = 1; g < KG; ++g) M_ = fmaxf(M_, sm[g * p.C + c]);
        float St = 0.f, At = 0.f;
        for (int g = 0; g < KG; ++g) {
            const float w = __expf(sm[g * p.C + c] - M_);
            St += ss[g * p.C + c] * w; At += sa[g * p.C + c] * w;
        }
        p.out[(size_t)bn * p.C + c] = At / St;
        p.msave[(size_t)bn * 2 * p.C + c] = M_; p.msave[(size_t)bn * 2 * p.C + p.C + c] = St;
    }
}

__global__ __launch_bounds__(THREADS) void sm_bwd_kernel(SmParams p) {
    __shared__ double rs[THREADS], rq[THREADS];
    const int bn = blockIdx.x, tid = threadIdx.x;
    const int c = tid % p.C, kg = tid / p.C, KG = THREADS / p.C;
    const float m5 = p.coef5[c], s5 = p.coef5[p.C + c], b5 = p.coef5[2 * p.C + c], is5 = p.mi5[p.C + c];
    const float m3 = p.coef3[c], s3 = p.coef3[p.C + c], b3 = p.coef3[2 * p.C + c];
    const float M_ = p.msave[(size_t)bn * 2 * p.C + c], inv = 1.f / p.msave[(size_t)bn * 2 * p.C + p.C + c];
    const float go = p.g_out[(size_t)bn * p.C + c], o = p.out[(size_t)bn * p.C + c];
    const size_t base = (size_t)bn * p.M * p.C + c;
    double ds = 0.0, dq = 0.0;
    for (int k = kg; k < p.M; k += KG) {
        const float y5v = p.y5[base + (size_t)k * p.C];
        const float z5 = (y5v - m5) * s5 + b5;
        const float h5 = act_f(z5, p.slope5);
        const float h3 = act_f((p.y3[base + (size_t)k * p.C] - m3) * s3 + b3, p.slope3);
        const float s = __expf(h5 - M_) * inv;
        float g5 = s * go * (h3 - o);
        g5 = z5 > 0.f ? g5 : g5 * p.slope5;
        p.gz5[base + (size_t)k * p.C] = g5;
        p.ga3[base + (size_t)k * p.C] = go * s;
        ds += g5; dq += (double)g5 * ((y5v - m5) * is5);
    }
    rs[tid] = ds; rq[tid] = dq;
    __syncthreads();
    if (kg == 0) {
        for (int g = 1; g < KG; ++g) { ds += rs[g * p.C + c]; dq += rq[g * p.C + c]; }
        double *rep = p.dsums5 + (size_t)(blockIdx.x % REP) * 2 * p.C;
        atomicAdd(rep + c, ds); atomicAdd(rep + p.C + c, dq);
    }
}

// ---------------------------------------------------------------------------------------------------
// Gradient of the position encoding of the all-pixel cost volume.  Its pre-BN tensor is an outer sum
//     ye[b,n,k,:] = enc_n[b,n,:] + enc_k[b,k,:]                                 (PPBackbone_center.py:416-418)
// so the gradients of the two small factors are the k- and n-sums of dL/dye.  pair_sum reads gz_e ONCE and
// produces both sums; the BN backward dL/dye = scale*(gz - m1 - xhat*m2) is then applied in closed form on
// the [B,N,C] / [B,M,C] factors (sum_k xhat = invstd*(M*enc_n + sum_k enc_k - M*mean)): the [B,N,M,C]
// gradient of ye is never written and never re-read by two reductions.
// ---------------------------------------------------------------------------------------------------
constexpr int PS_NL = 16;         // point rows per block (halves the sum_n atomics of 8)

__global__ __launch_bounds__(THREADS) void pair_sum_kernel(int B, int N, int M, int C, const float4 *__restrict__ g,
                                                            float *__restrict__ sum_k, float *__restrict__ sum_n) {
    __shared__ float4 red[THREADS][PS_NL];                     // 64 KB
    const int cv = C >> 2, ks = THREADS / cv;                   // k-slots per block
    const int c4 = threadIdx.x % cv, kslot = threadIdx.x / cv;
    const int chunks = (N + PS_NL - 1) / PS_NL;
    const int b = blockIdx.x / chunks, n0 = (blockIdx.x - b * chunks) * PS_NL;
    const int kseg = (M + gridDim.y - 1) / gridDim.y, k_lo = blockIdx.y * kseg, k_hi = min(M, k_lo + kseg);
    float4 acc[PS_NL];
#pragma unroll
    for (int j = 0; j < PS_NL; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = k_lo + kslot; k < k_hi; k += ks) {
        float4 v[PS_NL];
#pragma unroll
        for (int j = 0; j < PS_NL; ++j) {
            const int n = n0 + j;
            v[j] = n < N ? g[(((size_t)b * N + n) * M + k) * cv + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < PS_NL; ++j) {
            acc[j].x += v[j].x; acc[j].y += v[j].y; acc[j].z += v[j].z; acc[j].w += v[j].w;
            t.x += v[j].x; t.y += v[j].y; t.z += v[j].z; t.w += v[j].w;
        }
        float *dst = sum_n + ((size_t)b * M + k) * C + c4 * 4;
        atomicAdd(dst + 0, t.x); atomicAdd(dst + 1, t.y); atomicAdd(dst + 2, t.z); atomicAdd(dst + 3, t.w);
    }
#pragma unroll
    for (int j = 0; j < PS_NL; ++j) red[threadIdx.x][j] = acc[j];
    __syncthreads();
    for (int i = threadIdx.x; i < PS_NL * cv; i += THREADS) {
        const int j = i / cv, c = i - j * cv;
        if (n0 + j >= N) continue;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = 0; q < ks; ++q) { const float4 r = red[q * cv + c][j]; a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w; }
        float *dk = sum_k + ((size_t)b * N + n0 + j) * C + c * 4;               // k-range slices add up (zeroed by the caller)
        atomicAdd(dk + 0, a.x); atomicAdd(dk + 1, a.y); atomicAdd(dk + 2, a.z); atomicAdd(dk + 3, a.w);
    }
}

// Deterministic variant of pair_sum: the same sweep, but every block STORES its partial sums into its own slab
// (sum_n: one slab per point chunk, sum_k: one per k segment) instead of adding them with fp32 atomics; the closed-form
// kernel below adds the slabs up in slab order.
constexpr int PSD_KSEG = 4;
__global__ __launch_bounds__(THREADS) void pair_sum_det_kernel(int B, int N, int M, int C, const float4 *__restrict__ g,
                                                                float *__restrict__ slab_k, float *__restrict__ slab_n) {
    __shared__ float4 red[THREADS][PS_NL];
    const int cv = C >> 2, ks = THREADS / cv;
    const int c4 = threadIdx.x % cv, kslot = threadIdx.x / cv;
    const int chunks = (N + PS_NL - 1) / PS_NL;
    const int b = blockIdx.x / chunks, chunk = blockIdx.x - b * chunks, n0 = chunk * PS_NL;
    const int kseg = (M + gridDim.y - 1) / gridDim.y, k_lo = blockIdx.y * kseg, k_hi = min(M, k_lo + kseg);
    float4 acc[PS_NL];
#pragma unroll
    for (int j = 0; j < PS_NL; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = k_lo + kslot; k < k_hi; k += ks) {
        float4 v[PS_NL];
#pragma unroll
        for (int j = 0; j < PS_NL; ++j) {
            const int n = n0 + j;
            v[j] = n < N ? g[(((size_t)b * N + n) * M + k) * cv + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < PS_NL; ++j) {
            acc[j].x += v[j].x; acc[j].y += v[j].y; acc[j].z += v[j].z; acc[j].w += v[j].w;
            t.x += v[j].x; t.y += v[j].y; t.z += v[j].z; t.w += v[j].w;
        }
        *reinterpret_cast<float4 *>(slab_n + (((size_t)chunk * B + b) * M + k) * C + c4 * 4) = t;      // unique writer
    }
#pragma unroll
    for (int j = 0; j < PS_NL; ++j) red[threadIdx.x][j] = acc[j];
    __syncthreads();
    for (int i = threadIdx.x; i < PS_NL * cv; i += THREADS) {
        const int j = i / cv, c = i - j * cv;
        if (n0 + j >= N) continue;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = 0; q < ks; ++q) { const float4 r = red[q * cv + c][j]; a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w; }
        *reinterpret_cast<float4 *>(slab_k + (((size_t)blockIdx.y * B + b) * N + n0 + j) * C + c * 4) = a;
    }
}

// sums the slabs (fixed order) into the two [B,.,C] arrays the closed-form kernel reads
__global__ void pair_slab_sum_kernel(int nslab, long long n, const float *__restrict__ slabs, float *__restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float a = 0.f;
        for (int s = 0; s < nslab; ++s) a += slabs[(size_t)s * n + i];
        out[i] = a;
    }
}

// PB_SLICES blocks per batch sample: E_k = sum_k enc_k[b], E_n = sum_n enc_n[b] (recomputed per block: tiny),
// then the closed-form BN backward on this block's slice of the two factors
constexpr int PB_SLICES = 8;

__global__ __launch_bounds__(THREADS) void pair_bias_bn_bwd_kernel(int B, int N, int M, int C, const float *__restrict__ sum_k,
                                                                    const float *__restrict__ sum_n,
                                                                    const float *__restrict__ enc_n,
                                                                    const float *__restrict__ enc_k,
                                                                    const double *__restrict__ dsums,
                                                                    const float *__restrict__ coef, const float *__restrict__ mi,
                                                                    float *__restrict__ d_enc_n, float *__restrict__ d_enc_k) {
    __shared__ float part[2][THREADS];
    __shared__ float Ek[256], En[256], m1s[256], m2s[256];
    const int b = blockIdx.x / PB_SLICES, slice = blockIdx.x % PB_SLICES;
    const int G = THREADS / C, grp = threadIdx.x / C, c0 = threadIdx.x % C;      // C divides THREADS
    const double rows = (double)B * N * M;
    // 8 independent loads in flight per thread (a plain `e += x[k]` loop issued them one L2 latency apart: 70 us)
    auto colsum = [&](const float *src, int rows_) {
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int k = grp;
        for (; k + 7 * G < rows_; k += 8 * G) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[((size_t)b * rows_ + k + u * G) * C + c0];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += v[u];
        }
        for (; k < rows_; k += G) a[0] += src[((size_t)b * rows_ + k) * C + c0];
        return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    };
    const float ek = colsum(enc_k, M), en = colsum(enc_n, N);
    part[0][threadIdx.x] = ek; part[1][threadIdx.x] = en;
    __syncthreads();
    if (threadIdx.x < C) {
        float a = 0.f, d = 0.f;
        for (int q = 0; q < G; ++q) { a += part[0][q * C + threadIdx.x]; d += part[1][q * C + threadIdx.x]; }
        Ek[threadIdx.x] = a; En[threadIdx.x] = d;
        double sd = 0.0, sx = 0.0;
        for (int r = 0; r < I2P_BN_REPLICAS; ++r) { sd += dsums[(size_t)r * 2 * C + threadIdx.x]; sx += dsums[(size_t)r * 2 * C + C + threadIdx.x]; }
        m1s[threadIdx.x] = (float)(sd / rows); m2s[threadIdx.x] = (float)(sx / rows);
    }
    __syncthreads();
    for (int i = slice * THREADS + threadIdx.x; i < N * C; i += PB_SLICES * THREADS) {
        const int c = i % C;
        const float scale = coef[C + c], mean = mi[c], is = mi[C + c];
        const float sxh = is * ((float)M * enc_n[(size_t)b * N * C + i] + Ek[c] - (float)M * mean);
        d_enc_n[(size_t)b * N * C + i] = scale * (sum_k[(size_t)b * N * C + i] - (float)M * m1s[c] - m2s[c] * sxh);
    }
    for (int i = slice * THREADS + threadIdx.x; i < M * C; i += PB_SLICES * THREADS) {
        const int c = i % C;
        const float scale = coef[C + c], mean = mi[c], is = mi[C + c];
        const float sxh = is * ((float)N * enc_k[(size_t)b * M * C + i] + En[c] - (float)N * mean);
        d_enc_k[(size_t)b * M * C + i] = scale * (sum_n[(size_t)b * M * C + i] - (float)N * m1s[c] - m2s[c] * sxh);
    }
}

}  // namespace

extern "C" int i2p_pair_bias_bn_bwd(int B, int N, int M, int C, const float *gz, const float *enc_n, const float *enc_k,
                                    const double *dsums, const float *coef, const float *mi, float *sum_k, float *sum_n,
                                    float *d_enc_n, float *d_enc_k, void *stream) {
    if (B <= 0 || N <= 0 || M <= 0 || C <= 0 || C > 256 || (C & 3) || THREADS % C) return I2P_ERR_BAD_ARG;
    if (!gz || !enc_n || !enc_k || !dsums || !coef || !mi || !sum_k || !sum_n || !d_enc_n || !d_enc_k) return I2P_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int chunks = (N + PS_NL - 1) / PS_NL;
    // k range split 4 ways: 4x the blocks (one block per CU left most of the HBM bandwidth unused)
    hipLaunchKernelGGL(pair_sum_kernel, dim3(B * chunks, 4), dim3(THREADS), 0, st, B, N, M, C, (const float4 *)gz, sum_k, sum_n);
    hipLaunchKernelGGL(pair_bias_bn_bwd_kernel, dim3(B * PB_SLICES), dim3(THREADS), 0, st, B, N, M, C, sum_k, sum_n, enc_n, enc_k, dsums,
                       coef, mi, d_enc_n, d_enc_k);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" long long i2p_pair_bias_bn_bwd_scratch(int B, int N, int M, int C) {
    const long long chunks = (N + PS_NL - 1) / PS_NL;
    return (long long)B * N * C * (PSD_KSEG + 1) + (long long)B * M * C * (chunks + 1);
}

// i2p_pair_bias_bn_bwd without atomics: scratch holds the slabs and the two summed arrays (i2p_pair_bias_bn_bwd_scratch floats)
extern "C" int i2p_pair_bias_bn_bwd_det(int B, int N, int M, int C, const float *gz, const float *enc_n, const float *enc_k,
                                        const double *dsums, const float *coef, const float *mi, float *scratch,
                                        float *d_enc_n, float *d_enc_k, void *stream) {
    if (B <= 0 || N <= 0 || M <= 0 || C <= 0 || C > 256 || (C & 3) || THREADS % C) return I2P_ERR_BAD_ARG;
    if (!gz || !enc_n || !enc_k || !dsums || !coef || !mi || !scratch || !d_enc_n || !d_enc_k) return I2P_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int chunks = (N + PS_NL - 1) / PS_NL;
    const long long nk = (long long)B * N * C, nn = (long long)B * M * C;
    float *slab_k = scratch, *sum_k = slab_k + PSD_KSEG * nk, *slab_n = sum_k + nk, *sum_n = slab_n + (long long)chunks * nn;
    hipLaunchKernelGGL(pair_sum_det_kernel, dim3(B * chunks, PSD_KSEG), dim3(THREADS), 0, st, B, N, M, C, (const float4 *)gz, slab_k, slab_n);
    hipLaunchKernelGGL(pair_slab_sum_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, st, PSD_KSEG, nk, slab_k, sum_k);
    hipLaunchKernelGGL(pair_slab_sum_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, chunks, nn, slab_n, sum_n);
    hipLaunchKernelGGL(pair_bias_bn_bwd_kernel, dim3(B * PB_SLICES), dim3(THREADS), 0, st, B, N, M, C, sum_k, sum_n, enc_n, enc_k, dsums,
                       coef, mi, d_enc_n, d_enc_k);
    I2P_RETURN_LAUNCH_STATUS();
}

// second half of i2p_pair_bias_bn_bwd on already formed k-/n-sums (shared with the bf16 entry, whose first half reads a
// bf16 gradient tensor)
extern "C" int i2p_pair_bias_bn_finish(int B, int N, int M, int C, const float *sum_k, const float *sum_n, const float *enc_n,
                                       const float *enc_k, const double *dsums, const float *coef, const float *mi,
                                       float *d_enc_n, float *d_enc_k, void *stream) {
    if (B <= 0 || N <= 0 || M <= 0 || C <= 0 || C > 256 || (C & 3) || THREADS % C) return I2P_ERR_BAD_ARG;
    hipLaunchKernelGGL(pair_bias_bn_bwd_kernel, dim3(B * PB_SLICES), dim3(THREADS), 0, (hipStream_t)stream, B, N, M, C, sum_k, sum_n,
                       enc_n, enc_k, dsums, coef, mi, d_enc_n, d_enc_k);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_cv_softmax_wsum_fwd(int B, int N, int M, int C, const float *y5, const float *coef5, float slope5,
                                       const float *y3, const float *coef3, float slope3, float *out, float *msave,
                                       void *stream) {
    if (B <= 0 || N <= 0 || M <= 0 || C <= 0 || C > 256 || 256 % C) return I2P_ERR_BAD_ARG;
    if (!y5 || !coef5 || !y3 || !coef3 || !out || !msave) return I2P_ERR_BAD_ARG;
    SmParams p{}; p.B = B; p.N = N; p.M = M; p.C = C; p.y5 = y5; p.coef5 = coef5; p.slope5 = slope5; p.y3 = y3;
    p.coef3 = coef3; p.slope3 = slope3; p.out = out; p.msave = msave;
    hipLaunchKernelGGL(sm_fwd_kernel, dim3(B * N), dim3(THREADS), 0, (hipStream_t)stream, p);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_cv_softmax_wsum_bwd(int B, int N, int M, int C, const float *g_out, const float *out,
                                       const float *msave, const float *y5, const float *coef5, const float *mi5,
                                       float slope5, const float *y3, const float *coef3, float slope3, float *gz5,
                                       double *dsums5, float *ga3, void *stream) {
    if (B <= 0 || N <= 0 || M <= 0 || C <= 0 || C > 256 || 256 % C) return I2P_ERR_BAD_ARG;
    if (!g_out || !out || !msave || !y5 || !coef5 || !mi5 || !y3 || !coef3 || !gz5 || !dsums5 || !ga3) return I2P_ERR_BAD_ARG;
    SmParams p{}; p.B = B; p.N = N; p.M = M; p.C = C; p.y5 = y5; p.coef5 = coef5; p.mi5 = mi5; p.slope5 = slope5;
    p.y3 = y3; p.coef3 = coef3; p.slope3 = slope3; p.g_out = g_out; p.out = const_cast<float *>(out);
    p.msave = const_cast<float *>(msave); p.gz5 = gz5; p.ga3 = ga3; p.dsums5 = dsums5;
    hipLaunchKernelGGL(sm_bwd_kernel, dim3(B * N), dim3(THREADS), 0, (hipStream_t)stream, p);
    I2P_RETURN_LAUNCH_STATUS();
}
