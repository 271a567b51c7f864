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

}  // namespace

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
