// Global-norm gradient clip + Adam (L2 weight decay) on the trainer's flat fp32 buffers in TWO launches
// (reference: train20v2learn_wandb_proj.py:198-205 torch.optim.Adam lr 1e-3 betas (0.9, 0.999) eps 1e-8 weight_decay 1e-4,
// :472-476 clip_grad_norm_(10) then optimizer.step()).  The eager formulation of i2pnet_amd/train.py::FlatAdam is ~25
// elementwise launches over the same 3.4 MB; every one of them sits on the step's critical path inside the hipGraph.
//
//   launch 1 (adam_norm_kernel):   per-block fp64 partial sums of (gscale*g)^2 in a fixed order; block 0 advances the step counter
//   launch 2 (adam_update_kernel): every block re-adds the <= 256 partials in index order (bit-identical total on every block),
//                                  forms the clip factor, and applies the update to its slice
//
// Arithmetic follows torch.optim.Adam's single-tensor path operation by operation (fp32, no contraction: the library is
// built with -ffp-contract=off):  g = g*gscale*clip;  g += wd*p;  m = m + (1-b1)*(g-m);  v = v*b2 + (1-b2)*g*g;
// denom = sqrt(v)/sqrt(1-b2^t) + eps;  p -= (m/denom) * (lr/(1-b1^t)).
#include "common.h"

namespace {

constexpr int ADAM_THREADS = 256;
constexpr int ADAM_MAX_BLOCKS = 256;

__global__ __launch_bounds__(ADAM_THREADS) void adam_norm_kernel(long long n4, const float4 *__restrict__ grad, float gscale,
                                                                 double *__restrict__ partials, float *__restrict__ step,
                                                                 const float *__restrict__ poison) {
    __shared__ double red[ADAM_THREADS / 64];
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * ADAM_THREADS + threadIdx.x; i < n4; i += (long long)gridDim.x * ADAM_THREADS) {
        const float4 g = grad[i];
        const float a = g.x * gscale, b = g.y * gscale, c = g.z * gscale, d = g.w * gscale;
        acc += (double)a * a + (double)b * b + (double)c * c + (double)d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < ADAM_THREADS / 64; ++w) t += red[w];
        partials[blockIdx.x] = t;
        if (blockIdx.x == 0 && !(poison && poison[0] != 0.f)) step[0] = step[0] + 1.0f;     // (launch 2 reads the advanced counter)
    }
}

__global__ __launch_bounds__(ADAM_THREADS) void adam_update_kernel(long long n4, float4 *__restrict__ param, float4 *__restrict__ grad,
                                                                   float4 *__restrict__ m, float4 *__restrict__ v,
                                                                   const float4 *__restrict__ mask, const double *__restrict__ partials,
                                                                   int nparts, const float *__restrict__ step, const float *__restrict__ lr,
                                                                   float b1, float b2, float w1, float w2, float eps, float wd, float clip,
                                                                   float gscale, float *__restrict__ total_out,
                                                                   const float *__restrict__ poison) {
    if (poison && poison[0] != 0.f) return;         // gradients of an abandoned grid barrier (mlp_chain.hip): the step is not applied
    __shared__ float s_scale, s_bc2s, s_lrbc1;
    __shared__ double s_tot;
    if (threadIdx.x < 64) {                 // the <= 256 partials: four per lane in index order, then a fixed shuffle tree (same result in every block)
        double t = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = threadIdx.x * 4 + u; if (i < nparts) t += partials[i]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (threadIdx.x == 0) s_tot = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t = s_tot;
        const float total = (float)sqrt(t);
        float sc = 1.0f;
        if (clip > 0.f) sc = fminf(clip / (total + 1e-6f), 1.0f);       // torch.clamp(clip / (total + 1e-6), max=1)
        const float st = step[0];
        const float bc1 = 1.0f - powf(b1, st);
        s_bc2s = sqrtf(1.0f - powf(b2, st));
        s_lrbc1 = lr[0] / bc1;
        s_scale = sc;
        if (blockIdx.x == 0 && total_out) total_out[0] = total;
    }
    __syncthreads();
    const float sc = s_scale, bc2s = s_bc2s, lrbc1 = s_lrbc1;
    for (long long i = (long long)blockIdx.x * ADAM_THREADS + threadIdx.x; i < n4; i += (long long)gridDim.x * ADAM_THREADS) {
        float4 g4 = grad[i], p4 = param[i], m4 = m[i], v4 = v[i];
        float4 k4 = mask ? mask[i] : make_float4(1.f, 1.f, 1.f, 1.f);
        float *g = &g4.x, *p = &p4.x, *mm = &m4.x, *vv = &v4.x, *k = &k4.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float ge = g[e] * gscale;
            ge = ge * sc;
            g[e] = ge;                                                 // the averaged, clipped gradient stays in flat_grad
            float a = ge;
            if (wd != 0.f) a = a + wd * p[e];
            if (mask) a = a * k[e];
            mm[e] = mm[e] + w1 * (a - mm[e]);                          // lerp_(g, 1-b1), weight < 0.5 form
            vv[e] = vv[e] * b2 + (w2 * a) * a;                         // mul_(b2).addcmul_(g, g, value=1-b2)
            const float denom = sqrtf(vv[e]) / bc2s + eps;
            p[e] = p[e] - (mm[e] / denom) * lrbc1;
        }
        grad[i] = g4; param[i] = p4; m[i] = m4; v[i] = v4;
    }
}

}  // namespace

// n % 4 == 0, all buffers 16-byte aligned (the trainer's flat layout pads every parameter to 4 floats).
// partials: >= 256 doubles of scratch.  step / lr: device scalars (fp32).  mask: nullptr or [n] 0/1.  total_out: nullptr or [1]
// (the pre-clip global norm).  clip <= 0: no clipping.
extern "C" int i2p_clip_adam(long long n, float *param, float *grad, float *exp_avg, float *exp_avg_sq, const float *mask,
                             double *partials, float *step, const float *lr, double beta1, double beta2, float eps, float weight_decay,
                             float clip, float gscale, float *total_out, const float *poison, void *stream) {
    if (n <= 0 || (n & 3) || !param || !grad || !exp_avg || !exp_avg_sq || !partials || !step || !lr) return I2P_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
         reinterpret_cast<uintptr_t>(exp_avg_sq) | reinterpret_cast<uintptr_t>(mask)) & 15)
        return I2P_ERR_BAD_ARG;
    const long long n4 = n >> 2;
    long long nb = (n4 + ADAM_THREADS * 4 - 1) / (ADAM_THREADS * 4);
    if (nb > ADAM_MAX_BLOCKS) nb = ADAM_MAX_BLOCKS;
    if (nb < 1) nb = 1;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_norm_kernel, dim3((unsigned)nb), dim3(ADAM_THREADS), 0, st, n4, reinterpret_cast<const float4 *>(grad), gscale,
                       partials, step, poison);
    long long nu = (n4 + ADAM_THREADS - 1) / ADAM_THREADS;       // the update streams 8 tensors: one float4 per thread, the whole chip
    if (nu > 4096) nu = 4096;
    hipLaunchKernelGGL(adam_update_kernel, dim3((unsigned)nu), dim3(ADAM_THREADS), 0, st, n4, reinterpret_cast<float4 *>(param),
                       reinterpret_cast<float4 *>(grad), reinterpret_cast<float4 *>(exp_avg), reinterpret_cast<float4 *>(exp_avg_sq),
                       reinterpret_cast<const float4 *>(mask), partials, (int)nb, step, lr, (float)beta1, (float)beta2,
                       (float)(1.0 - beta1), (float)(1.0 - beta2),        // the lerp / addcmul weights as torch forms them: in double, then rounded
                       eps, weight_decay, clip, gscale, total_out, poison);
    I2P_RETURN_LAUNCH_STATUS();
}

// ---- kernel-only timing hook (see common.h) ---------------------------------------------------------------------------------
namespace {
bool g_ktime_on = false;
hipEvent_t g_kt0 = nullptr, g_kt1 = nullptr;
}
void i2p_ktime_begin(hipStream_t st) { if (g_ktime_on) (void)hipEventRecord(g_kt0, st); }
void i2p_ktime_end(hipStream_t st) { if (g_ktime_on) (void)hipEventRecord(g_kt1, st); }
extern "C" int i2p_ktime_enable(int on) {
    if (on && !g_kt0) {
        if (hipEventCreate(&g_kt0) != hipSuccess || hipEventCreate(&g_kt1) != hipSuccess) return I2P_ERR_BAD_ARG;
    }
    g_ktime_on = on != 0;
    return 0;
}
extern "C" float i2p_ktime_last_us(void) {
    if (!g_kt1) return -1.f;
    if (hipEventSynchronize(g_kt1) != hipSuccess) return -1.f;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_kt0, g_kt1) != hipSuccess) return -1.f;
    return ms * 1e3f;
}
