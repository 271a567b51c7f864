// Forward of the factored first cost-volume layer (pair mode, 128 -> 128) with bf16 output — third form of this layer
// (BASELINE configs[2] / [4]; reference: PPBackbone_center.py:383-433, first 1x1 conv of mlp1; factored form: DESIGN.md section 4):
//
//   y[(b,n,k), :] = bf16( bf16(f[b,n,:] * g[b,k,:]) . W^T + bias_n[b,n,:] + bias_k[b,k,:] ),   sums += {sum y, sum y^2} of the rounded y
//
// Only y touches HBM (rows * 256 B).  rg_fwd_kernel<4,false,PAIR> (rows in memory order) ran at 0.16 of 8 TB/s, pair_fwd_ps_kernel
// (pixel tile stationary in ONE wave: 398 registers, one wave per SIMD, 32 MFMAs + 64 LDS operand reads per 32-row strip, phases
// strictly in sequence) at 0.27.  Here the strip (b, kt, n) — 32 pixels of point n = 32 consecutive rows of y — is shared by the four
// waves of a block, like csrc/pair_bwd_bf16.hip does for the backward:
//   wave w forms two of the eight 16-channel steps of x' = bf16(f * g) (its pixel factors in 16 registers) into a shared LDS image,
//   wave w computes output-channel tile w: 8 MFMAs with its W tile as A operand IN REGISTERS (32) and x' from the image,
//   the tile goes back through a shared output image and leaves as whole 256-byte rows (one 16-byte store per thread and half strip)
//   with the BN statistics taken on the way.
// Images double-buffered, ONE LDS-only s_barrier per strip, persistent blocks over equal contiguous strip ranges, two blocks per CU.
// The accumulation order (bias_k, + bias_n, then the eight k-steps in order) is the one of the other two kernels: bit-identical y.
#include "bf16_common.h"
#include <cstdlib>

namespace {

constexpr int REP = I2P_BN_REPLICAS;
constexpr int F3_THREADS = 256, F3_C = 128, F3_PX = 32;
constexpr int F3_PW = 272;                               // padded pitch of a 256-byte image row (conflict-free operand accesses)
constexpr int F3_IMG = F3_PX * F3_PW;

struct PairFwd3P {
    int B, N, M, KT;
    long long S;
    const float *f, *g, *bias_n, *bias_k, *w;
    bf16_t *y;
    double *sums;
};

__device__ __forceinline__ int w_perm3(int m) { return 16 * ((m >> 2) & 1) + (m & 3) + 4 * (m >> 3); }
__device__ __forceinline__ void f3_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct It3 {
    int b, kt, n;
    __device__ __forceinline__ void next(int N, int KT) { if (++n == N) { n = 0; if (++kt == KT) { kt = 0; ++b; } } }
};

__global__ __launch_bounds__(F3_THREADS, 2) void pair_fwd3_bf16_kernel(PairFwd3P p) {
    extern __shared__ uint4 smem[];
    char *Xi = reinterpret_cast<char *>(smem);           // 2 x x' image [32 px] rows of 16 chunks
    char *Oi = Xi + 2 * F3_IMG;                          // 2 x output image
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = p.N, M = p.M, KT = p.KT;
    const int n = lane & 31, h = lane >> 5;

    // W tile of this wave as MFMA A operand: row i of the tile = output channel 32 w + w_perm3(i), 8 k-steps of 16 input channels
    i2p_bf16x8 wa[8];
    {
        const float *wr = p.w + (size_t)(32 * wave + w_perm3(n)) * F3_C;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const float4 a = *reinterpret_cast<const float4 *>(wr + (2 * ks + h) * 8), c = *reinterpret_cast<const float4 *>(wr + (2 * ks + h) * 8 + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
            wa[ks] = __builtin_bit_cast(i2p_bf16x8, bf_pack8(v));
        }
    }
    const long long s_begin = (long long)blockIdx.x * p.S / gridDim.x, s_end = (long long)(blockIdx.x + 1) * p.S / gridDim.x;
    const int count = __builtin_amdgcn_readfirstlane((int)(s_end - s_begin));
    double ssum[8], ssq[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { ssum[q] = 0.0; ssq[q] = 0.0; }

    if (count > 0) {
        It3 it_a, it_b;                                   // strip being staged (x') / being computed
        {
            const long long t = s_begin / N;
            it_a.n = __builtin_amdgcn_readfirstlane((int)(s_begin - t * N));
            it_a.b = __builtin_amdgcn_readfirstlane((int)(t / KT));
            it_a.kt = __builtin_amdgcn_readfirstlane((int)(t - (long long)(t / KT) * KT));
            it_b = it_a;
        }
        // staging role: this wave's two k-steps 2w, 2w+1 -> chunks ca = 4w + h, cb = 4w + 2 + h of pixel row n
        const int ca = 4 * wave + h, cb = ca + 2;
        float gA[8], gB[8];                               // g[b, k0 + n, 8 ca .. ], g[b, k0 + n, 8 cb .. ]  (tile of it_a)
        float bk[16];                                     // bias_k[b, k0 + n, 32 w + 16 h + e]               (tile of it_b)
        auto load_g = [&](const It3 &it) {
            const int k = it.kt * F3_PX + n;
            const float *gp = p.g + ((size_t)it.b * M + (k < M ? k : M - 1)) * F3_C;
            const float4 a0 = *reinterpret_cast<const float4 *>(gp + 8 * ca), a1 = *reinterpret_cast<const float4 *>(gp + 8 * ca + 4);
            const float4 b0 = *reinterpret_cast<const float4 *>(gp + 8 * cb), b1 = *reinterpret_cast<const float4 *>(gp + 8 * cb + 4);
            const bool ok = k < M;
            gA[0] = ok ? a0.x : 0.f; gA[1] = ok ? a0.y : 0.f; gA[2] = ok ? a0.z : 0.f; gA[3] = ok ? a0.w : 0.f;
            gA[4] = ok ? a1.x : 0.f; gA[5] = ok ? a1.y : 0.f; gA[6] = ok ? a1.z : 0.f; gA[7] = ok ? a1.w : 0.f;
            gB[0] = ok ? b0.x : 0.f; gB[1] = ok ? b0.y : 0.f; gB[2] = ok ? b0.z : 0.f; gB[3] = ok ? b0.w : 0.f;
            gB[4] = ok ? b1.x : 0.f; gB[5] = ok ? b1.y : 0.f; gB[6] = ok ? b1.z : 0.f; gB[7] = ok ? b1.w : 0.f;
        };
        auto load_bk = [&](const It3 &it) {
            const int k = it.kt * F3_PX + n;
            const float *bp = p.bias_k + ((size_t)it.b * M + (k < M ? k : M - 1)) * F3_C + 32 * wave + 16 * h;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 v = *reinterpret_cast<const float4 *>(bp + 4 * j);
                bk[4 * j] = k < M ? v.x : 0.f; bk[4 * j + 1] = k < M ? v.y : 0.f; bk[4 * j + 2] = k < M ? v.z : 0.f; bk[4 * j + 3] = k < M ? v.w : 0.f;
            }
        };
        // per-strip rows: f[b,n, 8ca.., 8cb..] for the staging role, bias_n[b,n, 32w + 16h ..] for the MFMA role
        float4 fr[4], bn[4];
        auto load_f = [&](const It3 &it) {
            const float *fp = p.f + ((size_t)it.b * N + it.n) * F3_C;
            fr[0] = *reinterpret_cast<const float4 *>(fp + 8 * ca); fr[1] = *reinterpret_cast<const float4 *>(fp + 8 * ca + 4);
            fr[2] = *reinterpret_cast<const float4 *>(fp + 8 * cb); fr[3] = *reinterpret_cast<const float4 *>(fp + 8 * cb + 4);
        };
        auto load_bn = [&](const It3 &it) {
            const float *bp = p.bias_n + ((size_t)it.b * N + it.n) * F3_C + 32 * wave + 16 * h;
#pragma unroll
            for (int j = 0; j < 4; ++j) bn[j] = *reinterpret_cast<const float4 *>(bp + 4 * j);
        };
        auto stage = [&](int buf) {                        // x' chunks of this wave for the strip of it_a (fr, gA / gB)
            char *X = Xi + buf * F3_IMG + n * F3_PW;
            const float pa[8] = {gA[0] * fr[0].x, gA[1] * fr[0].y, gA[2] * fr[0].z, gA[3] * fr[0].w, gA[4] * fr[1].x, gA[5] * fr[1].y, gA[6] * fr[1].z, gA[7] * fr[1].w};
            const float pb[8] = {gB[0] * fr[2].x, gB[1] * fr[2].y, gB[2] * fr[2].z, gB[3] * fr[2].w, gB[4] * fr[3].x, gB[5] * fr[3].y, gB[6] * fr[3].z, gB[7] * fr[3].w};
            *reinterpret_cast<uint4 *>(X + ca * 16) = bf_pack8(pa);
            *reinterpret_cast<uint4 *>(X + cb * 16) = bf_pack8(pb);
        };
        auto compute = [&](int buf) {                      // output tile of this wave for the strip of it_b (bn, bk) -> output image
            const char *X = Xi + buf * F3_IMG + n * F3_PW + h * 16;
            i2p_f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = bk[e];
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[4 * j] += bn[j].x; acc[4 * j + 1] += bn[j].y; acc[4 * j + 2] += bn[j].z; acc[4 * j + 3] += bn[j].w; }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const i2p_bf16x8 xb = __builtin_bit_cast(i2p_bf16x8, *reinterpret_cast<const uint4 *>(X + ks * 32));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ks], xb, acc, 0, 0, 0);
            }
            float lo[8], hi[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { lo[q] = acc[q]; hi[q] = acc[8 + q]; }
            char *O = Oi + buf * F3_IMG + n * F3_PW + (4 * wave + 2 * h) * 16;
            *reinterpret_cast<uint4 *>(O) = bf_pack8(lo);
            *reinterpret_cast<uint4 *>(O + 16) = bf_pack8(hi);
        };
        float s1[8], s2[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { s1[q] = 0.f; s2[q] = 0.f; }
        auto store = [&](const It3 &it, int buf) {         // whole rows out: thread (row r = tid >> 4 (+16), chunk tid & 15)
            const int k0 = it.kt * F3_PX, nv = min(F3_PX, M - k0);
            const size_t row0 = ((size_t)it.b * N + it.n) * M + k0;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int r = 16 * half + (tid >> 4);
                if (r < nv) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(Oi + buf * F3_IMG + r * F3_PW + (tid & 15) * 16);
                    float fq[8]; bf_unpack8(v, fq);
#pragma unroll
                    for (int q = 0; q < 8; ++q) { s1[q] += fq[q]; s2[q] = __builtin_fmaf(fq[q], fq[q], s2[q]); }
                    st_u4_stream(p.y + (row0 + r) * F3_C + (tid & 15) * 8, v);
                }
            }
        };
        auto flush_stats = [&]() {
#pragma unroll
            for (int q = 0; q < 8; ++q) { ssum[q] += (double)s1[q]; ssq[q] += (double)s2[q]; s1[q] = 0.f; s2[q] = 0.f; }
        };

        load_g(it_a); load_bk(it_b); load_f(it_a); load_bn(it_b);
        stage(0);
        if (count > 1) { It3 nx = it_a; nx.next(N, KT); if (nx.n == 0) load_g(nx); it_a = nx; load_f(it_a); }
        f3_barrier();
        It3 it_c = it_b;                                   // strip whose output image is stored next
        // order inside an iteration: the requests of the next strips go out BEFORE the stores of the previous strip — the memory
        // counter retires in order, so a wait for loads issued behind stores would wait for the stores' acknowledgements as well
        for (int t = 0; t < count; ++t) {
            const It3 it_prev = it_c;
            compute(t & 1);
            it_c = it_b;
            if (t + 1 < count) {
                stage((t + 1) & 1);                        // strip t + 1 (it_a, fr loaded an iteration ago)
                it_b.next(N, KT);
                if (it_b.n == 0) load_bk(it_b);
                load_bn(it_b);
                if (t + 2 < count) { It3 nx = it_a; nx.next(N, KT); if (nx.n == 0) load_g(nx); it_a = nx; load_f(it_a); }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t > 0) store(it_prev, (t - 1) & 1);
            if ((t & 7) == 7) flush_stats();
            f3_barrier();
        }
        store(it_c, (count - 1) & 1);
        flush_stats();
    }
    // ---- statistics: threads with equal (tid & 15) own the same 8 channels: lanes 16 / 32 apart, then the four waves through LDS ----
    if (p.sums) {
        f3_barrier();
        double *red = reinterpret_cast<double *>(smem);    // [4 waves][16 chunks][16]
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            double a = ssum[q], c = ssq[q];
            a += __shfl_xor(a, 16); c += __shfl_xor(c, 16);
            a += __shfl_xor(a, 32); c += __shfl_xor(c, 32);
            if (lane < 16) { red[(wave * 16 + lane) * 16 + q] = a; red[(wave * 16 + lane) * 16 + 8 + q] = c; }
        }
        f3_barrier();
        if (tid < 128) {                                   // channel tid: chunk tid >> 3, q = tid & 7
            const int ch = tid >> 3, q = tid & 7;
            double a = 0.0, c = 0.0;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) { a += red[(w2 * 16 + ch) * 16 + q]; c += red[(w2 * 16 + ch) * 16 + 8 + q]; }
            double *rep = p.sums + (size_t)(blockIdx.x % REP) * 2 * F3_C;
            atomicAdd(rep + tid, a); atomicAdd(rep + F3_C + tid, c);
        }
    }
}

}  // namespace

bool i2p_pair_fwd3_bf16_ok(int B, int N, int M, int cin, int cout) {
    const char *e = getenv("I2P_NO_PAIR_FWD3");                     // (read per call: the tests compare the kernels in one process)
    if (e && e[0] == '1') return false;
    return cin == F3_C && cout == F3_C && (long long)B * N * M >= 16384;
}

int i2p_pair_fwd3_bf16(int B, int N, int M, const float *f, const float *g, const float *bias_n, const float *bias_k, const float *w,
                       unsigned short *y, double *sums, void *stream) {
    if (!f || !g || !bias_n || !bias_k || !w || !y) return I2P_ERR_BAD_ARG;
    PairFwd3P p;
    p.B = B; p.N = N; p.M = M; p.KT = (M + F3_PX - 1) / F3_PX; p.S = (long long)B * p.KT * N;
    p.f = f; p.g = g; p.bias_n = bias_n; p.bias_k = bias_k; p.w = w; p.y = y; p.sums = sums;
    const size_t bytes = 4 * (size_t)F3_IMG;
    const unsigned grid = (unsigned)(p.S < 512 ? p.S : 512);
    hipLaunchKernelGGL(pair_fwd3_bf16_kernel, dim3(grid), dim3(F3_THREADS), bytes, (hipStream_t)stream, p);
    I2P_RETURN_LAUNCH_STATUS();
}
