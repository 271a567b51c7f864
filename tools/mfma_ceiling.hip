// fp32 MFMA ceiling probes for gfx950 (tools/mfma_ceiling.py drives them): what rate does v_mfma_f32_16x16x4_f32 /
// v_mfma_f32_32x32x2_f32 reach (i) from many waves per SIMD with register operands (the pattern behind the guide's 155 TF),
// (ii) from ONE wave per SIMD holding 512 VGPRs (the weights-in-registers layer kernels of csrc/mlp_wreg.hip run like this),
// (iii) like (ii) with the HBM streams of a layer kernel on top (R float4 loads + S float4 stores per lane per 256 MFMAs).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// (i) NACC independent 16x16x4 accumulators per wave, `iters` rounds; occupancy set by the launch (waves per SIMD)
template <int NACC>
__global__ __launch_bounds__(256) void mfma_many_waves(float *out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[j].x + acc[j].y + acc[j].z + acc[j].w;
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

// (ii)/(iii) one wave per SIMD: 64 A-operand registers (a 128-wide weight panel's worth per k-group) and 32 accumulator tiles
// (128 VGPRs) like a 16-row x 128-column strip; 256 MFMAs per strip, fully unrolled; R loads / S stores of float4 per lane per strip
// (iv) like (iii) with the loads software-pipelined one strip ahead (issued right after the previous strip's operands were
// consumed, waited for a whole strip later) and the stores of a strip issued under the next strip's MFMAs: what a
// hand-scheduled layer kernel does.  The MFMA stream never waits for a load it has just issued.
template <int R, int S, bool WIDE>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mfma_one_wave_pipelined(float *out, const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, long long strips_per_wave, float a0) {
    float wreg[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) wreg[j] = a0 + (float)j * 1e-3f + threadIdx.x * 1e-6f;
    f32x4 acc[32];
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const f32x4 *sp = src + (size_t)wave * strips_per_wave * R * 64 + lane;
    f32x4 *dp = dst + (size_t)wave * strips_per_wave * S * 64 + lane;
    f32x4 x[R], xn[R], keep[S];
#pragma unroll
    for (int r = 0; r < R; ++r) x[r] = __builtin_nontemporal_load(sp + (size_t)r * 64);
#pragma unroll
    for (int q = 0; q < S; ++q) keep[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float carry = 0.f;
    for (long long s = 0; s < strips_per_wave; ++s) {
        const long long sn = s + 1 < strips_per_wave ? s + 1 : s;
#pragma unroll
        for (int r = 0; r < R; ++r) xn[r] = __builtin_nontemporal_load(sp + ((size_t)sn * R + r) * 64);      // next strip: a strip of latency budget
        if (s > 0) {
#pragma unroll
            for (int q = 0; q < S; ++q) __builtin_nontemporal_store(keep[q], dp + ((size_t)(s - 1) * S + q) * 64);  // previous strip's rows
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[j] = (f32x4){carry, 0.f, 0.f, 0.f};
        if (!WIDE) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float bk = x[k % R][k & 3];
#pragma unroll
                for (int j = 0; j < 32; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[(k * 8 + j) & 63], bk, acc[j], 0, 0, 0);
            }
        } else {
            f32x16 *wa = reinterpret_cast<f32x16 *>(acc);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float bk = x[k % R][k & 3];
#pragma unroll
                for (int j = 0; j < 8; ++j) wa[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[(k * 8 + j) & 63], bk, wa[j], 0, 0, 0);
            }
        }
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) t += acc[j].x;
        carry = t * 1e-30f;
#pragma unroll
        for (int q = 0; q < S; ++q) keep[q] = acc[q];
#pragma unroll
        for (int r = 0; r < R; ++r) x[r] = xn[r];
    }
#pragma unroll
    for (int q = 0; q < S; ++q) __builtin_nontemporal_store(keep[q], dp + ((size_t)(strips_per_wave - 1) * S + q) * 64);
    if (carry == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = carry;
}

template <int R, int S, bool WIDE>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mfma_one_wave(float *out, const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, long long strips_per_wave, float a0) {
    float wreg[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) wreg[j] = a0 + (float)j * 1e-3f + threadIdx.x * 1e-6f;
    f32x4 acc[32];
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const f32x4 *sp = src + (size_t)wave * strips_per_wave * (R > 0 ? R : 1) * 64 + lane;
    f32x4 *dp = dst + (size_t)wave * strips_per_wave * (S > 0 ? S : 1) * 64 + lane;
    f32x4 x[R > 0 ? R : 1];
#pragma unroll
    for (int r = 0; r < (R > 0 ? R : 1); ++r) x[r] = (f32x4){1.f, 2.f, 3.f, 4.f};
    float carry = 0.f;
    for (long long s = 0; s < strips_per_wave; ++s) {
        if (R > 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) x[r] = __builtin_nontemporal_load(sp + ((size_t)s * R + r) * 64);
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[j] = (f32x4){carry, 0.f, 0.f, 0.f};
        if (!WIDE) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {                       // 8 k-groups x 32 tiles = 256 MFMAs (16x16x4)
                const float bk = x[k % (R > 0 ? R : 1)][k & 3];
#pragma unroll
                for (int j = 0; j < 32; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[(k * 8 + j) & 63], bk, acc[j], 0, 0, 0);
            }
        } else {                                                 // the same flops as 128 MFMAs of 32x32x2 (8 accumulator tiles of 16 regs)
            f32x16 *wa = reinterpret_cast<f32x16 *>(acc);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float bk = x[k % (R > 0 ? R : 1)][k & 3];
#pragma unroll
                for (int j = 0; j < 8; ++j) wa[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[(k * 8 + j) & 63], bk, wa[j], 0, 0, 0);
            }
        }
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) t += acc[j].x;
        carry = t * 1e-30f;
        if (S > 0) {
#pragma unroll
            for (int q = 0; q < S; ++q) __builtin_nontemporal_store(acc[q], dp + ((size_t)s * S + q) * 64);
        }
    }
    if (carry == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = carry;
}

extern "C" {
// returns flop launched; all launches on `stream`
double probe_many_waves(int nacc, int blocks, int iters, float *out, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (nacc == 4) hipLaunchKernelGGL(mfma_many_waves<4>, dim3(blocks), dim3(256), 0, st, out, iters, 1.0f, 1.0f);
    else hipLaunchKernelGGL(mfma_many_waves<8>, dim3(blocks), dim3(256), 0, st, out, iters, 1.0f, 1.0f);
    return (double)blocks * 4 * iters * nacc * 2048.0;
}
double probe_one_wave(int mode, int blocks, long long strips, float *out, const void *src, void *dst, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const f32x4 *s = (const f32x4 *)src; f32x4 *d = (f32x4 *)dst;
    switch (mode) {
        case 0: hipLaunchKernelGGL((mfma_one_wave<0, 0, false>), dim3(blocks), dim3(256), 0, st, out, s, d, strips, 1.0f); break;
        case 1: hipLaunchKernelGGL((mfma_one_wave<8, 8, false>), dim3(blocks), dim3(256), 0, st, out, s, d, strips, 1.0f); break;     // forward: x in, y out
        case 2: hipLaunchKernelGGL((mfma_one_wave<24, 8, false>), dim3(blocks), dim3(256), 0, st, out, s, d, strips, 1.0f); break;    // dgrad: gz, y, x in, gz_in out
        case 3: hipLaunchKernelGGL((mfma_one_wave<0, 0, true>), dim3(blocks), dim3(256), 0, st, out, s, d, strips, 1.0f); break;
        case 4: hipLaunchKernelGGL((mfma_one_wave<8, 8, true>), dim3(blocks), dim3(256), 0, st, out, s, d, strips, 1.0f); break;
        case 5: hipLaunchKernelGGL((mfma_one_wave_pipelined<8, 8, false>), dim3(blocks), dim3(256), 0, st, out, s, d, strips, 1.0f); break;
        case 6: hipLaunchKernelGGL((mfma_one_wave_pipelined<8, 8, true>), dim3(blocks), dim3(256), 0, st, out, s, d, strips, 1.0f); break;
        case 7: hipLaunchKernelGGL((mfma_one_wave_pipelined<24, 8, false>), dim3(blocks), dim3(256), 0, st, out, s, d, strips, 1.0f); break;
        case 8: hipLaunchKernelGGL((mfma_one_wave_pipelined<24, 8, true>), dim3(blocks), dim3(256), 0, st, out, s, d, strips, 1.0f); break;
        default: return -1.0;
    }
    return (double)blocks * 4 * strips * 256 * 2048.0;
}
}
