// bf16 storage helpers shared by the bf16 kernels (mlp_bf16.hip, bf16_stream.hip).
//
// bf16 mode (BASELINE.json configs[2] / configs[4]) stores the pre-BN [rows, C] tensors of the fused layer chains and
// the gradients that flow between their layers as bf16 (round-to-nearest-even), halving the HBM round trips that
// batch-statistics BN forces (437 MB per sample per forward in fp32).  All arithmetic stays fp32
// (MFMA accumulators, BN/activation math, softmax) or fp64 (BN statistics); parameters stay fp32.
//
// One rule keeps forward and backward consistent: EVERY bf16 kernel evaluates a BN as  z = fmaf(y, a, b)  with
// a = invstd*gamma, b = beta - mean*a  on the bf16-rounded y it finds in memory, so the activation's sign (and with
// it the derivative mask) is bit-identical wherever it is recomputed.
#pragma once
#include "common.h"

typedef __bf16 i2p_bf16x2 __attribute__((ext_vector_type(2)));
typedef float i2p_f32x2 __attribute__((ext_vector_type(2)));
using i2p_bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using i2p_f32x16 = __attribute__((ext_vector_type(16))) float;
typedef unsigned short bf16_t;                       // storage type at the C ABI (raw bits)

__device__ __forceinline__ unsigned bf_pack2(float lo, float hi) {          // v_cvt_pk_bf16_f32 (RNE)
    const i2p_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, i2p_bf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float bf_round(float f) { return bf_lo(bf_pack2(f, 0.f)); }

__device__ __forceinline__ void bf_unpack8(const uint4 &v, float (&f)[8]) {
    f[0] = bf_lo(v.x); f[1] = bf_hi(v.x); f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
    f[4] = bf_lo(v.z); f[5] = bf_hi(v.z); f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}
__device__ __forceinline__ uint4 bf_pack8(const float (&f)[8]) {
    return make_uint4(bf_pack2(f[0], f[1]), bf_pack2(f[2], f[3]), bf_pack2(f[4], f[5]), bf_pack2(f[6], f[7]));
}
__device__ __forceinline__ float bf_act(float z, float slope) { return z > 0.f ? z : z * slope; }
__device__ __forceinline__ float bf_bnz(float y, float a, float b) { return __builtin_fmaf(y, a, b); }

// streaming 16-byte accesses (tensors touched once per launch)
typedef unsigned i2p_u32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_u4_stream(const void *ptr) {
    const i2p_u32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const i2p_u32x4_nt *>(ptr));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_u4_stream(void *ptr, const uint4 &v) {
    const i2p_u32x4_nt o = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(o, reinterpret_cast<i2p_u32x4_nt *>(ptr));
}

// LDS images of [rows][CP] 16-byte chunks (CP = 1 << cps, 2 <= CP <= 32), chunk index XOR-swizzled by the row so
// that the 16 lanes of every ds_read_b128 / ds_write_b128 group (rows with distinct row&15, equal logical chunk) fall on
// 16 distinct bank quads:  physical chunk = c ^ swz(row).
__device__ __forceinline__ int bf_swz(int row, int cps) {
    return cps >= 4 ? (row & 15) : ((row >> (4 - cps)) & ((1 << cps) - 1));
}
__device__ __forceinline__ int bf_chunk(int row, int c, int cps) { return (row << cps) + (c ^ bf_swz(row, cps)); }

// BN-backward constants of the BN BEHIND a layer, formed in the consumer kernel's prologue from the replicated fp64 sums (what the
// 64-thread launch bnbwd_coef_bf16 of mlp_bf16.hip computes; the same arithmetic in the same order, so the constants are
// bit-identical): dsums [REP][2c] = {sum gz, sum gz * xhat}, coef [3][c] = mean, scale, beta, mi [2][c] = mean, invstd.
// out8 (block 0 only, may be null): rows 6, 7 of the [8][c] scratch tail = dbeta, dgamma for the caller.
struct BnBwdSrc { const double *dsums; const float *coef, *mi; long long rows; float *out8; int c; };
__device__ __forceinline__ void bn_bwd_consts(const BnBwdSrc &s, int ch, float &gA, float &gB, float &gC) {
    // all 64 loads first (one round trip), then the sums in replica order
    double d0[I2P_BN_REPLICAS], d1[I2P_BN_REPLICAS];
#pragma unroll
    for (int r = 0; r < I2P_BN_REPLICAS; ++r) { d0[r] = s.dsums[(size_t)r * 2 * s.c + ch]; d1[r] = s.dsums[(size_t)r * 2 * s.c + s.c + ch]; }
    double sd = 0.0, sx = 0.0;
#pragma unroll
    for (int r = 0; r < I2P_BN_REPLICAS; ++r) { sd += d0[r]; sx += d1[r]; }
    const float m1 = (float)(sd / (double)s.rows), m2 = (float)(sx / (double)s.rows);
    const float sc = s.coef[s.c + ch], mu = s.mi[ch], is = s.mi[s.c + ch];
    gA = sc; gB = -(sc * m2) * is; gC = -(sc * m1) - gB * mu;
    if (s.out8 && blockIdx.x == 0) { s.out8[6 * s.c + ch] = (float)sd; s.out8[7 * s.c + ch] = (float)sx; }
}
