// Weight gradients of the plain linear layers left outside the fused layer kernels (cin > 160 or cout > 128 layers of
// levels 3-4 / the up-convolutions / the flow predictors, and the factored first cost-volume layer's per-point and
// per-pixel terms: modules.py Conv2d.forward, CostVolume._pi_all_pixels): out[m][n] = sum_r a[r][m] * b[r][n] with
// rows >> m, n.  A BLAS gives such a product one workgroup per output tile and a 15 000-iteration K loop (rocBLAS:
// 120 us for 256 x 128 over 14 848 rows, 55 us for 3 x 64 over 3 744); here the ROWS are cut over the grid, each
// block keeps a 64 x 64 output tile of its row chunk in registers (4 x 4 per thread, operands through LDS), and the
// chunks are summed in a fixed order by a second launch (bit-reproducible; no atomics).
#include "common.h"

namespace {

constexpr int GT_TILE = 64, GT_STAGE = 32, GT_THREADS = 256, GT_TARGET_BLOCKS = 512;

struct GemmTnGeom { int tiles_m, tiles_n, chunk_rows, nchunks; };

GemmTnGeom gemm_tn_geom(long long rows, int m, int n) {
    GemmTnGeom g;
    g.tiles_m = (m + GT_TILE - 1) / GT_TILE; g.tiles_n = (n + GT_TILE - 1) / GT_TILE;
    const long long tiles = (long long)g.tiles_m * g.tiles_n;
    long long want = GT_TARGET_BLOCKS / tiles; if (want < 1) want = 1;
    long long cr = (rows + want - 1) / want;
    cr = (cr + GT_STAGE - 1) / GT_STAGE * GT_STAGE; if (cr < GT_STAGE) cr = GT_STAGE;
    g.chunk_rows = (int)cr;
    g.nchunks = (int)((rows + cr - 1) / cr); if (g.nchunks < 1) g.nchunks = 1;
    return g;
}

__global__ __launch_bounds__(GT_THREADS) void gemm_tn_kernel(long long rows, int m, int n, const float *__restrict__ a, int lda,
                                                             const float *__restrict__ b, int ldb, int tiles_n, int chunk_rows,
                                                             float *__restrict__ partial) {
    __shared__ float sa[GT_STAGE][GT_TILE], sb[GT_STAGE][GT_TILE];
    const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
    const int m0 = (blockIdx.x / tiles_n) * GT_TILE, n0 = (blockIdx.x % tiles_n) * GT_TILE;
    const long long r_begin = (long long)blockIdx.y * chunk_rows;
    long long r_end = r_begin + chunk_rows; if (r_end > rows) r_end = rows;
    float acc[4][4] = {};
    const int lc = t & 63, lr = t >> 6;                       // staging: thread -> column lc of rows lr, lr+4, ...
    const bool am = m0 + lc < m, bn = n0 + lc < n;
    for (long long r0 = r_begin; r0 < r_end; r0 += GT_STAGE) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < GT_STAGE / 4; ++q) {
            const long long r = r0 + lr + 4 * q;
            const bool in = r < r_end;
            sa[lr + 4 * q][lc] = (in && am) ? a[r * lda + m0 + lc] : 0.f;
            sb[lr + 4 * q][lc] = (in && bn) ? b[r * ldb + n0 + lc] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < GT_STAGE; ++rr) {
            const float4 av = *reinterpret_cast<const float4 *>(&sa[rr][4 * ty]);
            const float4 bv = *reinterpret_cast<const float4 *>(&sb[rr][4 * tx]);
            const float ai[4] = {av.x, av.y, av.z, av.w}, bj[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(ai[i], bj[j], acc[i][j]);
        }
    }
    float *dst = partial + (size_t)blockIdx.y * m * n;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int mi = m0 + 4 * ty + i;
        if (mi >= m) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nj = n0 + 4 * tx + j;
            if (nj < n) dst[(size_t)mi * n + nj] = acc[i][j];
        }
    }
}

// out[o] = sum over chunks, in chunk order within each of 8 interleaved lanes, then lanes in order: the same sum every run
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(int nparts, int count, const float *__restrict__ parts, float *__restrict__ out) {
    __shared__ float red[8][32];
    const int o = blockIdx.x * 32 + (threadIdx.x & 31), pl = threadIdx.x >> 5;
    float s = 0.f;
    if (o < count)
        for (int c = pl; c < nparts; c += 8) s += parts[(size_t)c * count + o];
    red[pl][threadIdx.x & 31] = s;
    __syncthreads();
    if (pl == 0 && o < count) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) v += red[q][threadIdx.x & 31];
        out[o] = v;
    }
}

}  // namespace

extern "C" long long i2p_gemm_tn_scratch(long long rows, int m, int n) {
    if (rows <= 0 || m <= 0 || n <= 0) return 0;
    const GemmTnGeom g = gemm_tn_geom(rows, m, n);
    return (long long)g.nchunks * m * n * (long long)sizeof(float);
}

extern "C" int i2p_gemm_tn(long long rows, int m, int n, const float *a, int lda, const float *b, int ldb, void *scratch, float *out,
                           void *stream) {
    if (rows <= 0 || m <= 0 || n <= 0 || !a || !b || !scratch || !out || lda < m || ldb < n) return I2P_ERR_BAD_ARG;
    const GemmTnGeom g = gemm_tn_geom(rows, m, n);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(g.tiles_m * g.tiles_n, g.nchunks), dim3(GT_THREADS), 0, st, rows, m, n, a, lda, b, ldb,
                       g.tiles_n, g.chunk_rows, (float *)scratch);
    const int count = m * n;
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((count + 31) / 32), dim3(256), 0, st, g.nchunks, count, (const float *)scratch, out);
    I2P_RETURN_LAUNCH_STATUS();
}
