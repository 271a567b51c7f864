// Weight gradients of the plain linear layers left outside the fused layer kernels (cin > 160 or cout > 128 layers of
// levels 3-4 / the up-convolutions / the flow predictors, and the factored first cost-volume layer's per-point and
// per-pixel terms: modules.py Conv2d.forward, CostVolume._pi_all_pixels): out[m][n] = sum_r a[r][m] * b[r][n] with
// rows >> m, n.  A BLAS gives such a product one workgroup per output tile and a 15 000-iteration K loop (rocBLAS:
// 120 us for 256 x 128 over 14 848 rows, 55 us for 3 x 64 over 3 744); here the ROWS are cut over the grid.  A block owns a
// 64 x 64 output tile of its row chunk; its four waves take interleaved groups of four rows (the K of
// v_mfma_f32_16x16x4_f32) and feed the MFMAs straight from global memory: the 16 MFMA tiles of the block tile are
// interleaved (tile t <-> columns 4i + t), so a lane's operands for all four tiles of a side are ONE float4 and the 16 lanes
// of a k-slot read a 256-byte run of the row.  The waves' accumulators are summed through LDS, the chunks by a second
// launch in a fixed order (bit-reproducible; no atomics).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float add(float a, float b) { return a + b; }
__device__ __forceinline__ float4 add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

constexpr int GT_TILE = 64, GT_THREADS = 256, GT_GROUP = 16 /* rows per block step: 4 waves x K=4 */, GT_UNROLL = 4;
constexpr int GT_ALIGN = GT_GROUP * GT_UNROLL, GT_TARGET_BLOCKS = 512;

struct GemmTnGeom { int tiles_m, tiles_n, chunk_rows, nchunks; };

GemmTnGeom gemm_tn_geom(long long rows, int m, int n) {
    GemmTnGeom g;
    g.tiles_m = (m + GT_TILE - 1) / GT_TILE; g.tiles_n = (n + GT_TILE - 1) / GT_TILE;
    const long long tiles = (long long)g.tiles_m * g.tiles_n;
    long long want = GT_TARGET_BLOCKS / tiles; if (want < 1) want = 1;
    long long cr = (rows + want - 1) / want;
    cr = (cr + GT_ALIGN - 1) / GT_ALIGN * GT_ALIGN; if (cr < GT_ALIGN) cr = GT_ALIGN;
    g.chunk_rows = (int)cr;
    g.nchunks = (int)((rows + cr - 1) / cr); if (g.nchunks < 1) g.nchunks = 1;
    return g;
}

// four consecutive columns c0 .. c0+3 of row r (zero outside [0, lim) and for rows >= r_end)
template <bool VEC>
__device__ __forceinline__ f32x4 ld4(const float *__restrict__ base, long long r, long long r_end, int ld, int c0, int lim) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r >= r_end) return v;
    const float *p = base + r * ld + c0;
    if constexpr (VEC) {
        if (c0 < lim) v = *reinterpret_cast<const f32x4 *>(p);           // lim % 4 == 0 here: the whole quad is inside
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c0 + e < lim) v[e] = p[e];
    }
    return v;
}

template <bool VA, bool VB>
__global__ __launch_bounds__(GT_THREADS) void gemm_tn_kernel(long long rows, int m, int n, const float *__restrict__ a, int lda,
                                                             const float *__restrict__ b, int ldb, int tiles_n, int chunk_rows,
                                                             float *__restrict__ partial) {
    __shared__ float red[4][4][4][4][64];                        // [wave][tm][tn][e][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, k = lane >> 4;
    const int m0 = (blockIdx.x / tiles_n) * GT_TILE, n0 = (blockIdx.x % tiles_n) * GT_TILE;
    const long long r_begin = (long long)blockIdx.y * chunk_rows;
    long long r_end = r_begin + chunk_rows; if (r_end > rows) r_end = rows;
    f32x4 acc[4][4];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ca = m0 + 4 * i, cb = n0 + 4 * i;
    f32x4 av[GT_UNROLL], bv[GT_UNROLL];
    long long r = r_begin + 4 * wave + k;                        // this lane's row of the wave's current group
#pragma unroll
    for (int u = 0; u < GT_UNROLL; ++u) {
        av[u] = ld4<VA>(a, r + u * GT_GROUP, r_end, lda, ca, m);
        bv[u] = ld4<VB>(b, r + u * GT_GROUP, r_end, ldb, cb, n);
    }
    for (long long base = r_begin; base < r_end; base += GT_ALIGN) {
        f32x4 an[GT_UNROLL], bn[GT_UNROLL];
        r += GT_ALIGN;
#pragma unroll
        for (int u = 0; u < GT_UNROLL; ++u) {                    // next step's operands in flight under this step's MFMAs
            an[u] = ld4<VA>(a, r + u * GT_GROUP, r_end, lda, ca, m);
            bn[u] = ld4<VB>(b, r + u * GT_GROUP, r_end, ldb, cb, n);
        }
#pragma unroll
        for (int u = 0; u < GT_UNROLL; ++u)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][tm], bv[u][tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < GT_UNROLL; ++u) { av[u] = an[u]; bv[u] = bn[u]; }
    }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave][tm][tn][e][lane] = acc[tm][tn][e];
    __syncthreads();
    // D of tile (tm, tn): lane l, register e = element (row 4*(l>>4) + e, column l & 15) -> output (m0 + 4*row + tm, n0 + 4*col + tn)
    float *dst = partial + (size_t)blockIdx.y * m * n;
    for (int item = threadIdx.x; item < 4 * 4 * 64; item += GT_THREADS) {
        const int l = item & 63, e = (item >> 6) & 3, tm = item >> 8;
        const int mi = m0 + 4 * (4 * (l >> 4) + e) + tm, nj = n0 + 4 * (l & 15);
        if (mi >= m || nj >= n) continue;
        float v[4];
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) v[tn] = (red[0][tm][tn][e][l] + red[1][tm][tn][e][l]) + (red[2][tm][tn][e][l] + red[3][tm][tn][e][l]);
        float *q = dst + (size_t)mi * n + nj;
        if (VB) *reinterpret_cast<float4 *>(q) = make_float4(v[0], v[1], v[2], v[3]);     // n % 4 == 0: the quad is inside, 16-byte aligned
        else
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
                if (nj + tn < n) q[tn] = v[tn];
    }
}

// out[o] = sum over chunks: 16 interleaved chunk lanes, each in chunk order, then the lanes in order — the same sum every run.
// 16-byte columns (count % 4 == 0) or single floats.
template <typename T>
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(int nparts, int count, const T *__restrict__ parts, T *__restrict__ out) {
    __shared__ T red[16][16];
    const int tx = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int o = blockIdx.x * 16 + tx;
    T s = T();
    if (o < count)
        for (int c = pl; c < nparts; c += 16) s = add(s, parts[(size_t)c * count + o]);
    red[pl][tx] = s;
    __syncthreads();
    if (pl == 0 && o < count) {
        T v = T();
#pragma unroll
        for (int q = 0; q < 16; ++q) v = add(v, red[q][tx]);
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
    const bool va = (m & 3) == 0 && (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0;
    const bool vb = (n & 3) == 0 && (ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(b) & 15) == 0 && (reinterpret_cast<uintptr_t>(scratch) & 15) == 0;
    const dim3 grid(g.tiles_m * g.tiles_n, g.nchunks), block(GT_THREADS);
#define GT_LAUNCH(A, B) hipLaunchKernelGGL((gemm_tn_kernel<A, B>), grid, block, 0, st, rows, m, n, a, lda, b, ldb, g.tiles_n, g.chunk_rows, (float *)scratch)
    if (va && vb) GT_LAUNCH(true, true); else if (va) GT_LAUNCH(true, false); else if (vb) GT_LAUNCH(false, true); else GT_LAUNCH(false, false);
#undef GT_LAUNCH
    const int count = m * n;
    if ((count & 3) == 0 && ((reinterpret_cast<uintptr_t>(scratch) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        if (!i2p_defer_reduce(0, g.nchunks, count / 4, scratch, out))
        hipLaunchKernelGGL(gemm_tn_reduce_kernel<float4>, dim3((count / 4 + 15) / 16), dim3(256), 0, st, g.nchunks, count / 4, (const float4 *)scratch,
                           (float4 *)out);
    } else
        hipLaunchKernelGGL(gemm_tn_reduce_kernel<float>, dim3((count + 15) / 16), dim3(256), 0, st, g.nchunks, count, (const float *)scratch, out);
    I2P_RETURN_LAUNCH_STATUS();
}
