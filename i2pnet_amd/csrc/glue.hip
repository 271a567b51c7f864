// Small one-launch replacements for clusters of elementwise ATen launches between the layer kernels.  Every kernel here
// replaces 2-4 launches of a few microseconds each that sit on the step's critical path inside the hipGraph (a kernel on a
// 200 KB tensor occupies the stream for ~4.5 us whatever it computes):
//   i2p_row_valid      any(x[r,:] != 0) as 0/1 floats            (utils.py:106-108 check_valid: ne + any + cast)
//   i2p_mask_fill      x*valid + fill*(1-valid) for 0/1 row masks  (modellearn_proj_center.py:318,376; PPBackbone_center.py:481:
//                      the reference's mul/sub/mul/add chain; here compare + fill + where)
//   i2p_pad_cols       [rows, c] -> [rows, cpad] with zero columns (weights of a layer whose input rows carry zero padding)
//   i2p_strided_pick2  the strided centre cells of two [B,H,W,3] range images in one launch (PPBackbone_center.py:94-95)
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void row_valid_kernel(long long rows, int c, const float *__restrict__ x, float *__restrict__ out) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    bool any = false;
    for (int i = 0; i < c; ++i) any |= x[r * c + i] != 0.f;
    out[r] = any ? 1.f : 0.f;
}

// one thread per float4 of the row-major [rows, c] tensor (c % 4 == 0) or per element
template <bool V4>
__global__ __launch_bounds__(256) void mask_fill_kernel(long long n, int per_row, const float *__restrict__ x, const float *__restrict__ valid,
                                                        float fill, float *__restrict__ out) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const bool keep = valid[t / per_row] > 0.f;
    if (V4) {
        const float4 v = reinterpret_cast<const float4 *>(x)[t];
        reinterpret_cast<float4 *>(out)[t] = keep ? v : make_float4(fill, fill, fill, fill);
    } else {
        out[t] = keep ? x[t] : fill;
    }
}

__global__ __launch_bounds__(256) void pad_cols_kernel(int rows, int c, int cpad, const float *__restrict__ w, float *__restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= rows * cpad) return;
    const int r = t / cpad, k = t - r * cpad;
    out[t] = k < c ? w[r * c + k] : 0.f;
}

__global__ __launch_bounds__(256) void strided_pick2_kernel(int B, int H, int W, int oh, int ow, int sh, int sw, const float *__restrict__ a,
                                                            const float *__restrict__ b, float *__restrict__ oa, float *__restrict__ ob) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)B * oh * ow * 3;
    if (t >= n) return;
    const int e = (int)(t % 3);
    long long q = t / 3;
    const int w = (int)(q % ow); q /= ow;
    const int h = (int)(q % oh); const int bi = (int)(q / oh);
    const size_t src = (((size_t)bi * H + (size_t)h * sh) * W + (size_t)w * sw) * 3 + e;
    oa[t] = a[src];
    if (b) ob[t] = b[src];
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward-validation feature of the first cost volume (PPBackbone_center.py:408-414): respond[b,k,c] = max over the valid
// points n of pts[b,n,c] * pix[b,k,c], in closed form — for a fixed pixel value g the maximum of fl(f_n * g) is
// fl(g * max_n f_n) if g >= 0 else fl(g * min_n f_n) (rounding is monotone) — -1e10 when the sample has no valid point.
// The torch formulation is 18 launches forward and 14 backward on [8,468,128] tensors; here one launch each way:
// a block owns (sample, 32 channels): it reduces max / min (+ arg) over the points, then streams the pixels.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int MR_CH = 32, MR_ROWS = 32, MR_THREADS = MR_CH * MR_ROWS;      // 1024 threads = 32 row lanes x 32 channels (few blocks: the waves of a block hide each other's latencies)

__global__ __launch_bounds__(MR_THREADS) void maxresp_fwd_kernel(int N, int M, int C, const float *__restrict__ pts, const float *__restrict__ pix,
                                                          const float *__restrict__ valid, float *__restrict__ respond,
                                                          float *__restrict__ fmaxmin, int *__restrict__ imaxmin, int *__restrict__ anyv) {
    __shared__ float smax[MR_ROWS][MR_CH], smin[MR_ROWS][MR_CH];
    __shared__ int simax[MR_ROWS][MR_CH], simin[MR_ROWS][MR_CH], sany[MR_ROWS][MR_CH];
    const int b = blockIdx.x, c0 = blockIdx.y * MR_CH, ch = threadIdx.x & (MR_CH - 1), r = threadIdx.x >> 5;
    const int c = c0 + ch;
    float vmax = -INFINITY, vmin = INFINITY; int imax = 0, imin = 0, any = 0;
    if (c < C)
        for (int n = r; n < N; n += MR_ROWS) {
            if (valid[(size_t)b * N + n] > 0.f) {
                const float v = pts[((size_t)b * N + n) * C + c];
                any = 1;
                if (v > vmax) { vmax = v; imax = n; }            // first maximum in point order within this lane's rows
                if (v < vmin) { vmin = v; imin = n; }
            }
        }
    smax[r][ch] = vmax; smin[r][ch] = vmin; simax[r][ch] = imax; simin[r][ch] = imin; sany[r][ch] = any;
    __syncthreads();
    // combine the row lanes (lowest point index wins ties, like a serial scan)
    vmax = smax[0][ch]; vmin = smin[0][ch]; imax = simax[0][ch]; imin = simin[0][ch]; any = sany[0][ch];
#pragma unroll
    for (int q = 1; q < MR_ROWS; ++q) {
        const float a = smax[q][ch], d = smin[q][ch];
        if (a > vmax || (a == vmax && simax[q][ch] < imax && sany[q][ch])) { vmax = a; imax = simax[q][ch]; }
        if (d < vmin || (d == vmin && simin[q][ch] < imin && sany[q][ch])) { vmin = d; imin = simin[q][ch]; }
        any |= sany[q][ch];
    }
    if (!any) { vmax = 0.f; vmin = 0.f; imax = imin = 0; }
    if (c < C && r == 0) {
        fmaxmin[((size_t)b * 2 + 0) * C + c] = vmax; fmaxmin[((size_t)b * 2 + 1) * C + c] = vmin;
        imaxmin[((size_t)b * 2 + 0) * C + c] = imax; imaxmin[((size_t)b * 2 + 1) * C + c] = imin;
        if (blockIdx.y == 0 && ch == 0) anyv[b] = any;
    }
    if (c >= C) return;
#pragma unroll 4
    for (int k = r; k < M; k += MR_ROWS) {
        const float g = pix[((size_t)b * M + k) * C + c];
        respond[((size_t)b * M + k) * C + c] = any ? g * (g >= 0.f ? vmax : vmin) : -1e10f;
    }
}

__global__ __launch_bounds__(MR_THREADS) void maxresp_bwd_kernel(int N, int M, int C, const float *__restrict__ g, const float *__restrict__ pix,
                                                          const float *__restrict__ fmaxmin, const int *__restrict__ imaxmin,
                                                          const int *__restrict__ anyv, float *__restrict__ d_pts, float *__restrict__ d_pix) {
    __shared__ float s1[MR_ROWS][MR_CH], s2[MR_ROWS][MR_CH];
    const int b = blockIdx.x, c0 = blockIdx.y * MR_CH, ch = threadIdx.x & (MR_CH - 1), r = threadIdx.x >> 5;
    const int c = c0 + ch;
    const bool any = anyv[b] != 0;
    float dmax = 0.f, dmin = 0.f;
    if (c < C) {
        const float vmax = fmaxmin[((size_t)b * 2 + 0) * C + c], vmin = fmaxmin[((size_t)b * 2 + 1) * C + c];
#pragma unroll 4
        for (int k = r; k < M; k += MR_ROWS) {
            const size_t o = ((size_t)b * M + k) * C + c;
            const float gv = any ? g[o] : 0.f, pv = pix[o];
            const bool pos = pv >= 0.f;
            d_pix[o] = gv * (pos ? vmax : vmin);
            const float t = gv * pv;
            if (pos) dmax += t; else dmin += t;
        }
        for (int n = r; n < N; n += MR_ROWS) d_pts[((size_t)b * N + n) * C + c] = 0.f;     // this block owns d_pts[b, :, c0..c0+32)
    }
    s1[r][ch] = dmax; s2[r][ch] = dmin;
    __syncthreads();
    if (r == 0 && c < C) {
        float a = 0.f, d = 0.f;
#pragma unroll
        for (int q = 0; q < MR_ROWS; ++q) { a += s1[q][ch]; d += s2[q][ch]; }
        const int imax = imaxmin[((size_t)b * 2 + 0) * C + c], imin = imaxmin[((size_t)b * 2 + 1) * C + c];
        // (the zero fill above was done by other threads of this block: ordered by the barrier; both rows may coincide)
        d_pts[((size_t)b * N + imax) * C + c] += a;
        d_pts[((size_t)b * N + imin) * C + c] += d;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Pose-head MLP (PPBackbone_center.py:553-562): hidden = W1 pooled + b1 -> dropout -> q = Wq h + bq, t = Wt h + bt ->
// q / (sqrt(|q|^2 + 1e-10) + 1e-10) on [B, C] -> [B, 256] -> [B, 7] — three addmm, a dropout and a normalisation forward,
// ~16 launches backward (two GEMMs + a bias reduction per layer), all on a few KB.  One block does each direction.
//   mask [B,H]: the dropout multiplier (0 or 1/(1-p)) or NULL.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PH_THREADS = 1024;      // one block, 16 waves: a one-block kernel has nothing else to hide its latencies

// W1 [H][C] -> LDS rows of pitch C+1; C % 4 == 0: float4 loads, eight of them in flight per thread before the first LDS store
__device__ __forceinline__ void stage_w1(float *sw, const float *__restrict__ w1, int H, int C, int tid) {
    if ((C & 3) == 0 && (reinterpret_cast<uintptr_t>(w1) & 15) == 0) {
        const int n4 = (H * C) >> 2;
        for (int i0 = tid; i0 < n4; i0 += PH_THREADS * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * PH_THREADS; v[u] = i < n4 ? reinterpret_cast<const float4 *>(w1)[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * PH_THREADS;
                if (i < n4) {
                    const int e = i * 4, r = e / C, c = e - r * C;
                    float *d = sw + r * (C + 1) + c;
                    d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
                }
            }
        }
    } else {
        for (int i = tid; i < H * C; i += PH_THREADS) sw[(i / C) * (C + 1) + (i % C)] = w1[i];
    }
}

__global__ __launch_bounds__(PH_THREADS) void pose_head_fwd_kernel(int B, int C, int H, const float *__restrict__ pooled, const float *__restrict__ w1,
                                                                   const float *__restrict__ b1, const float *__restrict__ mask,
                                                                   const float *__restrict__ wq, const float *__restrict__ bq,
                                                                   const float *__restrict__ wt, const float *__restrict__ bt,
                                                                   float *__restrict__ hid, float *__restrict__ qraw, float *__restrict__ q,
                                                                   float *__restrict__ t) {
    extern __shared__ float ph[];                           // pooled [B][C], hidden [B][H], out [B][8], W1 [H][C+1]
    float *sp = ph, *sh = ph + B * C, *so = sh + B * H, *sw = so + B * 8;
    const int tid = threadIdx.x;
    float *swo = sw + H * (C + 1);                          // [7][H]: the two output layers' weights
    // A one-block kernel pays ~2 us for every dependent global round trip and the compiler will not move one staging loop's loads
    // above the previous loop's LDS stores: request everything first (registers), store afterwards — one round trip for the prologue.
    if (B * C <= 2 * PH_THREADS && 7 * H <= 7 * PH_THREADS && (C & 3) == 0 && H * C <= 16 * 4 * PH_THREADS && (reinterpret_cast<uintptr_t>(w1) & 15) == 0) {
        float rp[2], ro[7];
        float4 rw[16];
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int i = tid + u * PH_THREADS; rp[u] = i < B * C ? pooled[i] : 0.f; }
#pragma unroll
        for (int u = 0; u < 7; ++u) { const int i = tid + u * PH_THREADS; ro[u] = i < 7 * H ? (i < 4 * H ? wq[i] : wt[i - 4 * H]) : 0.f; }
        const int n4 = (H * C) >> 2;
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int i = tid + u * PH_THREADS; rw[u] = i < n4 ? reinterpret_cast<const float4 *>(w1)[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int i = tid + u * PH_THREADS; if (i < B * C) sp[i] = rp[u]; }
#pragma unroll
        for (int u = 0; u < 7; ++u) { const int i = tid + u * PH_THREADS; if (i < 7 * H) swo[i] = ro[u]; }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = tid + u * PH_THREADS;
            if (i < n4) {
                const int e = i * 4, r = e / C, c = e - r * C;
                float *d = sw + r * (C + 1) + c;
                d[0] = rw[u].x; d[1] = rw[u].y; d[2] = rw[u].z; d[3] = rw[u].w;
            }
        }
    } else {
        for (int i = tid; i < B * C; i += PH_THREADS) sp[i] = pooled[i];
        for (int i = tid; i < 7 * H; i += PH_THREADS) swo[i] = i < 4 * H ? wq[i] : wt[i - 4 * H];
        stage_w1(sw, w1, H, C, tid);                        // coalesced 16-byte loads, 8 in flight; row pitch C+1: conflict-free row reads
    }
    __syncthreads();
    // phase 1: thread = (hidden unit h, batch slice): PH_THREADS / H slices, a slice takes batches part, part + nb, ...
    const int nb = H <= PH_THREADS ? PH_THREADS / H : 1;
    for (int item = tid; item < H * nb; item += PH_THREADS) {
        const int h = item % H, part = item / H;
        for (int bb = part; bb < B; bb += 8 * nb) {
            const float bias = b1[h];
            float mk[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) mk[j] = (mask && bb + j * nb < B) ? mask[(bb + j * nb) * H + h] : 1.f;
            float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int c = 0; c < C; ++c) {
                const float w = sw[h * (C + 1) + c];
#pragma unroll
                for (int j = 0; j < 8; ++j) if (bb + j * nb < B) a[j] = __builtin_fmaf(w, sp[(bb + j * nb) * C + c], a[j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (bb + j * nb < B) {
                    float v = a[j] + bias;
                    if (mask) v *= mk[j];
                    sh[(bb + j * nb) * H + h] = v; hid[(bb + j * nb) * H + h] = v;
                }
        }
    }
    __syncthreads();
    // the 7*B output dot products of length H: one 16-lane group per (b, o), partial sums combined with DPP row adds
    for (int i = tid >> 4; i < B * 7; i += PH_THREADS / 16) {
        const int b = i / 7, o = i - b * 7, l = tid & 15;
        const float *w = swo + o * H;
        float a = 0.f;
#pragma unroll 8
        for (int h = l; h < H; h += 16) a = __builtin_fmaf(w[h], sh[b * H + h], a);
        a = __uint_as_float(i2p_row16_add_f32(a));
        if (l == 0) {
            a += o < 4 ? bq[o] : bt[o - 4];
            so[b * 8 + o] = a;
            if (o < 4) qraw[b * 4 + o] = a; else t[b * 3 + o - 4] = a;
        }
    }
    __syncthreads();
    for (int b = tid; b < B; b += PH_THREADS) {
        const float x = so[b * 8], y = so[b * 8 + 1], z = so[b * 8 + 2], w = so[b * 8 + 3];
        const float d = sqrtf((x * x + y * y + z * z + w * w) + 1e-10f) + 1e-10f;
        q[b * 4] = x / d; q[b * 4 + 1] = y / d; q[b * 4 + 2] = z / d; q[b * 4 + 3] = w / d;
    }
}

__global__ __launch_bounds__(PH_THREADS) void pose_head_bwd_kernel(int B, int C, int H, const float *__restrict__ gq, const float *__restrict__ gt,
                                                                   const float *__restrict__ qraw, const float *__restrict__ hid,
                                                                   const float *__restrict__ mask, const float *__restrict__ pooled,
                                                                   const float *__restrict__ w1, const float *__restrict__ wq,
                                                                   const float *__restrict__ wt, float *__restrict__ d_pooled,
                                                                   float *__restrict__ dw1, float *__restrict__ db1, float *__restrict__ dwq,
                                                                   float *__restrict__ dbq, float *__restrict__ dwt, float *__restrict__ dbt) {
    extern __shared__ float ph[];                           // pooled [B][C], d_hidden [B][H], d_out [B][8], W1 [H][C+1]
    float *sp = ph, *sd = ph + B * C, *so = sd + B * H, *sw = so + B * 8;
    const int tid = threadIdx.x;
    for (int i = tid; i < B * C; i += PH_THREADS) sp[i] = pooled[i];
    if (d_pooled) stage_w1(sw, w1, H, C, tid);
    for (int b = tid; b < B; b += PH_THREADS) {             // normalisation backward (quat_unit_bwd_kernel, mode 1), then [d_qraw | d_t]
        const float x = qraw[b * 4], y = qraw[b * 4 + 1], z = qraw[b * 4 + 2], w = qraw[b * 4 + 3];
        const float g0 = gq ? gq[b * 4] : 0.f, g1 = gq ? gq[b * 4 + 1] : 0.f, g2 = gq ? gq[b * 4 + 2] : 0.f, g3 = gq ? gq[b * 4 + 3] : 0.f;
        const float n2 = (x * x + y * y + z * z + w * w) + 1e-10f, rr = sqrtf(n2), d = rr + 1e-10f;
        const float k = (g0 * x + g1 * y + g2 * z + g3 * w) / (d * d * rr);
        so[b * 8] = g0 / d - x * k; so[b * 8 + 1] = g1 / d - y * k; so[b * 8 + 2] = g2 / d - z * k; so[b * 8 + 3] = g3 / d - w * k;
        for (int o = 0; o < 3; ++o) so[b * 8 + 4 + o] = gt ? gt[b * 3 + o] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < 7; i += PH_THREADS) {             // bias gradients of the two output layers
        float a = 0.f;
        for (int b = 0; b < B; ++b) a += so[b * 8 + i];
        if (i < 4) dbq[i] = a; else dbt[i - 4] = a;
    }
    for (int h = tid; h < H; h += PH_THREADS) {
        float wcol[7];
#pragma unroll
        for (int o = 0; o < 7; ++o) wcol[o] = o < 4 ? wq[(size_t)o * H + h] : wt[(size_t)(o - 4) * H + h];
        float dwo[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dbh = 0.f;
        for (int b0 = 0; b0 < B; b0 += 8) {
            float hv[8], mk[8];                              // (all global reads of the chunk in flight before the arithmetic)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                hv[j] = b0 + j < B ? hid[(b0 + j) * H + h] : 0.f;
                mk[j] = (mask && b0 + j < B) ? mask[(b0 + j) * H + h] : 1.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int b = b0 + j;
                if (b < B) {
                    float dh = 0.f;
#pragma unroll
                    for (int o = 0; o < 7; ++o) { dh = __builtin_fmaf(so[b * 8 + o], wcol[o], dh); dwo[o] = __builtin_fmaf(so[b * 8 + o], hv[j], dwo[o]); }
                    if (mask) dh *= mk[j];
                    sd[b * H + h] = dh; dbh += dh;
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 7; ++o) { if (o < 4) dwq[(size_t)o * H + h] = dwo[o]; else dwt[(size_t)(o - 4) * H + h] = dwo[o]; }
        db1[h] = dbh;
    }
    __syncthreads();
    for (int i = tid; i < H * C; i += PH_THREADS) {         // dW1[h,c] = sum_b d_hidden[b,h] * pooled[b,c]
        const int h = i / C, c = i - h * C;
        float a = 0.f;
        for (int b = 0; b < B; ++b) a = __builtin_fmaf(sd[b * H + h], sp[b * C + c], a);
        dw1[i] = a;
    }
    if (d_pooled)
        for (int i = tid; i < B * C; i += PH_THREADS) {     // d_pooled[b,c] = sum_h d_hidden[b,h] * W1[h,c]
            const int b = i / C, c = i - b * C;
            // (a serial 256-step chain of two LDS reads + one FMA is latency-bound at ~130 cycles per step in a one-block kernel:
            // four independent partial sums, eight steps in flight)
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int h = 0;
#pragma unroll 2
            for (; h + 3 < H; h += 4) {
                a0 = __builtin_fmaf(sd[b * H + h], sw[h * (C + 1) + c], a0);
                a1 = __builtin_fmaf(sd[b * H + h + 1], sw[(h + 1) * (C + 1) + c], a1);
                a2 = __builtin_fmaf(sd[b * H + h + 2], sw[(h + 2) * (C + 1) + c], a2);
                a3 = __builtin_fmaf(sd[b * H + h + 3], sw[(h + 3) * (C + 1) + c], a3);
            }
            for (; h < H; ++h) a0 = __builtin_fmaf(sd[b * H + h], sw[h * (C + 1) + c], a0);
            d_pooled[i] = (a0 + a1) + (a2 + a3);
        }
}

}  // namespace

static size_t pose_head_lds(int B, int C, int H) { return ((size_t)B * C + (size_t)B * H + (size_t)B * 8 + (size_t)H * (C + 1) + (size_t)7 * H) * sizeof(float); }
static void pose_head_attr() {
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pose_head_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pose_head_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        done = true;
    }
}

// pooled [B,C], w1 [H,C], b1 [H], mask [B,H] or NULL, wq [4,H], bq [4], wt [3,H], bt [3] -> hid [B,H] (post-dropout hidden, saved for the
// backward), qraw [B,4] (unnormalised), q [B,4] = qraw / (sqrt(|qraw|^2 + 1e-10) + 1e-10), t [B,3]
extern "C" int i2p_pose_head_fwd(int B, int C, int H, const float *pooled, const float *w1, const float *b1, const float *mask, const float *wq,
                                 const float *bq, const float *wt, const float *bt, float *hid, float *qraw, float *q, float *t, void *stream) {
    if (B <= 0 || C <= 0 || H <= 0 || pose_head_lds(B, C, H) > 150 * 1024) return I2P_ERR_BAD_ARG;
    if (!pooled || !w1 || !b1 || !wq || !bq || !wt || !bt || !hid || !qraw || !q || !t) return I2P_ERR_BAD_ARG;
    pose_head_attr();
    hipLaunchKernelGGL(pose_head_fwd_kernel, dim3(1), dim3(PH_THREADS), pose_head_lds(B, C, H), (hipStream_t)stream, B, C, H, pooled, w1, b1, mask,
                       wq, bq, wt, bt, hid, qraw, q, t);
    I2P_RETURN_LAUNCH_STATUS();
}

// gq [B,4] = dL/dq, gt [B,3] = dL/dt (either may be NULL = zero) -> every parameter gradient and d_pooled [B,C] (NULL: skipped)
extern "C" int i2p_pose_head_bwd(int B, int C, int H, const float *gq, const float *gt, const float *qraw, const float *hid, const float *mask,
                                 const float *pooled, const float *w1, const float *wq, const float *wt, float *d_pooled, float *dw1, float *db1,
                                 float *dwq, float *dbq, float *dwt, float *dbt, void *stream) {
    if (B <= 0 || C <= 0 || H <= 0 || pose_head_lds(B, C, H) > 150 * 1024) return I2P_ERR_BAD_ARG;
    if (!qraw || !hid || !pooled || !w1 || !wq || !wt || !dw1 || !db1 || !dwq || !dbq || !dwt || !dbt) return I2P_ERR_BAD_ARG;
    pose_head_attr();
    hipLaunchKernelGGL(pose_head_bwd_kernel, dim3(1), dim3(PH_THREADS), pose_head_lds(B, C, H), (hipStream_t)stream, B, C, H, gq, gt, qraw, hid,
                       mask, pooled, w1, wq, wt, d_pooled, dw1, db1, dwq, dbq, dwt, dbt);
    I2P_RETURN_LAUNCH_STATUS();
}

namespace {
}  // namespace

// pts [B,N,C], pix [B,M,C], valid [B,N] (0/1) -> respond [B,M,C]; saves fmaxmin [B,2,C], imaxmin i32 [B,2,C], anyv i32 [B]
extern "C" int i2p_max_response_fwd(int B, int N, int M, int C, const float *pts, const float *pix, const float *valid, float *respond,
                                    float *fmaxmin, int *imaxmin, int *anyv, void *stream) {
    if (B <= 0 || N <= 0 || M <= 0 || C <= 0 || !pts || !pix || !valid || !respond || !fmaxmin || !imaxmin || !anyv) return I2P_ERR_BAD_ARG;
    hipLaunchKernelGGL(maxresp_fwd_kernel, dim3(B, (C + MR_CH - 1) / MR_CH), dim3(MR_THREADS), 0, (hipStream_t)stream, N, M, C, pts, pix, valid, respond,
                       fmaxmin, imaxmin, anyv);
    I2P_RETURN_LAUNCH_STATUS();
}

// g = dL/drespond [B,M,C] -> d_pts [B,N,C] (written completely), d_pix [B,M,C]
extern "C" int i2p_max_response_bwd(int B, int N, int M, int C, const float *g, const float *pix, const float *fmaxmin, const int *imaxmin,
                                    const int *anyv, float *d_pts, float *d_pix, void *stream) {
    if (B <= 0 || N <= 0 || M <= 0 || C <= 0 || !g || !pix || !fmaxmin || !imaxmin || !anyv || !d_pts || !d_pix) return I2P_ERR_BAD_ARG;
    hipLaunchKernelGGL(maxresp_bwd_kernel, dim3(B, (C + MR_CH - 1) / MR_CH), dim3(MR_THREADS), 0, (hipStream_t)stream, N, M, C, g, pix, fmaxmin, imaxmin,
                       anyv, d_pts, d_pix);
    I2P_RETURN_LAUNCH_STATUS();
}

// K_inv[b] = inverse of the intrinsic matrix rescaled to a feature map (fx, cx by sx; fy, cy by sy): change_intrinsic + the adjugate
// inverse of model.py (modellearn_proj_center.py:457-463, :282 — torch.inverse on the CPU there) in one launch instead of eight
// (scale multiply, three cross products, multiply, sum, stack, divide).  Same arithmetic, operation for operation (no contraction).
__global__ void intrinsic_inverse_kernel(int B, const float *__restrict__ K, float sx, float sy, float *__restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float *k = K + 9 * b;
    const float r0[3] = {k[0] * sx, k[1] * 1.0f, k[2] * sx}, r1[3] = {k[3] * 1.0f, k[4] * sy, k[5] * sy}, r2[3] = {k[6], k[7], k[8]};
    auto cross = [](const float *a, const float *c, float *o) {
        o[0] = a[1] * c[2] - a[2] * c[1]; o[1] = a[2] * c[0] - a[0] * c[2]; o[2] = a[0] * c[1] - a[1] * c[0];
    };
    float c0[3], c1[3], c2[3];
    cross(r1, r2, c0); cross(r2, r0, c1); cross(r0, r1, c2);
    const float det = (r0[0] * c0[0] + r0[1] * c0[1]) + r0[2] * c0[2];
    float *o = out + 9 * b;
#pragma unroll
    for (int i = 0; i < 3; ++i) { o[3 * i] = c0[i] / det; o[3 * i + 1] = c1[i] / det; o[3 * i + 2] = c2[i] / det; }
}

extern "C" int i2p_intrinsic_inverse(int B, const float *K, float sx, float sy, float *out, void *stream) {
    if (B < 0) return I2P_ERR_BAD_ARG;
    if (B == 0) return 0;
    if (!K || !out) return I2P_ERR_BAD_ARG;
    hipLaunchKernelGGL(intrinsic_inverse_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, B, K, sx, sy, out);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_row_valid(long long rows, int c, const float *x, float *out, void *stream) {
    if (rows < 0 || c <= 0) return I2P_ERR_BAD_ARG;
    if (rows == 0) return 0;
    if (!x || !out) return I2P_ERR_BAD_ARG;
    hipLaunchKernelGGL(row_valid_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, c, x, out);
    I2P_RETURN_LAUNCH_STATUS();
}

// out[r, :] = valid[r] > 0 ? x[r, :] : fill   (x, out [rows, c]; valid [rows]); the backward is the same call on the gradient with fill = 0
extern "C" int i2p_mask_fill(long long rows, int c, const float *x, const float *valid, float fill, float *out, void *stream) {
    if (rows < 0 || c <= 0) return I2P_ERR_BAD_ARG;
    if (rows == 0) return 0;
    if (!x || !valid || !out) return I2P_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if ((c & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        const long long n = rows * (c >> 2);
        hipLaunchKernelGGL(mask_fill_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, c >> 2, x, valid, fill, out);
    } else {
        const long long n = rows * c;
        hipLaunchKernelGGL(mask_fill_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, c, x, valid, fill, out);
    }
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_pad_cols(int rows, int c, int cpad, const float *w, float *out, void *stream) {
    if (rows <= 0 || c <= 0 || cpad < c || (long long)rows * cpad > (1LL << 30) || !w || !out) return I2P_ERR_BAD_ARG;
    hipLaunchKernelGGL(pad_cols_kernel, dim3((unsigned)(((long long)rows * cpad + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, c, cpad, w, out);
    I2P_RETURN_LAUNCH_STATUS();
}

// oa[b,h,w,:] = a[b, h*sh, w*sw, :] (and ob from b when b != NULL) for h < oh, w < ow; (oh-1)*sh < H and (ow-1)*sw < W
extern "C" int i2p_strided_pick2(int B, int H, int W, int oh, int ow, int sh, int sw, const float *a, const float *b, float *oa, float *ob,
                                 void *stream) {
    if (B <= 0 || H <= 0 || W <= 0 || oh <= 0 || ow <= 0 || sh <= 0 || sw <= 0 || (oh - 1) * sh >= H || (ow - 1) * sw >= W) return I2P_ERR_BAD_ARG;
    if (!a || !oa || (b && !ob)) return I2P_ERR_BAD_ARG;
    const long long n = (long long)B * oh * ow * 3;
    hipLaunchKernelGGL(strided_pick2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B, H, W, oh, ow, sh, sw, a, b, oa, ob);
    I2P_RETURN_LAUNCH_STATUS();
}
