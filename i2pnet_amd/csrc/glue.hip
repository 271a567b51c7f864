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

}  // namespace

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
