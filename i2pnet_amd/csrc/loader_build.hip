// Device-side build of a batch of KITTI odometry samples (i2pnet_amd/data.py::DeviceSampleBuilder) in two launches per BATCH:
// the per-point work of src/kitti_odometry_corr_lidarnone_proj.py:524-533 (shuffle), :332-343 (jitter), :654-656 (extrinsic
// product in float64), :699-711 (zero padding to 150 000 rows) and the per-pixel work of :713-747 (drop the top rows, cv2.resize
// x0.5 INTER_LINEAR, 160x512 crop, float) — what the torch path of data.py does in ~50 small kernels per SAMPLE (1.1 ms of host
// launch time each: the loader-inclusive step was host-bound).  The arithmetic is the torch path's, operation
// for operation (this file is compiled with -ffp-contract=off), so the two agree bit for bit (tests/test_data_pipeline.py); the
// torch path itself is pinned to sample dicts of the imported reference loader (tests/golden/loader_kitti.npz).
#include <hip/hip_runtime.h>
#include <cstdint>

#include "common.h"

namespace {

// one row of the per-batch parameter table (16 x 8 bytes, written by the host into pinned memory, one H2D copy per batch)
struct PointParams {
    const float *scan;           // [N, 4] x, y, z, intensity
    const long long *perm;       // [N] the sample's shuffle
    long long n;                 // rows kept = min(N, sample_point)
    long long reserved;
    double E[12];                // init_extrinsic, row-major 3 x 4
};
static_assert(sizeof(PointParams) == 128, "table layout");

// lidar / raw [B, SP, 3], feats [B, SP, 1]; noise [B, SP, 3] standard normal draws or nullptr
__global__ __launch_bounds__(256) void kitti_points_kernel(const PointParams *__restrict__ tab, int SP, const float *__restrict__ noise,
                                                           float *__restrict__ lidar, float *__restrict__ raw, float *__restrict__ feats) {
    const int b = blockIdx.y;
    const PointParams p = tab[b];
    const size_t base = (size_t)b * SP;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < SP; i += (long long)gridDim.x * blockDim.x) {
        float x = 0.f, y = 0.f, z = 0.f, w = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
        if (i < p.n) {
            const float4 s = *reinterpret_cast<const float4 *>(p.scan + 4 * (size_t)p.perm[i]);
            x = s.x; y = s.y; z = s.z; w = s.w;
            if (noise) {                                         // pc + clamp(0.01 * N(0,1), -0.05, 0.05), float32
                const float *r = noise + 3 * (base + i);
                x = x + fminf(fmaxf(0.01f * r[0], -0.05f), 0.05f);
                y = y + fminf(fmaxf(0.01f * r[1], -0.05f), 0.05f);
                z = z + fminf(fmaxf(0.01f * r[2], -0.05f), 0.05f);
            }
            const double dx = (double)x, dy = (double)y, dz = (double)z;
            cx = (float)(((dx * p.E[0] + dy * p.E[1]) + dz * p.E[2]) + p.E[3]);      // data.affine_f64: left to right, no contraction
            cy = (float)(((dx * p.E[4] + dy * p.E[5]) + dz * p.E[6]) + p.E[7]);
            cz = (float)(((dx * p.E[8] + dy * p.E[9]) + dz * p.E[10]) + p.E[11]);
        }
        float *l = lidar + 3 * (base + i), *q = raw + 3 * (base + i);
        l[0] = cx; l[1] = cy; l[2] = cz;
        q[0] = x; q[1] = y; q[2] = z;
        feats[base + i] = w;
    }
}

struct ImageParams {
    const unsigned char *img;    // [H, W, 3] uint8, the top rows already dropped (pointer advanced)
    long long H, W;              // source size
    long long oh, ow;            // size after cv2.resize
    long long dx, dy;            // crop offset in the resized image
    long long exact2;            // H == 2 oh and W == 2 ow: INTER_AREA's 2x2 mean (cv2's fast path for an exact 2x shrink)
};
static_assert(sizeof(ImageParams) == 64, "table layout");

// taps of one output coordinate along one axis, as data._linear_taps computes them (cv2 resize.cpp, 8-bit linear branch)
__device__ __forceinline__ void linear_tap(int d, int n_out, int n_in, bool clamp_weight, int &i0, int &i1, int &w0, int &w1) {
    const double scale = (double)n_in / (double)n_out;
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    const float s = floorf(f);
    f = f - s;
    int si = (int)s;
    if (clamp_weight) {
        const bool lo = si < 0, hi = si >= n_in - 1;
        if (lo || hi) f = 0.f;
        si = lo ? 0 : (hi ? n_in - 1 : si);
    }
    w0 = (int)rintf((1.0f - f) * 2048.0f);                      // cvRound: round half to even
    w1 = (int)rintf(f * 2048.0f);
    i0 = si < 0 ? 0 : (si > n_in - 1 ? n_in - 1 : si);
    i1 = si + 1 < 0 ? 0 : (si + 1 > n_in - 1 ? n_in - 1 : si + 1);
}

// rgb [B, 3, OH, OW] float in 0..255
__global__ __launch_bounds__(256) void kitti_image_kernel(const ImageParams *__restrict__ tab, int OH, int OW, float *__restrict__ rgb) {
    const int b = blockIdx.z, r = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= OW) return;
    const ImageParams p = tab[b];
    const int W = (int)p.W, H = (int)p.H;
    const int oy = (int)p.dy + r, ox = (int)p.dx + c;          // coordinates in the resized image
    int out[3];
    if (p.exact2) {
        const unsigned char *r0 = p.img + ((size_t)(2 * oy) * W + 2 * ox) * 3, *r1 = r0 + (size_t)W * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) out[k] = ((int)r0[k] + (int)r0[3 + k] + (int)r1[k] + (int)r1[3 + k] + 2) >> 2;
    } else {
        int x0, x1, a0, a1, y0, y1, b0, b1;
        linear_tap(ox, (int)p.ow, W, true, x0, x1, a0, a1);
        linear_tap(oy, (int)p.oh, H, false, y0, y1, b0, b1);
        const unsigned char *s0 = p.img + (size_t)y0 * W * 3, *s1 = p.img + (size_t)y1 * W * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const long long h0 = (long long)s0[3 * x0 + k] * a0 + (long long)s0[3 * x1 + k] * a1;     // horizontal pass (int)
            const long long h1 = (long long)s1[3 * x0 + k] * a0 + (long long)s1[3 * x1 + k] * a1;
            const long long v = ((((long long)b0 * (h0 >> 4)) >> 16) + (((long long)b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            out[k] = (int)(v & 255);                             // `.to(torch.uint8)`
        }
    }
    const size_t plane = (size_t)OH * OW;
    float *dst = rgb + (size_t)b * 3 * plane + (size_t)r * OW + c;
    dst[0] = (float)out[0]; dst[plane] = (float)out[1]; dst[2 * plane] = (float)out[2];
}

}  // namespace

extern "C" int i2p_kitti_points_build(int B, int sample_point, const void *table, const float *noise, float *lidar, float *raw,
                                      float *feats, void *stream) {
    if (B < 0 || sample_point <= 0 || !table || !lidar || !raw || !feats) return I2P_ERR_BAD_ARG;
    if (B == 0) return 0;
    const unsigned gx = (unsigned)((sample_point + 255) / 256);
    hipLaunchKernelGGL(kitti_points_kernel, dim3(gx < 1024 ? gx : 1024, B), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const PointParams *>(table), sample_point, noise, lidar, raw, feats);
    I2P_RETURN_LAUNCH_STATUS();
}

extern "C" int i2p_kitti_image_build(int B, int out_h, int out_w, const void *table, float *rgb, void *stream) {
    if (B < 0 || out_h <= 0 || out_w <= 0 || !table || !rgb) return I2P_ERR_BAD_ARG;
    if (B == 0) return 0;
    hipLaunchKernelGGL(kitti_image_kernel, dim3((out_w + 255) / 256, out_h, B), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const ImageParams *>(table), out_h, out_w, rgb);
    I2P_RETURN_LAUNCH_STATUS();
}
