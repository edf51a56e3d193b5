// Bilinear up-sampling with align_corners=True on plain NCHW fp32 planes (row n4): nn.Upsample(scale_factor, mode='bilinear',
// align_corners=True) of the reference's Upsample block (core/block.py:965-973) -- DBNet's x8 (core/model.py:223), the
// up_mode='bilinear' option of NestFuse / UNFusion / MAFusion.  Source coordinate of output row Y: sy = Y * (h-1)/(H-1), taken in
// fp32 exactly as ATen's area_pixel_compute_source_index does, y0 = floor(sy), y1 = min(y0 + 1, h-1), weight ly = sy - y0.
//   forward   out[Y][X] = (1-ly)(1-lx) x[y0][x0] + (1-ly) lx x[y0][x1] + ly (1-lx) x[y1][x0] + ly lx x[y1][x1]
//   backward  the adjoint in GATHER form (deterministic, no atomics): input pixel (iy, ix) sums g[Y][X] * wy(Y -> iy) * wx(X -> ix)
//             over the few output rows / columns whose y0 or y1 (x0 or x1) is iy (ix).
#include <math.h>

#include "common.hpp"

namespace mmif {

__device__ inline void bl_src(int Y, float r, int h, int& y0, int& y1, float& l) {
    const float s = r * (float)Y;
    y0 = min((int)s, h - 1);
    y1 = min(y0 + 1, h - 1);
    l = s - (float)y0;
}

__global__ void bilinear_up_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, long long planes, int h, int w, int H, int W,
                                       float rh, float rw) {
    const long long total = planes * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int X = (int)(i % W), Y = (int)((i / W) % H);
        const float* pl = x + (i / ((long long)H * W)) * h * w;
        int y0, y1, x0, x1;
        float ly, lx;
        bl_src(Y, rh, h, y0, y1, ly);
        bl_src(X, rw, w, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        out[i] = hy * (hx * pl[(long long)y0 * w + x0] + lx * pl[(long long)y0 * w + x1]) +
                 ly * (hx * pl[(long long)y1 * w + x0] + lx * pl[(long long)y1 * w + x1]);
    }
}

// weight with which output index Y contributes to input index iy
__device__ inline float bl_weight(int Y, float r, int h, int iy) {
    int y0, y1;
    float l;
    bl_src(Y, r, h, y0, y1, l);
    return (y0 == iy ? 1.f - l : 0.f) + (y1 == iy ? l : 0.f);
}

__global__ void bilinear_up_bwd_kernel(const float* __restrict__ g, float* __restrict__ dx, long long planes, int h, int w, int H, int W,
                                       float rh, float rw) {
    const long long total = planes * h * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ix = (int)(i % w), iy = (int)((i / w) % h);
        const float* pl = g + (i / ((long long)h * w)) * H * W;
        // candidate outputs: sy in (iy - 1, iy + 1)  <=>  Y in ((iy-1)/r, (iy+1)/r); one extra on each side absorbs fp32 rounding
        int Ylo = 0, Yhi = H - 1, Xlo = 0, Xhi = W - 1;
        if (rh > 0.f) { Ylo = max(0, (int)floorf((float)(iy - 1) / rh) - 1); Yhi = min(H - 1, (int)ceilf((float)(iy + 1) / rh) + 1); }
        if (rw > 0.f) { Xlo = max(0, (int)floorf((float)(ix - 1) / rw) - 1); Xhi = min(W - 1, (int)ceilf((float)(ix + 1) / rw) + 1); }
        float s = 0.f;
        for (int Y = Ylo; Y <= Yhi; ++Y) {
            const float wy = bl_weight(Y, rh, h, iy);
            if (wy == 0.f) continue;
            float row = 0.f;
            for (int X = Xlo; X <= Xhi; ++X) {
                const float wx = bl_weight(X, rw, w, ix);
                if (wx != 0.f) row += wx * pl[(long long)Y * W + X];
            }
            s += wy * row;
        }
        dx[i] = s;
    }
}

static int grid1d_r(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 65535 * 16 ? 65535 * 16 : b));
}
}  // namespace mmif

using namespace mmif;

extern "C" int mmif_bilinear_up_fwd(const float* x, float* out, int64_t planes, int32_t h, int32_t w, int32_t H, int32_t W, void* stream) {
    MMIF_REQUIRE(x != nullptr && out != nullptr && planes > 0 && h > 0 && w > 0 && H >= h && W >= w, "bilinear_up_fwd: bad arguments");
    const float rh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, rw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    hipLaunchKernelGGL(bilinear_up_fwd_kernel, dim3(grid1d_r(planes * H * W)), dim3(256), 0, (hipStream_t)stream, x, out, (long long)planes, h, w,
                       H, W, rh, rw);
    return check_launch("bilinear_up_fwd");
}

extern "C" int mmif_bilinear_up_bwd(const float* g, float* dx, int64_t planes, int32_t h, int32_t w, int32_t H, int32_t W, void* stream) {
    MMIF_REQUIRE(g != nullptr && dx != nullptr && planes > 0 && h > 0 && w > 0 && H >= h && W >= w, "bilinear_up_bwd: bad arguments");
    const float rh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, rw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    hipLaunchKernelGGL(bilinear_up_bwd_kernel, dim3(grid1d_r(planes * h * w)), dim3(256), 0, (hipStream_t)stream, g, dx, (long long)planes, h, w,
                       H, W, rh, rw);
    return check_launch("bilinear_up_bwd");
}

// ------------------------------------------------------------------ plain-NCHW resampling glue of the layer-by-layer blocks
// nn.MaxPool2d(k, k) (Downsample, core/block.py:941-950; k = 2, 4), nn.Upsample(scale_factor, mode='nearest') (:968-969) and the
// nn.ReflectionPad2d((l, r, t, b)) with which Upsample / Downsample match a target shape (:983-991; negative amounts crop).
namespace mmif {

// floor-mode pooling: out[oy][ox] = max over the k x k window; idx = offset (dy * k + dx) of the FIRST maximum in row-major window
// order (the element max_pool2d_with_indices_backward routes the gradient to)
__global__ void maxpool_nchw_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx, long long planes, int h,
                                        int w, int ho, int wo, int k) {
    const long long total = planes * ho * wo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % wo), oy = (int)((i / wo) % ho);
        const float* pl = x + (i / ((long long)ho * wo)) * h * w;
        float best = pl[(long long)(oy * k) * w + ox * k];
        int bi = 0;
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx) {
                const float v = pl[(long long)(oy * k + dy) * w + ox * k + dx];
                if (v > best || v != v) {   // NaN propagates like ATen's
                    if (!(best != best)) { best = v; bi = dy * k + dx; }
                }
            }
        y[i] = best;
        idx[i] = (unsigned char)bi;
    }
}

__global__ void maxpool_nchw_bwd_kernel(const float* __restrict__ g, const unsigned char* __restrict__ idx, float* __restrict__ dx, long long planes,
                                        int h, int w, int ho, int wo, int k) {
    const long long total = planes * h * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % w), y = (int)((i / w) % h);
        const long long plane = i / ((long long)h * w);
        const int oy = y / k, ox = x / k;
        float r = 0.f;
        if (oy < ho && ox < wo) {
            const long long o = (plane * ho + oy) * wo + ox;
            if ((int)idx[o] == (y - oy * k) * k + (x - ox * k)) r = g[o];
        }
        dx[i] = r;
    }
}

__global__ void nearest_up_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long planes, int h, int w, int s) {
    const int H = h * s, W = w * s;
    const long long total = planes * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int X = (int)(i % W), Y = (int)((i / W) % H);
        y[i] = x[((i / ((long long)H * W)) * h + Y / s) * w + X / s];
    }
}

__global__ void nearest_up_bwd_kernel(const float* __restrict__ g, float* __restrict__ dx, long long planes, int h, int w, int s) {
    const int H = h * s, W = w * s;
    const long long total = planes * h * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % w), y = (int)((i / w) % h);
        const float* pl = g + (i / ((long long)h * w)) * H * W;
        float r = 0.f;
        for (int dy = 0; dy < s; ++dy)
            for (int dxx = 0; dxx < s; ++dxx) r += pl[(long long)(y * s + dy) * W + x * s + dxx];
        dx[i] = r;
    }
}

// out [H][W] = reflect-pad (or crop, negative amounts) of x [h][w]: out(Y, X) = x(R(Y - top), R(X - left))
__global__ void reflect_pad_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long planes, int h, int w, int H, int W, int top,
                                       int left) {
    const long long total = planes * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int X = (int)(i % W), Y = (int)((i / W) % H);
        const int sy = reflect_idx(Y - top, h), sx = reflect_idx(X - left, w);
        y[i] = x[((i / ((long long)H * W)) * h + sy) * w + sx];
    }
}

// adjoint, gathered: dx(y, x) = sum of g over the output positions that read (y, x): the direct one and its mirrors about 0 / h-1
__global__ void reflect_pad_bwd_kernel(const float* __restrict__ g, float* __restrict__ dx, long long planes, int h, int w, int H, int W, int top,
                                       int left) {
    const long long total = planes * h * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % w), y = (int)((i / w) % h);
        const float* pl = g + (i / ((long long)h * w)) * H * W;
        int ys[3] = {y, -y, 2 * (h - 1) - y}, xs[3] = {x, -x, 2 * (w - 1) - x};
        const int ny = (y == 0 || y == h - 1) ? (h == 1 ? 1 : 2) : 3, nx = (x == 0 || x == w - 1) ? (w == 1 ? 1 : 2) : 3;
        if (y == 0) ys[1] = 2 * (h - 1) - y;        // (-0 duplicates the direct image: keep the other mirror)
        if (x == 0) xs[1] = 2 * (w - 1) - x;
        float r = 0.f;
        for (int a = 0; a < ny; ++a) {
            const int Y = ys[a] + top;
            if (Y < 0 || Y >= H) continue;
            for (int b = 0; b < nx; ++b) {
                const int X = xs[b] + left;
                if (X >= 0 && X < W) r += pl[(long long)Y * W + X];
            }
        }
        dx[i] = r;
    }
}
}  // namespace mmif

extern "C" int mmif_maxpool_nchw_fwd(const float* x, float* y, unsigned char* idx, int64_t planes, int32_t h, int32_t w, int32_t k, void* stream) {
    MMIF_REQUIRE(x != nullptr && y != nullptr && idx != nullptr && planes > 0 && k >= 1 && k <= 15 && h >= k && w >= k, "maxpool_nchw_fwd: bad arguments");
    const int ho = h / k, wo = w / k;
    hipLaunchKernelGGL(maxpool_nchw_fwd_kernel, dim3(grid1d_r(planes * ho * wo)), dim3(256), 0, (hipStream_t)stream, x, y, idx, (long long)planes, h,
                       w, ho, wo, k);
    return check_launch("maxpool_nchw_fwd");
}
extern "C" int mmif_maxpool_nchw_bwd(const float* g, const unsigned char* idx, float* dx, int64_t planes, int32_t h, int32_t w, int32_t k,
                                     void* stream) {
    MMIF_REQUIRE(g != nullptr && dx != nullptr && idx != nullptr && planes > 0 && k >= 1 && k <= 15 && h >= k && w >= k, "maxpool_nchw_bwd: bad arguments");
    hipLaunchKernelGGL(maxpool_nchw_bwd_kernel, dim3(grid1d_r(planes * h * w)), dim3(256), 0, (hipStream_t)stream, g, idx, dx, (long long)planes, h, w,
                       h / k, w / k, k);
    return check_launch("maxpool_nchw_bwd");
}
extern "C" int mmif_nearest_up_fwd(const float* x, float* y, int64_t planes, int32_t h, int32_t w, int32_t scale, void* stream) {
    MMIF_REQUIRE(x != nullptr && y != nullptr && planes > 0 && h > 0 && w > 0 && scale >= 1, "nearest_up_fwd: bad arguments");
    hipLaunchKernelGGL(nearest_up_fwd_kernel, dim3(grid1d_r(planes * h * w * scale * scale)), dim3(256), 0, (hipStream_t)stream, x, y,
                       (long long)planes, h, w, scale);
    return check_launch("nearest_up_fwd");
}
extern "C" int mmif_nearest_up_bwd(const float* g, float* dx, int64_t planes, int32_t h, int32_t w, int32_t scale, void* stream) {
    MMIF_REQUIRE(g != nullptr && dx != nullptr && planes > 0 && h > 0 && w > 0 && scale >= 1, "nearest_up_bwd: bad arguments");
    hipLaunchKernelGGL(nearest_up_bwd_kernel, dim3(grid1d_r(planes * h * w)), dim3(256), 0, (hipStream_t)stream, g, dx, (long long)planes, h, w, scale);
    return check_launch("nearest_up_bwd");
}
// (left, right, top, bottom) as nn.ReflectionPad2d; negative amounts crop; every amount must be < the corresponding input extent
extern "C" int mmif_reflect_pad_fwd(const float* x, float* y, int64_t planes, int32_t h, int32_t w, int32_t left, int32_t right, int32_t top,
                                    int32_t bottom, void* stream) {
    const int H = h + top + bottom, W = w + left + right;
    MMIF_REQUIRE(x != nullptr && y != nullptr && planes > 0 && h > 0 && w > 0 && H > 0 && W > 0, "reflect_pad_fwd: bad arguments");
    MMIF_REQUIRE(left < w && right < w && top < h && bottom < h, "reflect_pad_fwd: padding must be smaller than the input");
    hipLaunchKernelGGL(reflect_pad_fwd_kernel, dim3(grid1d_r(planes * H * W)), dim3(256), 0, (hipStream_t)stream, x, y, (long long)planes, h, w, H, W,
                       top, left);
    return check_launch("reflect_pad_fwd");
}
extern "C" int mmif_reflect_pad_bwd(const float* g, float* dx, int64_t planes, int32_t h, int32_t w, int32_t left, int32_t right, int32_t top,
                                    int32_t bottom, void* stream) {
    const int H = h + top + bottom, W = w + left + right;
    MMIF_REQUIRE(g != nullptr && dx != nullptr && planes > 0 && h > 0 && w > 0 && H > 0 && W > 0, "reflect_pad_bwd: bad arguments");
    MMIF_REQUIRE(left < w && right < w && top < h && bottom < h, "reflect_pad_bwd: padding must be smaller than the input");
    hipLaunchKernelGGL(reflect_pad_bwd_kernel, dim3(grid1d_r(planes * h * w)), dim3(256), 0, (hipStream_t)stream, g, dx, (long long)planes, h, w, H, W,
                       top, left);
    return check_launch("reflect_pad_bwd");
}
