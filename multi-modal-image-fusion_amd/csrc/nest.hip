// NestFuse / RFN-Nest glue kernels on blocked-NHWC tensors (all HBM-bound, granule-wise):
//   2x2 max-pool fwd/bwd            (reference nn.MaxPool2d(2,2), core/model.py:332-335)
//   nearest x2 upsample + reflect pad to the skip's shape, fwd/bwd   (core/block.py:965-991)
//   attention_fusion 'sa' | 'ca' | 'sca' with spatial 'l1' / channel 'avg' pooling, fwd/bwd (core/fusion.py:32-124)
//   in-place ReLU mask of a gradient
// Gradient tensors may be halo-1 (padded-domain) buffers: reads go through load_grad_fold (honours the
// FOLDED flag), writes hit the interior at (+halo, +halo).
#include "common.hpp"

namespace mmif {

static int grid_for(long long total) {
    long long b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

#define GRID_STRIDE(i, total) \
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (long long)gridDim.x * blockDim.x)

template <typename T>
__device__ inline void ld(const TV& t, int n, int c, int ys, int xs, float (&v)[8]) {
    Elem<T>::load(t.base + t.gidx(n, c, ys, xs) * Elem<T>::gran_bytes, v);
}
template <typename T>
__device__ inline void st(const TV& t, int n, int c, int ys, int xs, const float (&v)[8]) {
    Elem<T>::store(t.base + t.gidx(n, c, ys, xs) * Elem<T>::gran_bytes, v);
}

// ------------------------------------------------------------------ max-pool 2x2 / 2
template <typename T>
__global__ void maxpool_fwd_kernel(TV x, TV y) {
    const long long total = (long long)y.n * y.cb * y.h * y.w;
    GRID_STRIDE(i, total) {
        int n, c, yo, xo;
        split_idx(i, y.cb, y.h, y.w, n, c, yo, xo);
        float a[8], b[8], d[8], e[8], o[8];
        ld<T>(x, n, c, 2 * yo, 2 * xo, a);
        ld<T>(x, n, c, 2 * yo, 2 * xo + 1, b);
        ld<T>(x, n, c, 2 * yo + 1, 2 * xo, d);
        ld<T>(x, n, c, 2 * yo + 1, 2 * xo + 1, e);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = fmaxf(fmaxf(a[k], b[k]), fmaxf(d[k], e[k]));
        st<T>(y, n, c, yo, xo, o);
    }
}

// gx[input pixel] (+)= g[pool cell] if this pixel is the FIRST maximum of its cell (torch's tie rule), else 0
template <typename T>
__global__ void maxpool_bwd_kernel(TV x, TV g, TV gx, int accumulate, int relu_mask) {
    const long long total = (long long)x.n * x.cb * x.h * x.w;
    GRID_STRIDE(i, total) {
        int n, c, yi, xi;
        split_idx(i, x.cb, x.h, x.w, n, c, yi, xi);
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int yo = yi >> 1, xo = xi >> 1;
        if (yo < g.h && xo < g.w) {
            float q[4][8], gv[8];
            ld<T>(x, n, c, 2 * yo, 2 * xo, q[0]);
            ld<T>(x, n, c, 2 * yo, 2 * xo + 1, q[1]);
            ld<T>(x, n, c, 2 * yo + 1, 2 * xo, q[2]);
            ld<T>(x, n, c, 2 * yo + 1, 2 * xo + 1, q[3]);
            load_grad_fold<T>(g, n, c, yo, xo, gv);
            const int me = (yi & 1) * 2 + (xi & 1);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int arg = 0;
                float m = q[0][k];
#pragma unroll
                for (int j = 1; j < 4; ++j)
                    if (q[j][k] > m) { m = q[j][k]; arg = j; }
                o[k] = (arg == me) ? gv[k] : 0.f;
            }
        }
        if (accumulate) {
            float old[8];
            ld<T>(gx, n, c, yi + gx.halo, xi + gx.halo, old);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] += old[k];
        }
        if (relu_mask) {   // this was the last contribution to the gradient of the ReLU output x: threshold_backward here, not in a pass of its own
            float xv[8];
            ld<T>(x, n, c, yi, xi, xv);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = xv[k] > 0.f ? o[k] : 0.f;
        }
        st<T>(gx, n, c, yi + gx.halo, xi + gx.halo, o);
    }
}

// ------------------------------------------------------------------ nearest x2 upsample (+ reflect pad to y's shape)
// y[yy][xx] = x[ R(yy - top, 2h) / 2 ][ R(xx - left, 2w) / 2 ]
__device__ inline int up_src(int o, int pad_lo, int len2) { return min(max(reflect_idx(o - pad_lo, len2), 0), len2 - 1) >> 1; }

template <typename T>
__global__ void upsample_fwd_kernel(TV x, TV y) {
    const int top = (y.h - 2 * x.h) / 2, left = (y.w - 2 * x.w) / 2;
    const long long total = (long long)y.n * y.cb * y.h * y.w;
    GRID_STRIDE(i, total) {
        int n, c, yo, xo;
        split_idx(i, y.cb, y.h, y.w, n, c, yo, xo);
        float v[8];
        ld<T>(x, n, c, up_src(yo, top, 2 * x.h), up_src(xo, left, 2 * x.w), v);
        st<T>(y, n, c, yo, xo, v);
    }
}

// gx[i][j] (+)= sum of g over every y position that reads x[i][j]
template <typename T>
__global__ void upsample_bwd_kernel(TV g, TV gx, int accumulate, TV xm, int relu_mask) {
    const int H2 = 2 * gx.h, W2 = 2 * gx.w;
    const int top = (g.h - H2) / 2, left = (g.w - W2) / 2;
    const long long total = (long long)gx.n * gx.cb * gx.h * gx.w;
    GRID_STRIDE(i, total) {
        int n, c, yi, xi;
        split_idx(i, gx.cb, gx.h, gx.w, n, c, yi, xi);
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // candidates: the 2 direct rows / cols plus the (few) reflected pad rows / cols that mirror onto them
        const int bot = g.h - top - H2, right = g.w - left - W2;
        const int ny = 2 + top + bot, nx = 2 + left + right;
        for (int a = 0; a < ny; ++a) {
            const int yy = a < 2 ? top + 2 * yi + a : (a < 2 + top ? a - 2 : top + H2 + (a - 2 - top));
            if (up_src(yy, top, H2) != yi) continue;
            for (int b = 0; b < nx; ++b) {
                const int xx = b < 2 ? left + 2 * xi + b : (b < 2 + left ? b - 2 : left + W2 + (b - 2 - left));
                if (up_src(xx, left, W2) != xi) continue;
                float gv[8];
                load_grad_fold<T>(g, n, c, yy, xx, gv);
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] += gv[k];
            }
        }
        if (accumulate) {
            float old[8];
            ld<T>(gx, n, c, yi + gx.halo, xi + gx.halo, old);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] += old[k];
        }
        if (relu_mask) {   // last contribution to the gradient of the ReLU output xm
            float xv[8];
            ld<T>(xm, n, c, yi, xi, xv);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = xv[k] > 0.f ? o[k] : 0.f;
        }
        st<T>(gx, n, c, yi + gx.halo, xi + gx.halo, o);
    }
}

// ------------------------------------------------------------------ in-place ReLU mask: g *= [x > 0] on g's stored domain
template <typename T>
__global__ void relu_mask_kernel(TV x, TV g) {
    const long long total = (long long)g.n * g.cb * g.hs * g.ws;
    GRID_STRIDE(i, total) {
        int n, c, ys, xs;
        split_idx(i, g.cb, g.hs, g.ws, n, c, ys, xs);
        float gv[8], xv[8];
        ld<T>(g, n, c, ys, xs, gv);
        load_act_reflect<T>(x, n, c, ys - g.halo, xs - g.halo, xv);
#pragma unroll
        for (int k = 0; k < 8; ++k) gv[k] = xv[k] > 0.f ? gv[k] : 0.f;
        st<T>(g, n, c, ys, xs, gv);
    }
}

// ------------------------------------------------------------------ attention fusion (core/fusion.py:32-124)
// per-(n, channel) plane sums: out[(n*C + ch)*NV + v]; mode 0: v0 = sum a, v1 = sum b (channel 'avg' pooling * HW);
// mode 1: v0 = sum g*(a-b)  (gradient w.r.t. the channel weight).  One block per (n, channel block).
constexpr int PS_SLICES = 32;

template <typename T, int MODE>
__global__ __launch_bounds__(256) void plane_sums_kernel(TV a, TV b, TV g, float* __restrict__ partial) {
    __shared__ float red[16];
    const int c = blockIdx.x % a.cb, n = blockIdx.x / a.cb, sl = blockIdx.y;
    float s0[8], s1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s0[k] = s1[k] = 0.f;
    for (int p = sl * 256 + threadIdx.x; p < a.h * a.w; p += 256 * PS_SLICES) {
        const int y = p / a.w, x = p % a.w;
        float va[8], vb[8];
        ld<T>(a, n, c, y, x, va);
        ld<T>(b, n, c, y, x, vb);
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { s0[k] += va[k]; s1[k] += vb[k]; }
        } else {
            float gv[8];
            load_grad_fold<T>(g, n, c, y, x, gv);
#pragma unroll
            for (int k = 0; k < 8; ++k) s0[k] += gv[k] * (va[k] - vb[k]);
        }
    }
    constexpr int NV = MODE == 0 ? 2 : 1;
    const int C = a.cb * 8;
    float* dst = partial + ((long long)sl * a.n * C + (long long)n * C + c * 8) * NV;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float t0 = block_sum(s0[k], red);
        if (threadIdx.x == 0) dst[k * NV + 0] = t0;
        if (MODE == 0) {
            const float t1 = block_sum(s1[k], red);
            if (threadIdx.x == 0) dst[k * NV + 1] = t1;
        }
    }
}

// out[i] = sum over the PS_SLICES partial planes (fixed order)
__global__ void plane_sums_finish(const float* __restrict__ partial, float* __restrict__ out, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float s = 0.f;
    for (int sl = 0; sl < PS_SLICES; ++sl) s += partial[(long long)sl * count + i];
    out[i] = s;
}

// per-(n, channel) coefficients of the channel branch, once per call instead of once per pixel (they were 8 divisions per granule in the
// pixel loops): wc = m1 / max(m1 + m2, eps); backward: da = (1/dc - m1/dc^2 [ms >= eps]) / HW, db = (-m1/dc^2 [ms >= eps]) / HW.
// Same expressions, same fp32 values as the per-pixel form.
__global__ void attn_coef_kernel(const float* __restrict__ csum, float* __restrict__ coef, int count, float inv_hw) {
    const float EPSV = 1e-7f;
    const int ci = blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= count) return;
    const float m1 = csum[ci * 2] * inv_hw, m2 = csum[ci * 2 + 1] * inv_hw;
    const float ms = m1 + m2, dc = fmaxf(ms, EPSV), wc = m1 / dc, pc = ms >= EPSV ? 1.f : 0.f;
    coef[ci] = wc;
    coef[count + ci] = (1.f / dc - m1 / (dc * dc) * pc) * inv_hw;
    coef[2 * count + ci] = (-m1 / (dc * dc) * pc) * inv_hw;
}

// mode bits: 1 = spatial branch, 2 = channel branch; both -> 'sca' = mean of the two
template <typename T>
__global__ void attn_fwd_kernel(TV a, TV b, TV o, const float* __restrict__ coef, int mode) {
    const float EPSV = 1e-7f;
    const long long total = (long long)o.n * o.h * o.w;
    const int C = a.cb * 8;
    const float scale = mode == 3 ? 0.5f : 1.f;
    GRID_STRIDE(i, total) {
        int n, c_unused, y, x;
        split_idx(i, 1, o.h, o.w, n, c_unused, y, x);
        float ws = 0.f;
        if (mode & 1) {
            float s1 = 0.f, s2 = 0.f;
            for (int c = 0; c < a.cb; ++c) {
                float va[8], vb[8];
                ld<T>(a, n, c, y, x, va);
                ld<T>(b, n, c, y, x, vb);
#pragma unroll
                for (int k = 0; k < 8; ++k) { s1 += fabsf(va[k]); s2 += fabsf(vb[k]); }
            }
            ws = s1 / fmaxf(s1 + s2, EPSV);
        }
        for (int c = 0; c < a.cb; ++c) {
            float va[8], vb[8], vo[8];
            ld<T>(a, n, c, y, x, va);
            ld<T>(b, n, c, y, x, vb);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float r = 0.f;
                if (mode & 1) r += ws * va[k] + (1.f - ws) * vb[k];
                if (mode & 2) {
                    const float wc = coef[(long long)n * C + c * 8 + k];
                    r += wc * va[k] + (1.f - wc) * vb[k];
                }
                vo[k] = r * scale;
            }
            st<T>(o, n, c, y, x, vo);
        }
    }
}

// backward of attention fusion; gsum = plane sums of g*(a-b) (MODE 1 above).  ga/gb: gradient views (interior written,
// accumulate optional).  Derivation: oracle/fusion_oracle.py:_attn_branch_bwd.
template <typename T>
__global__ void attn_bwd_kernel(TV a, TV b, TV g, TV ga, TV gb, const float* __restrict__ coef, const float* __restrict__ gsum,
                                int mode, int accumulate) {
    const float EPSV = 1e-7f;
    const long long total = (long long)a.n * a.h * a.w;
    const int C = a.cb * 8;
    const long long cnt = (long long)a.n * C;
    const float scale = mode == 3 ? 0.5f : 1.f;
    GRID_STRIDE(i, total) {
        int n, c_unused, y, x;
        split_idx(i, 1, a.h, a.w, n, c_unused, y, x);
        float s1 = 0.f, s2 = 0.f, gw = 0.f;
        if (mode & 1) {
            for (int c = 0; c < a.cb; ++c) {
                float va[8], vb[8], gv[8];
                ld<T>(a, n, c, y, x, va);
                ld<T>(b, n, c, y, x, vb);
                load_grad_fold<T>(g, n, c, y, x, gv);
#pragma unroll
                for (int k = 0; k < 8; ++k) { s1 += fabsf(va[k]); s2 += fabsf(vb[k]); gw += gv[k] * (va[k] - vb[k]); }
            }
        }
        const float ssum = s1 + s2, d = fmaxf(ssum, EPSV), ws = s1 / d;
        const float pass = ssum >= EPSV ? 1.f : 0.f;          // clamp_(min=eps) passes the gradient where s >= eps
        const float gs1 = gw * scale * (1.f / d - s1 / (d * d) * pass), gs2 = gw * scale * (-s1 / (d * d) * pass);
        for (int c = 0; c < a.cb; ++c) {
            float va[8], vb[8], gv[8], oa[8], ob[8];
            ld<T>(a, n, c, y, x, va);
            ld<T>(b, n, c, y, x, vb);
            load_grad_fold<T>(g, n, c, y, x, gv);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float gk = gv[k] * scale;
                float ra = 0.f, rb = 0.f;
                if (mode & 1) {
                    ra += gk * ws + gs1 * (va[k] > 0.f ? 1.f : (va[k] < 0.f ? -1.f : 0.f));
                    rb += gk * (1.f - ws) + gs2 * (vb[k] > 0.f ? 1.f : (vb[k] < 0.f ? -1.f : 0.f));
                }
                if (mode & 2) {
                    const long long ci = (long long)n * C + c * 8 + k;
                    const float wc = coef[ci];
                    const float gwc = gsum[ci] * scale;     // dL/dwc
                    ra += gk * wc + gwc * coef[cnt + ci];
                    rb += gk * (1.f - wc) + gwc * coef[2 * cnt + ci];
                }
                oa[k] = ra;
                ob[k] = rb;
            }
            if (accumulate) {
                float o1[8], o2[8];
                ld<T>(ga, n, c, y + ga.halo, x + ga.halo, o1);
                ld<T>(gb, n, c, y + gb.halo, x + gb.halo, o2);
#pragma unroll
                for (int k = 0; k < 8; ++k) { oa[k] += o1[k]; ob[k] += o2[k]; }
            }
            st<T>(ga, n, c, y + ga.halo, x + ga.halo, oa);
            st<T>(gb, n, c, y + gb.halo, x + gb.halo, ob);
        }
    }
}

}  // namespace mmif

using namespace mmif;

#define LAUNCH_T(dtype, KERNEL, grid, ...)                                                              \
    do {                                                                                                \
        if ((dtype) == MMIF_F32) hipLaunchKernelGGL((KERNEL<float>), dim3(grid), dim3(256), 0, st, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<bf16_t>), dim3(grid), dim3(256), 0, st, __VA_ARGS__);           \
    } while (0)

static int same_nc(const mmif_tensor* a, const mmif_tensor* b) { return a->n == b->n && a->cb == b->cb && a->dtype == b->dtype; }

extern "C" int mmif_maxpool2x2_fwd(const mmif_tensor* x, const mmif_tensor* y, void* stream) {
    if (int rc = validate_tensor(x, "x")) return rc;
    if (int rc = validate_tensor(y, "y")) return rc;
    MMIF_REQUIRE(same_nc(x, y) && x->halo == 0 && y->halo == 0 && y->h == x->h / 2 && y->w == x->w / 2 && y->h > 0 && y->w > 0,
                 "maxpool2x2_fwd: shape mismatch");
    hipStream_t st = (hipStream_t)stream;
    TV tx = make_tv(x), ty = make_tv(y);
    LAUNCH_T(x->dtype, maxpool_fwd_kernel, grid_for((long long)ty.n * ty.cb * ty.h * ty.w), tx, ty);
    return check_launch("maxpool_fwd");
}

static int maxpool2x2_bwd_impl(const mmif_tensor* x, const mmif_tensor* g, const mmif_tensor* gx, int32_t accumulate, int32_t relu_mask,
                               void* stream) {
    if (int rc = validate_tensor(x, "x")) return rc;
    if (int rc = validate_tensor(g, "g")) return rc;
    if (int rc = validate_tensor(gx, "gx")) return rc;
    MMIF_REQUIRE(same_nc(x, g) && same_nc(x, gx) && x->halo == 0 && g->h == x->h / 2 && g->w == x->w / 2 && gx->h == x->h && gx->w == x->w,
                 "maxpool2x2_bwd: shape mismatch");
    hipStream_t st = (hipStream_t)stream;
    TV tx = make_tv(x), tg = make_tv(g), tgx = make_tv(gx);
    LAUNCH_T(x->dtype, maxpool_bwd_kernel, grid_for((long long)tx.n * tx.cb * tx.h * tx.w), tx, tg, tgx, accumulate, relu_mask);
    return check_launch("maxpool_bwd");
}
extern "C" int mmif_maxpool2x2_bwd(const mmif_tensor* x, const mmif_tensor* g, const mmif_tensor* gx, int32_t accumulate, void* stream) {
    return maxpool2x2_bwd_impl(x, g, gx, accumulate, 0, stream);
}
// ... followed by gx *= [x > 0] (x, the pool's input, is a ReLU output and this was the last contribution to its gradient)
extern "C" int mmif_maxpool2x2_bwd_relu(const mmif_tensor* x, const mmif_tensor* g, const mmif_tensor* gx, int32_t accumulate, void* stream) {
    return maxpool2x2_bwd_impl(x, g, gx, accumulate, 1, stream);
}

extern "C" int mmif_upsample2x_fwd(const mmif_tensor* x, const mmif_tensor* y, void* stream) {
    if (int rc = validate_tensor(x, "x")) return rc;
    if (int rc = validate_tensor(y, "y")) return rc;
    MMIF_REQUIRE(same_nc(x, y) && x->halo == 0 && y->halo == 0 && y->h >= 2 * x->h && y->w >= 2 * x->w && y->h - 2 * x->h < 2 * x->h &&
                     y->w - 2 * x->w < 2 * x->w, "upsample2x_fwd: target shape must be >= 2x the source (reflect-padded up)");
    hipStream_t st = (hipStream_t)stream;
    TV tx = make_tv(x), ty = make_tv(y);
    LAUNCH_T(x->dtype, upsample_fwd_kernel, grid_for((long long)ty.n * ty.cb * ty.h * ty.w), tx, ty);
    return check_launch("upsample_fwd");
}

static int upsample2x_bwd_impl(const mmif_tensor* g, const mmif_tensor* gx, int32_t accumulate, const mmif_tensor* xmask, void* stream) {
    if (int rc = validate_tensor(g, "g")) return rc;
    if (int rc = validate_tensor(gx, "gx")) return rc;
    MMIF_REQUIRE(same_nc(g, gx) && g->h >= 2 * gx->h && g->w >= 2 * gx->w, "upsample2x_bwd: shape mismatch");
    if (xmask != nullptr) {
        if (int rc = validate_tensor(xmask, "x")) return rc;
        MMIF_REQUIRE(same_nc(xmask, gx) && xmask->halo == 0 && xmask->h == gx->h && xmask->w == gx->w, "upsample2x_bwd_relu: x does not match gx");
    }
    hipStream_t st = (hipStream_t)stream;
    TV tg = make_tv(g), tgx = make_tv(gx), tm = make_tv(xmask != nullptr ? xmask : gx);
    LAUNCH_T(g->dtype, upsample_bwd_kernel, grid_for((long long)tgx.n * tgx.cb * tgx.h * tgx.w), tg, tgx, accumulate, tm, xmask != nullptr ? 1 : 0);
    return check_launch("upsample_bwd");
}
extern "C" int mmif_upsample2x_bwd(const mmif_tensor* g, const mmif_tensor* gx, int32_t accumulate, void* stream) {
    return upsample2x_bwd_impl(g, gx, accumulate, nullptr, stream);
}
// ... followed by gx *= [x > 0] (x: the ReLU output whose gradient gx is; this was its last contribution)
extern "C" int mmif_upsample2x_bwd_relu(const mmif_tensor* g, const mmif_tensor* gx, int32_t accumulate, const mmif_tensor* x, void* stream) {
    MMIF_REQUIRE(x != nullptr, "upsample2x_bwd_relu: x is NULL");
    return upsample2x_bwd_impl(g, gx, accumulate, x, stream);
}

extern "C" int mmif_relu_mask(const mmif_tensor* x, const mmif_tensor* g, void* stream) {
    if (int rc = validate_tensor(x, "x")) return rc;
    if (int rc = validate_tensor(g, "g")) return rc;
    MMIF_REQUIRE(same_nc(x, g) && x->halo == 0 && x->h == g->h && x->w == g->w, "relu_mask: shape mismatch");
    hipStream_t st = (hipStream_t)stream;
    TV tx = make_tv(x), tg = make_tv(g);
    LAUNCH_T(x->dtype, relu_mask_kernel, grid_for((long long)tg.n * tg.cb * tg.hs * tg.ws), tx, tg);
    return check_launch("relu_mask");
}

static int attn_mode(int32_t mode) { return mode == 0 ? 1 : (mode == 1 ? 2 : (mode == 2 ? 3 : -1)); }  // sa, ca, sca

extern "C" size_t mmif_fuse_attn_workspace(int32_t n, int32_t c) {
    return (size_t)n * ((c + 7) / 8 * 8) * (6 + 2 * PS_SLICES) * sizeof(float);  // sums + per-slice partials + per-channel coefficients
}

extern "C" int mmif_fuse_attn_fwd(const mmif_tensor* a, const mmif_tensor* b, const mmif_tensor* out, int32_t mode,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = validate_tensor(a, "a")) return rc;
    if (int rc = validate_tensor(b, "b")) return rc;
    if (int rc = validate_tensor(out, "out")) return rc;
    const int m = attn_mode(mode);
    if (m < 0) {
        set_error("only supported ['sa', 'ca', 'sca', 'wavg'] mode");
        return MMIF_EINVAL;
    }
    MMIF_REQUIRE(same_nc(a, b) && same_nc(a, out) && a->h == b->h && a->w == b->w && a->h == out->h && a->w == out->w && a->halo == 0 &&
                     b->halo == 0 && out->halo == 0, "fuse_attn_fwd: shape mismatch");
    if (workspace_bytes < mmif_fuse_attn_workspace(a->n, a->cb * 8)) {
        set_error("fuse_attn_fwd: workspace too small");
        return MMIF_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    TV ta = make_tv(a), tb = make_tv(b), to = make_tv(out);
    float* csum = (float*)workspace;
    float* coef = csum + (size_t)a->n * a->cb * 8 * (3 + 2 * PS_SLICES);
    if (m & 2) {
        const int cnt = a->n * a->cb * 8;
        float* part = csum + (size_t)cnt * 3;
        if (a->dtype == MMIF_F32) hipLaunchKernelGGL((plane_sums_kernel<float, 0>), dim3(ta.n * ta.cb, PS_SLICES), dim3(256), 0, st, ta, tb, ta, part);
        else hipLaunchKernelGGL((plane_sums_kernel<bf16_t, 0>), dim3(ta.n * ta.cb, PS_SLICES), dim3(256), 0, st, ta, tb, ta, part);
        hipLaunchKernelGGL(plane_sums_finish, dim3((cnt * 2 + 255) / 256), dim3(256), 0, st, part, csum, cnt * 2);
        hipLaunchKernelGGL(attn_coef_kernel, dim3((cnt + 255) / 256), dim3(256), 0, st, csum, coef, cnt, 1.f / ((float)a->h * a->w));
        if (int rc = check_launch("attn plane sums")) return rc;
    }
    LAUNCH_T(a->dtype, attn_fwd_kernel, grid_for((long long)to.n * to.h * to.w), ta, tb, to, coef, m);
    return check_launch("attn_fwd");
}

static int fuse_attn_bwd_impl(const mmif_tensor* a, const mmif_tensor* b, const mmif_tensor* g, const mmif_tensor* ga,
                              const mmif_tensor* gb, int32_t mode, int32_t accumulate, void* workspace, size_t workspace_bytes,
                              void* stream, bool cached) {
    if (int rc = validate_tensor(a, "a")) return rc;
    if (int rc = validate_tensor(b, "b")) return rc;
    if (int rc = validate_tensor(g, "g")) return rc;
    if (int rc = validate_tensor(ga, "ga")) return rc;
    if (int rc = validate_tensor(gb, "gb")) return rc;
    const int m = attn_mode(mode);
    if (m < 0) {
        set_error("only supported ['sa', 'ca', 'sca', 'wavg'] mode");
        return MMIF_EINVAL;
    }
    MMIF_REQUIRE(same_nc(a, b) && same_nc(a, g) && same_nc(a, ga) && same_nc(a, gb) && a->h == g->h && a->w == g->w && ga->h == a->h &&
                     gb->h == a->h && ga->w == a->w && gb->w == a->w && a->halo == 0 && b->halo == 0, "fuse_attn_bwd: shape mismatch");
    if (workspace_bytes < mmif_fuse_attn_workspace(a->n, a->cb * 8)) {
        set_error("fuse_attn_bwd: workspace too small");
        return MMIF_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    TV ta = make_tv(a), tb = make_tv(b), tg = make_tv(g), tga = make_tv(ga), tgb = make_tv(gb);
    float* csum = (float*)workspace;
    float* gsum = csum + (size_t)a->n * a->cb * 8 * 2;
    float* coef = csum + (size_t)a->n * a->cb * 8 * (3 + 2 * PS_SLICES);
    if (m & 2) {
        const int cnt = a->n * a->cb * 8;
        float* part = csum + (size_t)cnt * 3;
        const dim3 grid(ta.n * ta.cb, PS_SLICES);
        if (!cached) {   // (cached: csum / coef of these a, b are still in the workspace from mmif_fuse_attn_fwd)
            if (a->dtype == MMIF_F32) hipLaunchKernelGGL((plane_sums_kernel<float, 0>), grid, dim3(256), 0, st, ta, tb, ta, part);
            else hipLaunchKernelGGL((plane_sums_kernel<bf16_t, 0>), grid, dim3(256), 0, st, ta, tb, ta, part);
            hipLaunchKernelGGL(plane_sums_finish, dim3((cnt * 2 + 255) / 256), dim3(256), 0, st, part, csum, cnt * 2);
            hipLaunchKernelGGL(attn_coef_kernel, dim3((cnt + 255) / 256), dim3(256), 0, st, csum, coef, cnt, 1.f / ((float)a->h * a->w));
        }
        if (a->dtype == MMIF_F32) hipLaunchKernelGGL((plane_sums_kernel<float, 1>), grid, dim3(256), 0, st, ta, tb, tg, part);
        else hipLaunchKernelGGL((plane_sums_kernel<bf16_t, 1>), grid, dim3(256), 0, st, ta, tb, tg, part);
        hipLaunchKernelGGL(plane_sums_finish, dim3((cnt + 255) / 256), dim3(256), 0, st, part, gsum, cnt);
        if (int rc = check_launch("attn plane sums (bwd)")) return rc;
    }
    LAUNCH_T(a->dtype, attn_bwd_kernel, grid_for((long long)ta.n * ta.h * ta.w), ta, tb, tg, tga, tgb, coef, gsum, m, accumulate);
    return check_launch("attn_bwd");
}

extern "C" int mmif_fuse_attn_bwd(const mmif_tensor* a, const mmif_tensor* b, const mmif_tensor* g, const mmif_tensor* ga,
                                  const mmif_tensor* gb, int32_t mode, int32_t accumulate, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    return fuse_attn_bwd_impl(a, b, g, ga, gb, mode, accumulate, workspace, workspace_bytes, stream, false);
}
// the same when `workspace` is the buffer mmif_fuse_attn_fwd ran on for these a, b (same mode) and nothing has written it since: the
// channel sums and coefficients are reused instead of recomputed (one pass over a and b less)
extern "C" int mmif_fuse_attn_bwd_cached(const mmif_tensor* a, const mmif_tensor* b, const mmif_tensor* g, const mmif_tensor* ga,
                                         const mmif_tensor* gb, int32_t mode, int32_t accumulate, void* workspace, size_t workspace_bytes,
                                         void* stream) {
    return fuse_attn_bwd_impl(a, b, g, ga, gb, mode, accumulate, workspace, workspace_bytes, stream, true);
}
