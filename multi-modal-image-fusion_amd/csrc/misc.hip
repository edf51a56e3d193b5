// Layout conversion, element-wise fusion, clip+Adam, and the library's error plumbing.
#include <math.h>
#include <stdarg.h>

#include "common.hpp"

namespace mmif {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int cached_num_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cus[dev] == 0) {
        int n = 0;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
        if (const char* e = getenv("MMIF_NUM_CUS")) n = atoi(e) > 0 ? atoi(e) : n;
        cus[dev] = n;
    }
    return cus[dev];
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return MMIF_ELAUNCH;
    }
    return MMIF_OK;
}

int validate_tensor(const mmif_tensor* t, const char* name) {
    if (t == nullptr || t->data == nullptr) {
        set_error("%s: null tensor", name);
        return MMIF_EINVAL;
    }
    if (t->dtype != MMIF_F32 && t->dtype != MMIF_BF16) {
        set_error("%s: bad dtype %d", name, t->dtype);
        return MMIF_EINVAL;
    }
    if (t->n <= 0 || t->h <= 0 || t->w <= 0 || t->halo < 0 || t->halo > 1 || t->cb <= 0 || t->cb_off < 0 ||
        t->cb_off + t->cb > t->cb_total) {
        set_error("%s: bad extent n=%d h=%d w=%d halo=%d cb_total=%d cb_off=%d cb=%d", name, t->n, t->h, t->w, t->halo,
                  t->cb_total, t->cb_off, t->cb);
        return MMIF_EINVAL;
    }
    if (t->cb > 64) {
        set_error("%s: views wider than 512 channels are not supported", name);
        return MMIF_EINVAL;
    }
    return MMIF_OK;
}

// ---------------------------------------------------------------- layout
template <typename T>
__global__ void nchw_to_blocked_kernel(const float* __restrict__ src, int c, TV dst) {
    const long long total = (long long)dst.n * dst.cb * dst.h * dst.w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int n, b, y, x;
        split_idx(i, dst.cb, dst.h, dst.w, n, b, y, x);
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int ch = b * 8 + k;
            v[k] = ch < c ? src[(((long long)n * c + ch) * dst.h + y) * dst.w + x] : 0.f;
        }
        Elem<T>::store(dst.base + dst.gidx(n, b, y + dst.halo, x + dst.halo) * Elem<T>::gran_bytes, v);
    }
}

template <typename T>
__global__ void blocked_to_nchw_kernel(TV src, float* __restrict__ dst, int c) {
    const long long total = (long long)src.n * src.cb * src.h * src.w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int n, b, y, x;
        split_idx(i, src.cb, src.h, src.w, n, b, y, x);
        float v[8];
        load_grad_fold<T>(src, n, b, y, x, v);  // halo 0: plain load; halo 1: folded gradient
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int ch = b * 8 + k;
            if (ch < c) dst[(((long long)n * c + ch) * src.h + y) * src.w + x] = v[k];
        }
    }
}

template <typename T>
__global__ void zero_kernel(TV t) {
    const long long per = (long long)t.cb * t.plane;
    const long long total = (long long)t.n * per;
    float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = i / per;
        const long long r = i % per;
        Elem<T>::store(t.base + ((long long)n * t.img + (long long)t.cb_off * t.plane + r) * Elem<T>::gran_bytes, z);
    }
}

// fold targets of one (n, channel block): rows 1 and h-2 (all x), then cols 1 and w-2 (remaining y)
// zero_src: also zero the halo granules this target folded in.  Every halo pixel mirrors onto exactly ONE interior pixel
// (reflect_idx is a function) and every target is handled by one thread, so there is no hazard and no second launch; only
// valid for h, w >= 4 (below that the two fold rows / cols coincide and the host keeps the separate zero_halo pass).
template <typename T>
__global__ void fold_targets_kernel(TV t, int zero_src) {
    const int per = 2 * t.w + 2 * t.h;
    const long long total = (long long)t.n * t.cb * per;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int j = i % per, c = (i / per) % t.cb, n = i / ((long long)per * t.cb);
        int y, x;
        bool dup = false;
        if (j < t.w) { y = 1; x = j; }
        else if (j < 2 * t.w) { y = t.h - 2; x = j - t.w; dup = (t.h - 2 == 1); }
        else if (j < 2 * t.w + t.h) { y = j - 2 * t.w; x = 1; dup = (y == 1 || y == t.h - 2); }
        else { y = j - 2 * t.w - t.h; x = t.w - 2; dup = (y == 1 || y == t.h - 2) || (t.w - 2 == 1); }
        if (dup || y < 0 || y >= t.h || x < 0 || x >= t.w) continue;
        float v[8];
        load_grad_fold<T>(t, n, c, y, x, v);   // reads this pixel's interior + halo sources only: no hazard between targets
        Elem<T>::store(t.base + t.gidx(n, c, y + 1, x + 1) * Elem<T>::gran_bytes, v);
        if (zero_src) {
            const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            // stored coordinates of the mirrored halo rows / cols (as in load_grad_fold), -1 = none
            const int ys[3] = {y + 1, (y == 1) ? 0 : -1, (y == t.h - 2) ? t.h + 1 : -1};
            const int xs[3] = {x + 1, (x == 1) ? 0 : -1, (x == t.w - 2) ? t.w + 1 : -1};
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b)
                    if ((a | b) != 0 && ys[a] >= 0 && xs[b] >= 0) Elem<T>::store(t.base + t.gidx(n, c, ys[a], xs[b]) * Elem<T>::gran_bytes, z);
        }
    }
}

template <typename T>
__global__ void zero_halo_kernel(TV t) {
    const int per = 2 * t.ws + 2 * t.hs;
    const long long total = (long long)t.n * t.cb * per;
    const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int j = i % per, c = (i / per) % t.cb, n = i / ((long long)per * t.cb);
        int ys, xs;
        if (j < t.ws) { ys = 0; xs = j; }
        else if (j < 2 * t.ws) { ys = t.hs - 1; xs = j - t.ws; }
        else if (j < 2 * t.ws + t.hs) { ys = j - 2 * t.ws; xs = 0; }
        else { ys = j - 2 * t.ws - t.hs; xs = t.ws - 1; }
        Elem<T>::store(t.base + t.gidx(n, c, ys, xs) * Elem<T>::gran_bytes, z);
    }
}

// ---------------------------------------------------------------- element fusion (core/fusion.py:21-29)
template <typename T>
__global__ void fuse_elem_fwd_kernel(TV a, TV b, TV o, int mode) {
    const long long total = (long long)o.n * o.cb * o.h * o.w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int n, c, y, x;
        split_idx(i, o.cb, o.h, o.w, n, c, y, x);
        float va[8], vb[8], vo[8];
        Elem<T>::load(a.base + a.gidx(n, c, y, x) * Elem<T>::gran_bytes, va);
        Elem<T>::load(b.base + b.gidx(n, c, y, x) * Elem<T>::gran_bytes, vb);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            vo[k] = mode == MMIF_FUSE_SUM ? va[k] + vb[k] : (mode == MMIF_FUSE_MEAN ? (va[k] + vb[k]) * 0.5f : fmaxf(va[k], vb[k]));
        Elem<T>::store(o.base + o.gidx(n, c, y, x) * Elem<T>::gran_bytes, vo);
    }
}

// position-wise on the STORED domain of g (g, ga, gb share the halo): ga = g * dmode_a * [a>0]
template <typename T>
__global__ void fuse_elem_bwd_kernel(TV a, TV b, TV g, TV ga, TV gb, int mode, int relu_mask) {
    const long long total = (long long)g.n * g.cb * g.hs * g.ws;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int n, c, ys, xs;
        split_idx(i, g.cb, g.hs, g.ws, n, c, ys, xs);
        float vg[8], va[8], vb[8], oa[8], ob[8];
        Elem<T>::load(g.base + g.gidx(n, c, ys, xs) * Elem<T>::gran_bytes, vg);
        const bool need_ab = relu_mask || mode == MMIF_FUSE_MAX;
        if (need_ab) {
            load_act_reflect<T>(a, n, c, ys - g.halo, xs - g.halo, va);
            load_act_reflect<T>(b, n, c, ys - g.halo, xs - g.halo, vb);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float da = 1.f, db = 1.f;
            if (mode == MMIF_FUSE_MEAN) da = db = 0.5f;
            else if (mode == MMIF_FUSE_MAX) {
                da = va[k] > vb[k] ? 1.f : (va[k] == vb[k] ? 0.5f : 0.f);
                db = 1.f - da;
            }
            if (relu_mask) {
                if (!(va[k] > 0.f)) da = 0.f;
                if (!(vb[k] > 0.f)) db = 0.f;
            }
            oa[k] = vg[k] * da;
            ob[k] = vg[k] * db;
        }
        Elem<T>::store(ga.base + ga.gidx(n, c, ys, xs) * Elem<T>::gran_bytes, oa);
        Elem<T>::store(gb.base + gb.gidx(n, c, ys, xs) * Elem<T>::gran_bytes, ob);
    }
}

// ---------------------------------------------------------------- clip_grad_norm_ + Adam (train.py:72-75,319)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n, float* __restrict__ partial) {
    __shared__ float red[16];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) s = fmaf(g[i], g[i], s);
    const float t = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// partial[np] holds sums of squares; stats[0] = scaled norm, stats[1] = clip coefficient * grad_scale come out of the Adam launch below
// clip coefficient + Adam in one launch: every block forms the total of the sums of squares itself -- the arithmetic of the former one-block
// finish launch with its 256 threads, so the norm, the coefficient and the update are bit-identical to the two-launch form -- block 0 also publishes them
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float step_size, float beta1,
                                                   float beta2, float inv_bc2_sqrt, float eps,
                                                   const float* __restrict__ partial, int np, float grad_scale, float max_norm,
                                                   float* __restrict__ stats, float* __restrict__ norm_out) {
    __shared__ float red[16];
    __shared__ float s_gs;
    float s = 0.f;
    for (int i = threadIdx.x; i < np; i += blockDim.x) s += partial[i];
    const float t = block_sum(s, red);
    if (threadIdx.x == 0) {
        const float norm = sqrtf(t) * grad_scale;
        float coef = 1.f;
        if (max_norm > 0.f) coef = fminf(1.f, max_norm / (norm + 1e-6f));
        s_gs = coef * grad_scale;
        if (blockIdx.x == 0) {
            stats[0] = norm;
            stats[1] = coef * grad_scale;
            if (norm_out) norm_out[0] = norm;
        }
    }
    __syncthreads();
    const float gs = s_gs;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i] * gs;
        const float mi = m[i] + (gi - m[i]) * (1.f - beta1);             // exp_avg.lerp_(grad, 1-beta1)
        const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;         // mul_(beta2).addcmul_(g, g, 1-beta2)
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

}  // namespace mmif

using namespace mmif;

extern "C" const char* mmif_version(void) { return "mmif-hip 0.1 (gfx950)"; }
extern "C" const char* mmif_last_error(void) { return g_err; }

static int grid_for(long long total) {
    long long b = (total + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" int mmif_nchw_to_blocked(const float* src, int32_t c, const mmif_tensor* dst, void* stream) {
    if (int rc = validate_tensor(dst, "dst")) return rc;
    MMIF_REQUIRE(c > 0 && c <= dst->cb * 8, "nchw_to_blocked: c=%d does not fit the view (%d blocks)", c, dst->cb);
    TV t = make_tv(dst);
    const long long total = (long long)t.n * t.cb * t.h * t.w;
    if (dst->dtype == MMIF_F32) hipLaunchKernelGGL(nchw_to_blocked_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, c, t);
    else hipLaunchKernelGGL(nchw_to_blocked_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, c, t);
    return check_launch("nchw_to_blocked");
}

extern "C" int mmif_blocked_to_nchw(const mmif_tensor* src, float* dst, int32_t c, void* stream) {
    if (int rc = validate_tensor(src, "src")) return rc;
    MMIF_REQUIRE(c > 0 && c <= src->cb * 8, "blocked_to_nchw: c=%d does not fit the view (%d blocks)", c, src->cb);
    TV t = make_tv(src);
    const long long total = (long long)t.n * t.cb * t.h * t.w;
    if (src->dtype == MMIF_F32) hipLaunchKernelGGL(blocked_to_nchw_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, t, dst, c);
    else hipLaunchKernelGGL(blocked_to_nchw_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, t, dst, c);
    return check_launch("blocked_to_nchw");
}

extern "C" int mmif_zero(const mmif_tensor* t, void* stream) {
    if (int rc = validate_tensor(t, "t")) return rc;
    TV v = make_tv(t);
    const long long total = (long long)v.n * v.cb * v.plane;
    if (t->dtype == MMIF_F32) hipLaunchKernelGGL(zero_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, v);
    else hipLaunchKernelGGL(zero_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, v);
    return check_launch("zero");
}

extern "C" int mmif_fold_halo(const mmif_tensor* t, void* stream) {
    if (int rc = validate_tensor(t, "t")) return rc;
    MMIF_REQUIRE(t->halo == 1, "fold_halo: tensor has no halo");
    MMIF_REQUIRE(t->h >= 2 && t->w >= 2, "fold_halo: reflect padding needs h,w >= 2");
    mmif_tensor u = *t;
    u.flags &= ~MMIF_T_FOLDED;  // the fold itself must read the halo
    TV v = make_tv(&u);
    const long long n1 = (long long)v.n * v.cb * (2 * v.w + 2 * v.h), n2 = (long long)v.n * v.cb * (2 * v.ws + 2 * v.hs);
    hipStream_t st = (hipStream_t)stream;
    const int fused = (v.h >= 4 && v.w >= 4) ? 1 : 0;
    if (t->dtype == MMIF_F32) hipLaunchKernelGGL(fold_targets_kernel<float>, dim3(grid_for(n1)), dim3(256), 0, st, v, fused);
    else hipLaunchKernelGGL(fold_targets_kernel<bf16_t>, dim3(grid_for(n1)), dim3(256), 0, st, v, fused);
    if (int rc = check_launch("fold_targets")) return rc;
    if (fused) return MMIF_OK;
    if (t->dtype == MMIF_F32) hipLaunchKernelGGL(zero_halo_kernel<float>, dim3(grid_for(n2)), dim3(256), 0, st, v);
    else hipLaunchKernelGGL(zero_halo_kernel<bf16_t>, dim3(grid_for(n2)), dim3(256), 0, st, v);
    return check_launch("zero_halo");
}

static int same_shape(const mmif_tensor* a, const mmif_tensor* b) {
    return a->n == b->n && a->h == b->h && a->w == b->w && a->cb == b->cb && a->dtype == b->dtype;
}

extern "C" int mmif_fuse_elem_fwd(const mmif_tensor* a, const mmif_tensor* b, const mmif_tensor* out, int32_t mode,
                                  void* stream) {
    if (int rc = validate_tensor(a, "a")) return rc;
    if (int rc = validate_tensor(b, "b")) return rc;
    if (int rc = validate_tensor(out, "out")) return rc;
    if (mode < MMIF_FUSE_SUM || mode > MMIF_FUSE_MAX) {
        set_error("only supported ['sum', 'mean', 'max'] mode");
        return MMIF_EINVAL;
    }
    MMIF_REQUIRE(same_shape(a, b) && same_shape(a, out) && a->halo == 0 && b->halo == 0 && out->halo == 0,
                 "fuse_elem_fwd: shape/halo mismatch");
    TV ta = make_tv(a), tb = make_tv(b), to = make_tv(out);
    const long long total = (long long)to.n * to.cb * to.h * to.w;
    if (a->dtype == MMIF_F32) hipLaunchKernelGGL(fuse_elem_fwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, ta, tb, to, mode);
    else hipLaunchKernelGGL(fuse_elem_fwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, ta, tb, to, mode);
    return check_launch("fuse_elem_fwd");
}

extern "C" int mmif_fuse_elem_bwd(const mmif_tensor* a, const mmif_tensor* b, const mmif_tensor* g, const mmif_tensor* ga,
                                  const mmif_tensor* gb, int32_t mode, int32_t relu_mask, void* stream) {
    if (int rc = validate_tensor(g, "g")) return rc;
    if (int rc = validate_tensor(ga, "ga")) return rc;
    if (int rc = validate_tensor(gb, "gb")) return rc;
    if (mode < MMIF_FUSE_SUM || mode > MMIF_FUSE_MAX) {
        set_error("only supported ['sum', 'mean', 'max'] mode");
        return MMIF_EINVAL;
    }
    MMIF_REQUIRE(same_shape(g, ga) && same_shape(g, gb) && g->halo == ga->halo && g->halo == gb->halo,
                 "fuse_elem_bwd: gradient shape/halo mismatch");
    TV tg = make_tv(g), tga = make_tv(ga), tgb = make_tv(gb);
    TV ta = tg, tb = tg;
    if (relu_mask || mode == MMIF_FUSE_MAX) {
        if (int rc = validate_tensor(a, "a")) return rc;
        if (int rc = validate_tensor(b, "b")) return rc;
        MMIF_REQUIRE(same_shape(a, g) && same_shape(b, g) && a->halo == 0 && b->halo == 0, "fuse_elem_bwd: a/b mismatch");
        ta = make_tv(a);
        tb = make_tv(b);
    }
    const long long total = (long long)tg.n * tg.cb * tg.hs * tg.ws;
    if (g->dtype == MMIF_F32) hipLaunchKernelGGL(fuse_elem_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, ta, tb, tg, tga, tgb, mode, relu_mask);
    else hipLaunchKernelGGL(fuse_elem_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, ta, tb, tg, tga, tgb, mode, relu_mask);
    return check_launch("fuse_elem_bwd");
}

constexpr int ADAM_PARTIALS = 1024;

extern "C" size_t mmif_clip_adam_workspace(int64_t numel) {
    (void)numel;
    return (size_t)(ADAM_PARTIALS + 8) * sizeof(float);
}

extern "C" int mmif_clip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t numel,
                                   float lr, float beta1, float beta2, float eps, int32_t step, float max_norm,
                                   float grad_scale, float* norm_out, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    MMIF_REQUIRE(numel > 0 && step >= 1, "clip_adam_step: numel=%lld step=%d", (long long)numel, step);
    if (workspace_bytes < mmif_clip_adam_workspace(numel)) {
        set_error("clip_adam_step: workspace too small");
        return MMIF_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    float* stats = partial + ADAM_PARTIALS;
    int nb = (int)((numel + 255) / 256);
    if (nb > ADAM_PARTIALS) nb = ADAM_PARTIALS;
    hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, st, grads, (long long)numel, partial);
    if (int rc = check_launch("sumsq")) return rc;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    hipLaunchKernelGGL(adam_kernel, dim3(nb), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, (long long)numel, step_size,
                       beta1, beta2, inv_bc2_sqrt, eps, partial, nb, grad_scale, max_norm, stats, norm_out);
    return check_launch("adam");
}
