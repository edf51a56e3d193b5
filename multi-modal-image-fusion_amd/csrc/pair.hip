// PFNetv2's self-learned fusion (reference core/model.py:120-124,134-141): the SAME tiny conv stack
// ConvLayer(2,2) -> ConvLayer(2,2) -> ConvLayer(2,1,act=None) is applied to every channel pair
// (feat1[:, i], feat2[:, i]), i = 0..63, and the 64 results are concatenated and added to feat1 + feat2.
// The reference runs it as a 64-iteration Python loop (192 conv launches per forward).  Here one "pair
// conv" is ONE launch over blocked tensors: every channel of the two operand tensors A, B is an independent
// 2-channel image sharing the 2x2x3x3 (or 1x2x3x3) weights, so the 8 channels of a granule are 8 lanes of the
// same arithmetic and nothing is stacked, reshaped or copied:
//     OA[c] = act(bias[0] + sum_tap w[0][0][tap]*A[c] + w[0][1][tap]*B[c])      (OB with w[1], when nout = 2)
// HBM-bound VALU work (18 FMA per output value): fp32 math, T only in storage.
// Gradients follow the engine's convention: dgrad writes the padded domain [h+2][w+2] (halo = 1), the caller
// folds the halo (mmif_fold_halo); ReLU masks of the producer are applied per channel block (mask_bits).
#include "common.hpp"

namespace mmif {

#define PAIR_GRID_STRIDE(i, total) \
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (long long)gridDim.x * blockDim.x)

static int pair_grid(long long total) {
    long long b = (total + 255) / 256;
    if (b > 16384) b = 16384;
    if (b < 1) b = 1;
    return (int)b;
}

template <typename T, int NOUT>
__global__ __launch_bounds__(256) void pairconv_fwd_kernel(TV a, TV b, const float* __restrict__ w, const float* __restrict__ bias,
                                                            TV oa, TV ob, int relu, TV r1, TV r2, int has_res) {
    const long long total = (long long)a.n * a.cb * a.h * a.w;
    PAIR_GRID_STRIDE(i, total) {
        const int x = i % a.w, y = (i / a.w) % a.h, c = (i / ((long long)a.w * a.h)) % a.cb, n = i / ((long long)a.w * a.h * a.cb);
        float acc[NOUT][8];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const float bo = bias ? bias[o] : 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[o][k] = bo;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float va[8], vb[8];
            load_act_reflect<T>(a, n, c, y + t / 3 - 1, x + t % 3 - 1, va);
            load_act_reflect<T>(b, n, c, y + t / 3 - 1, x + t % 3 - 1, vb);
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                const float wa = w[(o * 2 + 0) * 9 + t], wb = w[(o * 2 + 1) * 9 + t];
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[o][k] = fmaf(wb, vb[k], fmaf(wa, va[k], acc[o][k]));
            }
        }
        if (relu) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[o][k] = fmaxf(acc[o][k], 0.f);
        }
        if (has_res) {  // + feat1 + feat2 (core/model.py:141), added to output 0
            float u[8], v[8];
            Elem<T>::load(r1.base + r1.gidx(n, c, y, x) * Elem<T>::gran_bytes, u);
            Elem<T>::load(r2.base + r2.gidx(n, c, y, x) * Elem<T>::gran_bytes, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[0][k] = (acc[0][k] + u[k]) + v[k];
        }
        Elem<T>::store(oa.base + oa.gidx(n, c, y, x) * Elem<T>::gran_bytes, acc[0]);
        if (NOUT == 2) Elem<T>::store(ob.base + ob.gidx(n, c, y, x) * Elem<T>::gran_bytes, acc[NOUT - 1]);
    }
}

// gxa/gxb (halo 1): padded-domain gradient w.r.t. the operands A, B:
//   gx_c[p] = sum_o sum_tap w[o][c][tap] * g_o[p - tap + 1]   (g zero outside the image), p in [-1, h] x [-1, w]
//   (+ add[p]: the residual path's gradient, same for A and B)   then  * [x_c(R(p)) > 0] for the masked channel blocks.
template <typename T, int NOUT>
__global__ __launch_bounds__(256) void pairconv_dgrad_kernel(TV ga, TV gb, const float* __restrict__ w, TV xa, TV xb, TV gxa, TV gxb,
                                                              unsigned long long mask_bits, TV add, int has_add) {
    const long long total = (long long)gxa.n * gxa.cb * gxa.hs * gxa.ws;
    PAIR_GRID_STRIDE(i, total) {
        const int xs = i % gxa.ws, ys = (i / gxa.ws) % gxa.hs, c = (i / ((long long)gxa.ws * gxa.hs)) % gxa.cb,
                  n = i / ((long long)gxa.ws * gxa.hs * gxa.cb);
        const int py = ys - gxa.halo, px = xs - gxa.halo;
        float da[8], db[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) da[k] = db[k] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int qy = py - (t / 3) + 1, qx = px - (t % 3) + 1;
            float g0[8], g1[8];
            load_grad_fold<T>(ga, n, c, qy, qx, g0);
            if (NOUT == 2) load_grad_fold<T>(gb, n, c, qy, qx, g1);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                da[k] = fmaf(w[(0 * 2 + 0) * 9 + t], g0[k], da[k]);
                db[k] = fmaf(w[(0 * 2 + 1) * 9 + t], g0[k], db[k]);
                if (NOUT == 2) {
                    da[k] = fmaf(w[(1 * 2 + 0) * 9 + t], g1[k], da[k]);
                    db[k] = fmaf(w[(1 * 2 + 1) * 9 + t], g1[k], db[k]);
                }
            }
        }
        if (has_add) {
            float r[8];
            load_grad_fold<T>(add, n, c, py, px, r);
#pragma unroll
            for (int k = 0; k < 8; ++k) { da[k] += r[k]; db[k] += r[k]; }
        }
        if ((mask_bits >> c) & 1ull) {
            float va[8], vb[8];
            load_act_reflect<T>(xa, n, c, py, px, va);
            load_act_reflect<T>(xb, n, c, py, px, vb);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                da[k] = va[k] > 0.f ? da[k] : 0.f;
                db[k] = vb[k] > 0.f ? db[k] : 0.f;
            }
        }
        Elem<T>::store(gxa.base + gxa.gidx(n, c, ys, xs) * Elem<T>::gran_bytes, da);
        Elem<T>::store(gxb.base + gxb.gidx(n, c, ys, xs) * Elem<T>::gran_bytes, db);
    }
}

// dw[o][c][tap] = sum over (n, channel, p) of g_o[p] * reflect_pad(x_c)[p + tap - 1];  db[o] = sum g_o.
// Persistent blocks, NOUT*19 register accumulators per thread, block reduction -> partial[block][NOUT*19].
constexpr int PAIR_WG_BLOCKS = 2048;

template <typename T, int NOUT>
__global__ __launch_bounds__(256) void pairconv_wgrad_kernel(TV xa, TV xb, TV ga, TV gb, float* __restrict__ partial) {
    constexpr int PER = NOUT * 19;
    __shared__ float red[4][PER];
    float acc[NOUT][2][9], accb[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        accb[o] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[o][0][t] = acc[o][1][t] = 0.f;
    }
    const long long total = (long long)xa.n * xa.cb * xa.h * xa.w;
    PAIR_GRID_STRIDE(i, total) {
        const int x = i % xa.w, y = (i / xa.w) % xa.h, c = (i / ((long long)xa.w * xa.h)) % xa.cb, n = i / ((long long)xa.w * xa.h * xa.cb);
        float g[NOUT][8];
        load_grad_fold<T>(ga, n, c, y, x, g[0]);
        if (NOUT == 2) load_grad_fold<T>(gb, n, c, y, x, g[NOUT - 1]);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += g[o][k];
            accb[o] += s;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float va[8], vb[8];
            load_act_reflect<T>(xa, n, c, y + t / 3 - 1, x + t % 3 - 1, va);
            load_act_reflect<T>(xb, n, c, y + t / 3 - 1, x + t % 3 - 1, vb);
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                float sa = 0.f, sb = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    sa = fmaf(g[o][k], va[k], sa);
                    sb = fmaf(g[o][k], vb[k], sb);
                }
                acc[o][0][t] += sa;
                acc[o][1][t] += sb;
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float v = acc[o][cc][t];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                if (lane == 0) red[wave][(o * 2 + cc) * 9 + t] = v;
            }
        float v = accb[o];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[wave][NOUT * 18 + o] = v;
    }
    __syncthreads();
    if (threadIdx.x < PER) {
        const int e = threadIdx.x;
        partial[(long long)blockIdx.x * PER + e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    }
}

// one block: output e = fixed-order sum of the G partials (4 interleaved chains of 64 lanes each -> tree)
__global__ __launch_bounds__(256) void pairconv_wgrad_reduce(const float* __restrict__ partial, int G, int per, int nout,
                                                             float* __restrict__ dw, float* __restrict__ db, int accumulate) {
    __shared__ float red[16];
    for (int e = 0; e < per; ++e) {
        float s = 0.f;
        for (int g = threadIdx.x; g < G; g += 256) s += partial[(long long)g * per + e];
        const float tot = block_sum(s, red);
        if (threadIdx.x == 0) {
            float* dst = e < nout * 18 ? dw + e : db + (e - nout * 18);
            *dst = accumulate ? *dst + tot : tot;
        }
    }
}

static bool same_shape(const mmif_tensor* a, const mmif_tensor* b) {
    return a->n == b->n && a->cb == b->cb && a->h == b->h && a->w == b->w && a->dtype == b->dtype;
}

}  // namespace mmif

using namespace mmif;

#define PAIR_LAUNCH(dtype, nout, kern, grid, ...)                                                                        \
    do {                                                                                                                 \
        if ((dtype) == MMIF_F32) {                                                                                       \
            if ((nout) == 2) hipLaunchKernelGGL((kern<float, 2>), dim3(grid), dim3(256), 0, st, __VA_ARGS__);            \
            else hipLaunchKernelGGL((kern<float, 1>), dim3(grid), dim3(256), 0, st, __VA_ARGS__);                        \
        } else {                                                                                                         \
            if ((nout) == 2) hipLaunchKernelGGL((kern<bf16_t, 2>), dim3(grid), dim3(256), 0, st, __VA_ARGS__);           \
            else hipLaunchKernelGGL((kern<bf16_t, 1>), dim3(grid), dim3(256), 0, st, __VA_ARGS__);                       \
        }                                                                                                                \
    } while (0)

extern "C" int mmif_pairconv_fwd(const mmif_tensor* a, const mmif_tensor* b, const float* w, const float* bias, int32_t nout,
                                 const mmif_tensor* oa, const mmif_tensor* ob, int32_t relu, const mmif_tensor* res1,
                                 const mmif_tensor* res2, void* stream) {
    if (int rc = validate_tensor(a, "a")) return rc;
    if (int rc = validate_tensor(b, "b")) return rc;
    if (int rc = validate_tensor(oa, "oa")) return rc;
    MMIF_REQUIRE(nout == 1 || nout == 2, "pairconv_fwd: nout must be 1 or 2 (got %d)", nout);
    MMIF_REQUIRE(w != nullptr, "pairconv_fwd: w is NULL");
    MMIF_REQUIRE(same_shape(a, b) && same_shape(a, oa) && a->halo == 0 && b->halo == 0 && oa->halo == 0, "pairconv_fwd: shape mismatch");
    MMIF_REQUIRE(a->h >= 2 && a->w >= 2, "pairconv_fwd: reflect padding needs h, w >= 2");
    if (nout == 2) {
        MMIF_REQUIRE(ob != nullptr, "pairconv_fwd: ob is NULL with nout = 2");
        if (int rc = validate_tensor(ob, "ob")) return rc;
        MMIF_REQUIRE(same_shape(a, ob) && ob->halo == 0, "pairconv_fwd: ob shape mismatch");
    }
    const bool has_res = res1 != nullptr && res2 != nullptr;
    if (has_res) {
        if (int rc = validate_tensor(res1, "res1")) return rc;
        if (int rc = validate_tensor(res2, "res2")) return rc;
        MMIF_REQUIRE(same_shape(a, res1) && same_shape(a, res2) && res1->halo == 0 && res2->halo == 0, "pairconv_fwd: residual shape mismatch");
    }
    hipStream_t st = (hipStream_t)stream;
    TV ta = make_tv(a), tb = make_tv(b), toa = make_tv(oa), tob = make_tv(nout == 2 ? ob : oa);
    TV t1 = make_tv(has_res ? res1 : a), t2 = make_tv(has_res ? res2 : a);
    const long long total = (long long)ta.n * ta.cb * ta.h * ta.w;
    PAIR_LAUNCH(a->dtype, nout, pairconv_fwd_kernel, pair_grid(total), ta, tb, w, bias, toa, tob, relu, t1, t2, has_res ? 1 : 0);
    return check_launch("pairconv_fwd");
}

extern "C" int mmif_pairconv_dgrad(const mmif_tensor* ga, const mmif_tensor* gb, const float* w, int32_t nout, const mmif_tensor* xa,
                                   const mmif_tensor* xb, const mmif_tensor* gxa, const mmif_tensor* gxb, uint64_t mask_bits,
                                   const mmif_tensor* add, void* stream) {
    if (int rc = validate_tensor(ga, "ga")) return rc;
    if (int rc = validate_tensor(gxa, "gxa")) return rc;
    if (int rc = validate_tensor(gxb, "gxb")) return rc;
    MMIF_REQUIRE(nout == 1 || nout == 2, "pairconv_dgrad: nout must be 1 or 2 (got %d)", nout);
    MMIF_REQUIRE(w != nullptr, "pairconv_dgrad: w is NULL");
    MMIF_REQUIRE(same_shape(ga, gxa) && same_shape(ga, gxb) && gxa->halo == 1 && gxb->halo == 1, "pairconv_dgrad: gx must be halo-1 views of g's shape");
    if (nout == 2) {
        MMIF_REQUIRE(gb != nullptr, "pairconv_dgrad: gb is NULL with nout = 2");
        if (int rc = validate_tensor(gb, "gb")) return rc;
        MMIF_REQUIRE(same_shape(ga, gb), "pairconv_dgrad: gb shape mismatch");
    }
    if (mask_bits) {
        MMIF_REQUIRE(xa != nullptr && xb != nullptr, "pairconv_dgrad: mask_bits without xa/xb");
        if (int rc = validate_tensor(xa, "xa")) return rc;
        if (int rc = validate_tensor(xb, "xb")) return rc;
        MMIF_REQUIRE(same_shape(ga, xa) && same_shape(ga, xb) && xa->halo == 0 && xb->halo == 0, "pairconv_dgrad: xa/xb shape mismatch");
    }
    if (add) {
        if (int rc = validate_tensor(add, "add")) return rc;
        MMIF_REQUIRE(same_shape(ga, add), "pairconv_dgrad: add shape mismatch");
    }
    hipStream_t st = (hipStream_t)stream;
    TV tga = make_tv(ga), tgb = make_tv(nout == 2 ? gb : ga), txa = make_tv(mask_bits ? xa : ga), txb = make_tv(mask_bits ? xb : ga);
    TV tgxa = make_tv(gxa), tgxb = make_tv(gxb), tadd = make_tv(add ? add : ga);
    const long long total = (long long)tgxa.n * tgxa.cb * tgxa.hs * tgxa.ws;
    PAIR_LAUNCH(ga->dtype, nout, pairconv_dgrad_kernel, pair_grid(total), tga, tgb, w, txa, txb, tgxa, tgxb, (unsigned long long)mask_bits, tadd,
                add ? 1 : 0);
    return check_launch("pairconv_dgrad");
}

extern "C" size_t mmif_pairconv_wgrad_workspace(void) { return (size_t)PAIR_WG_BLOCKS * 2 * 19 * sizeof(float); }

extern "C" int mmif_pairconv_wgrad(const mmif_tensor* xa, const mmif_tensor* xb, const mmif_tensor* ga, const mmif_tensor* gb, int32_t nout,
                                   float* dw, float* db, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = validate_tensor(xa, "xa")) return rc;
    if (int rc = validate_tensor(xb, "xb")) return rc;
    if (int rc = validate_tensor(ga, "ga")) return rc;
    MMIF_REQUIRE(nout == 1 || nout == 2, "pairconv_wgrad: nout must be 1 or 2 (got %d)", nout);
    MMIF_REQUIRE(dw != nullptr && db != nullptr, "pairconv_wgrad: dw/db is NULL");
    MMIF_REQUIRE(same_shape(xa, xb) && same_shape(xa, ga) && xa->halo == 0 && xb->halo == 0, "pairconv_wgrad: shape mismatch");
    if (nout == 2) {
        MMIF_REQUIRE(gb != nullptr, "pairconv_wgrad: gb is NULL with nout = 2");
        if (int rc = validate_tensor(gb, "gb")) return rc;
        MMIF_REQUIRE(same_shape(xa, gb), "pairconv_wgrad: gb shape mismatch");
    }
    if (workspace == nullptr || workspace_bytes < mmif_pairconv_wgrad_workspace()) {
        set_error("pairconv_wgrad: workspace too small");
        return MMIF_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    TV txa = make_tv(xa), txb = make_tv(xb), tga = make_tv(ga), tgb = make_tv(nout == 2 ? gb : ga);
    const long long total = (long long)txa.n * txa.cb * txa.h * txa.w;
    int G = pair_grid(total);
    if (G > PAIR_WG_BLOCKS) G = PAIR_WG_BLOCKS;
    float* partial = (float*)workspace;
    PAIR_LAUNCH(xa->dtype, nout, pairconv_wgrad_kernel, G, txa, txb, tga, tgb, partial);
    hipLaunchKernelGGL(pairconv_wgrad_reduce, dim3(1), dim3(256), 0, st, partial, G, nout * 19, nout, dw, db, accumulate);
    return check_launch("pairconv_wgrad");
}
