// Image-side convolution layers: Cin == 1 (first encoder conv) and Cout == 1 (last decoder conv).
// Reference: ConvLayer(1,16) core/model.py:73,77,118,169 / ConvLayer(1,16,ksize=1) :326 and
// ConvLayer(16,1,act=None) :86,131,179 / ConvLayer(64,1,ksize=1) :344.  Images are fp32 [n][h][w].
// These layers are HBM-bound (K = 9 or M = 1): plain VALU kernels, fp32 math, T only on the
// feature-map side.
#include "common.hpp"
#include "reduce_defer.hpp"

namespace mmif {

constexpr int ITILE = 16;

__device__ inline float img_reflect(const float* img, int h, int w, int y, int x) {
    y = min(max(reflect_idx(y, h), 0), h - 1);
    x = min(max(reflect_idx(x, w), 0), w - 1);
    return img[(long long)y * w + x];
}

// ---------------------------------------------------------------- Cin == 1 forward
template <typename T, int KS>
__global__ __launch_bounds__(256) void image_in_fwd_kernel(const float* __restrict__ img, TV ty, const float* __restrict__ w,
                                                           const float* __restrict__ bias, int cout, int relu, int tiles_x) {
    constexpr int KK = KS * KS, P = KS / 2;
    extern __shared__ float sm[];  // [cout*KK] weights, [cout] bias
    float* wsm = sm;
    float* bsm = sm + cout * KK;
    for (int e = threadIdx.x; e < cout * KK; e += 256) wsm[e] = w[e];
    for (int e = threadIdx.x; e < cout; e += 256) bsm[e] = bias ? bias[e] : 0.f;
    __syncthreads();
    const int tx = threadIdx.x & 15, tyy = threadIdx.x >> 4;
    const int x = (blockIdx.x % tiles_x) * ITILE + tx, y = (blockIdx.x / tiles_x) * ITILE + tyy;
    const int in_ = blockIdx.y;
    if (y >= ty.h || x >= ty.w) return;
    const float* im = img + (long long)in_ * ty.h * ty.w;
    float nb[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) nb[t] = img_reflect(im, ty.h, ty.w, y + t / KS - P, x + t % KS - P);
    for (int b = 0; b < ty.cb; ++b) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int o = b * 8 + i;
            float r = 0.f;
            if (o < cout) {
                r = bsm[o];
#pragma unroll
                for (int t = 0; t < KK; ++t) r = fmaf(nb[t], wsm[o * KK + t], r);
                if (relu) r = fmaxf(r, 0.f);
            }
            v[i] = r;
        }
        Elem<T>::store(ty.base + ty.gidx(in_, b, y, x) * Elem<T>::gran_bytes, v);
    }
}

// ---------------------------------------------------------------- Cin == 1 wgrad
// dw[o][tap] = sum_p g[p][o] * img[R(p+tap-P)], db[o] = sum_p g[p][o].
// blockIdx.y = channel block (8 output channels per thread: 8*KK + 8 accumulators as packed fp32 pairs, v_pk_fma_f32);
// one pixel per thread per iteration (grid-stride), wave-shuffle + LDS block reduction at the end; the partial of
// (block, 16-channel group og) keeps the layout [16 o][KK] then [16] bias sums, each channel block filling its half.
typedef float f32x2_i __attribute__((ext_vector_type(2)));
template <typename T, int KS>
__global__ __launch_bounds__(256) void image_in_wgrad_kernel(const float* __restrict__ img, TV tg, float* __restrict__ partial,
                                                             int cout, long long npix) {
    constexpr int KK = KS * KS, P = KS / 2, PER = 16 * KK + 16, HALF = 8 * KK + 8;
    __shared__ float red[4][HALF];
    const int tid = threadIdx.x, cb = blockIdx.y, n_og = (gridDim.y + 1) / 2;
    f32x2_i acc[4][KK], accb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        accb[k] = (f32x2_i){0.f, 0.f};
#pragma unroll
        for (int t = 0; t < KK; ++t) acc[k][t] = (f32x2_i){0.f, 0.f};
    }
    const int hw = tg.h * tg.w;
    for (long long pix = (long long)blockIdx.x * 256 + tid; pix < npix; pix += (long long)gridDim.x * 256) {
        const int in_ = (int)(pix / hw), r = (int)(pix % hw);
        const int y = r / tg.w, x = r % tg.w;
        float v[8];
        load_grad_fold<T>(tg, in_, cb, y, x, v);
        const float* im = img + (long long)in_ * hw;
        // reflect the KS rows / cols once (the index arithmetic, not the FMAs, dominated this kernel)
        int ry[KS], rx[KS];
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            ry[u] = min(max(reflect_idx(y + u - P, tg.h), 0), tg.h - 1) * tg.w;
            rx[u] = min(max(reflect_idx(x + u - P, tg.w), 0), tg.w - 1);
        }
        float iv[KK];
#pragma unroll
        for (int t = 0; t < KK; ++t) iv[t] = im[ry[t / KS] + rx[t % KS]];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x2_i gp = {v[2 * k], v[2 * k + 1]};
            accb[k] += gp;
#pragma unroll
            for (int t = 0; t < KK; ++t) acc[k][t] = __builtin_elementwise_fma(gp, (f32x2_i){iv[t], iv[t]}, acc[k][t]);
        }
    }
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
#pragma unroll
            for (int t = 0; t < KK; ++t) {
                float v = hlf ? acc[k][t].y : acc[k][t].x;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                if (lane == 0) red[wave][(2 * k + hlf) * KK + t] = v;
            }
            float v = hlf ? accb[k].y : accb[k].x;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) red[wave][8 * KK + 2 * k + hlf] = v;
        }
    __syncthreads();
    float* dst = partial + ((long long)blockIdx.x * n_og + (cb >> 1)) * PER;
    for (int e = tid; e < HALF; e += 256) {
        const float t = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
        if (e < 8 * KK) dst[(cb & 1) * 8 * KK + e] = t;
        else dst[16 * KK + (cb & 1) * 8 + (e - 8 * KK)] = t;
    }
    (void)cout;
}

template <int KS>
__global__ __launch_bounds__(64 * RED_SLICES) void image_in_wgrad_reduce(const float* __restrict__ partial, float* __restrict__ dw,
                                                             float* __restrict__ db, int cout, int G, int n_og, int accumulate) {
    constexpr int KK = KS * KS, PER = 16 * KK + 16;
    __shared__ float red[RED_SLICES][64];
    const int o_local = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + o_local;
    long long off = 0;
    bool valid = false;
    if (idx < cout * KK) {
        const int o = idx / KK, tap = idx % KK;
        off = (long long)(o / 16) * PER + (o % 16) * KK + tap;
        valid = true;
    } else if (idx < cout * KK + cout) {
        const int o = idx - cout * KK;
        off = (long long)(o / 16) * PER + 16 * KK + (o % 16);
        valid = true;
    }
    const float t = partial_sum(partial, off, (long long)n_og * PER, G, valid, red);
    if (slice == 0 && valid) {
        if (idx < cout * KK) dw[idx] = accumulate ? dw[idx] + t : t;
        else if (db) db[idx - cout * KK] = accumulate ? db[idx - cout * KK] + t : t;
    }
}

// ---------------------------------------------------------------- Cout == 1 forward
template <typename T, int KS>
__global__ __launch_bounds__(256) void image_out_fwd_kernel(TV tx, const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ img, int cin, int relu, int tiles_x) {
    constexpr int KK = KS * KS, P = KS / 2;
    extern __shared__ float wsm[];  // [cin][KK]
    for (int e = threadIdx.x; e < cin * KK; e += 256) wsm[e] = w[e];
    __syncthreads();
    const int txx = threadIdx.x & 15, tyy = threadIdx.x >> 4;
    const int x = (blockIdx.x % tiles_x) * ITILE + txx, y = (blockIdx.x / tiles_x) * ITILE + tyy;
    const int in_ = blockIdx.y;
    if (y >= tx.h || x >= tx.w) return;
    float r = bias ? bias[0] : 0.f;
    // reflected rows / cols once per pixel (granule offsets inside a plane)
    int ry[KS], rx[KS];
#pragma unroll
    for (int u = 0; u < KS; ++u) {
        ry[u] = min(max(reflect_idx(y + u - P, tx.h), 0), tx.h - 1) * tx.ws;
        rx[u] = min(max(reflect_idx(x + u - P, tx.w), 0), tx.w - 1);
    }
    for (int b = 0; b < tx.cb; ++b) {
        const char* plane = tx.base + tx.gidx(in_, b, 0, 0) * Elem<T>::gran_bytes;
#pragma unroll
        for (int t = 0; t < KK; ++t) {
            float v[8];
            Elem<T>::load(plane + (long long)(ry[t / KS] + rx[t % KS]) * Elem<T>::gran_bytes, v);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (b * 8 + i < cin) r = fmaf(v[i], wsm[(b * 8 + i) * KK + t], r);
        }
    }
    if (relu) r = fmaxf(r, 0.f);
    img[((long long)in_ * tx.h + y) * tx.w + x] = r;
}

// bf16, 3x3, 16 input channels (the last layer of every PFNet / DenseFuse decoder): the kernel above issues 18 granule loads per
// pixel through the vector-memory path (288 B per pixel for 32 B of data: load-issue bound, 50 us at B=32 256x256).  Here a block
// stages the 18 x 18 x 2-plane reflect-padded tile into LDS once (2.5 loads per thread) and every pixel reads its 18 granules from
// there; weights live in registers.  Same FMA order (channel block, tap, channel) => bit-identical.
// (round 3: also for fp32 tensors -- 32-byte granules, the same order: 117 -> 45 us in the fp32 step)
template <typename T>
__global__ __launch_bounds__(256) void image_out_fwd_tiled_kernel(TV tx, const float* __restrict__ w, const float* __restrict__ bias,
                                                                  float* __restrict__ img, int relu, int tiles_x) {
    constexpr int TP = ITILE + 2;
    constexpr int GU = Elem<T>::gran_bytes / 16;   // uint4 per granule
    // rows of 32 granules (512 B): ds_read_b128 serves the wave's four 16-pixel rows in lane groups that mix two rows; a row stride that
    // is a multiple of 256 B keeps each group on 64 distinct banks (the natural 18-granule stride made half of them 2-way conflicts)
    constexpr int TS = 32;
    __shared__ __attribute__((aligned(16))) uint4 s_x[2][TP * TS * GU];
    const int tid = threadIdx.x;
    const int x0 = (blockIdx.x % tiles_x) * ITILE, y0 = (blockIdx.x / tiles_x) * ITILE;
    const int in_ = blockIdx.y;
    for (int e = tid; e < 2 * TP * TP; e += 256) {
        const int b = e / (TP * TP), p = e - b * (TP * TP);
        const int y = min(max(reflect_idx(y0 + p / TP - 1, tx.h), 0), tx.h - 1);
        const int x = min(max(reflect_idx(x0 + p % TP - 1, tx.w), 0), tx.w - 1);
        const uint4* src = reinterpret_cast<const uint4*>(tx.base + tx.gidx(in_, b, y, x) * Elem<T>::gran_bytes);
#pragma unroll
        for (int q = 0; q < GU; ++q) s_x[b][((p / TP) * TS + p % TP) * GU + q] = src[q];
    }
    float wr[16][9];
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[c][t] = w[c * 9 + t];
    __syncthreads();
    const int txx = tid & 15, tyy = tid >> 4;
    const int x = x0 + txx, y = y0 + tyy;
    if (y >= tx.h || x >= tx.w) return;
    float r = bias ? bias[0] : 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float v[8];
            Elem<T>::load(&s_x[b][((tyy + t / 3) * TS + txx + t % 3) * GU], v);
#pragma unroll
            for (int i = 0; i < 8; ++i) r = fmaf(v[i], wr[b * 8 + i][t], r);
        }
    if (relu) r = fmaxf(r, 0.f);
    img[((long long)in_ * tx.h + y) * tx.w + x] = r;
}

// ---------------------------------------------------------------- Cout == 1 dgrad
// gx[y][x][c] (+)= sum_{u,v} W[0][c][k-1-u][k-1-v] * g0(y+u-p, x+v-p) over the stored domain of gx.
template <typename T, int KS>
__global__ __launch_bounds__(256) void image_out_dgrad_kernel(const float* __restrict__ gimg, const float* __restrict__ yimg,
                                                              const float* __restrict__ w, TV tmask, TV tgx, int cin,
                                                              unsigned long long mask_bits, unsigned long long accum_bits,
                                                              int tiles_x) {
    constexpr int KK = KS * KS, P = KS / 2;
    extern __shared__ float wsm[];  // [cin][KK]
    for (int e = threadIdx.x; e < cin * KK; e += 256) wsm[e] = w[e];
    __syncthreads();
    const int txx = threadIdx.x & 15, tyy = threadIdx.x >> 4;
    const unsigned tile = xcd_tile(blockIdx.x, gridDim.x);   // neighbouring tiles share cache lines of gx: keep them on one XCD
    const int xs = (int)(tile % (unsigned)tiles_x) * ITILE + txx, ys = (int)(tile / (unsigned)tiles_x) * ITILE + tyy;
    const int in_ = blockIdx.y;
    if (ys >= tgx.hs || xs >= tgx.ws) return;
    const int y = ys - tgx.halo, x = xs - tgx.halo;
    const int H = tgx.h, W = tgx.w;
    float g[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) {
        const int yy = y + t / KS - P, xx = x + t % KS - P;
        float v = 0.f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
            const long long i = ((long long)in_ * H + yy) * W + xx;
            v = gimg[i];
            if (yimg != nullptr && !(yimg[i] > 0.f)) v = 0.f;
        }
        g[t] = v;
    }
    for (int b = 0; b < tgx.cb; ++b) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float r = 0.f;
            const int c = b * 8 + i;
            if (c < cin) {
#pragma unroll
                for (int t = 0; t < KK; ++t) r = fmaf(g[t], wsm[c * KK + (KK - 1 - t)], r);
            }
            v[i] = r;
        }
        char* dst = tgx.base + tgx.gidx(in_, b, ys, xs) * Elem<T>::gran_bytes;
        if ((accum_bits >> b) & 1ull) {
            float old[8];
            Elem<T>::load(dst, old);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += old[i];
        }
        if ((mask_bits >> b) & 1ull) {
            float xm[8];
            load_act_reflect<T>(tmask, in_, b, y, x, xm);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = xm[i] > 0.f ? v[i] : 0.f;
        }
        Elem<T>::store(dst, v);
    }
}

// ---------------------------------------------------------------- Cout == 1 wgrad
// dw[0][c][tap] = sum_p g[p] * xpad[p+tap-P][c] (xpad = reflect-padded x), db = sum_p g[p].
// Re-indexed by the SOURCE position pp = p + tap - P of the padded domain [-P, h-1+P] x [-P, w-1+P]:
//   dw[c][tap] = sum_pp xpad[pp][c] * g[pp - tap + P]        (g zero outside the image)
// so each thread loads ONE activation granule per position (instead of one per tap) plus KK scalar g values (fp32 image,
// cached).  blockIdx.y = channel block (8 input channels, packed fp32 pair accumulators); partial per (block, 16-channel
// group cg): [16 c][KK] then [1] bias (written by channel block 0).
template <typename T, int KS>
__global__ __launch_bounds__(256) void image_out_wgrad_kernel(TV tx, const float* __restrict__ gimg, const float* __restrict__ yimg,
                                                              float* __restrict__ partial, long long npix) {
    constexpr int KK = KS * KS, P = KS / 2, PER = 16 * KK + 1, HALF = 8 * KK + 1;
    __shared__ float red[4][HALF];
    const int tid = threadIdx.x, cb = blockIdx.y, n_cg = (gridDim.y + 1) / 2;
    f32x2_i acc[4][KK];
    float accb = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int t = 0; t < KK; ++t) acc[k][t] = (f32x2_i){0.f, 0.f};
    const int H = tx.h, W = tx.w, hp = H + 2 * P, wp = W + 2 * P;
    const long long hwp = (long long)hp * wp, total = (long long)tx.n * hwp;
    for (long long pos = (long long)blockIdx.x * 256 + tid; pos < total; pos += (long long)gridDim.x * 256) {
        const int in_ = (int)(pos / hwp), r = (int)(pos % hwp);
        const int py = r / wp - P, px = r % wp - P;
        float v[8];
        load_act_reflect<T>(tx, in_, cb, py, px, v);
        const float* gi = gimg + (long long)in_ * H * W;
        const float* yi = yimg ? yimg + (long long)in_ * H * W : nullptr;
        // KS rows / cols of g around pp: clamped offsets + validity once per row / col, branch-free loads
        int oy[KS], ox[KS];
        bool vy[KS], vx[KS];
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            const int yy = py - u + P, xx = px - u + P;
            vy[u] = yy >= 0 && yy < H;
            vx[u] = xx >= 0 && xx < W;
            oy[u] = min(max(yy, 0), H - 1) * W;
            ox[u] = min(max(xx, 0), W - 1);
        }
        float g[KK];
#pragma unroll
        for (int t = 0; t < KK; ++t) {
            const int o = oy[t / KS] + ox[t % KS];
            float gv = gi[o];
            if (yi != nullptr && !(yi[o] > 0.f)) gv = 0.f;
            g[t] = (vy[t / KS] && vx[t % KS]) ? gv : 0.f;
        }
        if (py >= 0 && py < H && px >= 0 && px < W) accb += g[P * KS + P];   // tap (P, P): g at pp itself
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x2_i xp = {v[2 * k], v[2 * k + 1]};
#pragma unroll
            for (int t = 0; t < KK; ++t) acc[k][t] = __builtin_elementwise_fma(xp, (f32x2_i){g[t], g[t]}, acc[k][t]);
        }
    }
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf)
#pragma unroll
            for (int t = 0; t < KK; ++t) {
                float v = hlf ? acc[k][t].y : acc[k][t].x;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                if (lane == 0) red[wave][(2 * k + hlf) * KK + t] = v;
            }
    {
        float v = accb;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[wave][8 * KK] = v;
    }
    __syncthreads();
    float* dst = partial + ((long long)blockIdx.x * n_cg + (cb >> 1)) * PER;
    for (int e = tid; e < HALF; e += 256) {
        const float t = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
        if (e < 8 * KK) dst[(cb & 1) * 8 * KK + e] = t;
        else if (cb == 0) dst[16 * KK] = t;
    }
    (void)npix;
}

template <int KS>
__global__ __launch_bounds__(64 * RED_SLICES) void image_out_wgrad_reduce(const float* __restrict__ partial, float* __restrict__ dw,
                                                              float* __restrict__ db, int cin, int G, int n_cg, int accumulate) {
    constexpr int KK = KS * KS, PER = 16 * KK + 1;
    __shared__ float red[RED_SLICES][64];
    const int o_local = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + o_local;
    long long off = 0;
    bool valid = false;
    if (idx < cin * KK) {
        const int c = idx / KK, tap = idx % KK;
        off = (long long)(c / 16) * PER + (c % 16) * KK + tap;
        valid = true;
    } else if (idx == cin * KK) {
        off = 16 * KK;  // bias sum lives in channel group 0
        valid = true;
    }
    const float t = partial_sum(partial, off, (long long)n_cg * PER, G, valid, red);
    if (slice == 0 && valid) {
        if (idx < cin * KK) dw[idx] = accumulate ? dw[idx] + t : t;
        else if (db) db[0] = accumulate ? db[0] + t : t;
    }
}

constexpr int IMG_G = 512;

// (csrc/image_bwd.hip: the fused backward of the 16 -> 1 layer leaves its block partials in image_out_wgrad_kernel's layout)
int image_out_wgrad_reduce3_launch(const float* ws, float* dw, float* db, int cin, int G, int n_cg, int accumulate, hipStream_t st) {
    const int n = cin * 9 + 1;
    hipLaunchKernelGGL((image_out_wgrad_reduce<3>), dim3(cdiv(n, 64)), dim3(64 * RED_SLICES), 0, st, ws, dw, db, cin, G, n_cg, accumulate);
    return check_launch("image_out_wgrad_reduce");
}

}  // namespace mmif

using namespace mmif;
int image_out_fwd16_launch(const TV& tx, const float* w, const float* bias, float* img, int relu, hipStream_t st);   // image_bwd.hip

#define DISPATCH_T_KS(dtype, ks, CALL)                 \
    do {                                               \
        if ((dtype) == MMIF_F32) {                     \
            if ((ks) == 3) { CALL(float, 3); } else { CALL(float, 1); } \
        } else {                                       \
            if ((ks) == 3) { CALL(bf16_t, 3); } else { CALL(bf16_t, 1); } \
        }                                              \
    } while (0)

extern "C" size_t mmif_conv2d_image_wgrad_workspace(int32_t c, int32_t ksize) {
    return (size_t)IMG_G * cdiv(c, 16) * (16 * ksize * ksize + 16) * sizeof(float);
}

extern "C" int mmif_conv2d_image_in_fwd(const float* img, const float* w, const float* bias, const mmif_tensor* y,
                                        int32_t cout, int32_t ksize, int32_t relu, void* stream) {
    if (int rc = validate_tensor(y, "y")) return rc;
    MMIF_REQUIRE(ksize == 1 || ksize == 3, "image_in_fwd: ksize must be 1 or 3");
    MMIF_REQUIRE(y->halo == 0 && cout <= y->cb * 8 && cout > 0, "image_in_fwd: bad output view");
    MMIF_REQUIRE(ksize == 1 || (y->h >= 2 && y->w >= 2), "reflect padding needs h,w >= 2");
    TV ty = make_tv(y);
    const int tiles_x = cdiv(ty.w, ITILE), tiles_y = cdiv(ty.h, ITILE);
    const size_t shm = (size_t)(cout * ksize * ksize + cout) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
#define CALL(T, KS) hipLaunchKernelGGL((image_in_fwd_kernel<T, KS>), dim3(tiles_x * tiles_y, ty.n), dim3(256), shm, st, img, ty, w, bias, cout, relu, tiles_x)
    DISPATCH_T_KS(y->dtype, ksize, CALL);
#undef CALL
    return check_launch("image_in_fwd");
}

extern "C" int mmif_conv2d_image_in_wgrad(const float* img, const mmif_tensor* gy, float* dw, float* db, int32_t cout,
                                          int32_t ksize, int32_t accumulate, void* workspace, size_t workspace_bytes,
                                          void* stream) {
    if (int rc = validate_tensor(gy, "gy")) return rc;
    MMIF_REQUIRE(ksize == 1 || ksize == 3, "image_in_wgrad: ksize must be 1 or 3");
    if (workspace_bytes < mmif_conv2d_image_wgrad_workspace(cout, ksize)) {
        set_error("image_in_wgrad: workspace too small");
        return MMIF_EWORKSPACE;
    }
    TV tg = make_tv(gy);
    const long long npix = (long long)tg.n * tg.h * tg.w;
    const int G = (int)(cdiv(npix, 256) < IMG_G ? cdiv(npix, 256) : IMG_G);
    const int n_og = cdiv(cout, 16);
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
#define CALL(T, KS) hipLaunchKernelGGL((image_in_wgrad_kernel<T, KS>), dim3(G, tg.cb), dim3(256), 0, st, img, tg, ws, cout, npix)
    DISPATCH_T_KS(gy->dtype, ksize, CALL);
#undef CALL
    if (int rc = check_launch("image_in_wgrad")) return rc;
    const int n = cout * ksize * ksize + cout;
    if (ksize == 3) hipLaunchKernelGGL((image_in_wgrad_reduce<3>), dim3(cdiv(n, 64)), dim3(64 * RED_SLICES), 0, st, ws, dw, db, cout, G, n_og, accumulate);
    else hipLaunchKernelGGL((image_in_wgrad_reduce<1>), dim3(cdiv(n, 64)), dim3(64 * RED_SLICES), 0, st, ws, dw, db, cout, G, n_og, accumulate);
    return check_launch("image_in_wgrad_reduce");
}

extern "C" int mmif_conv2d_image_out_fwd(const mmif_tensor* x, const float* w, const float* bias, float* img, int32_t cin,
                                         int32_t ksize, int32_t relu, void* stream) {
    if (int rc = validate_tensor(x, "x")) return rc;
    MMIF_REQUIRE(ksize == 1 || ksize == 3, "image_out_fwd: ksize must be 1 or 3");
    MMIF_REQUIRE(x->halo == 0 && cin <= x->cb * 8 && cin > 0, "image_out_fwd: bad input view");
    MMIF_REQUIRE(ksize == 1 || (x->h >= 2 && x->w >= 2), "reflect padding needs h,w >= 2");
    TV tx = make_tv(x);
    const int tiles_x = cdiv(tx.w, ITILE), tiles_y = cdiv(tx.h, ITILE);
    const size_t shm = (size_t)cin * ksize * ksize * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (ksize == 3 && cin == 16 && x->cb == 2) {
        // bf16: the persistent kernel of csrc/image_bwd.hip (round 6; 36 us against the tiled kernel's 40 at B = 32 256 x 256); fp32: the tiled kernel
        if (x->dtype == MMIF_BF16) return image_out_fwd16_launch(tx, w, bias, img, relu, st);
        hipLaunchKernelGGL(image_out_fwd_tiled_kernel<float>, dim3(tiles_x * tiles_y, tx.n), dim3(256), 0, st, tx, w, bias, img, relu, tiles_x);
        return check_launch("image_out_fwd");
    }
#define CALL(T, KS) hipLaunchKernelGGL((image_out_fwd_kernel<T, KS>), dim3(tiles_x * tiles_y, tx.n), dim3(256), shm, st, tx, w, bias, img, cin, relu, tiles_x)
    DISPATCH_T_KS(x->dtype, ksize, CALL);
#undef CALL
    return check_launch("image_out_fwd");
}

extern "C" int mmif_conv2d_image_out_dgrad(const float* gimg, const float* y_img, const float* w, const mmif_tensor* x,
                                           const mmif_tensor* gx, int32_t cin, int32_t ksize, uint64_t mask_bits,
                                           uint64_t accum_bits, void* stream) {
    if (int rc = validate_tensor(gx, "gx")) return rc;
    MMIF_REQUIRE(ksize == 1 || ksize == 3, "image_out_dgrad: ksize must be 1 or 3");
    MMIF_REQUIRE(gx->halo >= ksize / 2, "image_out_dgrad: gx needs halo >= ksize/2");
    MMIF_REQUIRE(cin <= gx->cb * 8, "image_out_dgrad: gx view too small");
    TV tgx = make_tv(gx);
    TV tm = tgx;
    if (mask_bits) {
        MMIF_REQUIRE(x != nullptr, "image_out_dgrad: mask_bits set but x is NULL");
        if (int rc = validate_tensor(x, "x")) return rc;
        MMIF_REQUIRE(x->halo == 0 && x->dtype == gx->dtype && x->h == gx->h && x->w == gx->w && x->n == gx->n && x->cb >= gx->cb,
                     "image_out_dgrad: x / gx mismatch");
        tm = make_tv(x);
    }
    const int tiles_x = cdiv(tgx.ws, ITILE), tiles_y = cdiv(tgx.hs, ITILE);
    const size_t shm = (size_t)cin * ksize * ksize * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
#define CALL(T, KS) hipLaunchKernelGGL((image_out_dgrad_kernel<T, KS>), dim3(tiles_x * tiles_y, tgx.n), dim3(256), shm, st, gimg, y_img, w, tm, tgx, cin, (unsigned long long)mask_bits, (unsigned long long)accum_bits, tiles_x)
    DISPATCH_T_KS(gx->dtype, ksize, CALL);
#undef CALL
    return check_launch("image_out_dgrad");
}

extern "C" int mmif_conv2d_image_out_wgrad(const mmif_tensor* x, const float* gimg, const float* y_img, float* dw, float* db,
                                           int32_t cin, int32_t ksize, int32_t accumulate, void* workspace,
                                           size_t workspace_bytes, void* stream) {
    if (int rc = validate_tensor(x, "x")) return rc;
    MMIF_REQUIRE(ksize == 1 || ksize == 3, "image_out_wgrad: ksize must be 1 or 3");
    MMIF_REQUIRE(x->halo == 0, "image_out_wgrad: x must be an activation (halo 0)");
    if (workspace_bytes < mmif_conv2d_image_wgrad_workspace(cin, ksize)) {
        set_error("image_out_wgrad: workspace too small");
        return MMIF_EWORKSPACE;
    }
    TV tx = make_tv(x);
    const long long npix = (long long)tx.n * tx.h * tx.w;
    const int G = (int)(cdiv(npix, 256) < IMG_G ? cdiv(npix, 256) : IMG_G);
    const int n_cg = cdiv(cin, 16);
    hipStream_t st = (hipStream_t)stream;
    float* ws = defer_ws((float*)workspace, (size_t)G * n_cg * (16 * ksize * ksize + 1) * sizeof(float));
#define CALL(T, KS) hipLaunchKernelGGL((image_out_wgrad_kernel<T, KS>), dim3(G, tx.cb), dim3(256), 0, st, tx, gimg, y_img, ws, npix)
    DISPATCH_T_KS(x->dtype, ksize, CALL);
#undef CALL
    if (int rc = check_launch("image_out_wgrad")) return rc;
    const int n = cin * ksize * ksize + 1;
    {
        RedJob J;
        J.partial = ws; J.dw = dw; J.db = db; J.type = RED_IMAGE_OUT; J.sl = RED_SLICES; J.G = G; J.accumulate = accumulate;
        J.p0 = cin; J.p1 = ksize; J.p2 = n_cg; J.p3 = 0; J.nvb = cdiv(n, 64);
        if (defer_push(J)) return MMIF_OK;
    }
    if (ksize == 3) hipLaunchKernelGGL((image_out_wgrad_reduce<3>), dim3(cdiv(n, 64)), dim3(64 * RED_SLICES), 0, st, ws, dw, db, cin, G, n_cg, accumulate);
    else hipLaunchKernelGGL((image_out_wgrad_reduce<1>), dim3(cdiv(n, 64)), dim3(64 * RED_SLICES), 0, st, ws, dw, db, cin, G, n_cg, accumulate);
    return check_launch("image_out_wgrad_reduce");
}
