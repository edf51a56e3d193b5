// Backward of the decoders' LAST layer -- ConvLayer(16, 1, 3x3, reflect) of PFNetv1 / DenseFuse / PFNetv2 / VIFNet (reference
// core/model.py:86,178; core/block.py:26-99) -- as ONE launch (round 6): input gradient (reflect-padding adjoint applied, ReLU mask of
// the layer's input), weight gradient and bias gradient.  It replaces image_out_wgrad_kernel + its reduce's producer, image_out_dgrad_kernel
// and the stand-alone fold of the halo (mmif_fold_halo): 98 us -> one pass at B = 32 256 x 256 (profiles/r06_*).
//
// Why one pass is natural here.  With e[t] = g0(pp - t + 1) (g0 = the image gradient, zero outside the image; t = (u, v) a tap), a position
// pp of the PADDED domain [-1, H] x [-1, W] contributes
//     gxpad[pp][c]  = sum_t W[c][t] e[t]                 (gradient of the padded input: a 9-tap correlation with the flipped kernel)
//     dW[c][t]     += xpad[pp][c] e[t]                   (weight gradient, indexed by the SOURCE position as image_out_wgrad_kernel does)
// and xpad[pp] = x[q], gx[q] += gxpad[pp] for the interior pixel q that pp reflects onto.  So a thread that owns (q, one 8-channel block)
// needs ONE activation granule (its own, straight from global memory: coalesced 16-byte lanes, no halo) and the 3 x 3 neighbourhood of g0
// around each pp in {q} u {ring positions that reflect onto q} (rows 1 / H-2, columns 1 / W-2 have one or three of those) -- a 12 x 36
// fp32 tile in LDS per 8 x 32-pixel tile.  Both products come out of the same nine values: 72 packed FMAs per position and channel block,
// weights in scalar registers (the channel block is wave uniform).
//
// Structure: persistent blocks of 8 waves (waves 0-3 channel block 0, waves 4-7 channel block 1), XCD-aware tile walk, the NEXT tile's
// activation granule and g0 values requested before the current tile's arithmetic (one barrier per tile, double-buffered g0 tile),
// weight-gradient accumulators in registers over all of a block's tiles, one partial per block reduced in fixed order by
// image_out_wgrad_reduce (deterministic, no atomics).  HBM-bound by design: 67 MB of activations in, 67 MB of gradient out, 8 MB of image.
#include "common.hpp"
#include "reduce_defer.hpp"

namespace mmif {

constexpr int IB_TW = 32, IB_TH = 8;             // output tile: 8 rows x 32 columns = 256 pixels
// g0 tile: rows y0 - 2 .. y0 + TH + 2, columns x0 - 2 .. x0 + TW + 2 (a ring position looks one pixel beyond the ring: pp = H reads rows
// H - 1 .. H + 1 and q = H - 2 may be the tile's last row)
constexpr int IB_GW = IB_TW + 5, IB_GH = IB_TH + 5;
constexpr int IB_GP = 40;                        // row pitch of the g0 tile in floats (a 32-lane group reads one row: conflict free at any pitch)
static_assert(IB_GW * IB_GH <= 512, "one g0 value per thread");
constexpr int IB_PER = 16 * 9 + 1;               // floats per block partial: dW[16][9], db  (image_out_wgrad_reduce<3>'s layout, n_cg = 1)

typedef float f32x2_b __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512, 2) void image_out_bwd16_kernel(TV tx, TV tgx, const float* __restrict__ gimg, const float* __restrict__ yimg,
                                                                 const float* __restrict__ w, float* __restrict__ partial, int tiles_x,
                                                                 int tiles_y, int total) {
    __shared__ float s_g[2][IB_GH * IB_GP];
    __shared__ float s_red[8][73];
    const int tid = threadIdx.x;
    const int cb = __builtin_amdgcn_readfirstlane(tid >> 8);     // wave uniform: the weights of this channel block live in SGPRs
    const int pix = tid & 255, ty = pix >> 5, txx = pix & 31;
    const int H = tx.h, W = tx.w;
    const TileWalk tw = xcd_walk(total, gridDim.x, blockIdx.x);
    const float* wq = w + cb * 72;      // [8 channels][9 taps] of this channel block
    float wr[8][9];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[c][t] = wq[c * 9 + t];
    f32x2_b acc[4][9];
    float accb = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[k][t] = (f32x2_b){0.f, 0.f};

    // ---- requests of a tile: this thread's activation granule + (threads < 432) one value of the g0 tile
    auto tile_origin = [&](int tile, int& in_, int& y0, int& x0) {
        const int tpi = tiles_x * tiles_y;
        in_ = tile / tpi;
        const int r = tile - in_ * tpi;
        y0 = (r / tiles_x) * IB_TH;
        x0 = (r % tiles_x) * IB_TW;
    };
    auto load_x = [&](int tile) {
        int in_, y0, x0;
        tile_origin(tile, in_, y0, x0);
        const int y = min(y0 + ty, H - 1), x = min(x0 + txx, W - 1);
        return *reinterpret_cast<const uint4*>(tx.base + tx.gidx(in_, cb, y, x) * 16);
    };
    auto load_g = [&](int tile) {
        float v = 0.f;
        if (tid < IB_GH * IB_GW) {
            int in_, y0, x0;
            tile_origin(tile, in_, y0, x0);
            const int y = y0 - 2 + tid / IB_GW, x = x0 - 2 + tid % IB_GW;
            if (y >= 0 && y < H && x >= 0 && x < W) {
                const long long i = ((long long)in_ * H + y) * W + x;
                v = gimg[i];
                if (yimg != nullptr && !(yimg[i] > 0.f)) v = 0.f;     // ReLU of THIS layer (act = ReLU variants); PFNet's last layer has none
            }
        }
        return v;
    };
    // Requests run AHEAD: the activation granules of the next three tiles and the g0 values of the next one are in flight while a tile is
    // worked on (one tile ahead left 16 KB of reads in flight per CU: 2 TB/s; the loop is unrolled by three so that the ring is registers)
    uint4 xq0 = make_uint4(0, 0, 0, 0), xq1 = xq0, xq2 = xq0;
    float gn = 0.f;
    if (tw.count > 0) { xq0 = load_x(tw.first); gn = load_g(tw.first); }
    if (tw.count > 1) xq1 = load_x(tw.first + tw.stride);
    if (tw.count > 2) xq2 = load_x(tw.first + 2 * tw.stride);
    auto body = [&](int k, uint4& xq) {
        if (k >= tw.count) return;
        const int tile = tw.first + k * tw.stride;
        if (tid < IB_GH * IB_GW) s_g[k & 1][(tid / IB_GW) * IB_GP + tid % IB_GW] = gn;
        const uint4 xc = xq;
        __syncthreads();
        if (k + 1 < tw.count) gn = load_g(tile + tw.stride);
        if (k + 3 < tw.count) xq = load_x(tile + 3 * tw.stride);
        int in_, y0, x0;
        tile_origin(tile, in_, y0, x0);
        const int qy = y0 + ty, qx = x0 + txx;
        if (qy < H && qx < W) {
            const uint32_t xw[4] = {xc.x, xc.y, xc.z, xc.w};
            f32x2_b xv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xv[i] = (f32x2_b){__uint_as_float(xw[i] << 16), __uint_as_float(xw[i] & 0xffff0000u)};
            f32x2_b gx[4] = {(f32x2_b){0.f, 0.f}, (f32x2_b){0.f, 0.f}, (f32x2_b){0.f, 0.f}, (f32x2_b){0.f, 0.f}};
            const float* sg = s_g[k & 1];
            // positions of the padded domain that reflect onto q: q itself, and the ring row / column for rows 1, H-2 / columns 1, W-2
            auto eval = [&](int py, int px, bool interior) {
                // e[t] = g0(pp - t + 1): the 3 x 3 neighbourhood of pp read back to front; LDS origin = (y0 - 2, x0 - 2)
                const float* c0 = sg + (py - y0 + 2) * IB_GP + (px - x0 + 2);
                float e[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) e[t] = c0[(1 - t / 3) * IB_GP + (1 - t % 3)];
                if (interior) accb += e[4];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const f32x2_b et = {e[t], e[t]};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        gx[i] = __builtin_elementwise_fma((f32x2_b){wr[2 * i][t], wr[2 * i + 1][t]}, et, gx[i]);
                        acc[i][t] = __builtin_elementwise_fma(xv[i], et, acc[i][t]);
                    }
                }
            };
            eval(qy, qx, true);
            const int ry = qy == 1 ? -1 : (qy == H - 2 ? H : -2), rx = qx == 1 ? -1 : (qx == W - 2 ? W : -2);
            // (H, W >= 4: a row / column is the target of at most one ring row / column)
            if (ry != -2) eval(ry, qx, false);
            if (rx != -2) eval(qy, rx, false);
            if (ry != -2 && rx != -2) eval(ry, rx, false);
            // ReLU mask of the layer's input (the previous layer's output x > 0), one rounding, into the interior of the halo-1 gradient
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float lo = xv[i].x > 0.f ? gx[i].x : 0.f, hi = xv[i].y > 0.f ? gx[i].y : 0.f;
                o[i] = pack_bf16x2(lo, hi);
            }
            *reinterpret_cast<uint4*>(tgx.base + tgx.gidx(in_, cb, qy + 1, qx + 1) * 16) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    };
    for (int k = 0; k < tw.count; k += 3) {
        body(k, xq0);
        body(k + 1, xq1);
        body(k + 2, xq2);
    }
    // ---- block partial: dW[8 cb + 2 i + half][t] summed over the channel block's four waves, db over channel block 0's
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float v = hlf ? acc[i][t].y : acc[i][t].x;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                if (lane == 0) s_red[wave][(2 * i + hlf) * 9 + t] = v;
            }
    {
        float v = accb;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) s_red[wave][72] = v;
    }
    __syncthreads();
    float* dst = partial + (long long)blockIdx.x * IB_PER;
    if (tid < 144) {
        const int c2 = tid / 72, e = tid % 72;
        dst[tid] = (s_red[4 * c2][e] + s_red[4 * c2 + 1][e]) + (s_red[4 * c2 + 2][e] + s_red[4 * c2 + 3][e]);
    } else if (tid == 144) {
        dst[144] = (s_red[0][72] + s_red[1][72]) + (s_red[2][72] + s_red[3][72]);
    }
}

// ---------------------------------------------------------------- forward of the same layer (round 6)
// image_out_fwd_tiled_kernel kept the 144 weights in vector registers (3 waves per SIMD) and ran load -> LDS -> barrier -> arithmetic -> store
// once per 16 x 16-tile block: 40 us for 75 MB at B = 32 256 x 256, latency bound.  Here: the recipe of the backward kernel above -- a thread
// owns (pixel, channel block), the channel block's 72 weights sit in scalar registers, persistent blocks walk 8 x 32-pixel tiles with the next
// tile's reflect-padded 10 x 34 x 2 granules requested one tile ahead (double-buffered LDS, one barrier for the tile + one for the exchange
// of the two channel blocks' partial sums).  y = (bias + sum over channel block 0) + (sum over channel block 1), each sum an FMA chain in
// (tap, channel) order -- the tiled kernel's single chain regrouped (fp32, ~1e-7).
constexpr int IF_GW = IB_TW + 2, IF_GH = IB_TH + 2;     // 34 x 10 granules per channel block
constexpr int IF_GP = 36;                               // row pitch in granules
constexpr int IF_N = IF_GW * IF_GH;                     // 340

__global__ __launch_bounds__(512, 2) void image_out_fwd16_kernel(TV tx, const float* __restrict__ w, const float* __restrict__ bias,
                                                                 float* __restrict__ img, int relu, int tiles_x, int tiles_y, int total) {
    __shared__ __attribute__((aligned(16))) uint4 s_x[2][2][IF_GH * IF_GP];
    __shared__ float s_p[2][256];
    const int tid = threadIdx.x;
    const int cb = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int pix = tid & 255, ty = pix >> 5, txx = pix & 31;
    const int H = tx.h, W = tx.w;
    const TileWalk tw = xcd_walk(total, gridDim.x, blockIdx.x);
    const float* wq = w + cb * 72;
    float wr[8][9];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[c][t] = wq[c * 9 + t];
    const float b0 = bias != nullptr ? bias[0] : 0.f;
    auto tile_origin = [&](int tile, int& in_, int& y0, int& x0) {
        const int tpi = tiles_x * tiles_y;
        in_ = tile / tpi;
        const int r = tile - in_ * tpi;
        y0 = (r / tiles_x) * IB_TH;
        x0 = (r % tiles_x) * IB_TW;
    };
    // staged granules of this thread: e = pix and pix + 256 (< 340) of its channel block
    const int e1 = min(pix + 256, IF_N - 1);
    const int py0 = pix / IF_GW, px0 = pix % IF_GW, py1 = e1 / IF_GW, px1 = e1 % IF_GW;
    auto load_x = [&](int tile, int py, int px) {
        int in_, y0, x0;
        tile_origin(tile, in_, y0, x0);
        const int y = min(max(reflect_idx(y0 - 1 + py, H), 0), H - 1), x = min(max(reflect_idx(x0 - 1 + px, W), 0), W - 1);
        return *reinterpret_cast<const uint4*>(tx.base + tx.gidx(in_, cb, y, x) * 16);
    };
    uint4 xa = make_uint4(0, 0, 0, 0), xb = xa;
    if (tw.count > 0) { xa = load_x(tw.first, py0, px0); xb = load_x(tw.first, py1, px1); }
    for (int k = 0; k < tw.count; ++k) {
        const int tile = tw.first + k * tw.stride;
        uint4* sx = s_x[k & 1][cb];
        sx[py0 * IF_GP + px0] = xa;
        if (pix + 256 < IF_N) sx[py1 * IF_GP + px1] = xb;
        __syncthreads();
        if (k + 1 < tw.count) { xa = load_x(tile + tw.stride, py0, px0); xb = load_x(tile + tw.stride, py1, px1); }
        float r = cb == 0 ? b0 : 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const uint4 q = sx[(ty + t / 3) * IF_GP + txx + t % 3];
            const uint32_t qw[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                r = fmaf(__uint_as_float(qw[i] << 16), wr[2 * i][t], r);
                r = fmaf(__uint_as_float(qw[i] & 0xffff0000u), wr[2 * i + 1][t], r);
            }
        }
        if (cb == 1) s_p[k & 1][pix] = r;
        __syncthreads();
        if (cb == 0) {
            int in_, y0, x0;
            tile_origin(tile, in_, y0, x0);
            const int y = y0 + ty, x = x0 + txx;
            if (y < H && x < W) {
                float v = r + s_p[k & 1][pix];
                if (relu) v = fmaxf(v, 0.f);
                img[((long long)in_ * H + y) * W + x] = v;
            }
        }
    }
}

int image_out_wgrad_reduce3_launch(const float* ws, float* dw, float* db, int cin, int G, int n_cg, int accumulate, hipStream_t st);   // conv_image.hip

}  // namespace mmif

using namespace mmif;

// (conv_image.hip's mmif_conv2d_image_out_fwd takes this kernel for bf16, 16 channels, 3x3)
int image_out_fwd16_launch(const TV& tx, const float* w, const float* bias, float* img, int relu, hipStream_t st) {
    const int tiles_x = cdiv(tx.w, IB_TW), tiles_y = cdiv(tx.h, IB_TH);
    const long long total = (long long)tiles_x * tiles_y * tx.n;
    if (total >= (1ll << 31)) { set_error("image_out_fwd: too many tiles"); return MMIF_EINVAL; }
    int G = 1024;      // four persistent blocks per CU (39 registers, 25 KB of LDS each)
    if (total < G) G = (int)total;
    if (G >= 8) G = G / 8 * 8;
    hipLaunchKernelGGL(image_out_fwd16_kernel, dim3(G), dim3(512), 0, st, tx, w, bias, img, relu, tiles_x, tiles_y, (int)total);
    return check_launch("image_out_fwd");
}

extern "C" int32_t mmif_conv2d_image_out_bwd_supported(int32_t dtype, int32_t cin, int32_t ksize, int32_t h, int32_t w) {
    return dtype == MMIF_BF16 && cin == 16 && ksize == 3 && h >= 4 && w >= 4 ? 1 : 0;
}

extern "C" int mmif_conv2d_image_out_bwd(const mmif_tensor* x, const float* gimg, const float* y_img, const float* w, const mmif_tensor* gx,
                                         float* dw, float* db, int32_t cin, int32_t ksize, int32_t accumulate, void* workspace,
                                         size_t workspace_bytes, void* stream) {
    if (int rc = validate_tensor(x, "x")) return rc;
    if (int rc = validate_tensor(gx, "gx")) return rc;
    MMIF_REQUIRE(mmif_conv2d_image_out_bwd_supported(x->dtype, cin, ksize, x->h, x->w), "image_out_bwd: bf16, 16 input channels, 3x3, h, w >= 4 only");
    MMIF_REQUIRE(x->halo == 0 && x->cb == 2 && gx->halo == 1 && gx->cb == 2 && gx->dtype == x->dtype && gx->n == x->n && gx->h == x->h && gx->w == x->w,
                 "image_out_bwd: x must be a 16-channel activation (halo 0), gx its halo-1 gradient");
    if (workspace_bytes < mmif_conv2d_image_wgrad_workspace(cin, ksize)) {
        set_error("image_out_bwd: workspace too small");
        return MMIF_EWORKSPACE;
    }
    TV tx = make_tv(x), tgx = make_tv(gx);
    const int tiles_x = cdiv(tx.w, IB_TW), tiles_y = cdiv(tx.h, IB_TH);
    const long long total = (long long)tiles_x * tiles_y * tx.n;
    MMIF_REQUIRE(total < (1ll << 31), "image_out_bwd: too many tiles");
    int G = 512;      // two persistent blocks per CU (512 = the workspace's block count, conv_image.hip IMG_G)
    if (total < G) G = (int)total;
    if (G >= 8) G = G / 8 * 8;
    hipStream_t st = (hipStream_t)stream;
    float* ws = defer_ws((float*)workspace, (size_t)G * IB_PER * sizeof(float));
    hipLaunchKernelGGL(image_out_bwd16_kernel, dim3(G), dim3(512), 0, st, tx, tgx, gimg, y_img, w, ws, tiles_x, tiles_y, (int)total);
    if (int rc = check_launch("image_out_bwd")) return rc;
    {
        RedJob J;
        J.partial = ws; J.dw = dw; J.db = db; J.type = RED_IMAGE_OUT; J.sl = RED_SLICES; J.G = G; J.accumulate = accumulate;
        J.p0 = cin; J.p1 = ksize; J.p2 = 1; J.p3 = 0; J.nvb = cdiv(cin * 9 + 1, 64);
        if (defer_push(J)) return MMIF_OK;
    }
    return image_out_wgrad_reduce3_launch(ws, dw, db, cin, G, 1, accumulate, st);
}
