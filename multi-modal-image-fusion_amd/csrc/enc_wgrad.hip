// Weight gradients of the whole DenseBlock encoder -- ConvLayer(1 -> 16) + DenseBlock(16, 16): dW, db of all FOUR layers
// (reference core/model.py:73-80, core/block.py:137-151; backward of train.py:70) -- in ONE pass over the activations.
//
// Layer by layer (wgrad_mfma_kernel x 3 + image_in_wgrad_kernel) the encoder's weight gradients read 160 channel planes per branch:
// the 48 -> 16 layer reads x0 x1 x2 + g3, the 32 -> 16 layer x0 x1 + g2, the 16 -> 16 layer x0 + g1, the first layer the image + g0
// -- x0 three times, x1 twice.  Here a block stages ONE 16 x 16-pixel tile of the concatenated activations (x0 | x1 | x2, 18 x 18
// with the reflect halo), of the four pre-activation gradients (g0 | g1 | g2 | g3) and of the image into LDS and forms every
// product from it: 112 planes + the image, and each shifted activation fragment (transposing LDS read, ds_read_b64_tr_b16) feeds
// up to three MFMAs (x0 serves layers 1, 2, 3).
//   dW_L[o][c][u][v] = sum_pixels gL[o](y, x) * xin[c](R(y+u-1), R(x+v-1))      K = pixels (32 per k-step = two tile rows)
//   wave u = 0, 1, 2: tap row u -> 3 taps x (3 + 2 + 1) (layer, 16-input-channel block) products = 18 accumulator tiles (bf16 MFMA)
//   wave 3: the first layer against the fp32 image on the EXACT fp32 matrix path (v_mfma_f32_16x16x4_f32: the image is not rounded;
//           N = 9 taps + a column of ones = db0) and db1..db3 (bf16 MFMA against ones)
// No K split between waves -> no cross-wave reduction; per-block partials are reduced in a fixed order by a second kernel
// (deterministic, no atomics).  ~250 B staged per pixel => HBM-bound; 2 blocks per CU (65 KB LDS each).
#include "enc_wgrad.hpp"
#include "reduce_defer.hpp"

namespace mmif {

typedef __attribute__((ext_vector_type(8))) __bf16 ew_bf16x8;
typedef __attribute__((ext_vector_type(4))) short ew_s16x4;
typedef __attribute__((ext_vector_type(8))) short ew_s16x8;
typedef __attribute__((ext_vector_type(4))) float ew_f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned ew_u32x4;   // granule as a NATIVE vector (a uint4 struct copy from global memory becomes
                                                                 // a memcpy into a private array that is never promoted to registers)
#define EW_LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

constexpr int EW_T = 16, EW_TP = 18;
constexpr int EW_XPL = 324;     // granules per x plane (18 x 18); 5184 B = 64 mod 256 (as wgrad_mfma_kernel's tiles)
constexpr int EW_GPL = 260;     // granules per g plane (256 used)
constexpr int EW_NX = 6, EW_NG = 8;
constexpr int EW_TILE_BYTES = (EW_NX * EW_XPL + EW_NG * EW_GPL) * 16 + EW_TP * EW_TP * 4;
constexpr int EW_SM_BYTES = EW_TILE_BYTES > EW_PER * 4 ? EW_TILE_BYTES : EW_PER * 4;
#ifndef EW_ABL
#define EW_ABL 0   // timing ablations (diagnostic builds, -DEW_ABL=n; results WRONG when non-zero): 1 no bf16 product waves, 2 no first-layer /
#endif             // bias wave, 4 no global prefetch after the first tile (tools/bench_enc.py)

__global__ __launch_bounds__(256, 2) void enc_wgrad_kernel(const float* __restrict__ img, TV tx, TV tg, float* __restrict__ partial,
                                                            int tiles_x, int tpi, int total, int G) {
    __shared__ __attribute__((aligned(16))) char smem[EW_SM_BYTES];
    ew_u32x4* s_x = reinterpret_cast<ew_u32x4*>(smem);
    ew_u32x4* s_g = s_x + EW_NX * EW_XPL;
    float* s_img = reinterpret_cast<float*>(s_g + EW_NG * EW_GPL);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sl = lane & 15, g = lane >> 4;
    // XCD-aware tile order: block b runs on XCD b % 8 and walks the tiles of one contiguous band
    int gi = blockIdx.x;
    const int H = tx.h, W = tx.w;

    // (wave 3's accumulators -- the first layer's four chains -- live in tiles of acc3 the wave does not otherwise use: the roles never meet, and a
    //  union keeps the kernel inside the 256-register budget of two blocks per CU)
    ew_f32x4 acc3[3][3], acc2[3][2], acc1[3], accbw = {0.f, 0.f, 0.f, 0.f};
#define acc0 acc3[0][0]
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        acc1[v] = (ew_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 3; ++b) acc3[v][b] = (ew_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 2; ++b) acc2[v][b] = (ew_f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const ew_bf16x8 ones = __builtin_bit_cast(ew_bf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));

    // transposing reads: in-group lane sl supplies pixel (sl >> 2), 4-channel chunk (sl & 3) of a 16-channel block
    const int tr_row = sl >> 2, tr_c = sl & 3;
    const int lane_plane = tr_c >> 1, lane_byte = (tr_c & 1) * 8;
    auto tr_frag = [&](const char* base) {
        const ew_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(EW_LDS_PTR(ew_s16x4, base));
        const ew_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(EW_LDS_PTR(ew_s16x4, base + 4 * 16));
        const ew_s16x8 c = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(ew_bf16x8, c);
    };

    constexpr int NXR = (EW_NX * EW_XPL + 255) / 256;   // 8 x granules per thread per tile
    ew_u32x4 rx[NXR], rg[EW_NG];
    float ri[2];
    // all prefetch loads are unconditional (clamped addresses, zero by select): nothing serialises them.  Address arithmetic in 32 bits from
    // per-thread descriptors packed once (channel block << 16 | tile row << 8 | tile column): recomputing e / 324, e % 18 and three
    // 64-bit products per load made the issue of a tile's 18 loads a 3.4 k-tick phase of its 14.8 k-tick period (`s_memtime` trace)
    unsigned xpk[NXR], ipk[2];
#pragma unroll
    for (int i = 0; i < NXR; ++i) {
        const int e = min(tid + 256 * i, EW_NX * EW_XPL - 1);
        const int cb = e / EW_XPL, p = e - cb * EW_XPL;
        xpk[i] = (unsigned)(cb << 16 | (p / EW_TP) << 8 | (p % EW_TP));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = min(tid + 256 * i, EW_TP * EW_TP - 1);
        ipk[i] = (unsigned)((e / EW_TP) << 8 | (e % EW_TP));
    }
    const unsigned xplane = (unsigned)tx.plane, gplane = (unsigned)tg.plane;   // (the host checks that an image stays below 2^31 granules)
    const unsigned xcb0 = (unsigned)tx.cb_off * xplane, gcb0 = (unsigned)tg.cb_off * gplane;
    auto prefetch = [&](int tile) {
        const int in_ = tile / tpi, tt = tile - in_ * tpi;
        const int y0 = (tt / tiles_x) * EW_T, x0 = (tt % tiles_x) * EW_T;
        const char* xb = tx.base + (long long)in_ * tx.img * 16;
        const char* gb = tg.base + (long long)in_ * tg.img * 16;
#pragma unroll
        for (int i = 0; i < NXR; ++i) {
            const int y = min(max(reflect_idx(y0 + (int)((xpk[i] >> 8) & 255u) - 1, H), 0), H - 1);
            const int x = min(max(reflect_idx(x0 + (int)(xpk[i] & 255u) - 1, W), 0), W - 1);
            const unsigned off = xcb0 + (xpk[i] >> 16) * xplane + (unsigned)(y + tx.halo) * (unsigned)tx.ws + (unsigned)(x + tx.halo);
            rx[i] = *reinterpret_cast<const ew_u32x4*>(xb + (unsigned long long)off * 16u);
        }
        const int gy = y0 + tid / EW_T, gx = x0 + tid % EW_T;
        const bool inside = gy < H && gx < W;
        const unsigned goff = gcb0 + (unsigned)(min(gy, H - 1) + tg.halo) * (unsigned)tg.ws + (unsigned)(min(gx, W - 1) + tg.halo);
#pragma unroll
        for (int i = 0; i < EW_NG; ++i) {
            const ew_u32x4 v = *reinterpret_cast<const ew_u32x4*>(gb + (unsigned long long)(goff + i * gplane) * 16u);
            rg[i] = inside ? v : (ew_u32x4){0u, 0u, 0u, 0u};
        }
        const float* im = img + (long long)in_ * H * W;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int y = min(max(reflect_idx(y0 + (int)(ipk[i] >> 8) - 1, H), 0), H - 1);
            const int x = min(max(reflect_idx(x0 + (int)(ipk[i] & 255u) - 1, W), 0), W - 1);
            ri[i] = im[(unsigned)(y * W + x)];
        }
    };
    const TileWalk tw = xcd_walk(total, G, gi);
    if (tw.count > 0) prefetch(tw.first);
    for (int it = 0, tile = tw.first; it < tw.count; ++it, tile += tw.stride) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NXR; ++i) {
            const int e = tid + 256 * i;
            if (e < EW_NX * EW_XPL) s_x[e] = rx[i];
        }
#pragma unroll
        for (int i = 0; i < EW_NG; ++i) s_g[i * EW_GPL + tid] = rg[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + 256 * i;
            if (e < EW_TP * EW_TP) s_img[e] = ri[i];
        }
        __syncthreads();
        if (it + 1 < tw.count && !(EW_ABL & 4)) prefetch(tile + tw.stride);   // in flight during the MFMAs below
        if (wave < 3 && !(EW_ABL & 1)) {
            const int u = wave;
            // 8 k-steps x 3 tap columns = 24 groups of (3 activation fragments -> 6 MFMAs); the fragments of group i + 1 (and the three
            // gradient fragments of the next k-step) are fetched under the MFMAs of group i: at two waves per SIMD nothing else hides
            // the ~130-cycle latency of the transposing LDS reads (compute-only time of this kernel 115 -> see DESIGN.md)
            auto ldA = [&](int s, ew_bf16x8 (&a)[3]) {
                const int row = 2 * s + (g >> 1), col0 = 8 * (g & 1);
#pragma unroll
                for (int L = 1; L <= 3; ++L)
                    a[L - 1] = tr_frag(reinterpret_cast<const char*>(s_g) + ((2 * L + lane_plane) * EW_GPL + row * EW_T + col0 + tr_row) * 16 + lane_byte);
            };
            auto ldB = [&](int s, int v, ew_bf16x8 (&bx)[3]) {
                const int row = 2 * s + (g >> 1), col0 = 8 * (g & 1);
#pragma unroll
                for (int b = 0; b < 3; ++b)
                    bx[b] = tr_frag(reinterpret_cast<const char*>(s_x) + ((2 * b + lane_plane) * EW_XPL + (row + u) * EW_TP + col0 + v + tr_row) * 16 + lane_byte);
            };
            ew_bf16x8 a[3], an[3], bx[2][3];   // (the gradient fragments of the next k-step arrive during its last group: one copy, after use)
            ldA(0, a);
            ldB(0, 0, bx[0]);
#pragma unroll 6   // two k-steps per trip: every fragment buffer index below is a compile-time constant, nothing in flight is ever copied
            for (int i = 0; i < 24; ++i) {
                const int s = i / 3, v = i % 3;
                if (i + 1 < 24) {
                    const int s1 = (i + 1) / 3, v1 = (i + 1) % 3;
                    if (v1 == 0) ldA(s1, an);
                    ldB(s1, v1, bx[(i + 1) & 1]);
                }
#pragma unroll
                for (int b = 0; b < 3; ++b) acc3[v][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], bx[i & 1][b], acc3[v][b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < 2; ++b) acc2[v][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], bx[i & 1][b], acc2[v][b], 0, 0, 0);
                acc1[v] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bx[i & 1][0], acc1[v], 0, 0, 0);
                // db of layer u + 1: a ones product of the gradient fragment this wave holds anyway (it used to cost wave 3, the
                // long pole of the tile, three more transposing fragment reads and MFMAs per k-step)
                if (v == 0) accbw = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u == 0 ? a[0] : (u == 1 ? a[1] : a[2]), ones, accbw, 0, 0, 0);
                if (i + 1 < 24) {   // next group's DS reads first,
                    if ((i + 1) % 3 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                                              // then this group's MFMAs
                if (v == 2) { a[0] = an[0]; a[1] = an[1]; a[2] = an[2]; }
                (void)s;
            }
        } else if (wave == 3 && !(EW_ABL & 2)) {
            // first layer: D[o][n] += sum_p g0[o](p) * B[p][n],  B[p][n] = image(p + tap n) for n < 9, 1 for n = 9 (-> db0), exact fp32
            const int un = sl / 3, vn = sl - 3 * un;
            // (four accumulator chains -- a dependent v_mfma_f32_16x16x4_f32 issues every 40 cycles, independent ones every 32 -- in
            //  tiles of acc3 / acc2 this wave does not otherwise use; summed when the partial is written)
#pragma unroll 2
            for (int s = 0; s < 8; ++s) {
                float af[8], bf[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int p = 4 * q + g, prow = 2 * s + (p >> 4), pcol = p & 15;
                    const unsigned short gb = *reinterpret_cast<const unsigned short*>(reinterpret_cast<const char*>(s_g) + ((sl >> 3) * EW_GPL + prow * EW_T + pcol) * 16 + (sl & 7) * 2);
                    af[q] = __uint_as_float((unsigned)gb << 16);
                    const float iv = s_img[(prow + min(un, 2)) * EW_TP + pcol + vn];
                    bf[q] = sl < 9 ? iv : (sl == 9 ? 1.f : 0.f);
                }
#pragma unroll
                for (int q = 0; q < 8; q += 4) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q], bf[q], acc0, 0, 0, 0);
                    acc3[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q + 1], bf[q + 1], acc3[0][1], 0, 0, 0);
                    acc3[0][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q + 2], bf[q + 2], acc3[0][2], 0, 0, 0);
                    acc3[2][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q + 3], bf[q + 3], acc3[2][0], 0, 0, 0);
                }
            }
        }
    }
    // ---- block partial: accumulators -> LDS (natural [o][c][u][v] order) -> one coalesced copy
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    if (wave < 3) {
        const int u = wave;
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const int t = 3 * u + v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int oc = 4 * g + r;
#pragma unroll
                for (int b = 0; b < 3; ++b) red[EW_OFF3 + (oc * 48 + 16 * b + sl) * 9 + t] = acc3[v][b][r];
#pragma unroll
                for (int b = 0; b < 2; ++b) red[EW_OFF2 + (oc * 32 + 16 * b + sl) * 9 + t] = acc2[v][b][r];
                red[EW_OFF1 + (oc * 16 + sl) * 9 + t] = acc1[v][r];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (sl == 0) red[EW_OFFB + u * 16 + 4 * g + r] = accbw[r];   // every column of a ones product is the sum
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oc = 4 * g + r;
            red[EW_OFF0 + oc * 16 + sl] = (acc0[r] + acc3[0][1][r]) + (acc3[0][2][r] + acc3[2][0][r]);

        }
    }
    __syncthreads();
    float* dst = partial + (long long)gi * EW_PER;
    for (int e = tid; e < EW_PER; e += 256) dst[e] = red[e];
}
#undef acc0

// ------------------------------------------------------------------ the same scheme for a single 3x3 layer ("tap-row" wgrad)
// dW, db of ONE ConvLayer with 16 NXB input and 16 NGB output channels (decoder tails: 64 -> 32, 32 -> 16; the DenseBlock convs when the
// fused encoder kernel is off): wgrad_mfma_kernel gives every 16-input-channel group its own block, so the gradient tile is re-read
// NXB times (64 -> 32: 192 planes staged for 96 of data).  Here ONE block stages the whole x tile (all input channels, 18 x 18) and
// the whole g tile once; wave u = 0, 1, 2 owns tap row u for every (input block, output block) pair -- 3 NXB NGB accumulator tiles,
// each transposed x fragment feeding NGB MFMAs -- and wave 3 the bias sums.  No K split, per-block partials in the natural
// [o][c][u][v] order, fixed-order reduction.
template <int NXB, int NGB>
__global__ __launch_bounds__(256, 2) void taprow_wgrad_kernel(TV tx, TV tg, float* __restrict__ partial, int tiles_x, int tpi, int total, int G) {
    constexpr int CIN = 16 * NXB, COUT = 16 * NGB;
    constexpr int PER = COUT * CIN * 9 + COUT;
    constexpr int TILE_BYTES = (2 * NXB * EW_XPL + 2 * NGB * EW_GPL) * 16;
    constexpr int SM_BYTES = TILE_BYTES > PER * 4 ? TILE_BYTES : PER * 4;
    __shared__ __attribute__((aligned(16))) char smem[SM_BYTES];
    ew_u32x4* s_x = reinterpret_cast<ew_u32x4*>(smem);
    ew_u32x4* s_g = s_x + 2 * NXB * EW_XPL;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sl = lane & 15, g = lane >> 4;
    const int gi = blockIdx.x;
    const int H = tx.h, W = tx.w;
    ew_f32x4 acc[3][NXB][NGB], accb[NGB];
#pragma unroll
    for (int m = 0; m < NGB; ++m) {
        accb[m] = (ew_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int b = 0; b < NXB; ++b) acc[v][b][m] = (ew_f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const ew_bf16x8 ones = __builtin_bit_cast(ew_bf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));
    const int tr_row = sl >> 2, tr_c = sl & 3;
    const int lane_plane = tr_c >> 1, lane_byte = (tr_c & 1) * 8;
    auto tr_frag = [&](const char* base) {
        const ew_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(EW_LDS_PTR(ew_s16x4, base));
        const ew_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(EW_LDS_PTR(ew_s16x4, base + 4 * 16));
        const ew_s16x8 c = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(ew_bf16x8, c);
    };
    constexpr int NXR = (2 * NXB * EW_XPL + 255) / 256;
    ew_u32x4 rx[NXR], rg[2 * NGB];
    auto prefetch = [&](int tile) {
        const int in_ = tile / tpi, tt = tile - in_ * tpi;
        const int y0 = (tt / tiles_x) * EW_T, x0 = (tt % tiles_x) * EW_T;
#pragma unroll
        for (int i = 0; i < NXR; ++i) {
            const int e = min(tid + 256 * i, 2 * NXB * EW_XPL - 1);
            const int cb = e / EW_XPL, p = e - cb * EW_XPL;
            const int y = min(max(reflect_idx(y0 + p / EW_TP - 1, H), 0), H - 1);
            const int x = min(max(reflect_idx(x0 + p % EW_TP - 1, W), 0), W - 1);
            rx[i] = *reinterpret_cast<const ew_u32x4*>(tx.base + tx.gidx(in_, cb, y, x) * 16);
        }
        const int gy = y0 + tid / EW_T, gx = x0 + tid % EW_T;
        const bool inside = gy < H && gx < W;
        const int cy = min(gy, H - 1) + tg.halo, cx = min(gx, W - 1) + tg.halo;
#pragma unroll
        for (int i = 0; i < 2 * NGB; ++i) {
            const ew_u32x4 v = *reinterpret_cast<const ew_u32x4*>(tg.base + tg.gidx(in_, i, cy, cx) * 16);
            rg[i] = inside ? v : (ew_u32x4){0u, 0u, 0u, 0u};
        }
    };
    const TileWalk tw = xcd_walk(total, G, gi);
    if (tw.count > 0) prefetch(tw.first);
    for (int it = 0, tile = tw.first; it < tw.count; ++it, tile += tw.stride) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NXR; ++i) {
            const int e = tid + 256 * i;
            if (e < 2 * NXB * EW_XPL) s_x[e] = rx[i];
        }
#pragma unroll
        for (int i = 0; i < 2 * NGB; ++i) s_g[i * EW_GPL + tid] = rg[i];
        __syncthreads();
        if (it + 1 < tw.count) prefetch(tile + tw.stride);
        if (wave < 3) {
            const int u = wave;
#pragma unroll 2
            for (int s = 0; s < 8; ++s) {
                const int row = 2 * s + (g >> 1), col0 = 8 * (g & 1);
                ew_bf16x8 a[NGB];
#pragma unroll
                for (int m = 0; m < NGB; ++m)
                    a[m] = tr_frag(reinterpret_cast<const char*>(s_g) + ((2 * m + lane_plane) * EW_GPL + row * EW_T + col0 + tr_row) * 16 + lane_byte);
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    ew_bf16x8 bx[NXB];
#pragma unroll
                    for (int b = 0; b < NXB; ++b)
                        bx[b] = tr_frag(reinterpret_cast<const char*>(s_x) + ((2 * b + lane_plane) * EW_XPL + (row + u) * EW_TP + col0 + v + tr_row) * 16 + lane_byte);
#pragma unroll
                    for (int b = 0; b < NXB; ++b)
#pragma unroll
                        for (int m = 0; m < NGB; ++m) acc[v][b][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], bx[b], acc[v][b][m], 0, 0, 0);
                }
            }
        } else {
#pragma unroll 2
            for (int s = 0; s < 8; ++s) {
                const int row = 2 * s + (g >> 1), col0 = 8 * (g & 1);
#pragma unroll
                for (int m = 0; m < NGB; ++m) {
                    const ew_bf16x8 a = tr_frag(reinterpret_cast<const char*>(s_g) + ((2 * m + lane_plane) * EW_GPL + row * EW_T + col0 + tr_row) * 16 + lane_byte);
                    accb[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, ones, accb[m], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    if (wave < 3) {
        const int u = wave;
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int b = 0; b < NXB; ++b)
#pragma unroll
                for (int m = 0; m < NGB; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[((16 * m + 4 * g + r) * CIN + 16 * b + sl) * 9 + 3 * u + v] = acc[v][b][m][r];
    } else if (sl == 0) {
#pragma unroll
        for (int m = 0; m < NGB; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[COUT * CIN * 9 + 16 * m + 4 * g + r] = accb[m][r];
    }
    __syncthreads();
    float* dst = partial + (long long)gi * PER;
    for (int e = tid; e < PER; e += 256) dst[e] = red[e];
}

// out[i] = sum_g partial[g][i] (fixed order); the first n_w entries are dW in its natural layout, the rest db
__global__ __launch_bounds__(64 * RED_SLICES) void taprow_wgrad_reduce(const float* __restrict__ partial, float* __restrict__ dw, float* __restrict__ db, int n_w,
                                                           int per, int G, int accumulate) {
    __shared__ float red[RED_SLICES][64];
    const int idx = blockIdx.x * 64 + (threadIdx.x & 63);
    const float t = partial_sum(partial, idx, per, G, idx < per, red);
    if ((threadIdx.x >> 6) != 0 || idx >= per) return;
    float* p = idx < n_w ? dw + idx : (db != nullptr ? db + (idx - n_w) : nullptr);
    if (p != nullptr) *p = accumulate ? *p + t : t;
}

bool wgrad_taprow_supported(int ks, int cin, int cout) {
    if (ks != 3 || cin % 16 || cout % 16) return false;
    const int nxb = cin / 16, ngb = cout / 16;
    return (ngb == 1 && nxb >= 1 && nxb <= 3) || (ngb == 2 && (nxb == 2 || nxb == 4));
}
size_t wgrad_taprow_workspace(int cin, int cout) { return (size_t)EW_MAXG * ((size_t)cout * cin * 9 + cout) * sizeof(float); }

// fixed-order reduction of G natural-layout partials ([cout][cin][3][3] then [cout]) -- shared with the fused backward kernel of conv_mfma.hip
int taprow_reduce_launch(const float* ws, float* dw, float* db, int cin, int cout, int G, int accumulate, hipStream_t st) {
    const int n_w = cout * cin * 9, per = n_w + cout;
    {
        RedJob J;
        J.partial = ws; J.dw = dw; J.db = db; J.type = RED_TAPROW; J.sl = RED_SLICES; J.G = G; J.accumulate = accumulate;
        J.p0 = n_w; J.p1 = per; J.p2 = 0; J.p3 = 0; J.nvb = cdiv(per, 64);
        if (defer_push(J)) return MMIF_OK;
    }
    hipLaunchKernelGGL(taprow_wgrad_reduce, dim3(cdiv(per, 64)), dim3(64 * RED_SLICES), 0, st, ws, dw, db, n_w, per, G, accumulate);
    return check_launch("wgrad_taprow_reduce");
}

int wgrad_taprow(const TV& tx, const TV& tg, float* dw, float* db, int cin, int cout, int accumulate, float* ws, hipStream_t st) {
    const int tiles_x = cdiv(tx.w, EW_T), tiles_y = cdiv(tx.h, EW_T);
    const int tpi = tiles_x * tiles_y, total = tpi * tx.n;
    const int G = total < EW_MAXG ? total : EW_MAXG;
    const int nxb = cin / 16, ngb = cout / 16;
#define GO(X_, G_) hipLaunchKernelGGL((taprow_wgrad_kernel<X_, G_>), dim3(G), dim3(256), 0, st, tx, tg, ws, tiles_x, tpi, total, G)
    if (ngb == 1) { switch (nxb) { case 1: GO(1, 1); break; case 2: GO(2, 1); break; default: GO(3, 1); break; } }
    else { if (nxb == 2) GO(2, 2); else GO(4, 2); }
#undef GO
    if (int rc = check_launch("wgrad_taprow")) return rc;
    const int n_w = cout * cin * 9, per = n_w + cout;
    hipLaunchKernelGGL(taprow_wgrad_reduce, dim3(cdiv(per, 64)), dim3(64 * RED_SLICES), 0, st, (const float*)ws, dw, db, n_w, per, G, accumulate);
    return check_launch("wgrad_taprow_reduce");
}


// 64 outputs x 4 slices of the G partials per block; fixed summation order
__global__ __launch_bounds__(64 * RED_SLICES) void enc_wgrad_reduce(const float* __restrict__ partial, EwDst D, int G, int accumulate) {
    __shared__ float red[RED_SLICES][64];
    const int idx = blockIdx.x * 64 + (threadIdx.x & 63);
    const float t = partial_sum(partial, idx, EW_PER, G, idx < EW_PER, red);
    if ((threadIdx.x >> 6) != 0 || idx >= EW_PER) return;
    float* p = nullptr;
    if (idx < EW_OFF2) p = D.dw[2] + idx;
    else if (idx < EW_OFF1) p = D.dw[1] + (idx - EW_OFF2);
    else if (idx < EW_OFF0) p = D.dw[0] + (idx - EW_OFF1);
    else if (idx < EW_OFFB) {
        const int oc = (idx - EW_OFF0) >> 4, n = (idx - EW_OFF0) & 15;
        if (n < 9) p = D.dw0 + oc * 9 + n;
        else if (n == 9 && D.db0 != nullptr) p = D.db0 + oc;
    } else {
        const int L = (idx - EW_OFFB) >> 4, oc = (idx - EW_OFFB) & 15;
        if (D.db[L] != nullptr) p = D.db[L] + oc;
    }
    if (p != nullptr) *p = accumulate ? *p + t : t;
}

// both branches of a fused encoder backward in ONE launch: blockIdx.y = branch when the destinations differ; a SHARED encoder (the second
// branch accumulates onto the first's gradients) runs its two sums one after the other in the same blocks -- the order of the two launches
// this replaces, bit for bit
struct EwPair { const float* partial[2]; EwDst D[2]; int accumulate[2]; int serial; };
__global__ __launch_bounds__(64 * RED_SLICES) void enc_wgrad_reduce_pair(EwPair P, int G) {
    __shared__ float red[RED_SLICES][64];
    const int idx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int b0 = P.serial ? 0 : blockIdx.y, b1 = P.serial ? 2 : b0 + 1;
    for (int b = b0; b < b1; ++b) {
        const float t = partial_sum(P.partial[b], idx, EW_PER, G, idx < EW_PER, red);
        __syncthreads();      // (red is reused by the second sum)
        if ((threadIdx.x >> 6) != 0 || idx >= EW_PER) continue;
        const EwDst& D = P.D[b];
        float* p = nullptr;
        if (idx < EW_OFF2) p = D.dw[2] + idx;
        else if (idx < EW_OFF1) p = D.dw[1] + (idx - EW_OFF2);
        else if (idx < EW_OFF0) p = D.dw[0] + (idx - EW_OFF1);
        else if (idx < EW_OFFB) {
            const int oc = (idx - EW_OFF0) >> 4, n = (idx - EW_OFF0) & 15;
            if (n < 9) p = D.dw0 + oc * 9 + n;
            else if (n == 9 && D.db0 != nullptr) p = D.db0 + oc;
        } else {
            const int L = (idx - EW_OFFB) >> 4, oc = (idx - EW_OFFB) & 15;
            if (D.db[L] != nullptr) p = D.db[L] + oc;
        }
        if (p != nullptr) *p = P.accumulate[b] ? *p + t : t;
    }
}

int enc_wgrad_reduce_pair_launch(const float* pa, const EwDst& Da, int acc_a, const float* pb, const EwDst& Db, int acc_b, int G, hipStream_t st) {
    EwPair P;
    P.partial[0] = pa; P.partial[1] = pb;
    P.D[0] = Da; P.D[1] = Db;
    P.accumulate[0] = acc_a; P.accumulate[1] = acc_b;
    // any destination in common => the second branch must see the first's result: serial inside the blocks
    bool shared = Da.dw0 == Db.dw0 || (Da.db0 != nullptr && Da.db0 == Db.db0);
    for (int i = 0; i < 3; ++i) shared = shared || Da.dw[i] == Db.dw[i] || (Da.db[i] != nullptr && Da.db[i] == Db.db[i]);
    P.serial = shared ? 1 : 0;
    hipLaunchKernelGGL(enc_wgrad_reduce_pair, dim3(cdiv(EW_PER, 64), shared ? 1 : 2), dim3(64 * RED_SLICES), 0, st, P, G);
    return check_launch("enc_wgrad_reduce_pair");
}

int enc_wgrad_reduce_launch(const float* partial, const EwDst& D, int G, int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(enc_wgrad_reduce, dim3(cdiv(EW_PER, 64)), dim3(64 * RED_SLICES), 0, st, partial, D, G, accumulate);
    return check_launch("enc_wgrad_reduce");
}

}  // namespace mmif

namespace mmif {
// conv_x3.hip: the fp32 (split-operand) form of this pass for the three DenseBlock convs
bool wgrad_x3_dense_supported(const TV& tx, const TV& tg);
size_t wgrad_x3_dense_workspace();
int wgrad_x3_dense(const TV& tx, const TV& tg, float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, int accumulate, float* ws,
                   hipStream_t st);
}  // namespace mmif

using namespace mmif;

extern "C" int mmif_conv2d_image_in_wgrad(const float* img, const mmif_tensor* gy, float* dw, float* db, int32_t cout, int32_t ksize, int32_t accumulate,
                                          void* workspace, size_t workspace_bytes, void* stream);

extern "C" size_t mmif_dense_encoder_wgrad_workspace(void) {
    const size_t a = (size_t)EW_MAXG * EW_PER * sizeof(float), b = wgrad_x3_dense_workspace();
    return a > b ? a : b;
}

extern "C" int mmif_dense_encoder_wgrad(const float* img, const mmif_tensor* x, const mmif_tensor* gz, float* dw0, float* db0, float* dw1,
                                        float* db1, float* dw2, float* db2, float* dw3, float* db3, int32_t accumulate, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    if (int rc = validate_tensor(x, "x")) return rc;
    if (int rc = validate_tensor(gz, "gz")) return rc;
    MMIF_REQUIRE(img != nullptr && dw0 != nullptr && dw1 != nullptr && dw2 != nullptr && dw3 != nullptr, "dense_encoder_wgrad: NULL image / dW");
    if (x->dtype == MMIF_F32 && gz->dtype == MMIF_F32) {
        // fp32 tensors: the first layer on the image-side kernel, the three DenseBlock convs in ONE split-operand pass (csrc/conv_x3.hip)
        MMIF_REQUIRE(x->halo == 0 && x->cb >= 6, "dense_encoder_wgrad: x must be a halo-0 view of >= 6 channel blocks (x0 | x1 | x2)");
        MMIF_REQUIRE(gz->cb == 8 && (gz->halo == 0 || (gz->flags & MMIF_T_FOLDED)), "dense_encoder_wgrad: gz must be an 8-block view, halo 0 or folded");
        MMIF_REQUIRE(x->n == gz->n && x->h == gz->h && x->w == gz->w, "dense_encoder_wgrad: x / gz mismatch");
        if (workspace == nullptr || workspace_bytes < mmif_dense_encoder_wgrad_workspace()) {
            set_error("dense_encoder_wgrad: workspace too small");
            return MMIF_EWORKSPACE;
        }
        mmif_tensor g0 = *gz, g123 = *gz;
        g0.cb = 2;
        g123.cb_off = gz->cb_off + 2;
        g123.cb = 6;
        const TV tx = make_tv(x), tg = make_tv(&g123);
        MMIF_REQUIRE(wgrad_x3_dense_supported(tx, tg), "dense_encoder_wgrad: tensors not covered by the split-operand kernels");
        if (int rc = mmif_conv2d_image_in_wgrad(img, &g0, dw0, db0, 16, 3, accumulate, workspace, workspace_bytes, stream)) return rc;
        return wgrad_x3_dense(tx, tg, dw1, db1, dw2, db2, dw3, db3, accumulate, (float*)workspace, (hipStream_t)stream);
    }
    MMIF_REQUIRE(x->dtype == MMIF_BF16 && gz->dtype == MMIF_BF16, "dense_encoder_wgrad: bf16 tensors expected");
    MMIF_REQUIRE(x->halo == 0 && x->cb >= 6, "dense_encoder_wgrad: x must be a halo-0 view of >= 6 channel blocks (x0 | x1 | x2)");
    MMIF_REQUIRE(gz->cb == 8 && (gz->halo == 0 || (gz->flags & MMIF_T_FOLDED)), "dense_encoder_wgrad: gz must be an 8-block view, halo 0 or folded");
    MMIF_REQUIRE(x->n == gz->n && x->h == gz->h && x->w == gz->w, "dense_encoder_wgrad: x / gz mismatch");
    MMIF_REQUIRE(x->h >= 2 && x->w >= 2, "reflect padding needs h,w >= 2");
    MMIF_REQUIRE((long long)x->cb_total * x->h * x->w < (1ll << 31) && (long long)gz->cb_total * (gz->h + 2) * (gz->w + 2) < (1ll << 31),
                 "dense_encoder_wgrad: one image of x / gz must stay below 2^31 granules (32-bit tile offsets)");
    if (workspace == nullptr || workspace_bytes < mmif_dense_encoder_wgrad_workspace()) {
        set_error("dense_encoder_wgrad: workspace too small");
        return MMIF_EWORKSPACE;
    }
    TV tx = make_tv(x), tg = make_tv(gz);
    const int tiles_x = cdiv(tx.w, EW_T), tiles_y = cdiv(tx.h, EW_T);
    const int tpi = tiles_x * tiles_y, total = tpi * tx.n;
    const int G = total < EW_MAXG ? total : EW_MAXG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(enc_wgrad_kernel, dim3(G), dim3(256), 0, st, img, tx, tg, (float*)workspace, tiles_x, tpi, total, G);
    if (int rc = check_launch("enc_wgrad")) return rc;
    EwDst D;
    D.dw0 = dw0; D.db0 = db0;
    D.dw[0] = dw1; D.dw[1] = dw2; D.dw[2] = dw3;
    D.db[0] = db1; D.db[1] = db2; D.db[2] = db3;
    hipLaunchKernelGGL(enc_wgrad_reduce, dim3(cdiv(EW_PER, 64)), dim3(64 * RED_SLICES), 0, st, (const float*)workspace, D, G, accumulate);
    return check_launch("enc_wgrad_reduce");
}
