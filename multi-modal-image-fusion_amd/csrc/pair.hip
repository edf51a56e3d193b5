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

// ---- LDS tile: 16x16 pixels of ONE channel block plus a 1-pixel ring, both operands, raw storage granules split
// into 16-byte quads (bf16: 1 quad, fp32: 2 quads in separate planes so ds_read_b128 stays conflict free)
constexpr int PT = 16, PTP = 18, PLN = PTP * PTP;
template <typename T> struct Quads;
template <> struct Quads<bf16_t> { static constexpr int NQ = 1; };
template <> struct Quads<float> { static constexpr int NQ = 2; };

// REFLECT: activation (halo 0), reflect padding by index.  !REFLECT: gradient (halo 0, or halo 1 already folded), zero
// outside the image.  (y0, x0) = logical coordinates of tile pixel (0, 0); LDS slot (i, j) holds logical (y0-1+i, x0-1+j).
template <typename T, bool REFLECT>
__device__ inline void stage_tile(const TV& t, int n, int c, int y0, int x0, uint4* __restrict__ s) {
    constexpr int NQ = Quads<T>::NQ;
    for (int e = threadIdx.x; e < PLN; e += 256) {
        int y = y0 - 1 + e / PTP, x = x0 - 1 + e % PTP;
        bool ok = true;
        if (REFLECT) {
            y = reflect_idx(y, t.h);
            x = reflect_idx(x, t.w);
        } else {
            ok = y >= 0 && y < t.h && x >= 0 && x < t.w;
        }
        y = min(max(y, 0), t.h - 1);
        x = min(max(x, 0), t.w - 1);
        const uint4* p = reinterpret_cast<const uint4*>(t.base + t.gidx(n, c, y + t.halo, x + t.halo) * Elem<T>::gran_bytes);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const uint4 v = p[q];
            s[q * PLN + e] = ok ? v : make_uint4(0, 0, 0, 0);
        }
    }
}

// granule -> 4 packed fp32 pairs (channels 2i, 2i+1): the arithmetic below is v_pk_fma_f32 (2 lanes-values per
// instruction); these kernels are VALU-bound, not HBM-bound, once the taps come from LDS
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ inline f32x2 splat(float v) { return (f32x2){v, v}; }
__device__ inline f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// two-phase variant for persistent kernels: fetch a tile into registers (in flight while the previous tile is being
// consumed), then drop it into LDS
constexpr int PSL = (PLN + 255) / 256;   // staged granules per thread (2)
template <typename T>
__device__ inline void fetch_tile_reflect(const TV& t, int n, int c, int y0, int x0, uint4 (&r)[PSL][Quads<T>::NQ]) {
#pragma unroll
    for (int i = 0; i < PSL; ++i) {
        const int e = min((int)threadIdx.x + 256 * i, PLN - 1);
        const int y = min(max(reflect_idx(y0 - 1 + e / PTP, t.h), 0), t.h - 1);
        const int x = min(max(reflect_idx(x0 - 1 + e % PTP, t.w), 0), t.w - 1);
        const uint4* p = reinterpret_cast<const uint4*>(t.base + t.gidx(n, c, y, x) * Elem<T>::gran_bytes);
#pragma unroll
        for (int q = 0; q < Quads<T>::NQ; ++q) r[i][q] = p[q];
    }
}
template <typename T>
__device__ inline void drop_tile(const uint4 (&r)[PSL][Quads<T>::NQ], uint4* __restrict__ s) {
#pragma unroll
    for (int i = 0; i < PSL; ++i) {
        const int e = threadIdx.x + 256 * i;
        if (e < PLN) {
#pragma unroll
            for (int q = 0; q < Quads<T>::NQ; ++q) s[q * PLN + e] = r[i][q];
        }
    }
}

// the same two phases for any operand: REFLECT as in stage_tile; returns the in-image bits of this thread's PSL slots
template <typename T, bool REFLECT>
__device__ inline unsigned fetch_tile(const TV& t, int n, int c, int y0, int x0, uint4 (&r)[PSL][Quads<T>::NQ]) {
    unsigned okm = 0;
#pragma unroll
    for (int i = 0; i < PSL; ++i) {
        const int e = min((int)threadIdx.x + 256 * i, PLN - 1);
        int y = y0 - 1 + e / PTP, x = x0 - 1 + e % PTP;
        bool ok = true;
        if (REFLECT) {
            y = reflect_idx(y, t.h);
            x = reflect_idx(x, t.w);
        } else {
            ok = y >= 0 && y < t.h && x >= 0 && x < t.w;
        }
        y = min(max(y, 0), t.h - 1);
        x = min(max(x, 0), t.w - 1);
        const uint4* p = reinterpret_cast<const uint4*>(t.base + t.gidx(n, c, y + t.halo, x + t.halo) * Elem<T>::gran_bytes);
#pragma unroll
        for (int q = 0; q < Quads<T>::NQ; ++q) r[i][q] = p[q];
        okm |= (ok ? 1u : 0u) << i;
    }
    return okm;
}
template <typename T>
__device__ inline void drop_tile_masked(const uint4 (&r)[PSL][Quads<T>::NQ], unsigned okm, uint4* __restrict__ s) {
#pragma unroll
    for (int i = 0; i < PSL; ++i) {
        const int e = threadIdx.x + 256 * i;
        if (e < PLN) {
#pragma unroll
            for (int q = 0; q < Quads<T>::NQ; ++q) s[q * PLN + e] = ((okm >> i) & 1u) ? r[i][q] : make_uint4(0, 0, 0, 0);
        }
    }
}

template <typename T>
__device__ inline void tile_read(const uint4* __restrict__ s, int idx, f32x2 (&v)[4]) {
    if (Quads<T>::NQ == 1) {
        const uint4 a = s[idx];
        const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (f32x2){__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)};
    } else {
        const uint4 a = s[idx], b = s[PLN + idx];
        v[0] = (f32x2){__uint_as_float(a.x), __uint_as_float(a.y)};
        v[1] = (f32x2){__uint_as_float(a.z), __uint_as_float(a.w)};
        v[2] = (f32x2){__uint_as_float(b.x), __uint_as_float(b.y)};
        v[3] = (f32x2){__uint_as_float(b.z), __uint_as_float(b.w)};
    }
}
template <typename T>
__device__ inline void load_pairs(const float (&u)[8], f32x2 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (f32x2){u[2 * i], u[2 * i + 1]};
}
__device__ inline void unpack_pairs(const f32x2 (&v)[4], float (&u)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { u[2 * i] = v[i].x; u[2 * i + 1] = v[i].y; }
}

// thread -> tile pixel (16 x 16 tiles, one granule per thread).  ds_read_b128 is served in four NON-contiguous 16-lane groups
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, and the same + 32): with lane = 16 ty + tx a group reads half of one tile row and half of
// the next, whose 16-byte slots overlap mod 256 B at a row pitch of 18 granules (2-way conflicts: SQ_LDS_BANK_CONFLICT was 1.6 x the
// active LDS cycles of pairconv_bwd_kernel).  Flipping the row bit of the lanes with tx in 4..11 makes every group one whole tile row =
// 256 contiguous bytes.  Global accesses touch the same lines as before (a permutation inside the wave).
__device__ inline int tile_row_flip(int tx) { return (tx >= 4 && tx < 12) ? 1 : 0; }

struct TileId { int n, c, y0, x0; };
// 32-bit arithmetic on purpose: 64-bit divisions of the (wave-uniform) tile index cost ~900 scalar instructions per block
__device__ inline TileId tile_of(unsigned tile, int tiles_x, int tiles_y, int cb) {
    TileId t;
    const unsigned tx = (unsigned)tiles_x, tyy = (unsigned)tiles_y, ucb = (unsigned)cb;
    unsigned r = tile / tx;
    t.x0 = (int)(tile - r * tx) * PT;
    unsigned r2 = r / tyy;
    t.y0 = (int)(r - r2 * tyy) * PT;
    const unsigned r3 = r2 / ucb;
    t.c = (int)(r2 - r3 * ucb);
    t.n = (int)r3;
    return t;
}

template <typename T, int NOUT>
__global__ __launch_bounds__(256, 4) void pairconv_fwd_kernel(TV a, TV b, const float* __restrict__ w, const float* __restrict__ bias,
                                                            TV oa, TV ob, int relu, TV r1, TV r2, int has_res, int tiles_x, int tiles_y) {
    constexpr int NQ = Quads<T>::NQ;
    __shared__ __attribute__((aligned(16))) uint4 s_a[NQ * PLN], s_b[NQ * PLN];
    const TileId ti = tile_of(blockIdx.x, tiles_x, tiles_y, a.cb);
    stage_tile<T, true>(a, ti.n, ti.c, ti.y0, ti.x0, s_a);
    stage_tile<T, true>(b, ti.n, ti.c, ti.y0, ti.x0, s_b);
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = (threadIdx.x >> 4) ^ tile_row_flip(tx);
    const int y = ti.y0 + ty, x = ti.x0 + tx, n = ti.n, c = ti.c;
    if (y >= a.h || x >= a.w) return;
    f32x2 acc[NOUT][4];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        const float bo = bias ? bias[o] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[o][k] = splat(bo);
    }
    // a real loop over tap rows: fully unrolled, the compiler runs output 0's whole chain first and keeps all 18 unpacked
    // granules alive for output 1 (160 VGPRs / spills)
#pragma unroll 1
    for (int u = 0; u < 3; ++u) {
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            f32x2 va[4], vb[4];
            const int idx = (ty + u) * PTP + tx + v;
            tile_read<T>(s_a, idx, va);
            tile_read<T>(s_b, idx, vb);
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                const f32x2 wa = splat(w[(o * 2 + 0) * 9 + u * 3 + v]), wb = splat(w[(o * 2 + 1) * 9 + u * 3 + v]);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[o][k] = pk_fma(wb, vb[k], pk_fma(wa, va[k], acc[o][k]));
            }
        }
    }
    float out[NOUT][8];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        unpack_pairs(acc[o], out[o]);
        if (relu) {
#pragma unroll
            for (int k = 0; k < 8; ++k) out[o][k] = fmaxf(out[o][k], 0.f);
        }
    }
    if (has_res) {  // + feat1 + feat2 (core/model.py:141), added to output 0
        float u[8], v[8];
        Elem<T>::load(r1.base + r1.gidx(n, c, y, x) * Elem<T>::gran_bytes, u);
        Elem<T>::load(r2.base + r2.gidx(n, c, y, x) * Elem<T>::gran_bytes, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) out[0][k] = (out[0][k] + u[k]) + v[k];
    }
    Elem<T>::store(oa.base + oa.gidx(n, c, y, x) * Elem<T>::gran_bytes, out[0]);
    if (NOUT == 2) Elem<T>::store(ob.base + ob.gidx(n, c, y, x) * Elem<T>::gran_bytes, out[NOUT - 1]);
}

// ---- bf16 forward with 1 x 4 output strips.  The kernel above is instruction bound (per output 144 v_pk_fma_f32 + 144 shift / and
// operations that turn 18 bf16 granules into fp32 pairs + addressing: ~540 vector instructions, 263 us of issue time at B = 32 256^2):
// neighbouring outputs unpack the same granules again.  Here a thread owns 4 consecutive outputs of a row and walks the 3 x 6 granule
// window once (9 granule unpacks per output and operand instead of 18 for both), 32 x 32 output tiles (halo re-reads 1.13 x instead of
// 1.27 x).  Taps are visited in the same order per output (row-major, a before b), so the results are bit-identical.
// LDS: 34 x 34 window, columns interleaved mod 4 so that the j-th granule of the 8 strips of a row is contiguous (conflict-free
// ds_read_b128), row pitch 40 granules = 640 B = 128 mod 256 so that the two rows of a 16-lane pass use disjoint banks.
constexpr int FT = 32, FW = FT + 2, FQ = 9, FPITCH = 40;
constexpr int FSL = (FW * FW + 255) / 256;   // staged granules per thread and operand (5)
__device__ inline void unpack_q(const uint4& a, f32x2 (&v)[4]) {
    const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (f32x2){__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)};
}

template <int NOUT>
__global__ __launch_bounds__(256, 3) void pairconv_fwd_strip_kernel(TV a, TV b, const float* __restrict__ w, const float* __restrict__ bias,
                                                                    TV oa, TV ob, int relu, TV r1, TV r2, int has_res, int tiles_x, int tiles_y) {
    typedef bf16_t T;
    constexpr unsigned GB = 16;
    __shared__ __attribute__((aligned(16))) uint4 s_a[FW * FPITCH], s_b[FW * FPITCH];
    // 32-bit tile decode (wave-uniform)
    const unsigned tile = blockIdx.x, utx = (unsigned)tiles_x, uty = (unsigned)tiles_y, ucb = (unsigned)a.cb;
    const unsigned q1 = tile / utx, q2 = q1 / uty, q3 = q2 / ucb;
    const int x0 = (int)(tile - q1 * utx) * FT, y0 = (int)(q1 - q2 * uty) * FT, c = (int)(q2 - q3 * ucb), n = (int)q3;
    {   // stage both operands in one round trip: 2 * FSL loads in flight, one shared 32-bit in-plane offset per slot
        const char* pa = a.base + ((long long)n * a.img + (long long)(a.cb_off + c) * a.plane) * GB;
        const char* pb = b.base + ((long long)n * b.img + (long long)(b.cb_off + c) * b.plane) * GB;
        uint4 ra[FSL], rb[FSL];
        int slot[FSL];
#pragma unroll
        for (int k = 0; k < FSL; ++k) {
            const int e = min((int)threadIdx.x + 256 * k, FW * FW - 1);
            const int i = e / FW, jp = e - i * FW;
            const int y = min(max(reflect_idx(y0 - 1 + i, a.h), 0), a.h - 1), x = min(max(reflect_idx(x0 - 1 + jp, a.w), 0), a.w - 1);
            const unsigned off = (unsigned)(y * a.ws + x) * GB;
            ra[k] = *reinterpret_cast<const uint4*>(pa + off);
            rb[k] = *reinterpret_cast<const uint4*>(pb + off);
            slot[k] = i * FPITCH + (jp & 3) * FQ + (jp >> 2);
        }
#pragma unroll
        for (int k = 0; k < FSL; ++k)
            if ((int)threadIdx.x + 256 * k < FW * FW) {
                s_a[slot[k]] = ra[k];
                s_b[slot[k]] = rb[k];
            }
    }
    __syncthreads();
    const int sx = threadIdx.x & 7, row = threadIdx.x >> 3;
    const int y = y0 + row, xb0 = x0 + 4 * sx;
    const bool active = y < a.h && xb0 < a.w;
    f32x2 acc[NOUT][4][4];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        const float bo = bias ? bias[o] : 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[o][p][k] = splat(bo);
    }
#pragma unroll 1   // (fully unrolled, the compiler hoists all 36 unpacked granules: spills)
    for (int u = 0; u < 3; ++u) {
        const int rb_ = (row + u) * FPITCH + sx;
#pragma unroll
        for (int j = 0; j < 6; ++j) {   // window column 4*sx + j feeds output p = j - v through tap column v
            f32x2 va[4], vb[4];
            unpack_q(s_a[rb_ + (j & 3) * FQ + (j >> 2)], va);
            unpack_q(s_b[rb_ + (j & 3) * FQ + (j >> 2)], vb);
#pragma unroll
            for (int v = 2; v >= 0; --v) {
                const int p = j - v;
                if (p < 0 || p > 3) continue;
#pragma unroll
                for (int o = 0; o < NOUT; ++o) {
                    const f32x2 wa = splat(w[(o * 2 + 0) * 9 + u * 3 + v]), wb = splat(w[(o * 2 + 1) * 9 + u * 3 + v]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[o][p][k] = pk_fma(wb, vb[k], pk_fma(wa, va[k], acc[o][p][k]));
                }
            }
        }
    }
    // epilogue through LDS: a thread's 4 granules are 64 B apart from its neighbour's (every store instruction would write a quarter
    // of 32 cache lines); drop them p-major into the (now free) windows and store rows of 32 granules = 4 whole lines instead.
    // residuals that are other tensors (PFNetv2's last fuse layer: operands H2, residuals feat1 / feat2), one output: the tile goes
    // through LDS in fp32 (channels 0-3 in one window, 4-7 in the other) and (out + r1) + r2 is formed in the row-contiguous phase
    const bool lazy_res = NOUT == 1 && has_res == 1;
    uint4 pk[NOUT + 1][4];
    if (active) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float out[NOUT][8];
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                unpack_pairs(acc[o][p], out[o]);
                if (relu) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) out[o][k] = fmaxf(out[o][k], 0.f);
                }
            }
            if (has_res == 2) {  // + feat1 + feat2 (core/model.py:141), added to output 0: the residuals ARE the operands, whose
                                 // centre granules sit in the windows (window column 4*sx + p + 1, row + 1)
                f32x2 ua[4], ub[4];
                const int ctr = (row + 1) * FPITCH + sx + ((p + 1) & 3) * FQ + ((p + 1) >> 2);
                unpack_q(s_a[ctr], ua);
                unpack_q(s_b[ctr], ub);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    out[0][2 * k] = (out[0][2 * k] + ua[k].x) + ub[k].x;
                    out[0][2 * k + 1] = (out[0][2 * k + 1] + ua[k].y) + ub[k].y;
                }
            } else if (has_res && !lazy_res && xb0 + p < a.w) {
                float uu[8], vv[8];
                Elem<T>::load(r1.base + r1.gidx(n, c, y, xb0 + p) * GB, uu);
                Elem<T>::load(r2.base + r2.gidx(n, c, y, xb0 + p) * GB, vv);
#pragma unroll
                for (int k = 0; k < 8; ++k) out[0][k] = (out[0][k] + uu[k]) + vv[k];
            }
            if (lazy_res) {   // fp32 halves: the residuals are added where they can be loaded a row at a time (below)
                pk[0][p] = make_uint4(__float_as_uint(out[0][0]), __float_as_uint(out[0][1]), __float_as_uint(out[0][2]), __float_as_uint(out[0][3]));
                pk[NOUT][p] = make_uint4(__float_as_uint(out[0][4]), __float_as_uint(out[0][5]), __float_as_uint(out[0][6]), __float_as_uint(out[0][7]));
            } else {
#pragma unroll
                for (int o = 0; o < NOUT; ++o)
                    pk[o][p] = make_uint4(pack_bf16x2(out[o][0], out[o][1]), pack_bf16x2(out[o][2], out[o][3]), pack_bf16x2(out[o][4], out[o][5]),
                                          pack_bf16x2(out[o][6], out[o][7]));
            }
        }
    }
    __syncthreads();   // every wave is done with the operand windows
    if (active) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            s_a[row * FPITCH + p * FQ + sx] = pk[0][p];
            if (NOUT == 2) s_b[row * FPITCH + p * FQ + sx] = pk[NOUT - 1][p];
            else if (lazy_res) s_b[row * FPITCH + p * FQ + sx] = pk[NOUT][p];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = threadIdx.x + 256 * k, r = e >> 5, col = e & 31;
        const int yy = y0 + r, xx = x0 + col;
        if (yy < a.h && xx < a.w) {
            const int sl = r * FPITCH + (col & 3) * FQ + (col >> 2);
            if (lazy_res) {
                const uint4 lo = s_a[sl], hi = s_b[sl];
                float o8[8] = {__uint_as_float(lo.x), __uint_as_float(lo.y), __uint_as_float(lo.z), __uint_as_float(lo.w),
                               __uint_as_float(hi.x), __uint_as_float(hi.y), __uint_as_float(hi.z), __uint_as_float(hi.w)};
                float uu[8], vv[8];
                Elem<T>::load(r1.base + r1.gidx(n, c, yy, xx) * GB, uu);
                Elem<T>::load(r2.base + r2.gidx(n, c, yy, xx) * GB, vv);
#pragma unroll
                for (int q = 0; q < 8; ++q) o8[q] = (o8[q] + uu[q]) + vv[q];
                Elem<T>::store(oa.base + oa.gidx(n, c, yy, xx) * GB, o8);
                continue;
            }
            *reinterpret_cast<uint4*>(oa.base + oa.gidx(n, c, yy, xx) * GB) = s_a[sl];
            if (NOUT == 2) *reinterpret_cast<uint4*>(ob.base + ob.gidx(n, c, yy, xx) * GB) = s_b[sl];
        }
    }
}

// gxa/gxb (halo 1): padded-domain gradient w.r.t. the operands A, B:
//   gx_c[p] = sum_o sum_tap w[o][c][tap] * g_o[p - tap + 1]   (g zero outside the image), p in [-1, h] x [-1, w]
//   (+ add[p]: the residual path's gradient, same for A and B)   then  * [x_c(R(p)) > 0] for the masked channel blocks.
// Tiles cover the STORED domain [h+2][w+2] of gx; g must be halo 0 or already folded.
template <typename T, int NOUT>
__global__ __launch_bounds__(256) void pairconv_dgrad_kernel(TV ga, TV gb, const float* __restrict__ w, TV xa, TV xb, TV gxa, TV gxb,
                                                              unsigned long long mask_bits, TV add, int has_add, int tiles_x, int tiles_y) {
    constexpr int NQ = Quads<T>::NQ;
    __shared__ __attribute__((aligned(16))) uint4 s_a[NQ * PLN], s_b[NOUT == 2 ? NQ * PLN : 1];
    const TileId ti = tile_of(blockIdx.x, tiles_x, tiles_y, gxa.cb);
    // stored (ys, xs) <-> logical p = (ys - 1, xs - 1)
    stage_tile<T, false>(ga, ti.n, ti.c, ti.y0 - 1, ti.x0 - 1, s_a);
    if (NOUT == 2) stage_tile<T, false>(gb, ti.n, ti.c, ti.y0 - 1, ti.x0 - 1, s_b);
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = (threadIdx.x >> 4) ^ tile_row_flip(tx);
    const int ys = ti.y0 + ty, xs = ti.x0 + tx, n = ti.n, c = ti.c;
    if (ys >= gxa.hs || xs >= gxa.ws) return;
    const int py = ys - 1, px = xs - 1;
    f32x2 pa[4], pb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) pa[k] = pb[k] = splat(0.f);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int idx = (ty + 2 - t / 3) * PTP + tx + 2 - t % 3;   // g at p - tap + 1
        f32x2 g0[4], g1[4];
        tile_read<T>(s_a, idx, g0);
        if (NOUT == 2) tile_read<T>(s_b, idx, g1);
        const f32x2 w00 = splat(w[(0 * 2 + 0) * 9 + t]), w01 = splat(w[(0 * 2 + 1) * 9 + t]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pa[k] = pk_fma(w00, g0[k], pa[k]);
            pb[k] = pk_fma(w01, g0[k], pb[k]);
        }
        if (NOUT == 2) {
            const f32x2 w10 = splat(w[(1 * 2 + 0) * 9 + t]), w11 = splat(w[(1 * 2 + 1) * 9 + t]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pa[k] = pk_fma(w10, g1[k], pa[k]);
                pb[k] = pk_fma(w11, g1[k], pb[k]);
            }
        }
    }
    float da[8], db[8];
    unpack_pairs(pa, da);
    unpack_pairs(pb, db);
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

// dw[o][c][tap] = sum over (n, channel, p) of g_o[p] * reflect_pad(x_c)[p + tap - 1];  db[o] = sum g_o.
// Persistent blocks walking tiles, NOUT*19 register accumulators per thread, block reduction -> partial[block][NOUT*19].
constexpr int PAIR_WG_BLOCKS = 2048;
constexpr int PAIR_BWD_BLOCKS = 2048;

template <typename T, int NOUT>
__global__ __launch_bounds__(256, 4) void pairconv_wgrad_kernel(TV xa, TV xb, TV ga, TV gb, float* __restrict__ partial, int tiles_x,
                                                                 int tiles_y, unsigned ntiles) {
    constexpr int PER = NOUT * 19;
    constexpr int NQ = Quads<T>::NQ;
    __shared__ __attribute__((aligned(16))) uint4 s_a[NQ * PLN], s_b[NQ * PLN];
    __shared__ float red[4][PER];
    // packed accumulators: lane-pair sums stay separate until the end (no horizontal add per tap)
    f32x2 acc[NOUT][2][9], accb[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        accb[o] = splat(0.f);
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[o][0][t] = acc[o][1][t] = splat(0.f);
    }
    const int tx = threadIdx.x & 15, ty = (threadIdx.x >> 4) ^ tile_row_flip(tx);
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const TileId ti = tile_of(tile, tiles_x, tiles_y, xa.cb);
        __syncthreads();
        stage_tile<T, true>(xa, ti.n, ti.c, ti.y0, ti.x0, s_a);
        stage_tile<T, true>(xb, ti.n, ti.c, ti.y0, ti.x0, s_b);
        __syncthreads();
        const int y = ti.y0 + ty, x = ti.x0 + tx;
        if (y >= xa.h || x >= xa.w) continue;
        float gf[NOUT][8];
        load_grad_fold<T>(ga, ti.n, ti.c, y, x, gf[0]);
        if (NOUT == 2) load_grad_fold<T>(gb, ti.n, ti.c, y, x, gf[NOUT - 1]);
        f32x2 g[NOUT][4];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            load_pairs<T>(gf[o], g[o]);
            accb[o] += (g[o][0] + g[o][1]) + (g[o][2] + g[o][3]);
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            f32x2 va[4], vb[4];
            const int idx = (ty + t / 3) * PTP + tx + t % 3;
            tile_read<T>(s_a, idx, va);
            tile_read<T>(s_b, idx, vb);
#pragma unroll
            for (int o = 0; o < NOUT; ++o)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc[o][0][t] = pk_fma(g[o][k], va[k], acc[o][0][t]);
                    acc[o][1][t] = pk_fma(g[o][k], vb[k], acc[o][1][t]);
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
                float v = acc[o][cc][t].x + acc[o][cc][t].y;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                if (lane == 0) red[wave][(o * 2 + cc) * 9 + t] = v;
            }
        float v = accb[o].x + accb[o].y;
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

// one block PER OUTPUT e: fixed-order sum of the G partials (256 interleaved chains -> block tree).  (A single block walking all
// 19 / 38 outputs in turn took 30 - 80 us per layer: 0.17 ms of a PFNetv2 step.)
__global__ __launch_bounds__(256) void pairconv_wgrad_reduce(const float* __restrict__ partial, int G, int per, int nout,
                                                             float* __restrict__ dw, float* __restrict__ db, int accumulate) {
    __shared__ float red[16];
    const int e = blockIdx.x;
    float s = 0.f;
    for (int g = threadIdx.x; g < G; g += 256) s += partial[(long long)g * per + e];
    const float tot = block_sum(s, red);
    if (threadIdx.x == 0) {
        float* dst = e < nout * 18 ? dw + e : db + (e - nout * 18);
        *dst = accumulate ? *dst + tot : tot;
    }
}

// ---- backward of one pair-conv layer in ONE pass: the dgrad above and the wgrad above read the same two tensors (the layer's
// output gradient g and its input x: taps for one, ReLU mask for the other), so apart they move 10 tensor passes through HBM and
// together 6.  Persistent blocks walk 16x16 tiles of gx's STORED domain [h+2][w+2]; thread (ty, tx) owns padded position
// p = (y0 + ty - 1, x0 + tx - 1): it writes gx_a/gx_b[p] and, when p is inside the image, adds g[p] * reflect_pad(x)[p + tap - 1]
// to its NOUT*19 accumulators.  bf16 storage: the weight-gradient products run as v_dot2c_f32_bf16 on the raw granules (two
// channels per instruction, no unpacking: the products are exact in fp32 either way); the dgrad keeps fp32 weights.
template <typename T>
__device__ inline void unpack_gran(const uint4 (&r)[Quads<T>::NQ], float (&v)[8]) {
    if (Quads<T>::NQ == 1) {
        const uint32_t w[4] = {r[0].x, r[0].y, r[0].z, r[0].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    } else {
        const uint4 a = r[0], b = r[Quads<T>::NQ - 1];
        const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __uint_as_float(w[i]);
    }
}
template <bool BF> struct WAcc;
template <> struct WAcc<true> {
    typedef float type;
    static __device__ inline float zero() { return 0.f; }
    static __device__ inline float hsum(float v) { return v; }
};
template <> struct WAcc<false> {
    typedef f32x2 type;
    static __device__ inline f32x2 zero() { return splat(0.f); }
    static __device__ inline float hsum(f32x2 v) { return v.x + v.y; }
};
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ inline float dot2_bf16(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}
__device__ inline float dot8_bf16(const uint4& a, const uint4& b, float c) {
    return dot2_bf16(a.w, b.w, dot2_bf16(a.z, b.z, dot2_bf16(a.y, b.y, dot2_bf16(a.x, b.x, c))));
}

template <typename T, int NOUT>
__global__ __launch_bounds__(256, Quads<T>::NQ == 1 ? 3 : 2) void pairconv_bwd_kernel(TV ga, TV gb, const float* __restrict__ w, TV xa, TV xb, TV gxa, TV gxb,
                                                              unsigned long long mask_bits, TV add, int has_add,
                                                              float* __restrict__ partial, int tiles_x, int tiles_y, unsigned ntiles) {
    constexpr int PER = NOUT * 19;
    constexpr int NQ = Quads<T>::NQ;
    constexpr bool BF = NQ == 1;
    typedef typename WAcc<BF>::type acc_t;
    // g is kept in LDS as fp32 (two planes, as the fp32 storage type has them anyway): its 9 dgrad taps per output are then plain
    // 16-byte reads instead of a read + 8 shift / and operations each (161 of ~700 vector instructions per output)
    __shared__ __attribute__((aligned(16))) uint4 s_g[NOUT][2 * PLN], s_x[2][NQ * PLN];
    __shared__ float red[4][PER];
    acc_t acc[NOUT][2][9], accb[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        accb[o] = WAcc<BF>::zero();
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[o][0][t] = acc[o][1][t] = WAcc<BF>::zero();
    }
    const int tx = threadIdx.x & 15, ty = (threadIdx.x >> 4) ^ tile_row_flip(tx);
    // two-phase staging: the next tile's 4 operand windows (and this thread's granule of the residual gradient) are in flight
    // in registers while this tile is consumed.  Addresses = wave-uniform plane base + a 32-bit in-plane offset that ga/gb and
    // xa/xb share (the C ABI checks equal shapes and halos).
    constexpr unsigned GB = Elem<T>::gran_bytes;
    uint4 rg[NOUT][PSL][NQ], rx[2][PSL][NQ], radd[NQ];
    unsigned okm = 0;
    int ei[PSL], ej[PSL];
#pragma unroll
    for (int k = 0; k < PSL; ++k) {
        const int e = min((int)threadIdx.x + 256 * k, PLN - 1);
        ei[k] = e / PTP;
        ej[k] = e - ei[k] * PTP;
    }
    const int H = ga.h, W = ga.w;
    auto fetch = [&](unsigned tile) {
        const TileId tn = tile_of(tile, tiles_x, tiles_y, gxa.cb);
        const char* pga = ga.base + ((long long)tn.n * ga.img + (long long)(ga.cb_off + tn.c) * ga.plane) * GB;
        const char* pgb = gb.base + ((long long)tn.n * gb.img + (long long)(gb.cb_off + tn.c) * gb.plane) * GB;
        const char* pxa = xa.base + ((long long)tn.n * xa.img + (long long)(xa.cb_off + tn.c) * xa.plane) * GB;
        const char* pxb = xb.base + ((long long)tn.n * xb.img + (long long)(xb.cb_off + tn.c) * xb.plane) * GB;
        okm = 0;
        // slot (i, j) <-> logical (y0 - 2 + i, x0 - 2 + j): g zero outside the image, x reflected
#pragma unroll
        for (int k = 0; k < PSL; ++k) {
            const int y = tn.y0 - 2 + ei[k], x = tn.x0 - 2 + ej[k];
            const bool ok = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            const unsigned og = (unsigned)((min(max(y, 0), H - 1) + ga.halo) * ga.ws + min(max(x, 0), W - 1) + ga.halo) * GB;
            const unsigned ox = (unsigned)(min(max(reflect_idx(y, H), 0), H - 1) * xa.ws + min(max(reflect_idx(x, W), 0), W - 1)) * GB;
            okm |= (ok ? 1u : 0u) << k;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                rg[0][k][q] = *reinterpret_cast<const uint4*>(pga + og + 16 * q);
                if (NOUT == 2) rg[NOUT - 1][k][q] = *reinterpret_cast<const uint4*>(pgb + og + 16 * q);
                rx[0][k][q] = *reinterpret_cast<const uint4*>(pxa + ox + 16 * q);
                rx[1][k][q] = *reinterpret_cast<const uint4*>(pxb + ox + 16 * q);
            }
        }
        if (has_add) {   // halo 0 or folded (ring zeroed): the stored value at the clamped position, masked by bit 2 of okm
            const int y = tn.y0 + ty - 1, x = tn.x0 + tx - 1;
            const bool ok = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            const char* pad = add.base + ((long long)tn.n * add.img + (long long)(add.cb_off + tn.c) * add.plane) * GB;
            const unsigned oa = (unsigned)((min(max(y, 0), H - 1) + add.halo) * add.ws + min(max(x, 0), W - 1) + add.halo) * GB;
#pragma unroll
            for (int q = 0; q < NQ; ++q) radd[q] = *reinterpret_cast<const uint4*>(pad + oa + 16 * q);
            okm |= (ok ? 1u : 0u) << 2;
        }
    };
    if (blockIdx.x < ntiles) fetch(blockIdx.x);
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const TileId ti = tile_of(tile, tiles_x, tiles_y, gxa.cb);
        const int n = ti.n, c = ti.c;
        __syncthreads();
        if constexpr (BF) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o)
#pragma unroll
                for (int i = 0; i < PSL; ++i) {
                    const int e = threadIdx.x + 256 * i;
                    if (e < PLN) {
                        const uint4 r = ((okm >> i) & 1u) ? rg[o][i][0] : make_uint4(0, 0, 0, 0);
                        s_g[o][e] = make_uint4(r.x << 16, r.x & 0xffff0000u, r.y << 16, r.y & 0xffff0000u);
                        s_g[o][PLN + e] = make_uint4(r.z << 16, r.z & 0xffff0000u, r.w << 16, r.w & 0xffff0000u);
                    }
                }
        } else {
            drop_tile_masked<T>(rg[0], okm, s_g[0]);
            if (NOUT == 2) drop_tile_masked<T>(rg[NOUT - 1], okm, s_g[NOUT - 1]);
        }
        drop_tile_masked<T>(rx[0], 3u, s_x[0]);
        drop_tile_masked<T>(rx[1], 3u, s_x[1]);
        __syncthreads();
        float addv[8];
        const bool add_ok = (okm >> 2) & 1u;
        if (has_add) unpack_gran<T>(radd, addv);
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);
        const int ys = ti.y0 + ty, xs = ti.x0 + tx;
        if (ys >= gxa.hs || xs >= gxa.ws) continue;
        const int py = ys - 1, px = xs - 1;
        const int ctr = (ty + 1) * PTP + tx + 1;
        {
            f32x2 pa[4], pb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) pa[k] = pb[k] = splat(0.f);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int idx = (ty + 2 - t / 3) * PTP + tx + 2 - t % 3;   // g at p - tap + 1
                f32x2 g0[4], g1[4];
                tile_read<float>(s_g[0], idx, g0);
                if (NOUT == 2) tile_read<float>(s_g[NOUT - 1], idx, g1);
                const f32x2 w00 = splat(w[(0 * 2 + 0) * 9 + t]), w01 = splat(w[(0 * 2 + 1) * 9 + t]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    pa[k] = pk_fma(w00, g0[k], pa[k]);
                    pb[k] = pk_fma(w01, g0[k], pb[k]);
                }
                if (NOUT == 2) {
                    const f32x2 w10 = splat(w[(1 * 2 + 0) * 9 + t]), w11 = splat(w[(1 * 2 + 1) * 9 + t]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        pa[k] = pk_fma(w10, g1[k], pa[k]);
                        pb[k] = pk_fma(w11, g1[k], pb[k]);
                    }
                }
            }
            float da[8], db[8];
            unpack_pairs(pa, da);
            unpack_pairs(pb, db);
            if (has_add && add_ok) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { da[k] += addv[k]; db[k] += addv[k]; }
            }
            if ((mask_bits >> c) & 1ull) {   // x(R(p)) sits at the tile centre
                f32x2 va[4], vb[4];
                tile_read<T>(s_x[0], ctr, va);
                tile_read<T>(s_x[1], ctr, vb);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    da[2 * k] = va[k].x > 0.f ? da[2 * k] : 0.f;
                    da[2 * k + 1] = va[k].y > 0.f ? da[2 * k + 1] : 0.f;
                    db[2 * k] = vb[k].x > 0.f ? db[2 * k] : 0.f;
                    db[2 * k + 1] = vb[k].y > 0.f ? db[2 * k + 1] : 0.f;
                }
            }
            Elem<T>::store(gxa.base + gxa.gidx(n, c, ys, xs) * Elem<T>::gran_bytes, da);
            Elem<T>::store(gxb.base + gxb.gidx(n, c, ys, xs) * Elem<T>::gran_bytes, db);
        }
        if (py < 0 || py >= gxa.h || px < 0 || px >= gxa.w) continue;
        if constexpr (BF) {
            uint4 g[NOUT];
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {   // back to the bf16 granule (exact: the fp32 values are widened bf16)
                const uint4 lo = s_g[o][ctr], hi = s_g[o][PLN + ctr];
                g[o] = make_uint4((lo.x >> 16) | lo.y, (lo.z >> 16) | lo.w, (hi.x >> 16) | hi.y, (hi.z >> 16) | hi.w);
            }
            const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
#pragma unroll
            for (int o = 0; o < NOUT; ++o) accb[o] = dot8_bf16(g[o], ones, accb[o]);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int idx = (ty + t / 3) * PTP + tx + t % 3;
                const uint4 va = s_x[0][idx], vb = s_x[1][idx];
#pragma unroll
                for (int o = 0; o < NOUT; ++o) {
                    acc[o][0][t] = dot8_bf16(g[o], va, acc[o][0][t]);
                    acc[o][1][t] = dot8_bf16(g[o], vb, acc[o][1][t]);
                }
            }
        } else {
            f32x2 g[NOUT][4];
            tile_read<T>(s_g[0], ctr, g[0]);
            if (NOUT == 2) tile_read<T>(s_g[NOUT - 1], ctr, g[NOUT - 1]);
#pragma unroll
            for (int o = 0; o < NOUT; ++o) accb[o] += (g[o][0] + g[o][1]) + (g[o][2] + g[o][3]);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                f32x2 va[4], vb[4];
                const int idx = (ty + t / 3) * PTP + tx + t % 3;
                tile_read<T>(s_x[0], idx, va);
                tile_read<T>(s_x[1], idx, vb);
#pragma unroll
                for (int o = 0; o < NOUT; ++o)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        acc[o][0][t] = pk_fma(g[o][k], va[k], acc[o][0][t]);
                        acc[o][1][t] = pk_fma(g[o][k], vb[k], acc[o][1][t]);
                    }
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
                float v = WAcc<BF>::hsum(acc[o][cc][t]);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                if (lane == 0) red[wave][(o * 2 + cc) * 9 + t] = v;
            }
        float v = WAcc<BF>::hsum(accb[o]);
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
    const int tiles_x = cdiv(ta.w, PT), tiles_y = cdiv(ta.h, PT);
    const long long ntiles = (long long)ta.n * ta.cb * tiles_x * tiles_y;
    MMIF_REQUIRE(ntiles < (1ll << 31), "pairconv_fwd: too many tiles");
    const char* env = getenv("MMIF_PAIR_STRIP");   // read per call (A/B and the bit-identity test flip it inside one process)
    const bool strips = !(env && env[0] == '0');
    if (a->dtype == MMIF_BF16 && strips && (long long)ta.h * ta.w * 16 < (1ll << 31)) {
        const int sx_ = cdiv(ta.w, FT), sy_ = cdiv(ta.h, FT);
        const long long nt = (long long)ta.n * ta.cb * sx_ * sy_;
        MMIF_REQUIRE(nt < (1ll << 31), "pairconv_fwd: too many tiles");
        // residuals that are the operands themselves (PFNetv2: feat1, feat2) come out of the LDS windows
        const bool res_ops = has_res && res1->data == a->data && res1->cb_off == a->cb_off && res1->cb_total == a->cb_total &&
                             res2->data == b->data && res2->cb_off == b->cb_off && res2->cb_total == b->cb_total;
        const int res_mode = has_res ? (res_ops ? 2 : 1) : 0;
        if (nout == 2)
            hipLaunchKernelGGL((pairconv_fwd_strip_kernel<2>), dim3((unsigned)nt), dim3(256), 0, st, ta, tb, w, bias, toa, tob, relu, t1, t2,
                               res_mode, sx_, sy_);
        else
            hipLaunchKernelGGL((pairconv_fwd_strip_kernel<1>), dim3((unsigned)nt), dim3(256), 0, st, ta, tb, w, bias, toa, tob, relu, t1, t2,
                               res_mode, sx_, sy_);
        return check_launch("pairconv_fwd");
    }
    PAIR_LAUNCH(a->dtype, nout, pairconv_fwd_kernel, (unsigned)ntiles, ta, tb, w, bias, toa, tob, relu, t1, t2, has_res ? 1 : 0, tiles_x, tiles_y);
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
    MMIF_REQUIRE(ga->halo == 0 || (ga->flags & MMIF_T_FOLDED), "pairconv_dgrad: a halo-1 gradient must be folded first (mmif_fold_halo)");
    if (nout == 2 && gb) MMIF_REQUIRE(gb->halo == 0 || (gb->flags & MMIF_T_FOLDED), "pairconv_dgrad: a halo-1 gradient must be folded first (mmif_fold_halo)");
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
    const int tiles_x = cdiv(tgxa.ws, PT), tiles_y = cdiv(tgxa.hs, PT);
    const long long ntiles = (long long)tgxa.n * tgxa.cb * tiles_x * tiles_y;
    MMIF_REQUIRE(ntiles < (1ll << 31), "pairconv_dgrad: too many tiles");
    PAIR_LAUNCH(ga->dtype, nout, pairconv_dgrad_kernel, (unsigned)ntiles, tga, tgb, w, txa, txb, tgxa, tgxb, (unsigned long long)mask_bits, tadd,
                add ? 1 : 0, tiles_x, tiles_y);
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
    const int tiles_x = cdiv(txa.w, PT), tiles_y = cdiv(txa.h, PT);
    const long long ntiles = (long long)txa.n * txa.cb * tiles_x * tiles_y;
    const int G = (int)(ntiles < PAIR_WG_BLOCKS ? ntiles : PAIR_WG_BLOCKS);
    float* partial = (float*)workspace;
    MMIF_REQUIRE(ntiles < (1ll << 31), "pairconv_wgrad: too many tiles");
    PAIR_LAUNCH(xa->dtype, nout, pairconv_wgrad_kernel, G, txa, txb, tga, tgb, partial, tiles_x, tiles_y, (unsigned)ntiles);
    hipLaunchKernelGGL(pairconv_wgrad_reduce, dim3(nout * 19), dim3(256), 0, st, partial, G, nout * 19, nout, dw, db, accumulate);
    return check_launch("pairconv_wgrad");
}

// dgrad + wgrad of one pair-conv layer in one pass (pairconv_bwd_kernel): the arguments of mmif_pairconv_dgrad followed by the
// outputs of mmif_pairconv_wgrad.  xa/xb are always required here (they are the weight gradient's operand).
extern "C" int mmif_pairconv_bwd(const mmif_tensor* ga, const mmif_tensor* gb, const float* w, int32_t nout, const mmif_tensor* xa,
                                 const mmif_tensor* xb, const mmif_tensor* gxa, const mmif_tensor* gxb, uint64_t mask_bits,
                                 const mmif_tensor* add, float* dw, float* db, int32_t accumulate, void* workspace, size_t workspace_bytes,
                                 void* stream) {
    if (int rc = validate_tensor(ga, "ga")) return rc;
    if (int rc = validate_tensor(xa, "xa")) return rc;
    if (int rc = validate_tensor(xb, "xb")) return rc;
    if (int rc = validate_tensor(gxa, "gxa")) return rc;
    if (int rc = validate_tensor(gxb, "gxb")) return rc;
    MMIF_REQUIRE(nout == 1 || nout == 2, "pairconv_bwd: nout must be 1 or 2 (got %d)", nout);
    MMIF_REQUIRE(w != nullptr && dw != nullptr && db != nullptr, "pairconv_bwd: w/dw/db is NULL");
    MMIF_REQUIRE(same_shape(ga, gxa) && same_shape(ga, gxb) && gxa->halo == 1 && gxb->halo == 1, "pairconv_bwd: gx must be halo-1 views of g's shape");
    MMIF_REQUIRE(same_shape(ga, xa) && same_shape(ga, xb) && xa->halo == 0 && xb->halo == 0, "pairconv_bwd: xa/xb shape mismatch");
    MMIF_REQUIRE(ga->h >= 2 && ga->w >= 2, "pairconv_bwd: reflect padding needs h, w >= 2");
    MMIF_REQUIRE(ga->halo == 0 || (ga->flags & MMIF_T_FOLDED), "pairconv_bwd: a halo-1 gradient must be folded first (mmif_fold_halo)");
    if (nout == 2) {
        MMIF_REQUIRE(gb != nullptr, "pairconv_bwd: gb is NULL with nout = 2");
        if (int rc = validate_tensor(gb, "gb")) return rc;
        MMIF_REQUIRE(same_shape(ga, gb), "pairconv_bwd: gb shape mismatch");
        MMIF_REQUIRE(gb->halo == 0 || (gb->flags & MMIF_T_FOLDED), "pairconv_bwd: a halo-1 gradient must be folded first (mmif_fold_halo)");
    }
    if (add) {
        if (int rc = validate_tensor(add, "add")) return rc;
        MMIF_REQUIRE(same_shape(ga, add), "pairconv_bwd: add shape mismatch");
        MMIF_REQUIRE(add->halo == 0 || (add->flags & MMIF_T_FOLDED), "pairconv_bwd: a halo-1 residual gradient must be folded first (mmif_fold_halo)");
    }
    MMIF_REQUIRE(nout == 1 || gb->halo == ga->halo, "pairconv_bwd: ga and gb must have the same halo");
    MMIF_REQUIRE((long long)(ga->h + 2) * (ga->w + 2) * 32 < (1ll << 31), "pairconv_bwd: plane too large for 32-bit in-plane offsets");
    if (workspace == nullptr || workspace_bytes < mmif_pairconv_wgrad_workspace()) {
        set_error("pairconv_bwd: workspace too small");
        return MMIF_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    TV tga = make_tv(ga), tgb = make_tv(nout == 2 ? gb : ga), txa = make_tv(xa), txb = make_tv(xb);
    TV tgxa = make_tv(gxa), tgxb = make_tv(gxb), tadd = make_tv(add ? add : ga);
    const int tiles_x = cdiv(tgxa.ws, PT), tiles_y = cdiv(tgxa.hs, PT);
    const long long ntiles = (long long)tgxa.n * tgxa.cb * tiles_x * tiles_y;
    MMIF_REQUIRE(ntiles < (1ll << 31), "pairconv_bwd: too many tiles");
    const int G = (int)(ntiles < PAIR_BWD_BLOCKS ? ntiles : PAIR_BWD_BLOCKS);
    float* partial = (float*)workspace;
    PAIR_LAUNCH(ga->dtype, nout, pairconv_bwd_kernel, G, tga, tgb, w, txa, txb, tgxa, tgxb, (unsigned long long)mask_bits, tadd, add ? 1 : 0,
                partial, tiles_x, tiles_y, (unsigned)ntiles);
    hipLaunchKernelGGL(pairconv_wgrad_reduce, dim3(nout * 19), dim3(256), 0, st, partial, G, nout * 19, nout, dw, db, accumulate);
    return check_launch("pairconv_bwd");
}
