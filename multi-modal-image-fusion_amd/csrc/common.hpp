// Shared device/host helpers for the gfx950 image-fusion kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mmif.h"

namespace mmif {

constexpr int CBS = 8;  // channels per block (one granule = 8 channels of one pixel)

// ---------------------------------------------------------------- error plumbing
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define MMIF_REQUIRE(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            mmif::set_error(__VA_ARGS__); \
            return MMIF_EINVAL;          \
        }                                \
    } while (0)

// timing-ablation switches ($MMIF_ABLATE, tools/sweep_*.sh): a non-zero value makes the kernels skip loads / stores / k-loops, i.e. the
// results are WRONG -- say so loudly, once, so that a stray variable in a training environment cannot pass unnoticed
// ONE variable for all of them: MMIF_ABLATE="conv=5,x3=8" (keys: conv = conv_dma_kernel, x3 = the split-operand kernels, bp = bwd_pair_dma_kernel,
// ec = enc_chain_bwd_kernel; the bit meanings are documented at the kernels)
inline int ablate_env(const char* key) {
    const char* e = getenv("MMIF_ABLATE");
    if (e == nullptr) return 0;
    const size_t kl = strlen(key);
    int v = 0;
    for (const char* p = e; *p != 0;) {
        if (strncmp(p, key, kl) == 0 && p[kl] == '=') v = atoi(p + kl + 1);
        while (*p != 0 && *p != ',') ++p;
        if (*p == ',') ++p;
    }
    if (v != 0) fprintf(stderr, "mmif: WARNING: MMIF_ABLATE %s=%d is a timing-ablation mode -- kernel results are WRONG (diagnostics only)\n", key, v);
    return v;
}

// ---------------------------------------------------------------- bf16 <-> f32 (round to nearest even)
typedef uint16_t bf16_t;

__host__ __device__ inline float bf16_to_f32(bf16_t v) {
    union { uint32_t u; float f; } c;
    c.u = ((uint32_t)v) << 16;
    return c.f;
}
__host__ __device__ inline bf16_t f32_to_bf16(float f) {
    union { uint32_t u; float f; } c;
    c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// device fast path: v_cvt_pk_bf16_f32 (round to nearest even), two values per instruction
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ inline uint32_t pack_bf16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(uint32_t, r);
}

// ---------------------------------------------------------------- granule load/store (8 channels)
template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int dtype = MMIF_F32;
    static constexpr int gran_bytes = 32;
    __device__ static inline void load(const void* p, float (&v)[8]) {
        const float4* q = reinterpret_cast<const float4*>(p);
        float4 a = q[0], b = q[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    __device__ static inline void store(void* p, const float (&v)[8]) {
        float4* q = reinterpret_cast<float4*>(p);
        q[0] = make_float4(v[0], v[1], v[2], v[3]);
        q[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
    __device__ static inline float quant(float f) { return f; }
};
template <> struct Elem<bf16_t> {
    static constexpr int dtype = MMIF_BF16;
    static constexpr int gran_bytes = 16;
    __device__ static inline void load(const void* p, float (&v)[8]) {
        uint4 a = *reinterpret_cast<const uint4*>(p);
        uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    __device__ static inline void store(void* p, const float (&v)[8]) {
        *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                                  pack_bf16x2(v[6], v[7]));
    }
    __device__ static inline float quant(float f) { return __uint_as_float(pack_bf16x2(f, 0.f) << 16); }
};

// ---------------------------------------------------------------- device-side tensor view
struct TV {
    char* base;            // allocation base
    int n, h, w, halo;     // logical extent
    int hs, ws;            // stored extent (h + 2*halo, w + 2*halo)
    int cb_total, cb_off, cb;
    int folded;            // halo already folded + zeroed
    long long plane;       // granules per channel-block plane = hs*ws
    long long img;         // granules per image = cb_total*plane

    // granule index of (n, view-channel-block c, stored row ys, stored col xs)
    __host__ __device__ inline long long gidx(int in, int c, int ys, int xs) const {
        return (long long)in * img + (long long)(cb_off + c) * plane + (long long)ys * ws + xs;
    }
};

inline TV make_tv(const mmif_tensor* t) {
    TV v;
    v.base = (char*)t->data;
    v.n = t->n; v.h = t->h; v.w = t->w; v.halo = t->halo;
    v.hs = t->h + 2 * t->halo; v.ws = t->w + 2 * t->halo;
    v.cb_total = t->cb_total; v.cb_off = t->cb_off; v.cb = t->cb;
    v.folded = (t->flags & MMIF_T_FOLDED) ? 1 : 0;
    v.plane = (long long)v.hs * v.ws;
    v.img = (long long)v.cb_total * v.plane;
    return v;
}

int validate_tensor(const mmif_tensor* t, const char* name);
// compute units of the current device, cached per device (hipGetDeviceProperties fills a multi-KB struct and can reach the driver: not on the
// per-launch path -- ADVICE r5); $MMIF_NUM_CUS overrides (experiments: persistent grids on part of the chip)
int cached_num_cus();

// reflect index R(t, L) of F.pad(mode='reflect') (edge pixel not repeated); L >= 2 for |offset| 1
__host__ __device__ inline int reflect_idx(int t, int L) {
    if (t < 0) t = -t;
    if (t >= L) t = 2 * (L - 1) - t;
    return t;
}

// Load one granule of an ACTIVATION (halo 0) at logical (y, x) with reflect padding; coordinates
// may lie up to one pixel outside.  Result in fp32.
template <typename T>
__device__ inline void load_act_reflect(const TV& t, int in, int c, int y, int x, float (&v)[8]) {
    y = min(max(reflect_idx(y, t.h), 0), t.h - 1);  // clamp only matters for tile overhang lanes
    x = min(max(reflect_idx(x, t.w), 0), t.w - 1);
    Elem<T>::load(t.base + t.gidx(in, c, y, x) * Elem<T>::gran_bytes, v);
}

// Load one granule of a GRADIENT at logical (y, x): zero outside [0,h)x[0,w); for halo-1
// (padded-domain) tensors the halo is folded onto rows/cols 1 and h-2 / w-2 while loading
// (adjoint of reflect padding by 1).
template <typename T>
__device__ inline void load_grad_fold(const TV& t, int in, int c, int y, int x, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
    if (y < 0 || y >= t.h || x < 0 || x >= t.w) return;
    if (t.halo == 0 || t.folded) {
        Elem<T>::load(t.base + t.gidx(in, c, y + t.halo, x + t.halo) * Elem<T>::gran_bytes, v);
        return;
    }
    // candidate sources: the interior pixel plus the mirrored halo rows / cols (-1 = absent)
    const int ya = y + 1, yb = (y == 1) ? 0 : -1, yc = (y == t.h - 2) ? t.h + 1 : -1;
    const int xa = x + 1, xb = (x == 1) ? 0 : -1, xc = (x == t.w - 2) ? t.w + 1 : -1;
    auto add = [&](int ys, int xs) {
        if (ys < 0 || xs < 0) return;
        float u[8];
        Elem<T>::load(t.base + t.gidx(in, c, ys, xs) * Elem<T>::gran_bytes, u);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += u[i];
    };
    add(ya, xa); add(ya, xb); add(ya, xc);
    add(yb, xa); add(yb, xb); add(yb, xc);
    add(yc, xa); add(yc, xb); add(yc, xc);
}

// block-wide sum of one float per thread (blockDim.x multiple of 64, <= 1024); result valid in
// thread 0.  Deterministic for a fixed block size.
// XCD-aware walk of `total` tiles by G persistent blocks (block b runs on XCD b % 8, each XCD has its own L2): XCD x owns the contiguous
// band [x total/8, (x+1) total/8) and its G/8 blocks walk it side by side, so the halos that neighbouring tiles share are hits in that
// XCD's L2.  (A plain b, b + G, b + 2G walk puts neighbouring tiles on different XCDs: the 18 x 18-tile kernels fetched 1.7x their
// algorithmic bytes through the fabric, `profiles/r02_pmc_tcc_*`.)  Tiles of block b: first + i * stride, i < count.
struct TileWalk {
    int first, stride, count;
};
__device__ inline TileWalk xcd_walk(int total, int G, int b) {
    TileWalk t;
    if ((G & 7) == 0) {
        const int xcd = b & 7, slot = b >> 3, nsl = G >> 3;
        const int q8 = total >> 3, r8 = total & 7;
        const int band0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
        const int blen = q8 + (xcd < r8 ? 1 : 0);
        t.first = band0 + slot;
        t.stride = nsl;
        t.count = blen > slot ? (blen - slot + nsl - 1) / nsl : 0;
    } else {
        t.first = b;
        t.stride = G;
        t.count = b < total ? (total - b + G - 1) / G : 0;
    }
    return t;
}

// one tile per block: the tile of block b when every XCD (b % 8) takes a contiguous run of the nb tiles, so that two tiles sharing a
// cache line (rows of a halo-1 tensor are not line aligned at tile borders) are written through the same L2
// (tools/ubench/row_align.hip: 1.9 -> 4.7 TB/s for that pattern)
__device__ inline unsigned xcd_tile(unsigned b, unsigned nb) {
    const unsigned q8 = nb >> 3, r8 = nb & 7u, xcd = b & 7u;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
}

__device__ inline float block_sum(float v, float* smem /* >= 16 floats */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) r += smem[i];
    }
    return r;
}

// Fixed-order sum of the G per-block partials of one output element (the partials sit `stride` floats apart), as the reduce kernels of the
// weight gradients need it: a block = 64 outputs x RED_SLICES slices of the G range (1024 threads), 4 independent load chains per thread,
// i.e. 64 loads of one output in flight and G / 64 dependent round trips (the partials were written by other XCDs: ~1-2 us each; with 4
// slices -- 256 threads -- a G = 256 reduce ran 16 of them: 8-14 us per launch, seven launches per train step).  Returns the sum in the
// slice-0 threads; every thread of the block must call it.  SL = 4 (256 threads) for short partial lists (G <= 64: the 64-channel-pair
// weight gradients), where 16 slices only add waves.
constexpr int RED_SLICES = 16;
template <int SL = RED_SLICES>
__device__ inline float partial_sum(const float* __restrict__ partial, long long off, long long stride, int G, bool valid, float (*red)[64]) {
    const int o_local = threadIdx.x & 63, slice = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (valid) {
        int g = slice;
        for (; g + 3 * SL < G; g += 4 * SL) {
            s0 += partial[g * stride + off];
            s1 += partial[(g + SL) * stride + off];
            s2 += partial[(g + 2 * SL) * stride + off];
            s3 += partial[(g + 3 * SL) * stride + off];
        }
        for (; g < G; g += SL) s0 += partial[g * stride + off];
    }
    red[slice][o_local] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    float t = 0.f;
    if (slice == 0) {
#pragma unroll
        for (int q = 0; q < SL; q += 4) t += (red[q][o_local] + red[q + 1][o_local]) + (red[q + 2][o_local] + red[q + 3][o_local]);
    }
    return t;
}

// linear index -> (n, c, y, x) of a [n][c][h][w] iteration space; 32-bit arithmetic whenever the index fits (it always does for
// the shapes on this path) instead of four 64-bit divisions per element.  (Measured: no effect on the glue kernels' time -- they
// stream at 5+ TB/s either way -- kept because it is the cheaper code.)
__device__ inline void split_idx(long long i, int cdim, int h, int w, int& n, int& c, int& y, int& x) {
    if (i < 0xffffffffll) {
        unsigned u = (unsigned)i;
        const unsigned q1 = u / (unsigned)w;
        x = (int)(u - q1 * (unsigned)w);
        const unsigned q2 = q1 / (unsigned)h;
        y = (int)(q1 - q2 * (unsigned)h);
        const unsigned q3 = q2 / (unsigned)cdim;
        c = (int)(q2 - q3 * (unsigned)cdim);
        n = (int)q3;
    } else {
        x = (int)(i % w);
        y = (int)((i / w) % h);
        c = (int)((i / ((long long)w * h)) % cdim);
        n = (int)(i / ((long long)w * h * cdim));
    }
}

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace mmif
