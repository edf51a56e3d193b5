// bf16 MFMA convolution kernels for gfx950 (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
//
// Implicit GEMM, one formulation for forward and dgrad (see conv_valu.hip for the maths):
//     D[oc][pixel] = sum_{tap, ic} Wk[tap][ic][oc] * IN(pixel + tap - p)[ic]
//   M = output channels (A operand = packed weights), N = 16 consecutive pixels of one row
//   (B operand = input tile), K = (tap, 8-channel block) "k-groups": lane group g = lane>>4 of a
//   k-step handles k-group 4*step+g, i.e. ONE 16-byte granule per lane per operand.
// Data layout: HBM tensors are blocked NHWC ([n][C/8][H][W][8]); the LDS tiles keep the same
//   granule structure ([channel block][tile row][tile col][8 ch], plane stride = 0 mod 256 B), so
//   staging is a pure 16-byte-granule copy with reflect / zero-fill / halo-fold applied to the
//   SOURCE index only, and every ds_read_b128 operand fetch is bank-conflict free (a lane group's
//   two k-groups read complementary pixel sets of planes that alias the same banks).
// Block = 4 waves, output tile 16x16 pixels x (MF*16) channels; wave w owns rows 4w..4w+3.
// K is consumed in chunks of 4 channel blocks (32 channels x k*k taps): per chunk the input halo
//   tile (18x18 granules per block) and the packed weight slab are staged to LDS, then
//   nsteps = ceil(k*k*ncb/4) MFMA k-steps run out of LDS.  Two blocks per CU overlap each
//   other's staging and MFMA phases.
//
// wgrad (dW = sum_pixels g (x) xpad) has K = pixels: both operands are "k-major" in the blocked
// layout, so fragments are fetched with the gfx950 LDS transpose read ds_read_b64_tr_b16
// (4 pixels x 16 channels per 16-lane group -> per lane 4 consecutive pixels of one channel).
#include "common.hpp"
#include "reduce_defer.hpp"
// the epilogues' 16-byte output stores.  (Round 5 tried them as inline-asm `global_store_dwordx4 ... sc0`, after the sc0 bit had made the
// streaming encoder's buffer stores 5 % faster: no change in the step once the asm was correct -- a first version without the two wait
// states a > 8-byte store needs before its data registers are rewritten stored garbage, and the step ran 12 % "faster" on the garbage
// (less switching in the matrix pipes, higher clocks): tests/test_gpu_bwd_pair.py caught it.  Plain stores: the compiler pads its own.)
#define MMIF_STORE_GRAN(p, v) (*reinterpret_cast<uint4*>(p) = (v))
#include <stdlib.h>

namespace mmif {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

constexpr int MT = 16;        // output tile edge
constexpr int DT_ROWS_C = 32; // rows of a conv_dma_kernel tile (= DT_ROWS below)
constexpr int CHUNK_CB = 4;   // channel blocks per K chunk

__host__ __device__ constexpr int plane_granules(int ks) { return ks == 3 ? 336 : 256; }  // bytes = 0 mod 256

static inline int pick_mf(int n_out) {
    const int fr = (n_out + 15) / 16;
    return fr <= 4 ? fr : 4;
}
// Order in which every 3x3 kernel of this file visits the taps: COLUMN-major -- (u, v) = (0,0), (1,0), (2,0), (0,1), ... -- so that
// the three steps of one column offset v read the same six tile rows (conv_dma_kernel keeps them in registers: 6 instead of 12
// operand fragments per column).  All kernels share the order, hence the summation order, hence stay bit-identical to each other.
__host__ __device__ inline int visit_tap(int i, int ks) { return ks == 3 ? (i % 3) * 3 + i / 3 : i; }

static inline int n_mblocks(int n_out) {
    const int fr = (n_out + 15) / 16, mf = pick_mf(n_out);
    return (fr + mf - 1) / mf;
}
// packed image: for each chunk, nkg_pad(chunk) k-group planes of [M16p][8] bf16, M16p = n_mblocks*MF*16
static size_t packed_bytes(int n_out, int n_in, int ks) {
    const int ncb = (n_in + 7) / 8, kk = ks * ks;
    const size_t m16p = (size_t)n_mblocks(n_out) * pick_mf(n_out) * 16;
    size_t kg = 0;
    for (int c0 = 0; c0 < ncb; c0 += CHUNK_CB) {
        const int n = ncb - c0 < CHUNK_CB ? ncb - c0 : CHUNK_CB;
        kg += (size_t)((kk * n + 3) / 4) * 4;
    }
    return kg * m16p * 16;
}

// ------------------------------------------------------------------ weight packing
// dgrad == 0: out channel = o, in channel = c, Wk[u][v] = W[o][c][u][v]
// dgrad == 1: out channel = c, in channel = o, Wk[u][v] = W[o][c][k-1-u][k-1-v]
__global__ void pack_weights_kernel(const float* __restrict__ w, int cout, int cin, int ks, int dgrad, int mf,
                                    int m16p, bf16_t* __restrict__ dst, long long total) {
    const int kk = ks * ks;
    const int n_out = dgrad ? cin : cout, n_in = dgrad ? cout : cin;
    const int ncb = (n_in + 7) / 8;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int e = idx & 7;
        const long long row = idx >> 3;             // (global k-group plane, oc row)
        const int oc = (int)(row % m16p);
        long long kgp = row / m16p;                 // padded k-group index over all chunks
        // locate the chunk
        int c0 = 0, n = 0, kg = 0;
        for (c0 = 0; c0 < ncb; c0 += CHUNK_CB) {
            n = ncb - c0 < CHUNK_CB ? ncb - c0 : CHUNK_CB;
            const int pad = ((kk * n + 3) / 4) * 4;
            if (kgp < pad) { kg = (int)kgp; break; }
            kgp -= pad;
        }
        float val = 0.f;
        if (kg < kk * n) {
            const int tap = kg / n, cb = kg % n;
            const int u = tap / ks, v = tap % ks;
            const int ic = (c0 + cb) * 8 + e;
            if (oc < n_out && ic < n_in) {
                if (dgrad) val = w[(((long long)ic * cin + oc) * ks + (ks - 1 - u)) * ks + (ks - 1 - v)];
                else val = w[(((long long)oc * cin + ic) * ks + u) * ks + v];
            }
        }
        dst[idx] = f32_to_bf16(val);
    }
    (void)mf;
}

// All of a model's operand images in ONE launch (20 launches of a few microseconds each per PFNetv1 step otherwise):
// blockIdx.y = image, the table travels as a kernel argument.
constexpr int PACK_MAX_IMAGES = 64;
struct PackImage { const float* w; bf16_t* dst; long long total; int cout, cin, ks, dgrad, m16p; };
struct PackTable { PackImage im[PACK_MAX_IMAGES]; };

// one GRANULE (8 input channels of one (k-group plane, output row)) per thread: the chunk / tap decode is done once per granule (round 6:
// with one element per thread NestFuse's 5.4 M elements took 71 us per step, index arithmetic bound)
__device__ inline void pack_gran(const PackImage& J, long long row) {
    const int ks = J.ks, kk = ks * ks, cin = J.cin, m16p = J.m16p;
    const int n_out = J.dgrad ? cin : J.cout, n_in = J.dgrad ? J.cout : cin;
    const int ncb = (n_in + 7) / 8;
    const int oc = (int)(row % m16p);
    long long kgp = row / m16p;
    int c0 = 0, n = 0, kg = 0;
    for (c0 = 0; c0 < ncb; c0 += CHUNK_CB) {
        n = ncb - c0 < CHUNK_CB ? ncb - c0 : CHUNK_CB;
        const int pad = ((kk * n + 3) / 4) * 4;
        if (kgp < pad) { kg = (int)kgp; break; }
        kgp -= pad;
    }
    float val[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) val[e] = 0.f;
    if (kg < kk * n && oc < n_out) {
        const int tap = kg / n, cb = kg % n;
        const int u = tap / ks, v = tap % ks;
        const int ic0 = (c0 + cb) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ic = ic0 + e;
            if (ic < n_in) {
                if (J.dgrad) val[e] = J.w[(((long long)ic * cin + oc) * ks + (ks - 1 - u)) * ks + (ks - 1 - v)];
                else val[e] = J.w[(((long long)oc * cin + ic) * ks + u) * ks + v];
            }
        }
    }
    *reinterpret_cast<uint4*>(J.dst + row * 8) = make_uint4(pack_bf16x2(val[0], val[1]), pack_bf16x2(val[2], val[3]), pack_bf16x2(val[4], val[5]), pack_bf16x2(val[6], val[7]));
}

__global__ void pack_weights_multi_kernel(PackTable tab) {
    const PackImage& J = tab.im[blockIdx.y];
    const long long rows = J.total >> 3;
    for (long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += (long long)gridDim.x * blockDim.x)
        pack_gran(J, row);
}

// ------------------------------------------------------------------ granule loaders (bf16, raw uint4)
__device__ inline uint4 ld_gran(const TV& t, int in_, int c, int ys, int xs) {
    return *reinterpret_cast<const uint4*>(t.base + t.gidx(in_, c, ys, xs) * 16);
}
__device__ inline uint4 load_in_reflect(const TV& t, int in_, int c, int y, int x) {
    y = min(max(reflect_idx(y, t.h), 0), t.h - 1);
    x = min(max(reflect_idx(x, t.w), 0), t.w - 1);
    return ld_gran(t, in_, c, y, x);
}
__device__ inline uint4 load_in_gradfold(const TV& t, int in_, int c, int y, int x) {
    if (y < 0 || y >= t.h || x < 0 || x >= t.w) return make_uint4(0, 0, 0, 0);
    if (t.halo == 0 || t.folded) return ld_gran(t, in_, c, y + t.halo, x + t.halo);
    const bool by = (y == 1) || (y == t.h - 2), bx = (x == 1) || (x == t.w - 2);
    if (!by && !bx) return ld_gran(t, in_, c, y + 1, x + 1);
    float v[8];
    load_grad_fold<bf16_t>(t, in_, c, y, x, v);  // fp32 fold, rounded once
    return make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}

// Round 6 (DenseFuse, `core/model.py:165-186`): the gradient of f1 + f2 IS the gradient of either encoder's output, but each encoder's last
// DenseBlock conv (channel blocks 6, 7 of its 8) still owes it the ReLU mask of its own x3.  A dgrad can leave those two masked COPIES itself:
// fragment `frag` (16 channels = blocks 2 frag, 2 frag + 1 of the gradient it computes) is also written to out's blocks [2 frag, + 1] masked
// by mask's blocks [2 frag, + 1], and to out's blocks [2 frag + 8, + 1] masked by mask's [2 frag + 8, + 1] (out: halo-1 tensor of gx's
// geometry, mask: the halo-0 activations [x(img1) | x(img2)]) -- mmif_fuse_elem_bwd's pass disappears.  frag < 0: off.
struct DupOut {
    TV out, mask;
    int frag;
};

// ------------------------------------------------------------------ shared epilogue
// After the MFMAs lane (g, j) holds oc = 16*m + 4*g + r (r = 0..3) of pixels (row 4*wave + n, col j).
// v_permlane16_swap pairs rows n, n+1: lanes with even g end up with all 8 channels of row n, odd g with all
// 8 channels of row n+1, so every lane issues ONE 16-byte store per row pair (the 8-byte version was store-issue
// bound: 16 stores ~ 10k cycles per block).  oxs / oys0: stored column / first stored row (of the lane's row pair 0).
// DUP (dgrad, no LM): the instantiation that writes struct DupOut's masked copies; its own output is neither masked nor accumulated
template <int MF, bool DGRAD, bool LM = false, bool DUP = false>
__device__ inline void conv_epilogue(const f32x4 (&acc)[MF][4], const TV& tout, const TV& tmask, const float* s_bias, int mb, int in_,
                                     int oxs, int oys0, int g, int relu, unsigned long long mask_bits, unsigned long long accum_bits,
                                     int xlim, int ylim,   // stored columns / rows >= xlim / ylim are not written
                                     const unsigned char* lmask = nullptr, int lrow0 = 0, int lcol = 0,
                                     const TV* told = nullptr /* dgrad: accumulate ONTO this tensor's values instead of tout's (same n, h, w, halo) */,
                                     const DupOut* dup = nullptr /* dgrad without LM: masked copies of one fragment, see struct DupOut */) {
    // lmask (conv_dma_kernel dgrad): the ReLU-mask SIGN BITS of this tile, one byte per (channel block, tile row, tile column),
    // staged into LDS by the loader waves ahead of time; lrow0 / lcol = this lane's first tile row / its column
    if (oxs >= xlim) return;
    const unsigned row_bytes = (unsigned)tout.ws * 16u;
    const unsigned pix_off = (unsigned)(oys0 * tout.ws + oxs) * 16u;   // inside one plane
    // dgrad: fetch the old gradient / the ReLU-mask activations of ALL the lane's outputs first -- one memory round trip for
    // the whole epilogue instead of one per 16-channel fragment (and the stores below may alias them as far as the compiler
    // can tell, which would serialise load -> store -> load)
    uint4 oldv[(DGRAD && !DUP) ? MF : 1][2], xmv[(DGRAD && !LM && !DUP) ? MF : 1][2];
    unsigned lbits[(DGRAD && LM) ? MF : 1][2];
    if (DGRAD && LM) {
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) lbits[LM ? m : 0][p2] = lmask[((2 * m + (g >> 1)) * DT_ROWS_C + lrow0 + 2 * p2) * MT + lcol];
    }
    if (DGRAD && !DUP) {
        const unsigned mpix_off = (unsigned)min(max(reflect_idx(oxs - tout.halo, tmask.w), 0), tmask.w - 1) * 16u;
#pragma unroll
        for (int m = 0; m < MF; ++m) {
            const int ocb = (mb * MF + m) * 2 + (g >> 1);
            const bool blk_ok = ocb < tout.cb;
            const char* oplane = tout.base + ((long long)in_ * tout.img + (long long)(tout.cb_off + min(ocb, tout.cb - 1)) * tout.plane) * 16 + pix_off;
            if (told != nullptr)
                oplane = told->base + ((long long)in_ * told->img + (long long)(told->cb_off + min(ocb, told->cb - 1)) * told->plane) * 16 +
                         (unsigned)(oys0 * told->ws + oxs) * 16u;
            const unsigned old_row_bytes = told != nullptr ? (unsigned)told->ws * 16u : row_bytes;
            const char* mplane = tmask.base + ((long long)in_ * tmask.img + (long long)(tmask.cb_off + min(ocb, tmask.cb - 1)) * tmask.plane) * 16 + mpix_off;
            const bool do_acc = (accum_bits >> ocb) & 1ull, do_mask = (mask_bits >> ocb) & 1ull;
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                oldv[m][p2] = make_uint4(0, 0, 0, 0);
                if (!LM) xmv[m][p2] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);   // 1.0: mask passes
                const int oys = oys0 + 2 * p2;
                if (blk_ok && oys < ylim) {
                    // (global address space spelled out: with the base selected between two tensors this was a FLAT load, which counts
                    // in lgkmcnt as well and made the waits around the LDS traffic wait for HBM)
                    if (do_acc) {
                        typedef unsigned gu32x4 __attribute__((ext_vector_type(4)));
                        const gu32x4 ov = *reinterpret_cast<const __attribute__((address_space(1))) gu32x4*>((unsigned long long)(oplane + (2 * p2) * old_row_bytes));
                        oldv[m][p2] = make_uint4(ov.x, ov.y, ov.z, ov.w);
                    }
                    if (do_mask && !LM) {
                        const int y = min(max(reflect_idx(oys - tout.halo, tmask.h), 0), tmask.h - 1);
                        xmv[LM ? 0 : m][p2] = *reinterpret_cast<const uint4*>(mplane + (unsigned)(y * tmask.ws) * 16u);
                    }
                }
            }
        }
    }
    // masked copies of one fragment (struct DupOut): the two mask granules of both row pairs are requested with the epilogue's other operands
    uint4 dxa[2], dxb[2];
    int dm = -1;
    if (DUP && dup->frag >= mb * MF && dup->frag < mb * MF + MF) {
        dm = dup->frag - mb * MF;
        const int ocb_d = dup->frag * 2 + (g >> 1);
        const TV& tk = dup->mask;
        const unsigned kx = (unsigned)min(max(reflect_idx(oxs - tout.halo, tk.w), 0), tk.w - 1) * 16u;
        const char* pa = tk.base + ((long long)in_ * tk.img + (long long)(tk.cb_off + ocb_d) * tk.plane) * 16 + kx;
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {
            const int y = min(max(reflect_idx(oys0 + 2 * p2 - tout.halo, tk.h), 0), tk.h - 1);
            dxa[p2] = *reinterpret_cast<const uint4*>(pa + (unsigned)(y * tk.ws) * 16u);
            dxb[p2] = *reinterpret_cast<const uint4*>(pa + (unsigned)(y * tk.ws) * 16u + 8ll * tk.plane * 16);
        }
    }
#pragma unroll
    for (int m = 0; m < MF; ++m) {
        const int ocb = (mb * MF + m) * 2 + (g >> 1);  // channel block inside the out view
        const bool blk_ok = ocb < tout.cb;
        float bv[8];
        if (!DGRAD) {
            const float4 b0 = *reinterpret_cast<const float4*>(&s_bias[m * 16 + (g >> 1) * 8]);
            const float4 b1 = *reinterpret_cast<const float4*>(&s_bias[m * 16 + (g >> 1) * 8 + 4]);
            bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
        }
        char* oplane = tout.base + ((long long)in_ * tout.img + (long long)(tout.cb_off + min(ocb, tout.cb - 1)) * tout.plane) * 16 + pix_off;
        // both row pairs are finished before the first is stored: a store between the two made the compiler wait for IT (vmcnt(0))
        // before it touched the second pair's pre-loaded operands -- one HBM write latency inside every dgrad epilogue
        uint4 outv[2];
        bool outok[2] = {false, false};
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {
            float c[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[m][2 * p2][r]), __float_as_uint(acc[m][2 * p2 + 1][r]), false, false);
                c[r] = __uint_as_float(sw[0]);
                c[4 + r] = __uint_as_float(sw[1]);
            }
            const int oys = oys0 + 2 * p2;
            if (!blk_ok || oys >= ylim) continue;
            if (!DGRAD) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float t = c[i] + bv[i];
                    c[i] = relu ? fmaxf(t, 0.f) : t;
                }
            } else if constexpr (!DUP) {
                const uint32_t ow[4] = {oldv[m][p2].x, oldv[m][p2].y, oldv[m][p2].z, oldv[m][p2].w};
                const uint32_t xw[4] = {xmv[LM ? 0 : m][p2].x, xmv[LM ? 0 : m][p2].y, xmv[LM ? 0 : m][p2].z, xmv[LM ? 0 : m][p2].w};
                if (LM) {
                    const unsigned bits = ((mask_bits >> ocb) & 1ull) ? lbits[LM ? m : 0][p2] : 0xffu;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        c[2 * i] += __uint_as_float(ow[i] << 16);
                        c[2 * i + 1] += __uint_as_float(ow[i] & 0xffff0000u);
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (!((bits >> i) & 1u)) c[i] = 0.f;
                } else
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    c[2 * i] += __uint_as_float(ow[i] << 16);          // zeros when not accumulating
                    c[2 * i + 1] += __uint_as_float(ow[i] & 0xffff0000u);
                    // bf16 > 0  <=>  sign bit clear and magnitude non-zero (activations are never NaN); 1.0 when unmasked
                    const uint32_t lo = xw[i] & 0xffffu, hi = xw[i] >> 16;
                    if (!((lo & 0x8000u) == 0 && (lo & 0x7fffu) != 0)) c[2 * i] = 0.f;
                    if (!((hi & 0x8000u) == 0 && (hi & 0x7fffu) != 0)) c[2 * i + 1] = 0.f;
                }
            }
            outv[p2] = make_uint4(pack_bf16x2(c[0], c[1]), pack_bf16x2(c[2], c[3]), pack_bf16x2(c[4], c[5]), pack_bf16x2(c[6], c[7]));
            outok[p2] = true;
        }
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2)
            if (outok[p2]) MMIF_STORE_GRAN(oplane + (2 * p2) * row_bytes, outv[p2]);
        if (DUP && m == dm) {
            // bf16 > 0 per 16-bit half: min(x, 1) then max(.., 0) as signed 16-bit values gives 1 / 0, times 0xffff the half's mask
            auto keep = [](uint32_t v, uint32_t x) {
                uint32_t q;
                __asm__("v_pk_min_i16 %0, %1, %2\n\tv_pk_max_i16 %0, %0, 0" : "=&v"(q) : "v"(x), "v"(0x00010001u));
                return v & (q * 0xffffu);
            };
            const TV& tdo = dup->out;
            char* da = tdo.base + ((long long)in_ * tdo.img + (long long)(tdo.cb_off + ocb) * tdo.plane) * 16 + pix_off;
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                if (!outok[p2]) continue;
                const uint4 v = outv[p2], xa = dxa[p2], xb = dxb[p2];
                MMIF_STORE_GRAN(da + (2 * p2) * row_bytes, make_uint4(keep(v.x, xa.x), keep(v.y, xa.y), keep(v.z, xa.z), keep(v.w, xa.w)));
                MMIF_STORE_GRAN(da + (2 * p2) * row_bytes + 8ll * tdo.plane * 16, make_uint4(keep(v.x, xb.x), keep(v.y, xb.y), keep(v.z, xb.z), keep(v.w, xb.w)));
            }
        }
    }
}

// The same epilogue with the arithmetic done BEFORE the row swap, on packed pairs (conv_dma_kernel: a wave's epilogue runs
// beside its SIMD partner's MFMAs, and a lone wave only reaches ~65 % of the MFMA rate -- `tools/trace_dma.py`: 3.4-3.7 k
// ticks per epilogue, ~550 wave-wide VALU instructions at 4 cycles each -- so every epilogue instruction is MFMA time):
// bias with v_pk_add_f32, rounding two values per v_cvt_pk_bf16_f32, ReLU as a signed 16-bit v_pk_max_i16 against 0 on
// the rounded pair (bf16 <= -0  <=>  negative int16; against -32768 when the layer has no ReLU), then TWO
// v_permlane16_swap per row pair instead of four, and for the dgrad the loader-staged mask byte expanded to four dword masks.
// Results are bit-identical to conv_epilogue (ReLU commutes with the rounding; the mask zeroes whole bf16 values).
// Forward, and dgrad with LDS mask bits and nothing to accumulate.
typedef short i16x2_t __attribute__((ext_vector_type(2)));
template <int MF, bool DGRAD>
__device__ inline void conv_epilogue_packed(const f32x4 (&acc)[MF][4], const TV& tout, const float* s_bias, int mb, int in_, int ox0, int oy0, int lrow,
                                            int relu, unsigned long long mask_bits, int xlim, int ylim, const unsigned char* lmask = nullptr,
                                            long long* tr = nullptr, int* tr_n = nullptr) {
    // ox0 / oy0: stored column of the tile's first column / stored row of this WAVE's first row; lrow: the wave's first tile row.
    // Everything lane dependent (j = lane & 15, g = lane >> 4) is recomputed here behind an asm the compiler cannot hoist: kept live
    // across the k-loop these were the values conv_dma_kernel spilled (scratch reloads + s_waitcnt vmcnt(0) at the head of an epilogue)
    int lane_;
    __asm__ volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_));
    const int lcol = lane_ & 15, g = lane_ >> 4, gh = lane_ >> 5;
    const int oxs = ox0 + lcol, oys0 = oy0 + (g & 1), lrow0 = lrow + (g & 1);
#ifdef EPI_TRACE
#define ESTAMP() do { if (tr != nullptr && *tr_n < 62) tr[(*tr_n)++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define ESTAMP() do { } while (0)
#endif
    if (oxs >= xlim) return;
    const unsigned row_bytes = (unsigned)tout.ws * 16u;
    const unsigned pix_off = (unsigned)(oys0 * tout.ws + oxs) * 16u;
    const uint32_t floor2 = relu ? 0u : 0x80008000u;
    const i16x2_t floorv = __builtin_bit_cast(i16x2_t, floor2);
    const bool row_ok[2] = {oys0 < ylim, oys0 + 2 < ylim};
    // every LDS operand of the epilogue up front: ONE round trip through an LDS pipe that is busy with the partner wave's k-loop
    float4 bq[DGRAD ? 1 : MF];
    unsigned lb[DGRAD ? MF : 1][2];
#pragma unroll
    for (int m = 0; m < MF; ++m) {
        if (!DGRAD) bq[DGRAD ? 0 : m] = *reinterpret_cast<const float4*>(&s_bias[m * 16 + g * 4]);   // this lane's channels 16 m + 4 g + r
        else {
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) lb[DGRAD ? m : 0][p2] = lmask[((2 * m + gh) * DT_ROWS_C + lrow0 + 2 * p2) * MT + lcol];
        }
    }
#pragma unroll
    for (int m = 0; m < MF; ++m) {
        const int ocb = (mb * MF + m) * 2 + gh;
        const bool blk_ok = ocb < tout.cb;
        f32x2_t b01 = {0.f, 0.f}, b23 = {0.f, 0.f};
        if (!DGRAD) {
            b01 = (f32x2_t){bq[DGRAD ? 0 : m].x, bq[DGRAD ? 0 : m].y};
            b23 = (f32x2_t){bq[DGRAD ? 0 : m].z, bq[DGRAD ? 0 : m].w};
        }
        char* oplane = tout.base + ((long long)in_ * tout.img + (long long)(tout.cb_off + min(ocb, tout.cb - 1)) * tout.plane) * 16 + pix_off;
        const bool do_mask = DGRAD && ((mask_bits >> ocb) & 1ull);
        ESTAMP();
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {
            uint32_t pk[2][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const f32x4 a = acc[m][2 * p2 + e];
                f32x2_t lo = {a[0], a[1]}, hi = {a[2], a[3]};
                if (!DGRAD) { lo += b01; hi += b23; }
                uint32_t w0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2_t));
                uint32_t w1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf16x2_t));
                if (!DGRAD) {
                    w0 = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, w0), floorv));
                    w1 = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, w1), floorv));
                }
                pk[e][0] = w0;
                pk[e][1] = w1;
            }
            // even g: all 8 channels of row 2 p2; odd g: of row 2 p2 + 1  (channels 4 g' .. 4 g' + 7, g' = g & ~1)
            const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
            uint4 o = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            if (DGRAD) {
                const unsigned bits = do_mask ? lb[DGRAD ? m : 0][p2] : 0xffu;
                auto dmask = [&](int i) {   // bits 2i, 2i+1 -> 0x0000ffff / 0xffff0000 halves
                    const uint32_t l = (uint32_t)__builtin_amdgcn_sbfe((int)bits, 2 * i, 1), h = (uint32_t)__builtin_amdgcn_sbfe((int)bits, 2 * i + 1, 1);
                    return (l & 0xffffu) | (h & 0xffff0000u);
                };
                o.x &= dmask(0); o.y &= dmask(1); o.z &= dmask(2); o.w &= dmask(3);
            }
            if (blk_ok && row_ok[p2]) MMIF_STORE_GRAN(oplane + (2 * p2) * row_bytes, o);
        }
    }
    ESTAMP();
#undef ESTAMP
}

// ------------------------------------------------------------------ forward / dgrad kernel
// (min waves per SIMD: the Cout <= 16 instantiations are HBM-bound and want bytes in flight, i.e. occupancy: 128 VGPRs -> 4 blocks/CU)
template <int KS, int MF, bool DGRAD>
__global__ __launch_bounds__(256, MF == 1 ? 4 : 2) void conv_mfma_kernel(TV tin, TV tout, TV tmask, const uint4* __restrict__ wpk,
                                                            const float* __restrict__ bias, int n_out, int m16p,
                                                            int relu, unsigned long long mask_bits,
                                                            unsigned long long accum_bits, int tiles_x, int tiles_y,
                                                            int nmb, long long* __restrict__ trace) {
    constexpr int KK = KS * KS, P = KS / 2;
    constexpr int TP = MT + KS - 1;             // input tile edge (18 / 16)
    constexpr int PL = plane_granules(KS);      // granules per LDS plane
    constexpr int MAXKG = KK * CHUNK_CB;        // 36 / 4
    constexpr int MAXKGP = (MAXKG + 3) / 4 * 4;
    constexpr int NIN = (CHUNK_CB * TP * TP + 255) / 256;   // staged input granules per thread per chunk
    constexpr int NW = (MAXKGP * MF * 16 + 255) / 256;      // staged weight granules per thread per chunk
    __shared__ __attribute__((aligned(16))) uint4 s_in[CHUNK_CB * PL];
    __shared__ __attribute__((aligned(16))) uint4 s_w[MAXKGP * MF * 16];
    __shared__ int2 s_tab[MAXKGP];
    __shared__ __attribute__((aligned(16))) float s_bias[MF * 16];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    // XCD-aware block order (block b runs on XCD b % 8; speed only, any placement is correct): every XCD walks
    // a CONTIGUOUS band of (image, tile row, tile col) with a tile's M-blocks back to back, so the second
    // M-block's input tile and the halos shared with neighbouring tiles are hits in that XCD's L2.
    // optional phase trace (diagnostics, tools/trace_conv.py): wave 0 of the first 1024 blocks stamps s_memtime
    int tr_n = 0;
    long long* tr = (trace != nullptr && blockIdx.x < 1024 && threadIdx.x == 0) ? trace + (long long)blockIdx.x * 64 : nullptr;
#define TRACE_STAMP() do { if (tr != nullptr && tr_n < 64) tr[tr_n++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    TRACE_STAMP();
    const int nb = gridDim.x, b = blockIdx.x;
    const int q8 = nb >> 3, r8 = nb & 7, xcd = b & 7;
    const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    const int mb = lin % nmb, tl = lin / nmb;
    const int tpi = tiles_x * tiles_y;
    const int in_ = tl / tpi, trem = tl % tpi;
    const int tile_x = trem % tiles_x, tile_y = trem / tiles_x;
    const int iy0 = tile_y * MT - tout.halo - P, ix0 = tile_x * MT - tout.halo - P;  // logical origin of the input tile

    // ---- per-thread staging descriptors (chunk independent) ----
    // input element i of this thread: chunk-local channel block icb[i], LDS slot ilds[i], and either a
    // plane-relative granule offset (imode 1), a zero (imode 0) or a halo-fold slow path (imode 2)
    // idesc = LDS slot | chunk-local channel block << 16 | mode << 20; ioff = byte offset from the chunk's first plane
    // (always a valid address); ioff0 = the same pixel in the chunk's plane 0 (used for the ragged last chunk)
    int idesc[NIN];
    unsigned ioff[NIN];
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
        const int e = tid + 256 * i;
        const int cb = e / (TP * TP), p = e % (TP * TP);
        const int py = p / TP, px = p % TP;
        int mode;  // cb >= CHUNK_CB for the tail elements of the last iteration
        int y = iy0 + py, x = ix0 + px;
        if (!DGRAD) {
            y = min(max(reflect_idx(y, tin.h), 0), tin.h - 1);
            x = min(max(reflect_idx(x, tin.w), 0), tin.w - 1);
            mode = 1;
            ioff[i] = (unsigned)(y * tin.ws + x) * 16u;
        } else {
            const bool inside = y >= 0 && y < tin.h && x >= 0 && x < tin.w;
            const bool border = tin.halo && !tin.folded && ((y == 1) || (y == tin.h - 2) || (x == 1) || (x == tin.w - 2));
            mode = !inside ? 0 : (border ? 2 : 1);
            ioff[i] = (unsigned)((min(max(y, 0), tin.h - 1) + tin.halo) * tin.ws + min(max(x, 0), tin.w - 1) + tin.halo) * 16u;
        }
        idesc[i] = (cb * PL + p) | (cb << 16) | (mode << 20);
        ioff[i] += (unsigned)min(cb, CHUNK_CB - 1) * (unsigned)(tin.plane * 16);
    }
    const char* in_img = tin.base + ((long long)in_ * tin.img + (long long)tin.cb_off * tin.plane) * 16;
    // does this block's input tile contain a fold row/col (1 or h-2 / w-2) of a halo-1 gradient?
    const bool fold_tile = DGRAD && tin.halo && !tin.folded &&
                           ((iy0 <= 1 && 1 < iy0 + TP) || (iy0 <= tin.h - 2 && tin.h - 2 < iy0 + TP) ||
                            (ix0 <= 1 && 1 < ix0 + TP) || (ix0 <= tin.w - 2 && tin.w - 2 < ix0 + TP));

    f32x4 acc[MF][4];
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int ncb_tot = tin.cb;
    const char* in_lane = reinterpret_cast<const char*>(s_in) + ((wave * 4) * TP + j) * 16;
    const char* w_lane = reinterpret_cast<const char*>(s_w) + j * 16;

    // byte offset of this thread's weight granule i inside a chunk's packed slab: granule i is (256 / (MF*16)) k-group
    // planes below granule 0 whenever MF*16 divides 256, so ONE lane offset + a scalar stride serves all of them
    constexpr bool WREG = (256 % (MF * 16)) == 0;
    constexpr int NWO = WREG ? 1 : NW;
    unsigned woff[NWO];
#pragma unroll
    for (int i = 0; i < NWO; ++i) {
        const int e = tid + 256 * i;
        woff[i] = (unsigned)((e / (MF * 16)) * m16p + mb * MF * 16 + e % (MF * 16)) * 16u;
    }
    const unsigned wstride = (unsigned)(256 / (MF * 16)) * (unsigned)m16p * 16u;   // uniform
    uint4 rin[NIN], rw[NW];
    // one staged load (compile-time index i after unrolling): input granule i < NIN, else weight granule i - NIN.
    // Unconditional loads from clamped (always valid) addresses, zero by select: nothing serialises them.
    auto issue_load = [&](int i, int c0, long long wchunk_off) {
        const int ncb = min(CHUNK_CB, ncb_tot - c0);
        if (i < NIN) {
            const int cb = (idesc[i] >> 16) & 15, mode = idesc[i] >> 20;
            const char* chunk = in_img + (long long)c0 * tin.plane * 16;          // wave-uniform base (SGPR pair)
            const uint4 v = *reinterpret_cast<const uint4*>(chunk + (cb < ncb ? ioff[i] : ioff[0]));   // + 32-bit lane offset (element 0 is always in plane 0)
            rin[i] = (cb < ncb && mode >= 1) ? v : make_uint4(0, 0, 0, 0);
        } else if (i < NIN + NW) {
            const int nkgp = (KK * ncb + 3) / 4 * 4;
            const char* wchunk = reinterpret_cast<const char*>(wpk + wchunk_off);   // wave-uniform base
            const int e = tid + 256 * (i - NIN);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (e / (MF * 16) < nkgp) {
                if (WREG) v = *reinterpret_cast<const uint4*>(wchunk + (long long)(i - NIN) * wstride + woff[0]);
                else v = *reinterpret_cast<const uint4*>(wchunk + woff[WREG ? 0 : i - NIN]);
            }
            rw[i - NIN] = v;
        }
    };
    auto prefetch = [&](int c0, long long wchunk_off) {
#pragma unroll
        for (int i = 0; i < NIN + NW; ++i) issue_load(i, c0, wchunk_off);
    };
    if (!DGRAD && tid < MF * 16) {   // bias of this block's channels -> LDS (read back in the epilogue)
        const int oc = mb * MF * 16 + tid;
        s_bias[tid] = (bias != nullptr && oc < n_out) ? bias[oc] : 0.f;
    }

    long long wchunk_off = 0;  // granule offset of the current chunk in the packed weights
    TRACE_STAMP();   // [1] prologue done
    prefetch(0, 0);
    TRACE_STAMP();   // [2] first prefetch issued
    for (int c0 = 0; c0 < ncb_tot; c0 += CHUNK_CB) {
        const int ncb = min(CHUNK_CB, ncb_tot - c0);
        const int nkg = KK * ncb, nkgp = (nkg + 3) / 4 * 4;
        __syncthreads();  // previous chunk's MFMAs are done with the LDS tiles
        TRACE_STAMP();    // chunk: after barrier A
#pragma unroll
        for (int i = 0; i < NIN; ++i)
            if (((idesc[i] >> 16) & 15) < ncb) s_in[idesc[i] & 0xffff] = rin[i];
        if (DGRAD && fold_tile) {
            // rare (border tiles of an UNFOLDED padded-domain gradient): pixels on row/col 1 or h-2/w-2 get the
            // mirrored halo folded in; deliberately NOT unrolled (keeps the hot path's registers free)
#pragma unroll 1
            for (int e = tid; e < ncb * TP * TP; e += 256) {
                const int cb = e / (TP * TP), p = e % (TP * TP);
                const int y = iy0 + p / TP, x = ix0 + p % TP;
                const bool inside = y >= 0 && y < tin.h && x >= 0 && x < tin.w;
                if (inside && ((y == 1) || (y == tin.h - 2) || (x == 1) || (x == tin.w - 2)))
                    s_in[cb * PL + p] = load_in_gradfold(tin, in_, c0 + cb, y, x);
            }
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int e = tid + 256 * i;
            if (e < nkgp * MF * 16) s_w[e] = rw[i];
        }
        // k-group offset table: .x = byte offset into s_in, .y = byte offset into s_w
        if (tid < nkgp) {   // visit position tid -> (tap, channel block) and the weight plane (packed tap-major) that holds it
            int tap = 0, cb = 0, plane = tid;
            if (tid < nkg) { tap = visit_tap(tid / ncb, KS); cb = tid % ncb; plane = tap * ncb + cb; }
            s_tab[tid] = make_int2((cb * PL + (tap / KS) * TP + (tap % KS)) * 16, plane * MF * 256);
        }
        TRACE_STAMP();    // chunk: LDS stores issued (includes the wait for the prefetched data)
        __syncthreads();
        TRACE_STAMP();    // chunk: after barrier B
        wchunk_off += (long long)nkgp * m16p;
        const bool have_next = c0 + CHUNK_CB < ncb_tot;
        const int nsteps = nkgp >> 2;
        // table-driven k-groups; the next chunk's global loads are issued up front and land during the MFMAs.
        // (Interleaving them into a fully unrolled k-loop was tried: the vector-memory issue cost is per instruction,
        // ~110 cycles each, and the unrolled body spilled 120+ VGPRs -- no gain.)
        if (have_next) prefetch(c0 + CHUNK_CB, wchunk_off);
        int2 off = s_tab[g];
        int2 nxt = s_tab[min(4, nkgp - 4) + g];
        bf16x8 b[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) b[n] = *reinterpret_cast<const bf16x8*>(in_lane + off.x + n * TP * 16);
        for (int s = 0; s < nsteps; ++s) {
            const int2 nxt2 = s_tab[min(4 * (s + 2), nkgp - 4) + g];
            bf16x8 a[MF], bn[4];
#pragma unroll
            for (int m = 0; m < MF; ++m) a[m] = *reinterpret_cast<const bf16x8*>(w_lane + off.y + m * 256);
#pragma unroll
            for (int n = 0; n < 4; ++n) bn[n] = *reinterpret_cast<const bf16x8*>(in_lane + nxt.x + n * TP * 16);
#pragma unroll
            for (int m = 0; m < MF; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < 4; ++n) b[n] = bn[n];
            off = nxt;
            nxt = nxt2;
        }
        TRACE_STAMP();    // chunk: k-loop done
    }

    conv_epilogue<MF, DGRAD>(acc, tout, tmask, s_bias, mb, in_, tile_x * MT + j, tile_y * MT + wave * 4 + (g & 1), g, relu, mask_bits,
                             accum_bits, tout.ws, tout.hs);
    TRACE_STAMP();        // epilogue stores issued
    if (tr != nullptr) tr[63] = tr_n;
#undef TRACE_STAMP
}

// ------------------------------------------------------------------ DMA-staged forward / dgrad kernel (3x3, 64-channel M-blocks)
// Same implicit GEMM, restructured around the two things the phase trace of the kernel above showed:
//   * staging through registers costs a vector-memory issue + an LDS-store phase per chunk and 60 VGPRs, and the packed
//     weight slab (64% of the staged bytes) is re-staged for every 256 pixels;
//   * two barriers per chunk, with the LDS tiles single-buffered.
// Here a block is 8 MFMA (consumer) waves on a 32x16-pixel tile (the weight slab serves 512 pixels) plus 4 loader waves,
// one per SIMD, that only issue the staging traffic; operands go global -> LDS directly
// (global_load_lds_dwordx4: lane l of a wave writes base + 16*l, tests/test_gpu_probe.py), the LDS tiles are double
// buffered (2 x 75 KB, one block per CU) and there is ONE barrier per chunk: the DMAs of chunk q+1 are issued right after
// the barrier that opens chunk q and have the whole k-loop to land.  Blocks are persistent: each walks its list of
// (tile, M-block) items (XCD-contiguous bands, as above), so the first chunk of the next tile is in flight during the last
// chunk of the current one, and a tile's epilogue (stores) runs after the NEXT chunk's DMAs are issued.
// Requires the input gradient of a dgrad to be a FOLDED halo-1 tensor (mmif_fold_halo): its zeroed halo ring is the zero fill.
constexpr int DT_ROWS = 32;                          // output tile rows (8 waves x 4)
static_assert(DT_ROWS == DT_ROWS_C, "tile rows");
constexpr int DTP_Y = DT_ROWS + 2, DTP_X = MT + 2;   // 34 x 18 input tile
constexpr int DPL = 624;                             // granules per LDS plane (612 used); 9984 B = 0 mod 256
constexpr int DIN_PIECES = CHUNK_CB * DPL / 64;      // 39 DMA pieces (64 granules = 1 KiB each)
constexpr int DW_PIECES = 36;                        // one k-group plane of the 64-row M-block per piece
constexpr int D_PIECES = DIN_PIECES + DW_PIECES;
constexpr int DBUF_BYTES = CHUNK_CB * DPL * 16 + DW_PIECES * 64 * 16;   // 76800
static_assert(CHUNK_CB * DPL % 64 == 0, "input planes must be whole DMA pieces");

#define MMIF_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define MMIF_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

struct DItem { int mb, in_, tile_y, tile_x; };

// ---- dgrad with the reflect-padding adjoint FOLDED IN (org = 1) --------------------------------------------------------------
// The padded output domain of a 3x3 dgrad is (h+2) x (w+2): tiling all of it costs 9 x 17 tiles of 32 x 16 for a 256 x 256 image
// where the interior needs 8 x 16, and the halo ring then has to be folded back by a second kernel.  With org = 1 the tiles cover
// the interior only and the tiles that own stored rows 2 / hs-3 or columns 2 / ws-3 (logical 1, h-2 / 1, w-2: the fold targets)
// add the ring pixels' values themselves: the ring pixel above stored row 2 is  sum_v Wk[2][v] g[1][X-1+v]  -- the operand
// fragment the row already uses for tap (0, v), times the weights of tap (2, v) -- so the fold costs 3 extra MFMA steps per chunk
// for that one row (12 for a border column, with every lane but the target's zeroed; 1 for a corner), no second pass over HBM
// and no bf16 rounding of the halo values.  The ring itself is never written (it stays zero: 'folded').
template <int N> struct IC { static constexpr int value = N; };

// One chunk's fold steps of the wave whose row n is stored row 1 + yl + n and whose lane j is stored column 1 + xl + j.
// in_c / w_c: this lane group's channel-block plane of the staged gradient tile (at the wave's first row, lane's column) and of
// the chunk's weight planes (tap t at w_c + t * ncb * MF * 256); pitch = granules per tile row.
template <int MF, int NBUF = (MF == 3 ? 1 : 2)>
__device__ inline void dgrad_fold_steps(f32x4 (&acc)[MF][4], const char* in_c, const char* w_c, int ncb, int pitch, int g, int j, int yl,
                                        int xl, int hs, int ws) {
    const int nb = hs - 4 - yl;                 // row n of stored row hs-3 (stored row 2 is row 1 of the wave with yl == 0)
    const int jl = 1 - xl, jr = ws - 4 - xl;    // lanes of stored columns 2 / ws-3
    const bool has_t = yl == 0, has_b = nb >= 0 && nb < 4, has_l = jl >= 0, has_r = jr >= 0 && jr < MT;
    if (!(has_t || has_b || has_l || has_r)) return;
    const bf16x8 zero8 = __builtin_bit_cast(bf16x8, make_uint4(0, 0, 0, 0));
    const bool kz = g >= ncb;   // lane group past the end of a ragged chunk
    // gradient fragment at (tile row brow relative to the wave's first, column offset bcol): all lanes (jsel < 0) or lane jsel only
    auto ldb = [&](int brow, int bcol, int jsel) {
        bf16x8 b = *reinterpret_cast<const bf16x8*>(in_c + (brow * pitch + bcol) * 16);
        if (kz || (jsel >= 0 && j != jsel)) b = zero8;
        return b;
    };
    auto lda = [&](int tap, int m) { return *reinterpret_cast<const bf16x8*>(w_c + tap * ncb * (MF * 256) + m * 256); };
    auto at_row = [&](int n, auto&& f) {   // run f(IC<n>) for a wave-uniform n in 0..3
        if (n == 0) f(IC<0>()); else if (n == 1) f(IC<1>()); else if (n == 2) f(IC<2>()); else f(IC<3>());
    };
    // a ring ROW folds onto one row of one wave: taps (urow, v), v = 0..2, on tile row brow -- all operands first, then the MFMAs
    auto row_steps = [&](auto nc, int tap0, int brow) {
        constexpr int n = decltype(nc)::value;
        bf16x8 b[3], a[3][MF];
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            b[v] = ldb(brow, v, -1);
#pragma unroll
            for (int m = 0; m < MF; ++m) a[v][m] = lda(tap0 + v, m);
        }
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int m = 0; m < MF; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[v][m], b[v], acc[m][n], 0, 0, 0);
    };
    // a ring COLUMN folds onto one lane of every row: three more k-steps (taps (u, vcol), u = 0..2) of the usual shape -- 4 weight
    // + 4 gradient fragments, 4 MF MFMAs -- with the next step's operands fetched under the current step's MFMAs
    auto col_steps = [&](int vcol, int bcol, int jsel) {
        constexpr int NB = NBUF;   // (default 1 for MF = 3: the 128-register budget of the thin kernel has no room for a second operand set)
        bf16x8 a[NB][MF], b[NB][4];
        auto fetch = [&](int u, int slot) {
#pragma unroll
            for (int m = 0; m < MF; ++m) a[slot][m] = lda(3 * u + vcol, m);
#pragma unroll
            for (int n = 0; n < 4; ++n) b[slot][n] = ldb(n + u, bcol, jsel);
        };
        if (NB == 2) fetch(0, 0);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (NB == 1) fetch(u, 0);
            else if (u + 1 < 3) fetch(u + 1, (u + 1) & 1);
#pragma unroll
            for (int m = 0; m < MF; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u & (NB - 1)][m], b[u & (NB - 1)][n], acc[m][n], 0, 0, 0);
        }
    };
    auto corner = [&](auto nc, int tap, int brow, int bcol, int jsel) {
        constexpr int n = decltype(nc)::value;
        const bf16x8 b = ldb(brow, bcol, jsel);
#pragma unroll
        for (int m = 0; m < MF; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lda(tap, m), b, acc[m][n], 0, 0, 0);
    };
    if (has_t) {   // ring row 0 -> stored row 2 (n = 1): taps (2, v) on g row 1 = tile row n + 0
        row_steps(IC<1>(), 6, 1);
        if (has_l) corner(IC<1>(), 8, 1, 0, jl);
        if (has_r) corner(IC<1>(), 6, 1, 2, jr);
    }
    if (has_b) {   // ring row hs-1 -> stored row hs-3: taps (0, v) on g row hs-2 = tile row n + 2
        at_row(nb, [&](auto nc) {
            constexpr int n = decltype(nc)::value;
            row_steps(nc, 0, n + 2);
            if (has_l) corner(nc, 2, n + 2, 0, jl);
            if (has_r) corner(nc, 0, n + 2, 2, jr);
        });
    }
    if (has_l) col_steps(2, 0, jl);   // ring column 0    -> stored column 2:    taps (u, 2) on g column 1
    if (has_r) col_steps(0, 2, jr);   // ring column ws-1 -> stored column ws-3: taps (u, 0) on g column ws-2
}

constexpr int D_CONS = 8;                                          // consumer waves (MFMA): wave w owns tile rows 4w..4w+3
constexpr int D_LOAD = 4;                                          // loader waves (LDS-DMA issue only), one per SIMD
constexpr int DL_ITERS = (D_PIECES + D_LOAD - 1) / D_LOAD;         // 19 pieces per loader wave per chunk
constexpr int DL_IN_ITERS = (DIN_PIECES + D_LOAD - 1) / D_LOAD;    // 10 of them may be input pieces

// ReLU sign bytes of an activation tensor, in the PADDED domain of its gradient: [n][cb][hs = h + 2][pitch] bytes, bit i of the byte
// at (row Y, column X) = channel 8 cb + i of image pixel (Y - 1, X - 1) is > 0.  The byte of column X lives at X + SIGN_XOFF, so that
// the 16 columns of a dgrad tile (first stored column 1 + 16 k) are one aligned 16-byte load.  Written by wgrad_dma_kernel (which has
// the activation tile in LDS anyway), read by conv_dma_kernel<true, 2>: 1/16 of the bytes of the activations themselves.
constexpr int SIGN_XOFF = 15;
struct SignMap {
    unsigned char* p;
    int cb, hs, pitch;
};
static inline int sign_pitch(int w) { return (cdiv(w, MT) + 2) * MT; }

// (LMASK: 0 = none, 1 = the loaders reduce the mask ACTIVATIONS to sign bytes, 2 = they fetch ready sign bytes -- struct SignMap)
// (DUP: the dgrad that also leaves struct DupOut's masked copies -- its own instantiation: as a run-time option of <true, 0> the extra
// operands spilled 28 registers at the 168-register budget)
template <bool DGRAD, int LMASK, bool DUP = false>
__global__ __launch_bounds__((D_CONS + D_LOAD) * 64, 3) void conv_dma_kernel(TV tin, TV tout, TV tmask, const uint4* __restrict__ wpk,
                                                                              const float* __restrict__ bias, int n_out, int m16p, int relu,
                                                                              unsigned long long mask_bits, unsigned long long accum_bits,
                                                                              int tiles_x, int tiles_y, int nmb, long long* __restrict__ trace,
                                                                              int abl, int org, SignMap sgn, DupOut dup) {
    constexpr int MF = 4;
    __shared__ __attribute__((aligned(16))) char s_buf[2 * DBUF_BYTES];
    __shared__ int2 s_tab[2][DW_PIECES];
    __shared__ __attribute__((aligned(16))) float s_bias[3][MF * 16];
    // dgrad: ReLU-mask sign bits of two tiles in flight ([item parity][channel block of the M-block][tile row][tile col], one byte =
    // 8 channels).  The consumers' epilogue used to fetch the mask activations itself -- 8 global round trips on the critical path of
    // every tile, and all of a tile's k-loop is only 2 chunks for a 64-channel input (decode.1's dgrad).  The LOADER waves have the
    // time: they fetch the tile's mask granules two chunks ahead, reduce them to bits and park them here.
    __shared__ __attribute__((aligned(16))) unsigned char s_mask[LMASK ? 2 : 1][LMASK ? 8 * DT_ROWS * MT : 16];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: roles, piece indices, LDS bases and M0 stay in SGPRs
    const int j = lane & 15, g = lane >> 4;
    // diagnostics (tools/trace_dma.py): lane 0 of every consumer wave of the first 128 blocks stamps s_memtime per phase
    int tr_n = 0;
    long long* tr = (trace != nullptr && blockIdx.x < 128 && lane == 0 && wave < D_CONS) ? trace + ((long long)blockIdx.x * 8 + wave) * 64 : nullptr;
#define DTRACE() do { if (tr != nullptr && tr_n < 62 && !(abl & 256)) tr[tr_n++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    // $MMIF_ABLATE conv= bit 8: stamp only the top of every third chunk (the period over the whole launch instead of the phases of
    // its first 12 chunks)
#define DTRACE_TOP(q) do { if (tr != nullptr && tr_n < 62 && (!(abl & 256) || (q) % 3 == 0)) tr[tr_n++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    if (tr != nullptr && tr_n < 62) tr[tr_n++] = (long long)__builtin_amdgcn_s_memtime();
    const int tpi = tiles_x * tiles_y;
    const int nitems = tpi * tout.n * nmb;
    // this block's items: first + i * stride, i < count (XCD x = blockIdx % 8 owns a contiguous band of items)
    int first, stride, count;
    {
        const int G = gridDim.x, b = blockIdx.x;
        if ((G & 7) == 0) {
            const int xcd = b & 7, slot = b >> 3, nsl = G >> 3;
            const int q8 = nitems >> 3, r8 = nitems & 7;
            const int band0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
            const int blen = q8 + (xcd < r8 ? 1 : 0);
            first = band0 + slot;
            stride = nsl;
            count = blen > slot ? (blen - slot + nsl - 1) / nsl : 0;
        } else {
            first = b;
            stride = G;
            count = b < nitems ? (nitems - b + G - 1) / G : 0;
        }
    }
    if (count == 0) return;
    auto decode = [&](int lin) {
        DItem it;
        it.mb = lin % nmb;
        const int tl = lin / nmb;
        it.in_ = tl / tpi;
        const int trem = tl - it.in_ * tpi;
        it.tile_y = trem / tiles_x;
        it.tile_x = trem - it.tile_y * tiles_x;
        // fused fold: a block's items are a whole number of tile rows apart, so without a skew one block would own ONLY left-border
        // tiles (three extra k-steps per chunk each) and set the kernel's makespan; rotate the columns by row and image
        if (org) it.tile_x = (it.tile_x + it.tile_y + it.in_) % tiles_x;
        return it;
    };
    const int ncb_tot = tin.cb;
    const int nch = (ncb_tot + CHUNK_CB - 1) / CHUNK_CB;
    const int ncb_last = ncb_tot - (nch - 1) * CHUNK_CB;
    const int total_q = count * nch;
    // loader-staged mask bits need two chunks of lead per tile (and a third buffer with one chunk per tile: not worth the LDS)
    // (LMASK: the host picks the instantiation -- dgrad, >= 2 chunks per tile, mask bits set)
    // k-group table of the ragged last chunk: .x = byte offset into the input tile, .y = byte offset into the weight slab
    if (tid < DW_PIECES) {
        const int kg = tid;
        int tap = 0, cb = 0, plane = kg;
        if (kg < 9 * ncb_last) { tap = visit_tap(kg / ncb_last, 3); cb = kg % ncb_last; plane = tap * ncb_last + cb; }
        s_tab[1][kg] = make_int2((cb * DPL + (tap / 3) * DTP_X + (tap % 3)) * 16, plane * MF * 256);
    }

    if (wave >= D_CONS) {
        // =============================== loader waves: stage chunk q+1 while the consumers run chunk q ===============================
        // Source of every staged granule is a plain address (nothing is selected on the issue path):
        //   fwd   reflect padding by index;
        //   dgrad out-of-image taps read the gradient's halo ring, which a FOLDED halo-1 tensor keeps at zero (mmif_fold_halo;
        //         include/mmif.h MMIF_T_FOLDED) -- stored coordinates are clamped into [0, hs) x [0, ws);
        //   channel blocks past the end of a ragged last chunk re-read the last valid plane (their packed weights are zero).
        const int lw = wave - D_CONS;
        unsigned d_off[DL_IN_ITERS];     // per input piece: byte offset of this lane's pixel inside a plane (per item)
        unsigned d_geo[DL_IN_ITERS];     // item independent: tile row << 8 | tile col
        unsigned d_cb = 0;               // chunk-local channel block of each piece, 2 bits per piece
#pragma unroll
        for (int i = 0; i < DL_IN_ITERS; ++i) {
            const int slot = (lw + D_LOAD * i) * 64 + lane;
            const int cb = min(slot / DPL, CHUNK_CB - 1);
            const int p = min(slot - cb * DPL, DTP_Y * DTP_X - 1);
            d_geo[i] = (unsigned)((p / DTP_X) << 8 | (p % DTP_X));
            d_cb |= (unsigned)cb << (2 * i);
        }
        const unsigned plane_bytes = (unsigned)(tin.plane * 16);
        auto make_desc = [&](const DItem& itm) {
            const int iy0 = itm.tile_y * DT_ROWS - tout.halo - 1 + org, ix0 = itm.tile_x * MT - tout.halo - 1 + org;
#pragma unroll
            for (int i = 0; i < DL_IN_ITERS; ++i) {
                int y = iy0 + (int)(d_geo[i] >> 8), x = ix0 + (int)(d_geo[i] & 255u);
                if (!DGRAD) {
                    y = min(max(reflect_idx(y, tin.h), 0), tin.h - 1);
                    x = min(max(reflect_idx(x, tin.w), 0), tin.w - 1);
                } else {
                    y = min(max(y + 1, 0), tin.hs - 1);
                    x = min(max(x + 1, 0), tin.ws - 1);
                }
                // pixel offset + the piece's chunk-local plane: nothing but the load itself is left per chunk (a v_mul_lo_u32 per
                // piece and chunk is quarter rate, and loader VALU time is MFMA time of the consumers on the same SIMD)
                d_off[i] = (__umul24((unsigned)y, (unsigned)tin.ws) + (unsigned)x) * 16u + ((d_cb >> (2 * i)) & 3u) * plane_bytes;
            }
        };
        // weight chunk (M-block, chunk) each buffer's weight region holds: a layer with at most two chunks per item (K <= 64: decode.1's
        // dgrad, DenseFuse's decode.0) whose block keeps one M-block finds its weights where the last item left them and does not request
        // them again (36 of a chunk's 75 pieces)
        int wres[2] = {-1, -1};
        auto issue_dma = [&](const DItem& itm, int c, int buf, int item_no = 0) {
            // timing ablations (results are garbage): $MMIF_ABLATE conv= bit 1 = no WEIGHT pieces on every second item (what a weight chunk
            // shared by two pixel tiles could save at most), bit 3 = no INPUT pieces on every second item (an input tile shared by two M-blocks)
            const int wkey = itm.mb * 256 + c;
            const bool skip_w = ((abl & 2) && (item_no & 1)) || (wres[buf] == wkey && !(abl & 64)), skip_in = (abl & 8) && (item_no & 1);
            wres[buf] = wkey;
            const int ncb = min(CHUNK_CB, ncb_tot - c * CHUNK_CB);
            const int nkgp = (9 * ncb + 3) / 4 * 4;
            char* dst_in = s_buf + buf * DBUF_BYTES;
            char* dst_w = dst_in + CHUNK_CB * DPL * 16;
            const char* src_in = tin.base + ((long long)itm.in_ * tin.img + (long long)(tin.cb_off + c * CHUNK_CB) * tin.plane) * 16;
            const char* src_w = reinterpret_cast<const char*>(wpk) + ((long long)c * DW_PIECES * m16p + itm.mb * MF * 16 + lane) * 16;
#pragma unroll
            for (int i = 0; i < DL_ITERS; ++i) {
                const int P = lw + D_LOAD * i;   // wave-uniform piece index
                if (i < DL_IN_ITERS && P < DIN_PIECES) {
                    unsigned off = d_off[i < DL_IN_ITERS ? i : 0];
                    if (ncb != CHUNK_CB) {   // ragged last chunk (wave uniform): planes past the end re-read the last valid one
                        const unsigned cb = (d_cb >> (2 * (i < DL_IN_ITERS ? i : 0))) & 3u;
                        off -= (cb - min(cb, (unsigned)(ncb - 1))) * plane_bytes;
                    }
                    if (!skip_in) __builtin_amdgcn_global_load_lds(MMIF_GPTR(src_in + off), MMIF_LPTR(dst_in + P * 1024), 16, 0, 0);
                } else if (P >= DIN_PIECES && P < D_PIECES) {
                    const int kg = P - DIN_PIECES;
                    if (kg < nkgp && !skip_w)
                        __builtin_amdgcn_global_load_lds(MMIF_GPTR(src_w + (long long)kg * m16p * 16), MMIF_LPTR(dst_w + kg * 1024), 16, 0, 0);
                }
            }
        };
        auto load_bias = [&](const DItem& itm, int slot) {
            const int t = tid - D_CONS * 64;
            if (!DGRAD && t < MF * 16) {
                const int oc = itm.mb * MF * 16 + t;
                s_bias[slot][t] = (bias != nullptr && oc < n_out) ? bias[oc] : 0.f;
            }
        };
        // ---- mask bits: thread t of the 256 loader threads owns tile pixels t and t + 256 of every channel block of the M-block
        uint4 mreg[LMASK == 1 ? 16 : 1];
        bool mpend = false;       // mask granules of item `mitem` are in flight / in registers
        int mpar = 0;
        auto issue_mask = [&](const DItem& itm) {
            const int t = tid - D_CONS * 64;
            if (LMASK == 2) {
                // ready-made sign bytes (written by the layer's weight-gradient kernel): thread t owns the 16 columns of
                // (channel block t >> 5, tile row t & 31) -- ONE 16-byte load per thread and tile instead of 16 granules
                const int cbl = t >> 5, row = t & 31;
                const int ocb = itm.mb * 8 + cbl;
                const bool need = ocb < tout.cb && ((mask_bits >> ocb) & 1ull);
                const int ys = min(org + itm.tile_y * DT_ROWS + row, sgn.hs - 1), xs = org + itm.tile_x * MT;
                mreg[0] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
                if (need) mreg[0] = *reinterpret_cast<const uint4*>(sgn.p + (((long long)itm.in_ * sgn.cb + min(ocb, sgn.cb - 1)) * sgn.hs + ys) * sgn.pitch + xs + SIGN_XOFF);
                return;
            }
#pragma unroll
            for (int cbl = 0; cbl < 8; ++cbl) {
                const int ocb = itm.mb * 8 + cbl;
                const bool need = ocb < tout.cb && ((mask_bits >> ocb) & 1ull);   // wave-uniform
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int p = t + 256 * k, row = p >> 4, col = p & 15;
                    const int oys = org + itm.tile_y * DT_ROWS + row, oxs = org + itm.tile_x * MT + col;
                    const int y = min(max(reflect_idx(oys - tout.halo, tmask.h), 0), tmask.h - 1);
                    const int x = min(max(reflect_idx(oxs - tout.halo, tmask.w), 0), tmask.w - 1);
                    if (need) mreg[LMASK == 1 ? 2 * cbl + k : 0] = *reinterpret_cast<const uint4*>(tmask.base + tmask.gidx(itm.in_, min(ocb, tmask.cb - 1), y, x) * 16);
                }
            }
        };
        auto park_mask = [&](const DItem& itm, int par) {
            const int t = tid - D_CONS * 64;
            if (LMASK == 2) {
                *reinterpret_cast<uint4*>(&s_mask[LMASK ? par : 0][LMASK ? t * MT : 0]) = mreg[0];   // [cbl][row][16 cols], t = cbl * 32 + row
                return;
            }
#pragma unroll
            for (int cbl = 0; cbl < 8; ++cbl) {
                const int ocb = itm.mb * 8 + cbl;
                const bool need = ocb < tout.cb && ((mask_bits >> ocb) & 1ull);
                if (!need) continue;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const uint4 v = mreg[LMASK == 1 ? 2 * cbl + k : 0];
                    const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
                    unsigned bits = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {   // bf16 > 0  <=>  positive as a signed 16-bit integer (activations are never NaN)
                        bits |= ((short)(wv[i] & 0xffffu) > 0 ? 1u : 0u) << (2 * i);
                        bits |= ((short)(wv[i] >> 16) > 0 ? 1u : 0u) << (2 * i + 1);
                    }
                    s_mask[LMASK ? par : 0][LMASK ? cbl * (DT_ROWS * MT) + t + 256 * k : 0] = (unsigned char)bits;
                }
            }
        };
        DItem cur = decode(first), mitem = cur;
        make_desc(cur);
        load_bias(cur, 0);
        issue_dma(cur, 0, 0);
        int c = 0, item_i = 0;
        if (LMASK && nch == 2) { issue_mask(cur); mitem = cur; mpar = 0; mpend = true; }   // chunk 0 is the first tile's chunk nch - 2
        for (int q = 0; q < total_q; ++q) {
            __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's pieces of chunk q (and any mask granules) have landed
            __builtin_amdgcn_s_barrier();         // consumers: chunk q is readable; loaders: the other buffer is free
            if (LMASK && mpend) {                 // the tile whose chunk nch - 2 just opened: its bits are visible at the NEXT barrier,
                park_mask(mitem, mpar);           // which opens its last chunk -- before any of its epilogues runs
                __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                mpend = false;
            }
            if (++c == nch) {
                c = 0;
                ++item_i;
                if (q + 1 < total_q) {
                    cur = decode(first + item_i * stride);
                    make_desc(cur);
                    load_bias(cur, item_i % 3);
                }
            }
            if (q + 1 < total_q && !(abl & 1)) issue_dma(cur, c, (q & 1) ^ 1, item_i);
            if (LMASK && q + 1 < total_q && c == nch - 2) { issue_mask(cur); mitem = cur; mpar = item_i & 1; mpend = true; }
        }
        return;
    }

    // =============================== consumer waves ===============================
    f32x4 acc[MF][4];
    // epilogue of item `itm` (bias slot / mask parity of that item): the packed form wherever it applies
#define DMA_EPILOGUE(itm, slot, par)                                                                                                          \
    do {                                                                                                                                      \
        const int oxs_ = org + (itm).tile_x * MT + j, oys0_ = org + (itm).tile_y * DT_ROWS + wave * 4 + (g & 1);                              \
        if constexpr (!DGRAD) {                                                                                                               \
            conv_epilogue_packed<MF, false>(acc, tout, s_bias[slot], (itm).mb, (itm).in_, org + (itm).tile_x * MT,                            \
                                            org + (itm).tile_y * DT_ROWS + wave * 4, wave * 4, relu, mask_bits, tout.ws - org, tout.hs - org,  \
                                            nullptr, tr, &tr_n);                                                                                     \
        } else if constexpr (LMASK != 0) {                                                                                                    \
            if (accum_bits == 0)                                                                                                              \
                conv_epilogue_packed<MF, true>(acc, tout, s_bias[slot], (itm).mb, (itm).in_, org + (itm).tile_x * MT,                         \
                                               org + (itm).tile_y * DT_ROWS + wave * 4, wave * 4, relu, mask_bits, tout.ws - org,             \
                                               tout.hs - org, s_mask[LMASK ? (par) : 0]);                                                     \
            else                                                                                                                              \
                conv_epilogue<MF, true, true>(acc, tout, tmask, s_bias[slot], (itm).mb, (itm).in_, oxs_, oys0_, g, relu, mask_bits,           \
                                              accum_bits, tout.ws - org, tout.hs - org, s_mask[LMASK ? (par) : 0], wave * 4 + (g & 1), j);    \
        } else {                                                                                                                              \
            conv_epilogue<MF, true, false, DUP>(acc, tout, tmask, s_bias[slot], (itm).mb, (itm).in_, oxs_, oys0_, g, relu, mask_bits,        \
                                                accum_bits, tout.ws - org, tout.hs - org, nullptr, wave * 4 + (g & 1), j, nullptr, &dup);     \
        }                                                                                                                                     \
    } while (0)
    DItem cur = decode(first), pend = cur;
    bool have_pend = false;
    int pend_slot = 0, pend_par = 0;
    int c = 0, item_i = 0;
    for (int q = 0; q < total_q; ++q) {
        const int buf = q & 1;
        DTRACE_TOP(q);   // chunk top
        __builtin_amdgcn_s_barrier();
        DTRACE();   // barrier passed
        // lane coordinates re-derived per chunk behind an asm the compiler cannot hoist: nothing lane dependent stays live across the
        // loop (at the 168-register budget the allocator otherwise spills one of them and reloads it -- scratch + vmcnt(0) -- per chunk)
        int lane_q;
        __asm__ volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_q));
        const int j = lane_q & 15, g = lane_q >> 4;
        if (have_pend) {   // waves 4..7: previous tile's outputs, stored under the partner wave's MFMAs (see below)
            __builtin_amdgcn_s_setprio(3);   // (measured neutral; the epilogue is ~180 VALU instructions either way)
            DMA_EPILOGUE(pend, pend_slot, pend_par);
            __builtin_amdgcn_s_setprio(0);
            have_pend = false;
        }
        DTRACE();   // pending epilogue done
        if (c == 0) {
#pragma unroll
            for (int m = 0; m < MF; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const int ncb = min(CHUNK_CB, ncb_tot - c * CHUNK_CB);
        const char* in_lane = s_buf + buf * DBUF_BYTES + ((wave * 4) * DTP_X + j) * 16;
        const char* w_lane = s_buf + buf * DBUF_BYTES + CHUNK_CB * DPL * 16 + j * 16;
        // a wave whose 4 rows lie below the output (ragged last tile row; 258 = 8*32 + 2 for the padded domain of a 256-row
        // dgrad) has nothing to compute: it leaves the MFMA pipe to its SIMD partner and just keeps the barrier count
        const bool rows_live = !DGRAD || org + cur.tile_y * DT_ROWS + wave * 4 < tout.hs - org;
        if (!rows_live) {
        } else if (ncb == CHUNK_CB) {
            // ---- full chunk: k-step s = tap s over channel blocks g = 0..3; fully unrolled (immediate LDS offsets).
            // Operands of k-step s+1 are fetched while the MFMAs of k-step s run (explicit double buffer, order pinned with
            // sched_group_barrier: the 8 LDS reads ride on the first 8 MFMAs, the other 8 cover the last reads' latency)
            const char* in_g = in_lane + g * (DPL * 16);
            const char* w_g = w_lane + g * (MF * 256);
            // Column-major tap order (visit_tap): the three steps of column offset v use tile rows u .. u+3 of the SAME six rows, so
            // a row fragment is read from LDS once per column (18 instead of 36 gradient / activation fragments per chunk; the 36
            // weight fragments stay).  rows[] is a ring of 8 fragments: row r of column v lives in slot (6 v + r) % 8 -- during step
            // (v, 2) the four live rows and the next column's first four fill it exactly.  Weights are double buffered per step.
            bf16x8 a[2][MF], rows[8];
            auto ld_row = [&](int v, int r) { return *reinterpret_cast<const bf16x8*>(in_g + (r * DTP_X + v) * 16); };
            auto ld_a = [&](int t, int slot) {
                const int tap = (t % 3) * 3 + t / 3;
#pragma unroll
                for (int m = 0; m < MF; ++m) a[slot][m] = *reinterpret_cast<const bf16x8*>(w_g + tap * 4 * MF * 256 + m * 256);
            };
#pragma unroll
            for (int r = 0; r < 4; ++r) rows[r] = ld_row(0, r);
            ld_a(0, 0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int v = t / 3, u = t % 3;
                int nreads = 0;
                if (t + 1 < 9) {   // operands of step t+1, in the order its MFMAs consume them
                    const int v1 = (t + 1) / 3, u1 = (t + 1) % 3;
                    if (u1 == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) rows[(6 * v1 + r) % 8] = ld_row(v1, r);
                        nreads += 4;
                    } else {
                        rows[(6 * v1 + 3 + u1) % 8] = ld_row(v1, 3 + u1);
                        nreads += 1;
                    }
                    ld_a(t + 1, (t + 1) & 1);
                    nreads += MF;
                }
#pragma unroll
                for (int m = 0; m < MF; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t & 1][m], rows[(6 * v + u + n) % 8], acc[m][n], 0, 0, 0);
                if (t + 1 < 9) {   // the LDS reads ride on the first MFMAs, the rest cover the last reads' latency
#pragma unroll
                    for (int r = 0; r < 5; ++r) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                    }
                    if ((t + 1) % 3 == 0) {   // a new column: 4 + 4 reads
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                    } else {                  // 1 + 4 reads
                        __builtin_amdgcn_sched_group_barrier(0x008, 11, 0);
                    }
                }
                (void)nreads;
            }
        } else {
            const int nkgp = (9 * ncb + 3) / 4 * 4, nsteps = nkgp >> 2;
            const int2* tab = s_tab[1];
            int2 off = tab[g];
            int2 nx = tab[min(4, nkgp - 4) + g];
            bf16x8 b[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) b[n] = *reinterpret_cast<const bf16x8*>(in_lane + off.x + n * DTP_X * 16);
            for (int s2 = 0; s2 < nsteps; ++s2) {
                const int2 nx2 = tab[min(4 * (s2 + 2), nkgp - 4) + g];
                bf16x8 a[MF], bn[4];
#pragma unroll
                for (int m = 0; m < MF; ++m) a[m] = *reinterpret_cast<const bf16x8*>(w_lane + off.y + m * 256);
#pragma unroll
                for (int n = 0; n < 4; ++n) bn[n] = *reinterpret_cast<const bf16x8*>(in_lane + nx.x + n * DTP_X * 16);
#pragma unroll
                for (int m = 0; m < MF; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < 4; ++n) b[n] = bn[n];
                off = nx;
                nx = nx2;
            }
        }
        if (DGRAD && org && rows_live && !(abl & 4))
            dgrad_fold_steps<MF>(acc, in_lane + min(g, ncb - 1) * (DPL * 16), w_lane + min(g, ncb - 1) * (MF * 256), ncb, DTP_X, g, j,
                                 cur.tile_y * DT_ROWS + wave * 4, cur.tile_x * MT, tout.hs, tout.ws);
        DTRACE();   // k-loop done
        if (++c == nch) {
            c = 0;
            // Waves 0..3 are the older wave of their SIMD and win the MFMA arbitration: they leave the k-loop first and store
            // their outputs now, under the partner's remaining MFMAs; waves 4..7 store after the next barrier, under the
            // partner's next k-loop.  (All eight storing at the same point leaves the MFMA pipe idle for a whole epilogue.)
            if (wave < 4) {
                DMA_EPILOGUE(cur, item_i % 3, item_i & 1);
            } else {
                pend = cur;
                pend_slot = item_i % 3;
                pend_par = item_i & 1;
                have_pend = true;
            }
            ++item_i;
            if (q + 1 < total_q) cur = decode(first + item_i * stride);
        }
        DTRACE();   // tile epilogue (waves 0..3) done
    }
    if (tr != nullptr) { tr[63] = tr_n; tr[62] = (long long)__builtin_amdgcn_s_memtime(); }   // [62]: end of the block's last chunk
#undef DTRACE
#undef DTRACE_TOP
    if (have_pend) DMA_EPILOGUE(pend, pend_slot, pend_par);
#undef DMA_EPILOGUE
}

// ------------------------------------------------------------------ thin layers: asynchronous loader / consumer kernel
// 3x3 layers with <= 48 output and <= 48 input channels (DenseBlock encoder convs and their dgrads, decode.3): HBM-bound and,
// with almost no MFMA work per tile, LATENCY-bound in both kernels above.  What they need is several independent tiles per CU in
// different phases with loads running ahead (a block-barrier ring with one consumer group was
// measured slower than four independent register-staged blocks).  So:
//   * persistent block = 4 loader waves + 3 consumer GROUPS of 4 waves (group q takes the block's tiles q, q+3, ...);
//   * the whole packed weight image (<= 54 KB) is resident in LDS; 16x16-pixel input tiles (all channel blocks) stream by LDS-DMA
//     through a ring of NS >= 4 slots, D = min(2, NS-3) tiles in flight per loader (counted vmcnt);
//   * NO block barrier in the steady state: loaders publish "tile landed" and consumers "slot free" through LDS counters
//     (ds_add_u32 by lane 0, polled with s_sleep), so the two groups and the loaders drift freely against each other.
// Spin loops are bounded (a failed hand-off would produce wrong numbers, never a hung GPU).
constexpr int TN_MAXCB = 6;                    // input channel blocks per tile (cin <= 48)
constexpr int TN_PL = 336;                     // granules per LDS plane (18x18 = 324 used); 5376 B = 0 mod 256
constexpr int TN_MAXKG = 2 * 36;               // k-group planes of two chunks
constexpr int TN_GROUPS = 3, TN_LOAD = 4, TN_MAXSLOTS = 8;   // 12 consumer + 4 loader waves = one 1024-thread block per CU
__host__ __device__ constexpr int tn_ring_bytes(int mf) { return mf == 1 ? 128 * 1024 : (mf == 2 ? 112 * 1024 : 96 * 1024); }

__device__ inline void tn_wait_vmcnt(int n) {   // s_waitcnt vmcnt(n), n wave-uniform in 0..16 (the immediate must be a constant)
    switch (n) {
        case 0: __builtin_amdgcn_s_waitcnt(0x0f70); break;
        case 1: __builtin_amdgcn_s_waitcnt(0x0f71); break;
        case 2: __builtin_amdgcn_s_waitcnt(0x0f72); break;
        case 3: __builtin_amdgcn_s_waitcnt(0x0f73); break;
        case 4: __builtin_amdgcn_s_waitcnt(0x0f74); break;
        case 5: __builtin_amdgcn_s_waitcnt(0x0f75); break;
        case 6: __builtin_amdgcn_s_waitcnt(0x0f76); break;
        case 7: __builtin_amdgcn_s_waitcnt(0x0f77); break;
        case 8: __builtin_amdgcn_s_waitcnt(0x0f78); break;
        case 9: __builtin_amdgcn_s_waitcnt(0x0f79); break;
        case 10: __builtin_amdgcn_s_waitcnt(0x0f7a); break;
        case 11: __builtin_amdgcn_s_waitcnt(0x0f7b); break;
        case 12: __builtin_amdgcn_s_waitcnt(0x0f7c); break;
        case 13: __builtin_amdgcn_s_waitcnt(0x0f7d); break;
        case 14: __builtin_amdgcn_s_waitcnt(0x0f7e); break;
        case 15: __builtin_amdgcn_s_waitcnt(0x0f7f); break;
        default: __builtin_amdgcn_s_waitcnt(0x4f70); break;   // vmcnt(16): bit 14 = vmcnt bit 4
    }
}
// wave-uniform bounded poll of an LDS counter published by other waves
// The read is an asm ds_read_b32: as a C++ `volatile` read of a generic pointer it compiled to flat_load_dword + s_waitcnt vmcnt(0) --
// i.e. every poll first waited for ALL of the wave's outstanding vector-memory operations: a consumer's epilogue stores of the previous
// tile (a full HBM write latency per tile and group: the kernel took ~1.5 us per tile and CU whatever its epilogue loaded), a loader's
// in-flight LDS-DMA of the tiles behind the one whose slot it waits for.
__device__ inline void tn_wait_counter(volatile unsigned* ctr, unsigned target) {
    const unsigned a = (unsigned)(size_t)ctr;   // LDS byte offset (low half of the generic address)
    for (int spin = 0; spin < (1 << 22); ++spin) {
        unsigned v;
        __asm__ volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        if (v >= target) break;
        __builtin_amdgcn_s_sleep(1);
    }
    __asm__ volatile("" ::: "memory");
}

// GROUPS / PL / MAXP / RING / TIGHT (round 4): the 64 -> 32 forward (decode.2 of every PFNet / DenseFuse decoder: 8 input channel blocks) does
// not fit the default geometry -- three consumer groups need four 45 KiB slots next to the 36 KiB image.  Its variant runs TWO consumer
// groups on a ring of THREE slots of exactly the tile's 41 pieces (324-granule planes; the loaders' padding pieces re-stage the last real
// piece in place, so a slot needs no room for them): 123 KiB + 36 KiB, 768 threads.  Every other layer keeps the geometry it was tuned on.
template <int MF, bool DGRAD, int GROUPS = TN_GROUPS, int PL = TN_PL, int MAXP = 8, int RING = tn_ring_bytes(MF), bool TIGHT = false>
__global__ __launch_bounds__((4 * GROUPS + TN_LOAD) * 64, GROUPS == TN_GROUPS ? 4 : 3) void thin_conv_async_kernel(
    TV tin, TV tout, TV tmask, const uint4* __restrict__ wpk, const float* __restrict__ bias, int n_out, int relu,
    unsigned long long mask_bits, unsigned long long accum_bits, int tiles_x, int tiles_y, int ntiles, int org, TV told, int use_old) {
    constexpr int TP = MT + 2;   // org: as in conv_dma_kernel (dgrad over the interior of the padded domain, fold steps in the border tiles)
    constexpr int NCONS = 4 * GROUPS;
    __shared__ __attribute__((aligned(16))) char s_in[RING];
    __shared__ __attribute__((aligned(16))) uint4 s_w[TN_MAXKG * MF * 16];
    __shared__ int2 s_tab[2][36];
    __shared__ __attribute__((aligned(16))) float s_bias[MF * 16];
    __shared__ unsigned s_ready[TN_MAXSLOTS], s_done[TN_MAXSLOTS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int G = gridDim.x, b = blockIdx.x;
    const int tpi = tiles_x * tiles_y;
    const int ncb_tot = tin.cb;
    const int nch = (ncb_tot + CHUNK_CB - 1) / CHUNK_CB;
    const int nmine = b < ntiles ? (ntiles - b + G - 1) / G : 0;
    if (nmine == 0) return;
    // XCD-aware tile order (block b and all of its tiles b + kG run on XCD b % 8 when G % 8 == 0): contiguous band per XCD
    auto tile_of = [&](int i, int& in_, int& tile_y, int& tile_x) {
        int lin = i;
        if ((G & 7) == 0) {
            const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = i & 7;
            lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (i >> 3);
        }
        in_ = lin / tpi;
        const int trem = lin - in_ * tpi;
        tile_y = trem / tiles_x;
        tile_x = trem - tile_y * tiles_x;
        if (org) tile_x = (tile_x + tile_y + in_) % tiles_x;   // (see conv_dma_kernel: spread the border tiles over the blocks)
    };
    const int in_pieces = (ncb_tot * PL + 63) / 64;           // <= 4 MAXP
    const int P = (in_pieces + TN_LOAD - 1) / TN_LOAD;        // pieces per loader wave per tile (every loader issues exactly P)
    const int slot_bytes = TIGHT ? in_pieces * 1024 : P * TN_LOAD * 1024;
    const int NS = min(TN_MAXSLOTS, RING / slot_bytes);   // >= GROUPS + 1 (checked by the host)
    const int D = min(2, NS - GROUPS);                              // tiles a loader keeps in flight behind the one it publishes

    // ---- one-time setup: k-group tables, bias, counters, resident weights; ONE block barrier, none afterwards
    if (tid < 72) {
        const int c = tid / 36, kg = tid % 36;
        const int ncb = min(CHUNK_CB, ncb_tot - c * CHUNK_CB);
        if (ncb > 0) {
            int tap = 0, cb = 0, plane = kg;
            if (kg < 9 * ncb) { tap = visit_tap(kg / ncb, 3); cb = kg % ncb; plane = tap * ncb + cb; }
            s_tab[c][kg] = make_int2(((c * CHUNK_CB + cb) * PL + (tap / 3) * TP + (tap % 3)) * 16, (c * 36 + plane) * MF * 256);
        }
    }
    if (!DGRAD && tid < MF * 16) s_bias[tid] = (bias != nullptr && tid < n_out) ? bias[tid] : 0.f;
    if (tid < TN_MAXSLOTS) { s_ready[tid] = 0u; s_done[tid] = 0u; }
    if (wave >= NCONS) {
        // the packed image of a single M-block (m16p == MF*16) is contiguous ([chunk][k-group][MF*16 rows][8]); chunk c starts
        // at k-group plane 36 c in both the image and s_w, and its size is a multiple of 1 KiB
        int wbytes = 0;
        for (int c = 0; c < nch; ++c) wbytes += ((9 * min(CHUNK_CB, ncb_tot - c * CHUNK_CB) + 3) / 4 * 4) * MF * 256;
        const char* src = reinterpret_cast<const char*>(wpk) + lane * 16;
        char* dst = reinterpret_cast<char*>(s_w);
        for (int pz = wave - NCONS; pz * 1024 < wbytes; pz += TN_LOAD)
            __builtin_amdgcn_global_load_lds(MMIF_GPTR(src + (long long)pz * 1024), MMIF_LPTR(dst + pz * 1024), 16, 0, 0);
        __builtin_amdgcn_s_waitcnt(0x0f70);
    }
    __syncthreads();

    if (wave >= NCONS) {
        // =============================== loader waves ===============================
        const int lw = wave - NCONS;
        unsigned geo[MAXP];   // tile independent: plane << 16 | tile row << 8 | tile col of the granule this lane stages (piece lw + 4 i)
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            // (pieces past in_pieces -- padding so that every loader issues exactly P -- re-stage the last real piece: same sources)
            const int slot = min(min(lw + TN_LOAD * i, in_pieces - 1) * 64 + lane, ncb_tot * PL - 1);
            const int pl = slot / PL, p = min(slot - pl * PL, TP * TP - 1);
            geo[i] = (unsigned)(pl << 16 | (p / TP) << 8 | (p % TP));
        }
        const unsigned plane_bytes = (unsigned)(tin.plane * 16);
        auto issue_tile = [&](int k) {
            int in_, ty0, tx0;
            tile_of(b + k * G, in_, ty0, tx0);
            const int iy0 = ty0 * MT - tout.halo - 1 + org, ix0 = tx0 * MT - tout.halo - 1 + org;
            const char* src = tin.base + ((long long)in_ * tin.img + (long long)tin.cb_off * tin.plane) * 16;
            char* dst = s_in + (k % NS) * slot_bytes;
#pragma unroll
            for (int i = 0; i < MAXP; ++i) {
                if (i < P) {
                    const int piece = min(lw + TN_LOAD * i, in_pieces - 1);
                    const int pl = (int)(geo[i] >> 16);
                    int y = iy0 + (int)((geo[i] >> 8) & 255u), x = ix0 + (int)(geo[i] & 255u);
                    if (!DGRAD) {
                        y = min(max(reflect_idx(y, tin.h), 0), tin.h - 1);
                        x = min(max(reflect_idx(x, tin.w), 0), tin.w - 1);
                    } else {   // FOLDED halo-1 gradient: out-of-image taps read its zeroed halo ring
                        y = min(max(y + 1, 0), tin.hs - 1);
                        x = min(max(x + 1, 0), tin.ws - 1);
                    }
                    const unsigned off = (unsigned)(y * tin.ws + x) * 16u + (unsigned)pl * plane_bytes;
                    __builtin_amdgcn_global_load_lds(MMIF_GPTR(src + off), MMIF_LPTR(dst + piece * 1024), 16, 0, 0);
                }
            }
        };
        auto publish = [&](int k) {   // this wave's part of tile k has landed
            // asm: before a C++ atomicAdd on LDS the compiler drains the wave's LDS-DMA (s_waitcnt vmcnt(0): "may alias"), i.e. ALL the
            // younger tiles this loader keeps in flight -- each publish then cost a full HBM latency and the kernel ~1.5 us per tile and
            // CU whatever D was.  tn_wait_vmcnt above has already waited for exactly the tile being published.
            if (lane == 0) {
                const unsigned a = (unsigned)(size_t)&s_ready[k % NS], one = 1u;
                __asm__ volatile("ds_add_u32 %0, %1" : : "v"(a), "v"(one) : "memory");
            }
        };
        for (int k = 0; k < nmine; ++k) {
            if (k >= NS) tn_wait_counter(&s_done[k % NS], 4u * (unsigned)(k / NS));   // the slot's previous tile is consumed
            issue_tile(k);
            if (k >= D) {
                tn_wait_vmcnt(D * P);   // tile k-D landed; the D younger tiles stay in flight
                publish(k - D);
            }
        }
        for (int k = max(nmine - D, 0); k < nmine; ++k) {
            tn_wait_vmcnt((nmine - 1 - k) * P);
            publish(k);
        }
        return;
    }

    // =============================== consumer groups: group q = tiles q, q+3, ...; wave r of a group owns tile rows 4r .. 4r+3
    const int q = wave >> 2, r = wave & 3;
    const char* w_lane = reinterpret_cast<const char*>(s_w) + j * 16;
    for (int k = q; k < nmine; k += GROUPS) {
        int in_, ty0, tx0;
        tile_of(b + k * G, in_, ty0, tx0);
        tn_wait_counter(&s_ready[k % NS], (unsigned)TN_LOAD * (unsigned)(k / NS + 1));
        const char* in_lane = s_in + (k % NS) * slot_bytes + ((r * 4) * TP + j) * 16;
        f32x4 acc[MF][4];
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nch; ++c) {
            const int ncb = min(CHUNK_CB, ncb_tot - c * CHUNK_CB);
            const int nkgp = (9 * ncb + 3) / 4 * 4, nsteps = nkgp >> 2;
            const int2* tab = s_tab[c];
            int2 off = tab[g];
            int2 nx = tab[min(4, nkgp - 4) + g];
            bf16x8 bq[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) bq[n] = *reinterpret_cast<const bf16x8*>(in_lane + off.x + n * TP * 16);
            for (int s2 = 0; s2 < nsteps; ++s2) {
                const int2 nx2 = tab[min(4 * (s2 + 2), nkgp - 4) + g];
                bf16x8 a[MF], bn[4];
#pragma unroll
                for (int m = 0; m < MF; ++m) a[m] = *reinterpret_cast<const bf16x8*>(w_lane + off.y + m * 256);
#pragma unroll
                for (int n = 0; n < 4; ++n) bn[n] = *reinterpret_cast<const bf16x8*>(in_lane + nx.x + n * TP * 16);
#pragma unroll
                for (int m = 0; m < MF; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], bq[n], acc[m][n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < 4; ++n) bq[n] = bn[n];
                off = nx;
                nx = nx2;
            }
        }
        if (DGRAD && org) {
            for (int c = 0; c < nch; ++c) {
                const int ncb = min(CHUNK_CB, ncb_tot - c * CHUNK_CB), gc = min(g, ncb - 1);
                dgrad_fold_steps<MF>(acc, in_lane + (c * CHUNK_CB + gc) * (PL * 16), w_lane + (c * 36 + gc) * (MF * 256), ncb, TP, g, j,
                                     ty0 * MT + r * 4, tx0 * MT, tout.hs, tout.ws);
            }
        }
        // every LDS read of this tile has been consumed by an MFMA above: hand the slot back before the (long) epilogue
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) atomicAdd(&s_done[k % NS], 1u);
        conv_epilogue<MF, DGRAD>(acc, tout, tmask, s_bias, 0, in_, org + tx0 * MT + j, org + ty0 * MT + r * 4 + (g & 1), g, relu, mask_bits,
                                 accum_bits, tout.ws - org, tout.hs - org, nullptr, 0, 0, (DGRAD && use_old) ? &told : nullptr);
    }
}

// ------------------------------------------------------------------ wgrad kernel (K = pixels)
// block: (pixel-tile group gi, input-channel group of 16, output-channel group of MFW*16).
// k-step = 2 tile rows = 32 pixels; lane group g of a k-step: tile row 2s + (g>>1), cols 8*(g&1)..+7.
// The 4 waves split K (KSPLIT halves/quarters of the tile's 8 k-steps) x M (MW = MFW*KSPLIT/4 M-frags
// each): A = g^T (oc x pixels), B = shifted x (pixels x (tap, ic)); acc[m][tap] are 16x16 blocks.
constexpr int WG_XPL = 324;   // x-tile plane stride in granules  (5184 B = 64 mod 256)
constexpr int WG_GPL = 260;   // g-tile plane stride in granules  (4160 B = 64 mod 256)

// ICF = 16-channel input fragments per block (round 4).  A 1x1 layer has one tap, so a block that owns 16 input channels runs 4 MFMAs
// per k-step behind a full stage of the 64-channel g tile: NestFuse's 152 -> 64 layer staged 10 x (2 + 8) planes per tile for 27 planes
// of data and ran at 2.1 TB/s.  With ICF = 4 a block owns 64 input channels (3 x (8 + 8) planes), the g fragments feed four times the
// MFMAs.  3x3 layers keep ICF = 1 (their 64-channel pairs go to wgrad_dma_kernel).
template <int KS, int MFW, int KSPLIT, int ICF = 1>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(TV tx, TV tg, float* __restrict__ partial, int tiles_x,
                                                         int tpi, int total, int G, int n_icg, int n_ocg) {
    constexpr int KK = KS * KS, P = KS / 2, TP = MT + KS - 1;
    constexpr int XPL = KS == 3 ? WG_XPL : WG_GPL;
    constexpr int MGROUPS = 4 / KSPLIT;                  // wave groups along M
    constexpr int MW = MFW / MGROUPS;                    // M-frags per wave
    constexpr int KSTEPS = 8 / KSPLIT;                   // k-steps per wave per tile
    constexpr int ICW = 16 * ICF;                        // input channels per block
    constexpr int PER = MFW * 16 * ICW * KK + MFW * 16;  // floats per block partial
    constexpr int TILE_BYTES = (2 * ICF * XPL + MFW * 2 * WG_GPL) * 16;
    constexpr int RED_BYTES = PER * 4;
    constexpr int SM_BYTES = TILE_BYTES > RED_BYTES ? TILE_BYTES : RED_BYTES;
    static_assert(MFW % MGROUPS == 0, "bad wave split");
    __shared__ __attribute__((aligned(16))) char smem[SM_BYTES];
    uint4* s_x = reinterpret_cast<uint4*>(smem);
    uint4* s_g = s_x + 2 * ICF * XPL;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sl = lane & 15, g = lane >> 4;
    const int km = wave % KSPLIT, mm = wave / KSPLIT;
    // XCD-aware block order: the n_icg*n_ocg blocks that walk the SAME tiles are adjacent in dispatch order
    // and land on the same XCD (block b runs on XCD b % 8), so the shared g / x tiles are L2 hits
    const int npairs = n_icg * n_ocg;
    const int b = blockIdx.x;
    int gi, pair;
    if ((G & 7) == 0) { pair = (b >> 3) % npairs; gi = ((b >> 3) / npairs) * 8 + (b & 7); }
    else { pair = b % npairs; gi = b / npairs; }
    const int icg = pair % n_icg, ocg = pair / n_icg;
    const int xcb0 = icg * 2 * ICF, gcb0 = ocg * MFW * 2;

    f32x4 acc[MW][KK * ICF];
    f32x4 accb[MW];
#pragma unroll
    for (int m = 0; m < MW; ++m) {
        accb[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < KK * ICF; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // constant all-ones bf16 operand for the bias-gradient column sums
    const uint4 ones_u = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_u);

    // per-lane transpose-read addressing: in-group lane sl supplies row (sl>>2) = pixel, chunk (sl&3)
    const int tr_row = sl >> 2, tr_c = sl & 3;
    const int lane_plane = tr_c >> 1, lane_byte = (tr_c & 1) * 8;

    constexpr int NX = (2 * ICF * TP * TP + 255) / 256;   // x granules per thread per tile
    constexpr int NG = MFW * 2;                         // g granules per thread per tile (one per plane)
    uint4 rx[NX], rg[NG];
    // all prefetch loads are unconditional (clamped addresses, zero by select): nothing serialises them
    auto prefetch = [&](int tile) {
        const int in_ = tile / tpi, tt = tile % tpi;
        const int y0 = (tt / tiles_x) * MT, x0 = (tt % tiles_x) * MT;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = min(tid + 256 * i, 2 * ICF * TP * TP - 1);
            const int cb = e / (TP * TP), p = e % (TP * TP);
            const int y = min(max(reflect_idx(y0 + p / TP - P, tx.h), 0), tx.h - 1);
            const int x = min(max(reflect_idx(x0 + p % TP - P, tx.w), 0), tx.w - 1);
            const uint4 v = ld_gran(tx, in_, min(xcb0 + cb, tx.cb - 1), y, x);
            rx[i] = (xcb0 + cb < tx.cb) ? v : make_uint4(0, 0, 0, 0);
        }
        const int gy = y0 + tid / MT, gx = x0 + tid % MT;
        const bool inside = gy < tg.h && gx < tg.w;
        const int cy = min(gy, tg.h - 1) + tg.halo, cx = min(gx, tg.w - 1) + tg.halo;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const uint4 v = ld_gran(tg, in_, min(gcb0 + i, tg.cb - 1), cy, cx);
            rg[i] = (inside && gcb0 + i < tg.cb) ? v : make_uint4(0, 0, 0, 0);
        }
    };
    const TileWalk tw = xcd_walk(total, G, gi);
    if (tw.count > 0) prefetch(tw.first);
    for (int it = 0, tile = tw.first; it < tw.count; ++it, tile += tw.stride) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = tid + 256 * i;
            const int cb = e / (TP * TP), p = e % (TP * TP);
            if (cb < 2 * ICF) s_x[cb * XPL + p] = rx[i];
        }
#pragma unroll
        for (int i = 0; i < NG; ++i) s_g[i * WG_GPL + tid] = rg[i];
        if (tg.halo && !tg.folded) {
            // rare: tiles containing row/col 1 or h-2 / w-2 of a padded-domain gradient fold the halo in
            const int in_ = tile / tpi, tt = tile % tpi;
            const int y0 = (tt / tiles_x) * MT, x0 = (tt % tiles_x) * MT;
            const bool fold_tile = (y0 <= 1 && 1 < y0 + MT) || (y0 <= tg.h - 2 && tg.h - 2 < y0 + MT) ||
                                   (x0 <= 1 && 1 < x0 + MT) || (x0 <= tg.w - 2 && tg.w - 2 < x0 + MT);
            if (fold_tile) {
                const int gy = y0 + tid / MT, gx = x0 + tid % MT;
                if (gy < tg.h && gx < tg.w && (gy == 1 || gy == tg.h - 2 || gx == 1 || gx == tg.w - 2)) {
#pragma unroll 1
                    for (int i = 0; i < NG; ++i)
                        if (gcb0 + i < tg.cb) s_g[i * WG_GPL + tid] = load_in_gradfold(tg, in_, gcb0 + i, gy, gx);
                }
            }
        }
        __syncthreads();
        if (it + 1 < tw.count) prefetch(tile + tw.stride);  // in flight during the MFMAs below
#pragma unroll
        for (int ss = 0; ss < KSTEPS; ++ss) {
            const int s = km * KSTEPS + ss;
            const int row = 2 * s + (g >> 1), col0 = 8 * (g & 1);
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            // A fragments: g^T, M-frag m = planes 2m, 2m+1
            bf16x8 a[MW];
#pragma unroll
            for (int m = 0; m < MW; ++m) {
                const char* base = reinterpret_cast<const char*>(s_g) +
                                   (((2 * (mm * MW + m) + lane_plane) * WG_GPL) + row * MT + col0 + tr_row) * 16 + lane_byte;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, base));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, base + 4 * 16));
                const s16x8 c = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                a[m] = __builtin_bit_cast(bf16x8, c);
            }
            if (icg == 0) {
#pragma unroll
                for (int m = 0; m < MW; ++m) accb[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], ones, accb[m], 0, 0, 0);
            }
#pragma unroll
            for (int f = 0; f < ICF; ++f)
#pragma unroll
            for (int t = 0; t < KK; ++t) {
                const int u = t / KS, v = t % KS;
                const char* base = reinterpret_cast<const char*>(s_x) +
                                   (((2 * f + lane_plane) * XPL) + (row + u) * TP + col0 + v + tr_row) * 16 + lane_byte;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, base));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, base + 4 * 16));
                const s16x8 c = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                const bf16x8 bb = __builtin_bit_cast(bf16x8, c);
#pragma unroll
                for (int m = 0; m < MW; ++m) acc[m][f * KK + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], bb, acc[m][f * KK + t], 0, 0, 0);
            }
        }
    }
    // ---- combine the KSPLIT K-slices through LDS, then write the block partial ----
    // lane (g, sl) reg r holds (oc = 16*frag + 4g + r, ic = sl) for tap t; bias sums: any column (all equal)
    float* red = reinterpret_cast<float*>(smem);  // [MFW*16 oc][16 ICF ic][KK] + [MFW*16]
    for (int w = 0; w < KSPLIT; ++w) {
        __syncthreads();
        if (km == w) {
#pragma unroll
            for (int m = 0; m < MW; ++m) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int oc = 16 * (mm * MW + m) + 4 * g + r;
#pragma unroll
                    for (int ft = 0; ft < KK * ICF; ++ft) {
                        float* pp = &red[(oc * ICW + (ft / KK) * 16 + sl) * KK + ft % KK];
                        *pp = (w == 0 ? 0.f : *pp) + acc[m][ft][r];
                    }
                    if (sl == 0) {
                        float* pp = &red[MFW * 16 * ICW * KK + oc];
                        *pp = (w == 0 ? 0.f : *pp) + accb[m][r];
                    }
                }
            }
        }
    }
    __syncthreads();
    float* dst = partial + (((long long)gi * n_icg + icg) * n_ocg + ocg) * PER;
    for (int e = tid; e < PER; e += 256) dst[e] = red[e];
}

// 64 outputs x 4 G-slices per block: coalesced across outputs, 4-way parallel over G, fixed order
template <int KS, int MFW, int ICF = 1>
__global__ __launch_bounds__(256) void wgrad_mfma_reduce(const float* __restrict__ partial, float* __restrict__ dw,
                                                         float* __restrict__ db, int cin, int cout, int G, int n_icg,
                                                         int n_ocg, int accumulate) {
    constexpr int KK = KS * KS, ICW = 16 * ICF;
    constexpr int PER = MFW * 16 * ICW * KK + MFW * 16;
    __shared__ float red[4][64];
    const int total_w = cout * cin * KK;
    const int o_local = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + o_local;
    long long off = -1;
    if (idx < total_w) {
        const int tap = idx % KK, c = (idx / KK) % cin, o = idx / (KK * cin);
        const int icg = c / ICW, ic = c % ICW, ocg = o / (MFW * 16), oc = o % (MFW * 16);
        off = ((long long)icg * n_ocg + ocg) * PER + (oc * ICW + ic) * KK + tap;
    } else if (idx < total_w + cout) {
        const int o = idx - total_w;
        const int ocg = o / (MFW * 16), oc = o % (MFW * 16);
        off = ((long long)0 * n_ocg + ocg) * PER + MFW * 16 * ICW * KK + oc;
    }
    float s = 0.f;
    if (off >= 0) {
        const long long stride = (long long)n_icg * n_ocg * PER;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // 4 independent chains keep 4+ loads in flight
        int gi = slice;
        for (; gi + 12 < G; gi += 16) {
            s0 += partial[gi * stride + off];
            s1 += partial[(gi + 4) * stride + off];
            s2 += partial[(gi + 8) * stride + off];
            s3 += partial[(gi + 12) * stride + off];
        }
        for (; gi < G; gi += 4) s0 += partial[gi * stride + off];
        s = (s0 + s1) + (s2 + s3);
    }
    red[slice][o_local] = s;
    __syncthreads();
    if (slice == 0 && off >= 0) {
        const float t = (red[0][o_local] + red[1][o_local]) + (red[2][o_local] + red[3][o_local]);
        if (idx < total_w) dw[idx] = accumulate ? dw[idx] + t : t;
        else if (db != nullptr) db[idx - total_w] = accumulate ? db[idx - total_w] + t : t;
    }
}

// ------------------------------------------------------------------ DMA-staged wgrad (3x3, 64 input x 64 output channels per block)
// Same maths and the same transpose-read fragments as wgrad_mfma_kernel, restructured like conv_dma_kernel: the register-staged
// kernel above re-stages the 64-channel g tile for every 16 input channels (135 staged bytes per MFMA); here a block owns a
// 64 x 64 channel pair, so one x tile (8 planes) + one g tile (8 planes) = 74 KB feed 8 consumer waves x 8 k-steps x 18 MFMAs
// (32 bytes per MFMA).  4 loader waves stage tile t+1 by LDS-DMA while the consumers run tile t (double-buffered, one barrier
// per tile); consumer wave w accumulates dW[out 32*(w>>2) .. +32][in 16*(w&3) .. +16][9 taps] over all of the block's tiles in
// registers (no K split, so no cross-wave reduction) and writes one partial per block at the end.
// Channel counts that are not multiples of 64 run with a ragged last group (used when >= 60 % of the 64 x 64 pairs is real work);
// requires a FOLDED halo-1 gradient (its zero halo ring pads ragged tiles).
constexpr int WD_XG = 8 * WG_XPL;                          // x-tile granules (8 planes of 18x18)
constexpr int WD_GG = 8 * WG_GPL;                          // g-tile granules (8 planes of 16x16, stride 260)
constexpr int WD_XPIECES = (WD_XG + 63) / 64;              // 41
constexpr int WD_GPIECES = (WD_GG + 63) / 64;              // 33
constexpr int WD_PIECES = WD_XPIECES + WD_GPIECES;         // 74
constexpr int WD_BUF_BYTES = WD_PIECES * 1024;             // 75776
constexpr int WDL_ITERS = (WD_PIECES + D_LOAD - 1) / D_LOAD;   // 19 pieces per loader wave per tile
constexpr int WD_PER = 64 * 64 * 9 + 64;                   // floats per block partial: dW[64 oc][64 ic][9], db[64]

__global__ __launch_bounds__((D_CONS + D_LOAD) * 64, 3) void wgrad_dma_kernel(TV tx, TV tg, float* __restrict__ partial, int tiles_x,
                                                                               int tpi, int total, int G, int n_icg, int n_ocg, SignMap sgn,
                                                                               int ragged_skip) {
    constexpr int TP = MT + 2;
    __shared__ __attribute__((aligned(16))) char s_buf[2 * WD_BUF_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int npairs = n_icg * n_ocg;
    const int b = blockIdx.x;
    int gi, pair;
    if ((G & 7) == 0) { pair = (b >> 3) % npairs; gi = ((b >> 3) / npairs) * 8 + (b & 7); }   // blocks sharing tiles: same XCD
    else { pair = b % npairs; gi = b / npairs; }
    const int icg = pair % n_icg, ocg = pair / n_icg;
    const TileWalk tw = xcd_walk(total, G, gi);   // (gi % 8 = the block's XCD when G is a multiple of 8)
    const int ntile = tw.count;

    if (wave >= D_CONS) {
        // ---------------- loader waves ----------------
        const int lw = wave - D_CONS;
        unsigned geo[WDL_ITERS];   // item independent: plane << 16 | tile row << 8 | tile col of the granule this lane stages
#pragma unroll
        for (int i = 0; i < WDL_ITERS; ++i) {
            const int P = lw + D_LOAD * i;
            if (P < WD_XPIECES) {
                const int slot = min(P * 64 + lane, WD_XG - 1);
                const int pl = slot / WG_XPL, p = min(slot - pl * WG_XPL, TP * TP - 1);
                geo[i] = (unsigned)(pl << 16 | (p / TP) << 8 | (p % TP));
            } else {
                const int slot = min((P - WD_XPIECES) * 64 + lane, WD_GG - 1);
                const int pl = slot / WG_GPL, p = min(slot - pl * WG_GPL, MT * MT - 1);
                geo[i] = (unsigned)(pl << 16 | (p / MT) << 8 | (p % MT));
            }
        }
        const unsigned xplane = (unsigned)(tx.plane * 16), gplane = (unsigned)(tg.plane * 16);
        // per piece, tile independent: byte offset of its plane (ragged last channel group: planes past the tensor re-read its last
        // plane -- those dW rows / columns are never reduced).  The per-tile part is one 24-bit multiply-add per piece.
        unsigned pl_off[WDL_ITERS];
        {
            const int xpl_max0 = min(8, tx.cb - icg * 8) - 1, gpl_max0 = min(8, tg.cb - ocg * 8) - 1;
#pragma unroll
            for (int i = 0; i < WDL_ITERS; ++i) {
                const int P = lw + D_LOAD * i, pl = (int)(geo[i] >> 16);
                pl_off[i] = P < WD_XPIECES ? (unsigned)min(pl, xpl_max0) * xplane : (unsigned)min(pl, gpl_max0) * gplane;
            }
        }
        // Ragged last channel group (round 6; NestFuse: 152 = 64 + 64 + 24 output, 304 = 4 x 64 + 48 input channels): the pieces that hold
        // only planes past the tensor are not requested at all -- the kernel runs at the rate its tiles are staged (74 KB per tile and block,
        // ~4.4 TB/s over the chip, measured round 6), and those pieces used to re-read the last real plane.  The consumers multiply whatever
        // the slots hold: those dW rows / columns and db entries are never reduced.  ($MMIF_WGRAD_RAGGED=0 stages them as before)
        const int px_end = ragged_skip ? (min(8, tx.cb - icg * 8) * WG_XPL + 63) / 64 : WD_XPIECES;
        const int pg_end = WD_XPIECES + (ragged_skip ? (min(8, tg.cb - ocg * 8) * WG_GPL + 63) / 64 : WD_GPIECES);
        auto issue = [&](int tile, int buf) {
            const int in_ = tile / tpi, tt = tile - in_ * tpi;
            const int y0 = (tt / tiles_x) * MT, x0 = (tt % tiles_x) * MT;
            const char* src_x = tx.base + ((long long)in_ * tx.img + (long long)(tx.cb_off + icg * 8) * tx.plane) * 16;
            const char* src_g = tg.base + ((long long)in_ * tg.img + (long long)(tg.cb_off + ocg * 8) * tg.plane) * 16;
            char* dst = s_buf + buf * WD_BUF_BYTES;
#pragma unroll
            for (int i = 0; i < WDL_ITERS; ++i) {
                const int P = lw + D_LOAD * i;   // wave uniform
                const int py = (int)((geo[i] >> 8) & 255u), px = (int)(geo[i] & 255u);
                if (P >= px_end && (P < WD_XPIECES || P >= pg_end)) continue;
                if (P < WD_XPIECES) {
                    const int y = min(max(reflect_idx(y0 + py - 1, tx.h), 0), tx.h - 1);
                    const int x = min(max(reflect_idx(x0 + px - 1, tx.w), 0), tx.w - 1);
                    const unsigned off = (__umul24((unsigned)y, (unsigned)tx.ws) + (unsigned)x) * 16u + pl_off[i];
                    __builtin_amdgcn_global_load_lds(MMIF_GPTR(src_x + off), MMIF_LPTR(dst + P * 1024), 16, 0, 0);
                } else if (P < WD_PIECES) {
                    // pixels of a ragged tile that lie outside the image read the zeroed halo ring (stored row h+1 / col w+1)
                    const int y = min(y0 + py, tg.h) + 1, x = min(x0 + px, tg.w) + 1;
                    const unsigned off = (__umul24((unsigned)y, (unsigned)tg.ws) + (unsigned)x) * 16u + pl_off[i];
                    __builtin_amdgcn_global_load_lds(MMIF_GPTR(src_g + off), MMIF_LPTR(dst + P * 1024), 16, 0, 0);
                }
            }
        };
        // ReLU sign bytes of the activation tile (struct SignMap; the layer's dgrad reads them instead of the activations), by the
        // blocks of the first output-channel group: loader thread t owns 8 adjacent pixels of one plane -- (plane t >> 5, tile row
        // (t & 31) >> 1, columns 8 (t & 1) .. + 7) -- 8 LDS granules in, ONE 8-byte store out.  The LDS reads are asm: the compiler
        // orders a C++ read of this buffer after ALL outstanding LDS-DMA (vmcnt(0)), i.e. after the NEXT tile's staging has landed.
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        auto emit_signs = [&](int tile, int buf) {
            const int in_ = tile / tpi, tt = tile - in_ * tpi;
            const int y0 = (tt / tiles_x) * MT, x0 = (tt % tiles_x) * MT;
            const int t = lw * 64 + lane, pl = t >> 5, py = (t & 31) >> 1, px0 = 8 * (t & 1);
            if (y0 + py >= tx.h || icg * 8 + pl >= tx.cb) return;
            const unsigned lds0 = (unsigned)(size_t)(s_buf + buf * WD_BUF_BYTES) + (unsigned)((pl * WG_XPL + (py + 1) * TP + px0 + 1) * 16);
            u32x4_t v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) __asm__ volatile("ds_read_b128 %0, %1" : "=v"(v[i]) : "v"(lds0 + 16u * i));
            __asm__ volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
            const unsigned one2 = 0x00010001u;
            unsigned out[2] = {0u, 0u};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                unsigned r = 0;
#pragma unroll
                for (int d = 0; d < 4; ++d) {   // per dword: bit 0 = low bf16 > 0, bit 16 = high bf16 > 0 (positive as int16)
                    unsigned q;
                    __asm__("v_pk_min_i16 %0, %1, %2\n\tv_pk_max_i16 %0, %0, 0" : "=&v"(q) : "v"(v[i][d]), "v"(one2));
                    r |= q << (2 * d);
                }
                out[i >> 2] |= ((r | (r >> 15)) & 0xffu) << (8 * (i & 3));
            }
            unsigned char* dst = sgn.p + (((long long)in_ * sgn.cb + icg * 8 + pl) * sgn.hs + y0 + py + 1) * sgn.pitch + x0 + px0 + 1 + SIGN_XOFF;
            *reinterpret_cast<uint2*>(dst) = make_uint2(out[0], out[1]);   // (columns past a ragged image edge land in the row's padding)
        };
        const bool signs = sgn.p != nullptr && ocg == 0;
        if (ntile > 0) issue(tw.first, 0);
        for (int k = 0; k < ntile; ++k) {
            __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's pieces of tile k have landed
            __builtin_amdgcn_s_barrier();
            if (k + 1 < ntile) issue(tw.first + (k + 1) * tw.stride, (k & 1) ^ 1);
            if (signs) emit_signs(tw.first + k * tw.stride, k & 1);
        }
        return;
    }

    // ---------------- consumer waves ----------------
    const int sl = lane & 15, g = lane >> 4;
    const int icf = wave & 3, mp = wave >> 2;   // input-channel fragment (16 ch), output-channel half (2 fragments of 16)
    f32x4 acc[2][9], accb[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        accb[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const uint4 ones_u = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_u);
    // per-lane transpose-read addressing (see wgrad_mfma_kernel): in-group lane sl supplies pixel row (sl>>2), chunk (sl&3)
    const int tr_row = sl >> 2, tr_c = sl & 3;
    const int lane_plane = tr_c >> 1, lane_byte = (tr_c & 1) * 8;
    for (int k = 0; k < ntile; ++k) {
        __builtin_amdgcn_s_barrier();
        const char* s_x = s_buf + (k & 1) * WD_BUF_BYTES;
        const char* s_g = s_x + WD_XPIECES * 1024;
        // K = pixels.  k-step s2 covers tile rows s2 (lane groups g = 0, 1: columns 0-7 / 8-15) and s2 + 8 (g = 2, 3), so the
        // activation fragment of tap (u, v) at step s2 -- rows s2 + u and s2 + 8 + u -- IS the fragment of tap (0, v) at step s2 + u:
        // a ring of three rows, ONE new row (3 fragments = 6 transposing reads) per k-step + the 4 reads of the two gradient
        // fragments = 10 reads of 512 B per 18 MFMAs.  (With rows 2 s2, 2 s2 + 1 per step and every tap re-read it was 22 -- 156 B / clk
        // on four SIMDs against the 128 B / clk of the LDS: decode.0's wgrad 0.55 ms; carrying only the u = 2 row 16: 0.45 ms.)
        const int rsel = 8 * (g >> 1), col0 = 8 * (g & 1);
        auto ld_x = [&](int xrow, int v) {
            const char* base = s_x + (((2 * icf + lane_plane) * WG_XPL) + (xrow + rsel) * TP + col0 + v + tr_row) * 16 + lane_byte;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, base));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, base + 4 * 16));
            return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        auto ld_g = [&](int grow, int m) {
            const char* base = s_g + (((2 * (mp * 2 + m) + lane_plane) * WG_GPL) + (grow + rsel) * MT + col0 + tr_row) * 16 + lane_byte;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, base));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, base + 4 * 16));
            return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        bf16x8 xr[3][3], a[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int v = 0; v < 3; ++v) xr[r][v] = ld_x(r, v);
#pragma unroll
        for (int m = 0; m < 2; ++m) a[0][m] = ld_g(0, m);
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
#pragma unroll
            for (int v = 0; v < 3; ++v) xr[(s2 + 2) % 3][v] = ld_x(s2 + 2, v);   // used by this step's last three taps
            if (s2 + 1 < 8) {
#pragma unroll
                for (int m = 0; m < 2; ++m) a[(s2 + 1) & 1][m] = ld_g(s2 + 1, m);
            }
            if (icf == 0) {
#pragma unroll
                for (int m = 0; m < 2; ++m) accb[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s2 & 1][m], ones, accb[m], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int u = t / 3, v = t % 3;
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s2 & 1][m], xr[(s2 + u) % 3][v], acc[m][t], 0, 0, 0);
            }
        }
    }
    // lane (g, sl) reg r of fragment m holds (oc = 16*(2*mp + m) + 4g + r, ic = 16*icf + sl) for tap t; bias sums: column sl == 0
    float* dst = partial + ((long long)gi * npairs + pair) * WD_PER;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oc = 16 * (2 * mp + m) + 4 * g + r, ic = 16 * icf + sl;
#pragma unroll
            for (int t = 0; t < 9; ++t) dst[(oc * 64 + ic) * 9 + t] = acc[m][t][r];
            if (icf == 0 && sl == 0) dst[64 * 64 * 9 + oc] = accb[m][r];
        }
}

// dw / db = fixed-order sum of the G block partials of each (icg, ocg) pair
template <int SL>
__global__ __launch_bounds__(64 * SL) void wgrad_dma_reduce(const float* __restrict__ partial, float* __restrict__ dw, float* __restrict__ db,
                                                        int cin, int cout, int G, int n_icg, int n_ocg, int accumulate) {
    __shared__ float red[SL][64];
    const int total_w = cout * cin * 9;
    const int idx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int npairs = n_icg * n_ocg;
    long long off = -1;
    if (idx < total_w) {
        const int tap = idx % 9, c = (idx / 9) % cin, o = idx / (9 * cin);
        off = (long long)((c / 64) + n_icg * (o / 64)) * WD_PER + ((o % 64) * 64 + (c % 64)) * 9 + tap;
    } else if (idx < total_w + cout) {
        const int o = idx - total_w;
        off = (long long)(0 + n_icg * (o / 64)) * WD_PER + 64 * 64 * 9 + (o % 64);
    }
    const float t = partial_sum<SL>(partial, off, (long long)npairs * WD_PER, G, off >= 0, red);
    if ((threadIdx.x >> 6) == 0 && off >= 0) {
        if (idx < total_w) dw[idx] = accumulate ? dw[idx] + t : t;
        else if (db != nullptr) db[idx - total_w] = accumulate ? db[idx - total_w] + t : t;
    }
}

// ------------------------------------------------------------------ host side
static long long* g_trace = nullptr;  // device buffer [1024][64] for the optional phase trace
// ------------------------------------------------------------------ fused backward of ONE thin 3x3 layer: dgrad + wgrad ("bwd pair")
// decode.2 (64 -> 32) and decode.3 (32 -> 16) of every PFNet / DenseFuse decoder: their dgrad (read g + the ReLU-mask activations x, write
// gx) and their wgrad (read x + g again) move 256 / 128 planes for 160 / 80 of data.  Here a block stages ONE 18 x 18 tile of g (zero
// ring) and of x (reflect halo) and both products come out of it:
//   waves 0..3  dgrad of tile rows 4w .. 4w+3 over all 16 NXB input channels (MFMA, the thin kernel's k-group order and fold steps =>
//               bit-identical to it), ReLU mask from the x tile already in LDS, gx stored in the folded convention (interior only);
//   waves 4..6  weight gradient, tap row u = wave - 4 (the tap-row scheme of enc_wgrad.hip); wave 7 the bias sums.
// The dgrad operand image (<= 36 KB) is resident in LDS; per-block wgrad partials in natural order, fixed-order reduction.
constexpr int BP_PL = 324;   // granules per staged plane (18 x 18)
template <int NXB, int NGB>
__global__ __launch_bounds__(512, NXB == 4 ? 1 : 2) void bwd_pair_kernel(TV tx, TV tg, TV tgx, const uint4* __restrict__ wpk, float* __restrict__ partial,
                                                                         int tiles_x, int tpi, int total, int G) {
    constexpr int MF = NXB, TP = MT + 2;
    constexpr int CIN = 16 * NXB, COUT = 16 * NGB;
    constexpr int NCB = 2 * NGB;                          // channel blocks of g = K chunk of the dgrad (<= 4: one chunk)
    constexpr int NKG = 9 * NCB, NKGP = (NKG + 3) / 4 * 4;
    constexpr int WBYTES = NKGP * MF * 256;
    constexpr int PER = COUT * CIN * 9 + COUT;
    constexpr int TILE_BYTES = (2 * NXB + NCB) * BP_PL * 16 + WBYTES;
    constexpr int SM_BYTES = TILE_BYTES > PER * 4 ? TILE_BYTES : PER * 4;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    __shared__ __attribute__((aligned(16))) char smem[SM_BYTES];
    __shared__ int2 s_tab[NKGP];
    u32x4* s_x = reinterpret_cast<u32x4*>(smem);
    u32x4* s_g = s_x + 2 * NXB * BP_PL;
    char* s_w = reinterpret_cast<char*>(s_g + NCB * BP_PL);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int gi = blockIdx.x;
    const int H = tx.h, W = tx.w;
    // resident dgrad operand image + k-group table (.x byte offset into a g plane set, .y into the image)
    for (int e = tid; e < WBYTES / 16; e += 512) reinterpret_cast<uint4*>(s_w)[e] = wpk[e];
    if (tid < NKGP) {
        int tap = 0, cb = 0, plane = tid;
        if (tid < NKG) { tap = visit_tap(tid / NCB, 3); cb = tid % NCB; plane = tap * NCB + cb; }
        s_tab[tid] = make_int2((cb * BP_PL + (tap / 3) * TP + (tap % 3)) * 16, plane * MF * 256);
    }
    const bf16x8 ones = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));
    const int tr_row = j >> 2, tr_c = j & 3;
    const int lane_plane = tr_c >> 1, lane_byte = (tr_c & 1) * 8;
    auto tr_frag = [&](const char* base) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, base));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, base + 4 * 16));
        const s16x8 c = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, c);
    };
    constexpr int NXR = (2 * NXB * BP_PL + 511) / 512, NGR = (NCB * BP_PL + 511) / 512;
    u32x4 rx[NXR], rg[NGR];
    // tile-independent part of every staged granule's address, packed once (channel block << 16 | tile row << 8 | tile column);
    // 32-bit offsets per tile (see enc_wgrad_kernel: e / 324, e % 18 and 64-bit products per load were a phase of their own)
    unsigned xpk[NXR], gpk[NGR];
#pragma unroll
    for (int i = 0; i < NXR; ++i) {
        const int e = min(tid + 512 * i, 2 * NXB * BP_PL - 1);
        const int cb = e / BP_PL, p = e - cb * BP_PL;
        xpk[i] = (unsigned)(cb << 16 | (p / TP) << 8 | (p % TP));
    }
#pragma unroll
    for (int i = 0; i < NGR; ++i) {
        const int e = min(tid + 512 * i, NCB * BP_PL - 1);
        const int cb = e / BP_PL, p = e - cb * BP_PL;
        gpk[i] = (unsigned)(cb << 16 | (p / TP) << 8 | (p % TP));
    }
    const unsigned xplane = (unsigned)tx.plane, gplane = (unsigned)tg.plane;   // (conv_api.hip: an image stays below 2^31 granules)
    const unsigned xcb0 = (unsigned)tx.cb_off * xplane, gcb0 = (unsigned)tg.cb_off * gplane;
    auto prefetch = [&](int tile) {
        const int in_ = tile / tpi, tt = tile - in_ * tpi;
        const int y0 = (tt / tiles_x) * MT, x0 = (tt % tiles_x) * MT;
        const char* xb = tx.base + (long long)in_ * tx.img * 16;
        const char* gb = tg.base + (long long)in_ * tg.img * 16;
#pragma unroll
        for (int i = 0; i < NXR; ++i) {
            const int y = min(max(reflect_idx(y0 + (int)((xpk[i] >> 8) & 255u) - 1, H), 0), H - 1);
            const int x = min(max(reflect_idx(x0 + (int)(xpk[i] & 255u) - 1, W), 0), W - 1);
            const unsigned off = xcb0 + (xpk[i] >> 16) * xplane + (unsigned)(y + tx.halo) * (unsigned)tx.ws + (unsigned)(x + tx.halo);
            rx[i] = *reinterpret_cast<const u32x4*>(xb + (unsigned long long)off * 16u);
        }
#pragma unroll
        for (int i = 0; i < NGR; ++i) {   // g in STORED coordinates (halo 1, folded: the ring is zero = the zero padding of the dgrad)
            const int ys = y0 + (int)((gpk[i] >> 8) & 255u), xs = x0 + (int)(gpk[i] & 255u);
            const bool inside = ys < tg.hs && xs < tg.ws;
            const unsigned off = gcb0 + (gpk[i] >> 16) * gplane + (unsigned)min(ys, tg.hs - 1) * (unsigned)tg.ws + (unsigned)min(xs, tg.ws - 1);
            const u32x4 v = *reinterpret_cast<const u32x4*>(gb + (unsigned long long)off * 16u);
            rg[i] = inside ? v : (u32x4){0u, 0u, 0u, 0u};
        }
    };
    // every wave runs this at the top of a tile: publish the prefetched tile, start the next prefetch (block-wide: two barriers)
    const TileWalk tw = xcd_walk(total, G, gi);
    auto stage = [&](int tile, int it) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NXR; ++i) {
            const int e = tid + 512 * i;
            if (e < 2 * NXB * BP_PL) s_x[e] = rx[i];
        }
#pragma unroll
        for (int i = 0; i < NGR; ++i) {
            const int e = tid + 512 * i;
            if (e < NCB * BP_PL) s_g[e] = rg[i];
        }
        __syncthreads();
        if (it + 1 < tw.count) prefetch(tile + tw.stride);
    };
    if (tw.count > 0) prefetch(tw.first);
    float* red = reinterpret_cast<float*>(smem);
    // The three roles keep separate tile loops (same trip count, same barriers): their accumulators never share a live range, so the
    // kernel needs max(role) registers, not the sum.
    if (wave < 4) {
        // ---------------- dgrad: rows 4 wave .. 4 wave + 3 of every tile, all MF m-fragments
        for (int it = 0, tile = tw.first; it < tw.count; ++it, tile += tw.stride) {
            stage(tile, it);
            const int in_ = tile / tpi, tt = tile - in_ * tpi;
            const int ty0 = tt / tiles_x, tx0 = tt % tiles_x;
            f32x4 acc[MF][4];
#pragma unroll
            for (int m = 0; m < MF; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const char* in_lane = reinterpret_cast<const char*>(s_g) + ((wave * 4) * TP + j) * 16;
            const char* w_lane = s_w + j * 16;
            int2 off = s_tab[g];
#pragma unroll 1
            for (int s = 0; s < NKGP / 4; ++s) {
                const int2 nxt = s_tab[min(4 * (s + 1), NKGP - 4) + g];
                bf16x8 a[MF], b[4];
#pragma unroll
                for (int m = 0; m < MF; ++m) a[m] = *reinterpret_cast<const bf16x8*>(w_lane + off.y + m * 256);
#pragma unroll
                for (int n = 0; n < 4; ++n) b[n] = *reinterpret_cast<const bf16x8*>(in_lane + off.x + n * TP * 16);
#pragma unroll
                for (int m = 0; m < MF; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
                off = nxt;
            }
            {
                const int gc = min(g, NCB - 1);
                dgrad_fold_steps<MF, 1>(acc, in_lane + gc * (BP_PL * 16), w_lane + gc * (MF * 256), NCB, TP, g, j, ty0 * MT + wave * 4, tx0 * MT, tgx.hs, tgx.ws);
            }
            // epilogue: pair rows -> one granule per lane, ReLU mask from the x tile in LDS, store the interior of the folded gradient
            const int oxs = 1 + tx0 * MT + j;
#pragma unroll
            for (int m = 0; m < MF; ++m) {
                const int ocb = 2 * m + (g >> 1);
                char* oplane = tgx.base + ((long long)in_ * tgx.img + (long long)(tgx.cb_off + ocb) * tgx.plane) * 16;
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {
                    float c[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[m][2 * p2][r]), __float_as_uint(acc[m][2 * p2 + 1][r]), false, false);
                        c[r] = __uint_as_float(sw[0]);
                        c[4 + r] = __uint_as_float(sw[1]);
                    }
                    const int row = wave * 4 + (g & 1) + 2 * p2;
                    const int oys = 1 + ty0 * MT + row;
                    if (oys >= tgx.hs - 1 || oxs >= tgx.ws - 1) continue;
                    const u32x4 xm = s_x[ocb * BP_PL + (row + 1) * TP + j + 1];
                    const uint32_t xw[4] = {xm.x, xm.y, xm.z, xm.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (!((short)(xw[i] & 0xffffu) > 0)) c[2 * i] = 0.f;
                        if (!((short)(xw[i] >> 16) > 0)) c[2 * i + 1] = 0.f;
                    }
                    *reinterpret_cast<uint4*>(oplane + ((long long)oys * tgx.ws + oxs) * 16) =
                        make_uint4(pack_bf16x2(c[0], c[1]), pack_bf16x2(c[2], c[3]), pack_bf16x2(c[4], c[5]), pack_bf16x2(c[6], c[7]));
                }
            }
        }
        __syncthreads();
    } else if (wave < 7) {
        // ---------------- wgrad, tap row u: g centre (tile rows 1..16, cols 1..16), x shifted by (u, v)
        const int u = wave - 4;
        f32x4 wacc[3][NXB][NGB];
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int b = 0; b < NXB; ++b)
#pragma unroll
                for (int m = 0; m < NGB; ++m) wacc[v][b][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0, tile = tw.first; it < tw.count; ++it, tile += tw.stride) {
            stage(tile, it);
#pragma unroll 2
            for (int s = 0; s < 8; ++s) {
                const int row = 2 * s + (g >> 1), col0 = 8 * (g & 1);
                bf16x8 a[NGB];
#pragma unroll
                for (int m = 0; m < NGB; ++m)
                    a[m] = tr_frag(reinterpret_cast<const char*>(s_g) + ((2 * m + lane_plane) * BP_PL + (row + 1) * TP + col0 + 1 + tr_row) * 16 + lane_byte);
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    bf16x8 bx[NXB];
#pragma unroll
                    for (int b = 0; b < NXB; ++b)
                        bx[b] = tr_frag(reinterpret_cast<const char*>(s_x) + ((2 * b + lane_plane) * BP_PL + (row + u) * TP + col0 + v + tr_row) * 16 + lane_byte);
#pragma unroll
                    for (int b = 0; b < NXB; ++b)
#pragma unroll
                        for (int m = 0; m < NGB; ++m) wacc[v][b][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], bx[b], wacc[v][b][m], 0, 0, 0);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int b = 0; b < NXB; ++b)
#pragma unroll
                for (int m = 0; m < NGB; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[((16 * m + 4 * g + r) * CIN + 16 * b + j) * 9 + 3 * u + v] = wacc[v][b][m][r];
    } else {
        f32x4 accb[NGB];
#pragma unroll
        for (int m = 0; m < NGB; ++m) accb[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0, tile = tw.first; it < tw.count; ++it, tile += tw.stride) {
            stage(tile, it);
#pragma unroll 2
            for (int s = 0; s < 8; ++s) {
                const int row = 2 * s + (g >> 1), col0 = 8 * (g & 1);
#pragma unroll
                for (int m = 0; m < NGB; ++m) {
                    const bf16x8 a = tr_frag(reinterpret_cast<const char*>(s_g) + ((2 * m + lane_plane) * BP_PL + (row + 1) * TP + col0 + 1 + tr_row) * 16 + lane_byte);
                    accb[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, ones, accb[m], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (j == 0) {
#pragma unroll
            for (int m = 0; m < NGB; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[COUT * CIN * 9 + 16 * m + 4 * g + r] = accb[m][r];
        }
    }
    __syncthreads();
    float* dst = partial + (long long)gi * PER;
    for (int e = tid; e < PER; e += 512) dst[e] = red[e];
}

// ---- round 4: the same kernel with its tiles staged by LDS-DMA.  bwd_pair_kernel fetches the next tile into registers (9 granules per
// thread), and every tile pays two block barriers around the register -> LDS copy with no MFMA running (SQ counters: the matrix pipe busy
// ~40 % of the launch, 12.9 k cycles per tile against 5.4 k of MFMAs on the busiest SIMD).  Here wave 7 is a LOADER: it requests tile
// k + 1 straight into the other half of a double-buffered LDS tile (global_load_lds, 61 / 31 one-KiB pieces) while waves 0..6 work on tile
// k -- ONE barrier per tile, no staging registers.  (The loader must be a wave that reads no LDS: with DMA in flight the compiler drains
// vmcnt before any LDS access of that wave.)  The bias sums moved into the tap-row-1 wave, which already holds the transposed gradient
// fragments (same MFMA(a, ones) sequence => bit-identical); dgrad / wgrad k-loops, fold steps and epilogue are bwd_pair_kernel's.
// LDS: 2 x 61 KiB + 36 KiB image = 158 KiB (<4, 2>: one block per CU), 2 x 31 KiB + 10 KiB (<2, 1>: two blocks per CU).
template <int NXB, int NGB>
__global__ __launch_bounds__(512, NXB == 4 ? 1 : 2) void bwd_pair_dma_kernel(TV tx, TV tg, TV tgx, const uint4* __restrict__ wpk, float* __restrict__ partial,
                                                                             int tiles_x, int tpi, int total, int G, int abl) {
    constexpr int MF = NXB, TP = MT + 2;
    constexpr int CIN = 16 * NXB, COUT = 16 * NGB;
    constexpr int NCB = 2 * NGB;
    constexpr int NKG = 9 * NCB, NKGP = (NKG + 3) / 4 * 4;
    constexpr int WBYTES = NKGP * MF * 256;
    constexpr int PER = COUT * CIN * 9 + COUT;
    constexpr int NX = 2 * NXB * BP_PL, NT = NX + NCB * BP_PL;   // granules of the x planes / of a whole tile (x planes, then g planes)
    constexpr int NP = (NT + 63) / 64;                            // DMA pieces per tile
    constexpr int BUF_BYTES = NP * 1024;
    constexpr int TILE_BYTES = 2 * BUF_BYTES + WBYTES;
    constexpr int SM_BYTES = TILE_BYTES > PER * 4 ? TILE_BYTES : PER * 4;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    __shared__ __attribute__((aligned(16))) char smem[SM_BYTES];
    __shared__ int2 s_tab[NKGP];
    char* s_w = smem + 2 * BUF_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int gi = blockIdx.x;
    const int H = tx.h, W = tx.w;
    for (int e = tid; e < WBYTES / 16; e += 512) reinterpret_cast<uint4*>(s_w)[e] = wpk[e];
    if (tid < NKGP) {
        int tap = 0, cb = 0, plane = tid;
        if (tid < NKG) { tap = visit_tap(tid / NCB, 3); cb = tid % NCB; plane = tap * NCB + cb; }
        s_tab[tid] = make_int2((cb * BP_PL + (tap / 3) * TP + (tap % 3)) * 16, plane * MF * 256);
    }
    const TileWalk tw = xcd_walk(total, G, gi);
    float* red = reinterpret_cast<float*>(smem);
    if (wave == 7) {
        // ---------------- loader: tile-independent part of every granule this lane requests (plane << 16 | tile row << 8 | tile column)
        unsigned pk[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int e = min(i * 64 + lane, NT - 1);   // (the last piece's spare lanes re-request the last granule into the buffer's padding)
            const int cb = e / BP_PL, p = e - cb * BP_PL;
            pk[i] = (unsigned)(cb << 16 | (p / TP) << 8 | (p % TP));
        }
        const unsigned xplane = (unsigned)tx.plane, gplane = (unsigned)tg.plane;
        const unsigned xcb0 = (unsigned)tx.cb_off * xplane, gcb0 = (unsigned)tg.cb_off * gplane;
        auto issue = [&](int tile, int sel) {
            const int in_ = tile / tpi, tt = tile - in_ * tpi;
            const int y0 = (tt / tiles_x) * MT, x0 = (tt % tiles_x) * MT;
            const char* xb = tx.base + (long long)in_ * tx.img * 16;
            const char* gb = tg.base + (long long)in_ * tg.img * 16;
            char* dst = smem + sel * BUF_BYTES;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const bool all_x = i * 64 + 63 < NX, all_g = i * 64 >= NX;
                const unsigned cb = pk[i] >> 16;
                const int r = (int)((pk[i] >> 8) & 255u), c = (int)(pk[i] & 255u);
                unsigned offx = 0u, offg = 0u;
                if (!all_g) {   // x: reflect halo
                    const int y = min(max(reflect_idx(y0 + r - 1, H), 0), H - 1);
                    const int x = min(max(reflect_idx(x0 + c - 1, W), 0), W - 1);
                    offx = xcb0 + cb * xplane + (unsigned)(y + tx.halo) * (unsigned)tx.ws + (unsigned)(x + tx.halo);
                }
                if (!all_x) {   // g in STORED coordinates (halo 1, folded): rows / columns past the stored domain read its zero ring
                    const int ys = min(y0 + r, tg.hs - 1), xs = min(x0 + c, tg.ws - 1);
                    offg = gcb0 + (cb - (unsigned)(2 * NXB)) * gplane + (unsigned)ys * (unsigned)tg.ws + (unsigned)xs;
                }
                const char* src = all_x ? xb + (unsigned long long)offx * 16u
                                        : (all_g ? gb + (unsigned long long)offg * 16u
                                                 : (cb < (unsigned)(2 * NXB) ? xb + (unsigned long long)offx * 16u : gb + (unsigned long long)offg * 16u));
                __builtin_amdgcn_global_load_lds(MMIF_GPTR(src), MMIF_LPTR(dst + i * 1024), 16, 0, 0);
            }
        };
        if (tw.count > 0) issue(tw.first, 0);
        for (int it = 0, tile = tw.first; it < tw.count; ++it, tile += tw.stride) {
            __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): tile `it` has landed
            __syncthreads();                      // ... and everybody is done with tile it - 1 (the other buffer)
            if (it + 1 < tw.count && !(abl & 1)) issue(tile + tw.stride, (it + 1) & 1);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
    } else if (wave < 4) {
        // ---------------- dgrad: rows 4 wave .. 4 wave + 3 of every tile, all MF m-fragments
        for (int it = 0, tile = tw.first; it < tw.count; ++it, tile += tw.stride) {
            __syncthreads();
            const u32x4* s_x = reinterpret_cast<const u32x4*>(smem + (it & 1) * BUF_BYTES);
            const u32x4* s_g = s_x + NX;
            const int in_ = tile / tpi, tt = tile - in_ * tpi;
            const int ty0 = tt / tiles_x, tx0 = tt % tiles_x;
            f32x4 acc[MF][4];
#pragma unroll
            for (int m = 0; m < MF; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const char* in_lane = reinterpret_cast<const char*>(s_g) + ((wave * 4) * TP + j) * 16;
            const char* w_lane = s_w + j * 16;
            // k-loop, software pipelined: the fragments of k-step s + 1 are requested before the MFMAs of step s (two waves per SIMD do not
            // hide an LDS round trip per step: the un-pipelined loop ran the matrix pipe 44 % of its time)
            constexpr int NS = NKGP / 4;
            bf16x8 a[2][MF], b[2][4];
            {
                const int2 off = s_tab[g];
#pragma unroll
                for (int m = 0; m < MF; ++m) a[0][m] = *reinterpret_cast<const bf16x8*>(w_lane + off.y + m * 256);
#pragma unroll
                for (int n = 0; n < 4; ++n) b[0][n] = *reinterpret_cast<const bf16x8*>(in_lane + off.x + n * TP * 16);
            }
            if (!(abl & 8))
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (s + 1 < NS) {
                    const int2 off = s_tab[4 * (s + 1) + g];
#pragma unroll
                    for (int m = 0; m < MF; ++m) a[(s + 1) & 1][m] = *reinterpret_cast<const bf16x8*>(w_lane + off.y + m * 256);
#pragma unroll
                    for (int n = 0; n < 4; ++n) b[(s + 1) & 1][n] = *reinterpret_cast<const bf16x8*>(in_lane + off.x + n * TP * 16);
                }
#pragma unroll
                for (int m = 0; m < MF; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s & 1][m], b[s & 1][n], acc[m][n], 0, 0, 0);
            }
            if (!(abl & 32)) {
                const int gc = min(g, NCB - 1);
                dgrad_fold_steps<MF, 1>(acc, in_lane + gc * (BP_PL * 16), w_lane + gc * (MF * 256), NCB, TP, g, j, ty0 * MT + wave * 4, tx0 * MT, tgx.hs, tgx.ws);
            }
            const int oxs = 1 + tx0 * MT + j;
#pragma unroll
            for (int m = 0; m < MF; ++m) {
                if (abl & 16) break;
                const int ocb = 2 * m + (g >> 1);
                char* oplane = tgx.base + ((long long)in_ * tgx.img + (long long)(tgx.cb_off + ocb) * tgx.plane) * 16;
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {
                    float c[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[m][2 * p2][r]), __float_as_uint(acc[m][2 * p2 + 1][r]), false, false);
                        c[r] = __uint_as_float(sw[0]);
                        c[4 + r] = __uint_as_float(sw[1]);
                    }
                    const int row = wave * 4 + (g & 1) + 2 * p2;
                    const int oys = 1 + ty0 * MT + row;
                    if (oys >= tgx.hs - 1 || oxs >= tgx.ws - 1 || (abl & 2)) continue;
                    const u32x4 xm = s_x[ocb * BP_PL + (row + 1) * TP + j + 1];
                    const uint32_t xw[4] = {xm.x, xm.y, xm.z, xm.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (!((short)(xw[i] & 0xffffu) > 0)) c[2 * i] = 0.f;
                        if (!((short)(xw[i] >> 16) > 0)) c[2 * i + 1] = 0.f;
                    }
                    *reinterpret_cast<uint4*>(oplane + ((long long)oys * tgx.ws + oxs) * 16) =
                        make_uint4(pack_bf16x2(c[0], c[1]), pack_bf16x2(c[2], c[3]), pack_bf16x2(c[4], c[5]), pack_bf16x2(c[6], c[7]));
                }
            }
        }
        __syncthreads();
    } else {
        // ---------------- wgrad, tap row u (waves 4..6); the tap-row-1 wave also sums the bias gradient from its g fragments
        const int u = wave - 4;
        const bf16x8 ones = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));
        const int tr_row = j >> 2, tr_c = j & 3;
        const int lane_plane = tr_c >> 1, lane_byte = (tr_c & 1) * 8;
        auto tr_frag = [&](const char* base) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, base));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, base + 4 * 16));
            const s16x8 c = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            return __builtin_bit_cast(bf16x8, c);
        };
        f32x4 wacc[3][NXB][NGB], accb[NGB];
#pragma unroll
        for (int m = 0; m < NGB; ++m) accb[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int b = 0; b < NXB; ++b)
#pragma unroll
                for (int m = 0; m < NGB; ++m) wacc[v][b][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0, tile = tw.first; it < tw.count; ++it, tile += tw.stride) {
            __syncthreads();
            const char* s_x = smem + (it & 1) * BUF_BYTES;
            const char* s_g = s_x + NX * 16;
            // 24 groups (pixel-row pair s, tap column v) of NXB x NGB MFMAs, software pipelined: the x fragments of group i + 1 (and the g
            // fragments of the next row pair) are requested before the MFMAs of group i
            const char* xl = s_x + (lane_plane * BP_PL + ((g >> 1) + u) * TP + 8 * (g & 1) + tr_row) * 16 + lane_byte;
            const char* gl = s_g + (lane_plane * BP_PL + ((g >> 1) + 1) * TP + 8 * (g & 1) + 1 + tr_row) * 16 + lane_byte;
            bf16x8 a[2][NGB], bx[2][NXB];
#pragma unroll
            for (int m = 0; m < NGB; ++m) a[0][m] = tr_frag(gl + (2 * m * BP_PL) * 16);
#pragma unroll
            for (int b = 0; b < NXB; ++b) bx[0][b] = tr_frag(xl + (2 * b * BP_PL) * 16);
            if (!(abl & 4))
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                const int s = i / 3, v = i % 3;
                if (i + 1 < 24) {
                    const int s1 = (i + 1) / 3, v1 = (i + 1) % 3;
#pragma unroll
                    for (int b = 0; b < NXB; ++b) bx[(i + 1) & 1][b] = tr_frag(xl + (2 * b * BP_PL + 2 * s1 * TP + v1) * 16);
                    if (v1 == 0) {
#pragma unroll
                        for (int m = 0; m < NGB; ++m) a[s1 & 1][m] = tr_frag(gl + (2 * m * BP_PL + 2 * s1 * TP) * 16);
                    }
                }
                if (u == 1 && v == 0) {
#pragma unroll
                    for (int m = 0; m < NGB; ++m) accb[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s & 1][m], ones, accb[m], 0, 0, 0);
                }
#pragma unroll
                for (int b = 0; b < NXB; ++b)
#pragma unroll
                    for (int m = 0; m < NGB; ++m) wacc[v][b][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s & 1][m], bx[i & 1][b], wacc[v][b][m], 0, 0, 0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int b = 0; b < NXB; ++b)
#pragma unroll
                for (int m = 0; m < NGB; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[((16 * m + 4 * g + r) * CIN + 16 * b + j) * 9 + 3 * u + v] = wacc[v][b][m][r];
        if (u == 1 && j == 0) {
#pragma unroll
            for (int m = 0; m < NGB; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[COUT * CIN * 9 + 16 * m + 4 * g + r] = accb[m][r];
        }
    }
    __syncthreads();
    float* dst = partial + (long long)gi * PER;
    for (int e = tid; e < PER; e += 512) dst[e] = red[e];
}

int taprow_reduce_launch(const float* ws, float* dw, float* db, int cin, int cout, int G, int accumulate, hipStream_t st);   // enc_wgrad.hip
static int num_cus_();
constexpr int BP_MAXG = 512;
static int g_bwd_pair_dma = -1;
void debug_set_bwd_pair_dma(int mode) { g_bwd_pair_dma = mode ? 1 : 0; }
bool bwd_pair_supported(int ks, int cin, int cout) { return ks == 3 && ((cin == 64 && cout == 32) || (cin == 32 && cout == 16)); }
size_t bwd_pair_workspace(int cin, int cout) { return (size_t)BP_MAXG * ((size_t)cout * cin * 9 + cout) * sizeof(float); }
int bwd_pair(const TV& tx, const TV& tg, const TV& tgx, const void* wpk_dgrad, float* dw, float* db, int cin, int cout, int accumulate, float* ws,
             hipStream_t st) {
    const int tiles_x = cdiv(tx.w, MT), tiles_y = cdiv(tx.h, MT);
    const int tpi = tiles_x * tiles_y, total = tpi * tx.n;
    const int cap = (cin == 64 ? 1 : 2) * num_cus_();
    const int G = total < cap ? total : (cap < BP_MAXG ? cap : BP_MAXG);
    ws = defer_ws(ws, (size_t)G * ((size_t)cout * cin * 9 + cout) * sizeof(float));
    // mmif_debug_set_bwd_pair_dma(0): the register-staged kernel (the tests' cross-check; bit-identical results)
    if (g_bwd_pair_dma < 0) g_bwd_pair_dma = 1;
    const bool use_dma = g_bwd_pair_dma == 1 && tg.halo == 1 && tg.folded;
    static int bp_abl = -1;   // $MMIF_ABLATE bp= (timing ablations, wrong results): 1 no tile requests after the first, 2 no gx stores, 4 no wgrad loops, 8 no dgrad k-loops
    if (bp_abl < 0) bp_abl = ablate_env("bp");   // (the loader reads the gradient's zero ring for rows / columns past the image)
    if (cin == 64 && use_dma)
        hipLaunchKernelGGL((bwd_pair_dma_kernel<4, 2>), dim3(G), dim3(512), 0, st, tx, tg, tgx, (const uint4*)wpk_dgrad, ws, tiles_x, tpi, total, G, bp_abl);
    else if (cin == 64)
        hipLaunchKernelGGL((bwd_pair_kernel<4, 2>), dim3(G), dim3(512), 0, st, tx, tg, tgx, (const uint4*)wpk_dgrad, ws, tiles_x, tpi, total, G);
    else if (use_dma)
        hipLaunchKernelGGL((bwd_pair_dma_kernel<2, 1>), dim3(G), dim3(512), 0, st, tx, tg, tgx, (const uint4*)wpk_dgrad, ws, tiles_x, tpi, total, G, bp_abl);
    else
        hipLaunchKernelGGL((bwd_pair_kernel<2, 1>), dim3(G), dim3(512), 0, st, tx, tg, tgx, (const uint4*)wpk_dgrad, ws, tiles_x, tpi, total, G);
    if (int rc = check_launch("bwd_pair")) return rc;
    return taprow_reduce_launch(ws, dw, db, cin, cout, G, accumulate, st);
}

// ---- backward of a WIDE 3x3 layer (Cin, Cout multiples of 64: decode.0 / decode.1) as one call: wgrad_dma_kernel also leaves the
// ReLU sign bytes of the layer's input activations, conv_dma_kernel<true, 2> (dgrad, fused fold) reads those instead of the
// activations themselves -- for decode.1 (128 -> 64, every input block masked) that is 0.54 of the dgrad's 1.34 GB.
static int launch_wgrad_dma(const TV& tx, const TV& tg, float* dw, float* db, int cin, int cout, int accumulate, float* ws, hipStream_t st, SignMap sgn);
static int launch_conv_dma(bool dgrad, const TV& tin, const TV& tout, const TV& tmask, const void* wpk, const float* bias, int n_out, int relu,
                           uint64_t mask_bits, uint64_t accum_bits, int org, hipStream_t st, SignMap sgn = SignMap{nullptr, 0, 0, 0},
                           const DupOut* dup = nullptr);
static void init_modes();
bool bwd_wide_supported(int ks, int cin, int cout) { return ks == 3 && cin >= 64 && cout >= 64 && cin % 64 == 0 && cout % 64 == 0 && cin <= 512; }
size_t bwd_wide_signs_bytes(int n, int cin, int h, int w) { return (size_t)n * (cin / 8) * (h + 2) * sign_pitch(w); }
int bwd_wide(const TV& tx, const TV& tg, const TV& tgx, const void* wpk_dgrad, float* dw, float* db, int cin, int cout, uint64_t mask_bits,
             int accumulate, float* ws, unsigned char* signs, hipStream_t st, int phase) {
    init_modes();
    const SignMap sgn{signs, tx.cb, tx.h + 2, sign_pitch(tx.w)};
    if (phase != 2)   // (phase 1 / 2: only the weight-gradient / only the input-gradient half -- per-kernel timing, mmif.h)
        if (int rc = launch_wgrad_dma(tx, tg, dw, db, cin, cout, accumulate, ws, st, mask_bits != 0 ? sgn : SignMap{nullptr, 0, 0, 0})) return rc;
    if (phase == 1) return MMIF_OK;
    return launch_conv_dma(true, tg, tgx, tx, wpk_dgrad, nullptr, cin, 0, mask_bits, 0, 1, st, sgn);
}

bool conv_mfma_supported(bool dgrad, int ks, int cin, int cout) {
    (void)dgrad;
    return (ks == 1 || ks == 3) && cin >= 1 && cout >= 1;
}

template <int KS, int MF>
static int launch_conv_mfma(bool dgrad, const TV& tin, const TV& tout, const TV& tmask, const void* wpk, const float* bias,
                            int n_out, int relu, uint64_t mask_bits, uint64_t accum_bits, hipStream_t st) {
    const int tiles_x = cdiv(tout.ws, MT), tiles_y = cdiv(tout.hs, MT);
    const int nmb = n_mblocks(n_out);
    const int m16p = nmb * MF * 16;
    dim3 grid(tiles_x * tiles_y * tout.n * nmb);
    if (dgrad)
        hipLaunchKernelGGL((conv_mfma_kernel<KS, MF, true>), grid, dim3(256), 0, st, tin, tout, tmask, (const uint4*)wpk, bias,
                           n_out, m16p, relu, (unsigned long long)mask_bits, (unsigned long long)accum_bits, tiles_x, tiles_y, nmb, g_trace);
    else
        hipLaunchKernelGGL((conv_mfma_kernel<KS, MF, false>), grid, dim3(256), 0, st, tin, tout, tmask, (const uint4*)wpk, bias,
                           n_out, m16p, relu, (unsigned long long)mask_bits, (unsigned long long)accum_bits, tiles_x, tiles_y, nmb, g_trace);
    return check_launch(dgrad ? "conv_mfma dgrad" : "conv_mfma fwd");
}

static int num_cus_() { return cached_num_cus(); }

static int g_dma_mode = 1;    // mmif_debug_set_conv_dma: 1 (default) = DMA-staged kernel where it applies, 0 = never (the tests' cross-check)
static int g_num_cus = 0;
constexpr int g_fuse_fold = 1;   // the DMA-staged dgrads fold the reflect halo themselves (round 2's A/B switch is gone: +5 % on the step)
static int g_abl = 0;            // $MMIF_ABLATE conv= (diagnostics): bit 0 = no staging DMAs after the first chunk

static int g_wgrad_ragged = 1;    // mmif_debug_set_ragged(0): wgrad_dma_kernel stages the padded planes of a ragged channel group too (the tests' cross-check)
static void init_modes() {
    static bool done = false;
    if (done) return;
    done = true;
    g_abl = ablate_env("conv");      // (timing ablations, diagnostics only: tools/sweep_staging.sh)
}
static int launch_conv_dma(bool dgrad, const TV& tin, const TV& tout, const TV& tmask, const void* wpk, const float* bias, int n_out,
                           int relu, uint64_t mask_bits, uint64_t accum_bits, int org, hipStream_t st, SignMap sgn, const DupOut* dup) {
    DupOut dupv;
    if (dup != nullptr) dupv = *dup;
    else { dupv.out = tout; dupv.mask = tout; dupv.frag = -1; }
    const int tiles_x = cdiv(tout.ws - 2 * org, MT), tiles_y = cdiv(tout.hs - 2 * org, DT_ROWS);
    const int nmb = n_mblocks(n_out);
    const int m16p = nmb * 4 * 16;
    const long long nitems = (long long)tiles_x * tiles_y * tout.n * nmb;
    g_num_cus = cached_num_cus();
    int G = g_num_cus / 8 * 8;   // one persistent block per CU (150 KB of LDS each)
    if (G < 8) G = 8;
    if (nitems < G) G = (int)nitems;
    // dgrad with ReLU masks and >= 2 chunks per tile: the loader waves stage the mask bits (-15 % on the dgrads against consumers fetching them)
    const bool lmask = dgrad && mask_bits != 0 && cdiv(tin.cb, CHUNK_CB) >= 2;
#define DMA_GO(D_, L_, ORG_)                                                                                                                  \
    hipLaunchKernelGGL((conv_dma_kernel<D_, L_>), dim3(G), dim3((D_CONS + D_LOAD) * 64), 0, st, tin, tout, tmask, (const uint4*)wpk, bias, n_out, \
                       m16p, relu, (unsigned long long)mask_bits, (unsigned long long)accum_bits, tiles_x, tiles_y, nmb, g_trace, g_abl, ORG_, sgn, dupv)
    if (dgrad && dupv.frag >= 0) {
        hipLaunchKernelGGL((conv_dma_kernel<true, 0, true>), dim3(G), dim3((D_CONS + D_LOAD) * 64), 0, st, tin, tout, tmask, (const uint4*)wpk, bias, n_out,
                           m16p, relu, 0ull, 0ull, tiles_x, tiles_y, nmb, g_trace, g_abl, org, sgn, dupv);
    } else if (dgrad && lmask && sgn.p != nullptr) DMA_GO(true, 2, org);
    else if (dgrad && lmask) DMA_GO(true, 1, org);
    else if (dgrad) DMA_GO(true, 0, org);
    else DMA_GO(false, 0, 0);
#undef DMA_GO
    return check_launch(dgrad ? "conv_dma dgrad" : "conv_dma fwd");
}

// fold (dgrad only): the caller wants fold_halo(gx) applied as well and guarantees that gx's halo ring is zero on entry; *folded
// reports whether the kernel chosen did it (interior tiles + fold steps, ring left zero) -- otherwise the caller runs the fold kernel.
// the 64 -> 32 forward's geometry (thin_conv_async_kernel: two consumer groups, three tight slots of 324-granule planes)
constexpr int TNW_GROUPS = 2, TNW_PL = 324, TNW_MAXCB = 8, TNW_MAXP = (TNW_MAXCB * TNW_PL + 63) / 64 / TN_LOAD + 1;   // 41 pieces -> 11 per loader
constexpr int TNW_RING = 3 * ((TNW_MAXCB * TNW_PL + 63) / 64) * 1024;                                                  // 125 952 B
static int g_thin_wide = -1;   // mmif_debug_set_thin_wide(0): decode.2's forward stays on the register-staged kernel (the tests' cross-check; bit-identical results)
static bool thin_wide_ok(bool dgrad, int ks, int mf, const TV& tin, const TV& tout) {
    if (g_thin_wide < 0) g_thin_wide = 1;
    if (!(g_thin_wide == 1 && g_dma_mode == 1 && !dgrad && ks == 3 && mf == 2 && tin.cb > TN_MAXCB && tin.cb <= TNW_MAXCB && tin.plane * 16 * TNW_MAXCB < (1ll << 31)))
        return false;
    const long long ntiles = (long long)cdiv(tout.ws, MT) * cdiv(tout.hs, MT) * tout.n;
    int G = num_cus_() / 8 * 8;
    if (G < 8) G = 8;
    return TNW_RING / (cdiv(tin.cb * TNW_PL, 64) * 1024) >= TNW_GROUPS + 1 && ntiles >= 2ll * G && ntiles < (1ll << 31);
}
static bool thin_async_ok(bool dgrad, int ks, int mf, const TV& tin, const TV& tout, int org) {
    if (!(g_dma_mode == 1 && ks == 3 && mf <= 3 && (dgrad || mf >= 2) && tin.cb <= TN_MAXCB && (!dgrad || (tin.halo == 1 && tin.folded)) &&
          tin.plane * 16 * TN_MAXCB < (1ll << 31)))
        return false;
    const int tiles_x = cdiv(tout.ws - 2 * org, MT), tiles_y = cdiv(tout.hs - 2 * org, MT);
    const long long ntiles = (long long)tiles_x * tiles_y * tout.n;
    const int P = cdiv(cdiv(tin.cb * TN_PL, 64), TN_LOAD), slot_bytes = P * TN_LOAD * 1024;
    int G = num_cus_() / 8 * 8;
    if (G < 8) G = 8;
    return tn_ring_bytes(mf) / slot_bytes >= TN_GROUPS + 1 && ntiles >= 2ll * G && ntiles < (1ll << 31);
}
bool conv_dgrad_onto_supported(int ks, int cin, int cout, const TV& tin, const TV& tout) {
    (void)cout;
    init_modes();
    const int org = (ks == 3 && g_fuse_fold == 1 && tout.halo == 1 && tout.h >= 4 && tout.w >= 4) ? 1 : 0;
    return org == 1 && thin_async_ok(true, ks, pick_mf(cin), tin, tout, org);
}

// told (dgrad): accumulate onto THAT tensor's values instead of tout's own; only the thin asynchronous kernel implements it -- the
// call fails (MMIF_EINVAL) when the layer / shape would take another kernel (conv_dgrad_onto_supported tells beforehand)
bool conv_dgrad_onto_supported(int ks, int cin, int cout, const TV& tin, const TV& tout);
bool conv1x1_stream_ok(bool dgrad, const TV& tin, const TV& tout, const TV& tmask, int n_out, int m16p, uint64_t mask_bits, uint64_t accum_bits);   // conv1x1.hip
int conv1x1_stream(bool dgrad, const TV& tin, const TV& tout, const TV& tmask, const void* wpk, const float* bias, int n_out, int m16p, int relu,
                   uint64_t mask_bits, hipStream_t st);
int conv_mfma(bool dgrad, int ks, const TV& tin, const TV& tout, const TV& tmask, const void* w_packed, const float* bias,
              int cin, int cout, int relu, uint64_t mask_bits, uint64_t accum_bits, hipStream_t st, bool fold, bool* folded, const TV* told) {
    if (folded != nullptr) *folded = false;
    const int n_out = dgrad ? cin : cout;
    const int mf = pick_mf(n_out);
    init_modes();
    const int org = (dgrad && fold && folded != nullptr && ks == 3 && g_fuse_fold == 1 && tout.halo == 1 && tout.h >= 4 && tout.w >= 4) ? 1 : 0;
    // 1x1 layers: the streaming kernel (csrc/conv1x1.hip: weights resident in LDS, B fragments straight from global memory)
    if (ks == 1 && told == nullptr) {
        const int m16p = n_mblocks(n_out) * mf * 16;
        if (conv1x1_stream_ok(dgrad, tin, tout, tmask, n_out, m16p, mask_bits, accum_bits))
            return conv1x1_stream(dgrad, tin, tout, tmask, w_packed, bias, n_out, m16p, relu, mask_bits, st);
    }
    // the DMA-staged kernel: 3x3, 64-row M-blocks, input gradient already folded, tensors within 32-bit plane offsets
    if (g_dma_mode == 1 && ks == 3 && mf == 4 && (!dgrad || (tin.halo == 1 && tin.folded)) &&
        tin.plane * 16 * CHUNK_CB < (1ll << 31))
    {
        if (told != nullptr) {
            set_error("conv dgrad: accumulate-onto-another-tensor is only implemented by the thin asynchronous kernel");
            return MMIF_EINVAL;
        }
        if (org) *folded = true;
        return launch_conv_dma(dgrad, tin, tout, tmask, w_packed, bias, n_out, relu, mask_bits, accum_bits, org, st);
    }
    // thin layers: asynchronous loader / consumer kernel with resident weights (one M-block of <= 48 channels, <= 48 input
    // channels, a ring of at least TN_GROUPS + 1 tile slots, at least two tiles per persistent block).  Measured (B=32 256x256,
    // vs conv_mfma_kernel<3,MF>): every dgrad -13 .. -21 %, forward with 32 / 48 outputs -14 % / -30 %; forward with 16 outputs
    // is +3 .. +16 % (the register-staged kernel runs 4 blocks per SIMD there), so that case stays on the old kernel.
    if (thin_wide_ok(dgrad, ks, mf, tin, tout)) {
        const int tiles_x = cdiv(tout.ws, MT), tiles_y = cdiv(tout.hs, MT);
        const long long ntiles = (long long)tiles_x * tiles_y * tout.n;
        int G = num_cus_() / 8 * 8;
        if (G < 8) G = 8;
        hipLaunchKernelGGL((thin_conv_async_kernel<2, false, TNW_GROUPS, TNW_PL, TNW_MAXP, TNW_RING, true>), dim3(G), dim3((4 * TNW_GROUPS + TN_LOAD) * 64), 0,
                           st, tin, tout, tmask, (const uint4*)w_packed, bias, n_out, relu, (unsigned long long)mask_bits,
                           (unsigned long long)accum_bits, tiles_x, tiles_y, (int)ntiles, 0, tout, 0);
        return check_launch("thin_conv_async fwd (wide)");
    }
    if (thin_async_ok(dgrad, ks, mf, tin, tout, org)) {
        const int tiles_x = cdiv(tout.ws - 2 * org, MT), tiles_y = cdiv(tout.hs - 2 * org, MT);
        const long long ntiles = (long long)tiles_x * tiles_y * tout.n;
        int G = num_cus_() / 8 * 8;
        if (G < 8) G = 8;
        {
            const TV told_v = told != nullptr ? *told : tout;
            const int use_old = told != nullptr ? 1 : 0;
#define TGO(MF_)                                                                                                                       \
    do {                                                                                                                               \
        if (dgrad)                                                                                                                     \
            hipLaunchKernelGGL((thin_conv_async_kernel<MF_, true>), dim3(G), dim3((4 * TN_GROUPS + TN_LOAD) * 64), 0, st, tin, tout,   \
                               tmask, (const uint4*)w_packed, bias, n_out, relu, (unsigned long long)mask_bits,                       \
                               (unsigned long long)accum_bits, tiles_x, tiles_y, (int)ntiles, org, told_v, use_old);                  \
        else                                                                                                                           \
            hipLaunchKernelGGL((thin_conv_async_kernel<MF_, false>), dim3(G), dim3((4 * TN_GROUPS + TN_LOAD) * 64), 0, st, tin, tout,  \
                               tmask, (const uint4*)w_packed, bias, n_out, relu, (unsigned long long)mask_bits,                       \
                               (unsigned long long)accum_bits, tiles_x, tiles_y, (int)ntiles, 0, told_v, 0);                          \
        if (org) *folded = true;                                                                                                       \
        return check_launch(dgrad ? "thin_conv_async dgrad" : "thin_conv_async fwd");                                                 \
    } while (0)
            switch (mf) { case 1: TGO(1); case 2: TGO(2); default: TGO(3); }
#undef TGO
        }
    }
    if (told != nullptr) {
        set_error("conv dgrad: accumulate-onto-another-tensor is only implemented by the thin asynchronous kernel (this layer / shape takes another)");
        return MMIF_EINVAL;
    }
#define GO(KS_, MF_) return launch_conv_mfma<KS_, MF_>(dgrad, tin, tout, tmask, w_packed, bias, n_out, relu, mask_bits, accum_bits, st)
    if (ks == 3) {
        switch (mf) { case 1: GO(3, 1); case 2: GO(3, 2); case 3: GO(3, 3); default: GO(3, 4); }
    } else {
        switch (mf) { case 1: GO(1, 1); case 2: GO(1, 2); case 3: GO(1, 3); default: GO(1, 4); }
    }
#undef GO
}

// dgrad (folded, nothing masked or accumulated on its own output) + the two masked copies of fragment `frag` (struct DupOut): the DMA-staged
// kernel only -- wide layers (>= 49 input channels), 3x3, the in-tile reflect fold, gy a folded halo-1 gradient
bool conv_dgrad_dup_supported(int ks, int cin, const TV& tin, const TV& tout) {
    init_modes();
    return g_dma_mode == 1 && g_fuse_fold == 1 && ks == 3 && pick_mf(cin) == 4 && tin.halo == 1 && tin.folded && tout.halo == 1 && tout.h >= 4 &&
           tout.w >= 4 && tin.plane * 16 * CHUNK_CB < (1ll << 31);
}
int conv_dgrad_dup(const TV& tin, const TV& tout, const void* wpk, int cin, int cout, const TV& tdup, const TV& tmask, int frag, hipStream_t st) {
    (void)cout;
    DupOut d;
    d.out = tdup; d.mask = tmask; d.frag = frag;
    return launch_conv_dma(true, tin, tout, tout, wpk, nullptr, cin, 0, 0, 0, 1, st, SignMap{nullptr, 0, 0, 0}, &d);
}

static inline int pick_mfw(int cout) { return cout <= 16 ? 1 : (cout <= 32 ? 2 : 4); }

// tile groups per (icg, ocg) pair of the register-staged wgrad.  The grid is G * nb persistent blocks; it must fit the resident
// capacity in ONE round (3 blocks/CU at MFW = 1 -- 136 VGPRs --, else 2; 256 CUs): 1032 blocks on 768 slots ran a second round
// at 34 % occupancy.
// input-channel fragments per block of the register-staged kernel: 1x1 layers with 64-row output groups take up to 4 (see the kernel)
static inline int pick_icf(int ks, int cin, int cout) { return (ks == 1 && pick_mfw(cout) == 4) ? (cin > 32 ? 4 : (cin > 16 ? 2 : 1)) : 1; }

static int wgrad_G(int cin, int cout, int icf = 1) {
    const int mfw = pick_mfw(cout);
    const int nb = cdiv(cin, 16 * icf) * cdiv(cout, mfw * 16);
    const int capacity = 256 * (mfw == 1 ? 3 : 2);
    int G = capacity / nb / 8 * 8;
    if (G > 512) G = 512;   // (768 blocks for a single pair measured 8 % slower than 512)
    return G < 8 ? 8 : G;
}

bool wgrad_mfma_supported(int ks, int cin, int cout) { return (ks == 1 || ks == 3) && cin >= 1 && cout >= 1; }

static bool wgrad_dma_shape(int ks, int cin, int cout) {
    const long long padded = (long long)cdiv(cin, 64) * 64 * cdiv(cout, 64) * 64;
    return ks == 3 && cin % 8 == 0 && cout % 8 == 0 && (long long)cin * cout * 10 >= padded * 6;
}
static int g_wgrad_dma_blocks = 256;   // mmif_debug_set_wgrad_dma_blocks: persistent blocks of wgrad_dma_kernel (one per CU); fewer leave CUs to
                                       // a kernel running concurrently on another stream (the intra-step overlap experiment)
static int wgrad_dma_G(int cin, int cout, int blocks = 256) {   // tile groups per (icg, ocg) pair: about one persistent block per CU in total
    const int npairs = cdiv(cin, 64) * cdiv(cout, 64);
    int G = blocks / npairs;
    if (G >= 8) G = G / 8 * 8;   // multiples of 8 keep the blocks that share tiles on one XCD
    return G < 1 ? 1 : G;
}

// enc_wgrad.hip: single-layer "tap-row" kernel (whole x / g tile staged once per block)
bool wgrad_taprow_supported(int ks, int cin, int cout);
size_t wgrad_taprow_workspace(int cin, int cout);
int wgrad_taprow(const TV& tx, const TV& tg, float* dw, float* db, int cin, int cout, int accumulate, float* ws, hipStream_t st);
static int g_taprow_mode = -1;   // $MMIF_WGRAD_TAPROW=0: keep the per-input-group kernel (A/B timing)

size_t wgrad_mfma_workspace(int cin, int cout, int ks) {
    const int mfw = pick_mfw(cout), icf = pick_icf(ks, cin, cout);
    size_t a = 0;
    for (int f = 1; f <= icf; f *= 2) {   // (either block width may run: $MMIF_WGRAD1X1_WIDE)
        const size_t per = (size_t)mfw * 16 * 16 * f * ks * ks + mfw * 16;
        const size_t b = (size_t)wgrad_G(cin, cout, f) * cdiv(cin, 16 * f) * cdiv(cout, mfw * 16) * per * sizeof(float);
        if (b > a) a = b;
    }
    if (wgrad_dma_shape(ks, cin, cout)) {
        const size_t b = (size_t)wgrad_dma_G(cin, cout) * cdiv(cin, 64) * cdiv(cout, 64) * WD_PER * sizeof(float);
        if (b > a) a = b;
    }
    if (wgrad_taprow_supported(ks, cin, cout)) {
        const size_t b = wgrad_taprow_workspace(cin, cout);
        if (b > a) a = b;
    }
    return a;
}

static int launch_wgrad_dma(const TV& tx, const TV& tg, float* dw, float* db, int cin, int cout, int accumulate, float* ws, hipStream_t st,
                            SignMap sgn = SignMap{nullptr, 0, 0, 0}) {
    const int tiles_x = cdiv(tx.w, MT), tiles_y = cdiv(tx.h, MT);
    const int tpi = tiles_x * tiles_y, total = tpi * tx.n;
    const int n_icg = cdiv(cin, 64), n_ocg = cdiv(cout, 64);
    int G = wgrad_dma_G(cin, cout, g_wgrad_dma_blocks);   // (the workspace is sized for the full grid)
    if (G > total) G = total;   // every tile group owns at least one tile (the reduce sums all G partials)
    ws = defer_ws(ws, (size_t)G * n_icg * n_ocg * WD_PER * sizeof(float));      // (csrc/reduce_defer.hpp: an arena slot while reductions are deferred)
    hipLaunchKernelGGL(wgrad_dma_kernel, dim3(G * n_icg * n_ocg), dim3((D_CONS + D_LOAD) * 64), 0, st, tx, tg, ws, tiles_x, tpi, total, G,
                       n_icg, n_ocg, sgn, g_wgrad_ragged);
    if (int rc = check_launch("wgrad_dma")) return rc;
    const int n = cout * cin * 9 + cout;
    const int RG = G;
    {
        RedJob J;
        J.partial = ws; J.dw = dw; J.db = db; J.type = RED_WGRAD_DMA; J.sl = RG > 64 ? 16 : 4; J.G = G; J.accumulate = accumulate;
        J.p0 = cin; J.p1 = cout; J.p2 = n_icg; J.p3 = n_ocg; J.nvb = cdiv(n, 64);
        if (defer_push(J)) return MMIF_OK;
    }
    if (RG > 64) hipLaunchKernelGGL(wgrad_dma_reduce<16>, dim3(cdiv(n, 64)), dim3(1024), 0, st, ws, dw, db, cin, cout, G, n_icg, n_ocg, accumulate);
    else hipLaunchKernelGGL(wgrad_dma_reduce<4>, dim3(cdiv(n, 64)), dim3(256), 0, st, ws, dw, db, cin, cout, G, n_icg, n_ocg, accumulate);
    return check_launch("wgrad_dma_reduce");
}

template <int KS, int MFW, int KSPLIT, int ICF = 1>
static int launch_wgrad_mfma(const TV& tx, const TV& tg, float* dw, float* db, int cin, int cout, int accumulate, float* ws,
                             hipStream_t st) {
    const int tiles_x = cdiv(tx.w, MT), tiles_y = cdiv(tx.h, MT);
    const int tpi = tiles_x * tiles_y, total = tpi * tx.n;
    int G = wgrad_G(cin, cout, ICF);
    if (G > total) G = total;  // (no longer a multiple of 8: the kernel falls back to the plain block order)
    const int n_icg = cdiv(cin, 16 * ICF), n_ocg = cdiv(cout, MFW * 16);
    hipLaunchKernelGGL((wgrad_mfma_kernel<KS, MFW, KSPLIT, ICF>), dim3(G * n_icg * n_ocg), dim3(256), 0, st, tx, tg, ws, tiles_x, tpi,
                       total, G, n_icg, n_ocg);
    if (int rc = check_launch("wgrad_mfma")) return rc;
    const int n = cout * cin * KS * KS + cout;
    hipLaunchKernelGGL((wgrad_mfma_reduce<KS, MFW, ICF>), dim3(cdiv(n, 64)), dim3(256), 0, st, ws, dw, db, cin, cout, G, n_icg, n_ocg,
                       accumulate);
    return check_launch("wgrad_mfma_reduce");
}

int wgrad_mfma(int ks, const TV& tx, const TV& tg, float* dw, float* db, int cin, int cout, int accumulate, float* ws,
               hipStream_t st) {
    const int mfw = pick_mfw(cout);
    init_modes();
    if (g_dma_mode == 1 && wgrad_dma_shape(ks, cin, cout) && tg.halo == 1 && tg.folded && tx.plane * 16 * 8 < (1ll << 31) &&
        tg.plane * 16 * 8 < (1ll << 31))
        return launch_wgrad_dma(tx, tg, dw, db, cin, cout, accumulate, ws, st);
    if (g_taprow_mode < 0) {
        const char* e = getenv("MMIF_WGRAD_TAPROW");
        g_taprow_mode = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    if (g_taprow_mode == 1 && wgrad_taprow_supported(ks, cin, cout) && (tg.halo == 0 || tg.folded) && tx.halo == 0)
        return wgrad_taprow(tx, tg, dw, db, cin, cout, accumulate, ws, st);
#define GO(KS_, M_, K_) return launch_wgrad_mfma<KS_, M_, K_>(tx, tg, dw, db, cin, cout, accumulate, ws, st)
    if (ks == 3) {
        switch (mfw) { case 1: GO(3, 1, 4); case 2: GO(3, 2, 2); default: GO(3, 4, 2); }
    } else {
        switch (mfw) {
            case 1: GO(1, 1, 4);
            case 2: GO(1, 2, 2);
            default:
                switch (pick_icf(1, cin, cout)) {
                    case 4: return launch_wgrad_mfma<1, 4, 2, 4>(tx, tg, dw, db, cin, cout, accumulate, ws, st);
                    case 2: return launch_wgrad_mfma<1, 4, 2, 2>(tx, tg, dw, db, cin, cout, accumulate, ws, st);
                    default: GO(1, 4, 2);
                }
        }
    }
#undef GO
}

}  // namespace mmif

using namespace mmif;

extern "C" void mmif_debug_set_trace(void* device_buf) { mmif::g_trace = (long long*)device_buf; }
// 1 (default) = use the DMA-staged kernels where they apply, 0 = register-staged kernels only
extern "C" void mmif_debug_set_conv_dma(int32_t mode) { mmif::g_dma_mode = mode ? 1 : 0; }
extern "C" void mmif_debug_set_bwd_pair_dma(int32_t mode) { mmif::debug_set_bwd_pair_dma(mode); }
extern "C" void mmif_debug_set_thin_wide(int32_t mode) { mmif::g_thin_wide = mode ? 1 : 0; }
extern "C" void mmif_debug_set_ragged(int32_t mode) { mmif::g_wgrad_ragged = mode ? 1 : 0; }
extern "C" void mmif_debug_set_wgrad_dma_blocks(int32_t blocks) { mmif::g_wgrad_dma_blocks = blocks < 8 ? 8 : (blocks > 256 ? 256 : blocks); }

extern "C" size_t mmif_packed_weight_bytes(int32_t cout, int32_t cin, int32_t ksize) {
    const size_t a = packed_bytes(cout, cin, ksize), b = packed_bytes(cin, cout, ksize);
    return a > b ? a : b;
}

namespace mmif { int conv_x3_pack_multi(const mmif_pack_job* jobs, int n_jobs, hipStream_t st); }   // conv_x3.hip: packs the MMIF_PACK_X3 jobs, skips the others

extern "C" int mmif_pack_weights_multi(const mmif_pack_job* jobs, int32_t n_jobs, void* stream) {
    MMIF_REQUIRE(jobs != nullptr && n_jobs > 0, "pack_weights_multi: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    PackTable tab;
    int n = 0;
    auto flush = [&]() -> int {
        if (n == 0) return MMIF_OK;
        hipLaunchKernelGGL(pack_weights_multi_kernel, dim3(64, n), dim3(256), 0, st, tab);
        n = 0;
        return check_launch("pack_weights_multi");
    };
    bool any_x3 = false;
    for (int i = 0; i < n_jobs; ++i) {
        const mmif_pack_job& jb = jobs[i];
        MMIF_REQUIRE(jb.ksize == 1 || jb.ksize == 3, "pack_weights_multi: ksize must be 1 or 3 (job %d)", i);
        MMIF_REQUIRE(jb.w != nullptr && jb.cout > 0 && jb.cin > 0, "pack_weights_multi: bad arguments (job %d)", i);
        MMIF_REQUIRE(jb.format == MMIF_PACK_BF16 || jb.format == MMIF_PACK_X3, "pack_weights_multi: bad format (job %d)", i);
        if (jb.format == MMIF_PACK_X3) { any_x3 = true; continue; }   // split-bf16 images of fp32 tensors: conv_x3.hip, below
        for (int d = 0; d < 2; ++d) {
            void* dst = d ? jb.packed_dgrad : jb.packed_fwd;
            if (dst == nullptr) continue;
            const int n_out = d ? jb.cin : jb.cout, n_in = d ? jb.cout : jb.cin;
            PackImage& im = tab.im[n++];
            im.w = jb.w; im.dst = (bf16_t*)dst; im.total = (long long)(packed_bytes(n_out, n_in, jb.ksize) / 2);
            im.cout = jb.cout; im.cin = jb.cin; im.ks = jb.ksize; im.dgrad = d;
            im.m16p = n_mblocks(n_out) * pick_mf(n_out) * 16;
            if (n == PACK_MAX_IMAGES)
                if (int rc = flush()) return rc;
        }
    }
    if (int rc = flush()) return rc;
    return any_x3 ? conv_x3_pack_multi(jobs, n_jobs, st) : MMIF_OK;
}

// ---- DenseBlock(16, 16) backward chain in GATHER form.  Layer by layer, the dgrad of DenseBlock conv L scatters into all of its
// inputs: x0's gradient is read-modify-written three times (by convs 3, 2, 1), x1's twice.  Per DESTINATION the same sums are
//     g(x_k) = mask(G_k + sum_{L > k} dgrad_L(g_L)[channels of x_k])          k = 2, 1, 0
// i.e. the dgrad of a VIRTUAL layer with 16 input channels (x_k) and 16 (3 - k) output channels -- convs k+1 .. 3 stacked, each
// restricted to its x_k input slice -- whose output gradient is the contiguous block range [g_{k+1} | .. | g_3].  One dgrad launch
// per destination (accumulate onto G_k + ReLU mask in the epilogue, fp32 sum of all contributions, ONE bf16 rounding) instead of
// read-modify-write passes over the lower blocks.  This kernel writes the three virtual layers' dgrad operand images.
struct ChainSrc { const float* w[3]; bf16_t* dst[3]; long long total[3]; };
struct ChainSrc2 { ChainSrc s[2]; };          // two encoder branches in one launch (blockIdx.y = 3 * branch + k)
__global__ void pack_dense_chain_kernel(ChainSrc2 S2) {
    const ChainSrc& S = S2.s[blockIdx.y / 3];
    const int k = blockIdx.y % 3;                   // destination x_k; virtual cout = 16 (3 - k), cin = 16
    const int n_in = 16 * (3 - k), ncb = 2 * (3 - k);
    const long long total = S.total[k];
    (void)n_in;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int e = idx & 7;
        const long long row = idx >> 3;
        const int oc = (int)(row % 16);             // channel of x_k
        long long kgp = row / 16;
        int c0 = 0, n = 0, kg = 0;
        for (c0 = 0; c0 < ncb; c0 += CHUNK_CB) {
            n = ncb - c0 < CHUNK_CB ? ncb - c0 : CHUNK_CB;
            const int pad = ((9 * n + 3) / 4) * 4;
            if (kgp < pad) { kg = (int)kgp; break; }
            kgp -= pad;
        }
        float val = 0.f;
        if (kg < 9 * n) {
            const int tap = kg / n, cb = kg % n;
            const int u = tap / 3, v = tap % 3;
            const int ic = (c0 + cb) * 8 + e;       // virtual output channel: conv L = k + 1 + ic / 16, its output ic % 16
            const int L = k + 1 + ic / 16, o = ic % 16;
            val = S.w[L - 1][(((long long)o * (16 * L) + 16 * k + oc) * 3 + (2 - u)) * 3 + (2 - v)];
        }
        S.dst[k][idx] = f32_to_bf16(val);
    }
}

extern "C" int mmif_pack_dense_chain(const float* w1, const float* w2, const float* w3, void* packed_v0, void* packed_v1, void* packed_v2,
                                     void* stream) {
    MMIF_REQUIRE(w1 != nullptr && w2 != nullptr && w3 != nullptr && packed_v0 != nullptr && packed_v1 != nullptr && packed_v2 != nullptr,
                 "pack_dense_chain: NULL argument");
    ChainSrc S;
    S.w[0] = w1; S.w[1] = w2; S.w[2] = w3;
    S.dst[0] = (bf16_t*)packed_v0; S.dst[1] = (bf16_t*)packed_v1; S.dst[2] = (bf16_t*)packed_v2;
    for (int k = 0; k < 3; ++k) S.total[k] = (long long)(packed_bytes(16, 16 * (3 - k), 3) / 2);
    ChainSrc2 S2;
    S2.s[0] = S; S2.s[1] = S;
    hipLaunchKernelGGL(pack_dense_chain_kernel, dim3(8, 3), dim3(256), 0, (hipStream_t)stream, S2);
    return check_launch("pack_dense_chain");
}

// the same for the two encoder branches of a PFNet-style model in ONE launch (w_a / packed_a: {w1, w2, w3} / {v0, v1, v2} of branch a)
extern "C" int mmif_pack_dense_chain_pair(const float* const* w_a, void* const* packed_a, const float* const* w_b, void* const* packed_b, void* stream) {
    MMIF_REQUIRE(w_a != nullptr && packed_a != nullptr && w_b != nullptr && packed_b != nullptr, "pack_dense_chain_pair: NULL argument");
    ChainSrc2 S2;
    for (int b = 0; b < 2; ++b)
        for (int k = 0; k < 3; ++k) {
            const float* w = (b ? w_b : w_a)[k];
            void* d = (b ? packed_b : packed_a)[k];
            MMIF_REQUIRE(w != nullptr && d != nullptr, "pack_dense_chain_pair: NULL weight / image");
            S2.s[b].w[k] = w;
            S2.s[b].dst[k] = (bf16_t*)d;
            S2.s[b].total[k] = (long long)(packed_bytes(16, 16 * (3 - k), 3) / 2);
        }
    hipLaunchKernelGGL(pack_dense_chain_kernel, dim3(8, 6), dim3(256), 0, (hipStream_t)stream, S2);
    return check_launch("pack_dense_chain_pair");
}

extern "C" int mmif_pack_weights(const float* w, int32_t cout, int32_t cin, int32_t ksize, void* packed_fwd, void* packed_dgrad,
                                 void* stream) {
    MMIF_REQUIRE(ksize == 1 || ksize == 3, "pack_weights: ksize must be 1 or 3");
    MMIF_REQUIRE(w != nullptr && cout > 0 && cin > 0, "pack_weights: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    for (int d = 0; d < 2; ++d) {
        void* dst = d ? packed_dgrad : packed_fwd;
        if (dst == nullptr) continue;
        const int n_out = d ? cin : cout, n_in = d ? cout : cin;
        const long long total = (long long)(packed_bytes(n_out, n_in, ksize) / 2);
        const int mf = pick_mf(n_out), m16p = n_mblocks(n_out) * mf * 16;
        int nb = (int)((total + 255) / 256);
        if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(pack_weights_kernel, dim3(nb), dim3(256), 0, st, w, cout, cin, ksize, d, mf, m16p, (bf16_t*)dst, total);
        if (int rc = check_launch("pack_weights")) return rc;
    }
    return MMIF_OK;
}
