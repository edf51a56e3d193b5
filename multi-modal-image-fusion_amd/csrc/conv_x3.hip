// fp32-grade 3x3 / 1x1 convolution on the gfx950 MATRIX pipe: split-operand ("x3") kernels for fp32 tensors.
//
// The reference computes its ConvLayers in fp32 (core/block.py:56-66, 98-99) and BASELINE.json's north star asks for results within
// 1e-3 of that.  gfx950 has no TF32 / xf32 MFMA and its fp32-input MFMA runs at the VALU rate (157 TFLOP/s), so the parity path used to
// be the fp32 FMA kernels of conv_valu.hip (0.42 k image pairs / s).  Here every fp32 operand is split ONCE, while it is staged into LDS,
// into 16-bit pieces
//        x = hi + lo + r,   hi = round16(x),   lo = round16(x - hi)
// and a product a * b is accumulated (fp32, inside the MFMA) as  a_lo b_hi + a_hi b_lo + a_hi b_hi: three v_mfma_f32_32x32x16 per K = 16
// step, i.e. 1/3 of the 2.5 PFLOP/s 16-bit peak = 830 TFLOP/s of "fp32-grade" work, 5.3x the fp32 pipe.
//   * FORWARD: fp16 pieces of the value scaled per staged tile (11 + 11 bits: 2^-23 per product -- fp32 grade, which the forward needs
//     because a ReLU decision on a pre-activation within the error of zero would differ from the reference's; x3_split_pair_h below);
//     bf16 pieces (8 + 8 bits, 2^-17) as `mmif_set_x3_forward_pieces(2)`, three bf16 pieces / six products as (3).
//   * BACKWARD (dgrad, wgrad): bf16 pieces, three products -- linear in the gradient, 1e-5 is plenty.
// HBM tensors stay plain fp32 blocked NHWC, so every other kernel of the fp32 path (image-side layers, losses, fold, fusion, optimiser)
// is untouched.
//
// conv_x3_kernel<MB, DGRAD, NP, NW, RJ, KS, F16>: forward and input gradient (one formulation, see conv_valu.hip).  Persistent blocks,
//   output tile 32 columns x (RJ x NW) rows x 32 MB channels, a wave owns RJ rows.  K runs in LDS chunks of 16 / 32 input channels x KS^2
//   taps; per chunk the input tile is loaded as fp32 into registers one chunk AHEAD, split and written to the (single) LDS buffer after
//   the chunk's MFMAs, between two barriers; the pre-split weight images come from mmif_pack_weights_x3.  One 32x32x16 MFMA = one tap x
//   16 channels: lanes 0-31 hold the channel block 2c of pixel / output channel (lane & 31), lanes 32-63 block 2c+1, so every operand
//   fetch is one ds_read_b128 of a contiguous 512-byte run per half wave (conflict free for any 16-byte aligned start:
//   MI355X_MICROARCH.md, LDS lane groups).  NW = 8, one block per CU: the 64-channel kernels.  NW = 4, three blocks per CU: layers
//   with <= 32 output and <= 64 input channels (latency bound: the blocks hide each other's loads).
// wgrad_x3_kernel: weight gradient, K = pixels.  Block of 12 waves owns a 64 x 64 (out, in) channel pair and walks 8 x 16 pixel tiles;
//   wave (v, jt, mt) keeps dW[32 oc][32 ic] of the three taps (0..2, v) in 48 accumulator registers; operands are fetched from the
//   pixel-major hi / lo tiles with the LDS transpose read (ds_read_b64_tr_b16), 16 pixels of one tile row per k-step.  It can leave the
//   ReLU sign map of x for the dgrad that follows (mmif_conv2d_reflect_bwd_wide).
// wgrad_x3_thin_kernel: the same for <= 48 input / <= 16 output channels: four-wave blocks, 3-5 per CU, 16x16x32 MFMAs.
#include "common.hpp"
#include <stdlib.h>
#include <mutex>
#include <unordered_map>

namespace mmif {

typedef __attribute__((ext_vector_type(8))) __bf16 x3_bf16x8;
typedef __attribute__((ext_vector_type(4))) short x3_s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float x3_f4;        // native vectors: HIP's float4 / uint4 are structs, and arrays of them that
typedef __attribute__((ext_vector_type(4))) unsigned x3_u4;     // travel through the staging lambdas were left in scratch / promoted to LDS
#define X3_LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

constexpr int X3_TW = 32;                        // output tile columns; rows = 2 per wave (16 with 8 waves, 8 with 4)

__host__ __device__ constexpr int x3_mb(int n_out) { return n_out > 32 ? 2 : 1; }
static inline int x3_nmb(int n_out) { return cdiv(n_out, 32 * x3_mb(n_out)); }
static inline int x3_nch(int n_in) { return cdiv(cdiv(n_in, 8), 2); }
// operand image: [m-block][chunk][hi | lo][tap u*3+v][channel block 0 / 1 of the chunk][32*MB out channels][8 in channels] bf16
static size_t x3_packed_bytes(int n_out, int n_in, int pieces, int ks = 3) {
    return (size_t)x3_nmb(n_out) * x3_nch(n_in) * pieces * ks * ks * 2 * 32 * x3_mb(n_out) * 16;
}

// scaled-fp16 pieces of the forward operands (see x3_split_pair_h below)
typedef _Float16 x3_h8 __attribute__((ext_vector_type(8)));
constexpr float X3_SW = 1024.f, X3_HMAX = 65000.f;   // weights: fixed 2^10 (|w| < 64; smaller than 1.2e-4 = subnormal lo piece)
constexpr int X3_SW_EXP = 10;
__host__ __device__ inline unsigned short x3_f16_bits(float v) { return __builtin_bit_cast(unsigned short, (_Float16)v); }
__host__ __device__ inline float x3_f16_val(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }

// ------------------------------------------------------------------ weight packing
constexpr int X3_PACK_MAX = 64;
struct X3PackImage { const float* w; bf16_t* dst; long long total; int cout, cin, dgrad, mbw, nch, pieces, ks, f16;
                     const float *w2, *w3; int chain_k; };   // chain_k >= 0: the DenseBlock's virtual gather layer k (mmif_pack_dense_chain_x3)
struct X3PackTable { X3PackImage im[X3_PACK_MAX]; };

// dgrad == 0: out = o, in = c, Wk[u][v] = W[o][c][u][v];  dgrad == 1: out = c, in = o, Wk[u][v] = W[o][c][2-u][2-v]
// image: [m-block][chunk][piece 0 .. pieces-1][tap][channel block of the chunk][32*MB out][8 in]
__device__ unsigned g_x3_sat_count = 0;   // weights clamped by the fp16 operand split since the last reset

__global__ void x3_pack_kernel(X3PackTable tab) {
    const X3PackImage& J = tab.im[blockIdx.y];
    const int n_out = J.dgrad ? J.cin : J.cout, n_in = J.dgrad ? J.cout : J.cin;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < J.total; idx += (long long)gridDim.x * blockDim.x) {
        // idx enumerates the bf16 elements of the FIRST piece's images; piece p sits p * 9*2*mbw*8 elements further
        const int e = (int)(idx & 7);
        long long r = idx >> 3;
        const int ocl = (int)(r % J.mbw); r /= J.mbw;
        const int cbl = (int)(r & 1); r >>= 1;
        const int ks = J.ks, taps = ks * ks;
        const int tap = (int)(r % taps); r /= taps;
        const int ch = (int)(r % J.nch);
        const int mb = (int)(r / J.nch);
        const int oc = mb * J.mbw + ocl, ic = (ch * 2 + cbl) * 8 + e;
        const int u = tap / ks, v = tap % ks;
        float val = 0.f;
        if (oc < n_out && ic < n_in) {
            if (J.chain_k >= 0) {
                // virtual layer k: "output" (gradient) channel ic = 16 (L - k - 1) + o of DenseBlock conv L = k+1 .. 3, input channel oc = x_k's
                // channel 16 k + oc of that conv; dgrad form (flipped taps)
                const int L = J.chain_k + 1 + ic / 16, o = ic % 16, cinL = 16 * L;
                const float* src = L == 1 ? J.w : (L == 2 ? J.w2 : J.w3);
                val = src[(((long long)o * cinL + 16 * J.chain_k + oc) * 3 + (2 - u)) * 3 + (2 - v)];
            } else if (J.dgrad) val = J.w[(((long long)ic * J.cin + oc) * ks + (ks - 1 - u)) * ks + (ks - 1 - v)];
            else val = J.w[(((long long)oc * J.cin + ic) * ks + u) * ks + v];
        }
        const long long piece = (long long)taps * 2 * J.mbw * 8;
        const long long within = ((long long)(tap * 2 + cbl) * J.mbw + ocl) * 8 + e;
        const long long base = ((long long)mb * J.nch + ch) * J.pieces * piece;
        if (J.f16) {   // two fp16 pieces of 2^10 w (saturating), see x3_split_pair_h
            // a weight with |w| >= ~63.5 does not fit the fixed 2^10 scale: it is clamped, and COUNTED (mmif_x3_pack_saturations) -- the
            // fp32 path is the parity-grade path, a silently saturated forward defeats its purpose
            if (fabsf(val * X3_SW) > X3_HMAX) atomicAdd(&g_x3_sat_count, 1u);
            val = fminf(fmaxf(val * X3_SW, -X3_HMAX), X3_HMAX);
            const unsigned short h = x3_f16_bits(val);
            J.dst[base + within] = h;
            J.dst[base + piece + within] = x3_f16_bits(val - x3_f16_val(h));
            continue;
        }
        for (int p = 0; p < J.pieces; ++p) {   // successive bf16 roundings of the remainder (each subtraction is exact in fp32)
            const bf16_t q = f32_to_bf16(val);
            J.dst[base + p * piece + within] = q;
            val -= bf16_to_f32(q);
        }
    }
}

// ------------------------------------------------------------------ staging helpers
struct X3Gran { x3_f4 a, b; };   // one fp32 granule (8 channels of one pixel)

// NP bf16 pieces of a granule: piece[0] = bf16(x), piece[1] = bf16(x - piece[0]), piece[2] = bf16(x - piece[0] - piece[1])
template <int NP>
__device__ inline void x3_split_pair(float v0, float v1, unsigned (&q)[NP]) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        q[p] = pack_bf16x2(v0, v1);
        if (p + 1 < NP) {
            v0 -= __uint_as_float(q[p] << 16);
            v1 -= __uint_as_float(q[p] & 0xffff0000u);
        }
    }
}
template <int NP>
__device__ inline void x3_split_gran(const X3Gran& g, x3_u4 (&out)[NP]) {
    unsigned q0[NP], q1[NP], q2[NP], q3[NP];
    x3_split_pair<NP>(g.a.x, g.a.y, q0);
    x3_split_pair<NP>(g.a.z, g.a.w, q1);
    x3_split_pair<NP>(g.b.x, g.b.y, q2);
    x3_split_pair<NP>(g.b.z, g.b.w, q3);
#pragma unroll
    for (int p = 0; p < NP; ++p) out[p] = (x3_u4){q0[p], q1[p], q2[p], q3[p]};
}

__device__ inline x3_bf16x8 x3_frag(const x3_u4& v) { return __builtin_bit_cast(x3_bf16x8, v); }

// ---- the FP16 form of the split (forward pass, "F16" kernels): two fp16 pieces of the value SCALED by a power of two,
//          s x = hi + lo + r,   hi = fp16(s x),   lo = fp16(s x - hi),   |r| <= 2^-24 |s x|        (11 + 11 significant bits)
// so the three products hi*hi + hi*lo + lo*hi already carry 2^-23 per product -- fp32 grade with THREE MFMAs instead of the six the bf16
// pieces (8 bits each) need.  What fp16 lacks is range, so the scale is ADAPTIVE: weights carry a fixed 2^10 (1.2e-4 <= |w| < 64 at
// full precision, saturating beyond), activations the largest power of two that keeps the block's staged tile below 2^15 -- the block
// takes the maximum |x| of every chunk it stages (registers -> wave reduce -> 8 LDS slots, read after the barrier the staging needs
// anyway) and the exponent of an item is the running minimum over its chunks; when a later chunk forces a smaller exponent the fp32
// accumulators are rescaled by that power of two (exact), and the epilogue scales back.  Every piece is therefore exact to 2^-24 of the
// LARGEST magnitude that enters the sum, whatever the tensor's scale (1e-30 or 1e30: test_fp16_forward_range_window).  The backward
// kernels keep the bf16 pieces (gradients sit around 1e-7 .. 1e-2 and 1e-5 is all they need).
__device__ inline void x3_split_pair_h(float v0, float v1, unsigned (&q)[2]) {   // v0, v1 already scaled (|v| < 2^15 by construction)
    const unsigned short h0 = x3_f16_bits(v0), h1 = x3_f16_bits(v1);
    q[0] = (unsigned)h0 | ((unsigned)h1 << 16);
    q[1] = (unsigned)x3_f16_bits(v0 - x3_f16_val(h0)) | ((unsigned)x3_f16_bits(v1 - x3_f16_val(h1)) << 16);
}
__device__ inline float x3_pow2(int e) { return __uint_as_float((unsigned)(127 + e) << 23); }   // 2^e, -126 <= e <= 127
__device__ inline void x3_split_gran_h(const X3Gran& g, float sx, x3_u4 (&out)[2]) {
    unsigned q0[2], q1[2], q2[2], q3[2];
    x3_split_pair_h(g.a.x * sx, g.a.y * sx, q0);
    x3_split_pair_h(g.a.z * sx, g.a.w * sx, q1);
    x3_split_pair_h(g.b.x * sx, g.b.y * sx, q2);
    x3_split_pair_h(g.b.z * sx, g.b.w * sx, q3);
#pragma unroll
    for (int p = 0; p < 2; ++p) out[p] = (x3_u4){q0[p], q1[p], q2[p], q3[p]};
}
template <bool F16>
__device__ inline f32x16 x3_mfma(const x3_bf16x8& a, const x3_bf16x8& b, const f32x16& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(x3_h8, a), __builtin_bit_cast(x3_h8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// products (A piece, B piece) accumulated per tap, smallest first.  NP = 2: lo*hi + hi*lo + hi*hi (error ~2^-17 per product);
// NP = 3: mid*mid + lo*hi + hi*lo + mid*hi + hi*mid + hi*hi (what is dropped is below 2^-25: fp32 grade)
template <int NP> struct X3Prod;
template <> struct X3Prod<2> { static constexpr int N = 3; static constexpr int A[3] = {1, 0, 0}; static constexpr int B[3] = {0, 1, 0}; };
template <> struct X3Prod<3> { static constexpr int N = 6; static constexpr int A[6] = {1, 2, 0, 1, 0, 0}; static constexpr int B[6] = {1, 0, 2, 0, 1, 0}; };

constexpr int X3_BIAS_G = 128;   // granules reserved behind the tiles for the bias vector (512 floats)

// ------------------------------------------------------------------ forward / dgrad
// NP = 2: operands as (hi, lo), 3 MFMAs per tap and accumulator tile; the BACKWARD kernels (linear in the gradient: 1e-5 is plenty).
// NP = 3: operands as (hi, mid, lo), 6 MFMAs -- the FORWARD pass's default: a ReLU decision on a pre-activation within the 2-piece
//         error (~1e-5 of its scale) of zero would differ from the reference's about ten times as often as between two fp32
//         implementations, and every such flip moves the parameter gradients by O(1e-3) whatever the image size.
// One LDS tile (single buffered, two barriers per chunk) of CKB channel blocks: 4 (32 channels) with 2 pieces, 2 with 3 -- either way
// 150 / 114 KB and 216 MFMAs per wave between barriers; the next chunk's global loads fly during the MFMAs, the split + LDS write
// follows them.  (First version: 16-channel chunks, double buffered, one barrier per 108 MFMAs: the staging phases of the two waves of a
// SIMD coincide at the barrier and are fully exposed -- 41 % of the MFMA peak against 52 % for the 216-MFMA chunks.)
// (A PIPELINED form -- two 16-channel sub-chunk buffers, one barrier per sub-chunk, the two waves of a SIMD taking "stage s+1" and
// "MFMAs of s" in opposite order so that one of them always feeds the matrix pipe -- was built for the bf16-piece kernels: 2.02 vs 2.02 ms
// on decode.0's dgrad, and 5.96 vs 5.86 ms on 64 CUs, where the clock is not the limit; with the mask operands requested a whole
// sub-chunk ahead of the epilogue, 2.10 vs 1.96.  Its ablations: no MFMAs 1.14 ms, no loads / split / MFMAs 0.69 ms -- the epilogue's
// 2 x 1 GB of fp32 gradient writes and mask reads plus the barriers; staging and MFMA time ADD in either form.)
// NW = 8 waves: one block per CU (the instantiation in use).  NW = 4 -- TWO blocks of four waves per CU on 16-channel chunks and 8-row
// tiles, so that the two waves of a SIMD are not tied to one barrier -- was built and measured: 1.99 vs 2.01 ms on decode.0's dgrad;
// the waves overlap better but each block stages the whole weight chunk for half the pixels.  What the SQ counters say about these
// kernels (profiles/r03_pmc_sq_x3.txt): the matrix pipe is busy 58 % (2 pieces) / 74 % (3 pieces) of the cycles, i.e. a constant
// ~5 k cycles per 16 input channels of split + LDS-write + issue work that no barrier arrangement removes, at 1.7-1.8 GHz (the chip
// clocks dense MFMA work down: the 2.5 PFLOP/s peak assumes 2.4 GHz).
template <int MB, bool DGRAD, int NP, int NW, int RJ, int KS, bool F16 = false, bool M16 = false>
__global__ __launch_bounds__(64 * NW, 2) void conv_x3_kernel(TV tin, TV tout, TV tmask, const uint4* __restrict__ wpk,
                                                              const float* __restrict__ bias, int n_out, int nch16, int nmb, int relu,
                                                              unsigned long long mask_bits, unsigned long long accum_bits, int tiles_x,
                                                              int tiles_per_img, int total_tiles, const unsigned* __restrict__ signs, int org) {
    // org = 1 (round 6; 3x3 dgrads of the 32-wide-tile kernels): the tiles cover the INTERIOR of the halo-1 output only and the reflect-padding
    // adjoint is applied inside the tiles that own its targets, as conv_dma_kernel does (dgrad_fold_steps in conv_mfma.hip): in stored
    // coordinates the ring row 0 is  sum_v A(2, v) G[1][X - 1 + v]  = the fragments output row 2 reads for its taps (0, v) times the weights
    // of taps (2, v), and so on -- extra MFMAs for one row of one wave (rows) or with every lane but the target's zeroed (columns, corners).
    // The padded 258 x 258 domain of a 256 x 256 image cost 9 x 17 tiles of 32 x 16 for the interior's 8 x 16, and a fold launch per tensor.
    constexpr int MBW = 32 * MB;
    constexpr int TAPS = KS * KS, PD = KS / 2;    // KS = 1: the same kernel without halo and with one tap (HBM-bound: NestFuse's 1x1 layers)
    constexpr int WG = TAPS * 2 * MBW;            // weight granules of one piece of a 16-channel sub-chunk
    constexpr int X3_THREADS = 64 * NW, X3_TH = RJ * NW, X3_IH = X3_TH + 2 * PD, X3_IW = X3_TW + 2 * PD;   // a wave owns RJ rows of the tile
    constexpr int CKB = KS == 1 ? (NP == 2 ? 8 : 4) : ((NP == 2 && NW == 8 && RJ == 2) ? 4 : 2);   // channel blocks per LDS chunk
    constexpr int KK = CKB / 2;                   // 16-channel sub-chunks (MFMA k-steps per tap) per chunk
    // granules of one channel-block plane of the input tile: 612 (340 on 8-row tiles).  M16: a multiple of 16 -- its fragment read takes lanes
    // 0-15 from plane 0 and lanes 16-31 from plane 1, and a ds_read_b128 lane group ({0-3, 12-15, 20-27}, ..) is conflict free when the two
    // runs sit a whole number of 256-byte bank sweeps apart (with 340 it was 2-way on a quarter of the group: 35 % extra LDS cycles)
    constexpr int PL = M16 ? (X3_IH * X3_IW + 15) / 16 * 16 : X3_IH * X3_IW;
    constexpr int ING = CKB * PL;                 // input granules per piece
    constexpr int IN_ROUNDS = (ING + X3_THREADS - 1) / X3_THREADS;
    constexpr int WGC = KK * NP * WG;             // weight granules per chunk: [sub-chunk][piece][tap][cb 0/1][oc][8], as packed
    constexpr int W_ROUNDS = (WGC + X3_THREADS - 1) / X3_THREADS;
    constexpr int BUF_G = NP * ING + WGC;
    constexpr int NPROD = X3Prod<NP>::N;
    __shared__ __attribute__((aligned(16))) x3_u4 s_buf[BUF_G + X3_BIAS_G + 2];   // + 8 floats: the waves' tile maxima (F16)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const TileWalk tw = xcd_walk(total_tiles, gridDim.x, blockIdx.x);
    // timing ablations ($MMIF_ABLATE x3=, results are garbage), each after the first step: 1 no input loads (the first step's data is
    // kept: an all-zero tile clocks ~25 % higher and measures DVFS, not the loads), 2 no weight loads, 4 no split + LDS writes, 8 no
    // MFMAs.  decode.0, 128 -> 128, B = 32: forward (6 products) 2.75 ms, 1: 2.57, 4: 2.62, 5: 2.45, 8: 0.88; dgrad 1.95 / 1.78 / 1.84 /
    // 1.69 / 0.97; wgrad 1.62 / 1.29 / 1.54 / 1.18 / 0.82 -- the weight gradient waits for its tile loads (two half-tile round trips per
    // 7 us tile; all four granules in flight at once did not fit the 168-register budget: 24 bytes of scratch, 1.70 ms; warming L2 two tiles
    // ahead with one 4-byte load per granule made it slower, 1.95 ms: the memory system is busy, not cold)
    const int abl = relu >> 8;
    relu &= 255;
    const int nch = (nch16 + KK - 1) / KK;        // LDS chunks per item
    const int per_tile = nch * nmb;
    const int nsteps = tw.count * per_tile;
    if (nsteps == 0) return;

    if (!DGRAD) {   // bias (zero padded to the m-blocks) behind the tile
        float* s_bias = reinterpret_cast<float*>(s_buf + BUF_G);
        for (int i = tid; i < X3_BIAS_G * 4; i += X3_THREADS) s_bias[i] = (bias != nullptr && i < n_out) ? bias[i] : 0.f;
    }

    // the granules this thread stages, chunk independent: channel block << 16 | tile row << 8 | tile column
    unsigned geo[IN_ROUNDS];
#pragma unroll
    for (int k = 0; k < IN_ROUNDS; ++k) {
        const int e = min(tid + X3_THREADS * k, ING - 1);
        const int cb = e / PL, rem = e - cb * PL, py = rem / X3_IW;
        geo[k] = (unsigned)(cb << 16 | py << 8 | (rem - py * X3_IW));
    }

    // item of step s: tile ti = s / per_tile, m-block mb = (s / nch) % nmb, chunk c = s % nch
    bool done_first = false;
    X3Gran rin[IN_ROUNDS];
    x3_u4 rw[W_ROUNDS];

    auto issue = [&](int s) {
        const int c = s % nch, mb = (s / nch) % nmb, ti = s / per_tile;
        const int tile = tw.first + ti * tw.stride;
        const int in_ = tile / tiles_per_img, tt = tile - in_ * tiles_per_img;
        const int ys0 = (tt / tiles_x) * X3_TH, xs0 = (tt % tiles_x) * X3_TW;
        const int iy0 = ys0 + org - tout.halo - PD, ix0 = xs0 + org - tout.halo - PD;   // logical origin of the input tile
        const char* base = tin.base + ((long long)in_ * tin.img + (long long)(tin.cb_off + c * CKB) * tin.plane) * 32;
        const int ncb = tin.cb - c * CKB;                                   // channel blocks this chunk really has
#pragma unroll
        for (int k = 0; k < IN_ROUNDS; ++k) {
            const int cb = (int)(geo[k] >> 16), py = (int)((geo[k] >> 8) & 255u), px = (int)(geo[k] & 255u);
            if (!((abl & 1) && s > 0)) {   // (ablation 1 keeps the first step's REAL data: an all-zero tile would clock ~25 % higher)
                rin[k].a = (x3_f4){0.f, 0.f, 0.f, 0.f};
                rin[k].b = rin[k].a;
            }
            int y = iy0 + py, x = ix0 + px;
            bool ok = cb < ncb && tid + X3_THREADS * k < ING;
            if (DGRAD) {
                ok = ok && y >= 0 && y < tin.h && x >= 0 && x < tin.w;
                y += tin.halo; x += tin.halo;
            } else {
                y = min(max(reflect_idx(y, tin.h), 0), tin.h - 1);
                x = min(max(reflect_idx(x, tin.w), 0), tin.w - 1);
            }
            if (ok && !((abl & 1) && s > 0)) {
                const x3_f4* p = reinterpret_cast<const x3_f4*>(base + ((unsigned)cb * (unsigned)tin.plane + (unsigned)(y * tin.ws + x)) * 32u);   // (32-bit: x3_small)
                rin[k].a = p[0];
                rin[k].b = p[1];
            }
        }
        const x3_u4* src = reinterpret_cast<const x3_u4*>(wpk) + ((long long)mb * nch16 + c * KK) * (NP * WG);
        const int nw = min(KK, nch16 - c * KK) * (NP * WG);
#pragma unroll
        for (int k = 0; k < W_ROUNDS; ++k) {
            const int e = tid + X3_THREADS * k;
            if (e < nw && !((abl & 2) && s > 0)) rw[k] = src[e];
        }
    };
    // F16: the exponent the chunk being staged is scaled with (running minimum over the item's chunks), set by stage_scale()
    int e_stage = 0;
    auto tile_max = [&]() {      // every wave's maximum |x| of the staged registers -> LDS slot (read after the next barrier)
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < IN_ROUNDS; ++k)
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(rin[k].a.x), fabsf(rin[k].a.y)), fmaxf(fabsf(rin[k].a.z), fabsf(rin[k].a.w))),
                             fmaxf(fmaxf(fabsf(rin[k].b.x), fabsf(rin[k].b.y)), fmaxf(fabsf(rin[k].b.z), fabsf(rin[k].b.w)))));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) reinterpret_cast<float*>(s_buf + BUF_G + X3_BIAS_G)[wave] = m;
    };
    auto stage_scale = [&](bool first_chunk) {   // after the barrier: block maximum -> exponent with max * 2^e in [2^14, 2^15)
        const float* sm = reinterpret_cast<const float*>(s_buf + BUF_G + X3_BIAS_G);
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < NW; ++i) m = fmaxf(m, sm[i]);
        const int k = (int)((__float_as_uint(m) >> 23) & 255u) - 126;       // m = f * 2^k, f in [0.5, 1)
        int e = m > 0.f ? 15 - k : 100;
        e = min(max(e, -100), 100);
        e_stage = first_chunk ? e : min(e_stage, e);
    };
    auto commit = [&]() {
        if (abl & 4) { if (done_first) return; done_first = true; }
        const float sx = F16 ? x3_pow2(e_stage) : 1.f;
#pragma unroll
        for (int k = 0; k < IN_ROUNDS; ++k) {
            const int e = tid + X3_THREADS * k;
            if (e < ING) {
                x3_u4 pc[NP];
                if constexpr (F16) x3_split_gran_h(rin[k], sx, pc);
                else x3_split_gran<NP>(rin[k], pc);
#pragma unroll
                for (int p = 0; p < NP; ++p) s_buf[p * ING + e] = pc[p];
            }
        }
#pragma unroll
        for (int k = 0; k < W_ROUNDS; ++k) {
            const int e = tid + X3_THREADS * k;
            if (e < WGC) s_buf[NP * ING + e] = rw[k];   // (sub-chunks past the layer's last one: stale registers, never read)
        }
    };

    f32x16 acc[MB][RJ];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int j = 0; j < RJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;
    // M16 (layers with <= 16 output channels): 16 x 16 x 32 MFMAs -- M = the 16 output channels (a 32 x 32 tile would be half padding,
    // and these layers' MFMA time, not their bytes, was what they took: 48 -> 16 0.21 ms against 0.11 for the bytes), N = 16 pixels,
    // K = TWO TAPS x 16 channels: lane l = output channel / pixel l & 15, k-group l >> 4 = (tap of the pair) x (channel block of the
    // chunk).  Taps (0,1) (2,3) (4,5) (6,7) (8, none): five MFMA steps per 16 channels instead of nine of twice the length.
    x3_f4 acc16[RJ][2];
#pragma unroll
    for (int j = 0; j < RJ; ++j) { acc16[j][0] = (x3_f4){0.f, 0.f, 0.f, 0.f}; acc16[j][1] = acc16[j][0]; }
    static_assert(!M16 || (MB == 1 && KS == 3 && NP == 2 && CKB == 2), "M16: the 16-channel-chunk, two-piece, 3x3 kernels only");

    const int cbl = lane >> 5, nl = lane & 31;
    const int bbase = cbl * PL + RJ * wave * X3_IW + nl;         // + kk * 2 * PL + row * X3_IW + v   (+ piece * ING)
    const int abase = cbl * MBW + nl;                           // + kk * NP * WG + piece * WG + tap * 2 * MBW + m * 32

    issue(0);
    if constexpr (F16) {
        tile_max();
        __syncthreads();
        stage_scale(true);
    }
    commit();
    __syncthreads();
    int e_acc = e_stage, e_cur = e_stage;   // F16: exponent the accumulators hold / the chunk in LDS was staged with

    // the next chunk's address arithmetic + loads (~300 instructions) leave the matrix pipe idle for the wave that issues them, and the two
    // waves of a SIMD run in step between the barriers: waves 0 .. NW/2-1 issue BEFORE the chunk's MFMAs, their SIMD partners (wave + NW/2)
    // between the chunk's two 16-channel k-steps, so that one of the two feeds the pipe meanwhile ($MMIF_ABLATE x3= bit 4: all up front)
    const bool issue_late = KK >= 2 && wave >= NW / 2 && !(abl & 16);
    for (int s = 0; s < nsteps; ++s) {
        if (s + 1 < nsteps && !issue_late) issue(s + 1);
        if constexpr (F16) {   // this chunk was staged with e_cur <= the exponent of the item's earlier chunks: bring the accumulators down
            if (s % nch == 0) e_acc = e_cur;
            else if (e_cur < e_acc) {
                const float f = x3_pow2(max(e_cur - e_acc, -126));
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int j = 0; j < RJ; ++j) acc[m][j] *= f;
                if constexpr (M16) {
#pragma unroll
                    for (int j = 0; j < RJ; ++j) { acc16[j][0] *= f; acc16[j][1] *= f; }
                }
                e_acc = e_cur;
            }
        }
        // ---------------- the chunk's KK x 9 taps: operand fragments of tap t+1 are fetched while tap t's MFMAs run ----------------
        const int nkk = min(KK, nch16 - (s % nch) * KK);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            if (kk == 1 && issue_late && s + 1 < nsteps) issue(s + 1);
            if constexpr (M16) {
                if (!((abl & 8) && s > 0)) {
                    const x3_u4* s_in = s_buf;
                    const x3_u4* s_w = s_buf + NP * ING;
                    const int i16 = lane & 15, t2 = (lane >> 5) & 1, cb16 = (lane >> 4) & 1;   // k-group l >> 4 = 2 * (tap of the pair) + channel block
                    const int b0 = cb16 * PL + RJ * wave * X3_IW + i16, a0 = cb16 * MBW + i16;
#pragma unroll
                    for (int pr = 0; pr < 5; ++pr) {
                        constexpr int dummy = 0; (void)dummy;
                        const int tA = 2 * pr, tB = 2 * pr + 1 < TAPS ? 2 * pr + 1 : 2 * pr;   // (the pad half of pair 4 reads tap 8 again and is zeroed)
                        const int tl = t2 ? tB : tA;
                        const int ul = tl / 3, vl = tl - 3 * ul;
                        x3_bf16x8 af[NP], bf[RJ][2][NP];
                        const bool pad = (2 * pr + 1 >= TAPS) && t2;
#pragma unroll
                        for (int p = 0; p < NP; ++p) {
                            x3_u4 w = s_w[p * WG + tl * 2 * MBW + a0];
                            if (pad) w = (x3_u4){0u, 0u, 0u, 0u};
                            af[p] = x3_frag(w);
                        }
#pragma unroll
                        for (int j = 0; j < RJ; ++j)
#pragma unroll
                            for (int h = 0; h < 2; ++h)
#pragma unroll
                                for (int p = 0; p < NP; ++p) bf[j][h][p] = x3_frag(s_in[p * ING + b0 + (j + ul) * X3_IW + 16 * h + vl]);
#pragma unroll
                        for (int q = 0; q < NPROD; ++q)
#pragma unroll
                            for (int j = 0; j < RJ; ++j)
#pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    if constexpr (F16)
                                        acc16[j][h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(x3_h8, af[X3Prod<NP>::A[q]]),
                                                                                             __builtin_bit_cast(x3_h8, bf[j][h][X3Prod<NP>::B[q]]), acc16[j][h], 0, 0, 0);
                                    else
                                        acc16[j][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[X3Prod<NP>::A[q]], bf[j][h][X3Prod<NP>::B[q]], acc16[j][h], 0, 0, 0);
                                }
                    }
                }
            } else
            if (kk < nkk && !((abl & 8) && s > 0)) {
                const x3_u4* s_in = s_buf + kk * 2 * PL;
                const x3_u4* s_w = s_buf + NP * ING + kk * NP * WG;
                x3_bf16x8 brow[RJ + 2 * PD][NP], afr[2][MB][NP];
                // taps in column-major order t = 3 v + u: tap (u, v) reads rows u .. u+RJ-1 of the wave's (RJ+2)-row window at column shift v
                auto ld_a = [&](int t, int slot) {
                    const int v = t / KS, u = t % KS, tap = u * KS + v;
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int p = 0; p < NP; ++p) afr[slot][m][p] = x3_frag(s_w[p * WG + abase + tap * 2 * MBW + m * 32]);
                };
                auto ld_b = [&](int row, int v) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) brow[row][p] = x3_frag(s_in[p * ING + bbase + row * X3_IW + v]);
                };
                ld_a(0, 0);
#pragma unroll
                for (int r = 0; r < RJ; ++r) ld_b(r, 0);
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const int u = t % KS;
                    int nld = 0;
                    // RJ > 2: rows 2 .. RJ-1 of a new column cannot be fetched during the previous tap (it still reads them at the old
                    // column shift): they are fetched here and consumed by the SECOND half of this tap's MFMAs (rows j >= 2)
                    if (RJ > 2 && u == 0 && t > 0) {
#pragma unroll
                        for (int r = 2; r < RJ; ++r) ld_b(r, t / KS);
                        nld += (RJ - 2) * NP;
                    }
                    if (t + 1 < TAPS) {
                        const int v1 = (t + 1) / KS, u1 = (t + 1) % KS;
                        ld_a(t + 1, (t + 1) & 1);
                        nld += MB * NP;
                        if (u1 == 0) {   // rows 0, 1 of the next column are dead from this tap (u = 2) on
#pragma unroll
                            for (int r = 0; r < 2; ++r) ld_b(r, v1);
                            nld += 2 * NP;
                        } else { ld_b(u1 + RJ - 1, v1); nld += NP; }
                    }
#pragma unroll
                    for (int j0 = 0; j0 < RJ; j0 += 2)
#pragma unroll
                        for (int q = 0; q < NPROD; ++q)
#pragma unroll
                            for (int j = j0; j < j0 + 2; ++j)
#pragma unroll
                                for (int m = 0; m < MB; ++m)
                                    acc[m][j] = x3_mfma<F16>(afr[t & 1][m][X3Prod<NP>::A[q]], brow[u + j][X3Prod<NP>::B[q]], acc[m][j]);
                    // the prefetch reads ride between this tap's first MFMAs
                    for (int i = 0; i < nld; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                    }
                    for (int i = nld; i < NPROD * RJ * MB; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                if constexpr (DGRAD && KS == 3 && RJ <= 2) {
                    if (org) {
                        // ---- reflect-padding adjoint of this sub-chunk (see the kernel's head): the tile of this step
                        const int ti_f = s / per_tile, tile_f = tw.first + ti_f * tw.stride;
                        const int tt_f = tile_f % tiles_per_img;
                        const int yw = (tt_f / tiles_x) * X3_TH + 1 + RJ * wave, xl0 = (tt_f % tiles_x) * X3_TW + 1;   // stored row of j = 0 / column of lane 0
                        const int jt = 2 - yw, jb = tout.hs - 3 - yw;                // the wave's row that is stored row 2 / hs - 3 (if in 0 .. RJ-1)
                        const int lt = 2 - xl0, lr = tout.ws - 3 - xl0;              // the lane that is stored column 2 / ws - 3 (if in 0 .. 31)
                        const bool has_t = jt >= 0 && jt < RJ, has_b = jb >= 0 && jb < RJ, has_l = lt >= 0 && lt < X3_TW, has_r = lr >= 0 && lr < X3_TW;
                        if (has_t || has_b || has_l || has_r) {
                            const x3_u4 z4 = {0u, 0u, 0u, 0u};
                            auto fa = [&](int tap, int m, int p) { return x3_frag(s_w[p * WG + abase + tap * 2 * MBW + m * 32]); };
                            auto fb = [&](int row, int v, int p, int lsel) {      // window row, column shift; lsel >= 0: that lane's column only
                                x3_u4 q = s_in[p * ING + bbase + row * X3_IW + v];
                                if (lsel >= 0 && nl != lsel) q = z4;
                                return x3_frag(q);
                            };
                            auto row_fold = [&](int j, int tap_row, int brow_u) {   // ring row -> output row j: taps (tap_row, v) on the fragments of taps (brow_u, v)
#pragma unroll
                                for (int v = 0; v < 3; ++v) {
                                    x3_bf16x8 b[NP];
#pragma unroll
                                    for (int p = 0; p < NP; ++p) b[p] = fb(brow_u + j, v, p, -1);
#pragma unroll
                                    for (int m = 0; m < MB; ++m) {
                                        x3_bf16x8 a[NP];
#pragma unroll
                                        for (int p = 0; p < NP; ++p) a[p] = fa(tap_row * 3 + v, m, p);
#pragma unroll
                                        for (int q = 0; q < NPROD; ++q) {
                                            if (j == 0) acc[m][0] = x3_mfma<F16>(a[X3Prod<NP>::A[q]], b[X3Prod<NP>::B[q]], acc[m][0]);
                                            else acc[m][RJ > 1 ? 1 : 0] = x3_mfma<F16>(a[X3Prod<NP>::A[q]], b[X3Prod<NP>::B[q]], acc[m][RJ > 1 ? 1 : 0]);
                                        }
                                    }
                                }
                            };
                            auto col_fold = [&](int tap_col, int b_v, int lsel) {   // ring column -> the lane lsel of every row: taps (u, tap_col) on the fragments of taps (u, b_v)
#pragma unroll
                                for (int u = 0; u < 3; ++u)
#pragma unroll
                                    for (int j = 0; j < RJ; ++j) {
                                        x3_bf16x8 b[NP];
#pragma unroll
                                        for (int p = 0; p < NP; ++p) b[p] = fb(u + j, b_v, p, lsel);
#pragma unroll
                                        for (int m = 0; m < MB; ++m) {
                                            x3_bf16x8 a[NP];
#pragma unroll
                                            for (int p = 0; p < NP; ++p) a[p] = fa(u * 3 + tap_col, m, p);
#pragma unroll
                                            for (int q = 0; q < NPROD; ++q) acc[m][j] = x3_mfma<F16>(a[X3Prod<NP>::A[q]], b[X3Prod<NP>::B[q]], acc[m][j]);
                                        }
                                    }
                            };
                            auto corner = [&](int j, int tap, int brow, int b_v, int lsel) {
                                x3_bf16x8 b[NP];
#pragma unroll
                                for (int p = 0; p < NP; ++p) b[p] = fb(brow, b_v, p, lsel);
#pragma unroll
                                for (int m = 0; m < MB; ++m) {
                                    x3_bf16x8 a[NP];
#pragma unroll
                                    for (int p = 0; p < NP; ++p) a[p] = fa(tap, m, p);
#pragma unroll
                                    for (int q = 0; q < NPROD; ++q) {
                                        if (j == 0) acc[m][0] = x3_mfma<F16>(a[X3Prod<NP>::A[q]], b[X3Prod<NP>::B[q]], acc[m][0]);
                                        else acc[m][RJ > 1 ? 1 : 0] = x3_mfma<F16>(a[X3Prod<NP>::A[q]], b[X3Prod<NP>::B[q]], acc[m][RJ > 1 ? 1 : 0]);
                                    }
                                }
                            };
                            if (has_t) {   // ring row 0 -> stored row 2: taps (2, v) on G row 1 = the fragments of this row's taps (0, v)
                                row_fold(jt, 2, 0);
                                if (has_l) corner(jt, 8, jt + 0, 0, lt);
                                if (has_r) corner(jt, 6, jt + 0, 2, lr);
                            }
                            if (has_b) {   // ring row hs-1 -> stored row hs-3: taps (0, v) on G row hs-2 = the fragments of this row's taps (2, v)
                                row_fold(jb, 0, 2);
                                if (has_l) corner(jb, 2, jb + 2, 0, lt);
                                if (has_r) corner(jb, 0, jb + 2, 2, lr);
                            }
                            if (has_l) col_fold(2, 0, lt);   // ring column 0 -> stored column 2: taps (u, 2) on G column 1 = the fragments of taps (u, 0)
                            if (has_r) col_fold(0, 2, lr);   // ring column ws-1 -> stored column ws-3: taps (u, 0) on the fragments of taps (u, 2)
                        }
                    }
                }
            }
        }
        // ---------------- epilogue after the item's last chunk ----------------
        if (M16 && s % nch == nch - 1) {
            // C layout of the 16 x 16 x 32 MFMA: lane l, reg r = output channel 4 (l >> 4) + r of pixel l & 15: one half granule per lane
            const int ti = s / per_tile;
            const int tile = tw.first + ti * tw.stride;
            const int in_ = tile / tiles_per_img, tt = tile - in_ * tiles_per_img;
            const int ys0 = (tt / tiles_x) * X3_TH, xs0 = (tt % tiles_x) * X3_TW;
            const int g4 = lane >> 4, ocb = g4 >> 1, half = g4 & 1;
            const bool has_cb = ocb < tout.cb;
            const bool masked = DGRAD && ((mask_bits >> ocb) & 1ull), accum = DGRAD && ((accum_bits >> ocb) & 1ull);
#pragma unroll
            for (int j = 0; j < RJ; ++j) {
                const int ys = ys0 + RJ * wave + j;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int xs = xs0 + 16 * h + (lane & 15);
                    const bool inside = ys < tout.hs && xs < tout.ws && has_cb;
                    x3_f4 v = acc16[j][h];
                    if (!DGRAD) {
                        if constexpr (F16) v *= x3_pow2(max(-(e_acc + X3_SW_EXP), -126));
                        v += *reinterpret_cast<const x3_f4*>(reinterpret_cast<const float*>(s_buf + BUF_G) + 4 * g4);
                        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    } else {
                        const int ysc = min(ys, tout.hs - 1), xsc = min(xs, tout.ws - 1), ocbc = min(ocb, tout.cb - 1);
                        if (accum_bits != 0ull) {
                            const x3_f4 old = *reinterpret_cast<const x3_f4*>(tout.base + tout.gidx(in_, ocbc, ysc, xsc) * 32 + half * 16);
                            if (accum) v += old;
                        }
                        if (mask_bits != 0ull) {
                            const int oy = min(max(reflect_idx(ys - tout.halo, tmask.h), 0), tmask.h - 1);
                            const int ox = min(max(reflect_idx(xs - tout.halo, tmask.w), 0), tmask.w - 1);
                            x3_f4 xm;
                            if (signs != nullptr) {
                                const unsigned sg = signs[(((long long)in_ * ((tmask.cb + 3) >> 2) + 0) * tmask.h + oy) * tmask.w + ox];
                                const unsigned nib = (sg >> (8 * ocb + 4 * half)) & 15u;
                                xm = (x3_f4){(nib & 1u) ? 1.f : 0.f, (nib & 2u) ? 1.f : 0.f, (nib & 4u) ? 1.f : 0.f, (nib & 8u) ? 1.f : 0.f};
                            } else {
                                const char* pm = (masked && ocb < tmask.cb) ? tmask.base + tmask.gidx(in_, ocb, oy, ox) * 32 + half * 16 : tmask.base;
                                xm = *reinterpret_cast<const x3_f4*>(pm);
                            }
                            if (masked) {
                                v.x = xm.x > 0.f ? v.x : 0.f; v.y = xm.y > 0.f ? v.y : 0.f;
                                v.z = xm.z > 0.f ? v.z : 0.f; v.w = xm.w > 0.f ? v.w : 0.f;
                            }
                        }
                    }
                    if (inside) *reinterpret_cast<x3_f4*>(tout.base + tout.gidx(in_, ocb, ys, xs) * 32 + half * 16) = v;
                    acc16[j][h] = (x3_f4){0.f, 0.f, 0.f, 0.f};
                }
            }
        } else
        if (s % nch == nch - 1) {
            const int mb = (s / nch) % nmb, ti = s / per_tile;
            const int tile = tw.first + ti * tw.stride;
            const int in_ = tile / tiles_per_img, tt = tile - in_ * tiles_per_img;
            const int ys0 = (tt / tiles_x) * X3_TH + org, xs0 = (tt % tiles_x) * X3_TW + org;
            const int xs = xs0 + nl, half = lane >> 5;
#pragma unroll
            for (int j = 0; j < RJ; ++j) {
                const int ys = ys0 + RJ * wave + j;
                const bool inside = ys < tout.hs - org && xs < tout.ws - org;
                const int oy = DGRAD ? min(max(reflect_idx(ys - tout.halo, tmask.h), 0), tmask.h - 1) : 0;
                const int ox = DGRAD ? min(max(reflect_idx(xs - tout.halo, tmask.w), 0), tmask.w - 1) : 0;
                const int ysc = min(ys, tout.hs - 1), xsc = min(xs, tout.ws - 1);   // in-range twin of the lane's position: every load below is
#pragma unroll                                                                      // unconditional per lane (a load under its own branch gets
                for (int m = 0; m < MB; ++m) {                                      // its own s_waitcnt vmcnt(0): 16 serial round trips per item)
                    x3_f4 old[4], xm[4];
                    if (DGRAD) {   // the four accumulate / mask operands of this 32-channel group in ONE round trip, then the arithmetic
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            old[q] = (x3_f4){0.f, 0.f, 0.f, 0.f};
                            xm[q] = (x3_f4){1.f, 1.f, 1.f, 1.f};
                        }
                        if (accum_bits != 0ull) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int ocbc = min((mb * MB + m) * 4 + q, tout.cb - 1);
                                old[q] = *reinterpret_cast<const x3_f4*>(tout.base + tout.gidx(in_, ocbc, ysc, xsc) * 32 + half * 16);
                            }
                        }
                        if (mask_bits != 0ull && signs != nullptr) {   // the sign map wgrad_x3_kernel left: one dword = the 4 blocks of this group
                            // (the last 64-wide m-block of a layer with cin % 64 in (0, 32] has a second group past the map: clamp the
                            // group like the `old` loads clamp their block -- its values are never stored, ocb >= tout.cb)
                            const int ngrp = (tmask.cb + 3) >> 2, grp = min(mb * MB + m, ngrp - 1);
                            const unsigned sg = signs[(((long long)in_ * ngrp + grp) * tmask.h + oy) * tmask.w + ox];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const unsigned nib = (sg >> (8 * q + 4 * half)) & 15u;
                                xm[q] = (x3_f4){(nib & 1u) ? 1.f : 0.f, (nib & 2u) ? 1.f : 0.f, (nib & 4u) ? 1.f : 0.f, (nib & 8u) ? 1.f : 0.f};
                            }
                        } else if (mask_bits != 0ull) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {   // blocks the mask does not cover read one hot line (uniform address select, no branch)
                                const int ocb = (mb * MB + m) * 4 + q;
                                const bool need = ((mask_bits >> ocb) & 1ull) && ocb < tmask.cb;
                                const char* pm = need ? tmask.base + tmask.gidx(in_, ocb, oy, ox) * 32 + half * 16 : tmask.base;
                                xm[q] = *reinterpret_cast<const x3_f4*>(pm);
                            }
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int ocb = (mb * MB + m) * 4 + q;
                            if (!((accum_bits >> ocb) & 1ull)) old[q] = (x3_f4){0.f, 0.f, 0.f, 0.f};
                            if (!((mask_bits >> ocb) & 1ull)) xm[q] = (x3_f4){1.f, 1.f, 1.f, 1.f};
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int ocb = (mb * MB + m) * 4 + q;          // channel block of the out view
                        x3_f4 v = {acc[m][j][4 * q], acc[m][j][4 * q + 1], acc[m][j][4 * q + 2], acc[m][j][4 * q + 3]};
                        if (!DGRAD) {
                            const x3_f4 b4 = *reinterpret_cast<const x3_f4*>(reinterpret_cast<const float*>(s_buf + BUF_G) + ocb * 8 + 4 * half);
                            if constexpr (F16) v *= x3_pow2(max(-(e_acc + X3_SW_EXP), -126));   // the accumulators hold 2^(e_acc + 10) z
                            v += b4;
                            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        } else {
                            v += old[q];
                            v.x = xm[q].x > 0.f ? v.x : 0.f; v.y = xm[q].y > 0.f ? v.y : 0.f;
                            v.z = xm[q].z > 0.f ? v.z : 0.f; v.w = xm[q].w > 0.f ? v.w : 0.f;
                        }
                        if (inside && ocb < tout.cb)
                            *reinterpret_cast<x3_f4*>(tout.base + tout.gidx(in_, ocb, ys, xs) * 32 + half * 16) = v;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;
                }
            }
        }
        if (s + 1 < nsteps) {
            if constexpr (F16) tile_max();
            __syncthreads();   // every wave has finished reading the tile
            if constexpr (F16) {
                stage_scale((s + 1) % nch == 0);
                e_cur = e_stage;
            }
            commit();
        }
        __syncthreads();
    }
}


// ------------------------------------------------------------------ wgrad
constexpr int XW_TW = 16;                        // pixel-tile columns = the 16 pixels of one k-step
constexpr int XW_THREADS = 768;                  // (the 12-wave instantiations; wgrad_x3_kernel's NWV)

// TH x 16 pixel tiles, NXC / NGC channel blocks of the activation / gradient group kept in LDS.  <8, 8, 8>: the general 64 x 64 channel
// pair.  <16, 6, 2>: layers with <= 48 input and <= 16 output channels (the DenseBlock convs, decode.3) -- the same LDS and staging
// budget spent on TWICE the pixels per tile: those layers run one tile per ~3 us of global-load latency whatever the tile holds
// (0.18-0.30 ms per launch with almost no MFMA work), so half the tiles is half the time.  Channel blocks past NXC / NGC are read
// from the last plane kept (those dW rows / columns belong to channels the layer does not have and are never reduced).
template <int TH, int NXC, int NGC, int KS, bool SIGNS = false, int NWV = 12>
__global__ __launch_bounds__(64 * NWV, 12 / NWV) void wgrad_x3_kernel(TV tx, TV tg, float* __restrict__ partial, int cin, int cout, int tiles_x, int tpi,
                                                               int total, int G, int n_icg, int n_ocg, unsigned* __restrict__ signs) {
    constexpr int XW_THREADS = 64 * NWV;             // 12 waves (v, jt, mt).  (NWV = 6, <8, 4, 2>: six-wave blocks, two per CU, for cin <= 32 -- measured
                                                     // slower, 0.224 vs 0.188 ms on 16 -> 16: twice the tiles cost more than the second block hides; not instantiated)
    constexpr int PD = KS / 2, TAPS = KS * KS;       // KS = 1: no halo, one tap; the three wave groups v split the k-steps instead of the tap columns
    constexpr int XW_TH = TH, XW_XH = TH + 2 * PD, XW_XW = XW_TW + 2 * PD;
    constexpr int XW_PER = 64 * 64 * TAPS + 64;      // floats per block partial: dW[64 oc][64 ic][taps], db[64]
    constexpr int XW_XPL = XW_XH * XW_XW + (KS == 1 ? 4 : 0);   // 180 / 324 / 132 granules per x plane (= 4 mod 16: the four planes a half wave's transposing
    constexpr int XW_GPL = XW_TH * XW_TW + 4;        // 132 / 260 granules per g plane    read touches fall on disjoint bank quarters)
    constexpr int XW_XG = NXC * XW_XPL, XW_GG = NGC * XW_GPL;      // one precision half of the x / g tile
    constexpr int XW_BUF_G = 2 * XW_XG + 2 * XW_GG;                // [x hi][x lo][g hi][g lo]: 4992 / 4928 granules
    constexpr int XW_ROUNDS = (NXC * XW_XH * XW_XW + NGC * XW_TH * XW_TW + XW_THREADS - 1) / XW_THREADS;   // 4
    static_assert(2 * XW_BUF_G * 16 <= 160 * 1024, "two tile buffers must fit the LDS");
    __shared__ __attribute__((aligned(16))) x3_u4 s_buf[2 * XW_BUF_G];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int npairs = n_icg * n_ocg;
    const int b = blockIdx.x;
    int gi, pair;
    if ((G & 7) == 0) { pair = (b >> 3) % npairs; gi = ((b >> 3) / npairs) * 8 + (b & 7); }   // blocks sharing tiles: same XCD
    else { pair = b % npairs; gi = b / npairs; }
    const int icg = pair % n_icg, ocg = pair / n_icg;
    const TileWalk tw = xcd_walk(total, G, gi);
    const int ntile = tw.count;
    const int abl = tiles_x >> 16;     // timing ablations as in conv_x3_kernel (1 no loads, 4 no split + LDS writes, 8 no MFMAs)
    tiles_x &= 0xffff;

    // wave (v, jt, mt): tap COLUMN v, input-channel tile jt, output-channel tile mt (32 each); it owns the three taps (0..2, v): the
    // activation row a tap (u, v) needs at k-step ry is tile row ry + u, i.e. a three-row register ring with ONE new row per k-step
    // (2 + 2 fragments per 9 MFMAs; with a fixed tap ROW per wave every k-step re-read all three column shifts: 6 + 2 -- 111 B / clk
    // of transposing reads on four SIMDs, at the LDS's limit for three waves per SIMD)
    const int v = wave % 3, jt = (wave / 3) & 1, mt = wave / 6;
    // (waves whose channel tile the layer does not have -- 9 of 12 for a 16 -> 16 DenseBlock conv -- only stage.  Splitting the tile's
    // k-steps among them instead, partial sums meeting in LDS after the last tile, was built and measured: 0.185 vs 0.187 ms -- those
    // layers wait for their tile loads, not for the matrix pipe)
    const bool active = (ocg * 64 + mt * 32 < cout) && (icg * 64 + jt * 32 < cin);
    const bool want_db = ((v == 0 || KS == 1) && jt == 0 && icg == 0 && ocg * 64 + mt * 32 < cout);

    // channel blocks this block's groups really have: only those are staged (the planes of the others stay zero from here on)
    const int nxcb = min(NXC, tx.cb - icg * 8), ngcb = min(NGC, tg.cb - ocg * 8);
    const int n_x = nxcb * (XW_XH * XW_XW), n_all = n_x + ngcb * (XW_TH * XW_TW);
    for (int i = tid; i < 2 * XW_BUF_G; i += XW_THREADS) s_buf[i] = (x3_u4){0u, 0u, 0u, 0u};
    unsigned geo[XW_ROUNDS];   // tile independent: is-g << 24 | channel block << 16 | tile row << 8 | tile column
#pragma unroll
    for (int k = 0; k < XW_ROUNDS; ++k) {
        const int e = min(tid + XW_THREADS * k, n_all - 1);
        if (e < n_x) {
            const int cb = e / (XW_XH * XW_XW), rem = e - cb * (XW_XH * XW_XW), py = rem / XW_XW;
            geo[k] = (unsigned)(cb << 16 | py << 8 | (rem - py * XW_XW));
        } else {
            const int e2 = e - n_x;
            const int cb = e2 / (XW_TH * XW_TW), rem = e2 - cb * (XW_TH * XW_TW), py = rem / XW_TW;
            geo[k] = (unsigned)(1u << 24 | cb << 16 | py << 8 | (rem - py * XW_TW));
        }
    }

    // staging of the next tile in two halves (rounds 0-1 during k-steps 0-3, rounds 2-3 during k-steps 4-7: the other LDS buffer is free for
    // the whole tile), so only two granules per thread are in flight -- four kept the kernel 20 registers over its budget (scratch)
    constexpr int HALF_R = (XW_ROUNDS + 1) / 2;
    int k_done = 0;
    X3Gran rin[HALF_R];
    // ReLU sign map of x for the dgrad that follows (mmif_conv2d_reflect_bwd_wide on fp32 tensors): one byte per pixel and channel block,
    // bit i = channel i > 0, as [n][ceil(cb / 4)][h][w][4 blocks]; every x granule passes through this kernel's staging anyway -- the blocks
    // of output-channel group 0 write the interior pixels of their tiles.  The dgrad then masks with 1/32 of the bytes of x itself.
    const bool write_signs = SIGNS && signs != nullptr && ocg == 0;   // (a template switch: the map costs the kernel ~5 registers it does not have)
    int c_in = 0, c_y0 = 0, c_x0 = 0;   // the tile whose staging registers are about to be committed
    auto issue = [&](int k_tile, int half) {
        const int tile = tw.first + k_tile * tw.stride;
        const int in_ = tile / tpi, tt = tile - in_ * tpi;
        const int y0 = (tt / tiles_x) * XW_TH, x0 = (tt % tiles_x) * XW_TW;
        c_in = in_; c_y0 = y0; c_x0 = x0;
        const char* bx = tx.base + ((long long)in_ * tx.img + (long long)(tx.cb_off + icg * 8) * tx.plane) * 32;
        const char* bg = tg.base + ((long long)in_ * tg.img + (long long)(tg.cb_off + ocg * 8) * tg.plane) * 32;
        const unsigned xplane = (unsigned)tx.plane, gplane = (unsigned)tg.plane;
#pragma unroll
        for (int kr = 0; kr < HALF_R; ++kr) {
            const int k = HALF_R * half + kr;
            if (k >= XW_ROUNDS) continue;
            const unsigned gk = half ? geo[min(HALF_R + kr, XW_ROUNDS - 1)] : geo[kr];
            const int cb = (int)((gk >> 16) & 255u), py = (int)((gk >> 8) & 255u), px = (int)(gk & 255u);
            if ((abl & 1) && k_tile > 0) continue;
            rin[kr].a = (x3_f4){0.f, 0.f, 0.f, 0.f};
            rin[kr].b = rin[kr].a;
            if (tid + XW_THREADS * k < n_all) {
                if (!(gk >> 24)) {
                    const int y = min(max(reflect_idx(y0 + py - PD, tx.h), 0), tx.h - 1);
                    const int x = min(max(reflect_idx(x0 + px - PD, tx.w), 0), tx.w - 1);
                    const x3_f4* p = reinterpret_cast<const x3_f4*>(bx + ((unsigned)cb * xplane + (unsigned)(y * tx.ws + x)) * 32u);   // (32-bit: x3_small)
                    rin[kr].a = p[0];
                    rin[kr].b = p[1];
                } else {
                    const int y = y0 + py, x = x0 + px;
                    if (y < tg.h && x < tg.w) {
                        const x3_f4* p = reinterpret_cast<const x3_f4*>(bg + ((unsigned)cb * gplane + (unsigned)((y + tg.halo) * tg.ws + x + tg.halo)) * 32u);
                        rin[kr].a = p[0];
                        rin[kr].b = p[1];
                    }
                }
            }
        }
    };
    auto commit = [&](int buf, int half) {
        if ((abl & 4) && k_done > 1) return;
        x3_u4* dst = s_buf + buf * XW_BUF_G;
#pragma unroll
        for (int kr = 0; kr < HALF_R; ++kr) {
            const int k = HALF_R * half + kr;
            if (k >= XW_ROUNDS) continue;
            const unsigned gk = half ? geo[min(HALF_R + kr, XW_ROUNDS - 1)] : geo[kr];
            if (tid + XW_THREADS * k < n_all) {
                const int cb = (int)((gk >> 16) & 255u), py = (int)((gk >> 8) & 255u), px = (int)(gk & 255u);
                x3_u4 pc[2];
                x3_split_gran<2>(rin[kr], pc);
                if (!(gk >> 24)) {
                    const int o = cb * XW_XPL + py * XW_XW + px;
                    dst[o] = pc[0];
                    dst[XW_XG + o] = pc[1];
                    if (SIGNS && write_signs) {
                        const int y = c_y0 + py - PD, x = c_x0 + px - PD;
                        if (py >= PD && py < PD + XW_TH && px >= PD && px < PD + XW_TW && y < tx.h && x < tx.w) {
                            const X3Gran& g = rin[kr];
                            const unsigned bits = (g.a.x > 0.f) | (g.a.y > 0.f) << 1 | (g.a.z > 0.f) << 2 | (g.a.w > 0.f) << 3 | (g.b.x > 0.f) << 4 |
                                                  (g.b.y > 0.f) << 5 | (g.b.z > 0.f) << 6 | (g.b.w > 0.f) << 7;
                            const int cbg = icg * 8 + cb;
                            // (32-bit offset: the map is 1/32 of x, which x3_small keeps below 2^31 bytes)
                            reinterpret_cast<unsigned char*>(signs)[(((unsigned)(c_in * ((tx.cb + 3) >> 2) + (cbg >> 2)) * (unsigned)tx.h + (unsigned)y) * (unsigned)tx.w + (unsigned)x) * 4u + (unsigned)(cbg & 3)] =
                                (unsigned char)bits;
                        }
                    }
                } else {
                    const int o = 2 * XW_XG + cb * XW_GPL + py * XW_TW + px;
                    dst[o] = pc[0];
                    dst[XW_GG + o] = pc[1];
                }
            }
        }
    };

    f32x16 acc[3];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; acc[2][r] = 0.f; }
    // bias gradient = row sums of the gradient fragments, on TWO v_mfma_f32_16x16x32 (8 accumulator registers instead of the 16 of a
    // 32 x 32 tile -- this kernel lives at its 168-register budget): read as a 16x16x32 A operand, lane l of the 32x32x16 fragment is
    // row l & 15, k-group l >> 4, i.e. k-groups 0 / 2 hold channels 0-15 (pixels 0-7 / 8-15) and k-groups 1 / 3 channels 16-31; a B
    // operand of ones in k-groups {0, 2} (resp. {1, 3}) and zeros elsewhere sums exactly one half
    x3_f4 accb[2] = {(x3_f4){0.f, 0.f, 0.f, 0.f}, (x3_f4){0.f, 0.f, 0.f, 0.f}};
    const unsigned one2 = 0x3f803f80u;
    const unsigned sel0 = ((lane >> 4) & 1) ? 0u : one2, sel1 = ((lane >> 4) & 1) ? one2 : 0u;
    const x3_bf16x8 ones0 = __builtin_bit_cast(x3_bf16x8, ((x3_u4){sel0, sel0, sel0, sel0}));
    const x3_bf16x8 ones1 = __builtin_bit_cast(x3_bf16x8, ((x3_u4){sel1, sel1, sel1, sel1}));

    // transposing read: within a 16-lane group, lane sl supplies the address of (pixel sl >> 2, 8-byte piece sl & 3 of the 16-channel
    // record = channel block (sl & 3) >> 1, byte 8 * (sl & 1)) and receives channel sl of pixels 0..3 (tests/test_gpu_probe.py)
    const int sl = lane & 15, chalf = (lane >> 4) & 1, kg = lane >> 5;
    const int lane_cb = 2 * chalf + ((sl & 3) >> 1), lane_byte = (sl & 1) * 8, lane_px = 8 * kg + (sl >> 2);

    __syncthreads();   // the zero fill
    if (ntile > 0) {
        issue(0, 0);
        commit(0, 0);
        issue(0, 1);
        commit(0, 1);
    }
    __syncthreads();
    for (int k = 0; k < ntile; ++k) {
        const int buf = k & 1;
        const bool more = k + 1 < ntile;
        k_done = k;
        if (more) issue(k + 1, 0);
        if (active && !((abl & 8) && k > 0)) {
            const char* s_xh = reinterpret_cast<const char*>(s_buf + buf * XW_BUF_G);
            const char* s_xl = s_xh + XW_XG * 16;
            const char* s_gh = s_xl + XW_XG * 16;
            const char* s_gl = s_gh + XW_GG * 16;
            auto ld_tr = [&](const char* base) {
                const x3_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(X3_LDS_PTR(x3_s16x4, base));
                const x3_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(X3_LDS_PTR(x3_s16x4, base + 4 * 16));
                return __builtin_bit_cast(x3_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            };
            const int goff = ((min(mt * 4 + lane_cb, NGC - 1) * XW_GPL) + lane_px) * 16 + lane_byte;          // + ry * XW_TW * 16
            const int xoff = ((min(jt * 4 + lane_cb, NXC - 1) * XW_XPL) + lane_px + (KS == 1 ? 0 : v)) * 16 + lane_byte;      // + row * XW_XW * 16
            if constexpr (KS == 1) {
                // one tap: wave group v takes the k-steps ry = v, v + 3, .. (its partial goes to its own slot, summed by the reduce kernel)
#pragma unroll
                for (int ry = 0; ry < XW_TH; ++ry) {
                    if (ry == XW_TH / 2 && more) {   // second half of the next tile's staging
                        commit(buf ^ 1, 0);
                        issue(k + 1, 1);
                    }
                    if (ry % 3 == v) {
                        const x3_bf16x8 xh = ld_tr(s_xh + xoff + ry * XW_XW * 16), xl = ld_tr(s_xl + xoff + ry * XW_XW * 16);
                        const x3_bf16x8 gh = ld_tr(s_gh + goff + ry * XW_TW * 16), gl = ld_tr(s_gl + goff + ry * XW_TW * 16);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gl, xh, acc[0], 0, 0, 0);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, xl, acc[0], 0, 0, 0);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, xh, acc[0], 0, 0, 0);
                        if (want_db) {
                            accb[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gl, ones0, accb[0], 0, 0, 0);
                            accb[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gl, ones1, accb[1], 0, 0, 0);
                            accb[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh, ones0, accb[0], 0, 0, 0);
                            accb[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh, ones1, accb[1], 0, 0, 0);
                        }
                    }
                }
            } else {
            // activation rows ry, ry + 1, ry + 2 live in a three-slot register ring (slot = row % 3): ONE new row per k-step
            x3_bf16x8 xh[3], xl[3];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                xh[r] = ld_tr(s_xh + xoff + r * XW_XW * 16);
                xl[r] = ld_tr(s_xl + xoff + r * XW_XW * 16);
            }
#pragma unroll
            for (int ry = 0; ry < XW_TH; ++ry) {
                if (ry == XW_TH / 2 && more) {   // second half of the next tile's staging (the three waves of a SIMD one k-step apart: +-0)
                    commit(buf ^ 1, 0);
                    issue(k + 1, 1);
                }
                xh[(ry + 2) % 3] = ld_tr(s_xh + xoff + (ry + 2) * XW_XW * 16);
                xl[(ry + 2) % 3] = ld_tr(s_xl + xoff + (ry + 2) * XW_XW * 16);
                {
                    const x3_bf16x8 gh = ld_tr(s_gh + goff + ry * XW_TW * 16), gl = ld_tr(s_gl + goff + ry * XW_TW * 16);
#pragma unroll
                    for (int u = 0; u < 3; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gl, xh[(ry + u) % 3], acc[u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < 3; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, xl[(ry + u) % 3], acc[u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < 3; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, xh[(ry + u) % 3], acc[u], 0, 0, 0);
                    if (want_db) {
                        accb[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gl, ones0, accb[0], 0, 0, 0);
                        accb[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gl, ones1, accb[1], 0, 0, 0);
                        accb[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh, ones0, accb[0], 0, 0, 0);
                        accb[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh, ones1, accb[1], 0, 0, 0);
                    }
                }
            }
            }
        }
        else if (more) {   // (a wave without a live channel tile still stages)
            commit(buf ^ 1, 0);
            issue(k + 1, 1);
        }
        if (more) commit(buf ^ 1, 1);
        __syncthreads();
    }
    // lane l reg r of acc[u]: oc = 32 mt + (r & 3) + 8 (r >> 2) + 4 (l >> 5), ic = 32 jt + (l & 31), tap (u, v)
    if (active) {
        float* dst = partial + ((long long)(KS == 1 ? gi * 3 + v : gi) * npairs + pair) * XW_PER;
        // stored REGISTER-major, [mt][jt][v][u][r][lane] (256 contiguous bytes per store instruction; in dW order every lane wrote its own
        // 4 bytes of a different cache line: 36 lines per instruction); wgrad_x3_reduce undoes the mapping
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (KS == 1) dst[((mt * 2 + jt) * 16 + r) * 64 + lane] = acc[0][r];
            else {
#pragma unroll
                for (int u = 0; u < 3; ++u) dst[((((mt * 2 + jt) * 3 + v) * 3 + u) * 16 + r) * 64 + lane] = acc[u][r];
            }
        }
        // 16x16x32 C layout: lane l, reg r = row 4 (l >> 4) + r, column l & 15; rows of accb[h] = channels 16 h .. 16 h + 15
        if (want_db && (lane & 15) == 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[64 * 64 * TAPS + 32 * mt + 16 * h + 4 * (lane >> 4) + r] = accb[h][r];
        }
    }
}


// ------------------------------------------------------------------ wgrad, thin layers (<= 48 in, <= 16 out: the DenseBlock convs, decode.3)
// wgrad_x3_kernel gives such a layer 3 of its 12 waves to compute with (on 32 x 32 tiles of which a quarter is real), keeps ONE tile's
// loads in flight per CU and pays a 12-wave barrier per tile: 0.19-0.28 ms per launch where the bytes take 0.06-0.11.  Here: blocks of
// FOUR waves on 8 x 16-pixel tiles, three of them per CU (44 KB of LDS, < 168 registers) -- the same trick as the thin forward / dgrad
// kernels: the blocks hide each other's load latency --, 16 x 16 x 32 MFMAs (no padding: M = 16 output channels, N = 16 input channels,
// K = the 32 pixels of two tile rows) and the same two-piece products.  Work item (j, v) = input-channel block j x tap column v: it owns
// dW[16][16 j .. 16 j + 15][u = 0..2][v] in 12 accumulator registers; a wave takes items wave, wave + 4, ...
constexpr int XT_TH = 8, XT_TW = 16, XT_XH = XT_TH + 2, XT_XW = XT_TW + 2;
constexpr int XT_XPL = XT_XH * XT_XW, XT_GPL = XT_TH * XT_TW + 4;        // 180 / 132 granules per plane: both = 4 mod 16, so that the two planes a 32-lane
                                                                         // group of the transposing read touches fall on disjoint bank halves (184 was 2-way: 35 % extra cycles)
constexpr int XT_THREADS = 256;
constexpr int XT_PER = 9 * 3 * 4 * 64 + 64;                            // floats per block partial: [item][u][reg][lane], db[16] (+ pad)
template <int NXC, bool SIGNS = false>
__global__ __launch_bounds__(XT_THREADS, 3) void wgrad_x3_thin_kernel(TV tx, TV tg, float* __restrict__ partial, int tiles_x, int tpi, int total,
                                                                      unsigned* __restrict__ signs) {
    constexpr int XG = NXC * XT_XPL, GG = 2 * XT_GPL;                  // granules of one piece of the x / g tile
    constexpr int N_X = NXC * XT_XH * XT_XW, N_ALL = N_X + 2 * XT_TH * XT_TW;
    constexpr int ROUNDS = (N_ALL + XT_THREADS - 1) / XT_THREADS;       // 3 / 4 / 6
    constexpr int NITEM = 3 * (NXC / 2), MAXI = (NITEM + 3) / 4;        // 3 / 6 / 9 items, 1 / 2 / 3 per wave
    __shared__ __attribute__((aligned(16))) x3_u4 s_buf[2 * XG + 2 * GG];   // [x hi][x lo][g hi][g lo]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const TileWalk tw = xcd_walk(total, gridDim.x, blockIdx.x);
    for (int i = tid; i < 2 * XG + 2 * GG; i += XT_THREADS) s_buf[i] = (x3_u4){0u, 0u, 0u, 0u};   // (the planes of absent channel blocks stay zero)
    unsigned geo[ROUNDS];   // is-g << 24 | channel block << 16 | tile row << 8 | tile column
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const int e = min(tid + XT_THREADS * k, N_ALL - 1);
        if (e < N_X) {
            const int cb = e / (XT_XH * XT_XW), rem = e - cb * (XT_XH * XT_XW), py = rem / XT_XW;
            geo[k] = (unsigned)(cb << 16 | py << 8 | (rem - py * XT_XW));
        } else {
            const int e2 = e - N_X;
            const int cb = e2 / (XT_TH * XT_TW), rem = e2 - cb * (XT_TH * XT_TW), py = rem / XT_TW;
            geo[k] = (unsigned)(1u << 24 | cb << 16 | py << 8 | (rem - py * XT_TW));
        }
    }
    X3Gran rin[ROUNDS];
    int c_in = 0, c_y0 = 0, c_x0 = 0;   // the tile in the staging registers (SIGNS: the ReLU sign map of x, as wgrad_x3_kernel writes it)
    auto issue = [&](int k_tile) {
        const int tile = tw.first + k_tile * tw.stride;
        const int in_ = tile / tpi, tt = tile - in_ * tpi;
        const int y0 = (tt / tiles_x) * XT_TH, x0 = (tt % tiles_x) * XT_TW;
        c_in = in_; c_y0 = y0; c_x0 = x0;
        const char* bx = tx.base + ((long long)in_ * tx.img + (long long)tx.cb_off * tx.plane) * 32;
        const char* bg = tg.base + ((long long)in_ * tg.img + (long long)tg.cb_off * tg.plane) * 32;
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int cb = (int)((geo[k] >> 16) & 255u), py = (int)((geo[k] >> 8) & 255u), px = (int)(geo[k] & 255u);
            rin[k].a = (x3_f4){0.f, 0.f, 0.f, 0.f};
            rin[k].b = rin[k].a;
            if (tid + XT_THREADS * k >= N_ALL) continue;
            if (!(geo[k] >> 24)) {
                if (cb < tx.cb) {
                    const int y = min(max(reflect_idx(y0 + py - 1, tx.h), 0), tx.h - 1);
                    const int x = min(max(reflect_idx(x0 + px - 1, tx.w), 0), tx.w - 1);
                    const x3_f4* p = reinterpret_cast<const x3_f4*>(bx + ((unsigned)cb * (unsigned)tx.plane + (unsigned)(y * tx.ws + x)) * 32u);
                    rin[k].a = p[0];
                    rin[k].b = p[1];
                }
            } else {
                const int y = y0 + py, x = x0 + px;
                if (cb < tg.cb && y < tg.h && x < tg.w) {
                    const x3_f4* p = reinterpret_cast<const x3_f4*>(bg + ((unsigned)cb * (unsigned)tg.plane + (unsigned)((y + tg.halo) * tg.ws + x + tg.halo)) * 32u);
                    rin[k].a = p[0];
                    rin[k].b = p[1];
                }
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            if (tid + XT_THREADS * k >= N_ALL) continue;
            const int cb = (int)((geo[k] >> 16) & 255u), py = (int)((geo[k] >> 8) & 255u), px = (int)(geo[k] & 255u);
            x3_u4 pc[2];
            x3_split_gran<2>(rin[k], pc);
            if (!(geo[k] >> 24)) {
                const int o = cb * XT_XPL + py * XT_XW + px;
                s_buf[o] = pc[0];
                s_buf[XG + o] = pc[1];
                if (SIGNS) {
                    const int y = c_y0 + py - 1, x = c_x0 + px - 1;
                    if (py >= 1 && py <= XT_TH && px >= 1 && px <= XT_TW && y < tx.h && x < tx.w && cb < tx.cb) {
                        const X3Gran& g = rin[k];
                        const unsigned bits = (g.a.x > 0.f) | (g.a.y > 0.f) << 1 | (g.a.z > 0.f) << 2 | (g.a.w > 0.f) << 3 | (g.b.x > 0.f) << 4 |
                                              (g.b.y > 0.f) << 5 | (g.b.z > 0.f) << 6 | (g.b.w > 0.f) << 7;
                        reinterpret_cast<unsigned char*>(signs)[(((unsigned)(c_in * ((tx.cb + 3) >> 2) + (cb >> 2)) * (unsigned)tx.h + (unsigned)y) * (unsigned)tx.w + (unsigned)x) * 4u + (unsigned)(cb & 3)] =
                            (unsigned char)bits;
                    }
                }
            } else {
                const int o = 2 * XG + cb * XT_GPL + py * XT_TW + px;
                s_buf[o] = pc[0];
                s_buf[GG + o] = pc[1];
            }
        }
    };

    typedef __attribute__((ext_vector_type(4))) float f32x4_;
    f32x4_ acc[MAXI][3], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
#pragma unroll
        for (int u = 0; u < 3; ++u) acc[i][u] = (f32x4_){0.f, 0.f, 0.f, 0.f};
    const x3_bf16x8 ones = __builtin_bit_cast(x3_bf16x8, ((x3_u4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}));
    // operand fragments of a 16 x 16 x 32 MFMA: lane l = channel l & 15, k-group l >> 4 = pixels 8 (l >> 4) .. + 7 of the k-step's 32
    // (two tile rows of 16); the transposing read (see wgrad_x3_kernel): lane sl of a 16-lane group addresses pixel sl >> 2 (+ 4 for the
    // second read), channel block (sl & 3) >> 1 of the 16-channel record, byte 8 (sl & 1), and receives channel sl of four pixels
    const int sl = lane & 15, g4 = lane >> 4;
    const int l_cb = (sl & 3) >> 1, l_byte = (sl & 1) * 8, l_row = g4 >> 1, l_col = 8 * (g4 & 1) + (sl >> 2);
    auto ld_tr = [&](const char* base) {
        const x3_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(X3_LDS_PTR(x3_s16x4, base));
        const x3_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(X3_LDS_PTR(x3_s16x4, base + 4 * 16));
        return __builtin_bit_cast(x3_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    const char* s_xh = reinterpret_cast<const char*>(s_buf);
    const char* s_xl = s_xh + XG * 16;
    const char* s_gh = s_xl + XG * 16;
    const char* s_gl = s_gh + GG * 16;
    const int goff = ((l_cb * XT_GPL) + l_row * XT_TW + l_col) * 16 + l_byte;                 // + ry * XT_TW * 16

    __syncthreads();   // the zero fill
    const int ntile = tw.count;
    if (ntile > 0) {
        issue(0);
        commit();
    }
    __syncthreads();
    for (int k = 0; k < ntile; ++k) {
        if (k + 1 < ntile) issue(k + 1);
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int item = wave + 4 * i;             // (wave uniform)
            if (item >= NITEM) break;
            const int j = item / 3, v = item - 3 * j;
            const int xoff = (((2 * j + l_cb) * XT_XPL) + l_row * XT_XW + l_col + v) * 16 + l_byte;   // + (ry + u) * XT_XW * 16
            x3_bf16x8 xh_keep, xl_keep;   // the tap-row u = 2 fragments (tile rows ry + 2, ry + 3) are the u = 0 fragments of the next k-step
#pragma unroll
            for (int ry = 0; ry < XT_TH; ry += 2) {
                const x3_bf16x8 gh = ld_tr(s_gh + goff + ry * XT_TW * 16), gl = ld_tr(s_gl + goff + ry * XT_TW * 16);
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    x3_bf16x8 xh, xl;
                    if (u == 0 && ry > 0) { xh = xh_keep; xl = xl_keep; }
                    else { xh = ld_tr(s_xh + xoff + (ry + u) * XT_XW * 16); xl = ld_tr(s_xl + xoff + (ry + u) * XT_XW * 16); }
                    if (u == 2) { xh_keep = xh; xl_keep = xl; }
                    acc[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gl, xh, acc[i][u], 0, 0, 0);
                    acc[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh, xl, acc[i][u], 0, 0, 0);
                    acc[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh, xh, acc[i][u], 0, 0, 0);
                }
                if (item == 0) {   // bias gradient: row sums of the gradient fragments
                    accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gl, ones, accb, 0, 0, 0);
                    accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh, ones, accb, 0, 0, 0);
                }
            }
        }
        __syncthreads();   // every wave has read the tile
        if (k + 1 < ntile) commit();
        __syncthreads();
    }
    // block partial, register-major: [item][u][reg][lane]; C layout of the MFMA: lane l reg r = dW[oc = 4 (l >> 4) + r][ic = 16 j + (l & 15)]
    float* dst = partial + (long long)blockIdx.x * XT_PER;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int item = wave + 4 * i;
        if (item >= NITEM) break;
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[((item * 3 + u) * 4 + r) * 64 + lane] = acc[i][u][r];
    }
    if (wave == 0 && (lane & 15) == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[9 * 3 * 4 * 64 + 4 * (lane >> 4) + r] = accb[r];
    }
}

// The three DenseBlock convs of one encoder (16 -> 16, 32 -> 16, 48 -> 16: core/block.py:137-151) in ONE pass over [x0 | x1 | x2] and
// [g1 | g2 | g3] -- the fp32 form of csrc/enc_wgrad.hip: the same blocks and items as wgrad_x3_thin_kernel, item = (layer L, input block
// j < L, tap column v), 18 of them; every activation tile is staged once instead of up to three times (0.8 GB per encoder instead of 1.2).
constexpr int XD_PER = 18 * 768 + 64;                                  // floats per block partial: [item][u][reg][lane], db[3][16] (+ pad)
__global__ __launch_bounds__(XT_THREADS, 2) void wgrad_x3_dense_kernel(TV tx, TV tg, float* __restrict__ partial, int tiles_x, int tpi, int total) {
    constexpr int NXC = 6;
    constexpr bool SIGNS = false;
    unsigned* const signs = nullptr;
    constexpr int XG = NXC * XT_XPL, GG = 6 * XT_GPL;                  // x0 | x1 | x2 and g1 | g2 | g3 (16 channels each), one piece
    constexpr int N_X = NXC * XT_XH * XT_XW, N_ALL = N_X + 6 * XT_TH * XT_TW;
    constexpr int ROUNDS = (N_ALL + XT_THREADS - 1) / XT_THREADS;       // 8
    constexpr int NITEM = 18, MAXI = 5;                                 // items (layer L = 1..3, input block j < L, tap column v): 3 + 6 + 9
    __shared__ __attribute__((aligned(16))) x3_u4 s_buf[2 * XG + 2 * GG];   // [x hi][x lo][g hi][g lo]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const TileWalk tw = xcd_walk(total, gridDim.x, blockIdx.x);
    for (int i = tid; i < 2 * XG + 2 * GG; i += XT_THREADS) s_buf[i] = (x3_u4){0u, 0u, 0u, 0u};   // (the planes of absent channel blocks stay zero)
    unsigned geo[ROUNDS];   // is-g << 24 | channel block << 16 | tile row << 8 | tile column
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const int e = min(tid + XT_THREADS * k, N_ALL - 1);
        if (e < N_X) {
            const int cb = e / (XT_XH * XT_XW), rem = e - cb * (XT_XH * XT_XW), py = rem / XT_XW;
            geo[k] = (unsigned)(cb << 16 | py << 8 | (rem - py * XT_XW));
        } else {
            const int e2 = e - N_X;
            const int cb = e2 / (XT_TH * XT_TW), rem = e2 - cb * (XT_TH * XT_TW), py = rem / XT_TW;
            geo[k] = (unsigned)(1u << 24 | cb << 16 | py << 8 | (rem - py * XT_TW));
        }
    }
    X3Gran rin[ROUNDS];
    int c_in = 0, c_y0 = 0, c_x0 = 0;   // the tile in the staging registers (SIGNS: the ReLU sign map of x, as wgrad_x3_kernel writes it)
    auto issue = [&](int k_tile) {
        const int tile = tw.first + k_tile * tw.stride;
        const int in_ = tile / tpi, tt = tile - in_ * tpi;
        const int y0 = (tt / tiles_x) * XT_TH, x0 = (tt % tiles_x) * XT_TW;
        c_in = in_; c_y0 = y0; c_x0 = x0;
        const char* bx = tx.base + ((long long)in_ * tx.img + (long long)tx.cb_off * tx.plane) * 32;
        const char* bg = tg.base + ((long long)in_ * tg.img + (long long)tg.cb_off * tg.plane) * 32;
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int cb = (int)((geo[k] >> 16) & 255u), py = (int)((geo[k] >> 8) & 255u), px = (int)(geo[k] & 255u);
            rin[k].a = (x3_f4){0.f, 0.f, 0.f, 0.f};
            rin[k].b = rin[k].a;
            if (tid + XT_THREADS * k >= N_ALL) continue;
            if (!(geo[k] >> 24)) {
                if (cb < tx.cb) {
                    const int y = min(max(reflect_idx(y0 + py - 1, tx.h), 0), tx.h - 1);
                    const int x = min(max(reflect_idx(x0 + px - 1, tx.w), 0), tx.w - 1);
                    const x3_f4* p = reinterpret_cast<const x3_f4*>(bx + ((unsigned)cb * (unsigned)tx.plane + (unsigned)(y * tx.ws + x)) * 32u);
                    rin[k].a = p[0];
                    rin[k].b = p[1];
                }
            } else {
                const int y = y0 + py, x = x0 + px;
                if (cb < tg.cb && y < tg.h && x < tg.w) {
                    const x3_f4* p = reinterpret_cast<const x3_f4*>(bg + ((unsigned)cb * (unsigned)tg.plane + (unsigned)((y + tg.halo) * tg.ws + x + tg.halo)) * 32u);
                    rin[k].a = p[0];
                    rin[k].b = p[1];
                }
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            if (tid + XT_THREADS * k >= N_ALL) continue;
            const int cb = (int)((geo[k] >> 16) & 255u), py = (int)((geo[k] >> 8) & 255u), px = (int)(geo[k] & 255u);
            x3_u4 pc[2];
            x3_split_gran<2>(rin[k], pc);
            if (!(geo[k] >> 24)) {
                const int o = cb * XT_XPL + py * XT_XW + px;
                s_buf[o] = pc[0];
                s_buf[XG + o] = pc[1];
                if (SIGNS) {
                    const int y = c_y0 + py - 1, x = c_x0 + px - 1;
                    if (py >= 1 && py <= XT_TH && px >= 1 && px <= XT_TW && y < tx.h && x < tx.w && cb < tx.cb) {
                        const X3Gran& g = rin[k];
                        const unsigned bits = (g.a.x > 0.f) | (g.a.y > 0.f) << 1 | (g.a.z > 0.f) << 2 | (g.a.w > 0.f) << 3 | (g.b.x > 0.f) << 4 |
                                              (g.b.y > 0.f) << 5 | (g.b.z > 0.f) << 6 | (g.b.w > 0.f) << 7;
                        reinterpret_cast<unsigned char*>(signs)[(((unsigned)(c_in * ((tx.cb + 3) >> 2) + (cb >> 2)) * (unsigned)tx.h + (unsigned)y) * (unsigned)tx.w + (unsigned)x) * 4u + (unsigned)(cb & 3)] =
                            (unsigned char)bits;
                    }
                }
            } else {
                const int o = 2 * XG + cb * XT_GPL + py * XT_TW + px;
                s_buf[o] = pc[0];
                s_buf[GG + o] = pc[1];
            }
        }
    };

    typedef __attribute__((ext_vector_type(4))) float f32x4_;
    f32x4_ acc[MAXI][3], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
#pragma unroll
        for (int u = 0; u < 3; ++u) acc[i][u] = (f32x4_){0.f, 0.f, 0.f, 0.f};
    const x3_bf16x8 ones = __builtin_bit_cast(x3_bf16x8, ((x3_u4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}));
    // operand fragments of a 16 x 16 x 32 MFMA: lane l = channel l & 15, k-group l >> 4 = pixels 8 (l >> 4) .. + 7 of the k-step's 32
    // (two tile rows of 16); the transposing read (see wgrad_x3_kernel): lane sl of a 16-lane group addresses pixel sl >> 2 (+ 4 for the
    // second read), channel block (sl & 3) >> 1 of the 16-channel record, byte 8 (sl & 1), and receives channel sl of four pixels
    const int sl = lane & 15, g4 = lane >> 4;
    const int l_cb = (sl & 3) >> 1, l_byte = (sl & 1) * 8, l_row = g4 >> 1, l_col = 8 * (g4 & 1) + (sl >> 2);
    auto ld_tr = [&](const char* base) {
        const x3_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(X3_LDS_PTR(x3_s16x4, base));
        const x3_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(X3_LDS_PTR(x3_s16x4, base + 4 * 16));
        return __builtin_bit_cast(x3_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    const char* s_xh = reinterpret_cast<const char*>(s_buf);
    const char* s_xl = s_xh + XG * 16;
    const char* s_gh = s_xl + XG * 16;
    const char* s_gl = s_gh + GG * 16;

    __syncthreads();   // the zero fill
    const int ntile = tw.count;
    if (ntile > 0) {
        issue(0);
        commit();
    }
    __syncthreads();
    for (int k = 0; k < ntile; ++k) {
        if (k + 1 < ntile) issue(k + 1);
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int item = wave + 4 * i;             // (wave uniform)
            if (item >= NITEM) break;
            const int L = item < 3 ? 1 : (item < 9 ? 2 : 3), rel = item - (L == 1 ? 0 : (L == 2 ? 3 : 9));
            const int j = rel / 3, v = rel - 3 * j;
            const int xoff = (((2 * j + l_cb) * XT_XPL) + l_row * XT_XW + l_col + v) * 16 + l_byte;   // + (ry + u) * XT_XW * 16
            const int goff = (((2 * (L - 1) + l_cb) * XT_GPL) + l_row * XT_TW + l_col) * 16 + l_byte;  // + ry * XT_TW * 16
            x3_bf16x8 xh_keep, xl_keep;   // the tap-row u = 2 fragments (tile rows ry + 2, ry + 3) are the u = 0 fragments of the next k-step
#pragma unroll
            for (int ry = 0; ry < XT_TH; ry += 2) {
                const x3_bf16x8 gh = ld_tr(s_gh + goff + ry * XT_TW * 16), gl = ld_tr(s_gl + goff + ry * XT_TW * 16);
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    x3_bf16x8 xh, xl;
                    if (u == 0 && ry > 0) { xh = xh_keep; xl = xl_keep; }
                    else { xh = ld_tr(s_xh + xoff + (ry + u) * XT_XW * 16); xl = ld_tr(s_xl + xoff + (ry + u) * XT_XW * 16); }
                    if (u == 2) { xh_keep = xh; xl_keep = xl; }
                    acc[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gl, xh, acc[i][u], 0, 0, 0);
                    acc[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh, xl, acc[i][u], 0, 0, 0);
                    acc[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh, xh, acc[i][u], 0, 0, 0);
                }
                if (rel == 0) {   // (items 0, 3, 9 = waves 0, 3, 1: one per wave) bias gradient of layer L: row sums of the gradient fragments
                    accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gl, ones, accb, 0, 0, 0);
                    accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh, ones, accb, 0, 0, 0);
                }
            }
        }
        __syncthreads();   // every wave has read the tile
        if (k + 1 < ntile) commit();
        __syncthreads();
    }
    // block partial, register-major: [item][u][reg][lane]; C layout of the MFMA: lane l reg r = dW[oc = 4 (l >> 4) + r][ic = 16 j + (l & 15)]
    float* dst = partial + (long long)blockIdx.x * XD_PER;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int item = wave + 4 * i;
        if (item >= NITEM) break;
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[((item * 3 + u) * 4 + r) * 64 + lane] = acc[i][u][r];
    }
    if ((wave == 0 || wave == 3 || wave == 1) && (lane & 15) == 0) {
        const int L = wave == 0 ? 1 : (wave == 3 ? 2 : 3);
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[18 * 768 + 16 * (L - 1) + 4 * (lane >> 4) + r] = accb[r];
    }
}

__global__ __launch_bounds__(64 * RED_SLICES) void wgrad_x3_dense_reduce(const float* __restrict__ partial, float* __restrict__ dw1, float* __restrict__ db1,
                                                                        float* __restrict__ dw2, float* __restrict__ db2, float* __restrict__ dw3,
                                                                        float* __restrict__ db3, int G, int accumulate) {
    __shared__ float red[RED_SLICES][64];
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);      // the partial's order
    float* dst = nullptr;
    if (e < 18 * 768) {
        const int ln = e & 63, r = (e >> 6) & 3, u = (e >> 8) % 3, item = e / 768;
        const int L = item < 3 ? 1 : (item < 9 ? 2 : 3), rel = item - (L == 1 ? 0 : (L == 2 ? 3 : 9));
        const int j = rel / 3, v = rel - 3 * j;
        const int o = 4 * (ln >> 4) + r, c = 16 * j + (ln & 15);
        float* dw = L == 1 ? dw1 : (L == 2 ? dw2 : dw3);
        dst = dw + ((long long)o * (16 * L) + c) * 9 + u * 3 + v;
    } else if (e < 18 * 768 + 48) {
        const int L = (e - 18 * 768) / 16 + 1, o = (e - 18 * 768) % 16;
        float* db = L == 1 ? db1 : (L == 2 ? db2 : db3);
        if (db != nullptr) dst = db + o;
    }
    const float t = partial_sum(partial, e, XD_PER, G, dst != nullptr, red);
    if ((threadIdx.x >> 6) == 0 && dst != nullptr) *dst = accumulate ? *dst + t : t;
}

__global__ __launch_bounds__(64 * RED_SLICES) void wgrad_x3_thin_reduce(const float* __restrict__ partial, float* __restrict__ dw, float* __restrict__ db,
                                                                       int cin, int cout, int G, int accumulate) {
    __shared__ float red[RED_SLICES][64];
    // the partial's order: [item][u][reg][lane] for the layer's items (all blocks but the last), then db[16] behind all nine (the last block)
    const int e = (blockIdx.x + 1 < gridDim.x ? blockIdx.x * 64 : 9 * 3 * 4 * 64) + (threadIdx.x & 63);
    long long dst = -1;
    if (e < 9 * 3 * 4 * 64) {
        const int ln = e & 63, r = (e >> 6) & 3, u = (e >> 8) % 3, item = e / 768;
        const int j = item / 3, v = item - 3 * j;
        const int o = 4 * (ln >> 4) + r, c = 16 * j + (ln & 15);
        if (o < cout && c < cin) dst = ((long long)o * cin + c) * 9 + u * 3 + v;
    } else if (e < 9 * 3 * 4 * 64 + 16) {
        const int o = e - 9 * 3 * 4 * 64;
        if (o < cout) dst = -2 - o;
    }
    const float t = partial_sum(partial, e, XT_PER, G, dst != -1, red);
    if ((threadIdx.x >> 6) == 0 && dst != -1) {
        if (dst >= 0) dw[dst] = accumulate ? dw[dst] + t : t;
        else if (db != nullptr) { const int o = (int)(-2 - dst); db[o] = accumulate ? db[o] + t : t; }
    }
}

// dw / db = fixed-order sum of the G block partials of each (icg, ocg) pair
template <int SL>
__global__ __launch_bounds__(64 * SL) void wgrad_x3_reduce(const float* __restrict__ partial, float* __restrict__ dw, float* __restrict__ db,
                                                       int cin, int cout, int G, int n_icg, int n_ocg, int accumulate, int taps) {
    // threads walk the PARTIAL's order (register-major tiles: 64 consecutive threads read 64 consecutive floats of every block's partial)
    // and scatter the one result each to dW's order; walking dW's order instead read 64 different cache lines per load instruction
    __shared__ float red[SL][64];
    const int XW_PER = 64 * 64 * taps + 64;
    const int npairs = n_icg * n_ocg;
    const long long idx = (long long)blockIdx.x * 64 + (threadIdx.x & 63);      // over npairs x XW_PER
    const int pair = (int)(idx / XW_PER), e = (int)(idx - (long long)pair * XW_PER);
    const int icg = pair % n_icg, ocg = pair / n_icg;
    long long dst = -1;           // index into dw (>= 0), or -2 - o for db[o]
    if (pair < npairs) {
        if (e < 64 * 64 * taps) {
            const int ln = e & 63, r = (e >> 6) & 15, tile = e >> 10;
            int mt, jt, tap;
            if (taps == 1) { mt = tile >> 1; jt = tile & 1; tap = 0; }
            else { const int u = tile % 3, v = (tile / 3) % 3, mj = tile / 9; mt = mj >> 1; jt = mj & 1; tap = u * 3 + v; }
            const int o = ocg * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), c = icg * 64 + jt * 32 + (ln & 31);
            if (o < cout && c < cin) dst = ((long long)o * cin + c) * taps + tap;
        } else if (icg == 0) {
            const int o = ocg * 64 + (e - 64 * 64 * taps);
            if (o < cout) dst = -2 - o;
        }
    }
    const float t = partial_sum<SL>(partial, idx, (long long)npairs * XW_PER, G, dst != -1, red);
    if ((threadIdx.x >> 6) == 0 && dst != -1) {
        if (dst >= 0) dw[dst] = accumulate ? dw[dst] + t : t;
        else if (db != nullptr) { const int o = (int)(-2 - dst); db[o] = accumulate ? db[o] + t : t; }
    }
}

// ------------------------------------------------------------------ host side
static int x3_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
        if (const char* e = getenv("MMIF_NUM_CUS")) n = atoi(e) > 0 ? atoi(e) : n;   // (experiments: persistent grids on part of the chip)
    }
    return n;
}

static bool x3_enabled() {   // $MMIF_X3=0: fp32 tensors stay on the fp32 FMA kernels (A/B timing, cross-checks)
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("MMIF_X3");
        on = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    return on == 1;
}

static bool x3_grad_ok(const TV& t) { return t.halo == 0 || (t.halo == 1 && t.folded); }
static bool x3_small(const TV& t) { return t.plane * 32 * 8 < (1ll << 32); }   // 8 planes addressed with 32-bit byte offsets

bool conv_x3_supported(bool dgrad, int ks, int cin, int cout, const TV& tin, const TV& tout) {
    if (!x3_enabled() || (ks != 3 && ks != 1) || cin < 1 || cout < 1) return false;
    if (dgrad && !x3_grad_ok(tin)) return false;
    if ((dgrad ? cin : cout) > 64 * 8) return false;   // mask / accum bits address 64 channel blocks
    return (ks == 1 || (tin.h >= 2 && tin.w >= 2)) && x3_small(tin) && x3_small(tout);
}

// operand format of the FORWARD pass: 16 (default) = two SCALED FP16 pieces, 3 products, fp32-grade (2^-23 per product); 3 = three bf16
// pieces, 6 products, fp32-grade, no range limits (activations beyond +-4.1e3 or weights beyond +-64 saturate in the fp16 form); 2 = two
// bf16 pieces, 3 products, activations within ~1e-5 (ReLU decisions differ from the reference's ~10x as often as between two fp32
// implementations).  Process wide; set it before packing (mmif_set_x3_forward_pieces, $MMIF_X3_FWD_PIECES = 16 | 3 | 2).  The
// backward kernels are linear in the gradient and always use two bf16 pieces.
static int g_fwd_pieces = 0;
static int x3_norm_mode(int m) { return (m == 2 || m == 3) ? m : 16; }
int x3_fwd_pieces() {
    if (g_fwd_pieces == 0) {
        const char* e = getenv("MMIF_X3_FWD_PIECES");
        g_fwd_pieces = x3_norm_mode(e != nullptr ? atoi(e) : 16);
    }
    return g_fwd_pieces;
}

// The format of every forward operand image this process has packed, by device address: the forward launches dispatch on the IMAGE's
// format, not on the process-wide mode -- a mode change between a pack and a later forward (mmif_set_x3_forward_pieces without a re-pack;
// ADVICE r3) can no longer make a kernel read fp16 pieces as bf16 pieces.  Host-side, in call order = stream order; a replayed hipGraph
// carries its own pack + forward launches with the format they were captured in.  An address never packed here falls back to the mode.
static std::mutex g_fmt_mu;
static std::unordered_map<const void*, int> g_fwd_fmt;
static void x3_note_format(const void* img, int pieces) {
    std::lock_guard<std::mutex> lk(g_fmt_mu);
    g_fwd_fmt[img] = pieces;
}
static int x3_image_format(const void* img) {
    std::lock_guard<std::mutex> lk(g_fmt_mu);
    auto it = g_fwd_fmt.find(img);
    return it != g_fwd_fmt.end() ? it->second : x3_fwd_pieces();
}

size_t conv_x3_packed_bytes(int cout, int cin, int ks) {
    if (ks != 3 && ks != 1) return 16;
    const size_t a = x3_packed_bytes(cout, cin, 3, ks), b = x3_packed_bytes(cin, cout, 2, ks);   // (room for either forward format)
    return a > b ? a : b;
}

int conv_x3_pack_multi(const mmif_pack_job* jobs, int n_jobs, hipStream_t st) {
    X3PackTable tab;
    int n = 0;
    auto flush = [&]() -> int {
        if (n == 0) return MMIF_OK;
        hipLaunchKernelGGL(x3_pack_kernel, dim3(64, n), dim3(256), 0, st, tab);
        n = 0;
        return check_launch("pack_weights_x3");
    };
    for (int i = 0; i < n_jobs; ++i) {
        const mmif_pack_job& jb = jobs[i];
        if (jb.format != MMIF_PACK_X3 || (jb.ksize != 3 && jb.ksize != 1)) continue;
        for (int d = 0; d < 2; ++d) {
            void* dst = d ? jb.packed_dgrad : jb.packed_fwd;
            if (dst == nullptr) continue;
            const int n_out = d ? jb.cin : jb.cout, n_in = d ? jb.cout : jb.cin;
            X3PackImage& im = tab.im[n++];
            im.w = jb.w; im.dst = (bf16_t*)dst; im.cout = jb.cout; im.cin = jb.cin; im.dgrad = d; im.chain_k = -1; im.w2 = im.w3 = nullptr;
            im.mbw = 32 * x3_mb(n_out); im.nch = x3_nch(n_in); im.ks = jb.ksize;
            im.f16 = (!d && x3_fwd_pieces() == 16) ? 1 : 0;
            im.pieces = (d || im.f16) ? 2 : x3_fwd_pieces();
            if (!d) x3_note_format(dst, x3_fwd_pieces());
            im.total = (long long)x3_nmb(n_out) * im.nch * jb.ksize * jb.ksize * 2 * im.mbw * 8;   // elements of the first piece's images
            if (n == X3_PACK_MAX)
                if (int rc = flush()) return rc;
        }
    }
    return flush();
}

template <int MB, int NP, int NW, int RJ, int KS = 3, bool F16 = false, bool M16 = false>
static int launch_conv_x3(bool dgrad, const TV& tin, const TV& tout, const TV& tmask, const void* wpk, const float* bias, int n_out, int n_in,
                          int relu, uint64_t mask_bits, uint64_t accum_bits, hipStream_t st, const unsigned* signs, bool* folded = nullptr) {
    // dgrad, 3x3, the 32 x 32 x 16 tiles, a caller that wants the fold (folded != nullptr) and an output whose fold targets are distinct:
    // interior tiles + in-tile fold steps (org = 1), the halo ring stays untouched (zero: the caller's contract for a fold-me call)
    const int org = (dgrad && KS == 3 && !M16 && RJ <= 2 && folded != nullptr && tout.halo == 1 && tout.h >= 4 && tout.w >= 4) ? 1 : 0;
    if (org) *folded = true;
    const int tiles_x = cdiv(tout.ws - 2 * org, X3_TW), tiles_y = cdiv(tout.hs - 2 * org, RJ * NW);
    const int tpi = tiles_x * tiles_y, total = tpi * tout.n;
    constexpr int nw4_blocks = 3;   // four-wave blocks per CU (persistent grid): 42 KB of LDS, < 168 VGPRs each (two: half the gain; four: over-subscribed)
    int G = x3_num_cus() * (NW == 4 ? nw4_blocks : 1);
    if (total < G) G = total;
    const int nch = x3_nch(n_in), nmb = x3_nmb(n_out);
    constexpr int X3_THREADS = 64 * NW;
    static int abl = -1;
    if (abl < 0) abl = ablate_env("x3");
    relu = (relu & 255) | (abl << 8);
    if (dgrad)
        hipLaunchKernelGGL((conv_x3_kernel<MB, true, NP, NW, RJ, KS, false, M16>), dim3(G), dim3(X3_THREADS), 0, st, tin, tout, tmask, (const uint4*)wpk, bias, n_out, nch, nmb,
                           relu, (unsigned long long)mask_bits, (unsigned long long)accum_bits, tiles_x, tpi, total, signs, org);
    else
        hipLaunchKernelGGL((conv_x3_kernel<MB, false, NP, NW, RJ, KS, F16, M16>), dim3(G), dim3(X3_THREADS), 0, st, tin, tout, tmask, (const uint4*)wpk, bias, n_out, nch, nmb,
                           relu, (unsigned long long)mask_bits, (unsigned long long)accum_bits, tiles_x, tpi, total, (const unsigned*)nullptr, 0);
    return check_launch(dgrad ? "conv_x3 dgrad" : "conv_x3 fwd");
}

// forward: tin = x, tout = y;  dgrad: tin = gy (halo 0 or folded halo 1), tout = gx (the padded domain is written; the caller folds)
// folded (dgrad): non-NULL = the caller wants fold_halo(gx) applied and guarantees a zero halo ring on entry; *folded says whether the kernel
// chosen did it (interior tiles + fold steps) -- otherwise the caller runs the fold kernel
int conv_x3(bool dgrad, const TV& tin, const TV& tout, const TV& tmask, const void* wpk, const float* bias, int cin, int cout, int relu,
            uint64_t mask_bits, uint64_t accum_bits, hipStream_t st, int ks, const unsigned* signs, bool* folded) {
    if (folded != nullptr) *folded = false;
    const int n_out = dgrad ? cin : cout, n_in = dgrad ? cout : cin;
    const int fmt = dgrad ? 2 : x3_image_format(wpk);
    const bool six = !dgrad && fmt == 3, h16 = !dgrad && fmt == 16;
#define X3_ARGS tin, tout, tmask, wpk, bias, n_out, n_in, relu, mask_bits, accum_bits, st, signs, folded
    if (ks == 1) {
        if (x3_mb(n_out) == 2) {
            if (h16) return launch_conv_x3<2, 2, 8, 2, 1, true>(false, X3_ARGS);
            return six ? launch_conv_x3<2, 3, 8, 2, 1>(false, X3_ARGS) : launch_conv_x3<2, 2, 8, 2, 1>(dgrad, X3_ARGS);
        }
        if (h16) return launch_conv_x3<1, 2, 8, 2, 1, true>(false, X3_ARGS);
        return six ? launch_conv_x3<1, 3, 8, 2, 1>(false, X3_ARGS) : launch_conv_x3<1, 2, 8, 2, 1>(dgrad, X3_ARGS);
    }
    if (x3_mb(n_out) == 2) {
        if (h16) return launch_conv_x3<2, 2, 8, 2, 3, true>(false, X3_ARGS);
        return six ? launch_conv_x3<2, 3, 8, 2>(false, X3_ARGS) : launch_conv_x3<2, 2, 8, 2>(dgrad, X3_ARGS);
    }
    // <= 32 output channels: one 32-channel accumulator tile per pixel row, so a wave takes FOUR rows (32 x 32 pixel tiles) in the
    // 3-piece forward: the same 216 MFMAs per wave and barrier as the 64-channel kernels for half the weight staging
    constexpr int rj4 = 1;
    // thin layers (<= 32 out, <= 64 in: the DenseBlock convs and the decoder's tail) have almost no MFMA work per tile and run at the
    // latency of ONE tile's loads per block: four-wave blocks (8-row tiles, 16-channel chunks, 42 KB of LDS, 146 VGPRs) put THREE
    // independent blocks on a CU: forward 16->16 102 -> 77 us, 32->16 161 -> 144, 48->16 258 -> 212, 64->32 348 -> 324; dgrad 16->16
    // 160 -> 123, 32->16 204 -> 175 (two blocks: half of that; four: over-subscribed, slower; the 64-out-channel kernels on four-wave
    // blocks: +-0)
    if (n_in <= 64 && !six) {
        if (n_out <= 16) {      // <= 16 output channels: 16 x 16 x 32 MFMAs, K = two taps x 16 channels (no padded half tile)
            if (h16) return launch_conv_x3<1, 2, 4, 2, 3, true, true>(false, X3_ARGS);
            return launch_conv_x3<1, 2, 4, 2, 3, false, true>(dgrad, X3_ARGS);
        }
        if (h16) return launch_conv_x3<1, 2, 4, 2, 3, true>(false, X3_ARGS);
        return launch_conv_x3<1, 2, 4, 2>(dgrad, X3_ARGS);
    }
    if (h16) return launch_conv_x3<1, 2, 8, 2, 3, true>(false, X3_ARGS);
    if (six) return rj4 ? launch_conv_x3<1, 3, 8, 4>(false, X3_ARGS) : launch_conv_x3<1, 3, 8, 2>(false, X3_ARGS);
    return launch_conv_x3<1, 2, 8, 2>(dgrad, X3_ARGS);
#undef X3_ARGS
}

bool wgrad_x3_supported(int ks, int cin, int cout, const TV& tx, const TV& tg) {
    return x3_enabled() && (ks == 3 || ks == 1) && cin >= 1 && cout >= 1 && x3_grad_ok(tg) && tx.halo == 0 && (ks == 1 || (tx.h >= 2 && tx.w >= 2)) && x3_small(tx) && x3_small(tg);
}

static int wgrad_x3_G(int cin, int cout) {
    const int npairs = cdiv(cin, 64) * cdiv(cout, 64);
    int G = x3_num_cus() / npairs;
    if (G >= 8) G &= ~7;
    return G < 1 ? 1 : G;
}

size_t wgrad_x3_workspace(int cin, int cout, int ks) {
    if (ks == 1) {   // three k-split partials per block
        const size_t npairs = (size_t)cdiv(cin, 64) * cdiv(cout, 64);
        size_t G = 256 / npairs;
        if (G < 1) G = 1;
        const size_t dyn = (size_t)wgrad_x3_G(cin, cout);
        if (dyn > G) G = dyn;
        return 3 * G * npairs * (64 * 64 + 64) * sizeof(float);
    }
    if (ks != 3) return 0;
    const size_t npairs = (size_t)cdiv(cin, 64) * cdiv(cout, 64);
    size_t G = 256 / npairs;   // upper bound of wgrad_x3_G on any gfx950 part (<= 256 CUs) without asking the device
    if (G < 1) G = 1;
    const size_t dyn = (size_t)wgrad_x3_G(cin, cout);
    if (dyn > G) G = dyn;
    return G * npairs * (64 * 64 * 9 + 64) * sizeof(float);
}

size_t x3_signs_bytes(int n, int cb, int h, int w) { return (size_t)n * ((cb + 3) / 4) * h * w * 4; }

int wgrad_x3(const TV& tx, const TV& tg, float* dw, float* db, int cin, int cout, int accumulate, float* ws, hipStream_t st, int ks, unsigned* signs) {
    if (ks == 1) {
        const int tiles_x = cdiv(tx.w, XW_TW), tiles_y = cdiv(tx.h, 8);
        const int tpi = tiles_x * tiles_y, total = tpi * tx.n;
        const int n_icg = cdiv(cin, 64), n_ocg = cdiv(cout, 64);
        int G = wgrad_x3_G(cin, cout);
        if (total < G) G = total;
        if (signs != nullptr) hipLaunchKernelGGL((wgrad_x3_kernel<8, 8, 8, 1, true>), dim3(G * n_icg * n_ocg), dim3(XW_THREADS), 0, st, tx, tg, ws, cin, cout, tiles_x, tpi, total, G, n_icg, n_ocg, signs);
        else hipLaunchKernelGGL((wgrad_x3_kernel<8, 8, 8, 1>), dim3(G * n_icg * n_ocg), dim3(XW_THREADS), 0, st, tx, tg, ws, cin, cout, tiles_x, tpi, total, G, n_icg, n_ocg, signs);
        if (int rc = check_launch("wgrad_x3 1x1")) return rc;
        const int n = n_icg * n_ocg * (64 * 64 + 64);   // the reduce walks the partial's order
        const int RG = 3 * G;
        if (RG > 64) hipLaunchKernelGGL(wgrad_x3_reduce<16>, dim3(cdiv(n, 64)), dim3(1024), 0, st, ws, dw, db, cin, cout, 3 * G, n_icg, n_ocg, accumulate, 1);
    else hipLaunchKernelGGL(wgrad_x3_reduce<4>, dim3(cdiv(n, 64)), dim3(256), 0, st, ws, dw, db, cin, cout, 3 * G, n_icg, n_ocg, accumulate, 1);
        return check_launch("wgrad_x3_reduce");
    }
    const bool thin = cin <= 48 && cout <= 16;
    if (thin) {
        const int tiles_x = cdiv(tx.w, XT_TW), tiles_y = cdiv(tx.h, XT_TH);
        const int tpi = tiles_x * tiles_y, total = tpi * tx.n;
        int G = x3_num_cus() * (cin <= 16 ? 5 : (cin <= 32 ? 4 : 3));   // blocks per CU: what the registers (92 / 128 / 168) and the LDS (20 / 32 / 44 KB) allow
        if (total < G) G = total;
        if ((size_t)G * XT_PER * sizeof(float) <= wgrad_x3_workspace(cin, cout, 3)) {
#define XT_LAUNCH(...) hipLaunchKernelGGL((wgrad_x3_thin_kernel<__VA_ARGS__>), dim3(G), dim3(XT_THREADS), 0, st, tx, tg, ws, tiles_x, tpi, total, signs)
            if (cin <= 16) { if (signs != nullptr) XT_LAUNCH(2, true); else XT_LAUNCH(2); }
            else if (cin <= 32) { if (signs != nullptr) XT_LAUNCH(4, true); else XT_LAUNCH(4); }
            else { if (signs != nullptr) XT_LAUNCH(6, true); else XT_LAUNCH(6); }
#undef XT_LAUNCH
            if (int rc = check_launch("wgrad_x3 thin")) return rc;
            const int n = (cin <= 16 ? 3 : (cin <= 32 ? 6 : 9)) * 768;   // the items this layer has (partial order; the bias sums sit behind all nine)
            hipLaunchKernelGGL(wgrad_x3_thin_reduce, dim3(n / 64 + 1), dim3(64 * RED_SLICES), 0, st, ws, dw, db, cin, cout, G, accumulate);
            return check_launch("wgrad_x3_thin_reduce");
        }
    }
    const int th = thin ? 16 : 8;
    const int tiles_x = cdiv(tx.w, XW_TW), tiles_y = cdiv(tx.h, th);
    const int tpi = tiles_x * tiles_y, total = tpi * tx.n;
    const int n_icg = cdiv(cin, 64), n_ocg = cdiv(cout, 64);
    int G = wgrad_x3_G(cin, cout);
    if (total < G) G = total;
    static int abl = -1;
    if (abl < 0) abl = ablate_env("x3");
    const int tx_abl = tiles_x | (abl << 16);
#define XW_LAUNCH(...) hipLaunchKernelGGL((wgrad_x3_kernel<__VA_ARGS__>), dim3(G * n_icg * n_ocg), dim3(XW_THREADS), 0, st, tx, tg, ws, cin, cout, tx_abl, tpi, total, G, n_icg, n_ocg, signs)
    if (thin) { if (signs != nullptr) XW_LAUNCH(16, 6, 2, 3, true); else XW_LAUNCH(16, 6, 2, 3); }
    else { if (signs != nullptr) XW_LAUNCH(8, 8, 8, 3, true); else XW_LAUNCH(8, 8, 8, 3); }
#undef XW_LAUNCH
    if (int rc = check_launch("wgrad_x3")) return rc;
    const int n = n_icg * n_ocg * (64 * 64 * 9 + 64);   // the reduce walks the partial's order
    const int RG = G;
    if (RG > 64) hipLaunchKernelGGL(wgrad_x3_reduce<16>, dim3(cdiv(n, 64)), dim3(1024), 0, st, ws, dw, db, cin, cout, G, n_icg, n_ocg, accumulate, 9);
    else hipLaunchKernelGGL(wgrad_x3_reduce<4>, dim3(cdiv(n, 64)), dim3(256), 0, st, ws, dw, db, cin, cout, G, n_icg, n_ocg, accumulate, 9);
    return check_launch("wgrad_x3_reduce");
}


// dW / db of the three DenseBlock convs of one encoder: tx = [x0 | x1 | x2] (>= 6 channel blocks), tg = [g1 | g2 | g3] (6 blocks)
bool wgrad_x3_dense_supported(const TV& tx, const TV& tg) {
    return x3_enabled() && x3_grad_ok(tg) && tx.halo == 0 && tx.h >= 2 && tx.w >= 2 && x3_small(tx) && x3_small(tg) && tx.cb >= 6 && tg.cb == 6;
}
size_t wgrad_x3_dense_workspace() { return (size_t)2 * 256 * XD_PER * sizeof(float); }
int wgrad_x3_dense(const TV& tx, const TV& tg, float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, int accumulate, float* ws,
                   hipStream_t st) {
    const int tiles_x = cdiv(tx.w, XT_TW), tiles_y = cdiv(tx.h, XT_TH);
    const int tpi = tiles_x * tiles_y, total = tpi * tx.n;
    int G = x3_num_cus() * 2;
    if (G > 512) G = 512;
    if (total < G) G = total;
    hipLaunchKernelGGL(wgrad_x3_dense_kernel, dim3(G), dim3(XT_THREADS), 0, st, tx, tg, ws, tiles_x, tpi, total);
    if (int rc = check_launch("wgrad_x3 dense")) return rc;
    hipLaunchKernelGGL(wgrad_x3_dense_reduce, dim3((18 * 768 + 64) / 64), dim3(64 * RED_SLICES), 0, st, ws, dw1, db1, dw2, db2, dw3, db3, G, accumulate);
    return check_launch("wgrad_x3_dense_reduce");
}
}  // namespace mmif

using namespace mmif;

extern "C" void mmif_set_x3_forward_pieces(int32_t pieces) { g_fwd_pieces = x3_norm_mode(pieces); }
extern "C" int32_t mmif_get_x3_forward_pieces(void) { return x3_fwd_pieces(); }
// 1 when fp32 tensors take the split-operand kernels ($MMIF_X3, read ONCE per process by the library): callers that plan their launches
// (the Python engine) ask here instead of re-reading the environment, so the two can never disagree
// number of weight values the scaled-fp16 forward images clamped (|w| >= 65000 / 2^10) since the last reset; SYNCHRONISES the device -- call it
// after loading / initialising weights, not per step.  A non-zero count means the fp32 forward is wrong for those weights: switch the layer
// to mmif_set_x3_forward_pieces(3) (bf16 pieces carry fp32's exponent range).
extern "C" int32_t mmif_x3_pack_saturations(int32_t reset) {
    unsigned v = 0;
    // the pack kernels run on the caller's (possibly non-blocking) streams, which a symbol copy on the null stream does not order against:
    // drain the device first.  Never legal under stream capture -- the engine does not call this while capturing.
    if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_x3_sat_count), sizeof(v)) != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (reset) {
        const unsigned z = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_x3_sat_count), &z, sizeof(z));
    }
    return (int32_t)(v > 0x7fffffffu ? 0x7fffffff : v);
}
extern "C" int32_t mmif_get_x3_enabled(void) { return x3_enabled() ? 1 : 0; }

extern "C" size_t mmif_packed_weight_bytes_x3(int32_t cout, int32_t cin, int32_t ksize) { return conv_x3_packed_bytes(cout, cin, ksize); }

extern "C" int mmif_pack_weights_x3(const float* w, int32_t cout, int32_t cin, int32_t ksize, void* packed_fwd, void* packed_dgrad, void* stream) {
    MMIF_REQUIRE(ksize == 3 || ksize == 1, "pack_weights_x3: ksize must be 1 or 3");
    MMIF_REQUIRE(w != nullptr && cout > 0 && cin > 0, "pack_weights_x3: bad arguments");
    mmif_pack_job jb;
    jb.w = w; jb.cout = cout; jb.cin = cin; jb.ksize = ksize; jb.format = MMIF_PACK_X3; jb.packed_fwd = packed_fwd; jb.packed_dgrad = packed_dgrad;
    return conv_x3_pack_multi(&jb, 1, (hipStream_t)stream);
}

// dgrad operand images (x3 format) of the DenseBlock's three virtual gather layers: layer k has 16 input channels (x_k) and 16 (3 - k)
// output channels (convs k+1 .. 3 stacked, each restricted to its x_k input slice) -- mmif_pack_dense_chain for fp32 tensors.
extern "C" int mmif_pack_dense_chain_x3(const float* w1, const float* w2, const float* w3, void* packed_v0, void* packed_v1, void* packed_v2,
                                        void* stream) {
    MMIF_REQUIRE(w1 != nullptr && w2 != nullptr && w3 != nullptr && packed_v0 != nullptr && packed_v1 != nullptr && packed_v2 != nullptr,
                 "pack_dense_chain_x3: NULL argument");
    X3PackTable tab;
    void* dst[3] = {packed_v0, packed_v1, packed_v2};
    for (int k = 0; k < 3; ++k) {
        X3PackImage& im = tab.im[k];
        const int n_out = 16, n_in = 16 * (3 - k);      // the dgrad kernel's view: out = x_k's channels, in = the stacked gradient channels
        im.w = w1; im.w2 = w2; im.w3 = w3; im.chain_k = k;
        im.dst = (bf16_t*)dst[k]; im.cout = n_in; im.cin = n_out; im.dgrad = 1;
        im.mbw = 32 * x3_mb(n_out); im.nch = x3_nch(n_in); im.ks = 3; im.f16 = 0; im.pieces = 2;
        im.total = (long long)x3_nmb(n_out) * im.nch * 9 * 2 * im.mbw * 8;
    }
    hipLaunchKernelGGL(x3_pack_kernel, dim3(64, 3), dim3(256), 0, (hipStream_t)stream, tab);
    return check_launch("pack_dense_chain_x3");
}
