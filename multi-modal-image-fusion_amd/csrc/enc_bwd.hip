// Backward of the DenseBlock encoder -- ConvLayer(1 -> 16) + DenseBlock(16, 16) of the PFNet / DenseFuse family (reference
// core/model.py:73-80, core/block.py:137-151 under autograd, train.py:71) -- as ONE streaming kernel (round 5): the gradient chain
//     g2 = [x2 > 0] (G2 + A32 g3)      g1 = [x1 > 0] (G1 + A21 g2 + A31 g3)      g0 = [x0 > 0] (G0 + A10 g1 + A20 g2 + A30 g3)
// (A_lk = adjoint of reflect pad + 3x3 correlation; csrc/enc_chain.hip, round 4) AND the weight gradients dW, db of all four layers
// (csrc/enc_wgrad.hip, round 2) from the same rows while they are in LDS.
//
// The two-kernel form moves 288 channel planes per branch through HBM: the chain reads G (64) + the masks x0..x2 (48) and WRITES
// [g0 | g1 | g2 | g3] (64), the weight-gradient pass reads them back (64) with x0..x2 (48) again.  The gradients g0..g2 have no other
// consumer.  Here they never leave the chip.  A wave needs 216 weight-gradient accumulator registers next to the chain's 72 + operands:
// too much for one wave -- so the work of a 32-column strip is split over a PAIR of waves on the same SIMD (warp specialisation; a block is
// four pairs: wave p = role A, wave p + 4 = role B, 256 registers each):
//   role A  walks down the strip's rows; per row step r it takes g3 row r+4 (registers -> LDS ring) and runs the chain in INPUT-STATIONARY
//           form (csrc/enc_stream2.hip): input row R of a layer feeds the three tap rows at once -- out[R+1] (fresh accumulator, its C operand
//           = the incoming gradient G of that row), out[R], out[R-1] (complete) -- layer 1 on g3 row r+3 -> g2 row r+2, layer 2 on [g2 | g3]
//           row r+2 -> g1 row r+1, layer 3 ONE STEP BEHIND on [g1 | g2 | g3] row r -> g0 row r-1 (its operands are a step old: its k-steps
//           fill the latencies of the dependent pair); epilogue = tile pairing by v_permlane16_swap, column fold of the padded-domain halo
//           (DPP row shift + FMA in edge strips), ReLU mask from x, one rounding; the rows stay in LDS rings (4 / 4 / 2 / 2 slots).  It also
//           forms dW1 (g1 row r against x0 rows r-1 .. r+1: nine MFMAs per step).
//   role B  forms dW3, dW2, the bias sums and the first layer's gradients: dW_L[o][c][u][v] += sum_px gL[o](r, px) x_in[c](R(r+u-1), px+v-1),
//           K = the strip's 32 pixels = ONE k-step, both operands k-major through transposing LDS reads (ds_read_b64_tr_b16); the first layer
//           against the fp32 image as three exact bf16 pieces (round 6; the fp32 matrix path before), two rows behind (g0 row r-2 left the
//           chain in step r-1).  It also issues
//           the LDS-DMA of the activation / image rows (ring of 6 / 8 rows) and waits for them: it has the slack.
// One s_barrier per row step couples the roles: everything a role reads was written at least one barrier earlier (table in DESIGN.md 4.1).
// One block per CU: the branch's (image, strip) columns form one line of rows cut into slices of equal weighted length (fb_geometry); a pair walks its slice in pieces, the
// weight gradients accumulate in registers across pieces; one partial sum per block, finished by enc_wgrad_reduce in fixed order.
// HBM traffic: G (64 planes) + x (48) + the image, NOTHING written but the block partials.  Pixels are counted once: the weight-gradient
// operand of gL is zeroed outside the strip's kept columns, rows outside the piece are skipped; strips / pieces recompute their margins
// (3 columns / rows).
#include "enc_wgrad.hpp"
#include <stdlib.h>
#include <mutex>
#pragma clang diagnostic ignored "-Winline-asm"   // (the LDS-DMA asm names m0 in its clobber list: "reserved register")

namespace mmif {

typedef __attribute__((ext_vector_type(8))) __bf16 fb_bf16x8;
typedef __attribute__((ext_vector_type(4))) float fb_f32x4;
typedef __attribute__((ext_vector_type(4))) short fb_s16x4;
typedef __attribute__((ext_vector_type(2))) short fb_s16x2;
typedef __attribute__((ext_vector_type(2))) unsigned short fb_u16x2;
typedef __attribute__((ext_vector_type(8))) short fb_s16x8;
typedef unsigned fb_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned fb_u32x2 __attribute__((ext_vector_type(2)));
#define FB_LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

constexpr int FB_W = 32, FB_KEEP = FB_W - 6;
// activation ring rows are PIXEL-major inside a layer's 16 channels: [x0 | x1 | x2][32 px][2 channel blocks][16 B] -- the 16 lanes of a
// transposing read (4 pixels x 16 channels) touch 128 contiguous bytes (no bank conflict); the DMA lane 2 px + cb fetches that granule
// gradient ring rows carry a zero guard granule on either side of their 32 pixels: the chain is a ZERO-padded correlation of the gradient
// rows (the adjoint of reflect padding is applied to its padded-domain result), so the operand reads of pixel -1 / 32 -- one granule
// before / after the row -- must be zero where the strip holds the image's padded column (enc_stream2.hip reads its neighbours' bytes
// there: its border pixels are overwritten by the ghost copy; here they FOLD onto columns 1 / w-2)
constexpr int FB_CBS = 544;                    // bytes of one channel block of a gradient ring row: [guard][32 px][guard] x 16 B
constexpr int FB_ROW = 2 * FB_CBS;             // one gradient ring slot: [cb 0][cb 1]
constexpr int FB_S3 = 4, FB_S2 = 4, FB_S1 = 2, FB_S0 = 2;   // ring slots of g3 / g2 / g1 / g0 (powers of two)
constexpr int FB_G3 = 0, FB_G2 = FB_S3 * FB_ROW, FB_G1 = FB_G2 + FB_S2 * FB_ROW, FB_G0 = FB_G1 + FB_S1 * FB_ROW, FB_GRING = FB_G0 + FB_S0 * FB_ROW;   // 13056
constexpr int FB_XS = 6;                       // slots of the activation ring (rows r-1 .. r+3 in use, r+4 in flight)
constexpr int FB_IS = 8;                       // slots of the image ring (rows r-3 .. r+3 in use: the first layer's products trail by two rows)
constexpr int FB_XROW = 3 * 1024;              // x0 | x1 | x2 row
constexpr int FB_IROW = 128;                   // image row: 32 fp32
constexpr int FB_PAIRS = 4;                    // wave pairs per block: waves 0..3 run the chain (role A), waves 4..7 the weight gradients (B);
constexpr int FB_WAVES = 2 * FB_PAIRS;         // wave p and wave p + 4 share SIMD p and one strip
constexpr int FB_NFRAG = 30;                   // chain A fragments in LDS: (2 + 3 + 5 k-steps) x 3 tap rows, 1 KB each
constexpr int FB_WBYTES = FB_NFRAG * 1024;
constexpr int FB_LDS = FB_WBYTES + FB_PAIRS * FB_GRING + 64;
// (64 zero bytes after the activation rings and after the image rings: the operand reads of pixel 32 run one granule past a row, into the
//  next row / slot / wave -- finite data, multiplied by a zeroed gradient -- and after the LAST row they must not find the fp32 image
//  ring, whose low halves read as bf16 are arbitrary bit patterns, NaN included)
constexpr int FB_L0C = FB_PAIRS * (FB_XS * FB_XROW + FB_IS * FB_IROW) + 192;
// (+ the constant columns of the first layer's B operand, read like an image row: 128 B of 1.0f -- column 9, the bias sum -- and 128 B of 0)
constexpr int FB_LDS_DMA = FB_L0C + 256;
static_assert(EW_PER * 4 <= FB_LDS, "the block partial is staged in the operand LDS");
static_assert(FB_LDS + FB_LDS_DMA <= 160 * 1024, "LDS budget of one CU");

struct BwdBranch {
    TV g3, glow, x;            // 2-block gradient view, 6-block view G0 | G1 | G2, 6-block view x0 | x1 | x2 (halo 0)
    const uint4* wpk[3];       // dgrad operand images of the virtual layers: [0] dst x0 (48 in), [1] dst x1 (32 in), [2] dst x2 (16 in)
    const float* img;          // [n][h][w] fp32
    float* partial;            // gridDim.x block partials of EW_PER floats
};
struct BwdArgs {
    BwdBranch br[2];
    int n, h, w;
    int nstrips;
    int rows_per_slot;         // what one wave pair walks of the branch's line of n * nstrips columns, in WEIGHTED rows: a column counts
                               // h + piece_cost (a piece -- the part of a column inside a slice -- pays about that many margin steps, so
                               // slices that straddle a column boundary get fewer rows and every pair about the same number of steps)
    int piece_cost;
    int nbarriers;             // barriers every wave executes: the largest (pieces + steps) of a pair
};

template <int N> struct FBI { static constexpr int value = N; };

// position on the line of rows (column c = rows [c h, (c + 1) h)) of the weighted position v: the piece_cost weighted rows at the end of a
// column map to the start of the next one
__host__ __device__ inline long long fb_line_pos(long long v, int h, int piece_cost, long long line_rows) {
    const long long hv = h + piece_cost, c = v / hv, y = v - c * hv;
    const long long pos = c * h + (y < h ? y : (long long)h);
    return pos < line_rows ? pos : line_rows;
}
// steps of the piece [y_lo, y_hi) of a column (set_piece below)
__host__ __device__ inline int fb_piece_steps(int y_lo, int y_hi) {
    const int a_lo = y_lo - 3 > 0 ? y_lo - 3 : 0;
    return 3 * ((y_hi + 2 - (a_lo - 3) + 2) / 3);
}

#define FB_FENCE() __builtin_amdgcn_sched_barrier(0)
// the pair's (and the block's) step barrier: LDS traffic of this wave retired, then s_barrier.  Inline asm with a memory clobber: the
// compiler keeps memory accesses on their side of it, and -- unlike __syncthreads() -- does not drain the vector-memory counter (the
// chain waves always have two steps of requests in flight)
#ifndef FB_ABL
#define FB_ABL 0
#endif
// FB_FRESH: how far lane coordinates are re-derived where they are used (fresh_lane below) instead of carried across the step loops: 0 nowhere
// (12 spilled registers, all outside the loops), 1 the tails and border rows, 2 + each role's own, 3 + each piece's (0 spills)
#ifndef FB_FRESH
#define FB_FRESH 3
#endif
#if FB_ABL & 64
#define FB_STEP_BARRIER() __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define FB_STEP_BARRIER() __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
// compile-time ablations (tools/build_ab_encbwd.sh; TIMING ONLY, results are wrong): 1 no weight-gradient products (role B idle, no w1),
// 2 no chain MFMAs, 4 no epilogues, 8 no global requests / LDS-DMA in the steps, 16 no first-layer products, 32 no chain operand reads,
// 64 no step barriers (the roles and the pairs run free)

__global__ __launch_bounds__(FB_WAVES * 64, 1) void enc_bwd_fused_kernel(BwdArgs A) {
    __shared__ __attribute__((aligned(16))) char smem[FB_LDS];
    // written by LDS-DMA only (inline asm: the compiler must not know, it would order every LDS access of the wave behind the pending
    // DMAs with s_waitcnt vmcnt(0), see csrc/enc_stream2.hip); read by plain loads (masks) and transposing reads (operands)
    __shared__ __attribute__((aligned(16))) char smem_dma[FB_LDS_DMA];
    const BwdBranch& B = A.br[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = wave & (FB_PAIRS - 1), role = wave / FB_PAIRS;
    int j = lane & 15, g = lane >> 4;      // (each role re-derives them -- and what hangs off them -- from its own lane id read: see fresh_lane)

    // ---- chain A fragments (tap-row major K order, csrc/enc_stream2.hip): virtual layer 1 = dst x2 (input g3), 2 = dst x1 ([g2 | g3]),
    // 3 = dst x0 ([g1 | g2] + g3)
    auto a_plane = [&](int kq, int u, int kg) {
        const int L = kq < 2 ? 1 : (kq < 5 ? 2 : 3), q = kq < 2 ? kq : (kq < 5 ? kq - 2 : kq - 5);
        if (L >= 2 && q < 3) return (u * 3 + q) * 4 + kg;
        const int q2 = L == 1 ? q : q - 3, base = L == 3 ? 36 : 0;
        return base + ((q2 == 1 && kg >= 2) ? 18 + (kg & 1) : (u * 3 + (q2 == 0 ? (kg >> 1) : 2)) * 2 + (kg & 1));
    };
    for (int e = tid; e < FB_NFRAG * 64; e += FB_WAVES * 64) {
        const int f = e >> 6, l = e & 63, oc = l & 15, kg = l >> 4;
        const int u = f % 3, kq = f / 3;
        // (lane group g of a fragment holds the k-group ((g & 1) << 1) | (g >> 1) of the packed image: the chain's B reads put the ring / tap
        //  selector in bit 0 of the lane group and the channel block in bit 1 -- with the channel block in bit 0, the 544-byte stride of the
        //  guarded rows put two of the four 16-lane groups of every ds_read_b128 on shared banks: 29 % of the chain's LDS cycles)
        reinterpret_cast<uint4*>(smem)[e] = B.wpk[kq < 2 ? 2 : (kq < 5 ? 1 : 0)][a_plane(kq, u, ((kg & 1) << 1) | (kg >> 1)) * 16 + oc];
    }
    const int ring = FB_WBYTES + pair * FB_GRING;
    const int xring = 64 + pair * (FB_XS * FB_XROW);                                // byte offsets inside smem_dma (64 zero bytes in front)
    const int iring = 64 + FB_PAIRS * FB_XS * FB_XROW + 64 + pair * (FB_IS * FB_IROW);
    if (role == 0) {
        for (int e = lane; e < FB_GRING / 16; e += 64) reinterpret_cast<uint4*>(smem + ring)[e] = make_uint4(0u, 0u, 0u, 0u);
    } else {
        for (int e = lane; e < FB_XS * FB_XROW / 16; e += 64) reinterpret_cast<uint4*>(smem_dma + xring)[e] = make_uint4(0u, 0u, 0u, 0u);
        for (int e = lane; e < FB_IS * FB_IROW / 16; e += 64) reinterpret_cast<uint4*>(smem_dma + iring)[e] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (tid < 4) {
        reinterpret_cast<uint4*>(smem + FB_WBYTES + FB_PAIRS * FB_GRING)[tid] = make_uint4(0u, 0u, 0u, 0u);
        reinterpret_cast<uint4*>(smem_dma)[tid] = make_uint4(0u, 0u, 0u, 0u);
        reinterpret_cast<uint4*>(smem_dma + 64 + FB_PAIRS * FB_XS * FB_XROW)[tid] = make_uint4(0u, 0u, 0u, 0u);
        reinterpret_cast<uint4*>(smem_dma + FB_L0C - 64)[tid] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (tid < 16) {
        const unsigned one = tid < 8 ? 0x3f800000u : 0u;
        reinterpret_cast<uint4*>(smem_dma + FB_L0C)[tid] = make_uint4(one, one, one, one);
    }
    __syncthreads();

    // ---- work split: the branch's (image, strip) columns of h rows form one line of n * nstrips * h rows; wave pair `slot` owns rows
    // [slot * rows_per_slot, + rows_per_slot) of it -- one or more PIECES (column, [y_lo, y_hi)), walked one after the other with the weight
    // gradients accumulating in registers across them (one block per CU, one partial sum per block).  Every wave of the block executes the
    // same number of barriers: its pieces' (1 + steps), then idle ones up to A.nbarriers.
    const int H = A.h, W = A.w;
    const long long line_rows = (long long)A.n * A.nstrips * H;
    const long long pos_begin = fb_line_pos((long long)(blockIdx.x * FB_PAIRS + pair) * A.rows_per_slot, H, A.piece_cost, line_rows);
    const long long pos_end = fb_line_pos((long long)(blockIdx.x * FB_PAIRS + pair + 1) * A.rows_per_slot, H, A.piece_cost, line_rows);
    int strip = 0, in_ = 0, y_lo = 0, y_hi = 0, NSTEP = 0;
    int r0 = 0, o_lo = 0, o_hi = 0, a_lo = 0, a_hi = 0, b_lo = 0, b_hi = 0, c_lo = 0, c_hi = 0, r_first = 0;
    unsigned km[4] = {0u, 0u, 0u, 0u};          // keep-mask of this lane's 8 pixels (the K order below; two bf16 per dword)
    // strip geometry (csrc/enc_chain.hip): region [r0, r0 + 32) of image columns, r0 = -1 for the first strip (column -1 and column w
    // belong to the edge strips, which fold them onto columns 1 / w-2); kept columns [o_lo, o_hi)
    auto strip_r0 = [&](int s) { return A.nstrips == 1 ? -1 : min(-1 + FB_KEEP * s, W - (FB_W - 1)); };
    auto strip_hi = [&](int r) { return (r + FB_W >= W + 1) ? W : min(r + FB_W - 3, W - 4); };
    auto set_piece = [&](long long pos) {          // the piece that starts at line position pos; returns the position after it
        const int col = (int)(pos / H);
        y_lo = (int)(pos - (long long)col * H);
        y_hi = (int)min((long long)H, y_lo + (pos_end - pos));
        strip = col % A.nstrips;
        in_ = col / A.nstrips;
        r0 = strip_r0(strip);
        o_hi = strip_hi(r0);
        o_lo = strip == 0 ? 0 : max(r0 + 3, strip_hi(strip_r0(strip - 1)));
        // rows each stage touches (as csrc/enc_chain.hip): g3 rows [a_lo, a_hi), g2 rows [b_lo, b_hi), g1 rows [c_lo, c_hi), g0 rows [y_lo, y_hi)
        a_lo = max(0, y_lo - 3); a_hi = min(H, y_hi + 3);
        b_lo = max(0, y_lo - 2); b_hi = min(H, y_hi + 2);
        c_lo = max(0, y_lo - 1); c_hi = min(H, y_hi + 1);
        r_first = a_lo - 3;
        NSTEP = fb_piece_steps(y_lo, y_hi);              // steps r_first .. y_hi + 1 (the first layer's products trail by two rows), in threes
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int c0 = r0 + (d < 2 ? 4 * g + 2 * d : 16 + 4 * g + 2 * (d - 2)), c1 = c0 + 1;
            km[d] = ((c0 >= o_lo && c0 < o_hi) ? 0xffffu : 0u) | ((c1 >= o_lo && c1 < o_hi) ? 0xffff0000u : 0u);
        }
        return pos + (y_hi - y_lo);
    };
    int nbar = 0;          // barriers executed so far

    // (the lane coordinates of the tail are re-derived from a lane id the compiler cannot tie to the one above: it would otherwise carry --
    //  spill -- j and g across the step loops for the tail's sake)
    auto fresh_lane = [&](int level = 1) __attribute__((always_inline)) {
        if (level > FB_FRESH) return lane;
        int l;
        __asm__ volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    // ---- lane constants of the transposing reads (weight-gradient operands: in-group lane sl supplies pixel sl >> 2 (+ 4), 4-channel chunk sl & 3)
    int tr_row = j >> 2, tr_c = j & 3;
    // K order of the weight-gradient products (any bijection pixel <-> k works as long as both operands use it): k = 8 g + e is pixel
    // 4 g + e for e < 4 and 16 + 4 g + (e - 4) for e >= 4 -- the 64 lanes of one transposing read then cover 16 CONSECUTIVE pixels (512
    // contiguous bytes of the pixel-major activation ring; lane groups 256 B apart shared their banks: a third of the LDS cycles of the
    // first version were bank conflicts), the second read of a fragment the other 16
    int ltr = (tr_c >> 1) * FB_CBS + 16 + (4 * g + tr_row) * 16 + (tr_c & 1) * 8;        // gradient rings
    int ltr_x = (4 * g + tr_row) * 32 + tr_c * 8;                                       // activation ring (pixel-major: 32 B = 16 channels)
    auto tr_frag = [&](int addr) __attribute__((always_inline)) {
        const fb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(FB_LDS_PTR(fb_s16x4, smem + addr));
        const fb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(FB_LDS_PTR(fb_s16x4, smem + addr + 256));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto tr_frag_x = [&](int addr) __attribute__((always_inline)) {
        const fb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(FB_LDS_PTR(fb_s16x4, smem_dma + addr));
        const fb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(FB_LDS_PTR(fb_s16x4, smem_dma + addr + 512));
        return __builtin_bit_cast(fb_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto masked_g = [&](int addr) __attribute__((always_inline)) {      // one gradient row, k-major, pixels outside the kept columns zeroed
        fb_u32x4 raw = __builtin_bit_cast(fb_u32x4, tr_frag(addr + ltr));
#pragma unroll
        for (int d = 0; d < 4; ++d) raw[d] &= km[d];
        return __builtin_bit_cast(fb_bf16x8, raw);
    };
    auto crow = [&](int y) { return (unsigned)min(max(y, 0), H - 1); };
    auto rrow = [&](int y) { return min(max(reflect_idx(y, H), 0), H - 1); };
    auto xslot = [&](int y) { return (int)((unsigned)y % (unsigned)FB_XS); };      // (y >= 0)
    const fb_bf16x8 ones = __builtin_bit_cast(fb_bf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));

    // ---- block partial (after the steps): the pairs' accumulators summed in LDS (natural [o][c][u][v] order, enc_wgrad.hpp) -> one
    // coalesced copy.  Both roles call it with their own accumulators (same number of barriers)
    auto block_partial = [&](auto&& puts) __attribute__((always_inline)) {
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        for (int wv = 0; wv < FB_PAIRS; ++wv) {
            if (pair == wv) puts([&](int idx, float v) { red[idx] = wv == 0 ? v : red[idx] + v; });
            __syncthreads();
        }
        float* dst = B.partial + (long long)blockIdx.x * EW_PER;
        for (int e = wave * 64 + fresh_lane(); e < EW_PER; e += FB_WAVES * 64) dst[e] = red[e];
    };
    const fb_f32x4 zero4 = (fb_f32x4){0.f, 0.f, 0.f, 0.f};
    auto img_base = [&](const TV& t) { return t.base + ((long long)in_ * t.img + (long long)t.cb_off * t.plane) * 16; };

    if (role == 1) {
        // ================= role B: dW3, dW2, db3, db2 of row r = g3 / g2 row r (final since step r - 2) against the activation rows
        // R(r-1), r, R(r+1): nine chunks of one tap (u, v), operands double-buffered
        {   // this role's own lane coordinates (nothing lane-derived is shared -- kept live, spilled -- across the role split)
            const int l2 = fresh_lane(2);
            j = l2 & 15; g = l2 >> 4; tr_row = j >> 2; tr_c = j & 3;
            ltr = (tr_c >> 1) * FB_CBS + 16 + (4 * g + tr_row) * 16 + (tr_c & 1) * 8;
            ltr_x = (4 * g + tr_row) * 32 + tr_c * 8;
        }
        fb_f32x4 w3[3][3][3], w2[3][3][2], accb[3], acc0 = zero4;
        const int un = min(j / 3, 2), vn = j - 3 * (j / 3);
        const int l0_const = FB_L0C + (j == 9 ? 0 : 128);
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v) {
#pragma unroll
                for (int b = 0; b < 3; ++b) w3[u][v][b] = zero4;
#pragma unroll
                for (int b = 0; b < 2; ++b) w2[u][v][b] = zero4;
            }
#pragma unroll
        for (int i = 0; i < 3; ++i) accb[i] = zero4;
        const unsigned x_plane = (unsigned)(B.x.plane * 16), x_rowb = (unsigned)B.x.ws * 16u;
#pragma unroll 1
        for (long long pos = pos_begin; pos < pos_end;) {
        pos = set_piece(pos);
        // activation / image rows by LDS-DMA (this role has the slack to issue them and to wait for them): ring pixel p = image column reflect(r0 + p) (the edge strips' ghost pixels hold the reflected column)
        const char* x_img = img_base(B.x);
        const char* im_img = reinterpret_cast<const char*>(B.img + (long long)in_ * H * W);
        const unsigned xring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_dma + (unsigned)xring;
        const unsigned iring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_dma + (unsigned)iring;
        const int cdma = min(max(reflect_idx(r0 + (lane & 31), W), 0), W - 1);          // image row: lane = pixel
        const int cdmx = min(max(reflect_idx(r0 + (lane >> 1), W), 0), W - 1);          // activation rows: lane = 2 pixel + channel block
        const unsigned xdma_off = (unsigned)(lane & 1) * x_plane + (unsigned)cdmx * 16u;
        // (scalar row base + one 32-bit lane offset: no 64-bit lane pointers to keep in -- or spill from -- the vector registers)
        auto dma16 = [&](const char* sbase, unsigned voff, unsigned lds_dst) __attribute__((always_inline)) {
            __asm__ volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_dst), "v"(voff), "s"(sbase) : "memory", "m0");
        };
        auto dma4 = [&](const char* sbase, unsigned voff, unsigned lds_dst) __attribute__((always_inline)) {
            __asm__ volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" : : "s"(lds_dst), "v"(voff), "s"(sbase) : "memory", "m0");
        };
        const unsigned idma_off = (unsigned)cdma * 4u;
        auto dma_rows = [&](int y) __attribute__((always_inline)) {   // x0 | x1 | x2 row (3 x 1 KiB) + image row of image row clamp(y)
            const int yy = (int)crow(y);
            const char* src = x_img + (unsigned long long)((unsigned)yy * x_rowb);
            const unsigned dst = xring_lds + (unsigned)(xslot(yy) * FB_XROW);
#pragma unroll
            for (int i = 0; i < 3; ++i) dma16(src + (unsigned long long)((unsigned)(2 * i) * x_plane), xdma_off, dst + (unsigned)i * 1024u);
            if (lane < 32) dma4(im_img + (long long)yy * W * 4, idma_off, iring_lds + (unsigned)((yy & (FB_IS - 1)) * FB_IROW));
        };
        // prologue: rows r_first - 1 .. r_first + 3 (step r requests row r + 4 at its start and has it landed at its barrier: the chain waves
        // first touch row r + 4 in step r + 2)
        if (!(FB_ABL & 8)) {
#pragma unroll 1
            for (int y = max(r_first - 1, 0); y < r_first + 4; ++y)
                if (y >= 0) dma_rows(y);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);
        FB_STEP_BARRIER();
#pragma unroll 1
        for (int s = 0; s < NSTEP; ++s) {
            const int r = r_first + s;
            if (!(FB_ABL & 8)) dma_rows(r + 4);
            // first layer, row r - 2 (g0 row r - 2 left the chain in step r - 1): D[o][n] += sum_p g0[o](p) B[p][n], B[p][n] = image(p + tap n)
            // for n < 9, 1 for n = 9 (-> db0): exact fp32.  MFMA e takes the pixels of k = 8 g + e (K order above): its A operand is element e of the transposed
            // fragment a bf16 product would use
            const bool l0 = !(FB_ABL & 16) && r - 2 >= y_lo && r - 2 < y_hi;
            if (!(FB_ABL & 1) && r >= y_lo && r < y_hi) {
                fb_bf16x8 ag2 = masked_g(ring + FB_G2 + (r & (FB_S2 - 1)) * FB_ROW), ag3 = masked_g(ring + FB_G3 + (r & (FB_S3 - 1)) * FB_ROW);
                int xs3[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) xs3[u] = xring + xslot(rrow(r + u - 1)) * FB_XROW + ltr_x;
                fb_bf16x8 bx[2][3];
                auto wg_load = [&](auto Cc) __attribute__((always_inline)) {
                    constexpr int C = decltype(Cc)::value, u = C / 3, v = C % 3;
#pragma unroll
                    for (int b = 0; b < 3; ++b) bx[C & 1][b] = tr_frag_x(xs3[u] + b * 1024 + (v - 1) * 32);
                };
                auto wg_mma = [&](auto Cc) __attribute__((always_inline)) {
                    constexpr int C = decltype(Cc)::value, u = C / 3, v = C % 3;
#pragma unroll
                    for (int b = 0; b < 3; ++b) w3[u][v][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ag3, bx[C & 1][b], w3[u][v][b], 0, 0, 0);
#pragma unroll
                    for (int b = 0; b < 2; ++b) w2[u][v][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ag2, bx[C & 1][b], w2[u][v][b], 0, 0, 0);
                };
                wg_load(FBI<0>());
                FB_FENCE();
                wg_load(FBI<1>()); wg_mma(FBI<0>());
                FB_FENCE();
                wg_load(FBI<2>()); wg_mma(FBI<1>());
                FB_FENCE();
                wg_load(FBI<3>()); wg_mma(FBI<2>());
                FB_FENCE();
                wg_load(FBI<4>()); wg_mma(FBI<3>());
                FB_FENCE();
                wg_load(FBI<5>()); wg_mma(FBI<4>());
                FB_FENCE();
                wg_load(FBI<6>()); wg_mma(FBI<5>());
                FB_FENCE();
                wg_load(FBI<7>()); wg_mma(FBI<6>());
                FB_FENCE();
                wg_load(FBI<8>()); wg_mma(FBI<7>());
                FB_FENCE();
                wg_mma(FBI<8>());
                accb[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ag2, ones, accb[1], 0, 0, 0);   // every column = the sum
                accb[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ag3, ones, accb[2], 0, 0, 0);
                accb[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(masked_g(ring + FB_G1 + (r & (FB_S1 - 1)) * FB_ROW), ones, accb[0], 0, 0, 0);
                FB_FENCE();
            }
            if (l0) {     // (operands read here, not ahead: this role has the slack, not the registers)
                // the fp32 image as THREE bf16 pieces (hi = the top 16 bits, mid = the top 16 bits of the exact remainder, lo = what is left:
                // 8 + 8 + 8 significand bits, every piece exact) against the bf16 gradient on the 16-cycle bf16 MFMA: bf16 x bf16 products are
                // exact in fp32, so the three products sum to the fp32 product -- 48 MFMA cycles instead of the 256 of eight 16x16x4 fp32 ones
                const fb_bf16x8 l0g = masked_g(ring + FB_G0 + ((r - 2) & (FB_S0 - 1)) * FB_ROW);
                const int ib = j < 9 ? iring + (rrow(r - 2 + un - 1) & (FB_IS - 1)) * FB_IROW + (vn - 1) * 4 + 16 * g : l0_const;   // (pixel 4 g + e | 16 + 4 g + e - 4)
                uint32_t hi[8], mid[8], lo[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = *reinterpret_cast<const float*>(smem_dma + ib + (e < 4 ? e * 4 : 64 + (e - 4) * 4));
                    hi[e] = __float_as_uint(x) & 0xffff0000u;
                    const float r1 = x - __uint_as_float(hi[e]);
                    mid[e] = __float_as_uint(r1) & 0xffff0000u;
                    lo[e] = __float_as_uint(r1 - __uint_as_float(mid[e]));
                }
                auto pk8 = [&](const uint32_t* v) __attribute__((always_inline)) {      // the top halves of eight dwords -> bf16 x 8
                    fb_u32x4 o;
#pragma unroll
                    for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_perm(v[2 * d + 1], v[2 * d], 0x07060302u);
                    return __builtin_bit_cast(fb_bf16x8, o);
                };
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l0g, pk8(lo), acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l0g, pk8(mid), acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l0g, pk8(hi), acc0, 0, 0, 0);
            }
            FB_FENCE();
            // the rows requested in the PREVIOUS step have landed (this role's only vector-memory operations are the DMAs, four per step:
            // row r + 4 is requested at the start of step r, published by the barrier of step r + 1, first touched by the chain in step r + 2)
            __builtin_amdgcn_s_waitcnt(0x0f70 | 4);
            FB_STEP_BARRIER();
        }
        nbar += 1 + NSTEP;
        }
#pragma unroll 1
        for (; nbar < A.nbarriers; ++nbar) FB_STEP_BARRIER();
        block_partial([&](auto put) {
            const int l2 = fresh_lane(), j = l2 & 15, g = l2 >> 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) put(EW_OFF0 + (4 * g + q) * 16 + j, acc0[q]);
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    const int t = 3 * u + v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int oc = 4 * g + q;
#pragma unroll
                        for (int b = 0; b < 3; ++b) put(EW_OFF3 + (oc * 48 + 16 * b + j) * 9 + t, w3[u][v][b][q]);
#pragma unroll
                        for (int b = 0; b < 2; ++b) put(EW_OFF2 + (oc * 32 + 16 * b + j) * 9 + t, w2[u][v][b][q]);
                    }
                }
            if (j == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int L = 0; L < 3; ++L) put(EW_OFFB + L * 16 + 4 * g + q, accb[L][q]);
            }
        });
    } else {
        // ================= role A: the gradient chain, dW1 / db1 and the first layer's gradients
        {   // this role's own lane coordinates (nothing lane-derived is shared -- kept live, spilled -- across the role split)
            const int l2 = fresh_lane(2);
            j = l2 & 15; g = l2 >> 4; tr_row = j >> 2; tr_c = j & 3;
            ltr = (tr_c >> 1) * FB_CBS + 16 + (4 * g + tr_row) * 16 + (tr_c & 1) * 8;
            ltr_x = (4 * g + tr_row) * 32 + tr_c * 8;
        }
        fb_f32x4 w1[3][3];
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v) w1[u][v] = zero4;
#pragma unroll 1
        for (long long pos = pos_begin; pos < pos_end;) {
        pos = set_piece(pos);
        const bool edgeL = r0 < 0, edgeR = r0 + FB_W >= W + 1;
        // ---- lane constants: chain operands
        const int h2 = g & 1, cbk = g >> 1;      // (bit 0: second ring / next tap column, bit 1: channel block -- see the fragment staging)
        const int la = g * 256 + j * 16;
        const int lb4 = ring + cbk * FB_CBS + j * 16;                 // (+ 16 guard - 16 for tap column 0)
        const int lb2 = ring + cbk * FB_CBS + (j + h2) * 16;
        // chain epilogue (after the row swap): this lane holds the granule of pixel 16 (g & 1) + j, channel block g >> 1
        const int px_e = 16 * (g & 1) + j, cb_e = g >> 1;
        const int x_e = r0 + px_e;
        const bool in_e = x_e >= 0 && x_e < W;
        const int lw_e = ring + 16 + cb_e * FB_CBS + px_e * 16;
        const uint32_t inm_e = in_e ? 0xffffffffu : 0u;
        const bool tgtL = edgeL && x_e == 1, tgtR = edgeR && x_e == W - 2;
        const float mulL = tgtL ? 1.f : 0.f, mulR = tgtR ? 1.f : 0.f;
        // g3 rows: lane = (pixel lane & 31, channel block lane >> 5)
        // (from a lane id read inside the piece loop: the plane products below are loop-invariant otherwise, hoisted and spilled)
        const int lane_p = fresh_lane(3), jp = lane_p & 15, gp = lane_p >> 4;
        const int px_a = lane_p & 31, cb_a = lane_p >> 5;
        const int x_a = r0 + px_a;
        const bool in_a = x_a >= 0 && x_a < W;
        const int lw_a = ring + FB_G3 + 16 + cb_a * FB_CBS + px_a * 16;

        // ---- global operands through buffer descriptors (32-bit lane offsets; bit 31 = beyond the descriptor = reads as zero)
        const unsigned g3_plane = (unsigned)(B.g3.plane * 16), g3_row = (unsigned)B.g3.ws * 16u, g3_org = (unsigned)(B.g3.halo * (B.g3.ws + 1)) * 16u;
        const unsigned gl_plane = (unsigned)(B.glow.plane * 16), gl_row = (unsigned)B.glow.ws * 16u, gl_org = (unsigned)(B.glow.halo * (B.glow.ws + 1)) * 16u;
        const __amdgpu_buffer_rsrc_t rs_g3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(img_base(B.g3)), 0, (int)((unsigned)(B.g3.cb_total - B.g3.cb_off) * g3_plane), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_gl = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(img_base(B.glow)), 0, (int)((unsigned)(B.glow.cb_total - B.glow.cb_off) * gl_plane), 0x00020000);
        const unsigned g3_off = in_a ? (unsigned)cb_a * g3_plane + (unsigned)x_a * 16u + g3_org : 0x80000000u;
        unsigned gl_off[2];      // G fragments in accumulator layout: this lane's channels 4 g .. 4 g + 3 of pixel 16 t + j = 8 bytes of a granule
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int c = r0 + 16 * t + jp;
            gl_off[t] = (c >= 0 && c < W) ? (unsigned)(gp >> 1) * gl_plane + (unsigned)c * 16u + (unsigned)(gp & 1) * 8u + gl_org : 0x80000000u;
        }
        fb_u32x4 pg3;            // (the g3 row travels one step ahead only: one register set)
        fb_u32x2 pG[3][2];       // C operands of the fresh accumulators of the NEXT step: each layer's pair is re-requested right after this
                                 // step's first k-step of that layer consumed it
        auto request_g3 = [&](int y) __attribute__((always_inline)) {
            pg3 = __builtin_amdgcn_raw_buffer_load_b128(rs_g3, (int)g3_off, (int)(crow(y) * g3_row), 0);
        };
        // input row of chain layer Lc at step r: R = r + RO(Lc) -- layer 1 on g3 row r + 3 -> g2 row r + 2, layer 2 on [g2 | g3] row r + 2 ->
        // g1 row r + 1 (the one dependent pair of a step), layer 3 one step BEHIND on [g1 | g2 | g3] row r -> g0 row r - 1 (its operands are a
        // step old: it fills the waits of the dependent pair); each opens out row R + 1 from G(3 - Lc) row R + 1
#define FB_RO(Lc) ((Lc) == 1 ? 3 : ((Lc) == 2 ? 2 : 0))
        auto request_G = [&](int Lc, int r) __attribute__((always_inline)) {      // for step r
#pragma unroll
            for (int t = 0; t < 2; ++t)
                pG[Lc - 1][t] = __builtin_amdgcn_raw_buffer_load_b64(rs_gl, (int)gl_off[t], (int)((unsigned)(2 * (3 - Lc)) * gl_plane + crow(r + FB_RO(Lc) + 1) * gl_row), 0);
        };
        auto g_c = [&](const fb_u32x2& v) {
            return (fb_f32x4){__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16), __uint_as_float(v[1] & 0xffff0000u)};
        };

        fb_f32x4 acc[3][3][2];
#pragma unroll
        for (int L = 0; L < 3; ++L)
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[L][q][t] = (fb_f32x4){0.f, 0.f, 0.f, 0.f};

        // ---- one row step (P = (r - r_first) % 3 at compile time; FAST: every stage active, no border row)
        auto step = [&](auto Pc, auto Fc, int r) __attribute__((always_inline)) {
            constexpr int P = decltype(Pc)::value;
            constexpr bool FAST = decltype(Fc)::value != 0;
            const int R1 = r + 3, R2 = r + 2, R3 = r;
            const bool on1 = FAST || (R1 >= a_lo && R1 < a_hi), on2 = FAST || (R2 >= b_lo && R2 < b_hi), on3 = FAST || (R3 >= c_lo && R3 < c_hi);
            const bool em1 = FAST || (R1 - 1 >= b_lo && R1 - 1 < b_hi), em2 = FAST || (R2 - 1 >= c_lo && R2 - 1 < c_hi),
                       em3 = FAST || (R3 - 1 >= y_lo && R3 - 1 < y_hi);
            const bool wg = !(FB_ABL & 1) && (FAST || (r >= y_lo && r < y_hi));              // dW1, db1 of row r
            const int bL1 = lb2 + FB_G3 + (R1 & (FB_S3 - 1)) * FB_ROW;
            const int bL2 = lb4 + (h2 ? FB_G3 + (R2 & (FB_S3 - 1)) * FB_ROW : FB_G2 + (R2 & (FB_S2 - 1)) * FB_ROW);
            const int bL3 = lb4 + (h2 ? FB_G2 + (R3 & (FB_S2 - 1)) * FB_ROW : FB_G1 + (R3 & (FB_S1 - 1)) * FB_ROW);
            const int bL3x = lb2 + FB_G3 + (R3 & (FB_S3 - 1)) * FB_ROW;
            fb_bf16x8 fa[3][3], fbr[3][2];     // three operand sets: a k-step's reads are issued two regions before its products

            auto load_k = [&](auto Nc, auto Bc) __attribute__((always_inline)) {
                constexpr int N = decltype(Nc)::value, S = decltype(Bc)::value;
                if (FB_ABL & 32) return;
                const char* pa = smem + la + N * 3 * 1024;
#pragma unroll
                for (int u = 0; u < 3; ++u) fa[S][u] = *reinterpret_cast<const fb_bf16x8*>(pa + u * 1024);
                const char* pb = smem + (N < 2 ? bL1 + 32 * N : (N < 5 ? bL2 + 16 * (N - 2) : (N < 8 ? bL3 + 16 * (N - 5) : bL3x + 32 * (N - 8))));
#pragma unroll
                for (int t = 0; t < 2; ++t) fbr[S][t] = *reinterpret_cast<const fb_bf16x8*>(pb + t * 256);
            };
            auto mma_k = [&](auto Nc, auto Bc) __attribute__((always_inline)) {
                constexpr int N = decltype(Nc)::value, S = decltype(Bc)::value;
                constexpr int Lc = N < 2 ? 1 : (N < 5 ? 2 : 3);
                constexpr bool first = N == 0 || N == 2 || N == 5;
                constexpr int i1 = (P + FB_RO(Lc)) % 3, i0 = (i1 + 1) % 3, i2 = (i1 + 2) % 3;      // accumulator rows of out rows R, R+1, R-1
                if (FB_ABL & 2) return;
                if (!(Lc == 1 ? on1 : (Lc == 2 ? on2 : on3))) return;
                const int R = r + FB_RO(Lc);
                // adjoint of reflect padding along y: padded row -1 (= tap row 2 of input row 0) folds onto row 1, padded row h onto row h-2
                const bool top = !FAST && R == 0, bot = !FAST && R == H - 1;
                if (!FAST && first && R == 0) {   // row 0 has no row above it to open its accumulator: start it from G row 0 here
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        acc[Lc - 1][i1][t] = g_c(__builtin_amdgcn_raw_buffer_load_b64(rs_gl, (int)gl_off[t], (int)((unsigned)(2 * (3 - Lc)) * gl_plane), 0));
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const fb_bf16x8 b = fbr[S][t];
                    acc[Lc - 1][i0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[S][0], b, first ? g_c(pG[Lc - 1][t]) : acc[Lc - 1][i0][t], 0, 0, 0);
                    acc[Lc - 1][i1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[S][1], b, acc[Lc - 1][i1][t], 0, 0, 0);
                    acc[Lc - 1][i2][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[S][2], b, acc[Lc - 1][i2][t], 0, 0, 0);
                    if (!FAST) {
                        if (top) acc[Lc - 1][i0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[S][2], b, acc[Lc - 1][i0][t], 0, 0, 0);
                        if (bot) acc[Lc - 1][i2][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[S][0], b, acc[Lc - 1][i2][t], 0, 0, 0);
                    }
                }
            };
            // ReLU mask of an epilogue: x(3 - Lc) row rho from the activation ring, read one region ahead of its use
            uint4 xq;
            auto mask_load = [&](auto Lc_) __attribute__((always_inline)) {
                constexpr int Lc = decltype(Lc_)::value;
                if ((FB_ABL & 4) || !(Lc == 1 ? em1 : (Lc == 2 ? em2 : em3))) return;
                const int rho = r + FB_RO(Lc) - 1;
                xq = *reinterpret_cast<const uint4*>(smem_dma + xring + xslot((int)crow(rho)) * FB_XROW + (3 - Lc) * 1024 + px_e * 32 + cb_e * 16);
            };
            // epilogue of chain layer Lc: out row rho = R - 1 of g(3 - Lc): pair the column tiles, fold the edge columns, round once, ReLU mask
            auto epilogue = [&](auto Lc_) __attribute__((always_inline)) {
                constexpr int Lc = decltype(Lc_)::value;
                constexpr int i2 = ((P + FB_RO(Lc)) % 3 + 2) % 3;
                constexpr int XO = Lc == 1 ? FB_G2 : (Lc == 2 ? FB_G1 : FB_G0), SO = Lc == 1 ? FB_S2 : (Lc == 2 ? FB_S1 : FB_S0);
                if (FB_ABL & 4) return;
                if (!(Lc == 1 ? em1 : (Lc == 2 ? em2 : em3))) return;
                const int rho = r + FB_RO(Lc) - 1;
                float c[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[Lc - 1][i2][0][q]), __float_as_uint(acc[Lc - 1][i2][1][q]), false, false);
                    c[q] = __uint_as_float(sw[0]);
                    c[4 + q] = __uint_as_float(sw[1]);
                }
                // the adjoint of reflect padding along x (edge strips; a lane is the target of at most one fold: w >= 4).  Branch-free steps: by
                // DPP row shifts -- a wave-uniform branch with LDS traffic inside (ds_bpermute) makes the compiler drain the whole LDS queue,
                // prefetched operands included, at its join; source and target share a 16-lane row there (fold_cross: no branch-free steps)
                if (FAST) {
                    // (one fused multiply-add per value and side, its source operand through DPP; the multiplier is the target lanes' 1.0 / 0.0)
                    if (edgeL) {
#pragma unroll
                        for (int i = 0; i < 8; ++i)      // row_shr:2: lane j <- lane j - 2 (0 where the row has no such lane)
                            c[i] = fmaf(__uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(c[i]), 0x112, 0xf, 0xf, true)), mulL, c[i]);
                    }
                    if (edgeR) {
#pragma unroll
                        for (int i = 0; i < 8; ++i)      // row_shl:2: lane j <- lane j + 2
                            c[i] = fmaf(__uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(c[i]), 0x102, 0xf, 0xf, true)), mulR, c[i]);
                    }
                } else if (edgeL || edgeR) {
                    // (border rows only: the source lane is worked out here, from a fresh lane id, not carried through the fast steps)
                    const int l2 = fresh_lane(), px2 = 16 * ((l2 >> 4) & 1) + (l2 & 15), cb2 = l2 >> 5;
                    const int x2 = r0 + px2;
                    const bool tL = edgeL && x2 == 1, tR = edgeR && x2 == W - 2;
                    const int ps = tL ? max(px2 - 2, 0) : min(px2 + 2, FB_W - 1);
                    const int src4 = ((cb2 * 2 + (ps >> 4)) * 16 + (ps & 15)) * 4;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float f = __uint_as_float((unsigned)__builtin_amdgcn_ds_bpermute(src4, (int)__float_as_uint(c[i])));
                        c[i] += (tL || tR) ? f : 0.f;
                    }
                }
                // bf16 > 0  <=>  as int16 > 0 (negative zero and negatives are <= 0): packed max(x, 0) -> min(., 1) -> 0 - . = 0xffff per kept half
                // (inline asm: the compiler turns the same three vector operations into two compares and two selects per dword); columns
                // outside the image (edge strips only): zero
                const uint32_t xw[4] = {xq.x, xq.y, xq.z, xq.w};
                uint32_t gr[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    uint32_t m;
                    __asm__("v_pk_max_i16 %0, %1, 0\n\tv_pk_min_u16 %0, %0, %2\n\tv_pk_sub_u16 %0, 0, %0" : "=&v"(m) : "v"(xw[i]), "s"(0x00010001u));
                    gr[i] = pack_bf16x2(c[2 * i], c[2 * i + 1]) & m;
                }
                if (edgeL || edgeR) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) gr[i] &= inm_e;
                }
                *reinterpret_cast<uint4*>(smem + lw_e + XO + (rho & (SO - 1)) * FB_ROW) = make_uint4(gr[0], gr[1], gr[2], gr[3]);
            };
            // ---- dW1 / db1 of row r: g1 row r (written by the previous step) against x0 rows R(r-1), r, R(r+1), one tap row per chunk
            fb_bf16x8 ag1, bx;
            auto w1_load = [&](auto Cc) __attribute__((always_inline)) {
                constexpr int C = decltype(Cc)::value, u = C / 3, v = C % 3;
                if (!wg) return;
                if (C == 0) ag1 = masked_g(ring + FB_G1 + (r & (FB_S1 - 1)) * FB_ROW);
                bx = tr_frag_x(xring + xslot(rrow(r + u - 1)) * FB_XROW + ltr_x + (v - 1) * 32);
            };
            auto w1_mma = [&](auto Cc) __attribute__((always_inline)) {
                constexpr int C = decltype(Cc)::value, u = C / 3, v = C % 3;
                if (!wg) return;
                w1[u][v] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ag1, bx, w1[u][v], 0, 0, 0);
            };
            constexpr FBI<0> b0;
            constexpr FBI<1> b1;
            constexpr FBI<2> b2;
            // order of the k-steps: 0 5 1 6 7 | E1 | 2 8 3 9 4 | E3 E2  (layers 1 -> 2 are the dependent pair, layer 3's k-steps 5..9 and
            // the dW1 chunks fill between them)
            load_k(FBI<0>(), b0); load_k(FBI<5>(), b1); w1_load(FBI<0>());
            FB_FENCE();
            load_k(FBI<1>(), b2); mma_k(FBI<0>(), b0); w1_mma(FBI<0>());
            if (!(FB_ABL & 8)) request_G(1, r + 1);      // (every step, whether or not the layer ran: the set always holds the NEXT step's rows)
            FB_FENCE();
            load_k(FBI<6>(), b0); w1_load(FBI<1>()); mma_k(FBI<5>(), b1);
            if (!(FB_ABL & 8)) request_G(3, r + 1);
            FB_FENCE();
            load_k(FBI<7>(), b1); mma_k(FBI<1>(), b2); w1_mma(FBI<1>());
            FB_FENCE();
            w1_load(FBI<2>()); mask_load(FBI<1>()); mma_k(FBI<6>(), b0);
            FB_FENCE();
            epilogue(FBI<1>());
            load_k(FBI<2>(), b2);      // (layer 2 reads the g2 row the epilogue above just wrote)
            w1_mma(FBI<2>());
            FB_FENCE();
            load_k(FBI<8>(), b0); w1_load(FBI<3>()); mma_k(FBI<7>(), b1);
            FB_FENCE();
            load_k(FBI<3>(), b1); mma_k(FBI<2>(), b2); w1_mma(FBI<3>());
            if (!(FB_ABL & 8)) request_G(2, r + 1);
            FB_FENCE();
            load_k(FBI<9>(), b2); w1_load(FBI<4>()); mma_k(FBI<8>(), b0);
            FB_FENCE();
            load_k(FBI<4>(), b0); mma_k(FBI<3>(), b1); w1_mma(FBI<4>());
            FB_FENCE();
            w1_load(FBI<5>()); mma_k(FBI<9>(), b2);
            FB_FENCE();
            mask_load(FBI<3>()); mma_k(FBI<4>(), b0); w1_mma(FBI<5>());
            FB_FENCE();
            epilogue(FBI<3>()); w1_load(FBI<6>()); mask_load(FBI<2>());
            FB_FENCE();
            w1_mma(FBI<6>()); w1_load(FBI<7>());
            FB_FENCE();
            epilogue(FBI<2>()); w1_mma(FBI<7>()); w1_load(FBI<8>());
            FB_FENCE();
            w1_mma(FBI<8>());
            FB_FENCE();
            FB_STEP_BARRIER();
            // ---- the step's global traffic, in one place, after the barrier (role B has finished with the slots overwritten here): g3 row
            // r + 4 (set of step r + 1) into its ring slot, this set reloaded for step r + 3, the activation / image rows r + 4 requested
            {
                const int ya = r + 4;
                if (FAST || (ya >= a_lo && ya < a_hi)) *reinterpret_cast<fb_u32x4*>(smem + lw_a + (ya & (FB_S3 - 1)) * FB_ROW) = pg3;
            }
            if (!(FB_ABL & 8)) request_g3(r + 5);
            FB_FENCE();
        };

        // ---- the pipeline: steps r = r_first .. r_first + NSTEP - 1, phase 0 at r_first.  Prologue: sets 0..2 (steps r_first .. r_first + 2),
        // activation / image rows r_first - 1 .. r_first + 3, g3 row r_first + 3 into its slot.
        request_g3(r_first + 3);
#pragma unroll
        for (int Lc = 1; Lc <= 3; ++Lc) request_G(Lc, r_first);
        if (r_first + 3 >= a_lo && r_first + 3 < a_hi) *reinterpret_cast<fb_u32x4*>(smem + lw_a + ((r_first + 3) & (FB_S3 - 1)) * FB_ROW) = pg3;
        FB_STEP_BARRIER();      // (the weight-gradient role's prologue rows have landed)
        request_g3(r_first + 4);
        // branch-free steps: every stage active and emitting, no stage at image row 0 / h-1, the g3 row written at the end inside [a_lo, a_hi)
        const int r_end = r_first + NSTEP;
        // (fold_cross: one strip of width 15 / 16 -- the right fold's source and target pixels sit in different 16-lane rows)
        const bool fold_cross = A.nstrips == 1 && (W == 15 || W == 16);
        const int f_lo = max(y_lo + 2, 1), f_hi = fold_cross ? f_lo : min(min(y_hi, a_hi - 4), H - 4);
        int r = r_first;
#pragma unroll 1
        for (int part = 0; part < 2; ++part) {
            const int stop = part == 0 ? min(r_end, f_lo) : r_end;
#pragma unroll 1
            for (; r < stop; r += 3) {
                step(FBI<0>(), FBI<0>(), r);
                step(FBI<1>(), FBI<0>(), r + 1);
                step(FBI<2>(), FBI<0>(), r + 2);
            }
            if (part == 0) {
#pragma unroll 1
                for (; r + 2 < f_hi; r += 3) {
                    step(FBI<0>(), FBI<1>(), r);
                    step(FBI<1>(), FBI<1>(), r + 1);
                    step(FBI<2>(), FBI<1>(), r + 2);
                }
            }
        }
        nbar += 1 + NSTEP;
        }
#pragma unroll 1
        for (; nbar < A.nbarriers; ++nbar) FB_STEP_BARRIER();
        block_partial([&](auto put) {
            const int l2 = fresh_lane(), j = l2 & 15, g = l2 >> 4;
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int v = 0; v < 3; ++v)
#pragma unroll
                    for (int q = 0; q < 4; ++q) put(EW_OFF1 + ((4 * g + q) * 16 + j) * 9 + 3 * u + v, w1[u][v][q]);
        });
    }
}
#undef FB_FENCE
#undef FB_RO

// one block per CU (and branch): the line of n * nstrips columns in slices of equal WEIGHTED length (at least 8 rows + one piece's cost);
// nbarriers = the most barriers any pair executes (a piece of k rows: 1 + its steps), counted here slice by slice; the weight of a piece is
// the candidate with the smallest such maximum.  (The last geometry is kept: a training loop asks for the same one every step.)
struct FbGeo { int n, h, w, nb, ncu, nstrips, rows_per_slot, piece_cost, nbarriers, nblocks; };
static void fb_geometry(int n, int h, int w, int nb, FbGeo& G) {
    static std::mutex mu;
    static FbGeo last = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int ncu = cached_num_cus();
    std::lock_guard<std::mutex> lk(mu);
    if (last.n == n && last.h == h && last.w == w && last.nb == nb && last.ncu == ncu) { G = last; return; }
    G.n = n; G.h = h; G.w = w; G.nb = nb; G.ncu = ncu;
    G.nstrips = w <= FB_W - 2 ? 1 : (w - (FB_W - 2) + FB_KEEP - 1) / FB_KEEP + 1;
    const long long cols = (long long)n * G.nstrips, line_rows = cols * h;
    const int gmax = std::max(1, std::min(ncu / nb, EW_MAXG / 2));          // (block partials of both branches share one workspace of EW_MAXG slots)
    G.nbarriers = 0;
    for (int cost = 0; cost <= 12; cost += 2) {
        const long long weighted = cols * (h + cost);
        const int rps = (int)std::max<long long>(8 + cost, (weighted + (long long)gmax * FB_PAIRS - 1) / ((long long)gmax * FB_PAIRS));
        const int nblocks = (int)((weighted + (long long)rps * FB_PAIRS - 1) / ((long long)rps * FB_PAIRS));
        int worst = 0;
        for (long long slot = 0; slot < (long long)nblocks * FB_PAIRS; ++slot) {
            long long pos = fb_line_pos(slot * rps, h, cost, line_rows);
            const long long end = fb_line_pos((slot + 1) * rps, h, cost, line_rows);
            int bars = 0;
            while (pos < end) {
                const int y_lo = (int)(pos % h), y_hi = (int)std::min<long long>(h, y_lo + (end - pos));
                bars += 1 + fb_piece_steps(y_lo, y_hi);
                pos += y_hi - y_lo;
            }
            worst = std::max(worst, bars);
        }
        if (G.nbarriers == 0 || worst < G.nbarriers) { G.nbarriers = worst; G.rows_per_slot = rps; G.piece_cost = cost; G.nblocks = nblocks; }
    }
    last = G;
}

}  // namespace mmif

using namespace mmif;

static int fb_check_branch(const mmif_dense_chain* c, const float* img, float* const* dwdb, const char* which) {
    MMIF_REQUIRE(c != nullptr && c->g3 != nullptr && c->glow != nullptr && c->x != nullptr && img != nullptr && dwdb != nullptr, "dense_encoder_bwd: %s: NULL argument", which);
    for (int i = 0; i < 3; ++i) MMIF_REQUIRE(c->packed[i] != nullptr, "dense_encoder_bwd: %s: operand image %d is NULL", which, i);
    for (int i = 0; i < 8; i += 2) MMIF_REQUIRE(dwdb[i] != nullptr, "dense_encoder_bwd: %s: dW%d is NULL", which, i / 2);
    if (int rc = validate_tensor(c->g3, "g3")) return rc;
    if (int rc = validate_tensor(c->glow, "glow")) return rc;
    if (int rc = validate_tensor(c->x, "x")) return rc;
    const mmif_tensor *a = c->g3, *b = c->glow, *x = c->x;
    MMIF_REQUIRE(a->dtype == MMIF_BF16 && b->dtype == MMIF_BF16 && x->dtype == MMIF_BF16, "dense_encoder_bwd: %s: bf16 tensors expected", which);
    MMIF_REQUIRE(a->cb == 2 && b->cb == 6 && x->cb == 6, "dense_encoder_bwd: %s: views of 2 / 6 / 6 channel blocks expected", which);
    MMIF_REQUIRE(x->halo == 0 && (a->halo == 0 || (a->flags & MMIF_T_FOLDED)) && (b->halo == 0 || (b->flags & MMIF_T_FOLDED)),
                 "dense_encoder_bwd: %s: x halo 0, gradients halo 0 or folded", which);
    MMIF_REQUIRE(a->n == x->n && b->n == x->n && a->h == x->h && b->h == x->h && a->w == x->w && b->w == x->w, "dense_encoder_bwd: %s: shape mismatch", which);
    MMIF_REQUIRE(x->h >= 4 && x->w >= 4, "dense_encoder_bwd: needs h, w >= 4 (rows / columns 1 and h-2 / w-2 are distinct fold targets)");
    for (const mmif_tensor* t : {a, b, x})
        MMIF_REQUIRE((long long)t->cb_total * (t->h + 2 * t->halo) * (t->w + 2 * t->halo) * 16 < (1ll << 31),
                     "dense_encoder_bwd: %s: one image of every allocation must stay below 2 GiB (32-bit lane offsets, bit 31 = masked)", which);
    return MMIF_OK;
}

// does every allocation the call would touch stay within the kernel's 32-bit lane offsets?  (cb_total channel blocks, halo: of the LARGEST of
// g3 / glow / x) -- the engine asks before it takes the fused path and falls back to the chain + weight-gradient launches otherwise
extern "C" int32_t mmif_dense_encoder_bwd_fits(int32_t cb_total, int32_t h, int32_t w, int32_t halo) {
    return h >= 4 && w >= 4 && (long long)cb_total * (h + 2 * halo) * (w + 2 * halo) * 16 < (1ll << 31) ? 1 : 0;
}

extern "C" size_t mmif_dense_encoder_bwd_workspace(void) { return (size_t)EW_MAXG * EW_PER * sizeof(float); }

// dwdb_x = {dW0, db0, dW1, db1, dW2, db2, dW3, db3} of the branch's encoder (db may be NULL); accumulate_x: onto what the pointers hold
// (the second branch of a shared encoder accumulates onto the first's).  chain_x->out is not used (nothing is written but the gradients).
extern "C" int mmif_dense_encoder_bwd(const mmif_dense_chain* chain_a, const float* img_a, float* const* dwdb_a, int32_t accumulate_a,
                                      const mmif_dense_chain* chain_b, const float* img_b, float* const* dwdb_b, int32_t accumulate_b,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = fb_check_branch(chain_a, img_a, dwdb_a, "branch a")) return rc;
    const int nb = chain_b != nullptr ? 2 : 1;
    if (nb == 2) {
        if (int rc = fb_check_branch(chain_b, img_b, dwdb_b, "branch b")) return rc;
        MMIF_REQUIRE(chain_a->x->n == chain_b->x->n && chain_a->x->h == chain_b->x->h && chain_a->x->w == chain_b->x->w, "dense_encoder_bwd: the two branches differ in shape");
    }
    if (workspace == nullptr || workspace_bytes < mmif_dense_encoder_bwd_workspace()) {
        set_error("dense_encoder_bwd: workspace too small");
        return MMIF_EWORKSPACE;
    }
    BwdArgs A;
    memset(&A, 0, sizeof(A));
    A.n = chain_a->x->n; A.h = chain_a->x->h; A.w = chain_a->x->w;
    FbGeo geo;
    fb_geometry(A.n, A.h, A.w, nb, geo);
    A.nstrips = geo.nstrips; A.rows_per_slot = geo.rows_per_slot; A.piece_cost = geo.piece_cost; A.nbarriers = geo.nbarriers;
    const int G = geo.nblocks;
    MMIF_REQUIRE(nb * G <= EW_MAXG, "dense_encoder_bwd: too many blocks for the partial-sum workspace");
    for (int b = 0; b < nb; ++b) {
        const mmif_dense_chain* c = b ? chain_b : chain_a;
        BwdBranch& Bb = A.br[b];
        Bb.g3 = make_tv(c->g3); Bb.glow = make_tv(c->glow); Bb.x = make_tv(c->x);
        for (int i = 0; i < 3; ++i) Bb.wpk[i] = (const uint4*)c->packed[i];
        Bb.img = b ? img_b : img_a;
        Bb.partial = (float*)workspace + (size_t)b * G * EW_PER;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(enc_bwd_fused_kernel, dim3(G, nb), dim3(FB_WAVES * 64), 0, st, A);
    if (int rc = check_launch("dense_encoder_bwd")) return rc;
    EwDst D[2];
    for (int b = 0; b < nb; ++b) {
        float* const* d = b ? dwdb_b : dwdb_a;
        D[b].dw0 = d[0]; D[b].db0 = d[1];
        D[b].dw[0] = d[2]; D[b].db[0] = d[3]; D[b].dw[1] = d[4]; D[b].db[1] = d[5]; D[b].dw[2] = d[6]; D[b].db[2] = d[7];
    }
    if (nb == 2) return enc_wgrad_reduce_pair_launch(A.br[0].partial, D[0], accumulate_a, A.br[1].partial, D[1], accumulate_b, G, st);
    if (int rc = enc_wgrad_reduce_launch(A.br[0].partial, D[0], G, accumulate_a, st)) return rc;
    return MMIF_OK;
}
