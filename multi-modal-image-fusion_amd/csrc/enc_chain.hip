// Streaming DenseBlock encoder, BACKWARD gradient chain (bf16 MFMA): the input gradients of DenseBlock(16, 16, 3 convs) of the PFNet /
// DenseFuse family (reference core/model.py:73-80, core/block.py:137-151; autograd of `x = cat(x, conv_i(x))`) as ONE kernel.
//
// With F = [x0 | x1 | x2 | x3] the block's output and G = [G0 | G1 | G2 | g3] its incoming gradient (g3 complete and ReLU-masked by the
// producer), the gradients the weight-gradient pass and the first layer need are
//     g2 = [x2 > 0] (G2 + A32 g3)      g1 = [x1 > 0] (G1 + A21 g2 + A31 g3)      g0 = [x0 > 0] (G0 + A10 g1 + A20 g2 + A30 g3)
// A_lk = adjoint of (reflect pad, 3x3 correlation with conv l's weights for its x_k input slice).  Round 2 ran them in GATHER form -- one
// dgrad launch per destination on a stacked virtual layer (mmif_pack_dense_chain) -- which moves 240 channel planes per branch through HBM
// for 112 of input and 48 of output, at the memory system's mixed read/write rate (5.3-5.9 TB/s: DESIGN.md section 4, item 15).  Here the
// three destinations run as the LINE-BUFFER PIPELINE of enc_stream.hip, mirrored: a wave owns a strip of 32 columns and walks down the
// image keeping the last 5 / 4 / 3 rows of g3 / g2 / g1 in a private LDS ring; per step it loads g3 row r+3 and produces
//     g2 row r+2  ->  g1 row r+1  ->  g0 row r          (bf16 MFMA on the virtual layers' operand images, fp32 accumulate, ONE rounding)
// HBM traffic = G (64 planes) + the masks x0..x2 (48) in, [g0 | g1 | g2 | g3] (64) out: 176 planes instead of 240 + three launches' tails.
//   * the adjoint of reflect padding is local: rows 1 and h-2 take one extra pass of the k-loop over row 0 / h-1 with the tap row that the
//     padded rows -1 / h would have seen (2 rows of the image run two passes); columns -1 and w are PART of the edge strips -- they hold
//     zeros in the ring, their lanes compute the padded-domain value like any other column, and one cross-lane add folds it onto
//     column 1 / w-2 before the lane's own value is dropped;
//   * out-of-image rows are a permanently zero ring slot, so the k-loop has no border logic;
//   * the result goes to a SEPARATE tensor (strips and row segments recompute their margins from the inputs: in place they would race).
// Same operand images, k-group order and rounding points as the gather-form launches: results agree within summation order / one bf16
// rounding (tests/test_gpu_enc_chain.py).
#include "common.hpp"
#include <stdlib.h>

namespace mmif {

typedef __attribute__((ext_vector_type(8))) __bf16 ec_bf16x8;
typedef __attribute__((ext_vector_type(4))) float ec_f32x4;

constexpr int EC_W = 32;                              // strip width = granules per ring row (512 B)
constexpr int EC_R0 = 5, EC_R1 = 4, EC_R2 = 3;        // ring rows of g3 / g2 / g1 per channel-block half
constexpr int EC_B0 = 0, EC_B1 = 2 * EC_R0, EC_B2 = EC_B1 + 2 * EC_R1;
constexpr int EC_SLOTS = EC_B2 + 2 * EC_R2;           // 24 row slots
constexpr int EC_ZERO = EC_SLOTS * EC_W * 16;         // byte offset of the zero row (slot 24)
constexpr int EC_RING_BYTES = (EC_SLOTS + 1) * EC_W * 16;   // 12800
constexpr int EC_WAVES = 4;
constexpr int EC_AHEAD = 4;                           // K steps of operand fetches in flight ahead of the MFMAs
constexpr int EC_KEEP = EC_W - 6;                     // 26 output columns per interior strip
constexpr int EC_P1 = 20, EC_P2 = 36, EC_P3 = 56;     // k-group planes of the virtual layers 16 -> 16, 32 -> 16, 48 -> 16
constexpr int EC_WPLANES = EC_P1 + EC_P2 + EC_P3;     // 112 planes = 28672 B

struct ChainBranch {
    TV g3;                     // 2-block view: gradient of x3 (complete, masked)
    TV glow;                   // 6-block view: G0 | G1 | G2 (what the decoder left for x0, x1, x2)
    TV x;                      // 6-block view: x0 | x1 | x2 (halo 0) -- the ReLU masks
    TV out;                    // 8-block view: g0 | g1 | g2 | g3
    const uint4* wpk[3];       // dgrad operand images of the virtual layers: [0] dst x0 (48 in), [1] dst x1 (32 in), [2] dst x2 (16 in)
};
struct ChainArgs {
    ChainBranch br[2];
    int n, h, w;
    int nstrips, nseg, seg_rows;
    int items;                 // per branch: n * nseg * nstrips
    int abl;                   // timing ablations ($MMIF_ABLATE ec=, results are garbage): 1 no operand requests after the first two, 2 no
};                             // output stores, 4 no k-loops

template <int N> struct ECI { static constexpr int value = N; };
__device__ inline int ec_tap(int i) { return (i % 3) * 3 + i / 3; }   // visit order of the 3x3 taps (conv_mfma.hip visit_tap)

struct ECPre {                 // one step's global operands, requested two steps ahead
    uint4 g3v;                 // g3 row r+3, lane = (pixel lane & 31, channel-block half lane >> 5)
    uint4 gk[3];               // accumulate operands G2 (row r+2), G1 (row r+1), G0 (row r), lane = epilogue mapping
    uint4 xk[3];               // mask operands x2, x1, x0 at the same places
};

__device__ inline void ec_touch(const uint4& v) { __asm__ volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w) : "memory"); }

__global__ __launch_bounds__(EC_WAVES * 64, 2) void enc_chain_bwd_kernel(ChainArgs A) {
    __shared__ __attribute__((aligned(16))) uint4 s_w[EC_WPLANES * 16];
    __shared__ __attribute__((aligned(16))) char s_ring[EC_WAVES][EC_RING_BYTES];
    const ChainBranch& B = A.br[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- resident operand images (order of use: dst x2, dst x1, dst x0) + cleared rings (the zero row must read 0)
    for (int e = tid; e < EC_P1 * 16; e += EC_WAVES * 64) s_w[e] = B.wpk[2][e];
    for (int e = tid; e < EC_P2 * 16; e += EC_WAVES * 64) s_w[EC_P1 * 16 + e] = B.wpk[1][e];
    for (int e = tid; e < EC_P3 * 16; e += EC_WAVES * 64) s_w[(EC_P1 + EC_P2) * 16 + e] = B.wpk[0][e];
    for (int e = tid; e < EC_WAVES * EC_RING_BYTES / 16; e += EC_WAVES * 64) reinterpret_cast<uint4*>(&s_ring[0][0])[e] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const int item = blockIdx.x * EC_WAVES + wave;
    if (item >= A.items) return;
    const int strip = item % A.nstrips;
    const int seg = (item / A.nstrips) % A.nseg;
    const int in_ = item / (A.nstrips * A.nseg);
    const int H = A.h, W = A.w;
    const int y_lo = seg * A.seg_rows, y_hi = min(H, y_lo + A.seg_rows);
    if (y_lo >= y_hi) return;
    // strip geometry: region [r0, r0 + 32) of image columns, r0 = -1 for the first strip (column -1 and column w belong to the edge strips);
    // kept output columns [o_lo, o_hi) = those whose 3-destination dependency cone lies inside the region or beyond the image edge
    auto strip_r0 = [&](int s) { return A.nstrips == 1 ? -1 : min(-1 + EC_KEEP * s, W - (EC_W - 1)); };
    // (a strip that does not hold column w cannot fold onto column w-2, and g0 at column c reaches g2 at c + 2: its kept columns end at
    // w - 5; the edge strip, whose region starts at w - 31, takes over from there)
    auto strip_hi = [&](int r) { return (r + EC_W >= W + 1) ? W : min(r + EC_W - 3, W - 4); };
    const int r0 = strip_r0(strip);
    const int o_hi = strip_hi(r0);
    const int o_lo = strip == 0 ? 0 : max(r0 + 3, strip_hi(strip_r0(strip - 1)));
    const bool edgeL = r0 < 0, edgeR = r0 + EC_W >= W + 1;       // wave-uniform: this strip holds column -1 / column w
    char* ring = s_ring[wave];
    const int j = lane & 15, g = lane >> 4;

    // ---- lane constants
    int colB[2][3];            // MFMA operand side: byte offset inside a ring row of fragment f, tap column v (clamped: margins are discarded)
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int v = 0; v < 3; ++v) colB[f][v] = min(max(16 * f + j + v - 1, 0), EC_W - 1) * 16;
    const int a_lane = g * 256 + j * 16;                 // A operand: k-group plane g of a step, row (output channel) j
    const int hsel = g >> 1, cbh = g & 1;                // 4-block chunk: lane group = (tensor half, channel-block half)
    // epilogue side (after v_permlane16_swap): this lane holds the granule (pixel 16 (g & 1) + j, channel-block half g >> 1)
    const int px_e = 16 * (g & 1) + j, cb_e = g >> 1;
    const int x_e = r0 + px_e;
    const bool in_e = x_e >= 0 && x_e < W;
    const bool keep_e = x_e >= o_lo && x_e < o_hi;
    const int xc_e = min(max(x_e, 0), W - 1);
    // column fold: the lane of column 1 takes the padded-domain value of column -1 (two pixels to its left), column w-2 that of column w
    const int srcL = (cb_e * 2 + (max(px_e - 2, 0) >> 4)) * 16 + (max(px_e - 2, 0) & 15);
    const int srcR = (cb_e * 2 + (min(px_e + 2, EC_W - 1) >> 4)) * 16 + (min(px_e + 2, EC_W - 1) & 15);
    const bool tgtL = edgeL && x_e == 1, tgtR = edgeR && x_e == W - 2;
    // g3 rows: lane = (pixel lane & 31, channel-block half lane >> 5)
    const int px_a = lane & 31, cb_a = lane >> 5;
    const int x_a = r0 + px_a;
    const bool in_a = x_a >= 0 && x_a < W;
    const bool keep_a = x_a >= o_lo && x_a < o_hi;
    const int xc_a = min(max(x_a, 0), W - 1);

    // wave-uniform 64-bit image bases + 32-bit lane offsets (an image of the widest view is < 4 GB: checked by the host)
    auto img_base = [&](const TV& t) { return t.base + ((long long)in_ * t.img + (long long)t.cb_off * t.plane) * 16; };
    const char* g3_img = img_base(B.g3);
    const char* gl_img = img_base(B.glow);
    const char* x_img = img_base(B.x);
    char* out_img = B.out.base + ((long long)in_ * B.out.img + (long long)B.out.cb_off * B.out.plane) * 16;
    const unsigned g3_plane = (unsigned)(B.g3.plane * 16), g3_row = (unsigned)B.g3.ws * 16u, g3_org = (unsigned)(B.g3.halo * (B.g3.ws + 1)) * 16u;
    const unsigned gl_plane = (unsigned)(B.glow.plane * 16), gl_row = (unsigned)B.glow.ws * 16u, gl_org = (unsigned)(B.glow.halo * (B.glow.ws + 1)) * 16u;
    const unsigned x_plane = (unsigned)(B.x.plane * 16), x_row = (unsigned)B.x.ws * 16u;
    const unsigned o_plane = (unsigned)(B.out.plane * 16), o_row = (unsigned)B.out.ws * 16u, o_org = (unsigned)(B.out.halo * (B.out.ws + 1)) * 16u;

    auto crow = [&](int y) { return (unsigned)min(max(y, 0), H - 1); };
    // request the global operands of step r (rows clamped into the image: what lies outside is never used)
    auto request = [&](int r, ECPre& S) {
        if ((A.abl & 1) && r > y_lo) return;
        S.g3v = *reinterpret_cast<const uint4*>(g3_img + (unsigned)cb_a * g3_plane + crow(r + 3) * g3_row + (unsigned)xc_a * 16u + g3_org);
#pragma unroll
        for (int k = 0; k < 3; ++k) {   // k = 0: dst x2 at row r + 2; 1: dst x1 at r + 1; 2: dst x0 at r
            const unsigned blk = (unsigned)(2 * (2 - k) + cb_e), yy = crow(r + 2 - k);
            S.gk[k] = *reinterpret_cast<const uint4*>(gl_img + blk * gl_plane + yy * gl_row + (unsigned)xc_e * 16u + gl_org);
            S.xk[k] = *reinterpret_cast<const uint4*>(x_img + blk * x_plane + yy * x_row + (unsigned)xc_e * 16u);
        }
    };

    // ---- one row of a destination: L = 1, 2, 3 <-> dst x2, x1, x0; reads the rings of g3 (A), g2 (B), g1 (C) at rows y-1, y, y+1 (zero
    // outside the image), adds G, masks, writes row y of its gradient (ring B / C for L = 1 / 2, and the output tensor)
    auto conv_row = [&](auto Lc, int y, const uint4& gq, const uint4& xq) {
        constexpr int L = decltype(Lc)::value;
        constexpr int WBASE = (L == 1 ? 0 : (L == 2 ? EC_P1 : EC_P1 + EC_P2)) * 256;
        const char* wl = reinterpret_cast<const char*>(s_w) + WBASE + a_lane;
        // K loop: 9 steps over the chunk of 4 channel blocks (L = 2: g2 | g3; L = 3: g1 | g2) -- every lane group takes its own block of ONE
        // tap per step -- and / or 5 steps over a chunk of 2 blocks (g3 for L = 1 and L = 3): groups 0, 1 take tap 2s, groups 2, 3 tap 2s + 1
        constexpr int NQ = L >= 2 ? 9 : 0, ND = (L == 1 || L == 3) ? 5 : 0, NS = NQ + ND;
        constexpr int CH = L == 1 ? 0 : EC_P2 * 256;   // second chunk of the 48 -> 16 image starts after its 36 planes
        ec_f32x4 acc[2];
        acc[0] = (ec_f32x4){0.f, 0.f, 0.f, 0.f};
        acc[1] = (ec_f32x4){0.f, 0.f, 0.f, 0.f};
        // pass 0: the row's own taps.  pass 1 (rows 1 and h-2 only): the adjoint of reflect padding -- padded row -1 (h) received tap row 2 (0)
        // times row 0 (h-1), and folds onto row 1 (h-2)
        const int npass = (A.abl & 4) ? 0 : ((y == 1 || y == H - 2) ? 2 : 1);
        for (int pass = 0; pass < npass; ++pass) {
            int rows[3];     // gradient row feeding tap row u, -1 = zero
            if (pass == 0) {
                rows[0] = y - 1;                      // (>= 0 or -1)
                rows[1] = y;
                rows[2] = y + 1 < H ? y + 1 : -1;
            } else if (y == 1) {
                rows[0] = -1; rows[1] = -1; rows[2] = 0;
            } else {
                rows[0] = H - 1; rows[1] = -1; rows[2] = -1;
            }
            int rowQ[3], rowD[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int rr = max(rows[u], 0);
                const int sA = (EC_B0 + cbh * EC_R0 + rr % EC_R0) * (EC_W * 16);
                const int sB = (EC_B1 + cbh * EC_R1 + rr % EC_R1) * (EC_W * 16);
                const int sC = (EC_B2 + cbh * EC_R2 + rr % EC_R2) * (EC_W * 16);
                const int q = L == 2 ? (hsel ? sA : sB) : (hsel ? sB : sC);
                rowQ[u] = rows[u] >= 0 ? q : EC_ZERO;
                rowD[u] = rows[u] >= 0 ? sA : EC_ZERO;
            }
            ec_bf16x8 fa[NS], fb0[NS], fb1[NS];
            auto fetch = [&](int s) {
                if (s < NQ) {
                    const int tap = ec_tap(s), u = tap / 3, v = tap % 3;
                    fa[s] = *reinterpret_cast<const ec_bf16x8*>(wl + tap * 4 * 256);
                    fb0[s] = *reinterpret_cast<const ec_bf16x8*>(ring + rowQ[u] + colB[0][v]);
                    fb1[s] = *reinterpret_cast<const ec_bf16x8*>(ring + rowQ[u] + colB[1][v]);
                } else {
                    const int d = s - NQ;
                    const int tA = ec_tap(2 * d), uA = tA / 3, vA = tA % 3;
                    const bool padB = 2 * d + 1 >= 9;
                    const int tB = padB ? 0 : ec_tap(2 * d + 1), uB = tB / 3, vB = tB % 3;
                    // a_lane carries g * 256 = hsel * 512 + cbh * 256; this lane's plane is tap * 2 + cbh (18 + cbh when padded)
                    const int pA = tA * 2 * 256, pB = padB ? 18 * 256 : tB * 2 * 256;
                    fa[s] = *reinterpret_cast<const ec_bf16x8*>(wl + CH + ((hsel ? pB : pA) - hsel * 512));
                    const int bA0 = rowD[uA] + colB[0][vA], bB0 = rowD[uB] + colB[0][vB];
                    const int bA1 = rowD[uA] + colB[1][vA], bB1 = rowD[uB] + colB[1][vB];
                    fb0[s] = *reinterpret_cast<const ec_bf16x8*>(ring + (hsel ? bB0 : bA0));
                    fb1[s] = *reinterpret_cast<const ec_bf16x8*>(ring + (hsel ? bB1 : bA1));
                }
            };
#pragma unroll
            for (int s = 0; s < EC_AHEAD && s < NS; ++s) fetch(s);
            __builtin_amdgcn_sched_group_barrier(0x100, 3 * (EC_AHEAD < NS ? EC_AHEAD : NS), 0);   // the prologue's DS reads come first
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (s + EC_AHEAD < NS) fetch(s + EC_AHEAD);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[s], fb0[s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[s], fb1[s], acc[1], 0, 0, 0);
                if (s + EC_AHEAD < NS) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);   // 3 DS reads
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                           // 2 MFMAs
            }
        }
        // epilogue: pair the two column fragments -> one granule per lane; fold the edge columns; + G; mask; round once
        float c[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[0][r]), __float_as_uint(acc[1][r]), false, false);
            c[r] = __uint_as_float(sw[0]);
            c[4 + r] = __uint_as_float(sw[1]);
        }
        if (edgeL || edgeR) {   // wave-uniform: the adjoint of reflect padding along x
            float fl[8], fr[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                fl[i] = __shfl(c[i], srcL, 64);
                fr[i] = __shfl(c[i], srcR, 64);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] += (tgtL ? fl[i] : 0.f) + (tgtR ? fr[i] : 0.f);
        }
        const uint32_t gw[4] = {gq.x, gq.y, gq.z, gq.w}, xw[4] = {xq.x, xq.y, xq.z, xq.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            c[2 * i] += __uint_as_float(gw[i] << 16);
            c[2 * i + 1] += __uint_as_float(gw[i] & 0xffff0000u);
            // bf16 > 0  <=>  sign bit clear and magnitude non-zero; columns outside the image hold zero
            const uint32_t lo = xw[i] & 0xffffu, hi = xw[i] >> 16;
            if (!in_e || !((lo & 0x8000u) == 0 && (lo & 0x7fffu) != 0)) c[2 * i] = 0.f;
            if (!in_e || !((hi & 0x8000u) == 0 && (hi & 0x7fffu) != 0)) c[2 * i + 1] = 0.f;
        }
        const uint4 gr = make_uint4(pack_bf16x2(c[0], c[1]), pack_bf16x2(c[2], c[3]), pack_bf16x2(c[4], c[5]), pack_bf16x2(c[6], c[7]));
        if (L < 3) {
            constexpr int OB = L == 1 ? EC_B1 : EC_B2, OR = L == 1 ? EC_R1 : EC_R2;
            *reinterpret_cast<uint4*>(ring + ((OB + cb_e * OR + y % OR) * EC_W + px_e) * 16) = gr;
        }
        return gr;     // (stored to the output tensor by the NEXT step, see the pipeline)
    };

    // ---- the pipeline
    const int a_lo = max(0, y_lo - 3), a_hi = min(H, y_hi + 3);
    const int b_lo = max(0, y_lo - 2), b_hi = min(H, y_hi + 2);
    const int c_lo = max(0, y_lo - 1), c_hi = min(H, y_hi + 1);
    // Global memory traffic of a step happens at ONE point, its top: loads and stores share one in-order counter on this hardware, so a
    // wait for prefetched operands is a wait for every store issued before it -- with the rows' stores inside the step (first version) each
    // of the four uses of prefetched data drained the store of the row just finished: 4.0 us per row step against the forward's 2.9.
    // Now a step (1) takes its operands (requested two steps ago: landed long since), (2) stores the rows the PREVIOUS step produced
    // (kept in 16 registers), (3) requests the operands of step r + 2, (4) computes -- nothing waits on anything younger than a step.
    ECPre S0, S1;
    const int r_first = a_lo - 3;
    request(r_first, S0);
    request(r_first + 1, S1);
    uint4 pend[4];                     // g3 row r+2 (copy), g2 row r+1, g1 row r, g0 row r-1 of the previous step r-1
#pragma unroll
    for (int i = 0; i < 4; ++i) pend[i] = make_uint4(0, 0, 0, 0);
    auto flush = [&](int rp) {         // rows produced by step rp: g3 copy at rp + 3, g2 at rp + 2, g1 at rp + 1, g0 at rp
        if (A.abl & 2) return;
        const int ya = rp + 3;
        if (ya >= a_lo && ya < a_hi && ya >= y_lo && ya < y_hi && keep_a)     // [g0 | g1 | g2 | g3] contiguous for the weight-gradient pass
            *reinterpret_cast<uint4*>(out_img + (unsigned)(6 + cb_a) * o_plane + (unsigned)ya * o_row + (unsigned)x_a * 16u + o_org) = pend[0];
#pragma unroll
        for (int k = 0; k < 3; ++k) {  // k = 0: g2 (blocks 4, 5) at row rp + 2 ... k = 2: g0 at row rp
            const int y = rp + 2 - k;
            if (y >= y_lo && y < y_hi && keep_e)
                *reinterpret_cast<uint4*>(out_img + (unsigned)(2 * (2 - k) + cb_e) * o_plane + (unsigned)y * o_row + (unsigned)x_e * 16u + o_org) = pend[1 + k];
        }
    };
    auto step = [&](int r, ECPre& S) {
        const ECPre cur = S;           // (1)  (the empty asm pins the wait HERE: left to the first use, it would sit behind the stores and
        ec_touch(cur.g3v);             //       requests below and become a wait for them)
#pragma unroll
        for (int k = 0; k < 3; ++k) { ec_touch(cur.gk[k]); ec_touch(cur.xk[k]); }
        flush(r - 1);                  // (2)
        request(r + 2, S);             // (3)
        const int ya = r + 3, yb = r + 2, yc = r + 1;
        if (ya >= a_lo && ya < a_hi) {
            pend[0] = in_a ? cur.g3v : make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(ring + ((EC_B0 + cb_a * EC_R0 + ya % EC_R0) * EC_W + px_a) * 16) = pend[0];
        }
        if (yb >= b_lo && yb < b_hi) pend[1] = conv_row(ECI<1>(), yb, cur.gk[0], cur.xk[0]);
        if (yc >= c_lo && yc < c_hi) pend[2] = conv_row(ECI<2>(), yc, cur.gk[1], cur.xk[1]);
        if (r >= y_lo && r < y_hi) pend[3] = conv_row(ECI<3>(), r, cur.gk[2], cur.xk[2]);
    };
    int r = r_first;
    for (; r < y_hi; r += 2) {
        step(r, S0);
        step(r + 1, S1);
    }
    flush(r - 1);
}

// items-per-launch heuristic (as enc_stream.hip): every (strip, segment, image, branch) is one wave; 2 blocks of 4 waves fit a CU
static void ec_geometry(int n, int h, int w, int nb, int& nstrips, int& nseg, int& seg_rows) {
    nstrips = w <= EC_W - 2 ? 1 : (w - (EC_W - 2) + EC_KEEP - 1) / EC_KEEP + 1;
    const long long slots = 256ll * 2 * EC_WAVES;
    long long best = -1;
    nseg = 1;
    for (int k = 1; k <= (h + 7) / 8; ++k) {
        const int rows = (h + k - 1) / k;
        const long long items = (long long)nb * n * nstrips * k;
        const long long cost = ((items + slots - 1) / slots) * (rows + 6);
        if (best < 0 || cost < best) { best = cost; nseg = k; }
    }
    seg_rows = (h + nseg - 1) / nseg;
    nseg = (h + seg_rows - 1) / seg_rows;   // drop empty trailing segments
}

}  // namespace mmif

using namespace mmif;

static int ec_check_branch(const mmif_dense_chain* c, const char* which) {
    MMIF_REQUIRE(c != nullptr && c->g3 != nullptr && c->glow != nullptr && c->x != nullptr && c->out != nullptr, "dense_encoder_chain: %s: NULL tensor", which);
    for (int i = 0; i < 3; ++i) MMIF_REQUIRE(c->packed[i] != nullptr, "dense_encoder_chain: %s: operand image %d is NULL", which, i);
    if (int rc = validate_tensor(c->g3, "g3")) return rc;
    if (int rc = validate_tensor(c->glow, "glow")) return rc;
    if (int rc = validate_tensor(c->x, "x")) return rc;
    if (int rc = validate_tensor(c->out, "out")) return rc;
    const mmif_tensor *a = c->g3, *b = c->glow, *x = c->x, *o = c->out;
    MMIF_REQUIRE(a->dtype == MMIF_BF16 && b->dtype == MMIF_BF16 && x->dtype == MMIF_BF16 && o->dtype == MMIF_BF16, "dense_encoder_chain: %s: bf16 tensors expected", which);
    MMIF_REQUIRE(a->cb == 2 && b->cb == 6 && x->cb == 6 && o->cb == 8, "dense_encoder_chain: %s: views of 2 / 6 / 6 / 8 channel blocks expected", which);
    MMIF_REQUIRE(x->halo == 0 && (a->halo == 0 || (a->flags & MMIF_T_FOLDED)) && (b->halo == 0 || (b->flags & MMIF_T_FOLDED)),
                 "dense_encoder_chain: %s: x halo 0, gradients halo 0 or folded", which);
    MMIF_REQUIRE(a->n == o->n && b->n == o->n && x->n == o->n && a->h == o->h && b->h == o->h && x->h == o->h && a->w == o->w && b->w == o->w && x->w == o->w,
                 "dense_encoder_chain: %s: shape mismatch", which);
    MMIF_REQUIRE(o->h >= 4 && o->w >= 4, "dense_encoder_chain: needs h, w >= 4 (rows / columns 1 and h-2 / w-2 are distinct fold targets)");
    for (const mmif_tensor* t : {a, b, x, o})
        MMIF_REQUIRE((long long)t->cb_total * (t->h + 2 * t->halo) * (t->w + 2 * t->halo) * 16 < (1ll << 32),
                     "dense_encoder_chain: %s: one image of every allocation must stay below 4 GiB (32-bit lane offsets)", which);
    // the output must not alias an input plane range (strips recompute their margins from the inputs)
    auto overlaps = [&](const mmif_tensor* t) {
        if (t->data != o->data) return false;
        return t->cb_off < o->cb_off + o->cb && o->cb_off < t->cb_off + t->cb;
    };
    MMIF_REQUIRE(!overlaps(a) && !overlaps(b), "dense_encoder_chain: %s: out overlaps an input gradient view (the kernel is not in-place)", which);
    return MMIF_OK;
}

extern "C" int mmif_dense_encoder_chain(const mmif_dense_chain* ca, const mmif_dense_chain* cb, void* stream) {
    if (int rc = ec_check_branch(ca, "branch a")) return rc;
    const int nb = cb != nullptr ? 2 : 1;
    if (nb == 2) {
        if (int rc = ec_check_branch(cb, "branch b")) return rc;
        MMIF_REQUIRE(ca->out->n == cb->out->n && ca->out->h == cb->out->h && ca->out->w == cb->out->w, "dense_encoder_chain: the two branches differ in shape");
    }
    ChainArgs A;
    memset(&A, 0, sizeof(A));
    for (int b = 0; b < nb; ++b) {
        const mmif_dense_chain* c = b ? cb : ca;
        ChainBranch& B = A.br[b];
        B.g3 = make_tv(c->g3); B.glow = make_tv(c->glow); B.x = make_tv(c->x); B.out = make_tv(c->out);
        for (int i = 0; i < 3; ++i) B.wpk[i] = (const uint4*)c->packed[i];
    }
    A.n = ca->out->n; A.h = ca->out->h; A.w = ca->out->w;
    ec_geometry(A.n, A.h, A.w, nb, A.nstrips, A.nseg, A.seg_rows);
    A.items = A.n * A.nseg * A.nstrips;
    static int abl = -1;
    if (abl < 0) abl = ablate_env("ec");
    A.abl = abl;
    hipLaunchKernelGGL(enc_chain_bwd_kernel, dim3(cdiv(A.items, EC_WAVES), nb), dim3(EC_WAVES * 64), 0, (hipStream_t)stream, A);
    return check_launch("dense_encoder_chain");
}
