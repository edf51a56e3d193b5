// Streaming DenseBlock encoder, forward (bf16 MFMA): ConvLayer(1 -> 16) + DenseBlock(16, 16, 3 convs) of the PFNet / DenseFuse
// family (reference core/model.py:73-80, core/block.py:137-151) as ONE kernel.
//
// Layer by layer the encoder moves 160 channel planes per branch through HBM (write 16, read 16 + write 16, read 32 + write 16,
// read 48 + write 16) for 64 planes of output, and each Cout = 16 launch sits at ~50 % of the HBM roofline because a 16-channel
// tile gives a wave almost no MFMA work to hide its loads under.  Here the four layers run as a LINE-BUFFER PIPELINE that walks
// down the image: one wave owns a strip of 32 columns and keeps the last 5 / 4 / 3 rows of x0 / x1 / x2 (16 channels each) in a
// private 12 KB LDS ring; per step it produces
//     x0 row r+3 (VALU, fp32 FMAs on the fp32 image)  ->  x1 row r+2  ->  x2 row r+1  ->  x3 row r      (bf16 MFMA, fp32 accumulate)
// each from the three ring rows above it, and stores every finished row once.  HBM traffic = the image + the 64 output planes.
//   * waves are AUTONOMOUS: no block barrier after the weights are resident (28 KB of packed operand images per block), the only
//     synchronisation is a wave's own LDS write -> read order;
//   * reflect padding is index arithmetic: row reflection on the (wave-uniform) ring slot, column reflection in the per-lane
//     column offsets, so image borders cost nothing;
//   * a strip computes 32 columns at every layer and keeps the 26 (29 at an image edge) whose 3-layer receptive field lies inside
//     the strip; a row segment warms the pipeline up with 6 extra rows.  Strips x segments x images x branches = one wave each.
// The k-group order, the operand images (mmif_pack_weights) and the bias / ReLU / rounding points are those of the layer-wise
// kernels (conv_image.hip, conv_mfma.hip), so the results are bit-identical to running the four layers one by one
// (tests/test_gpu_enc_stream.py).
#include "enc_stream.hpp"
#include <stdlib.h>

namespace mmif {

typedef __attribute__((ext_vector_type(8))) __bf16 es_bf16x8;
typedef __attribute__((ext_vector_type(4))) float es_f32x4;

constexpr int ES_W = 32;                              // strip width = granules per ring row (512 B)
constexpr int ES_R0 = 5, ES_R1 = 4, ES_R2 = 3;        // ring rows of x0 / x1 / x2 per channel-block half
constexpr int ES_B0 = 0, ES_B1 = 2 * ES_R0, ES_B2 = ES_B1 + 2 * ES_R1;
constexpr int ES_SLOTS = ES_B2 + 2 * ES_R2;           // 24 row slots
constexpr int ES_RING_BYTES = ES_SLOTS * ES_W * 16;   // 12288
constexpr int ES_WAVES = 4;
#ifndef ES_ABL
#define ES_ABL 0   // timing ablations (diagnostic builds only, -DES_ABL=n; results are WRONG when non-zero): 1 no weight-fragment reads,
#endif             // 2 no input-fragment reads, 4 no global stores, 8 no first-layer FMAs (tools/bench_enc.py)
constexpr int ES_AHEAD = 4;                          // K steps of operand fetches in flight ahead of the MFMAs
constexpr int ES_KEEP = ES_W - 6;                     // 26 output columns per interior strip
constexpr int ES_WPLANES = ES_P1 + ES_P2 + ES_P3;     // 112 planes = 28672 B

template <int N> struct ESI { static constexpr int value = N; };
__device__ inline int es_tap(int i) { return (i % 3) * 3 + i / 3; }   // visit order of the 3x3 taps (conv_mfma.hip visit_tap)

__global__ __launch_bounds__(ES_WAVES * 64, 2) void enc_stream_fwd_kernel(EncArgs A) {
    __shared__ __attribute__((aligned(16))) uint4 s_w[ES_WPLANES * 16];
    __shared__ __attribute__((aligned(16))) char s_ring[ES_WAVES][ES_RING_BYTES];
    __shared__ __attribute__((aligned(16))) float s_bias[3][16];
    const EncBranch& B = A.br[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- resident operand images + biases (the only block-wide step)
    for (int e = tid; e < ES_P1 * 16; e += ES_WAVES * 64) s_w[e] = B.wpk[0][e];
    for (int e = tid; e < ES_P2 * 16; e += ES_WAVES * 64) s_w[ES_P1 * 16 + e] = B.wpk[1][e];
    for (int e = tid; e < ES_P3 * 16; e += ES_WAVES * 64) s_w[(ES_P1 + ES_P2) * 16 + e] = B.wpk[2][e];
    if (tid < 48) s_bias[tid >> 4][tid & 15] = B.bias[tid >> 4] != nullptr ? B.bias[tid >> 4][tid & 15] : 0.f;
    __syncthreads();
    const int item = blockIdx.x * ES_WAVES + wave;
    if (item >= A.items) return;
    const int strip = item % A.nstrips;
    const int seg = (item / A.nstrips) % A.nseg;
    const int in_ = item / (A.nstrips * A.nseg);
    const int H = A.h, W = A.w;
    const int y_lo = seg * A.seg_rows, y_hi = min(H, y_lo + A.seg_rows);
    if (y_lo >= y_hi) return;
    // strip geometry: region [r0, r0 + 32); kept output columns [o_lo, o_hi) = those whose 3-layer receptive field lies inside the
    // region (26 per interior strip, 29 at an image edge).  (Strips of 24 aligned columns -- whole 128-byte lines per stored run --
    // were measured slower: 11 instead of 10 strips per 256 columns and no gain on the store side)
    const int r0 = W > ES_W ? min(ES_KEEP * strip, W - ES_W) : 0;
    const int o_hi = (r0 + ES_W >= W) ? W : r0 + ES_W - 3;
    int o_lo = 0;
    if (strip > 0) {
        const int rp = min(ES_KEEP * (strip - 1), W - ES_W);
        o_lo = max(r0 + 3, rp + ES_W - 3);
    }
    char* ring = s_ring[wave];
    const int j = lane & 15, g = lane >> 4;

    // ---- lane constants
    // MFMA operand side: column offsets (bytes inside a ring row) of fragment f, tap column v, with the image's column reflection
    int colB[2][3];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            int c = reflect_idx(r0 + 16 * f + j + v - 1, W);
            c = min(max(c, 0), W - 1) - r0;
            colB[f][v] = min(max(c, 0), ES_W - 1) * 16;
        }
    const int a_lane = g * 256 + j * 16;                 // A operand: k-group plane g of a step, row (output channel) j
    const int hsel = g >> 1, cbh = g & 1;                // ncb = 2 chunks: lane group = (tap half, channel-block half)
    // epilogue side (after v_permlane16_swap): this lane holds the granule (pixel 16 (g & 1) + j, channel-block half g >> 1)
    const int px_e = 16 * (g & 1) + j, cb_e = g >> 1;
    const int x_e = r0 + px_e;
    const bool col_ok_e = x_e >= o_lo && x_e < o_hi;
    // first layer (VALU): lane = (pixel lane & 31, channel-block half lane >> 5); its 8 channels' weights live in registers
    const int px_a = lane & 31, cb_a = lane >> 5;
    const int x_a = r0 + px_a;
    const bool col_ok_a = x_a >= o_lo && x_a < o_hi;
    int cimg[3];
#pragma unroll
    for (int v = 0; v < 3; ++v) cimg[v] = min(max(reflect_idx(x_a + v - 1, W), 0), W - 1);
    float w0r[8][9], b0r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        b0r[i] = B.b0 != nullptr ? B.b0[cb_a * 8 + i] : 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) w0r[i][t] = B.w0[(cb_a * 8 + i) * 9 + t];
    }
    const float* img = B.img + (long long)in_ * H * W;
    // stores: wave-uniform 64-bit image base + a 32-bit lane offset (an image of the widest view is < 4 GB); geometry copied out of
    // the argument block once (left behind a reference, every store re-read it through the scalar cache and drained lgkmcnt)
    const unsigned out_plane_b = (unsigned)(B.out.plane * 16), out_row_b = (unsigned)B.out.ws * 16u;
    char* out_img = B.out.base + ((long long)in_ * B.out.img + (long long)B.out.cb_off * B.out.plane) * 16;

    auto rrow = [&](int y) { return min(max(reflect_idx(y, H), 0), H - 1); };
    auto ld_img_row = [&](int y, float (&dst)[3]) {
        const float* p = img + (long long)rrow(y) * W;
#pragma unroll
        for (int v = 0; v < 3; ++v) dst[v] = p[cimg[v]];
    };

    // ---- one row of a DenseBlock conv: L = 1, 2, 3 reads tensors x0 .. x(L-1) at rows R(y-1), y, R(y+1), writes x(L) row y
    auto conv_row = [&](auto Lc, int y) {
        constexpr int L = decltype(Lc)::value;
        constexpr int WBASE = (L == 1 ? 0 : (L == 2 ? ES_P1 : ES_P1 + ES_P2)) * 256;
        const char* wl = reinterpret_cast<const char*>(s_w) + WBASE + a_lane;
        int rr[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) rr[u] = rrow(y + u - 1);
        // K loop: 9 steps over the chunk of 4 channel blocks (x0, x1) -- every lane group takes its own block g of ONE tap per step --
        // and / or 5 steps over a chunk of 2 blocks (x0 for L = 1, x2 for L = 3): lane groups 0, 1 take tap 2s, groups 2, 3 tap 2s + 1
        // (k-groups 18, 19 are the zero planes of the operand image).  Same order as the layer-wise kernel (chunk by chunk).
        constexpr int NQ = L >= 2 ? 9 : 0, ND = (L == 1 || L == 3) ? 5 : 0, NS = NQ + ND;
        constexpr int TB = L == 1 ? ES_B0 : ES_B2, TR = L == 1 ? ES_R0 : ES_R2;
        constexpr int CH = L == 1 ? 0 : ES_P2 * 256;   // second chunk of the 48 -> 16 image starts after its 36 planes
        int rowQ[3], rowD[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int s0 = (ES_B0 + cbh * ES_R0 + rr[u] % ES_R0) * (ES_W * 16);
            const int s1 = (ES_B1 + cbh * ES_R1 + rr[u] % ES_R1) * (ES_W * 16);
            rowQ[u] = hsel ? s1 : s0;
            rowD[u] = (TB + cbh * TR + rr[u] % TR) * (ES_W * 16);
        }
        es_bf16x8 fa[NS], fb0[NS], fb1[NS];
        // operands of step s (compile-time s after unrolling): one weight fragment, two column fragments of the input rows
        auto fetch = [&](int s) {
            if (ES_ABL & 3) {   // diagnostics only
                if (!(ES_ABL & 1) || s == 0) fa[s] = *reinterpret_cast<const es_bf16x8*>(wl + (s % 5) * 1024); else fa[s] = fa[0];
                if (!(ES_ABL & 2) || s == 0) {
                    fb0[s] = *reinterpret_cast<const es_bf16x8*>(ring + rowQ[s % 3] + colB[0][s % 3]);
                    fb1[s] = *reinterpret_cast<const es_bf16x8*>(ring + rowQ[s % 3] + colB[1][s % 3]);
                } else { fb0[s] = fb0[0]; fb1[s] = fb1[0]; }
                return;
            }
            if (s < NQ) {
                const int tap = es_tap(s), u = tap / 3, v = tap % 3;
                fa[s] = *reinterpret_cast<const es_bf16x8*>(wl + tap * 4 * 256);
                fb0[s] = *reinterpret_cast<const es_bf16x8*>(ring + rowQ[u] + colB[0][v]);
                fb1[s] = *reinterpret_cast<const es_bf16x8*>(ring + rowQ[u] + colB[1][v]);
            } else {
                const int d = s - NQ;
                const int tA = es_tap(2 * d), uA = tA / 3, vA = tA % 3;
                const bool padB = 2 * d + 1 >= 9;
                const int tB = padB ? 0 : es_tap(2 * d + 1), uB = tB / 3, vB = tB % 3;
                // a_lane carries g * 256 = hsel * 512 + cbh * 256; this lane's plane is tap * 2 + cbh (18 + cbh when padded)
                const int pA = tA * 2 * 256, pB = padB ? 18 * 256 : tB * 2 * 256;
                fa[s] = *reinterpret_cast<const es_bf16x8*>(wl + CH + ((hsel ? pB : pA) - hsel * 512));
                const int bA0 = rowD[uA] + colB[0][vA], bB0 = rowD[uB] + colB[0][vB];
                const int bA1 = rowD[uA] + colB[1][vA], bB1 = rowD[uB] + colB[1][vB];
                fb0[s] = *reinterpret_cast<const es_bf16x8*>(ring + (hsel ? bB0 : bA0));
                fb1[s] = *reinterpret_cast<const es_bf16x8*>(ring + (hsel ? bB1 : bA1));
            }
        };
        es_f32x4 acc[2];
        acc[0] = (es_f32x4){0.f, 0.f, 0.f, 0.f};
        acc[1] = (es_f32x4){0.f, 0.f, 0.f, 0.f};
        // software pipeline: the operands of step s + ES_AHEAD are fetched under the MFMAs of step s (a wave has one SIMD
        // partner at most, so nothing else hides the ~100-cycle LDS latency of each fetch); order pinned with sched_group_barrier
#pragma unroll
        for (int s = 0; s < ES_AHEAD && s < NS; ++s) fetch(s);
        __builtin_amdgcn_sched_group_barrier(0x100, 3 * (ES_AHEAD < NS ? ES_AHEAD : NS), 0);   // the prologue's DS reads come first
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s + ES_AHEAD < NS) fetch(s + ES_AHEAD);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[s], fb0[s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[s], fb1[s], acc[1], 0, 0, 0);
            if (s + ES_AHEAD < NS) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);   // 3 DS reads
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                           // 2 MFMAs
        }
        // epilogue: pair the two column fragments -> one 16-byte granule per lane, + bias, ReLU, round once
        float c[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[0][r]), __float_as_uint(acc[1][r]), false, false);
            c[r] = __uint_as_float(sw[0]);
            c[4 + r] = __uint_as_float(sw[1]);
        }
        const float4 bv0 = *reinterpret_cast<const float4*>(&s_bias[L - 1][cb_e * 8]);
        const float4 bv1 = *reinterpret_cast<const float4*>(&s_bias[L - 1][cb_e * 8 + 4]);
        const float bv[8] = {bv0.x, bv0.y, bv0.z, bv0.w, bv1.x, bv1.y, bv1.z, bv1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = fmaxf(c[i] + bv[i], 0.f);
        const uint4 gr = make_uint4(pack_bf16x2(c[0], c[1]), pack_bf16x2(c[2], c[3]), pack_bf16x2(c[4], c[5]), pack_bf16x2(c[6], c[7]));
        if (L < 3) {
            constexpr int OB = L == 1 ? ES_B1 : ES_B2, OR = L == 1 ? ES_R1 : ES_R2;
            *reinterpret_cast<uint4*>(ring + ((OB + cb_e * OR + y % OR) * ES_W + px_e) * 16) = gr;
        }
        if (y >= y_lo && y < y_hi && col_ok_e && !(ES_ABL & 4))
            *reinterpret_cast<uint4*>(out_img + ((ES_ABL & 32) ? (unsigned)((y & 7) * 64 + lane) * 16u : (unsigned)(2 * L + cb_e) * out_plane_b + (unsigned)y * out_row_b + (unsigned)x_e * 16u)) = gr;   // (32: all stores into one L2-resident spot)
    };

    // ---- the pipeline
    const int a_lo = max(0, y_lo - 3), a_hi = min(H, y_hi + 3);
    const int b_lo = max(0, y_lo - 2), b_hi = min(H, y_hi + 2);
    const int c_lo = max(0, y_lo - 1), c_hi = min(H, y_hi + 1);
    // image rows: a 3-row window in registers plus two rows in flight.  The row a step shifts in was requested two steps earlier
    // into a FIXED register set (the row loop is unrolled by two so that no in-flight destination is ever copied), i.e. its HBM /
    // L2 latency is never on the wave's (serial) critical path.
    float win[3][3], fifo[2][3];
    ld_img_row(a_lo - 1, win[0]);
    ld_img_row(a_lo, win[1]);
    ld_img_row(a_lo + 1, win[2]);
    ld_img_row(a_lo + 2, fifo[0]);
    ld_img_row(a_lo + 3, fifo[1]);
    auto step = [&](int r, float (&slot)[3]) {
        const int ya = r + 3, yb = r + 2, yc = r + 1;
        if (ya < a_hi) {   // (ya >= a_lo by construction)
            float v8[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float acc0 = b0r[i];
#pragma unroll
                for (int t = 0; t < ((ES_ABL & 8) ? 1 : 9); ++t) acc0 = fmaf(win[t / 3][t % 3], w0r[i][t], acc0);
                v8[i] = A.relu0 ? fmaxf(acc0, 0.f) : acc0;
            }
            const uint4 gr = make_uint4(pack_bf16x2(v8[0], v8[1]), pack_bf16x2(v8[2], v8[3]), pack_bf16x2(v8[4], v8[5]), pack_bf16x2(v8[6], v8[7]));
            *reinterpret_cast<uint4*>(ring + ((ES_B0 + cb_a * ES_R0 + ya % ES_R0) * ES_W + px_a) * 16) = gr;
            if (ya >= y_lo && ya < y_hi && col_ok_a && !(ES_ABL & 4))
                *reinterpret_cast<uint4*>(out_img + ((ES_ABL & 32) ? (unsigned)((ya & 7) * 64 + lane) * 16u : (unsigned)cb_a * out_plane_b + (unsigned)ya * out_row_b + (unsigned)x_a * 16u)) = gr;
            // window of the next row: rows ya, ya + 1, R(ya + 2) (requested two steps ago); request row ya + 4 into the same slot
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                win[0][v] = win[1][v];
                win[1][v] = win[2][v];
                win[2][v] = slot[v];
            }
            if (!(ES_ABL & 16)) ld_img_row(ya + 4, slot);   // (16: no per-step image loads -> no vmcnt waits in the loop)
        }
        if (yb >= b_lo && yb < b_hi) conv_row(ESI<1>(), yb);
        if (yc >= c_lo && yc < c_hi) conv_row(ESI<2>(), yc);
        if (r >= y_lo && r < y_hi) conv_row(ESI<3>(), r);
    };
    for (int r = a_lo - 3; r < y_hi; r += 2) {
        step(r, fifo[0]);
        step(r + 1, fifo[1]);
    }
}


// items-per-launch heuristic: every (strip, segment, image, branch) is one wave; 2 blocks of 4 waves fit a CU (LDS), so 2048 waves
// are resident at once.  More segments fill the chip but each pays 6 warm-up rows: minimise rounds x rows per wave.
static void es_geometry(int n, int h, int w, int nb, int& nstrips, int& nseg, int& seg_rows) {
    nstrips = w > ES_W ? (w - 6 + ES_KEEP - 1) / ES_KEEP : 1;
    const long long slots = 256ll * 2 * ES_WAVES;
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

static int es_check_branch(const mmif_dense_encoder* e, const mmif_tensor* out, const char* which) {
    MMIF_REQUIRE(e != nullptr && e->img != nullptr && e->w0 != nullptr, "dense_encoder_fwd: %s: NULL image / first-layer weights", which);
    for (int i = 0; i < 3; ++i) MMIF_REQUIRE(e->packed[i] != nullptr, "dense_encoder_fwd: %s: packed operand image %d is NULL", which, i);
    if (int rc = validate_tensor(out, "out")) return rc;
    MMIF_REQUIRE(out->dtype == MMIF_BF16 && out->halo == 0 && out->cb == 8, "dense_encoder_fwd: %s: out must be a bf16 halo-0 view of 8 channel blocks", which);
    MMIF_REQUIRE(out->h >= 2 && out->w >= 2, "reflect padding needs h,w >= 2");
    MMIF_REQUIRE((long long)out->cb_total * out->h * out->w * 16 < (1ll << 32),
                 "dense_encoder_fwd: %s: one image of the output allocation must stay below 4 GiB (32-bit lane offsets)", which);
    return MMIF_OK;
}

// DenseFuse's encoder pass (round 6): both images through ONE shared encoder and their element-wise sum, one launch (csrc/enc_stream2.hip, dual form)
extern "C" int32_t mmif_dense_encoder_fwd_sum_supported(const mmif_dense_encoder* enc_a, const mmif_dense_encoder* enc_b, int32_t n, int32_t h, int32_t w) {
    EncArgs P;
    memset(&P, 0, sizeof(P));
    P.n = n; P.h = h; P.w = w;
    if (enc_a == nullptr || enc_b == nullptr || !enc_stream2_ok(P, 0)) return 0;
    if ((long long)16 * h * w * 16 >= (1ll << 31)) return 0;      // (a 128-channel feature buffer's image must stay below 2 GiB)
    if (enc_a->w0 != enc_b->w0 || enc_a->b0 != enc_b->b0) return 0;
    for (int i = 0; i < 3; ++i)
        if (enc_a->packed[i] != enc_b->packed[i] || enc_a->bias[i] != enc_b->bias[i]) return 0;
    return 1;
}
extern "C" int mmif_dense_encoder_fwd_sum(const mmif_dense_encoder* enc_a, const mmif_tensor* out_a, const mmif_dense_encoder* enc_b,
                                          const mmif_tensor* out_b, const mmif_tensor* sum, void* stream) {
    if (int rc = es_check_branch(enc_a, out_a, "branch a")) return rc;
    if (int rc = es_check_branch(enc_b, out_b, "branch b")) return rc;
    if (int rc = validate_tensor(sum, "sum")) return rc;
    MMIF_REQUIRE(out_a->n == out_b->n && out_a->h == out_b->h && out_a->w == out_b->w, "dense_encoder_fwd_sum: the two branches differ in shape");
    MMIF_REQUIRE(sum->dtype == MMIF_BF16 && sum->halo == 0 && sum->cb == 8 && sum->n == out_a->n && sum->h == out_a->h && sum->w == out_a->w,
                 "dense_encoder_fwd_sum: sum must be a bf16 halo-0 view of 8 channel blocks of the branches' shape");
    MMIF_REQUIRE(mmif_dense_encoder_fwd_sum_supported(enc_a, enc_b, out_a->n, out_a->h, out_a->w),
                 "dense_encoder_fwd_sum: the two branches must share ONE set of weights (and $MMIF_ENC_STREAM2 must not be 0)");
    EncArgs A;
    memset(&A, 0, sizeof(A));
    for (int b = 0; b < 2; ++b) {
        const mmif_dense_encoder* e = b ? enc_b : enc_a;
        EncBranch& B = A.br[b];
        B.img = e->img; B.w0 = e->w0; B.b0 = e->b0;
        for (int i = 0; i < 3; ++i) { B.wpk[i] = (const uint4*)e->packed[i]; B.bias[i] = e->bias[i]; }
        B.out = make_tv(b ? out_b : out_a);
    }
    A.sum = make_tv(sum);
    A.n = out_a->n; A.h = out_a->h; A.w = out_a->w;
    A.relu0 = 1;
    return enc_stream2_launch_dual(A, (hipStream_t)stream);
}

extern "C" int mmif_dense_encoder_fwd(const mmif_dense_encoder* enc_a, const mmif_tensor* out_a, const mmif_dense_encoder* enc_b,
                                      const mmif_tensor* out_b, void* stream) {
    if (int rc = es_check_branch(enc_a, out_a, "branch a")) return rc;
    const int nb = enc_b != nullptr ? 2 : 1;
    if (nb == 2) {
        if (int rc = es_check_branch(enc_b, out_b, "branch b")) return rc;
        MMIF_REQUIRE(out_a->n == out_b->n && out_a->h == out_b->h && out_a->w == out_b->w, "dense_encoder_fwd: the two branches differ in shape");
    }
    EncArgs A;
    memset(&A, 0, sizeof(A));
    for (int b = 0; b < nb; ++b) {
        const mmif_dense_encoder* e = b ? enc_b : enc_a;
        EncBranch& B = A.br[b];
        B.img = e->img; B.w0 = e->w0; B.b0 = e->b0;
        for (int i = 0; i < 3; ++i) { B.wpk[i] = (const uint4*)e->packed[i]; B.bias[i] = e->bias[i]; }
        B.out = make_tv(b ? out_b : out_a);
    }
    A.n = out_a->n; A.h = out_a->h; A.w = out_a->w;
    A.relu0 = 1;
    // round 5: the 64-column input-stationary kernel (csrc/enc_stream2.hip); $MMIF_ENC_STREAM2=0 / mmif_debug_set_enc_stream2(0): this file's
    if (enc_stream2_ok(A, nb)) return enc_stream2_launch(A, nb, (hipStream_t)stream);
    es_geometry(A.n, A.h, A.w, nb, A.nstrips, A.nseg, A.seg_rows);
    A.items = A.n * A.nseg * A.nstrips;
    hipLaunchKernelGGL(enc_stream_fwd_kernel, dim3(cdiv(A.items, ES_WAVES), nb), dim3(ES_WAVES * 64), 0, (hipStream_t)stream, A);
    return check_launch("dense_encoder_fwd");
}
