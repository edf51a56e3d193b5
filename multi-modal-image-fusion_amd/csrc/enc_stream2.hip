// Streaming DenseBlock encoder, forward (bf16 MFMA), second generation (round 5): ConvLayer(1 -> 16) + DenseBlock(16, 16, 3 convs) of the
// PFNet / DenseFuse family (reference core/model.py:73-80, core/block.py:137-151) as ONE kernel, like enc_stream.hip, rebuilt around what
// that kernel's counters said (~760 instructions per wave and 32-pixel row step, 1.5 LDS operand
// reads per MFMA, the four layers of a row step a serial chain):
//
//   * INPUT-STATIONARY accumulation.  With 16 output channels a layer has ONE M-tile, so in the output-stationary order every MFMA
//     needs its own B fragment (16 pixels x 32 k).  Here a wave reads the B fragments of input row R once and multiplies them with the
//     weights of all THREE tap rows:  out[R+1] = W(u=0) x[R] (a fresh accumulator that starts from the bias),  out[R] += W(u=1) x[R],
//     out[R-1] += W(u=2) x[R] (which completes row R-1) -- three accumulator rows per layer rotate through the row loop (unrolled by
//     three).  One B read + three A reads feed twelve MFMAs: 0.58 LDS reads per MFMA instead of 1.5.  K is ordered tap-row major, so
//     the 16 -> 16 layer pads 48 to 64 (two k-steps) and the 48 -> 16 layer 144 to 160 (five): 30 MFMAs per 16-pixel tile and row
//     instead of 28.
//   * 64-column strips, one wave per SIMD (512 registers: 3 layers x 3 rows x 4 tiles of accumulators = 144).  A strip keeps 58 of
//     its 64 columns (60 at an image edge) instead of 26 / 29 of 32, and the per-row bookkeeping is paid once per 64 pixels.
//   * SKEWED pipeline: step s produces x0 row s (fp32 FMAs on the image), feeds x0 row s-1 to layer 1 (completing x1 row s-2), rows
//     s-3 of [x0 | x1] to layer 2 (x2 row s-4) and rows s-5 of [x0 | x1 | x2] to layer 3 (x3 row s-6): everything a step READS from
//     the wave's LDS ring was written in an earlier step, so the four stages of a step are independent instruction streams for the
//     scheduler instead of a chain of LDS round trips.  Ring: 8 / 4 / 2 row slots of x0 / x1 / x2 (28 KB per wave).
//   * Reflect padding.  Rows: input row 1 also feeds out row 0 through W(u=0), input row h-2 also out row h-1 through W(u=2) (extra
//     MFMAs in the few steps that touch those rows; the row loop has a branch-free fast body for all others).  Columns: an edge strip
//     carries the padded column (-1 or w) as a GHOST pixel of its ring rows -- x0 computes it from the reflected image columns, x1 / x2
//     copy it from column 1 / w-2 after their epilogue -- so every operand read is base + immediate with no per-lane column tables.
//   * Epilogue on packed pairs: v_cvt_pk_bf16_f32, ReLU as v_pk_max_i16 against 0, two v_permlane16_swap per 32 pixels (the
//     conv_dma_kernel recipe), the bias rides in as the C operand of the fresh accumulator's first MFMA.
//
// Rounding points are the layer-wise kernels' (every x_k rounded to bf16 once, fp32 accumulation), the accumulation ORDER is not, so
// the results are no longer bit-identical to them: tests/test_gpu_enc_stream.py holds this kernel to the fp64 definition of every stage
// on the kernel's own bf16 inputs at one bf16 rounding (as tests/test_gpu_enc_chain.py does for the backward chain).  x0 (fp32 FMAs in
// the old order) stays bit-identical.  $MMIF_ENC_STREAM2=0 selects the round-2 kernel.
#include "enc_stream.hpp"
#include <stdlib.h>
#pragma clang diagnostic ignored "-Winline-asm"   // (the LDS-DMA asm names m0 in its clobber list: "reserved register")

namespace mmif {

typedef __attribute__((ext_vector_type(8))) __bf16 e2_bf16x8;
typedef __attribute__((ext_vector_type(4))) float e2_f32x4;
typedef float e2_f32x2 __attribute__((ext_vector_type(2)));
typedef short e2_i16x2 __attribute__((ext_vector_type(2)));
typedef unsigned e2_u32x4 __attribute__((ext_vector_type(4)));
#define E2_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define E2_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int E2_S0 = 8, E2_S1 = 4, E2_S2 = 2; // ring slots of x0 / x1 / x2 (powers of two: slot = row & (S - 1))
constexpr int E2_NFRAG = 24;                   // A fragments in LDS: (3 + 5 k-steps of the 32->16 / 48->16 convs) x 3 tap rows, 1 KB each
constexpr int E2_WBYTES = E2_NFRAG * 1024;     // (the 16->16 conv's six fragments live in registers)
// Geometry of the two instantiations: NT = 4 column tiles (64-pixel strips, four waves per CU, one per SIMD) and NT = 2 (32-pixel strips,
// eight waves per CU, two per SIMD).  Same work per SIMD either way; with two waves a SIMD keeps computing while one of them sits in a
// store that HBM back-pressure holds at issue -- measured on the NT = 4 build: compute 133 us and stores 105 us did not overlap at all
// (profiles/r05_ubench_enc_stream2_ablation.txt).
// DUAL (round 6, NT = 4 only): the wave's 64 ring pixels are TWO 32-pixel strips -- the same columns of the two images of a pair, column
// tiles 0, 1 = image a, tiles 2, 3 = image b -- run through ONE shared set of weights (DenseFuse: `core/model.py:165-186`, one encoder for
// both images).  Every tile is its own MFMA stream, the two halves only meet in the operand reads next to pixel 31 | 32, which land in columns
// a 32-pixel strip discards anyway; the epilogue then holds the granules of BOTH images of a pixel in one lane and emits their sum
// f1 + f2 (`core/fusion.py:21-29`, element_fusion 'sum') next to the two branches' own rows: mmif_fuse_elem_fwd's pass over 192 channel
// planes disappears.
template <int NT, bool DUAL = false> struct E2G {
    static_assert(!DUAL || NT == 4, "the dual form is the four-tile instantiation");
    static constexpr int W = DUAL ? 32 : 16 * NT;  // strip width in pixels (of one image)
    static constexpr int KEEP = W - 6;             // columns an interior strip keeps (3 lost per side over the three 3x3 layers)
    static constexpr int EDGE = W - 4;             // ... the first / last strip of an image (one ghost pixel + 3 lost on the inner side)
    static constexpr int CBS = 256 * NT;           // bytes of one channel block of a ring row (16 NT pixels x 16 B)
    static constexpr int ROW = 2 * CBS;            // one ring slot: [cb 0][cb 1]
    static constexpr int X0 = 0, X1 = E2_S0 * ROW, X2 = X1 + E2_S1 * ROW, RING = X2 + E2_S2 * ROW;
    static constexpr int WAVES = NT == 4 ? 4 : 8;
    static constexpr int NP = NT / 2;              // tile pairs of the epilogue
    static constexpr int NST = NP + (DUAL ? 1 : 0);   // stores of one layer's epilogue (DUAL: + the sum's)
    static constexpr int OPS = 4 * NST + (NT == 4 ? 2 : 1);    // vector-memory operations of a branch-free step (stores + image DMAs)
    static constexpr int AHEAD = 2 + (64 + OPS - 1) / OPS;     // image rows are requested this many steps ahead (> 63 operations): 9 / 15 (dual: 7)
    static constexpr int IMGS = 16;                            // slots of the image ring (> AHEAD)
    static constexpr int IMGROW = NT == 4 ? 288 : 160;         // bytes per slot (W + 2 used entries per image, + the pad group's)
    static constexpr int LDS = E2_WBYTES + WAVES * RING + 64;  // (+ 64: operand reads run up to two granules past a ring row)
    static constexpr int LDS_IMG = WAVES * IMGS * IMGROW;      // the image ring is an LDS variable of its own, see the kernel
};

template <int N> struct E2I { static constexpr int value = N; };
#ifndef E2_STORE_AUX
#define E2_STORE_AUX 1      // cache-policy bits of the output stores: sc0 (1) measured 196 us against 205 (0), nt (2) 231, sc1 (16) 225, 3 / 17 / 18 / 19 223-233
                             // (scope bits only ever strengthen coherence; the kernel boundary publishes the rows either way)
#endif
#ifndef E2_ABL
#define E2_ABL 0   // timing ablations (diagnostic builds only, tools/build_ab_enc2.sh; results are WRONG when non-zero): 1 no global stores (32: the stores go to one cache-resident 16 KiB per block instead),
#endif             // 2 no bf16 MFMAs, 4 no LDS operand reads after a step's first, 8 no epilogues at all, 16 no first-layer MFMAs / image loads

template <int NT, bool DUAL = false>
__global__ __launch_bounds__((E2G<NT, DUAL>::WAVES * 64), 1) void enc_stream2_fwd_kernel(EncArgs A) {
    using G = E2G<NT, DUAL>;
    constexpr int E2_X0 = G::X0, E2_X1 = G::X1, E2_X2 = G::X2, E2_RING = G::RING, E2_WAVES = G::WAVES, E2_W = G::W, E2_KEEP = G::KEEP;
    constexpr int E2_IMGS = G::IMGS, E2_IMGROW = G::IMGROW, CBS = G::CBS, ROWB = G::ROW, NP = G::NP;
    __shared__ __attribute__((aligned(16))) char smem[G::LDS];
    // The image ring is written by LDS-DMA.  The compiler orders every C++ read of an LDS object a pending DMA may write behind ALL
    // outstanding vector-memory operations (s_waitcnt vmcnt(0) -- with this kernel's stores in flight: microseconds, in every step), so
    // the ring is a SEPARATE object that C++ code never reads (its reads are inline asm) -- nothing aliases the operand rings / fragments.
    __shared__ __attribute__((aligned(16))) char smem_img[G::LDS_IMG];
    const EncBranch& B = A.br[DUAL ? 0 : blockIdx.y];       // (DUAL: the shared weights come from branch 0, blockIdx.y = 0)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;

    // ---- A fragments of (layer, k-step, tap row u), gathered from the layer-wise operand images (planes of [16 oc][8 ch] bf16):
    // fragment f = (kq * 3 + u), kq = 0,1: 16->16 | 2..4: 32->16 | 5..9: 48->16; lane (oc = j, k-group g) takes the plane of its (tap, cb).
    // Four-block chunks [x0 | x1] (k-step = tap column v, k-group = channel block); two-block chunks (x0 for the first conv, x2 for
    // the third): k-step 0 = tap columns 0 | 1 (k-groups 0,1 | 2,3), k-step 1 = tap column 2 | the image's zero planes.
    auto a_plane = [&](int kq, int u, int kg) {
        const int L = kq < 2 ? 1 : (kq < 5 ? 2 : 3), q = kq < 2 ? kq : (kq < 5 ? kq - 2 : kq - 5);
        if (L >= 2 && q < 3) return (u * 3 + q) * 4 + kg;
        const int q2 = L == 1 ? q : q - 3, base = L == 3 ? ES_P2 : 0;
        return base + ((q2 == 1 && kg >= 2) ? 18 + (kg & 1) : (u * 3 + (q2 == 0 ? (kg >> 1) : 2)) * 2 + (kg & 1));
    };
    for (int e = tid; e < E2_NFRAG * 64; e += E2_WAVES * 64) {
        const int f = e >> 6, l = e & 63, oc = l & 15, kg = l >> 4;
        const int u = f % 3, kq = f / 3 + 2;
        const int L = kq < 5 ? 2 : 3;
        const int plane = a_plane(kq, u, kg);
        reinterpret_cast<uint4*>(smem)[e] = B.wpk[L - 1][plane * 16 + oc];
    }
    // the wave's ring starts zeroed: pad k-groups, warm-up rows and the granules next to a row are read before they are ever written
    // (their products are multiplied by zero weights or only reach discarded columns, but NaN bit patterns would not stay there)
    const int ring = E2_WBYTES + wave * E2_RING;     // byte address in LDS
    for (int e = lane; e < E2_RING / 16; e += 64) reinterpret_cast<uint4*>(smem + ring)[e] = make_uint4(0u, 0u, 0u, 0u);
    if (wave == E2_WAVES - 1 && lane < 4) reinterpret_cast<uint4*>(smem + E2_WBYTES + E2_WAVES * E2_RING)[lane] = make_uint4(0u, 0u, 0u, 0u);
    const int iring = wave * (E2_IMGS * E2_IMGROW);      // byte offset inside smem_img
    for (int e = lane; e < E2_IMGS * E2_IMGROW / 16; e += 64) reinterpret_cast<uint4*>(smem_img + iring)[e] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();

    const int item = blockIdx.x * E2_WAVES + wave;
    if (item >= A.items) return;
    const int strip = item % A.nstrips;
    const int seg = (item / A.nstrips) % A.nseg;
    const int in_ = item / (A.nstrips * A.nseg);
    const int H = A.h, W = A.w;
    const int y_lo = seg * A.seg_rows, y_hi = min(H, y_lo + A.seg_rows);
    if (y_lo >= y_hi) return;

    // ---- strip geometry.  Ring pixel p of the strip is image column r0 + p.  The first strip starts at column -1 (its ghost), the last
    // one ends at column w (its ghost); a single strip (w <= W - 2) has both.  Kept columns [o_lo, o_hi): 3 px from an interior strip edge.
    int r0, o_lo, o_hi;
    if (A.nstrips == 1) { r0 = -1; o_lo = 0; o_hi = W; }
    else if (strip == 0) { r0 = -1; o_lo = 0; o_hi = G::EDGE; }
    else {
        o_lo = G::EDGE + E2_KEEP * (strip - 1);
        if (strip == A.nstrips - 1) { r0 = W - (E2_W - 1); o_hi = W; }
        else { r0 = o_lo - 3; o_hi = o_lo + E2_KEEP; }
    }
    const bool ghost_l = r0 < 0, ghost_r = r0 + E2_W > W;
    const int pg_r = W - r0;                          // ring pixel of the right ghost (column w) when ghost_r

    // ---- lane constants
    const int h2 = g >> 1, cbk = g & 1;
    const int la = g * 256 + j * 16;                                   // A operand: k-group plane g of a fragment, row (output channel) j
    const int lb4 = ring + cbk * CBS + j * 16 - 16;                     // B operand, four-block chunk: tensor h2 (x0 | x1), block cbk, tap column 0
    const int lb2 = ring + cbk * CBS + (j + h2) * 16 - 16;              // two-block chunk: tap column h2 (k-step 0); + 32: tap column 2 (k-step 1)
    e2_bf16x8 aL1[2][3];                                               // the 16->16 conv's A fragments (k-steps 0, 1 x tap rows), from its operand image
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int u = 0; u < 3; ++u) aL1[q][u] = __builtin_bit_cast(e2_bf16x8, B.wpk[0][a_plane(q, u, g) * 16 + j]);
    // epilogue side (after the row swap): this lane holds the granule of pixel 32 p + 16 (g & 1) + j, channel block g >> 1
    const int px_e = 16 * (g & 1) + j, cb_e = g >> 1;
    const int lw_e = ring + cb_e * CBS + px_e * 16;
    bool ok_e[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int ce = r0 + px_e + (DUAL ? 0 : 32 * p);     // (DUAL: pair p = image p, the same columns)
        ok_e[p] = ce >= o_lo && ce < o_hi;
    }
    // first layer on the fp32 matrix path (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate): one MFMA per tap ROW u, its
    // K = 4 the three tap columns (k-group g = column g - 1; g = 3 is padding).  The B operand of lane (pixel j, group g) is the image
    // value at (row s - 1 + u, column reflect(16 t + j) + g - 1) -- loaded straight into the operand register, no VALU work at all --,
    // the A operand the weight w0[oc j][u][g].  Output layout = the bf16 layers', so x0 shares their epilogue.
    // Image rows reach the wave through its LDS image ring: entry e of a row = image column reflect(r0 - 1 + e), so the operand of ring
    // pixel p, tap column g - 1, is entry p + g (the strip's ghost pixels come out wrong that way -- column -1 would see columns 2, 1, 0
    // instead of 0, 1, 2 -- and are overwritten by the epilogue's ghost copy like those of x1 / x2).
    float a0[3];
    const int li = (int)(size_t)(__attribute__((address_space(3))) char*)smem_img + iring + (j + g) * 4;   // operand read (LDS byte address): + 64 t + slot
    // DUAL: entries 0..33 = image a's columns r0 - 1 .. r0 + 32, entries 34..67 = image b's (tiles 2, 3 read 8 bytes further: read_img_row)
    const int cdma0 = 4 * min(max(reflect_idx(r0 - 1 + (DUAL && lane >= 34 ? lane - 34 : lane), W), 0), W - 1);  // DMA source column (bytes) of entries 0..63
    const int cdma1 = 4 * min(max(reflect_idx(DUAL ? r0 + 29 + (lane & 3) : r0 + 63 + (lane & 1), W), 0), W - 1);   // entries 64, 65 (lanes 0, 1; 64-pixel strips only) | DUAL: 64..67
#pragma unroll
    for (int u = 0; u < 3; ++u) a0[u] = g < 3 ? B.w0[j * 9 + u * 3 + g] : 0.f;
    e2_f32x4 biasC0;
#pragma unroll
    for (int r = 0; r < 4; ++r) biasC0[r] = B.b0 != nullptr ? B.b0[4 * g + r] : 0.f;
    e2_f32x4 biasC[3];    // C operand of a fresh accumulator: this lane's output channels 4 g .. 4 g + 3
#pragma unroll
    for (int L = 0; L < 3; ++L)
#pragma unroll
        for (int r = 0; r < 4; ++r) biasC[L][r] = B.bias[L] != nullptr ? B.bias[L][4 * g + r] : 0.f;

    // global traffic through buffer descriptors: a 32-bit lane offset (constant per lane) + a scalar row offset + an immediate, no 64-bit
    // lane addresses (an image of the widest view is < 4 GiB; out-of-range offsets -- the left ghost's -16 -- are dropped by the hardware)
    const unsigned plane_b = (unsigned)(B.out.plane * 16), row_b = (unsigned)B.out.ws * 16u;
    char* out_img = B.out.base + ((long long)in_ * B.out.img + (long long)B.out.cb_off * B.out.plane) * 16;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out_img, 0, (int)((unsigned)(B.out.cb_total - B.out.cb_off) * plane_b), 0x00020000);
    const char* img = reinterpret_cast<const char*>(B.img + (long long)in_ * H * W);
    // DUAL: image b and its output view, the sum's view (same h, w, halo 0 => the same plane / row pitch: checked by the host)
    const EncBranch& Bb = A.br[DUAL ? 1 : 0];
    const char* img_b = reinterpret_cast<const char*>(Bb.img + (long long)in_ * H * W);
    const char* img_l = (DUAL && lane >= 34) ? img_b : img;      // per-lane source image of the first DMA of a row
    char* out_img_b = Bb.out.base + ((long long)in_ * Bb.out.img + (long long)Bb.out.cb_off * Bb.out.plane) * 16;
    const __amdgpu_buffer_rsrc_t rs_out_b = __builtin_amdgcn_make_buffer_rsrc(out_img_b, 0, (int)((unsigned)(Bb.out.cb_total - Bb.out.cb_off) * plane_b), 0x00020000);
    char* sum_img = DUAL ? A.sum.base + ((long long)in_ * A.sum.img + (long long)A.sum.cb_off * A.sum.plane) * 16 : out_img;
    const __amdgpu_buffer_rsrc_t rs_sum = __builtin_amdgcn_make_buffer_rsrc(sum_img, 0, (int)((unsigned)((DUAL ? A.sum.cb_total - A.sum.cb_off : 1)) * plane_b), 0x00020000);
    // per-lane store offsets; a lane outside the strip's kept columns carries bit 31 = beyond the descriptor's range = dropped
    unsigned st_e[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) st_e[p] = ok_e[p] ? (unsigned)cb_e * plane_b + (unsigned)(r0 + px_e + (DUAL ? 0 : 32 * p)) * 16u : 0x80000000u;

    // rows each stage touches: x0 rows [a_lo, a_hi) feed layer 1, x1 rows [b_lo, b_hi) layer 2, x2 rows [c_lo, c_hi) layer 3
    const int a_lo = max(0, y_lo - 3), a_hi = min(H, y_hi + 3);
    const int b_lo = max(0, y_lo - 2), b_hi = min(H, y_hi + 2);
    const int c_lo = max(0, y_lo - 1), c_hi = min(H, y_hi + 1);
    const int f_lo = max(y_lo + 6, 7), f_hi = min(a_hi, H - 1);   // steps whose three layers are all active, emit, and touch no reflect row

    e2_f32x4 acc[3][3][NT];
#pragma unroll
    for (int L = 0; L < 3; ++L)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[L][r][t] = (e2_f32x4){0.f, 0.f, 0.f, 0.f};

    auto rrow = [&](int y) { return min(max(reflect_idx(y, H), 0), H - 1); };
    // image rows s-1, s, s+1 (one operand value per column tile) in register sets (P+2) % 3, P, (P+1) % 3 of `win`.  A row is REQUESTED
    // (LDS-DMA into the image ring: no destination registers, nothing the compiler tracks) SEVEN steps before the step that reads it out
    // of LDS, i.e. more than 63 vector-memory operations earlier, so that in the steady state it needs no s_waitcnt at all (see the read
    // at the end of a step).  The reads are inline asm: a C++ read of DMA-written LDS makes the compiler wait for ALL outstanding
    // vector-memory operations.
    float win[3][NT];
    // (inline asm, not __builtin_amdgcn_global_load_lds: the compiler orders every later LDS access of the wave -- reads AND writes, of any
    // LDS object -- behind a pending LDS-DMA it knows of with s_waitcnt vmcnt(0), i.e. behind all of the step's stores)
    auto dma_issue = [&](const char* gsrc, unsigned lds_dst) __attribute__((always_inline)) {
        __asm__ volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" : : "s"(lds_dst), "v"(gsrc) : "memory", "m0");
    };
    const unsigned iring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_img + (unsigned)iring;
    auto dma_img_row = [&](int y) __attribute__((always_inline)) {
        const char* src = img + (long long)rrow(y) * W * 4;
        const unsigned dst = iring_lds + (unsigned)((y & (E2_IMGS - 1)) * E2_IMGROW);
        if (DUAL) {
            const long long ro = (long long)rrow(y) * W * 4;
            dma_issue(img_l + ro + cdma0, dst);
            if (lane < 4) dma_issue(img_b + ro + cdma1, dst + 256);
        } else if (NT == 4) {
            dma_issue(src + cdma0, dst);
            if (lane < 2) dma_issue(src + cdma1, dst + 256);
        } else if (lane < E2_W + 2) dma_issue(src + cdma0, dst);
    };
    auto read_img_row = [&](int y, float (&dst)[NT]) __attribute__((always_inline)) {
        const int a = li + (y & (E2_IMGS - 1)) * E2_IMGROW;
        if constexpr (DUAL)      // (image b's entries start at 34: its tiles read two entries further)
            __asm__ volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:64\n\tds_read_b32 %2, %4 offset:136\n\tds_read_b32 %3, %4 offset:200\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]) : "v"(a) : "memory");
        else if constexpr (NT == 4)
            __asm__ volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:64\n\tds_read_b32 %2, %4 offset:128\n\tds_read_b32 %3, %4 offset:192\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]) : "v"(a) : "memory");
        else
            __asm__ volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:64\n\ts_waitcnt lgkmcnt(0)" : "=&v"(dst[0]), "=&v"(dst[1]) : "v"(a) : "memory");
    };
    const e2_i16x2 zero2 = {0, 0};
    auto relu2 = [&](uint32_t w) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(e2_i16x2, w), zero2)); };

    // ---- one row step (P = (s - a_lo) % 3 at compile time; FAST: no stage is idle, no reflect row is touched).
    // The step is a FLAT software pipeline over its ten k-steps (16->16: 2, 32->16: 3, 48->16: 5): region n requests the operand
    // fragments of k-step n + 1 and runs the 12 MFMAs of k-step n on the fragments requested one region earlier, plus one slice of the
    // step's VALU work (the two halves of the x0 row, the epilogue of the layer that finished one region ago).  The regions are
    // separated by scheduling fences: left alone, the scheduler hoists every LDS read of the step to its top (one wave per SIMD = a
    // 512-register budget it is happy to fill) and the allocator spills -- scratch reloads wait on vmcnt(0), i.e. on the step's stores.
#define E2_FENCE() __builtin_amdgcn_sched_barrier(0)
    auto step = [&](auto Pc, auto Fc, int s) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        constexpr bool FAST = decltype(Fc)::value != 0;
        const int R1 = s - 1, R2 = s - 3, R3 = s - 5;      // input row of layer L = s - (2 L - 1); it completes out row R - 1
        const bool x0_on = FAST || (s >= a_lo && s < a_hi);
        const bool on1 = FAST || (R1 >= a_lo && R1 < a_hi), on2 = FAST || (R2 >= b_lo && R2 < b_hi), on3 = FAST || (R3 >= c_lo && R3 < c_hi);
        const bool em1 = FAST || (R1 - 1 >= b_lo && R1 - 1 < b_hi), em2 = FAST || (R2 - 1 >= c_lo && R2 - 1 < c_hi),
                   em3 = FAST || (R3 - 1 >= y_lo && R3 - 1 < y_hi);
        // operand bases of this step (lane groups 0,1 of a four-block chunk read the x0 ring, 2,3 the x1 ring)
        const int bL1 = lb2 + E2_X0 + (R1 & (E2_S0 - 1)) * ROWB;
        const int bL2 = lb4 + (h2 ? E2_X1 + (R2 & (E2_S1 - 1)) * ROWB : E2_X0 + (R2 & (E2_S0 - 1)) * ROWB);
        const int bL3 = lb4 + (h2 ? E2_X1 + (R3 & (E2_S1 - 1)) * ROWB : E2_X0 + (R3 & (E2_S0 - 1)) * ROWB);
        const int bL3x = lb2 + E2_X2 + (R3 & (E2_S2 - 1)) * ROWB;
        e2_bf16x8 fa[2][3], fb[2][NT];      // double-buffered fragments: k-step n lives in set n & 1

        // k-step n: 0,1 = 16->16 on x0 | 2..4 = 32->16 on [x0 | x1] | 5..7 = 48->16 on [x0 | x1], 8,9 = on x2
        auto load_k = [&](auto Nc) __attribute__((always_inline)) {
            constexpr int N = decltype(Nc)::value;
            if ((E2_ABL & 4) && N >= 2) {
#pragma unroll
                for (int u = 0; u < 3; ++u) fa[N & 1][u] = fa[N & 1][u];
                return;
            }
            const char* pa = smem + la + (N >= 2 ? N - 2 : 0) * 3 * 1024;
#pragma unroll
            for (int u = 0; u < 3; ++u) fa[N & 1][u] = N < 2 ? aL1[N < 2 ? N : 0][u] : *reinterpret_cast<const e2_bf16x8*>(pa + u * 1024);
            const char* pb = smem + (N < 2 ? bL1 + 32 * N : (N < 5 ? bL2 + 16 * (N - 2) : (N < 8 ? bL3 + 16 * (N - 5) : bL3x + 32 * (N - 8))));
#pragma unroll
            for (int t = 0; t < NT; ++t) fb[N & 1][t] = *reinterpret_cast<const e2_bf16x8*>(pb + t * 256);
        };
        auto mma_k = [&](auto Nc) __attribute__((always_inline)) {
            constexpr int N = decltype(Nc)::value;
            if (E2_ABL & 2) return;
            constexpr int L = N < 2 ? 1 : (N < 5 ? 2 : 3), LAG = 2 * L - 1;
            constexpr bool first = N == 0 || N == 2 || N == 5;
            constexpr int i0 = (P + 9 - LAG + 1) % 3, i1 = (P + 9 - LAG) % 3, i2 = (P + 9 - LAG - 1) % 3;   // accumulator rows of out rows R+1, R, R-1
            const int R = s - LAG;
            const bool top = !FAST && R == 1, bot = !FAST && R == H - 2;
            if (!FAST && first && R == 0) {   // image row 0 has no row above it to open its accumulator: start it from the bias here
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[L - 1][i1][t] = biasC[L - 1];
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const e2_bf16x8 b = fb[N & 1][t];
                acc[L - 1][i0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[N & 1][0], b, first ? biasC[L - 1] : acc[L - 1][i0][t], 0, 0, 0);
                acc[L - 1][i1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[N & 1][1], b, acc[L - 1][i1][t], 0, 0, 0);
                acc[L - 1][i2][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[N & 1][2], b, acc[L - 1][i2][t], 0, 0, 0);
                if (!FAST) {
                    if (top) acc[L - 1][i2][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[N & 1][0], b, acc[L - 1][i2][t], 0, 0, 0);   // row -1 = row 1
                    if (bot) acc[L - 1][i0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[N & 1][2], b, acc[L - 1][i0][t], 0, 0, 0);   // row h = row h-2
                }
            }
        };
        // x0 row s: three fp32 MFMAs per column tile (tap rows u = 0, 1, 2 on image rows s - 1, s, s + 1), the bias as the first C
        e2_f32x4 acc0[NT];
        auto x0_mma = [&]() __attribute__((always_inline)) {
            if (E2_ABL & 16) {
#pragma unroll
                for (int t = 0; t < NT; ++t) acc0[t] = biasC0;
                return;
            }
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], win[(P + 2 + u) % 3][t], u == 0 ? biasC0 : acc0[t], 0, 0, 0);
        };
        // epilogue of a finished row (x0: row s; layer L >= 1: out row R - 1 = s - 2 L): round, ReLU (as a signed 16-bit max on the rounded
        // pair), pair the column tiles, ring + global stores, ghost pixels
        // a general step whose layer does not emit still issues that layer's store instructions (offset beyond the descriptor: dropped by
        // the hardware): EVERY step issues the same number of vector-memory operations, which is what lets the image rows do without
        // an s_waitcnt (see the read at the end of the step)
        auto dummy_stores = [&]() __attribute__((always_inline)) {
            if (E2_ABL & 1) return;
#pragma unroll
            for (int p = 0; p < G::NST; ++p)
                __builtin_amdgcn_raw_buffer_store_b128((e2_u32x4){0u, 0u, 0u, 0u}, rs_out, (int)0x80000000u, 0, 0);
        };
        auto epilogue = [&](auto Lc) __attribute__((always_inline)) {
            constexpr int L = decltype(Lc)::value;
            if (E2_ABL & 8) return;
            constexpr int LAG = 2 * L - 1;
            constexpr int i2 = (P + 9 - LAG - 1) % 3;
            constexpr int XO = L == 0 ? E2_X0 : (L == 1 ? E2_X1 : E2_X2), SO = L == 0 ? E2_S0 : (L == 1 ? E2_S1 : E2_S2);    // ring of the OUTPUT (L < 3)
            const int r = L == 0 ? s : s - LAG - 1;
            const int wb = lw_e + XO + (r & (SO - 1)) * ROWB;
            const unsigned own = (r >= y_lo && r < y_hi) ? 0u : 0x80000000u;
            const int orow = (int)((unsigned)(2 * L) * plane_b + (unsigned)r * row_b);
            uint4 o[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                uint32_t pk[2][2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const e2_f32x4 a = L == 0 ? acc0[2 * p + e] : acc[L > 0 ? L - 1 : 0][i2][2 * p + e];
                    pk[e][0] = relu2(pack_bf16x2(a[0], a[1]));
                    pk[e][1] = relu2(pack_bf16x2(a[2], a[3]));
                }
                const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                o[p] = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                if (L < 3) *reinterpret_cast<uint4*>(smem + wb + p * 512) = o[p];
                if (E2_ABL & 32)      // every store of a block into the same 16 KiB (cache resident): the store INSTRUCTIONS without the HBM write stream
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(e2_u32x4, o[p]), rs_out, (int)((st_e[p] & 0x3ff0u) | own), (int)(blockIdx.x * 16384u), 0);
                else if (!(E2_ABL & 1)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(e2_u32x4, o[p]), (DUAL && p == 1) ? rs_out_b : rs_out, (int)(st_e[p] | own), orow, E2_STORE_AUX);
            }
            if constexpr (DUAL) {
                // f1 + f2 of this lane's pixel and channel block: the two images' bf16 values added in fp32, rounded once (mmif_fuse_elem_fwd's
                // arithmetic: bit-identical to the separate pass)
                auto add2 = [](uint32_t a, uint32_t b) {
                    return pack_bf16x2(__uint_as_float(a << 16) + __uint_as_float(b << 16), __uint_as_float(a & 0xffff0000u) + __uint_as_float(b & 0xffff0000u));
                };
                const uint4 ob = o[NP - 1];
                const e2_u32x4 sm = {add2(o[0].x, ob.x), add2(o[0].y, ob.y), add2(o[0].z, ob.z), add2(o[0].w, ob.w)};
                if (!(E2_ABL & 1)) __builtin_amdgcn_raw_buffer_store_b128(sm, rs_sum, (int)(st_e[0] | own), orow, E2_STORE_AUX);
            }
            if (L < 3) {
                // ghost pixels of an edge strip, AFTER both halves of the row are in the ring (a narrow image's right ghost lies inside
                // the computed range): column -1 := column 1 (ring pixel 0 := 2), column w := column w - 2
                if (ghost_l && px_e == 2) {
                    *reinterpret_cast<uint4*>(smem + wb - 32) = o[0];
                    if constexpr (DUAL) *reinterpret_cast<uint4*>(smem + wb + 512 - 32) = o[NP - 1];      // image b's strip starts at ring pixel 32
                }
                if (ghost_r) {
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        if (px_e + (DUAL ? 0 : 32 * p) == pg_r - 2) *reinterpret_cast<uint4*>(smem + wb + p * 512 + 32) = o[p];
                }
            }
        };

        load_k(E2I<0>());
        E2_FENCE();
        load_k(E2I<1>()); if (on1) mma_k(E2I<0>()); if (x0_on) x0_mma();
        E2_FENCE();
        load_k(E2I<2>()); if (on1) mma_k(E2I<1>()); if (x0_on) epilogue(E2I<0>());
        E2_FENCE();
        // (the image row that step s + 1 shifts in -- row s + 2 -- replaces row s - 1 in its register set: requested as soon as the x0
        // row is done, a whole step ahead of its use)
        load_k(E2I<3>()); if (on2) mma_k(E2I<2>());
        if (x0_on && !(E2_ABL & 16)) dma_img_row(s + G::AHEAD);   // (read out of the ring at the end of step s + AHEAD - 2)
        if (em1) epilogue(E2I<1>()); else dummy_stores();
        E2_FENCE();
        load_k(E2I<4>()); if (on2) mma_k(E2I<3>());
        E2_FENCE();
        load_k(E2I<5>()); if (on2) mma_k(E2I<4>());
        E2_FENCE();
        load_k(E2I<6>()); if (on3) mma_k(E2I<5>()); if (em2) epilogue(E2I<2>()); else dummy_stores();
        E2_FENCE();
        load_k(E2I<7>()); if (on3) mma_k(E2I<6>());
        E2_FENCE();
        load_k(E2I<8>()); if (on3) mma_k(E2I<7>());
        E2_FENCE();
        load_k(E2I<9>()); if (on3) mma_k(E2I<8>());
        E2_FENCE();
        if (on3) mma_k(E2I<9>());
        E2_FENCE();
        if (em3) epilogue(E2I<3>()); else dummy_stores();
        if (x0_on && !(E2_ABL & 16)) {
            // image row s + 2 for the next step, into the set of row s - 1 (dead since this step's first-layer MFMAs).  NO WAIT: the row
            // was requested either before the first step (the prologue waits once, with no store in flight yet) or AHEAD - 2 steps ago,
            // and every step issues exactly OPS vector-memory operations -- at least 64 since the request.  A wave has at most 63 in
            // flight and they complete in order, so the row has landed.  That is the point of the whole arrangement: any
            // s_waitcnt vmcnt(n) also waits for STORES, and with the HBM write queue full (this kernel's steady state) stores stay in
            // flight for microseconds -- requested one step ahead, or four steps ahead with a counted wait, the image rows cost the
            // kernel ~100 us (profiles/r05_ubench_enc_stream2_ablation.txt).
            read_img_row(s + 2, win[(P + 2) % 3]);
        }
        E2_FENCE();
    };
#undef E2_FENCE

    // ---- the pipeline: s = a_lo .. (last x3 row + 6); phase 0 at s = a_lo
    static_assert(G::AHEAD + 1 <= E2_IMGS && 2 * NP + (G::AHEAD - 2) * G::OPS >= 64, "image ring / request distance");
#pragma unroll 1
    for (int y = a_lo - 1; y < a_lo + G::AHEAD; ++y) dma_img_row(y);
    __builtin_amdgcn_s_waitcnt(0x0f70);      // (the only vector-memory wait of the kernel: nothing but these requests is in flight)
    read_img_row(a_lo - 1, win[2]);
    read_img_row(a_lo, win[0]);
    read_img_row(a_lo + 1, win[1]);
    const int s_end = y_hi + 6;      // x3 row y_hi - 1 is emitted at step y_hi + 5
    // Three stretches of steps, in groups of three (the accumulator rotation): general steps up to the first group that lies inside
    // [f_lo, f_hi), the branch-free groups, general steps again to the end.  The general body and the fast body are SEPARATE loops of one
    // outer loop -- as the two arms of an if inside a single row loop they made the register allocator shuffle the 144 accumulator
    // registers between the arms in every iteration (512 registers + scratch; either body alone needs 172 + 144 / 194 + 240).
    int s = a_lo;
#pragma unroll 1
    for (int part = 0; part < 2; ++part) {
        const int stop = part == 0 ? min(s_end, f_lo) : s_end;
#pragma unroll 1
        for (; s < stop; s += 3) {
            step(E2I<0>(), E2I<0>(), s);
            step(E2I<1>(), E2I<0>(), s + 1);
            step(E2I<2>(), E2I<0>(), s + 2);
        }
        if (part == 0) {
#pragma unroll 1
            for (; s + 2 < f_hi; s += 3) {
                step(E2I<0>(), E2I<1>(), s);
                step(E2I<1>(), E2I<1>(), s + 1);
                step(E2I<2>(), E2I<1>(), s + 2);
            }
        }
    }
}

// items-per-launch heuristic: every (strip, segment, image, branch) is one wave, WAVES waves a block, one block per CU (LDS).  More
// segments fill the chip but each pays 3 warm-up rows of x0 plus the 6-step pipeline skew: minimise rounds x steps per wave.
template <int NT, bool DUAL = false>
static void e2_geometry(int n, int h, int w, int nb, int& nstrips, int& nseg, int& seg_rows) {
    using G = E2G<NT, DUAL>;
    nstrips = w <= G::W - 2 ? 1 : 2 + (w > 2 * G::EDGE ? (w - 2 * G::EDGE + G::KEEP - 1) / G::KEEP : 0);
    const int ncu = cached_num_cus();
    long long best = -1;
    nseg = 1;
    for (int k = 1; k <= (h + 7) / 8; ++k) {
        const int rows = (h + k - 1) / k;
        const long long blocks = (long long)nb * (((long long)n * nstrips * k + G::WAVES - 1) / G::WAVES);
        const long long cost = ((blocks + ncu - 1) / ncu) * (rows + 12);
        if (best < 0 || cost < best) { best = cost; nseg = k; }
    }
    seg_rows = (h + nseg - 1) / nseg;
    nseg = (h + seg_rows - 1) / seg_rows;   // drop empty trailing segments
}

static int g_es2 = 2;       // mmif_debug_set_enc_stream2: 0 = the round-2 kernel, 1 = 64-pixel strips (four waves per CU), 2 (default) = 32-pixel strips (eight)
static void e2_init() {}
// (the size limits of this kernel -- 32-bit lane offsets with bit 31 = masked lane -- are part of the predicate: a frame beyond them takes the
// round-2 kernel, which allows 4 GiB per image, instead of failing: ADVICE r5)
bool enc_stream2_ok(const EncArgs& A, int nb) {
    e2_init();
    if (!(g_es2 >= 1 && A.h >= 2 && A.w >= 2 && (long long)A.h * A.w * 4 < (1ll << 31))) return false;
    for (int b = 0; b < nb; ++b)
        if ((long long)A.br[b].out.cb_total * A.br[b].out.plane * 16 >= (1ll << 31)) return false;
    return true;
}

template <int NT>
static int e2_launch(EncArgs& A, int nb, hipStream_t st) {
    e2_geometry<NT>(A.n, A.h, A.w, nb, A.nstrips, A.nseg, A.seg_rows);
    A.items = A.n * A.nseg * A.nstrips;
    hipLaunchKernelGGL(enc_stream2_fwd_kernel<NT>, dim3(cdiv(A.items, E2G<NT>::WAVES), nb), dim3(E2G<NT>::WAVES * 64), 0, st, A);
    return check_launch("dense_encoder_fwd (stream2)");
}

// both images of a pair through one shared encoder + their sum (A.sum), one wave per (strip, segment, pair)
int enc_stream2_launch_dual(EncArgs& A, hipStream_t st) {
    MMIF_REQUIRE(A.relu0 == 1, "dense_encoder_fwd_sum: the first layer's ReLU is compiled in");
    MMIF_REQUIRE((long long)A.h * A.w * 4 < (1ll << 31), "dense_encoder_fwd_sum: one image must stay below 2 GiB");
    for (int b = 0; b < 2; ++b)
        MMIF_REQUIRE((long long)A.br[b].out.cb_total * A.br[b].out.plane * 16 < (1ll << 31),
                     "dense_encoder_fwd_sum: one image of an output allocation must stay below 2 GiB");
    MMIF_REQUIRE((long long)A.sum.cb_total * A.sum.plane * 16 < (1ll << 31), "dense_encoder_fwd_sum: one image of the sum's allocation must stay below 2 GiB");
    e2_geometry<4, true>(A.n, A.h, A.w, 1, A.nstrips, A.nseg, A.seg_rows);
    A.items = A.n * A.nseg * A.nstrips;
    hipLaunchKernelGGL((enc_stream2_fwd_kernel<4, true>), dim3(cdiv(A.items, E2G<4, true>::WAVES), 1), dim3(E2G<4, true>::WAVES * 64), 0, st, A);
    return check_launch("dense_encoder_fwd_sum (stream2 dual)");
}

int enc_stream2_launch(EncArgs& A, int nb, hipStream_t st) {
    MMIF_REQUIRE(A.relu0 == 1, "dense_encoder_fwd (stream2): the first layer's ReLU is compiled in");
    MMIF_REQUIRE((long long)A.h * A.w * 4 < (1ll << 31), "dense_encoder_fwd (stream2): one image must stay below 2 GiB");
    for (int b = 0; b < nb; ++b)
        MMIF_REQUIRE((long long)A.br[b].out.cb_total * A.br[b].out.plane * 16 < (1ll << 31),
                     "dense_encoder_fwd (stream2): one image of the output allocation must stay below 2 GiB (bit 31 of a store offset = masked lane)");
    e2_init();
    return g_es2 == 1 ? e2_launch<4>(A, nb, st) : e2_launch<2>(A, nb, st);
}

}  // namespace mmif

extern "C" void mmif_debug_set_enc_stream2(int32_t mode) { mmif::g_es2 = mode < 0 ? 0 : (mode > 2 ? 2 : mode); }
