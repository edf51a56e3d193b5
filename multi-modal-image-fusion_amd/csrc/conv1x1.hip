// Streaming 1x1 convolution, forward and dgrad, bf16 MFMA (v_mfma_f32_16x16x32_bf16, fp32 accumulate): the second layer of every
// ConvBlock of NestFuse / RFN-Nest (reference core/block.py:708-722: ConvLayer(in // 2, out, ksize=1)), RFN's 2C -> C layer
// (core/block.py:749) -- 14 layers of NestFuse's train step, 17 % of it on the register-staged kernels (profiles/r03_kernel_stats_nestfuse_*).
//
// A 1x1 layer has no spatial structure: y[o, p] = sum_c W[o, c] x[c, p] over the pixels p of a plane taken as ONE linear run, and in the
// blocked layout ([n][C/8][h][w][8]) the B fragment of the MFMA -- lane (j, g) = 8 consecutive K values of column j -- IS a granule: the
// 8 channels of block 4 chunk + g at pixel p0 + j.  So the input never touches LDS: every lane loads its granules straight from global
// memory (16 lanes = 256 contiguous bytes of one plane), the packed weights (the operand image mmif_pack_weights already builds:
// [chunk][k-group][M16p][8]) are resident in LDS for the whole launch, and a wave owns 32 consecutive pixels x ALL output channels:
//     per chunk of 32 input channels: 2 granule loads per lane, M16p / 16 weight fragments from LDS, each feeding two MFMAs.
// The register-staged conv_mfma_kernel<1, MF, *> staged every 16 x 16 tile through LDS behind two block barriers per chunk and re-read the
// input once per 64-channel M-block.  Here HBM traffic = one read of x + one write of y (+ one read of the ReLU-mask tensor in dgrad).
//   * forward: bias + optional ReLU, one rounding;  dgrad (the operand image is the transposed one): ReLU mask [x > 0] per output channel
//     block (mask_bits), halo-1 gradient tensors are walked over their whole padded plane (the zero ring of gy gives a zero ring of gx);
//   * persistent blocks of 4 autonomous waves (no barrier after the weights are resident), as many per CU as LDS / registers admit;
//   * epilogue: v_permlane16_swap pairs the two 16-pixel halves so that every lane stores one 16-byte granule per 16-channel fragment.
// Same k-group order and rounding points as conv_mfma_kernel<1, ...>: results are bit-identical (tests/test_gpu_conv1x1.py).
#include "common.hpp"
#include <stdlib.h>

namespace mmif {

typedef __attribute__((ext_vector_type(8))) __bf16 c1_bf16x8;
typedef __attribute__((ext_vector_type(4))) float c1_f32x4;

constexpr int C1_WAVES = 4;
constexpr int C1_PX = 32;   // pixels per wave item (two MFMA N tiles)

template <int NMT>
struct C1Occ { static constexpr int waves_per_eu = NMT <= 4 ? 4 : (NMT <= 8 ? 3 : 2); };

template <int NMT, bool DGRAD>
__global__ __launch_bounds__(C1_WAVES * 64, C1Occ<NMT>::waves_per_eu) void conv1x1_stream_kernel(
    TV tin, TV tout, TV tmask, const uint4* __restrict__ wpk, const float* __restrict__ bias, int n_out, int m16p, int nch, int relu,
    unsigned long long mask_bits, int groups_per_img, int total_items) {
    extern __shared__ __attribute__((aligned(16))) char c1_smem[];
    uint4* s_w = reinterpret_cast<uint4*>(c1_smem);
    float* s_bias = reinterpret_cast<float*>(c1_smem + (size_t)nch * 4 * m16p * 16);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- resident operand image + bias (the only block-wide step)
    const int nw = nch * 4 * m16p;
    for (int e = tid; e < nw; e += C1_WAVES * 64) s_w[e] = wpk[e];
    if (!DGRAD)
        for (int e = tid; e < m16p; e += C1_WAVES * 64) s_bias[e] = (bias != nullptr && e < n_out) ? bias[e] : 0.f;
    __syncthreads();
    const int j = lane & 15, g = lane >> 4;
    const int nmt = min(NMT, (n_out + 15) >> 4);            // live 16-row fragments (wave-uniform)
    const int P = (int)tin.plane;                           // stored pixels of one plane (same geometry for tin and tout)
    const unsigned in_plane_b = (unsigned)P * 16u;
    const int cb_hi = tin.cb - 1;
    // this lane's granule after the row swap: pixel half (g & 1), channel-block half (g >> 1) of each 16-row fragment
    const int ph = g & 1, cbh = g >> 1;
    const int wid = blockIdx.x * C1_WAVES + wave, nwaves = gridDim.x * C1_WAVES;
    for (int item = wid; item < total_items; item += nwaves) {
        const int in_ = item / groups_per_img, grp = item - in_ * groups_per_img;
        const int p0 = grp * C1_PX;
        // B operand: pixels p0 + j and p0 + 16 + j of channel block 4 c + g (clamped: tail lanes / blocks past the tensor read valid
        // data that meets zero weight rows or is never stored)
        const unsigned offa = (unsigned)min(p0 + j, P - 1) * 16u, offb = (unsigned)min(p0 + 16 + j, P - 1) * 16u;
        const char* in_img = tin.base + ((long long)in_ * tin.img + (long long)tin.cb_off * tin.plane) * 16;
        c1_f32x4 acc[NMT][2];
#pragma unroll
        for (int m = 0; m < NMT; ++m) {
            acc[m][0] = (c1_f32x4){0.f, 0.f, 0.f, 0.f};
            acc[m][1] = (c1_f32x4){0.f, 0.f, 0.f, 0.f};
        }
        auto ld_b = [&](int c, c1_bf16x8& t0, c1_bf16x8& t1) {
            const char* pl = in_img + (unsigned long long)min(4 * c + g, cb_hi) * in_plane_b;
            t0 = *reinterpret_cast<const c1_bf16x8*>(pl + offa);
            t1 = *reinterpret_cast<const c1_bf16x8*>(pl + offb);
        };
        auto mm = [&](int c, const c1_bf16x8& t0, const c1_bf16x8& t1) {
            const char* wl = reinterpret_cast<const char*>(s_w) + ((size_t)(c * 4 + g) * m16p + j) * 16;
#pragma unroll
            for (int m = 0; m < NMT; ++m) {
                if (m < nmt) {
                    const c1_bf16x8 a = *reinterpret_cast<const c1_bf16x8*>(wl + m * 256);
                    acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, t0, acc[m][0], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, t1, acc[m][1], 0, 0, 0);
                }
            }
        };
        // K loop, the next chunk's granules in flight under this chunk's MFMAs (two explicit register sets: no indexed registers)
        c1_bf16x8 x0, x1, y0, y1;
        ld_b(0, x0, x1);
        int c = 0;
        for (; c + 1 < nch; c += 2) {
            ld_b(c + 1, y0, y1);
            mm(c, x0, x1);
            if (c + 2 < nch) ld_b(c + 2, x0, x1);
            mm(c + 1, y0, y1);
        }
        if (c < nch) mm(c, x0, x1);
        // ---- epilogue.  After the swap lane (g, j) holds channels 8 (2 m + cbh) .. + 7 of pixel p0 + 16 ph + j
        const int ps = p0 + 16 * ph + j;
        const bool px_ok = ps < P;
        const unsigned out_off = (unsigned)min(ps, P - 1) * 16u;
        char* out_img = tout.base + ((long long)in_ * tout.img + (long long)tout.cb_off * tout.plane) * 16;
        const unsigned out_plane_b = (unsigned)tout.plane * 16u;
        bool inside = true;
        unsigned moff = 0;
        const char* m_img = nullptr;
        unsigned m_plane_b = 0;
        if (DGRAD) {
            // ReLU mask: the layer's forward input x (halo 0) at this pixel; gx / gy may carry a halo ring (zero gradient there)
            const int ws = tout.ws, ys = min(ps, P - 1) / ws, xs = min(ps, P - 1) - ys * ws;
            const int yy = ys - tout.halo, xx = xs - tout.halo;
            inside = yy >= 0 && yy < tmask.h && xx >= 0 && xx < tmask.w;
            moff = (unsigned)(min(max(yy, 0), tmask.h - 1) * tmask.ws + min(max(xx, 0), tmask.w - 1)) * 16u;
            m_img = tmask.base + ((long long)in_ * tmask.img + (long long)tmask.cb_off * tmask.plane) * 16;
            m_plane_b = (unsigned)tmask.plane * 16u;
        }
        // fragments in batches of four: a batch's mask granules are fetched in one round trip, then its rows are paired, finished and stored
#pragma unroll
        for (int mb = 0; mb < NMT; mb += 4) {
            if (mb < nmt) {
                uint4 xm[4];
                if (DGRAD) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        xm[q] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
                        const int ocb = 2 * (mb + q) + cbh;
                        if (mb + q < nmt && ocb < tmask.cb && ((mask_bits >> ocb) & 1ull))
                            xm[q] = *reinterpret_cast<const uint4*>(m_img + (unsigned long long)ocb * m_plane_b + moff);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = mb + q;
                    if (m < nmt) {
                        const int ocb = 2 * m + cbh;
                        float cv[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[m][0][r]), __float_as_uint(acc[m][1][r]), false, false);
                            cv[r] = __uint_as_float(sw[0]);
                            cv[4 + r] = __uint_as_float(sw[1]);
                        }
                        if (!DGRAD) {
                            const float4 b0 = *reinterpret_cast<const float4*>(&s_bias[ocb * 8]);
                            const float4 b1 = *reinterpret_cast<const float4*>(&s_bias[ocb * 8 + 4]);
                            const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const float t = cv[i] + bv[i];
                                cv[i] = relu ? fmaxf(t, 0.f) : t;
                            }
                        } else {
                            const uint32_t xw[4] = {xm[q].x, xm[q].y, xm[q].z, xm[q].w};
#pragma unroll
                            for (int i = 0; i < 4; ++i) {   // bf16 > 0  <=>  sign bit clear and magnitude non-zero
                                const uint32_t lo = xw[i] & 0xffffu, hi = xw[i] >> 16;
                                if (!((lo & 0x8000u) == 0 && (lo & 0x7fffu) != 0)) cv[2 * i] = 0.f;
                                if (!((hi & 0x8000u) == 0 && (hi & 0x7fffu) != 0)) cv[2 * i + 1] = 0.f;
                            }
                            if (!inside) {
#pragma unroll
                                for (int i = 0; i < 8; ++i) cv[i] = 0.f;   // the halo ring of a gradient stays zero
                            }
                        }
                        if (px_ok && ocb < tout.cb)
                            *reinterpret_cast<uint4*>(out_img + (unsigned long long)ocb * out_plane_b + out_off) =
                                make_uint4(pack_bf16x2(cv[0], cv[1]), pack_bf16x2(cv[2], cv[3]), pack_bf16x2(cv[4], cv[5]), pack_bf16x2(cv[6], cv[7]));
                    }
                }
            }
        }
    }
}

static int c1_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

static int g_c1_mode = -1;   // mmif_debug_set_conv1x1_stream(0): keep the register-staged kernel (the tests' cross-check)
constexpr size_t C1_MAX_LDS = 128 * 1024;

static size_t c1_lds_bytes(int nch, int m16p) { return (size_t)nch * 4 * m16p * 16 + (size_t)m16p * 4; }

// shapes the streaming kernel takes; everything else stays on conv_mfma_kernel<1, ...>
bool conv1x1_stream_ok(bool dgrad, const TV& tin, const TV& tout, const TV& tmask, int n_out, int m16p, uint64_t mask_bits, uint64_t accum_bits) {
    if (g_c1_mode < 0) g_c1_mode = 1;
    if (g_c1_mode == 0 || accum_bits != 0) return false;
    if (tin.hs != tout.hs || tin.ws != tout.ws || tin.n != tout.n) return false;          // same stored geometry: one linear walk
    if (tin.halo != 0 && !tin.folded) return false;                                       // an unfolded halo-1 gradient needs the fold-on-load path
    if (m16p > 256 || m16p % 16 != 0 || n_out > m16p) return false;
    const int nch = (tin.cb + 3) / 4;
    if (c1_lds_bytes(nch, m16p) > C1_MAX_LDS) return false;
    if (tin.plane * 16 >= (1ll << 31) || tout.plane * 16 >= (1ll << 31)) return false;      // 32-bit in-plane offsets
    if (dgrad && mask_bits != 0) {
        if (tmask.halo != 0 || tmask.h != tout.h || tmask.w != tout.w || tmask.n != tout.n || tmask.plane * 16 >= (1ll << 31)) return false;
    }
    return (long long)tin.n * cdiv(tin.plane, C1_PX) < (1ll << 31);
}

template <int NMT>
static int c1_launch(bool dgrad, const TV& tin, const TV& tout, const TV& tmask, const void* wpk, const float* bias, int n_out, int m16p, int relu,
                     uint64_t mask_bits, hipStream_t st) {
    const int nch = (tin.cb + 3) / 4;
    const size_t lds = c1_lds_bytes(nch, m16p);
    const int groups = cdiv(tin.plane, C1_PX), total = groups * tin.n;
    int bpc = (int)((160 * 1024) / (lds + 512));
    bpc = bpc < 1 ? 1 : (bpc > C1Occ<NMT>::waves_per_eu ? C1Occ<NMT>::waves_per_eu : bpc);
    int grid = c1_num_cus() * bpc;
    if (grid > cdiv(total, C1_WAVES)) grid = cdiv(total, C1_WAVES);
    if (grid < 1) grid = 1;
    if (dgrad) {
        static size_t set_d = 0;
        if (lds > set_d) {
            if (hipFuncSetAttribute((const void*)conv1x1_stream_kernel<NMT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C1_MAX_LDS) != hipSuccess) {
                (void)hipGetLastError();
                set_error("conv1x1_stream: cannot raise the dynamic LDS limit");
                return MMIF_EINVAL;
            }
            set_d = C1_MAX_LDS;
        }
        hipLaunchKernelGGL((conv1x1_stream_kernel<NMT, true>), dim3(grid), dim3(C1_WAVES * 64), lds, st, tin, tout, tmask, (const uint4*)wpk, bias, n_out,
                           m16p, nch, relu, (unsigned long long)mask_bits, groups, total);
    } else {
        static size_t set_f = 0;
        if (lds > set_f) {
            if (hipFuncSetAttribute((const void*)conv1x1_stream_kernel<NMT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C1_MAX_LDS) != hipSuccess) {
                (void)hipGetLastError();
                set_error("conv1x1_stream: cannot raise the dynamic LDS limit");
                return MMIF_EINVAL;
            }
            set_f = C1_MAX_LDS;
        }
        hipLaunchKernelGGL((conv1x1_stream_kernel<NMT, false>), dim3(grid), dim3(C1_WAVES * 64), lds, st, tin, tout, tmask, (const uint4*)wpk, bias, n_out,
                           m16p, nch, relu, (unsigned long long)mask_bits, groups, total);
    }
    return check_launch(dgrad ? "conv1x1_stream dgrad" : "conv1x1_stream fwd");
}

int conv1x1_stream(bool dgrad, const TV& tin, const TV& tout, const TV& tmask, const void* wpk, const float* bias, int n_out, int m16p, int relu,
                   uint64_t mask_bits, hipStream_t st) {
    const int nmt = (n_out + 15) / 16;
    if (nmt <= 4) return c1_launch<4>(dgrad, tin, tout, tmask, wpk, bias, n_out, m16p, relu, mask_bits, st);
    if (nmt <= 8) return c1_launch<8>(dgrad, tin, tout, tmask, wpk, bias, n_out, m16p, relu, mask_bits, st);
    if (nmt <= 12) return c1_launch<12>(dgrad, tin, tout, tmask, wpk, bias, n_out, m16p, relu, mask_bits, st);
    return c1_launch<16>(dgrad, tin, tout, tmask, wpk, bias, n_out, m16p, relu, mask_bits, st);
}

}  // namespace mmif

extern "C" void mmif_debug_set_conv1x1_stream(int32_t mode) { mmif::g_c1_mode = mode ? 1 : 0; }
