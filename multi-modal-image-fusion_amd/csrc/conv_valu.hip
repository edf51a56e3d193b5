// LDS-tiled VALU convolution kernels (fp32 accumulate) for blocked-NHWC tensors.
//
// These are the fp32 PARITY path (results within 1e-3 of the reference, checked against the CPU
// oracle) and the independent cross-check of the bf16 MFMA kernels in conv_mfma.hip.  They take
// the fp32 master weights in nn.Conv2d layout directly (no packing step), quantised through the
// storage type T so that bf16 runs are comparable element for element with the MFMA kernels.
//
// One formulation covers forward and dgrad (reference: core/block.py:98-99 and autograd's
// convolution_backward + reflection_pad2d_backward):
//     out[y][x] = sum_{u,v} Wk[u][v] * IN(y + u - p, x + v - p),     p = k/2
//   forward: IN = reflect-padded activation, Wk = W[o][c][u][v], (y,x) in [0,H)x[0,W)
//   dgrad  : IN = zero-extended (halo-folded) gradient, Wk = W[o][c][k-1-u][k-1-v] with the roles
//            of o and c swapped, (y,x) in [-halo, H+halo) = the padded domain of the input.
#include "common.hpp"

namespace mmif {

constexpr int TILE = 16;   // output tile edge (stored positions)
constexpr int COG = 16;    // output channels per block pass (two channel blocks)

template <typename T, int KS, bool DGRAD>
__global__ __launch_bounds__(256) void conv_valu_kernel(TV tin, TV tout, TV tmask, const float* __restrict__ w,
                                                        const float* __restrict__ bias, int cin, int cout, int relu,
                                                        unsigned long long mask_bits, unsigned long long accum_bits,
                                                        int tiles_x) {
    constexpr int P = KS / 2;
    constexpr int IT = TILE + KS - 1;  // input tile edge
    __shared__ float xin[8][IT][IT + 1];
    __shared__ float wsm[KS * KS][8][COG];

    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int tile_x = blockIdx.x % tiles_x, tile_y = blockIdx.x / tiles_x;
    const int in_ = blockIdx.y;
    const int og = blockIdx.z;  // output channel group of COG channels
    // stored coordinates of this thread's output position, logical coordinates
    const int oys = tile_y * TILE + ty, oxs = tile_x * TILE + tx;
    const int oy = oys - tout.halo, ox = oxs - tout.halo;
    // logical origin of the input tile
    const int iy0 = tile_y * TILE - tout.halo - P, ix0 = tile_x * TILE - tout.halo - P;

    // number of channels on each side in THIS formulation
    const int n_in = DGRAD ? cout : cin;    // channels of tin
    const int n_out = DGRAD ? cin : cout;   // channels of tout
    const int in_blocks = (n_in + 7) / 8;

    float acc[COG];
#pragma unroll
    for (int i = 0; i < COG; ++i) acc[i] = 0.f;

    for (int icb = 0; icb < in_blocks; ++icb) {
        __syncthreads();
        // ---- stage the input tile of channel block icb (fp32 in LDS, [ci][py][px]) ----
        for (int e = tid; e < IT * IT; e += 256) {
            const int py = e / IT, px = e % IT;
            float v[8];
            if (DGRAD) load_grad_fold<T>(tin, in_, icb, iy0 + py, ix0 + px, v);
            else load_act_reflect<T>(tin, in_, icb, iy0 + py, ix0 + px, v);
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) xin[ci][py][px] = v[ci];
        }
        // ---- stage the weights of (icb, og): wsm[tap][ci][co] ----
        for (int e = tid; e < KS * KS * 8 * COG; e += 256) {
            const int co = e % COG, ci = (e / COG) % 8, tap = e / (COG * 8);
            const int u = tap / KS, v = tap % KS;
            const int oc = og * COG + co;   // channel of tout
            const int ic = icb * 8 + ci;    // channel of tin
            float val = 0.f;
            if (oc < n_out && ic < n_in) {
                if (DGRAD) val = w[(((long long)ic * cin + oc) * KS + (KS - 1 - u)) * KS + (KS - 1 - v)];
                else val = w[(((long long)oc * cin + ic) * KS + u) * KS + v];
            }
            wsm[tap][ci][co] = Elem<T>::quant(val);
        }
        __syncthreads();
        // ---- accumulate ----
#pragma unroll
        for (int tap = 0; tap < KS * KS; ++tap) {
            const int u = tap / KS, v = tap % KS;
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) {
                const float xv = xin[ci][ty + u][tx + v];
                const float4* wp = reinterpret_cast<const float4*>(&wsm[tap][ci][0]);
#pragma unroll
                for (int q = 0; q < COG / 4; ++q) {
                    const float4 w4 = wp[q];
                    acc[4 * q + 0] = fmaf(xv, w4.x, acc[4 * q + 0]);
                    acc[4 * q + 1] = fmaf(xv, w4.y, acc[4 * q + 1]);
                    acc[4 * q + 2] = fmaf(xv, w4.z, acc[4 * q + 2]);
                    acc[4 * q + 3] = fmaf(xv, w4.w, acc[4 * q + 3]);
                }
            }
        }
    }

    if (oys >= tout.hs || oxs >= tout.ws) return;
#pragma unroll
    for (int b = 0; b < COG / 8; ++b) {
        const int ocb = og * (COG / 8) + b;  // channel block of the out view
        if (ocb >= tout.cb) break;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = acc[b * 8 + i];
        char* dst = tout.base + tout.gidx(in_, ocb, oys, oxs) * Elem<T>::gran_bytes;
        if (!DGRAD) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int oc = ocb * 8 + i;
                float r = v[i] + ((bias != nullptr && oc < cout) ? bias[oc] : 0.f);
                if (relu) r = fmaxf(r, 0.f);
                v[i] = (oc < cout) ? r : 0.f;
            }
        } else {
            if ((accum_bits >> ocb) & 1ull) {
                float old[8];
                Elem<T>::load(dst, old);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += old[i];
            }
            if ((mask_bits >> ocb) & 1ull) {
                float xm[8];
                load_act_reflect<T>(tmask, in_, ocb, oy, ox, xm);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = xm[i] > 0.f ? v[i] : 0.f;
            }
        }
        Elem<T>::store(dst, v);
    }
}

// ------------------------------------------------------------------------------------------
// wgrad: dw[o][c][u][v] = sum_{n,y,x} g[n][o][y][x] * xpad[n][c][y+u][x+v],  db[o] = sum g
// grid = (G pixel-tile groups, cin blocks, cout groups of 16); each block loops over its share of
// pixel tiles and writes ONE partial [16 co][8 ci][k*k] (+ 16 db) into the workspace; a second
// kernel reduces the G partials in a fixed order (deterministic).
// ------------------------------------------------------------------------------------------
template <typename T, int KS>
__global__ __launch_bounds__(256) void wgrad_valu_kernel(TV tx, TV tg, float* __restrict__ partial, int cin, int cout,
                                                         int tiles_x, int tiles_per_img, int total_tiles) {
    constexpr int P = KS / 2;
    constexpr int IT = TILE + KS - 1;
    constexpr int KK = KS * KS;
    __shared__ float xin[8][IT][IT + 1];
    __shared__ float gsm[COG][TILE * TILE + 1];
    __shared__ float red[2][COG * 8 * KK + COG];

    const int tid = threadIdx.x;
    const int ci = tid & 7, co = (tid >> 3) & 15, half = tid >> 7;
    const int icb = blockIdx.y, og = blockIdx.z;
    const int G = gridDim.x;

    float acc[KK];
#pragma unroll
    for (int i = 0; i < KK; ++i) acc[i] = 0.f;
    float accb = 0.f;

    for (int tile = blockIdx.x; tile < total_tiles; tile += G) {
        const int in_ = tile / tiles_per_img;
        const int tt = tile % tiles_per_img;
        const int y0 = (tt / tiles_x) * TILE, x0 = (tt % tiles_x) * TILE;
        __syncthreads();
        for (int e = tid; e < IT * IT; e += 256) {
            const int py = e / IT, px = e % IT;
            float v[8];
            load_act_reflect<T>(tx, in_, icb, y0 + py - P, x0 + px - P, v);
#pragma unroll
            for (int c = 0; c < 8; ++c) xin[c][py][px] = v[c];
        }
        for (int e = tid; e < TILE * TILE * (COG / 8); e += 256) {
            const int b = e / (TILE * TILE), p = e % (TILE * TILE);
            const int py = p / TILE, px = p % TILE;
            float v[8];
            const int gcb = og * (COG / 8) + b;
            if (gcb < tg.cb) load_grad_fold<T>(tg, in_, gcb, y0 + py, x0 + px, v);  // zero outside the image
            else {
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = 0.f;
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) gsm[b * 8 + c][p] = v[c];
        }
        __syncthreads();
        for (int py = half * (TILE / 2); py < (half + 1) * (TILE / 2); ++py) {
#pragma unroll 4
            for (int px = 0; px < TILE; ++px) {
                const float g = gsm[co][py * TILE + px];
                if (ci == 0) accb += g;
#pragma unroll
                for (int tap = 0; tap < KK; ++tap) acc[tap] = fmaf(g, xin[ci][py + tap / KS][px + tap % KS], acc[tap]);
            }
        }
    }
    // combine the two pixel halves, then write the partial
#pragma unroll
    for (int tap = 0; tap < KK; ++tap) red[half][(co * 8 + ci) * KK + tap] = acc[tap];
    if (ci == 0) red[half][COG * 8 * KK + co] = accb;
    __syncthreads();
    constexpr int PER = COG * 8 * KK + COG;
    float* dst = partial + (((long long)blockIdx.x * gridDim.y + icb) * gridDim.z + og) * PER;
    for (int e = tid; e < PER; e += 256) dst[e] = red[0][e] + red[1][e];
}

template <int KS>
__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, float* __restrict__ db,
                                    int cin, int cout, int G, int n_icb, int n_og, int accumulate) {
    constexpr int KK = KS * KS;
    constexpr int PER = COG * 8 * KK + COG;
    const int total_w = cout * cin * KK;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < total_w) {
        const int tap = idx % KK, c = (idx / KK) % cin, o = idx / (KK * cin);
        const int icb = c / 8, ci = c % 8, og = o / COG, co = o % COG;
        float s = 0.f;
        for (int g = 0; g < G; ++g) s += partial[(((long long)g * n_icb + icb) * n_og + og) * PER + (co * 8 + ci) * KK + tap];
        dw[idx] = accumulate ? dw[idx] + s : s;
    } else if (idx < total_w + cout && db != nullptr) {
        const int o = idx - total_w;
        const int og = o / COG, co = o % COG;
        float s = 0.f;
        for (int g = 0; g < G; ++g) s += partial[(((long long)g * n_icb + 0) * n_og + og) * PER + COG * 8 * KK + co];
        db[o] = accumulate ? db[o] + s : s;
    }
}

// ------------------------------------------------------------------------------------------ host
template <typename T, int KS>
static int launch_conv(bool dgrad, const TV& tin, const TV& tout, const TV& tmask, const float* w, const float* bias,
                       int cin, int cout, int relu, uint64_t mask_bits, uint64_t accum_bits, hipStream_t st) {
    const int tiles_x = cdiv(tout.ws, TILE), tiles_y = cdiv(tout.hs, TILE);
    const int n_out = dgrad ? cin : cout;
    dim3 grid(tiles_x * tiles_y, tout.n, cdiv(n_out, COG));
    if (dgrad)
        hipLaunchKernelGGL((conv_valu_kernel<T, KS, true>), grid, dim3(256), 0, st, tin, tout, tmask, w, bias, cin, cout,
                           relu, mask_bits, accum_bits, tiles_x);
    else
        hipLaunchKernelGGL((conv_valu_kernel<T, KS, false>), grid, dim3(256), 0, st, tin, tout, tmask, w, bias, cin, cout,
                           relu, mask_bits, accum_bits, tiles_x);
    return check_launch(dgrad ? "conv_valu dgrad" : "conv_valu fwd");
}

int conv_valu(bool dgrad, int dtype, int ks, const TV& tin, const TV& tout, const TV& tmask, const float* w,
              const float* bias, int cin, int cout, int relu, uint64_t mask_bits, uint64_t accum_bits, hipStream_t st) {
    if (dtype == MMIF_F32) {
        if (ks == 3) return launch_conv<float, 3>(dgrad, tin, tout, tmask, w, bias, cin, cout, relu, mask_bits, accum_bits, st);
        return launch_conv<float, 1>(dgrad, tin, tout, tmask, w, bias, cin, cout, relu, mask_bits, accum_bits, st);
    }
    if (ks == 3) return launch_conv<bf16_t, 3>(dgrad, tin, tout, tmask, w, bias, cin, cout, relu, mask_bits, accum_bits, st);
    return launch_conv<bf16_t, 1>(dgrad, tin, tout, tmask, w, bias, cin, cout, relu, mask_bits, accum_bits, st);
}

constexpr int WGRAD_G = 128;  // pixel-tile groups (partials per output element)

size_t wgrad_valu_workspace(int cin, int cout, int ks) {
    const size_t per = (size_t)COG * 8 * ks * ks + COG;
    return (size_t)WGRAD_G * cdiv(cin, 8) * cdiv(cout, COG) * per * sizeof(float);
}

template <typename T, int KS>
static int launch_wgrad(const TV& tx, const TV& tg, float* dw, float* db, int cin, int cout, int accumulate,
                        float* ws, hipStream_t st) {
    const int tiles_x = cdiv(tx.w, TILE), tiles_y = cdiv(tx.h, TILE);
    const int tpi = tiles_x * tiles_y, total = tpi * tx.n;
    const int G = total < WGRAD_G ? total : WGRAD_G;
    const int n_icb = cdiv(cin, 8), n_og = cdiv(cout, COG);
    hipLaunchKernelGGL((wgrad_valu_kernel<T, KS>), dim3(G, n_icb, n_og), dim3(256), 0, st, tx, tg, ws, cin, cout, tiles_x,
                       tpi, total);
    int rc = check_launch("wgrad_valu");
    if (rc) return rc;
    const int n = cout * cin * KS * KS + cout;
    hipLaunchKernelGGL((wgrad_reduce_kernel<KS>), dim3(cdiv(n, 256)), dim3(256), 0, st, ws, dw, db, cin, cout, G, n_icb,
                       n_og, accumulate);
    return check_launch("wgrad_reduce");
}

int wgrad_valu(int dtype, int ks, const TV& tx, const TV& tg, float* dw, float* db, int cin, int cout, int accumulate,
               float* ws, hipStream_t st) {
    if (dtype == MMIF_F32) return ks == 3 ? launch_wgrad<float, 3>(tx, tg, dw, db, cin, cout, accumulate, ws, st)
                                          : launch_wgrad<float, 1>(tx, tg, dw, db, cin, cout, accumulate, ws, st);
    return ks == 3 ? launch_wgrad<bf16_t, 3>(tx, tg, dw, db, cin, cout, accumulate, ws, st)
                   : launch_wgrad<bf16_t, 1>(tx, tg, dw, db, cin, cout, accumulate, ws, st);
}

}  // namespace mmif
