// The SSIMLoss modes beyond 'ssim' and TVLoss (reference core/loss.py):
//   'w-ssim'   :259-266  per-sample weights gamma_b = mean sigma1 / (mean sigma1 + mean sigma2), sigma = clamp(var(source), 1e-4)
//   'msw-ssim' :211-237  windows 11 / 9 / 7 / 5 / 3 (sigma 1.5 | 0.15 (k - 1), :33-39), per-PIXEL gamma maps, mean over the 5 windows
//   'ms-ssim'  :113-160  5-level avg-pool pyramid (reflect pad of odd sizes), prod_i clamp(v_i, eps)^w_i of the per-sample cs (levels
//                        0..3) / ssim (level 4) means
//   TVLoss     :347-358  NormLoss('l1'|'l2') of the vertical plus of the horizontal first differences
// Same machinery as csrc/loss.hip (separable window through LDS, closed-form gradient via the adjoint correlation, block sums +
// fixed-order second stage), generalised: window size is a template parameter, every map pixel carries the two pair weights
// (constant x per-sample x per-pixel gamma), the quantity is ssim or cs.  gamma depends only on the source images, so the
// gradient w.r.t. the fused image flows through the ssim / cs maps alone, exactly as in the reference's autograd graph.
// Not on the train.py hot path ('ssim' keeps its own tuned kernels); value + d/dimgf in one call like the other losses.
#include <math.h>

#include "common.hpp"

namespace mmif {

struct WinK {
    float t[11];
};

constexpr int MT_ = 16;   // tile edge

static void make_window(WinK& w, int k) {
    // core/loss.py:24-39: taps in double -> float32, divided by their float32 sum
    const double sigma = k == 11 ? 1.5 : 0.15 * (k - 1);
    float g[11];
    double sum = 0.0;
    for (int i = 0; i < k; ++i) {
        g[i] = (float)exp(-(double)((i - k / 2) * (i - k / 2)) / (2.0 * sigma * sigma));
        sum += (double)g[i];
    }
    const float fs = (float)sum;
    for (int i = 0; i < 11; ++i) w.t[i] = i < k ? g[i] / fs : 0.f;
}

// ------------------------------------------------------------------ pass 1: values + weighted adjoint inputs
// per map pixel and pair p (x1|x2 vs f): Q_p = ssim or cs; effective weight g_p = c_p * (w_p ? w_p[sample] : 1) * (pixel gamma?)
// partial[q][sample][block]: q0 = sum g_1 Q_1, q1 = sum g_2 Q_2, q2 = sum sigma_1, q3 = sum sigma_2   (sigma = clamp(var x, 1e-4))
// maps (optional) [4][n][Hm][Wm]: A = g1 A1 + g2 A2, B = g1 B1 + g2 B2, C1 = g1 C1', C2 = g2 C2'
template <int WIN>
__global__ __launch_bounds__(256) void ssimx_stats_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                          const float* __restrict__ f, int H, int W, WinK win, float C1, float C2,
                                                          float ca, float cb, const float* __restrict__ wa, const float* __restrict__ wb,
                                                          int pix_gamma, int quantity_cs, float* __restrict__ maps,
                                                          float* __restrict__ partial, int tiles_x) {
    constexpr int LIN = MT_ + WIN - 1;
    __shared__ float in[3][LIN][LIN + 1];
    __shared__ float hb[8][LIN][MT_ + 1];
    __shared__ float red[16];
    const int Hm = H - WIN + 1, Wm = W - WIN + 1;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int mx0 = (blockIdx.x % tiles_x) * MT_, my0 = (blockIdx.x / tiles_x) * MT_;
    const int in_ = blockIdx.y;
    const long long ibase = (long long)in_ * H * W;
    for (int e = tid; e < LIN * LIN; e += 256) {
        const int py = e / LIN, px = e % LIN;
        const int y = my0 + py, x = mx0 + px;
        const bool ok = (y < H) && (x < W);
        const long long i = ibase + (long long)y * W + x;
        in[0][py][px] = ok ? x1[i] : 0.f;
        in[1][py][px] = ok ? x2[i] : 0.f;
        in[2][py][px] = ok ? f[i] : 0.f;
    }
    __syncthreads();
    for (int e = tid; e < LIN * MT_; e += 256) {
        const int py = e / MT_, px = e % MT_;
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < WIN; ++k) {
            const float a = in[0][py][px + k], b = in[1][py][px + k], c = in[2][py][px + k], wk = win.t[k];
            s[0] = fmaf(wk, a, s[0]);
            s[1] = fmaf(wk, b, s[1]);
            s[2] = fmaf(wk, c, s[2]);
            s[3] = fmaf(wk, a * a, s[3]);
            s[4] = fmaf(wk, b * b, s[4]);
            s[5] = fmaf(wk, c * c, s[5]);
            s[6] = fmaf(wk, a * c, s[6]);
            s[7] = fmaf(wk, b * c, s[7]);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) hb[q][py][px] = s[q];
    }
    __syncthreads();
    float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < WIN; ++k) {
        const float wk = win.t[k];
#pragma unroll
        for (int q = 0; q < 8; ++q) m[q] = fmaf(wk, hb[q][ty + k][tx], m[q]);
    }
    const int my = my0 + ty, mx = mx0 + tx;
    const bool valid = (my < Hm) && (mx < Wm);
    float sums[4] = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
        const float mu1 = m[0], mu2 = m[1], muf = m[2];
        const float sf_raw = m[5] - muf * muf;
        const float sf = fmaxf(sf_raw, 0.f);
        const float kf = sf_raw > 0.f ? 1.f : 0.f;
        const float s1 = fmaxf(m[3] - mu1 * mu1, 0.f), s2 = fmaxf(m[4] - mu2 * mu2, 0.f);
        const float sg1 = fmaxf(s1, 1e-4f), sg2 = fmaxf(s2, 1e-4f);
        float g1 = ca * (wa != nullptr ? wa[in_] : 1.f), g2 = cb * (wb != nullptr ? wb[in_] : 1.f);
        if (pix_gamma) {
            const float gam = sg1 / fmaxf(sg1 + sg2, 1e-7f);
            g1 *= gam;
            g2 *= 1.f - gam;
        }
        float A = 0.f, B = 0.f, Cs[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float mux = s == 0 ? mu1 : mu2;
            const float sx = s == 0 ? s1 : s2;
            const float exf = s == 0 ? m[6] : m[7];
            const float gw = s == 0 ? g1 : g2;
            const float sxf = exf - mux * muf;
            const float m1 = 2.f * mux * muf + C1, m2 = mux * mux + muf * muf + C1;
            const float v1 = 2.f * sxf + C2, v2 = sx + sf + C2;
            float Q, Ap, Bp, Cp;
            if (quantity_cs) {
                Q = v1 / v2;
                Bp = -(Q / v2) * kf;
                Cp = 2.f / v2;
                Ap = -mux * Cp - 2.f * muf * Bp;
            } else {
                const float inv = 1.f / (m2 * v2);
                Q = m1 * v1 * inv;
                Bp = -(Q / v2) * kf;
                Cp = 2.f * m1 * inv;
                Ap = 2.f * mux * v1 * inv - 2.f * muf * Q / m2 - 2.f * mux * m1 * inv - 2.f * muf * Bp;
            }
            sums[s] = gw * Q;
            A += gw * Ap;
            B += gw * Bp;
            Cs[s] = gw * Cp;
        }
        sums[2] = sg1;
        sums[3] = sg2;
        if (maps != nullptr) {
            const long long msz = (long long)gridDim.y * Hm * Wm;
            const long long mi = ((long long)in_ * Hm + my) * Wm + mx;
            maps[mi] = A;
            maps[msz + mi] = B;
            maps[2 * msz + mi] = Cs[0];
            maps[3 * msz + mi] = Cs[1];
        }
    }
    const long long pstride = (long long)gridDim.y * gridDim.x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float bs = block_sum(sums[q], red);
        if (tid == 0) partial[q * pstride + (long long)in_ * gridDim.x + blockIdx.x] = bs;
    }
}

// ------------------------------------------------------------------ pass 2: adjoint correlation
// grad[y][x] (+)= scale * ( (G^T*A) + 2 f (G^T*B) + x1 (G^T*C1) + x2 (G^T*C2) ),  (G^T*M)[y][x] = sum_{u,v} G[u][v] M[y-u][x-v]
template <int WIN>
__global__ __launch_bounds__(256) void ssimx_grad_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                         const float* __restrict__ f, int H, int W, WinK win,
                                                         const float* __restrict__ maps, float scale, int accumulate,
                                                         float* __restrict__ grad, int tiles_x) {
    constexpr int LIN = MT_ + WIN - 1;
    __shared__ float in[4][LIN][LIN + 1];
    __shared__ float hb[4][LIN][MT_ + 1];
    const int Hm = H - WIN + 1, Wm = W - WIN + 1;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int x0 = (blockIdx.x % tiles_x) * MT_, y0 = (blockIdx.x / tiles_x) * MT_;
    const int in_ = blockIdx.y;
    const long long msz = (long long)gridDim.y * Hm * Wm;
    for (int e = tid; e < LIN * LIN; e += 256) {
        const int py = e / LIN, px = e % LIN;
        const int my = y0 - (WIN - 1) + py, mx = x0 - (WIN - 1) + px;
        const bool ok = my >= 0 && my < Hm && mx >= 0 && mx < Wm;
        const long long mi = ((long long)in_ * Hm + my) * Wm + mx;
#pragma unroll
        for (int q = 0; q < 4; ++q) in[q][py][px] = ok ? maps[q * msz + mi] : 0.f;
    }
    __syncthreads();
    for (int e = tid; e < LIN * MT_; e += 256) {
        const int py = e / MT_, px = e % MT_;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < WIN; ++k) {
            const float wk = win.t[k];
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] = fmaf(wk, in[q][py][px + (WIN - 1) - k], s[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) hb[q][py][px] = s[q];
    }
    __syncthreads();
    float m[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < WIN; ++k) {
        const float wk = win.t[k];
#pragma unroll
        for (int q = 0; q < 4; ++q) m[q] = fmaf(wk, hb[q][ty + (WIN - 1) - k][tx], m[q]);
    }
    const int y = y0 + ty, x = x0 + tx;
    if (y < H && x < W) {
        const long long i = ((long long)in_ * H + y) * W + x;
        const float g = scale * (m[0] + 2.f * f[i] * m[1] + x1[i] * m[2] + x2[i] * m[3]);
        grad[i] = accumulate ? grad[i] + g : g;
    }
}

// sums[q][b] = sum over the blocks of sample b (one block per (q, b)); deterministic
__global__ __launch_bounds__(256) void row_sums_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ sums) {
    __shared__ float red[16];
    const float* p = partial + (long long)blockIdx.x * nblk;
    float s = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) s += p[i];
    const float t = block_sum(s, red);
    if (threadIdx.x == 0) sums[blockIdx.x] = t;
}

// w-ssim: gamma_b from the per-sample sigma means (sums[2][b], sums[3][b])
__global__ void wssim_gamma_kernel(const float* __restrict__ sums, int n, float inv_nmap, float* __restrict__ wa, float* __restrict__ wb) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    const float a = sums[2 * n + b] * inv_nmap, c = sums[3 * n + b] * inv_nmap;
    const float g = a / fmaxf(a + c, 1e-7f);
    wa[b] = g;
    wb[b] = 1.f - g;
}

// loss (=|-=) : first: loss = weight * (1 - val / K), else loss -= weight * val / K;  val = inv_count * sum_b (sums[0][b] + sums[1][b])
__global__ void value_finish_kernel(const float* __restrict__ sums, int n, float inv_count, float weight, float inv_k, int first,
                                    float* __restrict__ loss) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float v = 0.f;
    for (int b = 0; b < n; ++b) v += sums[b] + sums[n + b];
    v *= inv_count;
    loss[0] = first ? weight * (1.f - v * inv_k) : loss[0] - weight * v * inv_k;
}

// ------------------------------------------------------------------ ms-ssim pyramid
// dst[y][x] = mean of the 2x2 cell of the source reflect-padded to even size (core/loss.py:146-153)
__global__ void avg_pool_pad_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int h, int w) {
    const int ho = (h + 1) / 2, wo = (w + 1) / 2;
    const long long total = (long long)n * ho * wo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = i % wo, y = (i / wo) % ho, b = i / ((long long)wo * ho);
        const float* p = src + (long long)b * h * w;
        const int y0 = 2 * y, y1 = (2 * y + 1 < h) ? 2 * y + 1 : h - 2, x0 = 2 * x, x1 = (2 * x + 1 < w) ? 2 * x + 1 : w - 2;
        dst[i] = 0.25f * ((p[(long long)y0 * w + x0] + p[(long long)y0 * w + x1]) + (p[(long long)y1 * w + x0] + p[(long long)y1 * w + x1]));
    }
}
// adjoint, accumulated into the finer level's gradient
__global__ void avg_pool_pad_bwd_kernel(const float* __restrict__ gc, float* __restrict__ gf, int n, int h, int w) {
    const int ho = (h + 1) / 2, wo = (w + 1) / 2;
    const long long total = (long long)n * h * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = i % w, y = (i / w) % h, b = i / ((long long)w * h);
        const float* g = gc + (long long)b * ho * wo;
        const int cy[2] = {y >> 1, ((h & 1) && y == h - 2) ? (h >> 1) : -1};
        const int cx[2] = {x >> 1, ((w & 1) && x == w - 2) ? (w >> 1) : -1};
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (cy[a] >= 0 && cx[c] >= 0) s += g[(long long)cy[a] * wo + cx[c]];
        gf[i] += 0.25f * s;
    }
}

struct MsMeta {
    float inv_nmap[5];
    float w[5];
};
// vals[p][l][b] = per-sample SUMS of cs (l < 4) / ssim (l = 4) of pair p; -> loss and the per-sample weights of pass 2
__global__ void ms_weights_kernel(const float* __restrict__ vals, int n, MsMeta meta, float weight, float* __restrict__ loss,
                                  float* __restrict__ wts) {
    __shared__ float red[16];
    float acc = 0.f;
    for (int b = threadIdx.x; b < n; b += blockDim.x) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float v[5], raw[5], ms = 1.f;
#pragma unroll
            for (int l = 0; l < 5; ++l) {
                raw[l] = vals[(p * 5 + l) * n + b] * meta.inv_nmap[l];
                v[l] = fmaxf(raw[l], 1e-7f);
                ms *= powf(v[l], meta.w[l]);
            }
            acc += ms;
#pragma unroll
            for (int l = 0; l < 5; ++l)
                wts[(p * 5 + l) * n + b] = raw[l] > 1e-7f ? -weight * 0.5f / (float)n * meta.w[l] * ms / v[l] * meta.inv_nmap[l] : 0.f;
        }
    }
    const float t = block_sum(acc, red);
    if (threadIdx.x == 0) loss[0] = weight * (1.f - 0.5f * t / (float)n);
}

// ------------------------------------------------------------------ TV loss
__global__ __launch_bounds__(256) void tv_loss_kernel(const float* __restrict__ x, int n, int H, int W, float sh, float sw, int l2,
                                                       float* __restrict__ grad, float* __restrict__ partial) {
    __shared__ float red[16];
    const long long total = (long long)n * H * W;
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int xx = i % W, yy = (i / W) % H;
        const float c = x[i];
        auto term = [&](float d, float sc) { return l2 ? 2.f * d * sc : (d > 0.f ? sc : (d < 0.f ? -sc : 0.f)); };
        float g = 0.f;
        if (yy >= 1) {
            const float d = c - x[i - W];
            s += sh * (l2 ? d * d : fabsf(d));
            g += term(d, sh);
        }
        if (yy + 1 < H) g -= term(x[i + W] - c, sh);
        if (xx >= 1) {
            const float d = c - x[i - 1];
            s += sw * (l2 ? d * d : fabsf(d));
            g += term(d, sw);
        }
        if (xx + 1 < W) g -= term(x[i + 1] - c, sw);
        if (grad) grad[i] = g;
    }
    const float t = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}
__global__ void tv_finish_kernel(const float* __restrict__ partial, int np, float* __restrict__ loss) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < np; i += blockDim.x) s += partial[i];
    const float t = block_sum(s, red);
    if (threadIdx.x == 0) loss[0] = t;
}

// ------------------------------------------------------------------ host helpers
struct SsimArgs {
    const float *x1, *x2, *f;
    int n, h, w;
    float C1, C2;
    hipStream_t st;
};

template <int WIN>
static int run_stats(const SsimArgs& a, float ca, float cb, const float* wa, const float* wb, int pix, int cs, float* maps,
                     float* partial, float* sums) {
    WinK win;
    make_window(win, WIN);
    const int Hm = a.h - WIN + 1, Wm = a.w - WIN + 1;
    const int tmx = cdiv(Wm, MT_), tmy = cdiv(Hm, MT_);
    hipLaunchKernelGGL((ssimx_stats_kernel<WIN>), dim3(tmx * tmy, a.n), dim3(256), 0, a.st, a.x1, a.x2, a.f, a.h, a.w, win, a.C1, a.C2, ca,
                       cb, wa, wb, pix, cs, maps, partial, tmx);
    if (int rc = check_launch("ssimx_stats")) return rc;
    hipLaunchKernelGGL(row_sums_kernel, dim3(4 * a.n), dim3(256), 0, a.st, partial, tmx * tmy, sums);
    return check_launch("ssimx_row_sums");
}
template <int WIN>
static int run_grad(const SsimArgs& a, const float* maps, float scale, int accumulate, float* grad) {
    WinK win;
    make_window(win, WIN);
    const int tx = cdiv(a.w, MT_), ty = cdiv(a.h, MT_);
    hipLaunchKernelGGL((ssimx_grad_kernel<WIN>), dim3(tx * ty, a.n), dim3(256), 0, a.st, a.x1, a.x2, a.f, a.h, a.w, win, maps, scale,
                       accumulate, grad, tx);
    return check_launch("ssimx_grad");
}
static int ew_grid(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace mmif

using namespace mmif;

// workspace layout (floats): [partial 4*n*tiles | sums 4n | wa n | wb n | vals 10n | wts 10n | maps 4*n*h*w | pyramid ...]
extern "C" size_t mmif_ssim_loss_mode_workspace(int32_t n, int32_t h, int32_t w, int32_t mode) {
    const size_t tiles = (size_t)cdiv(h, MT_) * cdiv(w, MT_);
    size_t fl = 4 * n * tiles + 26 * (size_t)n + 64 + 4 * (size_t)n * h * w;
    if (mode == 2) {   // ms-ssim: 3 pyramid images + 1 gradient per coarser level
        size_t hl = h, wl = w;
        for (int l = 1; l < 5; ++l) {
            hl = (hl + 1) / 2;
            wl = (wl + 1) / 2;
            fl += 4 * (size_t)n * hl * wl + 64;
        }
    }
    return fl * sizeof(float);
}

extern "C" int mmif_ssim_loss_mode(const float* img1, const float* img2, const float* imgf, int32_t n, int32_t h, int32_t w, float weight,
                                   float data_range, int32_t mode, float* loss_out, float* grad_out, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    if (mode < 1 || mode > 3) {
        set_error("only supported ['ssim', 'w-ssim', 'ms-ssim', 'msw-ssim'] mode");   // core/loss.py:279-282 ('ssim': mmif_ssim_loss)
        return MMIF_EINVAL;
    }
    MMIF_REQUIRE(img1 && img2 && imgf && loss_out && workspace, "ssim_loss_mode: NULL argument");
    MMIF_REQUIRE(n > 0 && h >= 11 && w >= 11, "ssim_loss_mode: image smaller than the 11x11 window (%dx%d)", h, w);
    if (workspace_bytes < mmif_ssim_loss_mode_workspace(n, h, w, mode)) {
        set_error("ssim_loss_mode: workspace too small");
        return MMIF_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const float C1 = (0.01f * data_range) * (0.01f * data_range), C2 = (0.03f * data_range) * (0.03f * data_range);
    const size_t tiles = (size_t)cdiv(h, MT_) * cdiv(w, MT_);
    float* partial = (float*)workspace;
    float* sums = partial + 4 * n * tiles;
    float* wa = sums + 4 * n;
    float* wb = wa + n;
    float* vals = wb + n;
    float* wts = vals + 10 * n;
    float* maps = wts + 10 * n;
    SsimArgs a{img1, img2, imgf, n, h, w, C1, C2, st};
    if (mode == 1) {   // w-ssim
        const float cnt = (float)(h - 10) * (float)(w - 10);
        if (int rc = run_stats<11>(a, 1.f, 1.f, nullptr, nullptr, 0, 0, nullptr, partial, sums)) return rc;   // sigma means
        hipLaunchKernelGGL(wssim_gamma_kernel, dim3(cdiv(n, 64)), dim3(64), 0, st, sums, n, 1.f / cnt, wa, wb);
        if (int rc = check_launch("wssim_gamma")) return rc;
        if (int rc = run_stats<11>(a, 1.f, 1.f, wa, wb, 0, 0, grad_out ? maps : nullptr, partial, sums)) return rc;
        hipLaunchKernelGGL(value_finish_kernel, dim3(1), dim3(64), 0, st, sums, n, 1.f / ((float)n * cnt), weight, 1.f, 1, loss_out);
        if (int rc = check_launch("wssim_finish")) return rc;
        if (grad_out) return run_grad<11>(a, maps, -weight / ((float)n * cnt), 0, grad_out);
        return MMIF_OK;
    }
    if (mode == 3) {   // msw-ssim
        int first = 1;
#define MSW_STEP(K)                                                                                                              \
    {                                                                                                                            \
        const float cnt = (float)(h - K + 1) * (float)(w - K + 1);                                                               \
        if (int rc = run_stats<K>(a, 1.f, 1.f, nullptr, nullptr, 1, 0, grad_out ? maps : nullptr, partial, sums)) return rc;      \
        hipLaunchKernelGGL(value_finish_kernel, dim3(1), dim3(64), 0, st, sums, n, 1.f / ((float)n * cnt), weight, 0.2f, first, \
                           loss_out);                                                                                            \
        if (int rc = check_launch("mswssim_finish")) return rc;                                                                  \
        if (grad_out)                                                                                                            \
            if (int rc = run_grad<K>(a, maps, -weight * 0.2f / ((float)n * cnt), first ? 0 : 1, grad_out)) return rc;            \
        first = 0;                                                                                                               \
    }
        MSW_STEP(11) MSW_STEP(9) MSW_STEP(7) MSW_STEP(5) MSW_STEP(3)
#undef MSW_STEP
        return MMIF_OK;
    }
    // ---- ms-ssim
    int hs[5], wsz[5];
    hs[0] = h;
    wsz[0] = w;
    for (int l = 1; l < 5; ++l) {
        hs[l] = (hs[l - 1] + 1) / 2;
        wsz[l] = (wsz[l - 1] + 1) / 2;
    }
    MMIF_REQUIRE(hs[4] >= 11 && wsz[4] >= 11, "ssim_loss_mode: ms-ssim needs images of at least 161x161 (11x11 window on level 4); got %dx%d", h, w);
    float* pyr = maps + 4 * (size_t)n * h * w;
    const float* lx1[5] = {img1};
    const float* lx2[5] = {img2};
    const float* lf[5] = {imgf};
    float* lg[5] = {grad_out};
    for (int l = 1; l < 5; ++l) {
        const size_t sz = (size_t)n * hs[l] * wsz[l];
        float* p1 = pyr;
        float* p2 = p1 + sz;
        float* pf = p2 + sz;
        lg[l] = pf + sz;
        pyr = lg[l] + sz;
        const int grid = ew_grid((long long)sz);
        hipLaunchKernelGGL(avg_pool_pad_kernel, dim3(grid), dim3(256), 0, st, lx1[l - 1], p1, n, hs[l - 1], wsz[l - 1]);
        hipLaunchKernelGGL(avg_pool_pad_kernel, dim3(grid), dim3(256), 0, st, lx2[l - 1], p2, n, hs[l - 1], wsz[l - 1]);
        hipLaunchKernelGGL(avg_pool_pad_kernel, dim3(grid), dim3(256), 0, st, lf[l - 1], pf, n, hs[l - 1], wsz[l - 1]);
        if (int rc = check_launch("avg_pool_pad")) return rc;
        lx1[l] = p1;
        lx2[l] = p2;
        lf[l] = pf;
    }
    MsMeta meta;
    const float msw[5] = {0.0448f, 0.2856f, 0.3001f, 0.2363f, 0.1333f};
    for (int l = 0; l < 5; ++l) {
        meta.w[l] = msw[l];
        meta.inv_nmap[l] = 1.f / ((float)(hs[l] - 10) * (float)(wsz[l] - 10));
        SsimArgs al{lx1[l], lx2[l], lf[l], n, hs[l], wsz[l], C1, C2, st};
        if (int rc = run_stats<11>(al, 1.f, 1.f, nullptr, nullptr, 0, l < 4 ? 1 : 0, nullptr, partial, sums)) return rc;
        // vals[p][l][b] <- sums[p][b]
        (void)hipMemcpyAsync(vals + (0 * 5 + l) * n, sums, n * sizeof(float), hipMemcpyDeviceToDevice, st);
        (void)hipMemcpyAsync(vals + (1 * 5 + l) * n, sums + n, n * sizeof(float), hipMemcpyDeviceToDevice, st);
    }
    hipLaunchKernelGGL(ms_weights_kernel, dim3(1), dim3(256), 0, st, vals, n, meta, weight, loss_out, wts);
    if (int rc = check_launch("ms_weights")) return rc;
    if (!grad_out) return MMIF_OK;
    for (int l = 4; l >= 0; --l) {
        SsimArgs al{lx1[l], lx2[l], lf[l], n, hs[l], wsz[l], C1, C2, st};
        if (int rc = run_stats<11>(al, 1.f, 1.f, wts + (0 * 5 + l) * n, wts + (1 * 5 + l) * n, 0, l < 4 ? 1 : 0, maps, partial, sums)) return rc;
        if (int rc = run_grad<11>(al, maps, 1.f, 0, lg[l])) return rc;
        if (l < 4) {
            hipLaunchKernelGGL(avg_pool_pad_bwd_kernel, dim3(ew_grid((long long)n * hs[l] * wsz[l])), dim3(256), 0, st, lg[l + 1], lg[l], n,
                               hs[l], wsz[l]);
            if (int rc = check_launch("avg_pool_pad_bwd")) return rc;
        }
    }
    return MMIF_OK;
}

// out[q][b] = sums[qsel][b] * scale  (tiny gather used by mmif_ssim_terms)
__global__ void pick_row_kernel(const float* __restrict__ sums, int n, int qsel, float scale, float* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n) out[b] = sums[qsel * n + b] * scale;
}

/* calc_ssim core/loss.py:52-110 with size_average=True: per-sample means of the ssim map, the cs map and the clamped source
 * variance sigma of (img1, img2) for a window of 3/5/7/9/11 taps -> out[3][n] (device).  Values only. */
extern "C" int mmif_ssim_terms(const float* img1, const float* img2, int32_t n, int32_t h, int32_t w, int32_t win_size, float data_range,
                               float* out, void* workspace, size_t workspace_bytes, void* stream) {
    MMIF_REQUIRE(img1 && img2 && out && workspace, "ssim_terms: NULL argument");
    MMIF_REQUIRE(win_size == 3 || win_size == 5 || win_size == 7 || win_size == 9 || win_size == 11, "ssim_terms: win_size must be 3, 5, 7, 9 or 11 (got %d)", win_size);
    MMIF_REQUIRE(n > 0 && h >= win_size && w >= win_size, "ssim_terms: image smaller than the window (%dx%d)", h, w);
    if (workspace_bytes < mmif_ssim_loss_mode_workspace(n, h, w, 1)) {
        set_error("ssim_terms: workspace too small");
        return MMIF_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const float C1 = (0.01f * data_range) * (0.01f * data_range), C2 = (0.03f * data_range) * (0.03f * data_range);
    const size_t tiles = (size_t)cdiv(h, MT_) * cdiv(w, MT_);
    float* partial = (float*)workspace;
    float* sums = partial + 4 * n * tiles;
    SsimArgs a{img1, img1, img2, n, h, w, C1, C2, st};   // pair 1 = (img1, img2); pair 2 unused
    const float inv = 1.f / ((float)(h - win_size + 1) * (float)(w - win_size + 1));
    for (int cs = 0; cs < 2; ++cs) {
        int rc;
        switch (win_size) {
            case 3: rc = run_stats<3>(a, 1.f, 0.f, nullptr, nullptr, 0, cs, nullptr, partial, sums); break;
            case 5: rc = run_stats<5>(a, 1.f, 0.f, nullptr, nullptr, 0, cs, nullptr, partial, sums); break;
            case 7: rc = run_stats<7>(a, 1.f, 0.f, nullptr, nullptr, 0, cs, nullptr, partial, sums); break;
            case 9: rc = run_stats<9>(a, 1.f, 0.f, nullptr, nullptr, 0, cs, nullptr, partial, sums); break;
            default: rc = run_stats<11>(a, 1.f, 0.f, nullptr, nullptr, 0, cs, nullptr, partial, sums); break;
        }
        if (rc) return rc;
        hipLaunchKernelGGL(pick_row_kernel, dim3(cdiv(n, 64)), dim3(64), 0, st, sums, n, 0, inv, out + cs * n);
        if (cs == 1) hipLaunchKernelGGL(pick_row_kernel, dim3(cdiv(n, 64)), dim3(64), 0, st, sums, n, 2, inv, out + 2 * n);
        if (int rc2 = check_launch("ssim_terms pick")) return rc2;
    }
    return MMIF_OK;
}

extern "C" size_t mmif_tv_loss_workspace(void) { return 4096 * sizeof(float); }

extern "C" int mmif_tv_loss(const float* x, int32_t n, int32_t h, int32_t w, float weight, int32_t l2, float* loss_out, float* grad_out,
                            void* workspace, size_t workspace_bytes, void* stream) {
    MMIF_REQUIRE(x && loss_out && workspace, "tv_loss: NULL argument");
    MMIF_REQUIRE(n > 0 && h >= 2 && w >= 2, "tv_loss: needs at least 2x2 images");
    if (workspace_bytes < mmif_tv_loss_workspace()) {
        set_error("tv_loss: workspace too small");
        return MMIF_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)n * h * w;
    const int nb = ew_grid(total);
    const float sh = weight / ((float)n * (h - 1) * w), sw = weight / ((float)n * h * (w - 1));
    hipLaunchKernelGGL(tv_loss_kernel, dim3(nb), dim3(256), 0, st, x, n, h, w, sh, sw, l2, grad_out, (float*)workspace);
    if (int rc = check_launch("tv_loss")) return rc;
    hipLaunchKernelGGL(tv_finish_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, nb, loss_out);
    return check_launch("tv_finish");
}
