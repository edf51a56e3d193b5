// General ConvLayer primitives for the nets outside the PFNet/DenseFuse hot path: kernel sizes 3/5/7
// (DeepFuse core/model.py:152-158, IFCNN, PMGI), stride 2 (DBNet :219-221, NestFuse down_mode='stride' :338-340) and
// ConvTranspose2d(k=3, s=2, p=1, output_padding=1) (SEDRFuse :258-259; core/block.py:67-76).  Plain NCHW fp32 tensors (these
// nets run layer by layer through autograd, so the boundary layout IS torch's), fp32 FMA arithmetic, LDS-tiled:
//
//   gconv_fwd_kernel   y[o][oy][ox]  = b[o] + sum_{c,u,v} W[o][c][u][v] * Xpad[c][oy*s+u-p][ox*s+v-p]          (reflect | zero padding)
//   gconv_tg_kernel    z[c][Y][X]   (= b[c]) + sum_{o,u,v} W[o][c][u][v] * G[o][(Y+off-u)/s][(X+off-v)/s]        (where divisible, in range)
//                         off = 0 on the padded domain [h+2p][w+2p]  -> input gradient of the conv before the reflect fold
//                         off = p on the output domain of a ConvTranspose2d -> its forward
//   gconv_fold_kernel  dx[y][x]      = sum over the reflect images of (y, x) in the padded domain                 (adjoint of ReflectionPad2d(p))
//   gconv_wg_kernel    dW[o][c][u][v] = sum_{n,oy,ox} G[o][oy][ox] * Xpad[c][oy*s+u-p][ox*s+v-p], db[o] = sum G   (+ fixed-order reduce)
//
// ConvTranspose2d(x; W[ci][co]) = tg(G := x, W, off = p); its input gradient = fwd(X := gy, W as [o = ci][c = co], zero padding,
// stride s); its weight gradient = wg(G := x, X := gy).  Everything deterministic (no atomics).
#include "common.hpp"

namespace mmif {

constexpr int GT = 16;     // output tile edge (256 threads = 16 x 16 pixels)
constexpr int G_OG = 8;    // output channels per block pass
constexpr int G_CC = 4;    // input channels staged per chunk

struct GC {
    int n, cin, cout;      // cin: channels of X / z, cout: channels of y / G   (nn.Conv2d naming)
    int hi, wi, ho, wo;    // X extent, G extent
    int k, s, p, reflect;
};

__device__ inline float gx_load(const float* __restrict__ plane, int h, int w, int y, int x, int reflect) {
    if (reflect) {
        y = min(max(reflect_idx(y, h), 0), h - 1);
        x = min(max(reflect_idx(x, w), 0), w - 1);
        return plane[(long long)y * w + x];
    }
    return (y >= 0 && y < h && x >= 0 && x < w) ? plane[(long long)y * w + x] : 0.f;
}

// ------------------------------------------------------------------ forward
// grid (tiles, cout groups of 8, n); LDS: x tile [G_CC][IT][IT] (IT = 15 s + k), weights [G_CC][k*k][8]
__global__ __launch_bounds__(256) void gconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ y, GC q, int relu, int tiles_x) {
    extern __shared__ float sm[];
    const int kk = q.k * q.k, IT = (GT - 1) * q.s + q.k;
    float* xs = sm;                          // [G_CC][IT][IT]
    float* ws = sm + G_CC * IT * IT;         // [G_CC][kk][G_OG]
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int ox0 = (blockIdx.x % tiles_x) * GT, oy0 = (blockIdx.x / tiles_x) * GT;
    const int og = blockIdx.y * G_OG, in_ = blockIdx.z;
    float acc[G_OG];
#pragma unroll
    for (int o = 0; o < G_OG; ++o) acc[o] = 0.f;
    for (int c0 = 0; c0 < q.cin; c0 += G_CC) {
        __syncthreads();
        for (int e = tid; e < G_CC * IT * IT; e += 256) {
            const int cc = e / (IT * IT), r = e % (IT * IT), iy = r / IT, ix = r % IT;
            const int c = c0 + cc;
            xs[e] = c < q.cin ? gx_load(x + ((long long)in_ * q.cin + c) * q.hi * q.wi, q.hi, q.wi, oy0 * q.s + iy - q.p, ox0 * q.s + ix - q.p, q.reflect)
                              : 0.f;
        }
        for (int e = tid; e < G_CC * kk * G_OG; e += 256) {
            const int o = e % G_OG, t = (e / G_OG) % kk, cc = e / (G_OG * kk);
            const int oc = og + o, c = c0 + cc;
            ws[e] = (oc < q.cout && c < q.cin) ? w[((long long)oc * q.cin + c) * kk + t] : 0.f;
        }
        __syncthreads();
        for (int cc = 0; cc < G_CC; ++cc) {
            const float* xp = xs + cc * IT * IT + (ty * q.s) * IT + tx * q.s;
            const float* wp = ws + cc * kk * G_OG;
            for (int u = 0; u < q.k; ++u)
                for (int v = 0; v < q.k; ++v) {
                    const float xv = xp[u * IT + v];
                    const float4 w0 = *reinterpret_cast<const float4*>(wp + (u * q.k + v) * G_OG);
                    const float4 w1 = *reinterpret_cast<const float4*>(wp + (u * q.k + v) * G_OG + 4);
                    acc[0] = fmaf(xv, w0.x, acc[0]); acc[1] = fmaf(xv, w0.y, acc[1]); acc[2] = fmaf(xv, w0.z, acc[2]); acc[3] = fmaf(xv, w0.w, acc[3]);
                    acc[4] = fmaf(xv, w1.x, acc[4]); acc[5] = fmaf(xv, w1.y, acc[5]); acc[6] = fmaf(xv, w1.z, acc[6]); acc[7] = fmaf(xv, w1.w, acc[7]);
                }
        }
    }
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy >= q.ho || ox >= q.wo) return;
#pragma unroll
    for (int o = 0; o < G_OG; ++o) {
        const int oc = og + o;
        if (oc >= q.cout) break;
        float r = acc[o] + (bias != nullptr ? bias[oc] : 0.f);
        if (relu) r = fmaxf(r, 0.f);
        y[(((long long)in_ * q.cout + oc) * q.ho + oy) * q.wo + ox] = r;
    }
}

// ------------------------------------------------------------------ transposed gather
// z[c][Y][X] over a [zh][zw] domain; G has q.cout channels on [q.ho][q.wo]; weights W[o][c][u][v] (o over q.cout, c over q.cin).
// grid (tiles of z, cin groups of 8, n); LDS: G tile [G_CC][R][R] with R = (15 + k - 1) / s + 2, weights [G_CC][k*k][8]
__device__ inline int floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

__global__ __launch_bounds__(256) void gconv_tg_kernel(const float* __restrict__ g, const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ z, GC q, int zh, int zw, int off, int tiles_x, int relu) {
    extern __shared__ float sm[];
    const int kk = q.k * q.k, R = (GT + q.k - 2) / q.s + 2;
    float* gs = sm;                         // [G_CC][R][R]
    float* ws = sm + G_CC * R * R;          // [G_CC][kk][G_OG]   (8 = z channels of this block)
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int X0 = (blockIdx.x % tiles_x) * GT, Y0 = (blockIdx.x / tiles_x) * GT;
    const int cg = blockIdx.y * G_OG, in_ = blockIdx.z;
    const int by = floor_div(Y0 + off - (q.k - 1), q.s), bx = floor_div(X0 + off - (q.k - 1), q.s);   // first staged G row / col
    const int Y = Y0 + ty, X = X0 + tx;
    float acc[G_OG];
#pragma unroll
    for (int c = 0; c < G_OG; ++c) acc[c] = 0.f;
    for (int o0 = 0; o0 < q.cout; o0 += G_CC) {
        __syncthreads();
        for (int e = tid; e < G_CC * R * R; e += 256) {
            const int oo = e / (R * R), r = e % (R * R), iy = by + r / R, ix = bx + r % R;
            const int o = o0 + oo;
            gs[e] = (o < q.cout && iy >= 0 && iy < q.ho && ix >= 0 && ix < q.wo)
                        ? g[(((long long)in_ * q.cout + o) * q.ho + iy) * q.wo + ix] : 0.f;
        }
        for (int e = tid; e < G_CC * kk * G_OG; e += 256) {
            const int c = e % G_OG, t = (e / G_OG) % kk, oo = e / (G_OG * kk);
            const int o = o0 + oo, ch = cg + c;
            ws[e] = (o < q.cout && ch < q.cin) ? w[((long long)o * q.cin + ch) * kk + t] : 0.f;
        }
        __syncthreads();
        for (int u = 0; u < q.k; ++u) {
            const int ry = Y + off - u;
            if (q.s > 1 && (ry < 0 || ry % q.s != 0)) continue;
            const int iy = (q.s > 1 ? ry / q.s : ry) - by;
            for (int v = 0; v < q.k; ++v) {
                const int rx = X + off - v;
                if (q.s > 1 && (rx < 0 || rx % q.s != 0)) continue;
                const int ix = (q.s > 1 ? rx / q.s : rx) - bx;
                // (staged window covers every (iy, ix) a thread of this tile can ask for; out-of-image entries are zero)
#pragma unroll
                for (int oo = 0; oo < G_CC; ++oo) {
                    const float gv = gs[oo * R * R + iy * R + ix];
                    const float* wp = ws + (oo * kk + u * q.k + v) * G_OG;
                    const float4 w0 = *reinterpret_cast<const float4*>(wp);
                    const float4 w1 = *reinterpret_cast<const float4*>(wp + 4);
                    acc[0] = fmaf(gv, w0.x, acc[0]); acc[1] = fmaf(gv, w0.y, acc[1]); acc[2] = fmaf(gv, w0.z, acc[2]); acc[3] = fmaf(gv, w0.w, acc[3]);
                    acc[4] = fmaf(gv, w1.x, acc[4]); acc[5] = fmaf(gv, w1.y, acc[5]); acc[6] = fmaf(gv, w1.z, acc[6]); acc[7] = fmaf(gv, w1.w, acc[7]);
                }
            }
        }
    }
    if (Y >= zh || X >= zw) return;
#pragma unroll
    for (int c = 0; c < G_OG; ++c) {
        const int ch = cg + c;
        if (ch >= q.cin) break;
        const float r = acc[c] + (bias != nullptr ? bias[ch] : 0.f);
        z[(((long long)in_ * q.cin + ch) * zh + Y) * zw + X] = relu ? fmaxf(r, 0.f) : r;
    }
}

// ------------------------------------------------------------------ reflect fold (adjoint of ReflectionPad2d(p)), planes = n * c
__global__ void gconv_fold_kernel(const float* __restrict__ zp, float* __restrict__ dx, long long planes, int h, int w, int p) {
    const int hp = h + 2 * p, wp = w + 2 * p;
    const long long total = planes * h * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % w), y = (int)((i / w) % h);
        const float* pl = zp + (i / ((long long)h * w)) * hp * wp;
        // images of y in padded coordinates: y + p, and the mirrors -y + p (1 <= y <= p), 2(h-1) - y + p (h-1-p <= y <= h-2)
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = y + p;
        if (y >= 1 && y <= p) ys[ny++] = p - y;
        if (y <= h - 2 && y >= h - 1 - p) ys[ny++] = 2 * (h - 1) - y + p;
        xs[nx++] = x + p;
        if (x >= 1 && x <= p) xs[nx++] = p - x;
        if (x <= w - 2 && x >= w - 1 - p) xs[nx++] = 2 * (w - 1) - x + p;
        float s = 0.f;
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b) s += pl[(long long)ys[a] * wp + xs[b]];
        dx[i] = s;
    }
}

// ------------------------------------------------------------------ weight gradient
// grid (G tile groups, cin groups of WG_CC, cout groups of 8); thread e <-> (c, tap) pairs of the group (e, e + 256), all 8 output
// channels; per pixel one x read + the 8 g values (two broadcast float4 reads).  partial[(gi, icg, ocg)][8][WG_CC][kk] then [8] db.
constexpr int WG_CC = 8;
__global__ __launch_bounds__(256) void gconv_wg_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ partial,
                                                       GC q, int tiles_x, int tiles_per_img, int total_tiles, int G) {
    extern __shared__ float sm[];
    const int kk = q.k * q.k, IT = (GT - 1) * q.s + q.k;
    float* xs = sm;                          // [WG_CC][IT][IT]
    float* gs = sm + WG_CC * IT * IT;        // [256 pixels][8]
    const int tid = threadIdx.x;
    const int gi = blockIdx.x, c0 = blockIdx.y * WG_CC, og = blockIdx.z * G_OG;
    const int nout = WG_CC * kk;             // (c, tap) pairs of this block: <= 392
    float acc[2][G_OG], accb = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int o = 0; o < G_OG; ++o) acc[r][o] = 0.f;
    int xoff[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int e = min(tid + 256 * r, nout - 1), cc = e / kk, t = e % kk;
        xoff[r] = cc * IT * IT + (t / q.k) * IT + (t % q.k);
    }
    for (int tl = gi; tl < total_tiles; tl += G) {
        const int in_ = tl / tiles_per_img, tr = tl % tiles_per_img;
        const int oy0 = (tr / tiles_x) * GT, ox0 = (tr % tiles_x) * GT;
        __syncthreads();
        for (int e = tid; e < WG_CC * IT * IT; e += 256) {
            const int cc = e / (IT * IT), r = e % (IT * IT), iy = r / IT, ix = r % IT;
            const int c = c0 + cc;
            xs[e] = c < q.cin ? gx_load(x + ((long long)in_ * q.cin + c) * q.hi * q.wi, q.hi, q.wi, oy0 * q.s + iy - q.p, ox0 * q.s + ix - q.p, q.reflect)
                              : 0.f;
        }
        {
            const int oy = oy0 + (tid >> 4), ox = ox0 + (tid & 15);
            const bool ok = oy < q.ho && ox < q.wo;
#pragma unroll
            for (int o = 0; o < G_OG; ++o) {
                const int oc = og + o;
                gs[tid * G_OG + o] = (ok && oc < q.cout) ? g[(((long long)in_ * q.cout + oc) * q.ho + oy) * q.wo + ox] : 0.f;
            }
        }
        __syncthreads();
        if (tid < G_OG && c0 == 0) {   // db: thread o sums its channel over the tile (fixed order)
            float s = 0.f;
            for (int px = 0; px < 256; ++px) s += gs[px * G_OG + tid];
            accb += s;
        }
        for (int px = 0; px < 256; ++px) {
            const float4 g0 = *reinterpret_cast<const float4*>(gs + px * G_OG);
            const float4 g1 = *reinterpret_cast<const float4*>(gs + px * G_OG + 4);
            const int pbase = ((px >> 4) * q.s) * IT + (px & 15) * q.s;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (tid + 256 * r >= nout) break;
                const float xv = xs[xoff[r] + pbase];
                acc[r][0] = fmaf(xv, g0.x, acc[r][0]); acc[r][1] = fmaf(xv, g0.y, acc[r][1]); acc[r][2] = fmaf(xv, g0.z, acc[r][2]);
                acc[r][3] = fmaf(xv, g0.w, acc[r][3]); acc[r][4] = fmaf(xv, g1.x, acc[r][4]); acc[r][5] = fmaf(xv, g1.y, acc[r][5]);
                acc[r][6] = fmaf(xv, g1.z, acc[r][6]); acc[r][7] = fmaf(xv, g1.w, acc[r][7]);
            }
        }
    }
    const int per = G_OG * WG_CC * kk + G_OG;
    float* dst = partial + (((long long)gi * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z) * per;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int e = tid + 256 * r;
        if (e >= nout) break;
#pragma unroll
        for (int o = 0; o < G_OG; ++o) dst[o * WG_CC * kk + e] = acc[r][o];   // [o][cc][tap]
    }
    if (tid < G_OG) dst[G_OG * WG_CC * kk + tid] = accb;
}

__global__ void gconv_wg_reduce(const float* __restrict__ partial, float* __restrict__ dw, float* __restrict__ db, GC q, int G, int n_icg,
                                int n_ocg) {
    const int kk = q.k * q.k, per = G_OG * WG_CC * kk + G_OG;
    const int total_w = q.cout * q.cin * kk;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total_w + q.cout) return;
    long long off;
    if (idx < total_w) {
        const int t = idx % kk, c = (idx / kk) % q.cin, o = idx / (kk * q.cin);
        off = ((long long)(c / WG_CC) * n_ocg + o / G_OG) * per + (o % G_OG) * WG_CC * kk + (c % WG_CC) * kk + t;
    } else {
        const int o = idx - total_w;
        off = ((long long)0 * n_ocg + o / G_OG) * per + G_OG * WG_CC * kk + (o % G_OG);
    }
    const long long stride = (long long)n_icg * n_ocg * per;
    float s = 0.f;
    for (int gi = 0; gi < G; ++gi) s += partial[gi * stride + off];
    if (idx < total_w) dw[idx] = s;
    else if (db != nullptr) db[idx - total_w] = s;
}

// ------------------------------------------------------------------ ReLU backward on plain tensors: g * [y > 0]
__global__ void relu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = y[i] > 0.f ? g[i] : 0.f;
}

// out[c] = sum over n and the plane of x[n][c][.]: one block per channel, fixed order (bias gradient of a ConvTranspose2d)
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int n, int c, long long hw) {
    __shared__ float red[16];
    const int ch = blockIdx.x;
    float s = 0.f;
    for (int in_ = 0; in_ < n; ++in_) {
        const float* pl = x + ((long long)in_ * c + ch) * hw;
        for (long long i = threadIdx.x; i < hw; i += 256) s += pl[i];
    }
    const float t = block_sum(s, red);
    if (threadIdx.x == 0) out[ch] = t;
}

// ------------------------------------------------------------------ depth-wise conv (groups == channels), k in {1, 3}, stride 1, reflect | zero padding
// (Res2Fusion's Res2ConvBlock.dwconvs, reference core/block.py:317-325: ConvLayer(width, width, ksize, groups=width, bias=False, act=None))
__global__ void dwconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
                                  long long planes, int c, int h, int wd, int k, int reflect) {
    const int p = k / 2, kk = k * k;
    const long long total = planes * h * wd;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % wd), yy = (int)((i / wd) % h);
        const long long plane = i / ((long long)h * wd);
        const int ch = (int)(plane % c);
        const float* pl = x + plane * h * wd;
        float r = bias != nullptr ? bias[ch] : 0.f;
        for (int u = 0; u < k; ++u)
            for (int v = 0; v < k; ++v) r = fmaf(w[ch * kk + u * k + v], gx_load(pl, h, wd, yy + u - p, xx + v - p, reflect), r);
        y[i] = r;
    }
}

// dx[y][x] = sum over the reflect images (iy, ix) of (y, x) in the padded domain of sum_{u,v} w[u][v] g[iy - u + p][ix - v + p]
__global__ void dwconv_dgrad_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ dx, long long planes, int c, int h,
                                    int wd, int k, int reflect) {
    const int p = k / 2, kk = k * k;
    const long long total = planes * h * wd;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % wd), y = (int)((i / wd) % h);
        const long long plane = i / ((long long)h * wd);
        const float* wc = w + (plane % c) * kk;
        const float* pl = g + plane * h * wd;
        int ys[3], xs[3], ny = 0, nx = 0;   // padded-domain coordinates minus p (i.e. logical, may be < 0 or >= h)
        ys[ny++] = y;
        xs[nx++] = x;
        if (reflect && p > 0) {
            if (y >= 1 && y <= p) ys[ny++] = -y;
            if (y <= h - 2 && y >= h - 1 - p) ys[ny++] = 2 * (h - 1) - y;
            if (x >= 1 && x <= p) xs[nx++] = -x;
            if (x <= wd - 2 && x >= wd - 1 - p) xs[nx++] = 2 * (wd - 1) - x;
        }
        float s = 0.f;
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b)
                for (int u = 0; u < k; ++u) {
                    const int oy = ys[a] + p - u;
                    if (oy < 0 || oy >= h) continue;
                    for (int v = 0; v < k; ++v) {
                        const int ox = xs[b] + p - v;
                        if (ox >= 0 && ox < wd) s = fmaf(wc[u * k + v], pl[(long long)oy * wd + ox], s);
                    }
                }
        dx[i] = s;
    }
}

// one block per channel: dw[c][tap] = sum_{n, pixels} g * xpad(shifted), db[c] = sum g   (fixed order)
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ dw,
                                                           float* __restrict__ db, int n, int c, int h, int wd, int k, int reflect) {
    __shared__ float red[16];
    const int ch = blockIdx.x, p = k / 2, kk = k * k;
    float acc[9], accb = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    const long long hw = (long long)h * wd;
    for (int in_ = 0; in_ < n; ++in_) {
        const float* xp = x + ((long long)in_ * c + ch) * hw;
        const float* gp = g + ((long long)in_ * c + ch) * hw;
        for (long long i = threadIdx.x; i < hw; i += 256) {
            const int xx = (int)(i % wd), yy = (int)(i / wd);
            const float gv = gp[i];
            accb += gv;
            for (int t = 0; t < kk; ++t) acc[t] = fmaf(gv, gx_load(xp, h, wd, yy + t / k - p, xx + t % k - p, reflect), acc[t]);
        }
    }
    for (int t = 0; t < kk; ++t) {
        const float s = block_sum(acc[t], red);
        if (threadIdx.x == 0) dw[ch * kk + t] = s;
    }
    const float s = block_sum(accb, red);
    if (threadIdx.x == 0 && db != nullptr) db[ch] = s;
}

static int grid1d(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 65535 * 16 ? 65535 * 16 : b));
}

static int check_gc(const char* what, const GC& q) {
    MMIF_REQUIRE(q.n > 0 && q.cin > 0 && q.cout > 0 && q.hi > 0 && q.wi > 0, "%s: bad extent", what);
    MMIF_REQUIRE(q.k == 1 || q.k == 3 || q.k == 5 || q.k == 7, "%s: ksize must be 1, 3, 5 or 7 (got %d)", what, q.k);
    MMIF_REQUIRE(q.s == 1 || q.s == 2, "%s: stride must be 1 or 2 (got %d)", what, q.s);
    MMIF_REQUIRE(q.p >= 0 && q.p <= q.k / 2, "%s: padding must be in [0, ksize/2]", what);
    MMIF_REQUIRE(!q.reflect || (q.p < q.hi && q.p < q.wi), "%s: reflect padding %d needs an input larger than that", what, q.p);
    MMIF_REQUIRE(q.ho > 0 && q.wo > 0, "%s: empty output", what);
    return MMIF_OK;
}
static GC make_gc(int n, int cin, int cout, int h, int w, int k, int s, int p, int reflect) {
    GC q;
    q.n = n; q.cin = cin; q.cout = cout; q.hi = h; q.wi = w; q.k = k; q.s = s; q.p = p; q.reflect = reflect;
    q.ho = (h + 2 * p - k) / s + 1;
    q.wo = (w + 2 * p - k) / s + 1;
    return q;
}
static int wg_groups(const GC& q, int total_tiles) {
    const int pairs = cdiv(q.cin, WG_CC) * cdiv(q.cout, G_OG);
    int G = 1024 / (pairs < 1 ? 1 : pairs);
    if (G < 1) G = 1;
    if (G > total_tiles) G = total_tiles;
    return G;
}

}  // namespace mmif

using namespace mmif;

// y = act(conv(x)): nn.Conv2d(cin, cout, k, stride, padding, padding_mode) + optional ReLU
extern "C" int mmif_gconv_fwd(const float* x, const float* w, const float* bias, float* y, int32_t n, int32_t cin, int32_t cout, int32_t h,
                              int32_t wd, int32_t ksize, int32_t stride, int32_t padding, int32_t reflect, int32_t relu, void* stream) {
    MMIF_REQUIRE(x != nullptr && w != nullptr && y != nullptr, "gconv_fwd: null pointer");
    const GC q = make_gc(n, cin, cout, h, wd, ksize, stride, padding, reflect);
    if (int rc = check_gc("gconv_fwd", q)) return rc;
    const int tiles_x = cdiv(q.wo, GT), tiles_y = cdiv(q.ho, GT);
    const int IT = (GT - 1) * q.s + q.k;
    const size_t lds = (size_t)(G_CC * IT * IT + G_CC * q.k * q.k * G_OG) * sizeof(float);
    hipLaunchKernelGGL(gconv_fwd_kernel, dim3(tiles_x * tiles_y, cdiv(cout, G_OG), n), dim3(256), lds, (hipStream_t)stream, x, w, bias, y, q, relu,
                       tiles_x);
    return check_launch("gconv_fwd");
}

extern "C" size_t mmif_gconv_dgrad_workspace(int32_t n, int32_t cin, int32_t h, int32_t wd, int32_t padding, int32_t reflect) {
    return (reflect && padding > 0) ? (size_t)n * cin * (h + 2 * padding) * (wd + 2 * padding) * sizeof(float) : 0;
}

// dx = d/dx of mmif_gconv_fwd (gy already masked by the activation): transposed gather on the padded domain + reflect fold
extern "C" int mmif_gconv_dgrad(const float* gy, const float* w, float* dx, int32_t n, int32_t cin, int32_t cout, int32_t h, int32_t wd,
                                int32_t ksize, int32_t stride, int32_t padding, int32_t reflect, void* workspace, size_t workspace_bytes,
                                void* stream) {
    MMIF_REQUIRE(gy != nullptr && w != nullptr && dx != nullptr, "gconv_dgrad: null pointer");
    const GC q = make_gc(n, cin, cout, h, wd, ksize, stride, padding, reflect);
    if (int rc = check_gc("gconv_dgrad", q)) return rc;
    const bool fold = reflect && padding > 0;
    if (workspace_bytes < mmif_gconv_dgrad_workspace(n, cin, h, wd, padding, reflect) || (fold && workspace == nullptr)) {
        set_error("gconv_dgrad: workspace too small");
        return MMIF_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    // zero padding: dx[y][x] = z[y + p][x + p] of the padded-domain map, i.e. gather with off = p directly on the image domain
    const int zh = fold ? h + 2 * padding : h, zw = fold ? wd + 2 * padding : wd, off = fold ? 0 : padding;
    const int tiles_x = cdiv(zw, GT), tiles_y = cdiv(zh, GT);
    const int R = (GT + q.k - 2) / q.s + 2;
    const size_t lds = (size_t)(G_CC * R * R + G_CC * q.k * q.k * G_OG) * sizeof(float);
    hipLaunchKernelGGL(gconv_tg_kernel, dim3(tiles_x * tiles_y, cdiv(cin, G_OG), n), dim3(256), lds, st, gy, w, (const float*)nullptr,
                       fold ? (float*)workspace : dx, q, zh, zw, off, tiles_x, 0);
    if (int rc = check_launch("gconv_dgrad")) return rc;
    if (!fold) return MMIF_OK;
    hipLaunchKernelGGL(gconv_fold_kernel, dim3(grid1d((long long)n * cin * h * wd)), dim3(256), 0, st, (const float*)workspace, dx,
                       (long long)n * cin, h, wd, padding);
    return check_launch("gconv_fold");
}

extern "C" size_t mmif_gconv_wgrad_workspace(int32_t cin, int32_t cout, int32_t ksize) {
    const size_t per = (size_t)G_OG * WG_CC * ksize * ksize + G_OG;
    const size_t pairs = (size_t)cdiv(cin, WG_CC) * cdiv(cout, G_OG);
    return (pairs > 1024 ? pairs : 1024) * per * sizeof(float);
}

extern "C" int mmif_gconv_wgrad(const float* x, const float* gy, float* dw, float* db, int32_t n, int32_t cin, int32_t cout, int32_t h, int32_t wd,
                                int32_t ksize, int32_t stride, int32_t padding, int32_t reflect, void* workspace, size_t workspace_bytes,
                                void* stream) {
    MMIF_REQUIRE(x != nullptr && gy != nullptr && dw != nullptr, "gconv_wgrad: null pointer");
    const GC q = make_gc(n, cin, cout, h, wd, ksize, stride, padding, reflect);
    if (int rc = check_gc("gconv_wgrad", q)) return rc;
    if (workspace == nullptr || workspace_bytes < mmif_gconv_wgrad_workspace(cin, cout, ksize)) {
        set_error("gconv_wgrad: workspace too small");
        return MMIF_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int tiles_x = cdiv(q.wo, GT), tiles_y = cdiv(q.ho, GT), tpi = tiles_x * tiles_y, total = tpi * n;
    const int G = wg_groups(q, total), n_icg = cdiv(cin, WG_CC), n_ocg = cdiv(cout, G_OG);
    const int IT = (GT - 1) * q.s + q.k;
    const size_t lds = (size_t)(WG_CC * IT * IT + 256 * G_OG) * sizeof(float);
    (void)hipFuncSetAttribute((const void*)gconv_wg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipLaunchKernelGGL(gconv_wg_kernel, dim3(G, n_icg, n_ocg), dim3(256), lds, st, x, gy, (float*)workspace, q, tiles_x, tpi, total, G);
    if (int rc = check_launch("gconv_wgrad")) return rc;
    const int nres = cout * cin * ksize * ksize + cout;
    hipLaunchKernelGGL(gconv_wg_reduce, dim3(cdiv(nres, 256)), dim3(256), 0, st, (const float*)workspace, dw, db, q, G, n_icg, n_ocg);
    return check_launch("gconv_wgrad_reduce");
}

// ---- ConvTranspose2d(cin, cout, k, stride, padding, output_padding), weight [cin][cout][k][k] (core/block.py:67-76) -------------
static int convt_out(int h, int k, int s, int p, int op) { return (h - 1) * s - 2 * p + k + op; }

extern "C" int mmif_gconvt_fwd(const float* x, const float* w, const float* bias, float* y, int32_t n, int32_t cin, int32_t cout, int32_t h,
                               int32_t wd, int32_t ksize, int32_t stride, int32_t padding, int32_t output_padding, int32_t relu, void* stream) {
    MMIF_REQUIRE(x != nullptr && w != nullptr && y != nullptr, "gconvt_fwd: null pointer");
    MMIF_REQUIRE(output_padding >= 0 && output_padding < stride, "gconvt_fwd: output_padding must be smaller than stride");
    const int ho = convt_out(h, ksize, stride, padding, output_padding), wo = convt_out(wd, ksize, stride, padding, output_padding);
    // the conv whose input gradient this is: X = y [cout ch, ho x wo], G = x [cin ch, h x w], W[o = cin][c = cout]
    GC q = make_gc(n, cout, cin, ho, wo, ksize, stride, padding, 0);
    if (int rc = check_gc("gconvt_fwd", q)) return rc;
    MMIF_REQUIRE(q.ho == h && q.wo == wd, "gconvt_fwd: inconsistent geometry");
    const int tiles_x = cdiv(wo, GT), tiles_y = cdiv(ho, GT);
    const int R = (GT + q.k - 2) / q.s + 2;
    const size_t lds = (size_t)(G_CC * R * R + G_CC * q.k * q.k * G_OG) * sizeof(float);
    hipLaunchKernelGGL(gconv_tg_kernel, dim3(tiles_x * tiles_y, cdiv(cout, G_OG), n), dim3(256), lds, (hipStream_t)stream, x, w, bias, y, q, ho, wo,
                       padding, tiles_x, relu);
    return check_launch("gconvt_fwd");
}

extern "C" int mmif_gconvt_dgrad(const float* gy, const float* w, float* dx, int32_t n, int32_t cin, int32_t cout, int32_t h, int32_t wd,
                                 int32_t ksize, int32_t stride, int32_t padding, int32_t output_padding, void* stream) {
    const int ho = convt_out(h, ksize, stride, padding, output_padding), wo = convt_out(wd, ksize, stride, padding, output_padding);
    // dx = conv(gy; W[o = cin][c = cout], stride, zero padding)
    return mmif_gconv_fwd(gy, w, nullptr, dx, n, cout, cin, ho, wo, ksize, stride, padding, 0, 0, stream);
}

extern "C" int mmif_gconvt_wgrad(const float* x, const float* gy, float* dw, float* db_scratch, int32_t n, int32_t cin, int32_t cout, int32_t h,
                                 int32_t wd, int32_t ksize, int32_t stride, int32_t padding, int32_t output_padding, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    const int ho = convt_out(h, ksize, stride, padding, output_padding), wo = convt_out(wd, ksize, stride, padding, output_padding);
    // dW[ci][co] = wgrad of the conv X = gy [cout ch] -> G = x [cin ch]; its "db" (sum of x per channel) is not a gradient of
    // the layer: the caller passes a scratch or NULL and takes the bias gradient as the plane sums of gy
    return mmif_gconv_wgrad(gy, x, dw, db_scratch, n, cout, cin, ho, wo, ksize, stride, padding, 0, workspace, workspace_bytes, stream);
}

extern "C" int mmif_relu_bwd(const float* g, const float* y, float* out, int64_t count, void* stream) {
    MMIF_REQUIRE(g != nullptr && y != nullptr && out != nullptr && count >= 0, "relu_bwd: bad arguments");
    if (count == 0) return MMIF_OK;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid1d(count)), dim3(256), 0, (hipStream_t)stream, g, y, out, (long long)count);
    return check_launch("relu_bwd");
}

extern "C" int mmif_channel_sum(const float* x, float* out, int32_t n, int32_t c, int64_t hw, void* stream) {
    MMIF_REQUIRE(x != nullptr && out != nullptr && n > 0 && c > 0 && hw > 0, "channel_sum: bad arguments");
    hipLaunchKernelGGL(channel_sum_kernel, dim3(c), dim3(256), 0, (hipStream_t)stream, x, out, n, c, (long long)hw);
    return check_launch("channel_sum");
}

// ---- depth-wise ConvLayer (groups == channels): x, y [n][c][h][w]; w [c][1][k][k]; k in {1, 3}, stride 1, padding k/2 ----
extern "C" int mmif_dwconv_fwd(const float* x, const float* w, const float* bias, float* y, int32_t n, int32_t c, int32_t h, int32_t wd,
                               int32_t ksize, int32_t reflect, void* stream) {
    MMIF_REQUIRE(x != nullptr && w != nullptr && y != nullptr && n > 0 && c > 0 && h > 0 && wd > 0, "dwconv_fwd: bad arguments");
    MMIF_REQUIRE(ksize == 1 || ksize == 3, "dwconv_fwd: ksize must be 1 or 3 (got %d)", ksize);
    MMIF_REQUIRE(!reflect || ksize == 1 || (h >= 2 && wd >= 2), "dwconv_fwd: reflect padding needs h,w >= 2");
    hipLaunchKernelGGL(dwconv_fwd_kernel, dim3(grid1d((long long)n * c * h * wd)), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, (long long)n * c,
                       c, h, wd, ksize, reflect);
    return check_launch("dwconv_fwd");
}

extern "C" int mmif_dwconv_dgrad(const float* gy, const float* w, float* dx, int32_t n, int32_t c, int32_t h, int32_t wd, int32_t ksize,
                                 int32_t reflect, void* stream) {
    MMIF_REQUIRE(gy != nullptr && w != nullptr && dx != nullptr && n > 0 && c > 0 && h > 0 && wd > 0, "dwconv_dgrad: bad arguments");
    MMIF_REQUIRE(ksize == 1 || ksize == 3, "dwconv_dgrad: ksize must be 1 or 3 (got %d)", ksize);
    hipLaunchKernelGGL(dwconv_dgrad_kernel, dim3(grid1d((long long)n * c * h * wd)), dim3(256), 0, (hipStream_t)stream, gy, w, dx, (long long)n * c,
                       c, h, wd, ksize, reflect);
    return check_launch("dwconv_dgrad");
}

extern "C" int mmif_dwconv_wgrad(const float* x, const float* gy, float* dw, float* db, int32_t n, int32_t c, int32_t h, int32_t wd, int32_t ksize,
                                 int32_t reflect, void* stream) {
    MMIF_REQUIRE(x != nullptr && gy != nullptr && dw != nullptr && n > 0 && c > 0 && h > 0 && wd > 0, "dwconv_wgrad: bad arguments");
    MMIF_REQUIRE(ksize == 1 || ksize == 3, "dwconv_wgrad: ksize must be 1 or 3 (got %d)", ksize);
    hipLaunchKernelGGL(dwconv_wgrad_kernel, dim3(c), dim3(256), 0, (hipStream_t)stream, x, gy, dw, db, n, c, h, wd, ksize, reflect);
    return check_launch("dwconv_wgrad");
}
