// Normalisation + activation epilogues of the reference's ConvLayer (core/block.py:78-92) for the nets outside the hot path
// (row n4): nn.BatchNorm2d (IFCNN, DIFNet, PMGI) and nn.GroupNorm(out_ch, out_ch) -- one group per channel, i.e. per-(sample,
// channel) statistics (SEDRFuse) -- followed by ReLU / LeakyReLU(0.2) / Tanh / nothing, on plain NCHW fp32 tensors.
//
//   norm_moments_kernel   per plane (n, c): sum x, sum x^2 in fp64 (one block per plane, fixed order)
//   norm_finish_kernel    BatchNorm: combine the n planes of a channel -> mean, rstd (+ running stats, momentum, unbiased var)
//                         GroupNorm: per plane -> mean, rstd
//   norm_act_fwd_kernel   y = act(gamma * (x - mean) * rstd + beta)
//   norm_bwd_sums_kernel  per plane: sum dz, sum dz * xhat   with dz = gy * act'(y)            (fp64)
//   norm_bwd_finish       per channel: dgamma = sum_n sum dz xhat, dbeta = sum_n sum dz (+ the per-statistics sums of BatchNorm)
//   norm_act_bwd_kernel   dx = gamma rstd (dz - mean(dz) - xhat mean(dz xhat))   (train) | gamma rstd dz (eval BatchNorm)
// act codes: 0 none, 1 ReLU, 2 LeakyReLU(slope), 3 Tanh, 4 ReLU6 -- every derivative is a function of the OUTPUT y, so only x and y are kept.
#include <math.h>

#include "common.hpp"

namespace mmif {

__device__ inline double block_sum_f64(double v, double* smem /* >= 16 */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) r += smem[i];
    }
    return r;
}

__device__ inline float act_fwd(float z, int act, float slope) {
    if (act == 1) return fmaxf(z, 0.f);
    if (act == 2) return z > 0.f ? z : z * slope;
    if (act == 3) return tanhf(z);
    if (act == 4) return fminf(fmaxf(z, 0.f), 6.f);
    return z;
}
__device__ inline float act_dydz(float y, int act, float slope) {
    if (act == 1) return y > 0.f ? 1.f : 0.f;
    if (act == 2) return y > 0.f ? 1.f : slope;
    if (act == 3) return 1.f - y * y;
    if (act == 4) return (y > 0.f && y < 6.f) ? 1.f : 0.f;   // hardtanh_backward: zero at and beyond both bounds
    return 1.f;
}

__global__ __launch_bounds__(256) void norm_moments_kernel(const float* __restrict__ x, double* __restrict__ mom, long long hw) {
    __shared__ double red[16];
    const float* pl = x + (long long)blockIdx.x * hw;
    double s1 = 0.0, s2 = 0.0;
    for (long long i = threadIdx.x; i < hw; i += 256) {
        const double v = (double)pl[i];
        s1 += v;
        s2 += v * v;
    }
    const double t1 = block_sum_f64(s1, red);
    const double t2 = block_sum_f64(s2, red);
    if (threadIdx.x == 0) { mom[2 * blockIdx.x] = t1; mom[2 * blockIdx.x + 1] = t2; }
}

// per_channel (BatchNorm): one thread per channel sums its n planes; else one thread per plane.  stats[2*i] = mean, [2*i+1] = rstd
__global__ void norm_finish_kernel(const double* __restrict__ mom, float* __restrict__ stats, int n, int c, long long hw, int per_channel,
                                   float eps, float* __restrict__ run_mean, float* __restrict__ run_var, float momentum) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (per_channel) {
        if (i >= c) return;
        double s1 = 0.0, s2 = 0.0;
        for (int in_ = 0; in_ < n; ++in_) { s1 += mom[2 * ((long long)in_ * c + i)]; s2 += mom[2 * ((long long)in_ * c + i) + 1]; }
        const double m = (double)n * (double)hw, mean = s1 / m;
        double var = s2 / m - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[2 * i] = (float)mean;
        stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
        if (run_mean != nullptr) {   // nn.BatchNorm2d training mode: running = (1 - momentum) running + momentum * batch (unbiased var)
            run_mean[i] = (1.f - momentum) * run_mean[i] + momentum * (float)mean;
            const double unb = m > 1.0 ? var * m / (m - 1.0) : var;
            run_var[i] = (1.f - momentum) * run_var[i] + momentum * (float)unb;
        }
    } else {
        if (i >= n * c) return;
        const double m = (double)hw, mean = mom[2 * i] / m;
        double var = mom[2 * i + 1] / m - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[2 * i] = (float)mean;
        stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// cross-rank BatchNorm (SyncBatchNorm): chan[2i], chan[2i+1] = sum_n of the plane moments of channel i (fixed order)
__global__ void norm_chan_reduce_kernel(const double* __restrict__ mom, double* __restrict__ chan, int n, int c, long long hw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    double s1 = 0.0, s2 = 0.0;
    for (int in_ = 0; in_ < n; ++in_) { s1 += mom[2 * ((long long)in_ * c + i)]; s2 += mom[2 * ((long long)in_ * c + i) + 1]; }
    chan[2 * i] = s1;
    chan[2 * i + 1] = s2;
    if (i == 0) chan[2 * c] = (double)n * (double)hw;   // this rank's element count per channel: summed by the same all-reduce
}

// ... and mean / rstd (+ running buffers) from channel sums over m = chan[2c] elements (the GLOBAL count after the all-reduce)
__global__ void norm_finish_chan_kernel(const double* __restrict__ chan, float* __restrict__ stats, int c, float eps,
                                        float* __restrict__ run_mean, float* __restrict__ run_var, float momentum) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    const double m = chan[2 * c];
    const double mean = chan[2 * i] / m;
    double var = chan[2 * i + 1] / m - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[2 * i] = (float)mean;
    stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean != nullptr) {
        run_mean[i] = (1.f - momentum) * run_mean[i] + momentum * (float)mean;
        const double unb = m > 1.0 ? var * m / (m - 1.0) : var;
        run_var[i] = (1.f - momentum) * run_var[i] + momentum * (float)unb;
    }
}

// eval-mode BatchNorm: statistics from the running buffers
__global__ void norm_running_stats_kernel(const float* __restrict__ run_mean, const float* __restrict__ run_var, float* __restrict__ stats, int c,
                                          float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    stats[2 * i] = run_mean[i];
    stats[2 * i + 1] = 1.f / sqrtf(run_var[i] + eps);
}

__global__ void norm_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, float* __restrict__ y, int c, long long hw, long long total, int per_channel,
                                    int act, float slope) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long plane = i / hw;
        const int ch = (int)(plane % c);
        const long long si = per_channel ? ch : plane;
        const float xh = (x[i] - stats[2 * si]) * stats[2 * si + 1];
        const float z = (gamma != nullptr ? gamma[ch] : 1.f) * xh + (beta != nullptr ? beta[ch] : 0.f);
        y[i] = act_fwd(z, act, slope);
    }
}

// per plane: sums[2p] = sum dz, sums[2p+1] = sum dz * xhat
__global__ __launch_bounds__(256) void norm_bwd_sums_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gy,
                                                            const float* __restrict__ stats, double* __restrict__ sums, int c, long long hw,
                                                            int per_channel, int act, float slope) {
    __shared__ double red[16];
    const long long plane = blockIdx.x;
    const long long si = per_channel ? (plane % c) : plane;
    const float mean = stats[2 * si], rstd = stats[2 * si + 1];
    const long long base = plane * hw;
    double s1 = 0.0, s2 = 0.0;
    for (long long i = threadIdx.x; i < hw; i += 256) {
        const float dz = gy[base + i] * act_dydz(y[base + i], act, slope);
        s1 += (double)dz;
        s2 += (double)dz * (double)((x[base + i] - mean) * rstd);
    }
    const double t1 = block_sum_f64(s1, red);
    const double t2 = block_sum_f64(s2, red);
    if (threadIdx.x == 0) { sums[2 * plane] = t1; sums[2 * plane + 1] = t2; }
}

// per channel: dgamma, dbeta (sum over the n planes); for BatchNorm also chan[2c] = sum dz, chan[2c+1] = sum dz xhat over the batch
__global__ void norm_bwd_finish_kernel(const double* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       double* __restrict__ chan, int n, int c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    double s1 = 0.0, s2 = 0.0;
    for (int in_ = 0; in_ < n; ++in_) { s1 += sums[2 * ((long long)in_ * c + i)]; s2 += sums[2 * ((long long)in_ * c + i) + 1]; }
    if (dbeta != nullptr) dbeta[i] = (float)s1;
    if (dgamma != nullptr) dgamma[i] = (float)s2;
    if (chan != nullptr) { chan[2 * i] = s1; chan[2 * i + 1] = s2; }
}

// mode 0: eval BatchNorm (statistics are constants); 1: per-channel batch statistics (sums in chan, m = n hw); 2: per-plane (sums, m = hw)
__global__ void norm_act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gy,
                                    const float* __restrict__ stats, const float* __restrict__ gamma, const double* __restrict__ sums,
                                    const double* __restrict__ chan, float* __restrict__ dx, double m_chan, const double* __restrict__ m_dev, int c,
                                    long long hw, long long total, int mode, int act, float slope) {
    if (m_dev != nullptr) m_chan = *m_dev;   // cross-rank BatchNorm: the global count lives on the device (no host sync)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long plane = i / hw;
        const int ch = (int)(plane % c);
        const long long si = mode == 2 ? plane : ch;
        const float mean = stats[2 * si], rstd = stats[2 * si + 1];
        const float g = gamma != nullptr ? gamma[ch] : 1.f;
        const float dz = gy[i] * act_dydz(y[i], act, slope);
        float r = dz;
        if (mode != 0) {
            const double m = mode == 1 ? m_chan : (double)hw;
            const double* s = mode == 1 ? chan + 2 * ch : sums + 2 * plane;
            const float xh = (x[i] - mean) * rstd;
            r = dz - (float)(s[0] / m) - xh * (float)(s[1] / m);
        }
        dx[i] = g * rstd * r;
    }
}

__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long total, int act, float slope) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        y[i] = act_fwd(x[i], act, slope);
}
__global__ void act_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y, float* __restrict__ dx, long long total, int act,
                               float slope) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        dx[i] = gy[i] * act_dydz(y[i], act, slope);
}

static int grid1d_n(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 65535 * 16 ? 65535 * 16 : b));
}
}  // namespace mmif

using namespace mmif;

// workspace (both directions): 2 doubles per plane + 2 doubles per channel
extern "C" size_t mmif_norm_workspace(int32_t n, int32_t c) { return ((size_t)n * c + c) * 2 * sizeof(double); }

// kind: 0 = BatchNorm2d (training: batch statistics, running buffers updated when given), 1 = BatchNorm2d eval (running buffers),
//       2 = GroupNorm with one group per channel.  stats: [c][2] (kinds 0, 1) or [n*c][2] (kind 2) floats = (mean, rstd), kept for backward.
extern "C" int mmif_norm_act_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, float* running_mean,
                                 float* running_var, int32_t n, int32_t c, int64_t hw, int32_t kind, float eps, float momentum, int32_t act,
                                 float slope, void* workspace, size_t workspace_bytes, void* stream) {
    MMIF_REQUIRE(x != nullptr && y != nullptr && stats != nullptr && n > 0 && c > 0 && hw > 0, "norm_act_fwd: bad arguments");
    MMIF_REQUIRE(kind >= 0 && kind <= 2 && act >= 0 && act <= 4, "norm_act_fwd: bad kind / activation");
    MMIF_REQUIRE(kind != 1 || (running_mean != nullptr && running_var != nullptr), "norm_act_fwd: eval mode needs the running statistics");
    if (kind != 1 && (workspace == nullptr || workspace_bytes < mmif_norm_workspace(n, c))) {
        set_error("norm_act_fwd: workspace too small");
        return MMIF_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)n * c * hw;
    if (kind == 1) {
        hipLaunchKernelGGL(norm_running_stats_kernel, dim3(cdiv(c, 256)), dim3(256), 0, st, running_mean, running_var, stats, c, eps);
    } else {
        double* mom = (double*)workspace;
        hipLaunchKernelGGL(norm_moments_kernel, dim3(n * c), dim3(256), 0, st, x, mom, (long long)hw);
        if (int rc = check_launch("norm_moments")) return rc;
        const int cnt = kind == 0 ? c : n * c;
        hipLaunchKernelGGL(norm_finish_kernel, dim3(cdiv(cnt, 256)), dim3(256), 0, st, mom, stats, n, c, (long long)hw, kind == 0 ? 1 : 0, eps,
                           kind == 0 ? running_mean : nullptr, kind == 0 ? running_var : nullptr, momentum);
    }
    if (int rc = check_launch("norm_stats")) return rc;
    hipLaunchKernelGGL(norm_act_fwd_kernel, dim3(grid1d_n(total)), dim3(256), 0, st, x, stats, gamma, beta, y, c, (long long)hw, total,
                       kind == 2 ? 0 : 1, act, slope);
    return check_launch("norm_act_fwd");
}

extern "C" int mmif_norm_act_bwd(const float* x, const float* y, const float* gy, const float* stats, const float* gamma, float* dx, float* dgamma,
                                 float* dbeta, int32_t n, int32_t c, int64_t hw, int32_t kind, int32_t act, float slope, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    MMIF_REQUIRE(x != nullptr && y != nullptr && gy != nullptr && stats != nullptr && dx != nullptr && n > 0 && c > 0 && hw > 0,
                 "norm_act_bwd: bad arguments");
    MMIF_REQUIRE(kind >= 0 && kind <= 2 && act >= 0 && act <= 4, "norm_act_bwd: bad kind / activation");
    if (workspace == nullptr || workspace_bytes < mmif_norm_workspace(n, c)) {
        set_error("norm_act_bwd: workspace too small");
        return MMIF_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    double* sums = (double*)workspace;
    double* chan = sums + 2 * (size_t)n * c;
    const long long total = (long long)n * c * hw;
    hipLaunchKernelGGL(norm_bwd_sums_kernel, dim3(n * c), dim3(256), 0, st, x, y, gy, stats, sums, c, (long long)hw, kind == 2 ? 0 : 1, act, slope);
    if (int rc = check_launch("norm_bwd_sums")) return rc;
    hipLaunchKernelGGL(norm_bwd_finish_kernel, dim3(cdiv(c, 256)), dim3(256), 0, st, sums, dgamma, dbeta, chan, n, c);
    if (int rc = check_launch("norm_bwd_finish")) return rc;
    hipLaunchKernelGGL(norm_act_bwd_kernel, dim3(grid1d_n(total)), dim3(256), 0, st, x, y, gy, stats, gamma, sums, chan, dx, (double)n * (double)hw,
                       (const double*)nullptr, c, (long long)hw, total, kind == 1 ? 0 : (kind == 0 ? 1 : 2), act, slope);
    return check_launch("norm_act_bwd");
}

// ---- cross-rank BatchNorm (the reference wraps its BatchNorm nets in nn.SyncBatchNorm for DDP, train.py:296): the statistics stage and the
// apply stage are separate calls; between them the HOST all-reduces the [c][2] fp64 sums (and the element count) over RCCL.
extern "C" int mmif_bn_moments(const float* x, double* chan_sums, int32_t n, int32_t c, int64_t hw, void* workspace, size_t workspace_bytes,
                               void* stream) {
    MMIF_REQUIRE(x != nullptr && chan_sums != nullptr && n > 0 && c > 0 && hw > 0, "bn_moments: bad arguments");
    if (workspace == nullptr || workspace_bytes < mmif_norm_workspace(n, c)) {
        set_error("bn_moments: workspace too small");
        return MMIF_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    double* mom = (double*)workspace;
    hipLaunchKernelGGL(norm_moments_kernel, dim3(n * c), dim3(256), 0, st, x, mom, (long long)hw);
    if (int rc = check_launch("norm_moments")) return rc;
    hipLaunchKernelGGL(norm_chan_reduce_kernel, dim3(cdiv(c, 256)), dim3(256), 0, st, mom, chan_sums, n, c, (long long)hw);
    return check_launch("bn_moments");
}

extern "C" int mmif_bn_apply_fwd(const float* x, const double* chan_sums, const float* gamma, const float* beta, float* y,
                                 float* stats, float* running_mean, float* running_var, int32_t n, int32_t c, int64_t hw, float eps,
                                 float momentum, int32_t act, float slope, void* stream) {
    MMIF_REQUIRE(x != nullptr && chan_sums != nullptr && y != nullptr && stats != nullptr && n > 0 && c > 0 && hw > 0, "bn_apply_fwd: bad arguments");
    MMIF_REQUIRE(act >= 0 && act <= 4, "bn_apply_fwd: bad activation");
    MMIF_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_apply_fwd: running_mean and running_var come together");
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)n * c * hw;
    hipLaunchKernelGGL(norm_finish_chan_kernel, dim3(cdiv(c, 256)), dim3(256), 0, st, chan_sums, stats, c, eps, running_mean, running_var,
                       momentum);
    if (int rc = check_launch("norm_finish_chan")) return rc;
    hipLaunchKernelGGL(norm_act_fwd_kernel, dim3(grid1d_n(total)), dim3(256), 0, st, x, stats, gamma, beta, y, c, (long long)hw, total, 1, act, slope);
    return check_launch("bn_apply_fwd");
}

// chan_sums[c][2] = this rank's (sum dz, sum dz xhat); dgamma / dbeta (either may be NULL) are this rank's LOCAL sums -- they travel in the
// gradient all-reduce like every other parameter gradient (torch's SyncBatchNorm does the same)
extern "C" int mmif_bn_bwd_sums(const float* x, const float* y, const float* gy, const float* stats, double* chan_sums, float* dgamma,
                                float* dbeta, int32_t n, int32_t c, int64_t hw, int32_t act, float slope, void* workspace,
                                size_t workspace_bytes, void* stream) {
    MMIF_REQUIRE(x != nullptr && y != nullptr && gy != nullptr && stats != nullptr && chan_sums != nullptr && n > 0 && c > 0 && hw > 0,
                 "bn_bwd_sums: bad arguments");
    MMIF_REQUIRE(act >= 0 && act <= 4, "bn_bwd_sums: bad activation");
    if (workspace == nullptr || workspace_bytes < mmif_norm_workspace(n, c)) {
        set_error("bn_bwd_sums: workspace too small");
        return MMIF_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    double* sums = (double*)workspace;
    hipLaunchKernelGGL(norm_bwd_sums_kernel, dim3(n * c), dim3(256), 0, st, x, y, gy, stats, sums, c, (long long)hw, 1, act, slope);
    if (int rc = check_launch("norm_bwd_sums")) return rc;
    hipLaunchKernelGGL(norm_bwd_finish_kernel, dim3(cdiv(c, 256)), dim3(256), 0, st, sums, dgamma, dbeta, chan_sums, n, c);
    return check_launch("bn_bwd_sums");
}

extern "C" int mmif_bn_apply_bwd(const float* x, const float* y, const float* gy, const float* stats, const float* gamma,
                                 const double* chan_sums, const double* count, float* dx, int32_t n, int32_t c, int64_t hw, int32_t act, float slope,
                                 void* stream) {
    MMIF_REQUIRE(x != nullptr && y != nullptr && gy != nullptr && stats != nullptr && chan_sums != nullptr && dx != nullptr && n > 0 && c > 0 &&
                     hw > 0 && count != nullptr,
                 "bn_apply_bwd: bad arguments");
    MMIF_REQUIRE(act >= 0 && act <= 4, "bn_apply_bwd: bad activation");
    const long long total = (long long)n * c * hw;
    hipLaunchKernelGGL(norm_act_bwd_kernel, dim3(grid1d_n(total)), dim3(256), 0, (hipStream_t)stream, x, y, gy, stats, gamma, nullptr, chan_sums, dx,
                       0.0, count, c, (long long)hw, total, 1, act, slope);
    return check_launch("bn_apply_bwd");
}

extern "C" int mmif_act_fwd(const float* x, float* y, int64_t count, int32_t act, float slope, void* stream) {
    MMIF_REQUIRE(x != nullptr && y != nullptr && count >= 0 && act >= 0 && act <= 4, "act_fwd: bad arguments");
    if (count == 0) return MMIF_OK;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid1d_n(count)), dim3(256), 0, (hipStream_t)stream, x, y, (long long)count, act, slope);
    return check_launch("act_fwd");
}

extern "C" int mmif_act_bwd(const float* gy, const float* y, float* dx, int64_t count, int32_t act, float slope, void* stream) {
    MMIF_REQUIRE(gy != nullptr && y != nullptr && dx != nullptr && count >= 0 && act >= 0 && act <= 4, "act_bwd: bad arguments");
    if (count == 0) return MMIF_OK;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid1d_n(count)), dim3(256), 0, (hipStream_t)stream, gy, y, dx, (long long)count, act, slope);
    return check_launch("act_bwd");
}
