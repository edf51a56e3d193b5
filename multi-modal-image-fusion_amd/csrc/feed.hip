// On-device data feed: the step right before the hot path.  Replaces, per batch, the reference's
//   FusionPatches.__getitem__ (data/patches.py:61-74): norm(patch) -> transform(patch, mode = random 0..7) -> float tensor [1,P,P]
//   norm   data/transform.py:15-29  (None: /255.0 | 'min-max' | 'z-score')
//   transform data/transform.py:38-66 (the 8 dihedral variants built from fliplr / flipud / rot90)
// plus the DataLoader's collate + pin_memory + H2D copy: the uint8 patch bank lives in HBM once (a 288 GB device holds any of
// the reference's datasets as patches), a batch is gathered by index with the augmentation applied on the fly.
// HBM-bound byte work: 1 B read + 4 B written per pixel.
#include "common.hpp"

namespace mmif {

// source coordinate of output pixel (y, x) for each data/transform.py mode (square P x P patches):
//   0 identity | 1 fliplr | 2 rot180 | 3 flipud | 4 rot90 (ccw) | 5 rot90 + flipud (= transpose)
//   6 rot270 (cw) | 7 rot270 + flipud (= anti-transpose)
__host__ __device__ inline void dihedral_src(int mode, int P, int y, int x, int& sy, int& sx) {
    const int q = P - 1;
    switch (mode) {
        case 1: sy = y; sx = q - x; break;
        case 2: sy = q - y; sx = q - x; break;
        case 3: sy = q - y; sx = x; break;
        case 4: sy = x; sx = q - y; break;
        case 5: sy = x; sx = y; break;
        case 6: sy = q - x; sx = y; break;
        case 7: sy = q - x; sx = q - y; break;
        default: sy = y; sx = x; break;
    }
}

// one block per output patch; norm_mode 0: v / 255, 1: (v - min) / max(max - min, eps), 2: (v - mean) / max(std, eps)
__global__ __launch_bounds__(256) void patch_feed_kernel(const uint8_t* __restrict__ bank, long long n_patches, int P,
                                                          const int* __restrict__ idx, const int* __restrict__ mode, int norm_mode,
                                                          float* __restrict__ out) {
    __shared__ float red[16];
    __shared__ float s_a, s_b;
    const int b = blockIdx.x;
    long long src = idx[b];
    src = src < 0 ? 0 : (src >= n_patches ? n_patches - 1 : src);
    const uint8_t* p = bank + src * (long long)P * P;
    const int m = mode != nullptr ? (mode[b] & 7) : 0;
    float sub = 0.f, div = 255.f;
    if (norm_mode != 0) {
        float s0 = norm_mode == 1 ? 255.f : 0.f, s1 = 0.f;   // min-max: (min, max); z-score: (sum, -)
        for (int e = threadIdx.x; e < P * P; e += 256) {
            const float v = (float)p[e];
            if (norm_mode == 1) { s0 = fminf(s0, v); s1 = fmaxf(s1, v); }
            else s0 += v;
        }
        if (norm_mode == 1) {
            // block min / max: reuse block_sum's layout with shuffles
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { s0 = fminf(s0, __shfl_down(s0, o, 64)); s1 = fmaxf(s1, __shfl_down(s1, o, 64)); }
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            if (lane == 0) { red[wave] = s0; red[4 + wave] = s1; }
            __syncthreads();
            if (threadIdx.x == 0) {
                const float mn = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
                const float mx = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
                s_a = mn;
                s_b = fmaxf(mx - mn, 1e-7f);
            }
        } else {
            const float tot = block_sum(s0, red);
            if (threadIdx.x == 0) s_a = tot / (float)(P * P);
            __syncthreads();
            const float mean = s_a;
            float sq = 0.f;
            for (int e = threadIdx.x; e < P * P; e += 256) {
                const float d = (float)p[e] - mean;
                sq += d * d;
            }
            const float tsq = block_sum(sq, red);
            if (threadIdx.x == 0) s_b = fmaxf(sqrtf(tsq / (float)(P * P)), 1e-7f);
        }
        __syncthreads();
        sub = s_a;
        div = s_b;
    }
    float* o = out + (long long)b * P * P;
    for (int e = threadIdx.x; e < P * P; e += 256) {
        const int y = e / P, x = e - y * P;
        int sy, sx;
        dihedral_src(m, P, y, x, sy, sx);
        o[e] = ((float)p[sy * P + sx] - sub) / div;   // IEEE division: bit-identical to numpy's img / 255.0
    }
}

}  // namespace mmif

using namespace mmif;

extern "C" int mmif_patch_feed(const uint8_t* bank, int64_t n_patches, int32_t patch, const int32_t* idx, const int32_t* mode,
                               int32_t batch, int32_t norm_mode, float* out, void* stream) {
    MMIF_REQUIRE(bank != nullptr && idx != nullptr && out != nullptr, "patch_feed: NULL argument");
    MMIF_REQUIRE(n_patches > 0 && patch > 0 && batch > 0, "patch_feed: n_patches, patch and batch must be positive");
    if (norm_mode < 0 || norm_mode > 2) {
        set_error("only supported ['min-max', 'z-score'] mode");   // data/transform.py:27
        return MMIF_EINVAL;
    }
    hipLaunchKernelGGL(patch_feed_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, bank, (long long)n_patches, patch, idx, mode,
                       norm_mode, out);
    return check_launch("patch_feed");
}
