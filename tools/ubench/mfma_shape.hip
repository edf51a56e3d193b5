// Micro-benchmark: does the MFMA shape change what the chip sustains under a conv-like operand stream?
//   A: v_mfma_f32_16x16x32_bf16, 4x4 fragments per wave (64 ch x 64 px), 8 ds_read_b128 per 16 MFMAs   (conv_dma_kernel today)
//   B: v_mfma_f32_32x32x16_bf16, 2x2 fragments x 2 k-halves,             8 ds_read_b128 per  8 MFMAs   (same tile, same LDS bytes)
// Both: 8 waves per block (2 per SIMD), one block per CU, operands re-read from LDS every k-step (pseudo-random bf16 data),
// one block barrier per 9 k-steps.  Same FLOPs per k-step (16 x 16384 = 8 x 32768).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ inline bf16x8 lds_frag(const uint4* lds, int idx) { return __builtin_bit_cast(bf16x8, lds[idx]); }

template <int SHAPE, bool LDSREAD>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];   // 64 KB = 4096 granules
    for (int i = threadIdx.x; i < 4096; i += 512) {
        unsigned h = (i * 2654435761u) ^ (blockIdx.x * 40503u);
        auto nxt = [&]() { h = h * 1664525u + 1013904223u; return ((h >> 9) & 0x807f807fu) | 0x3f003f00u | ((h >> 3) & 0x00800080u); };
        lds[i] = make_uint4(nxt(), nxt(), nxt(), nxt());
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (SHAPE == 0) {
        f32x4 acc[4][4];
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0, 0, 0, 0};
        bf16x8 a[4], b[4];
        for (int i = 0; i < 4; ++i) { a[i] = lds_frag(lds, i * 64 + lane); b[i] = lds_frag(lds, (i + 4) * 64 + lane); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int st = 0; st < 9; ++st) {
                const int base = ((it * 9 + st) * 8 + wave * 72) & 63;   // walks the 64 KB
                if (LDSREAD) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { a[i] = lds_frag(lds, ((base + i) & 63) * 64 + lane); b[i] = lds_frag(lds, ((base + 4 + i) & 63) * 64 + lane); }
                }
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
            }
            __builtin_amdgcn_s_barrier();
        }
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) s += acc[m][n][0];
    } else {
        f32x16 acc[2][2];
        for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
        bf16x8 a[2][2], b[2][2];
        for (int i = 0; i < 2; ++i) for (int h = 0; h < 2; ++h) { a[i][h] = lds_frag(lds, (i * 2 + h) * 64 + lane); b[i][h] = lds_frag(lds, (i * 2 + h + 4) * 64 + lane); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int st = 0; st < 9; ++st) {
                const int base = ((it * 9 + st) * 8 + wave * 72) & 63;
                if (LDSREAD) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            a[i][h] = lds_frag(lds, ((base + i * 2 + h) & 63) * 64 + lane);
                            b[i][h] = lds_frag(lds, ((base + 4 + i * 2 + h) & 63) * 64 + lane);
                        }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][h], b[n][h], acc[m][n], 0, 0, 0);
            }
            __builtin_amdgcn_s_barrier();
        }
        for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) s += acc[m][n][0] + acc[m][n][7];
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int SHAPE, bool LDSREAD>
void run(const char* name) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * sizeof(float)); hipMalloc(&cyc, 8);
    const int iters = 4000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k<SHAPE, LDSREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    k<SHAPE, LDSREAD><<<grid, 512, 65536>>>(out, 10, cyc);
    hipEventRecord(e0);
    k<SHAPE, LDSREAD><<<grid, 512, 65536>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-58s %7.3f ms  %6.0f TFLOP/s  ticks per chunk %7.0f (MFMA-bound 4608)  ticks/us %.0f\n", name, ms,
           (double)grid * 8 * iters * 144 * 16384 / ms / 1e9, (double)c / iters, (double)c / (ms * 1e3));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run<0, false>("16x16x32, operands in registers");
        run<1, false>("32x32x16, operands in registers");
        run<0, true>("16x16x32, 8 ds_read_b128 per 16 MFMAs");
        run<1, true>("32x32x16, 8 ds_read_b128 per 8 MFMAs");
    }
    return 0;
}
