// Micro-benchmark: v_mfma_f32_16x16x32_bf16 issue rate, 2 waves per SIMD (8-wave block), a block barrier every 144 MFMAs
// (the conv kernels' chunk shape), with N cycles of non-MFMA "admin" VALU work per chunk placed either at the same
// point in every wave or at opposite ends in the two waves of a SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// MODE 0: admin first in all waves; 1: admin first in waves 0-3, last in waves 4-7; 2: like 0 but setprio(1) on MFMAs of waves 4-7
template <int MODE, bool RANDOM>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, int admin, long long* cyc) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        uint4 ua, ub;
        if (RANDOM) {   // pseudo-random bf16 values in (-2, 2): full mantissa + sign toggling, like real activations / weights
            unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u) ^ (i * 0x9e3779b9u);
            auto nxt = [&]() { h = h * 1664525u + 1013904223u; return ((h >> 9) & 0x807f807fu) | 0x3f003f00u | ((h >> 3) & 0x00800080u); };
            ua = make_uint4(nxt(), nxt(), nxt(), nxt());
            ub = make_uint4(nxt(), nxt(), nxt(), nxt());
        } else {
            ua = ub = make_uint4(0x3f803f80u + threadIdx.x, 0x3f813f80u, 0x3f803f82u, 0x3f833f80u);
        }
        a[i] = __builtin_bit_cast(bf16x8, ua);
        b[i] = __builtin_bit_cast(bf16x8, ub);
    }
    f32x4 acc[4][4];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0, 0, 0, 0};
    float v = threadIdx.x * 0.001f;
    const int wave = threadIdx.x >> 6;
    const bool late = (MODE == 1) && wave >= 4;
    if (MODE == 2 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (!late) for (int i = 0; i < admin; ++i) v = v * 1.0001f + 0.5f;   // dependent VALU chain: ~8 cycles each
#pragma unroll
        for (int s = 0; s < 9; ++s)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
        if (late) for (int i = 0; i < admin; ++i) v = v * 1.0001f + 0.5f;
        __builtin_amdgcn_s_barrier();
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = v;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) s += acc[m][n][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE, bool RANDOM = false>
void run(const char* name, int admin) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * sizeof(float)); hipMalloc(&cyc, 8);
    const int iters = 4000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, RANDOM><<<grid, 512>>>(out, 10, admin, cyc);
    hipEventRecord(e0);
    k<MODE, RANDOM><<<grid, 512>>>(out, iters, admin, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-46s admin %4d: %7.3f ms  %6.0f TFLOP/s  ticks per chunk %7.0f (MFMA-bound 4608)  us per chunk %.2f\n", name, admin, ms,
           (double)grid * 8 * iters * 144 * 16384 / ms / 1e9, (double)c / iters, ms * 1e3 / iters);
    hipFree(out); hipFree(cyc);
}

int main() {
    // operand data: near-constant vs pseudo-random (the clock the chip sustains depends on it)
    run<0, false>("pure MFMA, near-constant operands", 0);
    run<0, true>("pure MFMA, pseudo-random operands", 0);
    run<0, false>("pure MFMA, near-constant operands", 0);
    run<0, true>("pure MFMA, pseudo-random operands", 0);
    for (int admin : {0, 50, 100, 200}) {
        run<0>("admin first in all waves", admin);
        run<1>("admin first (w0-3) / last (w4-7)", admin);
        run<2>("admin first in all, setprio 1 on w4-7", admin);
    }
    return 0;
}
