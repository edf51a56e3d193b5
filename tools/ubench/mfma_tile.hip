// Micro-benchmark: what does the chip SUSTAIN (seconds, not milliseconds) under a conv-like MFMA + LDS operand stream, and does the
// wave tile (LDS operand bytes per MFMA) move that number?
//   T0: 64 ch x  64 px per wave, 2 waves / SIMD, operands held in registers (no LDS reads)             -- MFMA only
//   T1: 64 ch x  64 px per wave, 2 waves / SIMD, 18 ds_read_b128 per 48 MFMAs (conv_dma_kernel's k-loop: 3 taps share 6 rows)
//   T2: 64 ch x 128 px per wave, 1 wave  / SIMD, 22 ds_read_b128 per 96 MFMAs
//   T3: 128 ch x 64 px per wave, 1 wave  / SIMD, 30 ds_read_b128 per 96 MFMAs
// One block per CU, pseudo-random bf16 operands, one block barrier per 9 k-steps; each configuration runs `secs` seconds back to back.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ inline bf16x8 lds_frag(const uint4* lds, int idx) { return __builtin_bit_cast(bf16x8, lds[idx]); }

template <int MF, int NF, bool LDSREAD, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void k(float* out, int iters, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];   // 96 KB: one block per CU; the first 64 KB are read
    for (int i = threadIdx.x; i < 4096; i += THREADS) {
        unsigned h = (i * 2654435761u) ^ (blockIdx.x * 40503u);
        auto nxt = [&]() { h = h * 1664525u + 1013904223u; return ((h >> 9) & 0x807f807fu) | 0x3f003f00u | ((h >> 3) & 0x00800080u); };
        lds[i] = make_uint4(nxt(), nxt(), nxt(), nxt());
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    f32x4 acc[MF][NF];
    for (int m = 0; m < MF; ++m) for (int n = 0; n < NF; ++n) acc[m][n] = (f32x4){0, 0, 0, 0};
    bf16x8 a[3][MF], b[NF + 2];
    for (int u = 0; u < 3; ++u) for (int m = 0; m < MF; ++m) a[u][m] = lds_frag(lds, (u * MF + m) * 64 + lane);
    for (int n = 0; n < NF + 2; ++n) b[n] = lds_frag(lds, (32 + n) * 64 + lane);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const int base = ((it * 3 + v) * 11 + wave * 17) & 63;   // walks the 64 KB
            if (LDSREAD) {
#pragma unroll
                for (int n = 0; n < NF + 2; ++n) b[n] = lds_frag(lds, ((base + n) & 63) * 64 + lane);
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (LDSREAD) {
#pragma unroll
                    for (int m = 0; m < MF; ++m) a[u][m] = lds_frag(lds, ((base + 16 + u * MF + m) & 63) * 64 + lane);
                }
#pragma unroll
                for (int m = 0; m < MF; ++m)
#pragma unroll
                    for (int n = 0; n < NF; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][m], b[n + u], acc[m][n], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_barrier();
    }
    for (int m = 0; m < MF; ++m) for (int n = 0; n < NF; ++n) s += acc[m][n][0];
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MF, int NF, bool LDSREAD, int THREADS>
void run(const char* name, double secs) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * sizeof(float)); hipMalloc(&cyc, 8);
    const int iters = 20000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k<MF, NF, LDSREAD, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    k<MF, NF, LDSREAD, THREADS><<<grid, THREADS, 98304>>>(out, 10, cyc);
    hipDeviceSynchronize();
    const double mfmas = (double)grid * (THREADS / 64) * iters * 9 * MF * NF;   // per launch
    double total_ms = 0, first_ms = 0, last_ms = 0; int launches = 0; long long c = 0;
    while (total_ms < secs * 1e3) {
        hipEventRecord(e0);
        k<MF, NF, LDSREAD, THREADS><<<grid, THREADS, 98304>>>(out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (launches == 0) first_ms = ms;
        last_ms = ms; total_ms += ms; ++launches;
    }
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-66s first %6.0f  last %6.0f TFLOP/s  (%d launches, %.1f s)  last: ticks per 9 k-steps / MFMA-bound %5.3f  ticks/us %.0f\n", name,
           mfmas * 16384 / first_ms / 1e9, mfmas * 16384 / last_ms / 1e9, launches, total_ms / 1e3,
           (double)c / iters / (9.0 * MF * NF * 16 * (THREADS / 256)), (double)c / (last_ms * 1e3));
    fflush(stdout);
    hipFree(out); hipFree(cyc);
}


// T5: 64 ch x 128 px per wave, 1 wave / SIMD, operands of the NEXT k-step fetched during the current one (explicit double buffer,
// reads interleaved with the MFMAs by sched_group_barrier): what a single consumer wave per SIMD can reach when it never waits for LDS
template <int THREADS>
__global__ __launch_bounds__(THREADS, 1) void kp(float* out, int iters, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    for (int i = threadIdx.x; i < 4096; i += THREADS) {
        unsigned h = (i * 2654435761u) ^ (blockIdx.x * 40503u);
        auto nxt = [&]() { h = h * 1664525u + 1013904223u; return ((h >> 9) & 0x807f807fu) | 0x3f003f00u | ((h >> 3) & 0x00800080u); };
        lds[i] = make_uint4(nxt(), nxt(), nxt(), nxt());
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int MF = 4, NF = 8;
    f32x4 acc[MF][NF];
    for (int m = 0; m < MF; ++m) for (int n = 0; n < NF; ++n) acc[m][n] = (f32x4){0, 0, 0, 0};
    const long long t0 = __builtin_amdgcn_s_memtime();
    bf16x8 a[2][MF], rows[12];   // rows ring: 10 live (NF + 2) + the next column's first two
    auto ld_a = [&](int t, int slot, int base) {
#pragma unroll
        for (int m = 0; m < MF; ++m) a[slot][m] = lds_frag(lds, ((base + 16 + t * MF + m) & 63) * 64 + lane);
    };
    for (int it = 0; it < iters; ++it) {
        const int base = (it * 11 + wave * 17) & 63;
#pragma unroll
        for (int r = 0; r < NF; ++r) rows[r] = lds_frag(lds, ((base + r) & 63) * 64 + lane);
        ld_a(0, 0, base);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int v = t / 3, u = t % 3;
            if (t + 1 < 9) {
                const int v1 = (t + 1) / 3, u1 = (t + 1) % 3;
                if (u1 == 0) {
#pragma unroll
                    for (int r = 0; r < NF; ++r) rows[(10 * v1 + r) % 12] = lds_frag(lds, ((base + 3 * v1 + r) & 63) * 64 + lane);
                } else {
                    rows[(10 * v1 + NF - 1 + u1) % 12] = lds_frag(lds, ((base + 3 * v1 + NF - 1 + u1) & 63) * 64 + lane);
                }
                ld_a(t + 1, (t + 1) & 1, base);
            }
#pragma unroll
            for (int m = 0; m < MF; ++m)
#pragma unroll
                for (int n = 0; n < NF; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t & 1][m], rows[(10 * v + u + n) % 12], acc[m][n], 0, 0, 0);
            if (t + 1 < 9) {
                if ((t + 1) % 3 == 0) {
#pragma unroll
                    for (int r = 0; r < 12; ++r) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); }
                    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                } else {
#pragma unroll
                    for (int r = 0; r < 5; ++r) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); }
                    __builtin_amdgcn_sched_group_barrier(0x008, 22, 0);
                }
            }
        }
        __builtin_amdgcn_s_barrier();
    }
    float s = 0.f;
    for (int m = 0; m < MF; ++m) for (int n = 0; n < NF; ++n) s += acc[m][n][0];
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

void run_kp(const char* name, double secs) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * sizeof(float)); hipMalloc(&cyc, 8);
    const int iters = 20000, grid = 256, THREADS = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)kp<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    kp<256><<<grid, THREADS, 98304>>>(out, 10, cyc);
    hipDeviceSynchronize();
    const double mfmas = (double)grid * (THREADS / 64) * iters * 9 * 32;
    double total_ms = 0, last_ms = 0; long long c = 0;
    while (total_ms < secs * 1e3) {
        hipEventRecord(e0);
        kp<256><<<grid, THREADS, 98304>>>(out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        last_ms = ms; total_ms += ms;
    }
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-66s last %6.0f TFLOP/s  ticks per 9 k-steps / MFMA-bound %5.3f  ticks/us %.0f\n", name, mfmas * 16384 / last_ms / 1e9,
           (double)c / iters / (9.0 * 32 * 16), (double)c / (last_ms * 1e3));
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 3.0;
    run<4, 4, false, 512>("T0  64x64  2 waves/SIMD, operands in registers", secs);
    run<4, 4, true, 512>("T1  64x64  2 waves/SIMD, 18 ds_read_b128 / 48 MFMA", secs);
    run<4, 8, true, 256>("T2  64x128 1 wave/SIMD,  22 ds_read_b128 / 96 MFMA", secs);
    run<8, 4, true, 256>("T3  128x64 1 wave/SIMD,  30 ds_read_b128 / 96 MFMA", secs);
    run<4, 8, false, 256>("T4  64x128 1 wave/SIMD,  operands in registers", secs);
    run<4, 4, true, 512>("T1  (again)", secs);
    run_kp("T5  64x128 1 wave/SIMD, next k-step prefetched (double buffer)", secs);
    return 0;
}
