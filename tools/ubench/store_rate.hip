// Micro-benchmark: sustained global STORE (and load) rate of the chip by store width and waves per CU.
// Every wave writes its own contiguous stream (1 KiB per wave-instruction at 16 B/lane) -- the pattern of an epilogue that stores
// finished rows -- so the figure is the vector-memory store path, not an access-pattern effect.  MODE 0: stores to a 2 GiB region
// (HBM), 1: stores wrapped into an L2-resident 16 MiB window, 2: loads from the 2 GiB region (for comparison).
//   hipcc --offload-arch=gfx950 -O3 -o store_rate store_rate.hip && ./store_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE, int WIDTH>   // WIDTH: bytes per lane (4, 8, 16)
__global__ __launch_bounds__(256) void k(char* buf, long long bytes_per_wave, long long window, float* sink) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long base = wave * bytes_per_wave;
    float acc = 0.f;
    for (long long off = 0; off < bytes_per_wave; off += 64 * WIDTH) {
        long long a = base + off + lane * WIDTH;
        if (MODE == 1) a %= window;
        if (MODE == 2) {
            if (WIDTH == 16) { const uint4 v = *reinterpret_cast<const uint4*>(buf + a); acc += __uint_as_float(v.x ^ v.y ^ v.z ^ v.w); }
            else if (WIDTH == 8) { const uint2 v = *reinterpret_cast<const uint2*>(buf + a); acc += __uint_as_float(v.x ^ v.y); }
            else acc += *reinterpret_cast<const float*>(buf + a);
        } else {
            if (WIDTH == 16) *reinterpret_cast<uint4*>(buf + a) = make_uint4(lane, 1, 2, 3);
            else if (WIDTH == 8) *reinterpret_cast<uint2*>(buf + a) = make_uint2(lane, 1);
            else *reinterpret_cast<unsigned*>(buf + a) = lane;
        }
    }
    if (MODE == 2 && acc == 123.456f) sink[0] = acc;
}

template <int MODE, int WIDTH>
static void run(char* buf, long long total, int waves_per_cu, float* sink) {
    const int blocks = 256 * waves_per_cu / 4;
    const long long per_wave = total / (256ll * waves_per_cu) / 1024 * 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, WIDTH>), dim3(blocks), dim3(256), 0, 0, buf, per_wave, 16ll << 20, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)per_wave * 256 * waves_per_cu;
    printf("%-22s %2d B/lane  %2d waves/CU  %7.1f GB/s  (%.1f B/clk/CU at 2.1 GHz)\n", MODE == 0 ? "store -> HBM" : (MODE == 1 ? "store -> L2 window" : "load <- HBM"),
           WIDTH, waves_per_cu, bytes / ms / 1e6, bytes / ms / 1e6 / 256 / 2.1);
}

int main() {
    const long long total = 2ll << 30;
    char* buf; float* sink;
    hipMalloc(&buf, total + (1 << 20)); hipMalloc(&sink, 16);
    hipMemset(buf, 0, total);
    for (int w : {4, 8, 16, 32}) { run<0, 16>(buf, total, w, sink); }
    for (int w : {8, 32}) { run<0, 8>(buf, total, w, sink); run<0, 4>(buf, total, w, sink); }
    for (int w : {8, 32}) { run<1, 16>(buf, total, w, sink); }
    for (int w : {8, 32}) { run<2, 16>(buf, total, w, sink); }
    return 0;
}
