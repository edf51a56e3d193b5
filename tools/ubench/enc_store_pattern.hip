// Store-pattern micro-benchmark for the streaming encoder kernels (csrc/enc_stream2.hip): every wave owns a strip of an image and walks
// down its rows; per row it issues 8 one-KiB store instructions into 8 channel-block planes of a blocked-NHWC buffer
// [n][16 cb][h][w][16 B].  Nothing is computed: what the store pattern alone sustains.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/enc_store_pattern.hip -o tools/ubench/enc_store_pattern && tools/ubench/enc_store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct P { int n, h, w, nstrips, nseg, seg_rows, items, keep, align, split; int waves; };

// split = 1: an instruction covers 32 pixels of TWO planes (2 x 512 B), the encoder kernel's epilogue layout; 0: 64 pixels of ONE plane
template <int MODE>
__global__ __launch_bounds__(256) void k(char* out, P p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + wave;
    if (item >= p.items) return;
    const int branch = blockIdx.y;
    const int strip = item % p.nstrips, seg = (item / p.nstrips) % p.nseg, in_ = item / (p.nstrips * p.nseg);
    const int y_lo = seg * p.seg_rows, y_hi = min(p.h, y_lo + p.seg_rows);
    const int o_lo = strip * p.keep, o_hi = min(p.w, o_lo + p.keep);
    const int r0 = p.align ? o_lo - 4 : max(-1, min(o_lo - 3, p.w - 63));
    const long long plane = (long long)p.h * p.w * 16, img = 16 * plane;
    char* base = out + (long long)in_ * img + (long long)branch * 8 * plane;
    const uint4 v = make_uint4(lane, item, 3, 4);
    for (int y = y_lo; y < y_hi; ++y) {
#pragma unroll
        for (int L = 0; L < 4; ++L) {
            if (MODE == 0) {          // two planes per instruction
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int px = (lane & 31) + 32 * q, cb = lane >> 5, col = r0 + px;
                    if (col >= o_lo && col < o_hi) *reinterpret_cast<uint4*>(base + (2 * L + cb) * plane + ((long long)y * p.w + col) * 16) = v;
                }
            } else {                  // one plane per instruction
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const int col = r0 + lane;
                    if (col >= o_lo && col < o_hi) *reinterpret_cast<uint4*>(base + (2 * L + cb) * plane + ((long long)y * p.w + col) * 16) = v;
                }
            }
        }
    }
}

// reference: every wave its own contiguous run of the same total size
__global__ __launch_bounds__(256) void kc(char* out, long long per_wave) {
    const int lane = threadIdx.x & 63;
    const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    char* b = out + w * per_wave;
    const uint4 v = make_uint4(lane, 2, 3, 4);
    for (long long o = 0; o < per_wave; o += 1024) *reinterpret_cast<uint4*>(b + o + lane * 16) = v;
}

int main() {
    const int n = 32, h = 256, w = 256;
    const size_t bytes = (size_t)n * 16 * h * w * 16;
    char* d;
    CK(hipMalloc(&d, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 20; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int it = 100;
        for (int i = 0; i < it; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-64s %8.1f us  %7.1f GB/s\n", name, ms / it * 1e3, bytes / (ms / it * 1e-3) / 1e9);
    };
    for (int keep : {58, 56, 64, 32}) for (int align = 0; align < 2; ++align) for (int nseg : {3, 2, 6}) {
        if (align && keep != 56 && keep != 64 && keep != 32) continue;
        P p; p.n = n; p.h = h; p.w = w; p.keep = keep; p.align = align;
        p.nstrips = (w + keep - 1) / keep; p.nseg = nseg; p.seg_rows = (h + nseg - 1) / nseg; p.items = n * p.nstrips * p.nseg;
        char nm[128];
        snprintf(nm, sizeof nm, "2 planes/instr  keep %2d %s  %d strips x %d segs (%d blocks)", keep, align ? "aligned" : "r0=o_lo-3", p.nstrips, nseg, 2 * ((p.items + 3) / 4));
        run(nm, [&] { hipLaunchKernelGGL(k<0>, dim3((p.items + 3) / 4, 2), dim3(256), 0, 0, d, p); });
        snprintf(nm, sizeof nm, "1 plane/instr   keep %2d %s  %d strips x %d segs", keep, align ? "aligned" : "r0=o_lo-3", p.nstrips, nseg);
        run(nm, [&] { hipLaunchKernelGGL(k<1>, dim3((p.items + 3) / 4, 2), dim3(256), 0, 0, d, p); });
    }
    for (int blocks : {240, 256, 512, 1024}) {
        char nm[128];
        snprintf(nm, sizeof nm, "contiguous run per wave, %d blocks of 4 waves", blocks);
        const long long per_wave = (long long)(bytes / (blocks * 4)) / 1024 * 1024;
        run(nm, [&] { hipLaunchKernelGGL(kc, dim3(blocks), dim3(256), 0, 0, d, per_wave); });
    }
    return 0;
}
