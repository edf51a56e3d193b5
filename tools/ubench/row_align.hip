// Micro-benchmark: does the 16-byte misalignment of halo-1 feature maps cost bandwidth?  A padded-domain gradient of a 256-wide image
// is stored with ws = 258 granules per row and its interior starts at granule 1, so a tile's 256- or 512-byte row segment never starts
// on a 128-byte line.  This copies [planes][256][256] interiors tile by tile (one block per tile, 16 B per lane, row segments of TW
// granules) between two tensors laid out (ws, xoff) = (258, 1) as the engine has it, and (264, 8) = rows padded to a line multiple and
// the interior starting on a line, and reports GB/s of payload (read + write).
//   hipcc --offload-arch=gfx950 -O3 -o row_align row_align.hip && ./row_align
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-value"

template <int TW, int MODE>   // MODE 0: copy, 1: read only, 2: write only
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ src, uint4* __restrict__ dst, int ws, int hs, int xoff, int tiles_x, int tiles_y, float* sink, int xcd) {
    // xcd: consecutive blocks go to different XCDs (block b -> XCD b % 8); give every XCD a contiguous run of tiles instead, so that the
    // two tiles that share a cache line are written through the same L2
    const unsigned t = xcd ? (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : blockIdx.x;
    const unsigned tx = t % tiles_x, r1 = t / tiles_x, ty = r1 % tiles_y, pl = r1 / tiles_y;
    const long long plane = (long long)pl * hs * ws;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < TW * 16 / 256; ++i) {
        const int idx = threadIdx.x + 256 * i, r = idx / TW, c = idx % TW;
        const long long a = plane + (long long)(1 + ty * 16 + r) * ws + xoff + tx * TW + c;
        if (MODE == 2) dst[a] = make_uint4(idx, 1, 2, 3);
        else {
            const uint4 v = src[a];
            if (MODE == 0) dst[a] = v; else acc += __uint_as_float(v.x ^ v.w);
        }
    }
    if (MODE == 1 && acc == 123.456f) sink[0] = acc;
}

template <int TW, int MODE>
static void run(const uint4* src, uint4* dst, int planes, int ws, int xoff, float* sink, int xcd = 0) {
    const int hs = 258, tiles_x = 256 / TW, tiles_y = 16;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<TW, MODE>), dim3(planes * tiles_x * tiles_y), dim3(256), 0, 0, src, dst, ws, hs, xoff, tiles_x, tiles_y, sink, xcd);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)planes * 256 * 256 * 16 * (MODE == 0 ? 2 : 1);
    printf("%-10s segment %3d B  ws %3d xoff %d %s: %7.1f GB/s  (%.3f ms)\n", MODE == 0 ? "copy" : (MODE == 1 ? "read" : "write"), TW * 16, ws, xoff, xcd ? "tiles by XCD " : "", bytes / best / 1e6, best);
}

int main() {
    const int planes = 2048;   // 32 images x 64 channel blocks... of 8 channels: 537 MB of payload per tensor
    const size_t bytes = (size_t)planes * 258 * 272 * 16;
    uint4 *a, *b; float* sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 16);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    const int lays[][2] = {{258, 1}, {264, 1}, {264, 8}, {272, 8}, {258, 0}, {256, 0}, {260, 2}, {264, 4}};
    for (int lay = 0; lay < 8; ++lay) {
        const int ws = lays[lay][0], xoff = lays[lay][1];
        run<16, 0>(a, b, planes, ws, xoff, sink); run<32, 0>(a, b, planes, ws, xoff, sink);
        run<16, 1>(a, b, planes, ws, xoff, sink); run<32, 1>(a, b, planes, ws, xoff, sink);
        run<16, 2>(a, b, planes, ws, xoff, sink); run<32, 2>(a, b, planes, ws, xoff, sink);
    }
    for (int lay = 0; lay < 3; lay += 2) {   // the same with an XCD-aware block -> tile map
        const int ws = lays[lay][0], xoff = lays[lay][1];
        run<16, 0>(a, b, planes, ws, xoff, sink, 1); run<32, 0>(a, b, planes, ws, xoff, sink, 1);
        run<16, 1>(a, b, planes, ws, xoff, sink, 1); run<32, 1>(a, b, planes, ws, xoff, sink, 1);
        run<16, 2>(a, b, planes, ws, xoff, sink, 1); run<32, 2>(a, b, planes, ws, xoff, sink, 1);
    }
    return 0;
}
