// Deferred weight-gradient reductions: see reduce_defer.hpp.  The kernel below restates the three reduce kernels' index arithmetic
// (wgrad_dma_reduce: csrc/conv_mfma.hip, taprow_wgrad_reduce: csrc/enc_wgrad.hip, image_out_wgrad_reduce: csrc/conv_image.hip) around the
// same partial_sum(): per output the same loads in the same order, so the results are bit-identical to the separate launches.
#include "reduce_defer.hpp"
#include <string.h>

namespace mmif {

constexpr int RD_MAXJOBS = 8;
constexpr int RD_WD_PER = 64 * 64 * 9 + 64;     // floats of one (input group, output group) partial of wgrad_dma_kernel (WD_PER, csrc/conv_mfma.hip)
struct RedTable { RedJob j[RD_MAXJOBS]; int start[RD_MAXJOBS + 1]; int n; };

// partial_sum() of common.hpp for a VIRTUAL block: thread vt of 64 * sl threads, its red[sl][64] rows at `red`
__device__ inline float partial_sum_v(const float* __restrict__ partial, long long off, long long stride, int G, bool valid, int sl, int vt,
                                      float (*red)[64]) {
    const int o_local = vt & 63, slice = vt >> 6;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (valid) {
        int g = slice;
        for (; g + 3 * sl < G; g += 4 * sl) {
            s0 += partial[g * stride + off];
            s1 += partial[(g + sl) * stride + off];
            s2 += partial[(g + 2 * sl) * stride + off];
            s3 += partial[(g + 3 * sl) * stride + off];
        }
        for (; g < G; g += sl) s0 += partial[g * stride + off];
    }
    red[slice][o_local] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    float t = 0.f;
    if (slice == 0) {
        for (int q = 0; q < sl; q += 4) t += (red[q][o_local] + red[q + 1][o_local]) + (red[q + 2][o_local] + red[q + 3][o_local]);
    }
    return t;
}

__global__ __launch_bounds__(1024) void reduce_multi_kernel(RedTable T) {
    __shared__ float red[16][64];
    int k = 0;
    while (k + 1 < T.n && (int)blockIdx.x >= T.start[k + 1]) ++k;
    const RedJob& J = T.j[k];
    const int lb = blockIdx.x - T.start[k];
    const int sub = J.sl == 16 ? 0 : threadIdx.x >> 8;              // four 256-thread virtual blocks per launch block when sl == 4
    const int vb = J.sl == 16 ? lb : 4 * lb + sub, vt = J.sl == 16 ? threadIdx.x : (threadIdx.x & 255);
    const int idx = vb * 64 + (vt & 63);
    long long off = -1, stride = 0;
    float* dst = nullptr;
    if (vb < J.nvb) {
        if (J.type == RED_WGRAD_DMA) {
            const int cin = J.p0, cout = J.p1, n_icg = J.p2, n_ocg = J.p3;
            const int total_w = cout * cin * 9;
            stride = (long long)(n_icg * n_ocg) * RD_WD_PER;
            if (idx < total_w) {
                const int tap = idx % 9, c = (idx / 9) % cin, o = idx / (9 * cin);
                off = (long long)((c / 64) + n_icg * (o / 64)) * RD_WD_PER + ((o % 64) * 64 + (c % 64)) * 9 + tap;
                dst = J.dw + idx;
            } else if (idx < total_w + cout) {
                const int o = idx - total_w;
                off = (long long)(0 + n_icg * (o / 64)) * RD_WD_PER + 64 * 64 * 9 + (o % 64);
                dst = J.db != nullptr ? J.db + o : nullptr;
            }
        } else if (J.type == RED_TAPROW) {
            const int n_w = J.p0, per = J.p1;
            stride = per;
            if (idx < per) {
                off = idx;
                dst = idx < n_w ? J.dw + idx : (J.db != nullptr ? J.db + (idx - n_w) : nullptr);
            }
        } else {
            const int cin = J.p0, KK = J.p1 * J.p1, n_cg = J.p2, PER = 16 * KK + 1;
            stride = (long long)n_cg * PER;
            if (idx < cin * KK) {
                const int c = idx / KK, tap = idx % KK;
                off = (long long)(c / 16) * PER + (c % 16) * KK + tap;
                dst = J.dw + idx;
            } else if (idx == cin * KK) {
                off = 16 * KK;      // the bias sum lives in channel group 0
                dst = J.db;
            }
        }
    }
    const float t = partial_sum_v(J.partial, off < 0 ? 0 : off, stride, J.G, off >= 0, J.sl, vt, red + 4 * sub);
    if ((vt >> 6) == 0 && off >= 0 && dst != nullptr) *dst = J.accumulate ? *dst + t : t;
}

static struct {
    char* arena = nullptr;
    size_t cap = 0, used = 0;
    RedJob jobs[RD_MAXJOBS];
    int n = 0;
    bool active = false;
    const float* last_slot = nullptr;
} g_rd;

float* defer_ws(float* ws, size_t bytes) {
    g_rd.last_slot = nullptr;
    if (!g_rd.active || g_rd.n >= RD_MAXJOBS) return ws;
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (g_rd.used + need > g_rd.cap) return ws;
    float* slot = reinterpret_cast<float*>(g_rd.arena + g_rd.used);
    g_rd.used += need;
    g_rd.last_slot = slot;
    return slot;
}

bool defer_push(const RedJob& job) {
    if (!g_rd.active || g_rd.last_slot == nullptr || job.partial != g_rd.last_slot || g_rd.n >= RD_MAXJOBS) return false;
    g_rd.jobs[g_rd.n++] = job;
    g_rd.last_slot = nullptr;
    return true;
}

}  // namespace mmif

using namespace mmif;

// arena: device memory the queued partial sums live in until the flush (the sum of the queued layers' weight-gradient workspaces; a layer
// that does not fit runs its reduce at once, as without deferral).  Stream-ordered like everything else: producers, flush and consumers of
// dW / db must be on one stream.
extern "C" int mmif_reduce_defer_begin(void* arena, size_t bytes) {
    MMIF_REQUIRE(arena != nullptr && bytes > 0, "reduce_defer_begin: NULL arena");
    // (jobs still queued belong to a backward pass that was abandoned half way -- an exception in the caller -- and are dropped)
    g_rd.arena = (char*)arena; g_rd.cap = bytes; g_rd.used = 0; g_rd.n = 0; g_rd.active = true; g_rd.last_slot = nullptr;
    return MMIF_OK;
}

// run every queued reduce as one launch; keep_deferring != 0: later producers are queued again (into the slots after the ones in use)
extern "C" int mmif_reduce_defer_flush(int32_t keep_deferring, void* stream) {
    int rc = MMIF_OK;
    if (g_rd.n > 0) {
        RedTable T;
        memset(&T, 0, sizeof(T));
        int nb = 0;
        for (int i = 0; i < g_rd.n; ++i) {
            T.j[i] = g_rd.jobs[i];
            T.start[i] = nb;
            nb += g_rd.jobs[i].sl == 16 ? g_rd.jobs[i].nvb : cdiv(g_rd.jobs[i].nvb, 4);
        }
        T.start[g_rd.n] = nb;
        T.n = g_rd.n;
        hipLaunchKernelGGL(reduce_multi_kernel, dim3(nb), dim3(1024), 0, (hipStream_t)stream, T);
        rc = check_launch("reduce_multi");
        g_rd.n = 0;
    }
    g_rd.last_slot = nullptr;
    if (!keep_deferring) { g_rd.active = false; g_rd.used = 0; }
    return rc;
}

extern "C" int32_t mmif_reduce_defer_pending(void) { return g_rd.n; }
