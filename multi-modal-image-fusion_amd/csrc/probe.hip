// Hardware layout probes (diagnostics only, used by tests/test_gpu_probe.py): dump what
// ds_read_b64_tr_b16 and v_mfma_f32_16x16x32_bf16 actually do on this chip so that the lane
// mappings assumed in conv_mfma.hip are checked against silicon, not documentation.
#include "common.hpp"

namespace mmif {
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// LDS holds u16 element e at index e (0..1023); lane l supplies byte address 8*perm[l]; out[l*4+j]
// = the element index each lane received.
__global__ void probe_tr16_kernel(const int* __restrict__ perm, short* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(lds + perm[threadIdx.x] * 4));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = t[j];
}

// D = A(16x32) * B(32x16) with A, B given as row-major bf16 [16][32] / [32][16] in global memory,
// operands gathered with the ASSUMED mapping (lane l: i|j = l&15, k = 8*(l>>4)+e); out[16][16] is
// written with the ASSUMED C/D mapping (row = 4*(l>>4)+r, col = l&15).
__global__ void probe_mfma_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, float* __restrict__ out) {
    const int l = threadIdx.x, ij = l & 15, g = l >> 4;
    uint16_t a[8], b[8];
    for (int e = 0; e < 8; ++e) {
        a[e] = A[ij * 32 + 8 * g + e];
        b[e] = B[(8 * g + e) * 16 + ij];
    }
    uint4 ua, ub;
    ua.x = a[0] | (a[1] << 16); ua.y = a[2] | (a[3] << 16); ua.z = a[4] | (a[5] << 16); ua.w = a[6] | (a[7] << 16);
    ub.x = b[0] | (b[1] << 16); ub.y = b[2] | (b[3] << 16); ub.z = b[4] | (b[5] << 16); ub.w = b[6] | (b[7] << 16);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[(4 * g + r) * 16 + ij] = acc[r];
}

// LDS-DMA (global_load_lds_dwordx4): 4 waves; wave w, lane l fetches the 16-byte granule src[idx[64w + l]] straight into LDS
// at (wave-uniform base given for wave w) + 16*l -- the ASSUMED destination mapping.  out = the LDS image, so the test sees
// where every lane's granule landed.  The base of wave w is lds + 64*slot[w] granules (slot: any permutation of 0..3).
__global__ void probe_dma_kernel(const uint4* __restrict__ src, const int* __restrict__ idx, const int* __restrict__ slot,
                                 uint4* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint4 lds[256];
    const int wave = threadIdx.x >> 6;
    lds[threadIdx.x] = make_uint4(0xdeadbeefu, 0, 0, 0);
    __syncthreads();
    const int base = __builtin_amdgcn_readfirstlane(slot[wave]) * 64;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + idx[threadIdx.x]),
                                     (__attribute__((address_space(3))) void*)(lds + base), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's DMA has landed
    __syncthreads();
    out[threadIdx.x] = lds[threadIdx.x];
}
}  // namespace mmif

extern "C" int mmif_probe_dma(const void* src, const int* idx, const int* slot, void* out, void* stream) {
    hipLaunchKernelGGL(mmif::probe_dma_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, idx, slot, (uint4*)out);
    return mmif::check_launch("probe_dma");
}
extern "C" int mmif_probe_tr16(const int* perm, short* out, void* stream) {
    hipLaunchKernelGGL(mmif::probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, perm, out);
    return mmif::check_launch("probe_tr16");
}
extern "C" int mmif_probe_mfma(const void* A, const void* B, float* out, void* stream) {
    hipLaunchKernelGGL(mmif::probe_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const mmif::bf16_t*)A,
                       (const mmif::bf16_t*)B, out);
    return mmif::check_launch("probe_mfma");
}
