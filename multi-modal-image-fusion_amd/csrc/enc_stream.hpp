// Shared argument block of the streaming DenseBlock-encoder forward kernels (enc_stream.hip: round 2, 32-column strips, bit-identical to
// the layer-wise kernels; enc_stream2.hip: round 5, 64-column strips, input-stationary accumulation).
#pragma once
#include "common.hpp"

namespace mmif {

// packed operand images of mmif_pack_weights (k-group planes of [16 oc][8] bf16 = 256 B): 16->16: 18 -> 20 planes, 32->16: 36, 48->16: 36 + 20
constexpr int ES_P1 = 20, ES_P2 = 36, ES_P3 = 56;

struct EncBranch {
    const float* img;          // [n][h][w] fp32
    const float* w0;           // first layer, [16][1][3][3]
    const float* b0;           // [16] or NULL
    const uint4* wpk[3];       // forward operand images of the three DenseBlock convs
    const float* bias[3];      // [16] each or NULL
    TV out;                    // 8-block view: x0 | x1 | x2 | x3
};
struct EncArgs {
    EncBranch br[2];
    int n, h, w;
    int nstrips, nseg, seg_rows;
    int items;                 // per branch: n * nseg * nstrips
    int relu0;                 // ReLU after the first layer (always 1 on the reference's path; the dense convs always have one)
    TV sum;                    // enc_stream2's dual form only: 8-block view that receives out(br[0]) + out(br[1])
};

// enc_stream2.hip
bool enc_stream2_ok(const EncArgs& A, int nb);
int enc_stream2_launch(EncArgs& A, int nb, hipStream_t st);
int enc_stream2_launch_dual(EncArgs& A, hipStream_t st);

}  // namespace mmif
