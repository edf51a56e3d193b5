// Partial-sum layout and reduction of the DenseBlock encoder's weight gradients, shared by csrc/enc_wgrad.hip (tile kernel, round 2) and
// csrc/enc_bwd.hip (the fused chain + weight-gradient kernel, round 5).
#pragma once
#include "common.hpp"

namespace mmif {

// per-block partial (floats): dW3 [16][48][9] | dW2 [16][32][9] | dW1 [16][16][9] | layer 0 [16 oc][16: taps 0..8, db0, 6 unused] | db1..3
constexpr int EW_OFF3 = 0, EW_OFF2 = 16 * 48 * 9, EW_OFF1 = EW_OFF2 + 16 * 32 * 9, EW_OFF0 = EW_OFF1 + 16 * 16 * 9;
constexpr int EW_OFFB = EW_OFF0 + 256, EW_PER = EW_OFFB + 48;
constexpr int EW_MAXG = 512;

struct EwDst { float* dw0; float* db0; float* dw[3]; float* db[3]; };
// fixed-order sum of G partials (EW_PER floats each) into the four layers' dW / db (accumulate: onto what is there)
int enc_wgrad_reduce_launch(const float* partial, const EwDst& D, int G, int accumulate, hipStream_t st);
// the same for two branches in one launch (same summation order as two launches; shared destinations are summed one after the other)
int enc_wgrad_reduce_pair_launch(const float* pa, const EwDst& Da, int acc_a, const float* pb, const EwDst& Db, int acc_b, int G, hipStream_t st);

}  // namespace mmif
