// Deferred weight-gradient reductions (round 5): every weight-gradient kernel of the decoder writes per-block partial sums that a small
// fixed-order reduce launch turns into dW / db -- five launches of 5-9 us each per PFNetv1 step, whatever little they do.  Between
// mmif_reduce_defer_begin() and mmif_reduce_defer_flush() those launches are QUEUED instead (their partials go to slots of a caller-supplied
// arena, so that later producers do not overwrite them) and flush runs them as ONE launch: the same per-output arithmetic in the same
// order (bit-identical results), one launch latency instead of five.
#pragma once
#include "common.hpp"

namespace mmif {

enum RedType { RED_WGRAD_DMA = 0, RED_TAPROW = 1, RED_IMAGE_OUT = 2 };
struct RedJob {
    const float* partial;
    float* dw;
    float* db;
    int type, sl;          // reduce flavour; slices of its fixed-order sum (16: 1024-thread blocks, 4: 256-thread blocks, four per launch block)
    int G, accumulate;
    int p0, p1, p2, p3;    // WGRAD_DMA: cin, cout, n_icg, n_ocg | TAPROW: n_w, per | IMAGE_OUT: cin, ksize, n_cg
    int nvb;               // virtual blocks of 64 outputs
};
// the buffer a producer should write its partials to: an arena slot while reductions are being deferred (and the arena / queue have room), else ws
float* defer_ws(float* ws, size_t bytes);
// queue the reduce of `partial` (true: queued, it runs at the next flush) -- only partials handed out by defer_ws() are queued
bool defer_push(const RedJob& job);

}  // namespace mmif
