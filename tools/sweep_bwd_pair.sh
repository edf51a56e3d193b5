#!/bin/bash
# Timing ablations of bwd_pair_dma_kernel inside the default bench (run on the GPU box from the repo root):  tools/sweep_bwd_pair.sh r04
# $MMIF_ABLATE bp= bits (results WRONG when non-zero): 1 no tile requests after the first, 2 no gx stores, 4 no weight-gradient loops,
# 8 no dgrad k-loops, 16 no dgrad epilogue, 32 no fold steps.  $MMIF_BWD_PAIR_DMA=0: the register-staged kernel.
tag=$1
out=gpurun_out/${tag}_ubench_bwd_pair_ablation.txt
echo "# avg us per launch inside bench.py (B=32 256x256 bf16, rocprofv3 kernel trace, 13 launches): decode.2 (64->32) | decode.3 (32->16)" > $out
row() {
  tools/prof_bench.sh bp_$1 --no-parity-path > /dev/null 2>&1
  a=$(grep "bwd_pair_$3kernel<4, 2>" gpurun_out/kstats_bp_$1.txt | awk '{print $(NF-3)}')
  b=$(grep "bwd_pair_$3kernel<2, 1>" gpurun_out/kstats_bp_$1.txt | awk '{print $(NF-3)}')
  printf "%-78s %8s %8s\n" "$2" "$a" "$b" >> $out
}
MMIF_BWD_PAIR_DMA=0 row reg "register-staged kernel (round 2: two barriers per tile)" ""
MMIF_ABLATE=bp=0 row 0 "DMA-staged kernel (loader wave, double-buffered tile, one barrier per tile)" "dma_"
MMIF_ABLATE=bp=1 row 1 "  no tile requests after the first" "dma_"
MMIF_ABLATE=bp=2 row 2 "  no gx stores" "dma_"
MMIF_ABLATE=bp=3 row 3 "  neither (compute only)" "dma_"
MMIF_ABLATE=bp=7 row 7 "  compute only, dgrad waves alone (no weight-gradient loops)" "dma_"
MMIF_ABLATE=bp=11 row 11 "  compute only, weight-gradient waves alone (no dgrad k-loops)" "dma_"
MMIF_ABLATE=bp=51 row 51 "  compute only, no dgrad epilogue, no fold steps" "dma_"
MMIF_ABLATE=bp=63 row 63 "  nothing (barriers, index arithmetic, weight image, partial sums)" "dma_"
cat $out
