#!/bin/bash
# staging ablations of conv_dma_kernel (run on the GPU box from the repo root):  tools/sweep_staging.sh > gpurun_out/staging.txt
for a in 0 2 8 10 1 0; do MMIF_ABLATE=conv=$a python3 tools/bench_staging.py 2>/dev/null; done
