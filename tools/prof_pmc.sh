#!/bin/bash
# HBM traffic of the bench's kernels from the TCC counters (run on the GPU box from the repo root):  tools/prof_pmc.sh <tag>
# Separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE (they do not fit one pass; --kernel-trace only, as the pool requires)
# -> gpurun_out/pmc_<tag>_{FETCH_SIZE,WRITE_SIZE}.txt (per-kernel means, tools/pmc_stats.py) and per-dispatch lists for the two
#    roofline kernels (tools/pmc_per_dispatch.py).
tag=$1
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_${tag}_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_${tag}_$c -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-path --no-other-configs > gpurun_out/pmc_${tag}_$c.log 2>&1
  csv=$(find gpurun_out/pmc_${tag}_$c -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_stats.py $csv > gpurun_out/pmc_${tag}_$c.txt
  python3 tools/pmc_per_dispatch.py $csv "conv_dma_kernel<false" > gpurun_out/pmc_${tag}_${c}_decode_fwd.txt
  python3 tools/pmc_per_dispatch.py $csv "enc_stream2_fwd_kernel" > gpurun_out/pmc_${tag}_${c}_enc_stream.txt
  head -12 gpurun_out/pmc_${tag}_$c.txt
  python3 tools/make_traffic.py ${tag} > gpurun_out/${tag}_traffic.json 2>/dev/null || true
done
find gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE -name "*.csv" -size +1M -delete
