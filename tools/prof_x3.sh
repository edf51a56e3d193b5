#!/bin/bash
# SQ counters of the split-operand (fp32 storage) kernels on decode.0's shape (run on the GPU box from the repo root):  tools/prof_x3.sh <tag>
# two rocprofv3 --pmc passes over tools/bench_x3.py (128 -> 128, B = 32, 256 x 256) -> gpurun_out/<tag>_pmc_sq_x3.txt
tag=$1
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
out=gpurun_out/${tag}_pmc_sq_x3.txt; : > $out
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"; do
  rm -rf gpurun_out/sqx3_$tag
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/sqx3_$tag -o p -- python3 tools/bench_x3.py 128 128 32 256 3 > gpurun_out/sqx3_$tag.log 2>&1
  csv=$(find gpurun_out/sqx3_$tag -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_stats.py $csv >> $out
done
rm -rf gpurun_out/sqx3_$tag
cat $out
