#!/bin/bash
# SQ counters of the bench's kernels (run on the GPU box from the repo root):  tools/prof_sq.sh <tag>
# one rocprofv3 --pmc pass (8 SQ slots): wave cycles, busy / wait breakdown, MFMA busy cycles, LDS activity and bank conflicts
tag=$1
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
rm -rf gpurun_out/sq_$tag
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT \
  --output-format csv -d gpurun_out/sq_$tag -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-path --no-other-configs > gpurun_out/sq_$tag.log 2>&1
csv=$(find gpurun_out/sq_$tag -name "*counter_collection.csv" | head -1)
python3 tools/pmc_stats.py $csv > gpurun_out/sq_$tag.txt
head -30 gpurun_out/sq_$tag.txt
find gpurun_out/sq_$tag -name "*.csv" -size +1M -delete
