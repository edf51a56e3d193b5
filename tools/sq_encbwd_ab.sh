#!/bin/bash
# LDS counters of the fused encoder backward for the product library and every ab/libmmif_eb_*.so ablation build (GPU box, repo root)
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
for lib in "" ab/libmmif_eb_*.so; do
  tag=$(basename "${lib:-product}" .so)
  rm -rf gpurun_out/sqab_$tag
  MMIF_LIB=${lib:+$R/$lib} timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS \
    --output-format csv -d gpurun_out/sqab_$tag -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-path > gpurun_out/sqab_$tag.log 2>&1
  csv=$(find gpurun_out/sqab_$tag -name "*counter_collection.csv" | head -1)
  echo "== $tag"; python3 tools/pmc_stats.py $csv | grep -i "^kernel\|enc_bwd"
  rm -rf gpurun_out/sqab_$tag
done
