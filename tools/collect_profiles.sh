#!/bin/bash
# copy one round's evidence from gpurun_out/ (scratch, merged back from the GPU box) into profiles/ (tracked):  tools/collect_profiles.sh r03
tag=$1
for f in bench_line.json decode_fwd_per_dispatch.txt kernel_stats_bench_pfnetv1_b32_256_bf16.txt kernel_stats_bench_pfnetv1_b32_256_fp32_x3.txt \
         pmc_sq_bench_pfnetv1_b32_256_bf16.txt pmc_sq_x3.txt pmc_tcc_fetch_size.txt pmc_tcc_write_size.txt traffic.json \
         kernel_stats_nestfuse_b4_512_bf16.txt kernel_stats_rfnnest_b4_512_bf16.txt kernel_stats_densefuse_b32_256_bf16.txt kernel_stats_infer_1224x1024_bf16.txt ubench_staging_ablation.txt ubench_chain_ablation.txt ubench_bwd_pair_ablation.txt; do
  [ -f gpurun_out/${tag}_$f ] && cp gpurun_out/${tag}_$f profiles/${tag}_$f
done
if [ -f gpurun_out/${tag}_config_sweep.txt ]; then
  (echo "# tools/sweep_configs.sh on one MI355X box (final library of the round; box-to-box spread of the same build: +-4 %)"; cat gpurun_out/${tag}_config_sweep.txt) > profiles/${tag}_config_sweep.txt
fi
sha256sum multi-modal-image-fusion_amd/libmmif_hip.so | cut -c1-64; grep lib_sha profiles/${tag}_traffic.json
