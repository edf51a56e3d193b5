#!/bin/bash
# Everything profiles/<tag>_* holds, regenerated on the library as built (run on the GPU box from the repo root):  tools/prof_all.sh r06
# then, back in the container:  tools/collect_profiles.sh r06
tag=$1
tools/prof_round.sh $tag > gpurun_out/${tag}_prof_round.log 2>&1
tools/prof_bench.sh ${tag}nf --model NestFuse --batch 4 --size 512 --no-parity-path > /dev/null 2>&1
cp gpurun_out/kstats_${tag}nf.txt gpurun_out/${tag}_kernel_stats_nestfuse_b4_512_bf16.txt
tools/prof_bench.sh ${tag}rf --model RFNNest --batch 4 --size 512 --no-parity-path > /dev/null 2>&1
cp gpurun_out/kstats_${tag}rf.txt gpurun_out/${tag}_kernel_stats_rfnnest_b4_512_bf16.txt
tools/prof_bench.sh ${tag}df --model DenseFuse --no-parity-path > /dev/null 2>&1
cp gpurun_out/kstats_${tag}df.txt gpurun_out/${tag}_kernel_stats_densefuse_b32_256_bf16.txt
tools/prof_bench.sh ${tag}inf --mode infer --batch 1 --size 1024 --width 1224 > /dev/null 2>&1
cp gpurun_out/kstats_${tag}inf.txt gpurun_out/${tag}_kernel_stats_infer_1224x1024_bf16.txt
tools/sweep_configs.sh > gpurun_out/${tag}_config_sweep.txt 2>&1
cp gpurun_out/${tag}_traffic.json profiles/${tag}_traffic.json   # (on the box's copy: the bench line below reads the counter pass of THIS build)
python3 bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
tail -c 1500 gpurun_out/${tag}_bench_line.json
cat gpurun_out/${tag}_config_sweep.txt
rm -rf gpurun_out/prof_${tag}*/ gpurun_out/prof_bp_*/ gpurun_out/pmc_${tag}*/ gpurun_out/sq_${tag}*/ 2>/dev/null
du -sh gpurun_out
