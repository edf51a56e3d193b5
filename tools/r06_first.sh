#!/bin/bash
# round 6, first GPU call: new tests, the bench line with other_configs, baseline kernel stats of configs 3 / 4 / 5
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_scripts.py -x -q -s -k "trajectory or other_configs or preflight or contract or 300" > gpurun_out/r06a_tests.log 2>&1
tail -15 gpurun_out/r06a_tests.log
timeout 600 python bench.py > gpurun_out/r06a_bench_line.json 2> gpurun_out/r06a_bench.err
tail -c 2500 gpurun_out/r06a_bench_line.json
tools/prof_bench.sh r06anf --model NestFuse --batch 4 --size 512 --no-parity-path > /dev/null 2>&1
cp gpurun_out/kstats_r06anf.txt gpurun_out/r06a_kernel_stats_nestfuse_b4_512_bf16.txt
tools/prof_bench.sh r06arf --model RFNNest --batch 4 --size 512 --no-parity-path > /dev/null 2>&1
cp gpurun_out/kstats_r06arf.txt gpurun_out/r06a_kernel_stats_rfnnest_b4_512_bf16.txt
tools/prof_bench.sh r06adf --model DenseFuse --no-parity-path > /dev/null 2>&1
cp gpurun_out/kstats_r06adf.txt gpurun_out/r06a_kernel_stats_densefuse_b32_256_bf16.txt
tools/prof_bench.sh r06ainf --mode infer --batch 1 --size 1024 --width 1224 > /dev/null 2>&1
cp gpurun_out/kstats_r06ainf.txt gpurun_out/r06a_kernel_stats_infer_1224x1024_bf16.txt
tools/prof_bench.sh r06a --no-parity-path > /dev/null 2>&1
cp gpurun_out/kstats_r06a.txt gpurun_out/r06a_kernel_stats_bench_pfnetv1_b32_256_bf16.txt
rm -rf gpurun_out/prof_r06a*/
head -30 gpurun_out/r06a_kernel_stats_infer_1224x1024_bf16.txt
