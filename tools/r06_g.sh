#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_x3.py tests/test_gpu_conv.py tests/test_gpu_models.py tests/test_gpu_nest.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_enc_stream.py -x -q > gpurun_out/r06g_tests.log 2>&1
tail -12 gpurun_out/r06g_tests.log
run() { python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-parity-path --no-other-configs "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'pairs/s', round(d['ms_per_step'],3), 'ms')"; }
for rep in 1 2; do
  echo -n "PFNetv1 fp32: "; run --dtype fp32
  echo -n "DenseFuse fp32: "; run --dtype fp32 --model DenseFuse
  echo -n "NestFuse fp32: "; run --dtype fp32 --model NestFuse --batch 4 --size 512 --steps 6 --warmup 2
done > gpurun_out/r06g_ab.txt 2>&1
cat gpurun_out/r06g_ab.txt
tools/prof_bench.sh r06gfp32 --dtype fp32 > /dev/null 2>&1
cp gpurun_out/kstats_r06gfp32.txt gpurun_out/r06g_kernel_stats_bench_pfnetv1_b32_256_fp32_x3.txt
rm -rf gpurun_out/prof_r06g*/
sed -n 1,14p gpurun_out/r06g_kernel_stats_bench_pfnetv1_b32_256_fp32_x3.txt
