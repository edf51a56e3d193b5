#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_image_bwd.py tests/test_gpu_nest.py -x -q > gpurun_out/r06c_tests.log 2>&1
tail -12 gpurun_out/r06c_tests.log
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path --no-other-configs "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'pairs/s', round(d['ms_per_step'],3), 'ms')"; }
for rep in 1 2 3; do
  echo -n "NestFuse skip on : "; run --model NestFuse --batch 4 --size 512
  echo -n "NestFuse skip off: "; MMIF_WGRAD_RAGGED=0 run --model NestFuse --batch 4 --size 512
  echo -n "PFNetv1 image_bwd on : "; run --steps 50 --warmup 15
  echo -n "PFNetv1 image_bwd off: "; MMIF_IMAGE_BWD=0 run --steps 50 --warmup 15
  echo -n "DenseFuse image_bwd on : "; run --model DenseFuse --steps 50 --warmup 15
  echo -n "DenseFuse image_bwd off: "; MMIF_IMAGE_BWD=0 run --model DenseFuse --steps 50 --warmup 15
done > gpurun_out/r06c_ab.txt 2>&1
cat gpurun_out/r06c_ab.txt
tools/prof_bench.sh r06c --no-parity-path > /dev/null 2>&1
cp gpurun_out/kstats_r06c.txt gpurun_out/r06c_kernel_stats_bench_pfnetv1_b32_256_bf16.txt
rm -rf gpurun_out/prof_r06c*/
head -24 gpurun_out/r06c_kernel_stats_bench_pfnetv1_b32_256_bf16.txt
