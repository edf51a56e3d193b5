#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_loss.py tests/test_gpu_image_bwd.py tests/test_gpu_models.py -x -q > gpurun_out/r06d_tests.log 2>&1
tail -12 gpurun_out/r06d_tests.log
run() { python bench.py --steps 50 --warmup 15 --no-cpu-baseline --no-parity-path --no-other-configs "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'pairs/s', round(d['ms_per_step'],3), 'ms')"; }
for rep in 1 2 3; do
  echo -n "PFNetv1 new : "; run
  echo -n "PFNetv1 image_bwd off: "; MMIF_IMAGE_BWD=0 run
  echo -n "DenseFuse new : "; run --model DenseFuse
done > gpurun_out/r06d_ab.txt 2>&1
cat gpurun_out/r06d_ab.txt
tools/prof_bench.sh r06d --no-parity-path > /dev/null 2>&1
cp gpurun_out/kstats_r06d.txt gpurun_out/r06d_kernel_stats_bench_pfnetv1_b32_256_bf16.txt
rm -rf gpurun_out/prof_r06d*/
sed -n 8,30p gpurun_out/r06d_kernel_stats_bench_pfnetv1_b32_256_bf16.txt
