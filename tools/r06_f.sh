#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_enc_stream.py tests/test_gpu_models.py tests/test_gpu_fullsize_oracle.py -x -q > gpurun_out/r06f_tests.log 2>&1
tail -12 gpurun_out/r06f_tests.log
run() { python bench.py --steps 50 --warmup 15 --no-cpu-baseline --no-parity-path --no-other-configs "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'pairs/s', round(d['ms_per_step'],3), 'ms')"; }
for rep in 1 2 3; do
  echo -n "DenseFuse dup on : "; run --model DenseFuse
  echo -n "DenseFuse dup off: "; MMIF_DGRAD_DUP=0 run --model DenseFuse
done > gpurun_out/r06f_ab.txt 2>&1
cat gpurun_out/r06f_ab.txt
tools/prof_bench.sh r06fdf --model DenseFuse --no-parity-path > /dev/null 2>&1
cp gpurun_out/kstats_r06fdf.txt gpurun_out/r06f_kernel_stats_densefuse_b32_256_bf16.txt
rm -rf gpurun_out/prof_r06f*/
sed -n 1,16p gpurun_out/r06f_kernel_stats_densefuse_b32_256_bf16.txt
