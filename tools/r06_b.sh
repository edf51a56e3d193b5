#!/bin/bash
# round 6, ragged channel groups: parity tests, A/B of NestFuse / RFN-Nest, kernel stats
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_nest.py tests/test_gpu_bwd_wide.py -x -q > gpurun_out/r06b_tests.log 2>&1
tail -8 gpurun_out/r06b_tests.log
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path --no-other-configs "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'pairs/s', round(d['ms_per_step'],3), 'ms')"; }
for rep in 1 2; do
for m in NestFuse RFNNest; do
  echo -n "$m ragged on : "; run --model $m --batch 4 --size 512
  echo -n "$m fwd/dgrad only: "; MMIF_WGRAD_RAGGED=0 run --model $m --batch 4 --size 512
  echo -n "$m wgrad only: "; MMIF_CONV_RAGGED_MB=0 run --model $m --batch 4 --size 512
  echo -n "$m ragged off: "; MMIF_CONV_RAGGED_MB=0 MMIF_WGRAD_RAGGED=0 run --model $m --batch 4 --size 512
done
done > gpurun_out/r06b_ab.txt 2>&1
cat gpurun_out/r06b_ab.txt
tools/prof_bench.sh r06bnf --model NestFuse --batch 4 --size 512 --no-parity-path > /dev/null 2>&1
cp gpurun_out/kstats_r06bnf.txt gpurun_out/r06b_kernel_stats_nestfuse_b4_512_bf16.txt
rm -rf gpurun_out/prof_r06b*/
head -24 gpurun_out/r06b_kernel_stats_nestfuse_b4_512_bf16.txt
