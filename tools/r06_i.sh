#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_bwd_wide.py tests/test_gpu_models.py tests/test_gpu_enc_stream.py -x -q > gpurun_out/r06i_tests.log 2>&1
tail -5 gpurun_out/r06i_tests.log
run() { python bench.py --steps 50 --warmup 15 --no-cpu-baseline --no-parity-path --no-other-configs "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'pairs/s', round(d['ms_per_step'],3), 'ms')"; }
for rep in 1 2 3; do
  echo -n "PFNetv1 resident on : "; run
  echo -n "PFNetv1 resident off: "; MMIF_ABLATE=conv=64 run
  echo -n "DenseFuse resident on : "; run --model DenseFuse
  echo -n "DenseFuse resident off: "; MMIF_ABLATE=conv=64 run --model DenseFuse
done > gpurun_out/r06i_ab.txt 2>&1
cat gpurun_out/r06i_ab.txt
