#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_nest.py tests/test_gpu_models.py tests/test_gpu_x3.py -x -q > gpurun_out/r06j_tests.log 2>&1
tail -5 gpurun_out/r06j_tests.log
tools/prof_bench.sh r06jnf --model NestFuse --batch 4 --size 512 --no-parity-path > /dev/null 2>&1
grep -n "pack_weights\|# gpurun" gpurun_out/kstats_r06jnf.txt
rm -rf gpurun_out/prof_r06j*/
