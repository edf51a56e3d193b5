#!/bin/bash
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_loss.py tests/test_gpu_models.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_trajectory.py tests/test_gpu_scripts.py -x -q > gpurun_out/r06r_tests.log 2>&1
tail -5 gpurun_out/r06r_tests.log
for i in 1 2 3 4; do
python bench.py --no-cpu-baseline --no-parity-path --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('run', $i, round(d['value'],1), round(d['ms_per_step'],3))"
done
