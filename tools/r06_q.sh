#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_scripts.py -x -q > gpurun_out/r06q_tests.log 2>&1
tail -4 gpurun_out/r06q_tests.log
tools/sweep_configs.sh > gpurun_out/r06_config_sweep.txt 2>&1
cat gpurun_out/r06_config_sweep.txt
for i in 1 2 3; do
python3 bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_line.json').read().strip().splitlines()[-1])
print('headline', round(d['value'],1), round(d['ms_per_step'],3), 'frac', round(d['step_frac_of_ideal'],3), 'roof', round(d['roofline']['frac'],3), d['roofline']['traffic'], 'parity', round(d['parity_path']['value'],1), {k: round(v['value'],1) for k,v in d['other_configs'].items()})
PY
done
