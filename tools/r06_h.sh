#!/bin/bash
mkdir -p gpurun_out
timeout 3400 python -m pytest tests -x -q -m gpu > gpurun_out/r06h_tests.log 2>&1
tail -15 gpurun_out/r06h_tests.log
timeout 600 python bench.py > gpurun_out/r06h_bench_line.json 2> gpurun_out/r06h_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06h_bench_line.json').read().strip().splitlines()[-1])
print('headline', round(d['value'],1), d['ms_per_step'], 'frac', d['step_frac_of_ideal'])
print('roofline', d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'])
print('parity', round(d['parity_path']['value'],1), d['parity_path']['rel_err_vs_oracle'], d['parity_path']['grad_rel_err_vs_oracle'], d['parity_path']['roofline']['frac'])
for k,v in d['other_configs'].items(): print(k, round(v['value'],1), round(v['ms_per_step'],3), round(v['step_frac_of_ideal'],3))
print('cpu', d['cpu_baseline']['value'])
PY
