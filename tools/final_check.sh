#!/bin/bash
mkdir -p gpurun_out
timeout 3400 python -m pytest tests -x -q -m gpu > gpurun_out/r06z_tests.log 2>&1
tail -6 gpurun_out/r06z_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
for i in 1 2; do
python3 bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_line.json').read().strip().splitlines()[-1])
print('headline', round(d['value'],1), round(d['ms_per_step'],3), 'frac', round(d['step_frac_of_ideal'],3), 'roof', round(d['roofline']['frac'],3), d['roofline']['traffic'], 'parity', round(d['parity_path']['value'],1), {k: round(v['value'],1) for k,v in d['other_configs'].items()})
PY
done
