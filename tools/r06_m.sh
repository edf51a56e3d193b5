#!/bin/bash
for i in $(seq 1 12); do
python bench.py --no-cpu-baseline --no-parity-path --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); rk=d['roofline_kernels']; print('run', $i, round(d['value'],1), round(d['ms_per_step'],3), {k: round(v['avg_launch_ms'],3) for k,v in rk.items()}, 'enc', round(d['roofline_hbm']['avg_launch_ms'],3), round(d['roofline_hbm_bwd']['avg_launch_ms'],3))"
done
