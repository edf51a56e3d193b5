#!/bin/bash
for i in $(seq 1 8); do
for ev in 8 1000; do
python bench.py --no-cpu-baseline --no-parity-path --no-other-configs --roofline-every $ev 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('every', $ev, 'run', $i, round(d['value'],1), round(d['ms_per_step'],3))"
done
done
