#!/bin/bash
mkdir -p gpurun_out
for i in 1 2 3; do
python bench.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', round(d['value'],1), round(d['ms_per_step'],3), 'parity', round(d['parity_path']['value'],1), {k: round(v['value'],1) for k,v in d['other_configs'].items()})"
done
python bench.py --model NestFuse --batch 8 --size 512 --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NestFuse B8', round(d['value'],1), round(d['ms_per_step'],3))"
python bench.py --model RFNNest --batch 8 --size 512 --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RFNNest B8', round(d['value'],1), round(d['ms_per_step'],3))"
python bench.py --model NestFuse --batch 16 --size 512 --steps 6 --warmup 2 --no-cpu-baseline --no-parity-path 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NestFuse B16', round(d['value'],1), round(d['ms_per_step'],3))"
