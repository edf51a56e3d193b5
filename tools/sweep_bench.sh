run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs "$@" 2>&1 | tail -1 | python -c "import sys,json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('$*', '->', round(d['value'],1), d['unit'], round(d['ms_per_step'],3),'ms')
except Exception as e:
    print('$*', 'FAILED', l[-300:])"; }
run --batch 64
run --batch 1
run --batch 3 --size 250 --width 330
run --batch 8 --size 512
run --batch 2 --size 1024
run --model DenseFuse
run --model VIFNet
run --model PFNetv2 --batch 5 --size 200 --width 120
run --model NestFuse --batch 4 --size 512
run --model RFNNest --batch 2 --size 256
run --batch 4 --size 64 --graph
run --mode infer --batch 1 --size 1024 --width 1224
run --model PFNetv2 --mode infer --batch 1 --size 1024 --width 1224
run --dtype fp32 --batch 4 --size 128
run --model PFNetv2 --dtype fp32 --batch 2 --size 96
run --model IFCNN --batch 4 --size 128
