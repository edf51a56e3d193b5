#!/bin/bash
for i in $(seq 1 10); do
BENCH_DIAG=1 python bench.py --no-cpu-baseline --no-parity-path --no-other-configs 2>&1 >/dev/null | grep DIAG
done
