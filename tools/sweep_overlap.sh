#!/bin/bash
# Intra-step overlap experiment (round-4 verdict item 3): decode.0's weight gradient on a second stream with a reduced persistent grid
# against the encoder's backward on the main stream ($MMIF_OVERLAP, $MMIF_OVERLAP_BLOCKS; mmif/engine.py fork_wgrad).
# Usage (GPU box): bash tools/sweep_overlap.sh > gpurun_out/overlap.txt 2>&1
run() {   # label, env..., -- bench args
  local label="$1"; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  for rep in 1 2; do
    env "${envs[@]}" python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-parity-path --roofline-every 1000 "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print('%-44s rep $rep  %8.3f ms/step  %9.1f pairs/s' % ('$label', d['ms_per_step'], d['value']))
except Exception as e:
    print('$label FAILED', l[-300:])"
  done
}
for model in PFNetv1 DenseFuse; do
  echo "== $model B=32 256x256 bf16, eager"
  run "serial (MMIF_OVERLAP=0)" MMIF_OVERLAP=0 -- --model $model
  run "serial, no sign bytes (MMIF_BWD_WIDE=0)" MMIF_OVERLAP=0 MMIF_BWD_WIDE=0 -- --model $model
  for b in 256 224 192 160 128 96; do
    run "overlap, wgrad on $b blocks" MMIF_OVERLAP=1 MMIF_OVERLAP_BLOCKS=$b -- --model $model
  done
  echo "== $model, hipGraph replay (--graph)"
  run "serial (MMIF_OVERLAP=0) --graph" MMIF_OVERLAP=0 -- --model $model --graph
  for b in 224 192 160; do
    run "overlap, wgrad on $b blocks --graph" MMIF_OVERLAP=1 MMIF_OVERLAP_BLOCKS=$b -- --model $model --graph
  done
done
