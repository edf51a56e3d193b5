#!/bin/bash
# All of a round's profile evidence for the default bench in one go (run on the GPU box from the repo root):  tools/prof_round.sh r03
#   kernel stats (bf16 default + fp32 parity path), per-dispatch durations of the two decode convs, SQ counters, TCC traffic passes,
#   traffic json stamped with the library build.  Copy gpurun_out/<tag>_* into profiles/ afterwards.
tag=$1
tools/prof_bench.sh ${tag} --no-parity-path > /dev/null 2>&1
cp gpurun_out/kstats_${tag}.txt gpurun_out/${tag}_kernel_stats_bench_pfnetv1_b32_256_bf16.txt
db=$(find gpurun_out/prof_${tag} -name "*.db" | head -1)
python3 - "$db" > gpurun_out/${tag}_decode_fwd_per_dispatch.txt <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
d = [(e - s) / 1e3 for n, s, e in rows if "conv_dma_kernel<false" in n]
print("# conv_dma_kernel<false, 0> dispatches in launch order: decode.0 (128->128) and decode.1 (128->64) alternate per step; durations in us")
d0, d1 = d[0::2], d[1::2]
print("decode.0 fwd:", " ".join(f"{v:.1f}" for v in d0), f"| mean {sum(d0) / len(d0):.1f}")
print("decode.1 fwd:", " ".join(f"{v:.1f}" for v in d1), f"| mean {sum(d1) / len(d1):.1f}")
for pat, nm in (("conv_dma_kernel<true", "dgrad decode.1 / decode.0"), ("wgrad_dma_kernel", "wgrad decode.1 / decode.0")):   # (the backward runs decode.1 first)
    d = [(e - s) / 1e3 for n, s, e in rows if pat in n]
    if d:
        a, b = d[0::2], d[1::2]
        print(f"{nm} (first launched / second launched per step): mean {sum(a) / len(a):.1f} / {sum(b) / max(1, len(b)):.1f}")
PY
tools/prof_bench.sh ${tag}fp32 --dtype fp32 > /dev/null 2>&1
cp gpurun_out/kstats_${tag}fp32.txt gpurun_out/${tag}_kernel_stats_bench_pfnetv1_b32_256_fp32_x3.txt
tools/prof_sq.sh ${tag} > /dev/null 2>&1
cp gpurun_out/sq_${tag}.txt gpurun_out/${tag}_pmc_sq_bench_pfnetv1_b32_256_bf16.txt
tools/prof_pmc.sh ${tag} > /dev/null 2>&1
cp gpurun_out/pmc_${tag}_FETCH_SIZE.txt gpurun_out/${tag}_pmc_tcc_fetch_size.txt
cp gpurun_out/pmc_${tag}_WRITE_SIZE.txt gpurun_out/${tag}_pmc_tcc_write_size.txt
