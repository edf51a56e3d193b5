#!/usr/bin/env python3
"""Instruction mix of the loops of a gfx950 .s file (hipcc -S --cuda-device-only): for every backward branch, the span it closes with
its MFMA / VALU / SALU / LDS / VMEM counts.  Usage: asm_mix.py file.s [min_mfma]"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
min_mfma = int(sys.argv[2]) if len(sys.argv) > 2 else 20
labels = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = i
for i, l in enumerate(lines):
    m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)|\s+s_branch\s+(\.LBB\d+_\d+)", l)
    if not m: continue
    t = m.group(1) or m.group(2)
    if t in labels and labels[t] < i:
        body = [x.strip() for x in lines[labels[t]:i] if x.startswith("\t") and not x.strip().startswith((".", ";"))]
        c = dict(mfma=0, valu=0, salu=0, lds=0, vmem=0, wait=0, nop=0, br=0, other=0)
        for x in body:
            op = x.split()[0]
            if op.startswith("v_mfma"): c["mfma"] += 1
            elif op.startswith("s_waitcnt"): c["wait"] += 1
            elif op.startswith("s_nop"): c["nop"] += 1
            elif op.startswith(("s_cbranch", "s_branch")): c["br"] += 1
            elif op.startswith("v_"): c["valu"] += 1
            elif op.startswith("s_"): c["salu"] += 1
            elif op.startswith("ds_"): c["lds"] += 1
            elif op.startswith(("buffer_", "global_", "scratch_", "flat_")): c["vmem"] += 1
            else: c["other"] += 1
        if c["mfma"] >= min_mfma:
            print(f"loop {t} lines {labels[t]}..{i}: {len(body)} instrs", c)
