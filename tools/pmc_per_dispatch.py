#!/usr/bin/env python3
"""Per-dispatch values of one counter for kernels matching a substring (rocprofv3 --pmc counter_collection.csv).
usage: pmc_per_dispatch.py csv kernel_substring"""
import csv, sys
from collections import defaultdict
path, sub = sys.argv[1], sys.argv[2]
vals = defaultdict(dict)
with open(path) as f:
    for r in csv.DictReader(f):
        if sub in r["Kernel_Name"]:
            vals[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
for d in sorted(vals):
    print(d, " ".join(f"{k}={v:.6g}" for k, v in sorted(vals[d].items())))
