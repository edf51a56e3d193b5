#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: mean of each counter per dispatch."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name).replace("mmif::", "").replace("void ", "")
    return name[:60]


def main():
    path = sys.argv[1]
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k].add(r["Dispatch_Id"])
    names = sorted({c for v in agg.values() for c in v})
    print("kernel".ljust(60), "disp", *[n.replace("SQ_", "")[:16].rjust(17) for n in names])
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", kv[1].get(names[0], 0))):
        n = len(cnt[k])
        print(k.ljust(60), f"{n:4d}", *[f"{v.get(c, 0) / n:17.4g}" for c in names])


if __name__ == "__main__":
    main()
