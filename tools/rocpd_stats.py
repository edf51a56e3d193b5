#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel stats table
(calls, total / avg / min / max duration, share), like `--stats` CSV.  Usage: rocpd_stats.py db [steps]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("mmif::", "")
    return name[:110]


def main():
    db = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        d = agg.setdefault(short(name), [0, 0, 1 << 62, 0])
        dur = e - s
        d[0] += 1
        d[1] += dur
        d[2] = min(d[2], dur)
        d[3] = max(d[3], dur)
    tot = sum(v[1] for v in agg.values())
    print(f"# {db}: {len(rows)} dispatches, total kernel time {tot / 1e6:.3f} ms ({tot / 1e6 / steps:.3f} ms per step over {steps} steps)")
    print(f"{'kernel':110s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:110s} {v[0]:7d} {v[1] / 1e6:10.3f} {v[1] / v[0] / 1e3:10.2f} {v[2] / 1e3:9.2f} {v[3] / 1e3:9.2f} {100.0 * v[1] / tot:6.2f}")


if __name__ == "__main__":
    main()
