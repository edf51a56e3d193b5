# One train step's launch sequence (name, duration, idle gap in front) from a rocprofv3 kernel-trace database:  python3 tools/launch_sequence.py gpurun_out/prof_<tag>/<...>.db
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
# find the last occurrence of adam_kernel and print the sequence between the previous adam and it
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
a, b = idx[-2], idx[-1]
prev_end = rows[a][2]
for n, s, e in rows[a + 1:b + 1]:
    print(f"{(s - prev_end) / 1e3:7.1f} gap  {(e - s) / 1e3:8.1f} us  {n[:90]}")
    prev_end = e
