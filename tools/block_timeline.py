#!/usr/bin/env python3
"""block_timeline.py -- from a rocprofv3 kernel-trace CSV: the kernel timeline of the LAST timed block of a bench.py run
(blocks are separated by the idle time of the barrier + synchronize bracket), one line per launch.

    python tools/block_timeline.py <..._kernel_trace.csv> [min idle us that separates blocks, default 150]
"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
# split where the whole GPU was idle for more than `idle` us
blocks, cur, busy_until = [], [], 0
for s, e, n, q in rows:
    if cur and s - busy_until > idle * 1e3:
        blocks.append(cur)
        cur = []
    cur.append((s, e, n, q))
    busy_until = max(busy_until, e)
blocks.append(cur)
print("blocks (launches, span us):", [(len(b), round((max(x[1] for x in b) - b[0][0]) / 1e3)) for b in blocks])
for b in blocks[-2:]:
    t0 = b[0][0]
    print("---- block of", len(b), "launches, span", round((max(x[1] for x in b) - t0) / 1e3, 1), "us")
    for s, e, n, q in b:
        short = n.split("(")[0].replace("frcnn::", "")[:34]
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q:>3}  {short}")
