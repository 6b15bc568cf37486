#!/usr/bin/env python3
"""trace_gaps.py -- from a rocprofv3 rocpd database: per-kernel durations and the idle gaps between consecutive
launches of the dominant kernel (how well the stage pipeline keeps the HBM-bound stage busy).

    python tools/trace_gaps.py gpurun_out/x/prof/*_results.db [kernel-substring]
"""
import sqlite3
import sys

import numpy as np

db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else "crop_resize"
rows = db.execute("select name, start, end from kernels order by start").fetchall()
names = sorted({r[0] for r in rows})
for n in names:
    d = np.array([r[2] - r[1] for r in rows if r[0] == n]) / 1e3
    print(f"{n[:60]:60s} calls {len(d):5d} avg {d.mean():8.1f} us  min {d.min():8.1f}  max {d.max():8.1f}")
k = [(r[1], r[2]) for r in rows if sub in r[0]]
gaps = np.array([k[i + 1][0] - k[i][1] for i in range(len(k) - 1)]) / 1e3
steady = gaps[gaps < 1000]
print(f"gaps between consecutive '{sub}' launches: median {np.median(steady):.1f} us, mean {steady.mean():.1f} us, p90 {np.percentile(steady, 90):.1f} us (n={len(steady)})")
per = np.array([k[i + 1][0] - k[i][0] for i in range(len(k) - 1)]) / 1e3
per = per[per < 1000]
print(f"start-to-start period: median {np.median(per):.1f} us, mean {per.mean():.1f} us")
