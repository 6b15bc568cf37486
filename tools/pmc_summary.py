"""Summarise what tools/pmc_case.sh collected: per kernel, average duration and average counter values per launch."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
rows = []
for path in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append(r)
print("kernel,calls,avg_us")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print(f'{r["Name"][:90]},{r["Calls"]},{float(r["AverageNs"]) / 1e3:.2f}')
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as fh:
        for r in csv.DictReader(fh):
            acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel,counter,avg_per_launch,launches")
for k, cs in acc.items():
    if not any(t in k for t in ("frcnn::", "crop", "iou", "bm_", "proposal", "decoder", "sort_select", "rpn_targets", "roi_")):
        continue
    for c, v in sorted(cs.items()):
        print(f"{k},{c},{sum(v) / len(v):.1f},{len(v)}")
