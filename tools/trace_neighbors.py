#!/usr/bin/env python3
"""For a rocprofv3 kernel_trace.csv: which kernels run right before/after each launch of kernels matching a pattern."""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2]
short = lambda k: k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
c = collections.Counter()
for i, r in enumerate(rows):
    if pat in r["Kernel_Name"]:
        prev = short(rows[i - 1]["Kernel_Name"]) if i else "-"
        nxt = short(rows[i + 1]["Kernel_Name"]) if i + 1 < len(rows) else "-"
        c[(prev, nxt)] += 1
for (p, n), k in c.most_common(25):
    print(k, "after", p, "| before", n)
