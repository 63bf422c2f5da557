#!/usr/bin/env python3
"""Print a per-step summary of a rocprofv3 kernel_stats.csv (usage: prof_summary.py file.csv nsteps)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot / n / 1e6:.2f} ms/step over {len(rows)} kernels")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 34]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{float(r['TotalDurationNs']) / n / 1e6:8.3f} ms {int(r['Calls']) / n:6.1f} calls  avg {float(r['AverageNs']) / 1e3:8.1f} us  {name[:90]}")
