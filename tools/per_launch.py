#!/usr/bin/env python3
"""rocprofv3 kernel_trace.csv -> per (kernel, grid) launch counts and durations.
usage: per_launch.py <kernel_trace.csv> [name-substring] [--small N]   (--small: only grids of < N workgroups)"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
small = int(sys.argv[sys.argv.index("--small") + 1]) if "--small" in sys.argv else None
d = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    if pat not in name:
        continue
    short = re.sub(r"\(anonymous namespace\)::|void ", "", name).split("(")[0][:58]
    wgs = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(
        1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    d[(short, wgs)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
for (k, wgs), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if small is not None and wgs >= small:
        continue
    tot += sum(v)
    print(f"{k:58s} wgs {wgs:7d}  n {len(v):5d}  avg {sum(v) / len(v):8.1f} us  total {sum(v) / 1e3:8.3f} ms")
print(f"total {tot / 1e3:.3f} ms over the capture")
