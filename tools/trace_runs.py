#!/usr/bin/env python3
"""Runs of consecutive launches of the same kernel in a rocprofv3 kernel_trace.csv (run length >= N)."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n_min = int(sys.argv[2]) if len(sys.argv) > 2 else 10
short = lambda k: k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
i = 0
while i < len(rows):
    j = i
    while j + 1 < len(rows) and rows[j + 1]["Kernel_Name"] == rows[i]["Kernel_Name"]:
        j += 1
    if j - i + 1 >= n_min:
        grids = sorted({int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)) for r in rows[i:j + 1]})
        print(j - i + 1, short(rows[i]["Kernel_Name"]), "| prev:", short(rows[i - 1]["Kernel_Name"]) if i else "-",
              "| next:", short(rows[j + 1]["Kernel_Name"]) if j + 1 < len(rows) else "-", "| grids", grids[:6], "..", grids[-3:])
    i = j + 1
