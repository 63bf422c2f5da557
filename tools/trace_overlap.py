#!/usr/bin/env python3
"""Did the contrast branch of a replayed step run UNDER the decoder's backward?  Reads a rocprofv3 --kernel-trace CSV of
tools/ab_step.py and prints, for the last replay, when the branch's kernels ran relative to the backbone backward's first one.
usage: python tools/trace_overlap.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"]
idx = [i for i, r in enumerate(rows) if "infonce_rows_kernel" in name(r)]
i = idx[-1]
sb = max(j for j, r in enumerate(rows) if "softmax_bwd_kernel" in name(r) and j < i + 400)
t0 = int(rows[sb]["Start_Timestamp"])
print("softmax_bwd (start of the backbone backward) at 0 us")
for j in range(max(0, min(i, sb) - 60), max(i, sb) + 40):
    r = rows[j]
    n = name(r).replace("void ", "").replace("(anonymous namespace)::", "")[:70]
    if any(k in n for k in ("infonce", "entropy_stats", "group_compact", "anchor_sample", "scatter_rows", "bilinear_bwd_rows", "softmax_bwd", "pl_select",
                            "focal", "lovasz", "wgrad_tr", "conv_x3f", "bilinear_rows")):
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} .. {(int(r['End_Timestamp']) - t0) / 1e3:9.1f} us  q{r.get('Queue_Id', '?')}  {n}")
