#!/usr/bin/env python3
"""Steady-state per-kernel statistics of a bench.py capture: rocprofv3's `--stats` summary divides EVERYTHING of the process by the
number of steps -- including what happens once (the weight upload: ~500 __amd_rocclr_copyBuffer launches; the first step's
layer-by-layer weight packs: ~140 pack_weights_kernel launches; AdamW's state allocation: ~380 fills), which a reader of the
summary takes for per-step work ("124 copies per step").  This takes the kernel TRACE of the same run, cuts it at the first
kernel of every training step (c3d_input_norm) and summarises the LAST `n` complete steps only.
usage: python tools/steady_stats.py <kernel_trace.csv> <out.csv> [n_last_steps = 2]
Output columns as rocprofv3's kernel_stats.csv (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs, StdDev), Calls and
TotalDurationNs PER STEP (mean over the steps taken)."""
import csv
import math
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n_last = int(sys.argv[3]) if len(sys.argv) > 3 else 2
starts = [i for i, r in enumerate(rows) if "input_norm" in r["Kernel_Name"]]
if len(starts) < n_last + 1:
    sys.exit(f"only {len(starts)} training steps in the trace")
# the last step runs to the end of the trace (nothing follows it in bench.py but the read-back of the loss)
a = starts[-n_last]
seg = rows[a:]
per = {}
for r in seg:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    per.setdefault(r["Kernel_Name"], []).append(d)
tot = sum(sum(v) for v in per.values())
out = []
for name, v in per.items():
    mean = sum(v) / len(v)
    sd = math.sqrt(sum((x - mean) ** 2 for x in v) / len(v))
    out.append((name, len(v) / n_last, sum(v) / n_last, mean, 100.0 * sum(v) / tot, min(v), max(v), sd))
out.sort(key=lambda t: -t[2])
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for t in out:
        w.writerow([t[0], f"{t[1]:g}", f"{t[2]:.0f}", f"{t[3]:.1f}", f"{t[4]:.2f}", t[5], t[6], f"{t[7]:.1f}"])
small = [(t[1], t[2]) for t in out if t[3] <= 10000]
print(f"steady state over the last {n_last} steps: {sum(t[1] for t in out):.0f} launches / step, {tot / n_last / 1e6:.3f} ms of kernels / step; "
      f"<= 10 us: {sum(c for c, _ in small):.0f} launches, {sum(d for _, d in small) / 1e6:.3f} ms")
