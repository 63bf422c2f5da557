#!/usr/bin/env python3
"""GPU idle time between kernels from a rocprofv3 --kernel-trace CSV: tools/gpu_gaps.py kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# take the last 60 % of the trace (steady-state steps)
t0 = ev[int(len(ev) * 0.4)][0]
ev = [e for e in ev if e[0] >= t0]
span = ev[-1][1] - ev[0][0]
busy, cur_end, gaps = 0, ev[0][0], []
for s, e, n in ev:
    if s > cur_end:
        gaps.append((s - cur_end, n))
        busy += e - s
        cur_end = e
    elif e > cur_end:
        busy += e - cur_end
        cur_end = e
print("span %.2f ms, busy %.2f ms (%.1f %%), %d kernels, %d gaps, idle %.2f ms" % (span / 1e6, busy / 1e6, 100.0 * busy / span, len(ev), len(gaps), (span - busy) / 1e6))
gaps.sort(reverse=True)
big = [g for g in gaps if g[0] > 20000]
print("gaps > 20 us: %d totalling %.2f ms" % (len(big), sum(g[0] for g in big) / 1e6))
for g, n in gaps[:15]:
    print("%8.1f us before %s" % (g / 1e3, n[:90]))
