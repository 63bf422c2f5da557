#!/usr/bin/env python3
"""Idle time between the kernels of one replayed step: reads a rocprofv3 --kernel-trace CSV of tools/ab_step.py, takes the last
replay (from one input_norm kernel to the next / the end) and prints wall time, summed kernel time, the gap histogram and the
largest gaps with the kernels around them.  usage: python tools/trace_gaps.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nm = lambda r: r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
starts = [i for i, r in enumerate(rows) if "input_norm" in nm(r)]
a, b = starts[-2], starts[-1]                      # the last complete step
step = rows[a:b]
t0, t1 = int(step[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
# union of busy intervals (two queues may overlap)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
union, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for s, e in iv[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        gaps.append((s - cur_e, cur_e))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print(f"{len(step)} kernels, wall {(t1 - t0) / 1e6:.3f} ms, summed kernel time {busy / 1e6:.3f} ms, GPU busy (union) {union / 1e6:.3f} ms, idle {(t1 - t0 - union) / 1e6:.3f} ms")
hist = {}
for g, _ in gaps:
    k = "<1us" if g < 1000 else "1-2us" if g < 2000 else "2-4us" if g < 4000 else "4-10us" if g < 10000 else ">10us"
    hist.setdefault(k, [0, 0])
    hist[k][0] += 1
    hist[k][1] += g
print({k: (v[0], round(v[1] / 1e3, 1)) for k, v in hist.items()}, "(count, total us)")
ends = {int(r["End_Timestamp"]): nm(r)[:60] for r in step}
for g, at in sorted(gaps, reverse=True)[:12]:
    nxt = next(nm(r)[:60] for r in step if int(r["Start_Timestamp"]) == at + g)
    print(f"{g / 1e3:7.1f} us after {ends.get(at, '?')}  before {nxt}")
