#!/usr/bin/env python3
"""Matrix-pipe busy / wait fractions per kernel instance over ALL dispatches of a rocprofv3 --pmc run of bench.py
(counters: SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE
SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE).  usage: pmc_step_summary.py counter_collection.csv [steps]

mfma_busy = sum SQ_VALU_MFMA_BUSY_CYCLES / (sum GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); clock = GRBM cycles / kernel time."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
disp = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if not any(t in k for t in ("mfma", "bfp", "x3", "pw3", "wgrad_tr")):
        continue
    d = disp.setdefault((k, r["Dispatch_Id"]), {"t": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.OrderedDict()
for (k, _), v in disp.items():
    a = agg.setdefault(k, collections.Counter())
    for c, x in v.items():
        a[c] += x
    a["n"] += 1
print("| kernel | launches/step | ms/step (profiled) | matrix pipe busy | waiting (s_waitcnt / barrier) | issue-stalled | LDS active / clk | clock GHz |")
print("|---|---|---|---|---|---|---|---|")
tot_busy = tot_cyc = 0.0
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["t"]):
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    wc = v["SQ_WAVE_CYCLES"]
    tot_busy += v["SQ_VALU_MFMA_BUSY_CYCLES"]
    tot_cyc += cyc * 1024
    print("| `%s` | %.0f | %.3f | %.2f | %.2f | %.2f | %.3f | %.2f |" % (
        k, v["n"] / steps, v["t"] / steps / 1e6, v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), v["SQ_WAIT_ANY"] / wc,
        v["SQ_WAIT_INST_ANY"] / wc, v.get("SQ_LDS_IDX_ACTIVE", 0) / cyc / 256, cyc / v["t"]))
print("\nall MFMA kernels: matrix pipe busy %.3f of their time" % (tot_busy / tot_cyc))
