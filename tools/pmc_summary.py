#!/usr/bin/env python3
"""Summarise a rocprofv3 counter_collection.csv of tools/pmc_conv.py: MFMA-busy fraction etc."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"]
    if not any(t in k for t in ("mfma", "bfp", "x3", "pw3", "wgrad_tr")):
        continue
    key = (k.replace("(anonymous namespace)::", "").replace("void ", "")[:44], r["Dispatch_Id"])
    agg.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
seen = set()
for (k, d), v in agg.items():
    if k in seen:
        continue
    seen.add(k)
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    wc = v["SQ_WAVE_CYCLES"]
    print("%-46s mfma_busy %.2f wait_any %.2f wait_inst %.2f waves/simd %.2f ldsconf/clk %.3f ms %.3f" % (
        k, v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), v["SQ_WAIT_ANY"] / wc, v["SQ_WAIT_INST_ANY"] / wc,
        wc * 4 / (cyc * 1024), v["SQ_LDS_BANK_CONFLICT"] / cyc / 256, cyc / 2.4e6) + (" lds_active/clk %.3f" % (v["SQ_LDS_IDX_ACTIVE"] / cyc / 256) if "SQ_LDS_IDX_ACTIVE" in v else ""))
