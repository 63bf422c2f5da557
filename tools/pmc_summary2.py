#!/usr/bin/env python3
"""Per-kernel SQ counter ratios from a rocprofv3 counter_collection.csv (last dispatch of each kernel)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"]
    if not any(s in k for s in ("wgrad_", "conv_bfp", "conv_x3", "conv_mfma", "conv_pw3")):
        continue
    agg.setdefault(k.replace("(anonymous namespace)::", "").replace("void ", "")[:52], {}).setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, disp in agg.items():
    v = disp[sorted(disp, key=int)[-1]]
    wc = v.get("SQ_WAVE_CYCLES", 1.0)
    out = ["%-54s" % k]
    for name, val in v.items():
        if name == "SQ_WAVE_CYCLES":
            out.append("wave_cyc(quad) %.3g" % val)
        elif name in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_MFMA", "SQ_WAVES", "GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
            out.append("%s %.4g" % (name[3:] if name.startswith("SQ_") else name, val))
        else:
            out.append("%s/wc %.3f" % (name[3:], val / wc))
    print(" | ".join(out))
