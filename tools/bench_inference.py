#!/usr/bin/env python3
"""Single-scan inference latency of the eval forward: eager launches vs hipGraph replay."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarse3d_amd.pc_processor.models import SalsaNextProto
from coarse3d_amd.serving import GraphedInference

dev = "cuda"
torch.manual_seed(1)
m = SalsaNextProto(5, 20, 20, 0).to(dev).eval()
res = {}
for b in (1, 8):
    x = torch.randn(b, 5, 64, 2048, device=dev)
    with torch.no_grad():
        for _ in range(3):
            ref = m(x, return_feat=False)["pred_2d"].clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            m(x, return_feat=False)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / n * 1e3
    gi = GraphedInference(m)
    out = gi(x)["pred_2d"]
    torch.cuda.synchronize()
    same = bool(torch.equal(out, ref))
    t0 = time.perf_counter()
    for _ in range(n):
        gi(x)
    torch.cuda.synchronize()
    graphed = (time.perf_counter() - t0) / n * 1e3
    res[f"bs{b}"] = {"eager_ms": round(eager, 3), "graph_ms": round(graphed, 3), "scans_per_s_graph": round(b / graphed * 1e3, 1),
                     "bit_identical": same}
print(json.dumps({"metric": "eval forward latency, 64x2048x5, fp32 (pred_2d only)", **res}))
