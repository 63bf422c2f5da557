#!/usr/bin/env python3
"""Multi-tap conv micro-benchmark, fused (round 3) vs phased (round 2) bf16x3 kernels of csrc/conv_x3.hip at the
step's shapes (forward = eight or six plane products by population, input gradient = six).  Run on the GPU box."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarse3d_amd import ops

ops.set_matrix_precision("bf16x3")
dev = "cuda"


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


shapes = [(8, 64, 2048, 32, 32, 3, 1, 1), (8, 64, 2048, 64, 64, 3, 2, 2), (8, 64, 2048, 64, 64, 3, 1, 1),
          (8, 64, 2048, 64, 64, 2, 2, 1), (8, 32, 1024, 128, 128, 3, 2, 2), (8, 16, 512, 256, 256, 3, 1, 1),
          (8, 64, 2048, 80, 32, 3, 1, 1), (8, 8, 256, 256, 256, 3, 2, 2)]
res = []
for (B, H, W, Ci, Co, k, dil, pad) in shapes:
    x = torch.randn(B, H, W, Ci, device=dev)
    w = torch.randn(Co, Ci, k, k, device=dev) * 0.05
    sc = torch.rand(Ci, device=dev) + 0.5
    sh = torch.randn(Ci, device=dev) * 0.1
    taps = ops.conv_taps(k, k, dil, pad)
    wp = ops.pack_weights(w, 0)
    src = ops.Source(x, sc, sh)
    out = torch.empty(B, H, W, Co, device=dev)
    part = torch.empty(Co, 2, ops.num_mtiles(B, H, W), device=dev)
    r = dict(shape=[B, H, W, Ci, Co, k, dil])
    for grad in (False, True):
        for fused in ("0", "1", "0", "1"):
            ops.CONV_VARIANT = 4 if str(fused) == "0" else 0
            ms = timeit(lambda: ops.conv_forward([src], wp, None, Co, taps, lrelu=True, out=out, stat_partial=part, grad=grad))
            key = ("six_" if grad else "fwd_") + ("fused" if fused == "1" else "phased")
            r.setdefault(key + "_ms", []).append(round(ms, 4))
        fl = 2.0 * B * H * W * Ci * Co * k * k
        for v in ("fused", "phased"):
            key = ("six_" if grad else "fwd_") + v
            r[key + "_tflops"] = round(fl / min(r[key + "_ms"]) / 1e9, 1)
    print(json.dumps(r), flush=True)
    res.append(r)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/bench_x3f.json", "w"), indent=1)
