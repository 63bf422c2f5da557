#!/usr/bin/env python3
"""Micro-benchmark of the wide pointwise kernel (csrc/conv_pw3.hip) at the step's shapes, fused vs phased
(ops.CONV_VARIANT = c3d_conv_desc.variant), bf16x3 engine.  Run on the GPU box: python tools/bench_pw3.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarse3d_amd import ops


def timeit(fn, n=20):
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


def main():
    ops.set_matrix_precision("bf16x3")
    dev = "cuda"
    shapes = [  # B, H, W, Cin, Cout, affine
        (8, 32, 1024, 704, 704, False), (8, 32, 1024, 704, 256, True), (8, 64, 2048, 256, 400, False),
        (8, 32, 1024, 384, 128, True), (8, 16, 512, 768, 256, True), (8, 16, 512, 256, 768, False),
        (8, 32, 1024, 128, 384, False), (8, 64, 2048, 64, 192, False),
    ]
    res = []
    for (B, H, W, Ci, Co, aff) in shapes:
        x = torch.randn(B, H, W, Ci, device=dev)
        w = torch.randn(Co, Ci, 1, 1, device=dev) * 0.05
        sc = torch.rand(Ci, device=dev) + 0.5 if aff else None
        sh = torch.randn(Ci, device=dev) * 0.1 if aff else None
        bias = torch.randn(Co, device=dev)
        wp = ops.pack_weights(w, 0)
        src = ops.Source(x, sc, sh, lrelu=aff)
        out = torch.empty(B, H, W, Co, device=dev)
        part = torch.empty(Co, 2, ops.num_mtiles(B, H, W), device=dev)
        flops = 2.0 * B * H * W * Ci * Co
        r = dict(shape=[B, H, W, Ci, Co], affine=aff)
        for fused in ("0", "1", "2", "0", "1", "2"):
            ops.CONV_VARIANT = {"0": 3, "1": 1, "2": 2}.get(str(fused), 0)
            ms = timeit(lambda: ops.conv_forward([src], wp, bias, Co, [(0, 0)], lrelu=True, out=out, stat_partial=part))
            key = {"0": "phased", "1": "fused8", "2": "fused4"}[fused]
            r.setdefault(key + "_ms", []).append(round(ms, 4))
            r[key + "_tflops"] = round(flops / min(r[key + "_ms"]) / 1e9, 1)
        print(json.dumps(r), flush=True)
        res.append(r)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/bench_pw3.json", "w"), indent=1)


if __name__ == "__main__":
    main()
