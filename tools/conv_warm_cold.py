#!/usr/bin/env python3
"""How much do the step's convolution launches lose when their inputs are cold (as in the step: the tensor was written many launches
ago) against warm (an isolated benchmark that re-reads one tensor: it sits in the 256 MB infinity cache)?  Forward launches with
BatchNorm affine + LeakyReLU on load and statistics off, and input-gradient launches (plain source); outputs rotate with the inputs.
usage (GPU box): python tools/conv_warm_cold.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarse3d_amd import ops
ops.set_matrix_precision("bf16x3")
dev = "cuda"
shapes = [(8, 64, 2048, 32, 32, 3, 1, 1), (8, 64, 2048, 32, 32, 3, 2, 2), (8, 64, 2048, 64, 64, 3, 2, 2), (8, 64, 2048, 64, 64, 2, 2, 1),
          (8, 32, 1024, 128, 128, 3, 2, 2), (8, 64, 2048, 64, 64, 1, 1, 0), (8, 64, 2048, 32, 32, 1, 1, 0), (8, 32, 1024, 192, 704, 1, 1, 0),
          (8, 32, 1024, 704, 256, 1, 1, 0)]
for (B, H, W, Ci, Co, k, dil, pad) in shapes:
    per_set = B * H * W * (Ci + Co) * 4
    nsets = max(2, int(1.5e9 // per_set) + 1)
    w = torch.randn(Co, Ci, k, k, device=dev) * 0.05
    wp = ops.pack_weights(w, 0)
    sc, sh = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.1
    xs = [torch.randn(B, H, W, Ci, device=dev) for _ in range(nsets)]
    os_ = [torch.empty(B, H, W, Co, device=dev) for _ in range(nsets)]
    r = dict(shape=[B, H, W, Ci, Co, k, dil], sets=nsets)
    for kind in ("forward", "dgrad"):
        taps = ops.conv_taps(k, k, dil, pad) if kind == "forward" else ops.negate_taps(ops.conv_taps(k, k, dil, pad))
        srcs = [ops.Source(x, sc, sh, lrelu=True) if kind == "forward" else ops.Source(x) for x in xs]
        for temp in ("warm", "cold"):
            best = 1e9
            for rnd in range(3):
                def fn(i):
                    j = i % nsets if temp == "cold" else 0
                    ops.conv_forward([srcs[j]], wp, None, Co, taps, out=os_[j], grad=(kind == "dgrad"))
                for i in range(nsets): fn(i)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 3 * nsets
                e0.record()
                for i in range(n): fn(i)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / n)
            r[f"{kind}_{temp}"] = round(best, 4)
        r[f"{kind}_cold_over_warm"] = round(r[f"{kind}_cold"] / r[f"{kind}_warm"], 3)
    print(json.dumps(r), flush=True)
