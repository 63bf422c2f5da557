#!/usr/bin/env python3
"""Fused nine-tap weight gradients, four + four waves (variant 256) against sixteen (0), with WARM inputs (the same tensors every
launch: they sit in the 256 MB infinity cache) and COLD ones (the launches rotate through enough tensor sets to exceed it) --
does the isolated A/B of profiles/round6_wgrad_roles.md under-represent the in-step gain because its inputs are warm?
usage (GPU box): python tools/ab_wgrad_cold.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarse3d_amd import ops
ops.set_matrix_precision("bf16x3")
dev = "cuda"
for (B, H, W, Ci, Co, k, dil, pad) in ((8, 64, 2048, 32, 32, 3, 2, 2), (8, 64, 2048, 32, 32, 3, 1, 1), (8, 64, 2048, 64, 64, 3, 2, 2), (8, 32, 1024, 128, 128, 3, 2, 2)):
    taps = ops.conv_taps(k, k, dil, pad)
    per_set = B * H * W * (Ci + 3 * Co) * 4
    nsets = max(2, int(1.5e9 // per_set) + 1)               # >= 1.5 GB in rotation
    sets = []
    for i in range(nsets):
        x = torch.randn(B, H, W, Ci, device=dev); dy = torch.randn(B, H, W, Co, device=dev)
        act = torch.randn(B, H, W, Co, device=dev); dzo = torch.empty_like(dy)
        sets.append((ops.Source(x, torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.1, lrelu=True), dy, act, dzo))
    kk = torch.randn(3, Co, device=dev) * 0.1
    out = {}
    for temp in ("warm", "cold"):
        for v in (256, 0):
            best = 1e9
            for rnd in range(3):
                ops.WGRAD_VARIANT = v
                dw = torch.zeros(Co, Ci, k, k, device=dev); db = torch.zeros(Co, device=dev)
                def fn(i):
                    s, dy, act, dzo = sets[i % nsets if temp == "cold" else 0]
                    ops.conv_wgrad(s, dzo, dw, taps, dbias=db, fuse=(dy, act, kk))
                for i in range(nsets): fn(i)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 2 * nsets
                e0.record()
                for i in range(n): fn(i)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / n)
            out[f"{temp}_{'old' if v else 'new'}"] = round(best, 4)
        out[f"{temp}_ratio"] = round(out[f"{temp}_old"] / out[f"{temp}_new"], 3)
    ops.WGRAD_VARIANT = 0
    print(json.dumps(dict(shape=[B, H, W, Ci, Co, k, dil], sets=nsets, **out)), flush=True)
