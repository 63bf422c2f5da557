#!/usr/bin/env python3
"""Cold-input timing of the HBM-bound 1x1 weight gradients (launches rotate through >= 1.5 GB of tensor sets), fused and unfused --
for a same-box A/B of two library builds through C3D_LIB (round 6: two against four tiles of loads in flight per CU).
usage (GPU box): [C3D_LIB=...] python tools/wgrad_1x1_cold.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarse3d_amd import ops
ops.set_matrix_precision("bf16x3")
dev = "cuda"
out = {}
for (B, H, W, Ci, Co) in ((8, 64, 2048, 64, 64), (8, 64, 2048, 32, 64), (8, 64, 2048, 32, 32), (8, 32, 1024, 128, 128), (8, 32, 1024, 64, 128), (8, 16, 512, 256, 256)):
    taps = ops.conv_taps(1, 1, 1, 0)
    per_set = B * H * W * (Ci + 3 * Co) * 4
    nsets = max(2, int(1.5e9 // per_set) + 1)
    sets = []
    for i in range(nsets):
        x = torch.randn(B, H, W, Ci, device=dev); dy = torch.randn(B, H, W, Co, device=dev)
        act = torch.randn(B, H, W, Co, device=dev); dzo = torch.empty_like(dy)
        sets.append((ops.Source(x, torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.1, lrelu=True), dy, act, dzo))
    kk = torch.randn(3, Co, device=dev) * 0.1
    for fused in (True, False):
        if fused and not ops.wgrad_fusable(sets[0][0], sets[0][2], Co):
            continue
        best = 1e9
        for rnd in range(3):
            dw = torch.zeros(Co, Ci, 1, 1, device=dev); db = torch.zeros(Co, device=dev)
            def fn(i):
                s, dy, act, dzo = sets[i % nsets]
                if fused: ops.conv_wgrad(s, dzo, dw, taps, dbias=db, fuse=(dy, act, kk))
                else: ops.conv_wgrad(s, dy, dw, taps)
            for i in range(nsets): fn(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 3 * nsets
            e0.record()
            for i in range(n): fn(i)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / n)
        byts = B * H * W * 4 * (Ci + (3 * Co if fused else Co))
        out[f"{H}x{W} {Ci}->{Co} {'fused' if fused else 'plain'}"] = [round(best, 4), round(byts / best / 1e9, 2)]
print(json.dumps(out), flush=True)
