#!/usr/bin/env python3
"""Round 6: the transform-free instance of the fused multi-tap kernel (conv_x3f_kernel<..., PLAIN = true>: input-gradient launches,
whose source dz carries no BatchNorm affine / LeakyReLU) against the general one (c3d_conv_desc.variant & 128), same launch.  GPU box."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarse3d_amd import ops

ops.set_matrix_precision("bf16x3")
dev = "cuda"


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


shapes = [(8, 64, 2048, 32, 32, 3, 1, 1), (8, 64, 2048, 32, 32, 3, 2, 2), (8, 64, 2048, 64, 64, 3, 2, 2), (8, 64, 2048, 64, 64, 3, 1, 1),
          (8, 64, 2048, 64, 64, 2, 2, 1), (8, 64, 2048, 32, 80, 3, 1, 1), (8, 32, 1024, 128, 128, 3, 2, 2), (8, 16, 512, 256, 256, 3, 1, 1),
          (16, 32, 1024, 32, 32, 3, 2, 2), (8, 48, 1808, 32, 32, 3, 2, 2)]
for (B, H, W, Ci, Co, k, dil, pad) in shapes:
    x = torch.randn(B, H, W, Ci, device=dev)
    w = torch.randn(Co, Ci, k, k, device=dev) * 0.05
    taps = ops.negate_taps(ops.conv_taps(k, k, dil, pad))
    wp = ops.pack_weights(w, 0)
    src = ops.Source(x)
    out = {}
    r = dict(shape=[B, H, W, Ci, Co, k, dil])
    for name, var in (("general", 128), ("plain", 0), ("general", 128), ("plain", 0)):
        ops.CONV_VARIANT = var
        o = torch.empty(B, H, W, Co, device=dev)
        ms = timeit(lambda: ops.conv_forward([src], wp, None, Co, taps, out=o, grad=True))
        r.setdefault(name + "_ms", []).append(round(ms, 4))
        out[name] = o
    ops.CONV_VARIANT = 0
    r["ratio"] = round(min(r["plain_ms"]) / min(r["general_ms"]), 3)
    r["bit_identical"] = bool(torch.equal(out["general"], out["plain"]))
    print(json.dumps(r), flush=True)
