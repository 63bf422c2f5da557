#!/usr/bin/env python3
"""conv / wgrad in the bf16 matrix mode with fp32 vs bf16 tensors (run on the GPU box)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarse3d_amd import ops
ops.set_matrix_precision("bf16")
dev = "cuda"
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (B, H, W, Ci, Co, k, dil, pad) in [(8, 32, 1024, 704, 704, 1, 1, 0), (8, 64, 2048, 64, 64, 3, 2, 2), (8, 64, 2048, 32, 32, 3, 1, 1)]:
    x = torch.randn(B, H, W, Ci, device=dev); dz = torch.randn(B, H, W, Co, device=dev)
    w = torch.randn(Co, Ci, k, k, device=dev) * 0.05
    sc = torch.rand(Ci, device=dev) + 0.5; sh = torch.randn(Ci, device=dev) * 0.1
    taps = ops.conv_taps(k, k, dil, pad); wp = ops.pack_weights(w, 0); dw = torch.zeros_like(w)
    for dt in (torch.float32, torch.bfloat16):
        xs, dzs = x.to(dt), dz.to(dt)
        out = torch.empty(B, H, W, Co, device=dev, dtype=dt)
        part = torch.empty(Co, 2, ops.num_mtiles(B, H, W), device=dev)
        t_c = timeit(lambda: ops.conv_forward([ops.Source(xs, sc, sh)], wp, None, Co, taps, lrelu=True, out=out, stat_partial=part))
        t_w = timeit(lambda: ops.conv_wgrad(ops.Source(xs, sc, sh), dzs, dw, taps))
        print(json.dumps(dict(shape=[B, H, W, Ci, Co, k], dtype=str(dt), conv_ms=round(t_c, 4), wgrad_ms=round(t_w, 4))), flush=True)
