#!/usr/bin/env python3
"""Multi-tap conv micro-benchmark in the bf16x3 / f32 matrix modes (run on the GPU box)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarse3d_amd import ops
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
ops.set_matrix_precision(mode)
dev = "cuda"
shapes = [(8, 64, 2048, 32, 32, 3, 1, 1), (8, 64, 2048, 64, 64, 3, 2, 2), (8, 32, 1024, 128, 128, 3, 2, 2),
          (8, 16, 512, 256, 256, 3, 1, 1), (8, 64, 2048, 64, 64, 2, 2, 1), (8, 64, 2048, 32, 80, 3, 1, 1)]
for (B, H, W, Ci, Co, k, dil, pad) in shapes:
    x = torch.randn(B, H, W, Ci, device=dev)
    w = torch.randn(Co, Ci, k, k, device=dev) * 0.05
    sc = torch.rand(Ci, device=dev) + 0.5; sh = torch.randn(Ci, device=dev) * 0.1
    taps = ops.conv_taps(k, k, dil, pad)
    wp = ops.pack_weights(w, 0)
    src = ops.Source(x, sc, sh)
    out = torch.empty(B, H, W, Co, device=dev)
    part = torch.empty(Co, 2, ops.num_mtiles(B, H, W), device=dev)
    fn = lambda: ops.conv_forward([src], wp, None, Co, taps, lrelu=True, out=out, stat_partial=part)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(json.dumps(dict(mode=mode, shape=[B, H, W, Ci, Co, k, dil], ms=round(ms, 4),
                          tflops=round(2.0 * B * H * W * Ci * Co * k * k / ms / 1e9, 1))), flush=True)
# pointwise shapes
for (B, H, W, Ci, Co) in [(8, 32, 1024, 704, 704), (8, 32, 1024, 704, 256), (1, 32768, 32, 256, 400), (8, 64, 2048, 192, 64), (8, 64, 2048, 64, 64)]:
    x = torch.randn(B, H, W, Ci, device=dev)
    w = torch.randn(Co, Ci, 1, 1, device=dev) * 0.05
    sc = torch.rand(Ci, device=dev) + 0.5; sh = torch.randn(Ci, device=dev) * 0.1
    wp = ops.pack_weights(w, 0)
    src = ops.Source(x, sc, sh)
    out = torch.empty(B, H, W, Co, device=dev)
    part = torch.empty(Co, 2, ops.num_mtiles(B, H, W), device=dev)
    fn = lambda: ops.conv_forward([src], wp, None, Co, [(0, 0)], lrelu=True, out=out, stat_partial=part)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    xs = x.reshape(-1, Ci)[::997].double() * sc.double() + sh.double()
    ref = torch.nn.functional.leaky_relu(xs @ w.reshape(Co, Ci).double().t(), 0.01)
    err = float((out.reshape(-1, Co)[::997].double() - ref).abs().max() / ref.abs().max())
    print(json.dumps(dict(mode=mode, shape=[B, H, W, Ci, Co, 1, 1], ms=round(ms, 4), tflops=round(2.0 * B * H * W * Ci * Co / ms / 1e9, 1), rel_err=err)), flush=True)
