#!/usr/bin/env python3
"""Weight-gradient micro-benchmark with an fp64 check (run on the GPU box): tools/bench_wgrad.py [f32|bf16|bf16x3] [fuse]
(fuse: the launches apply a BatchNorm / LeakyReLU backward on load, conv_wgrad(fuse=...): timing only for those)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from coarse3d_amd import ops
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
ops.set_matrix_precision(mode, storage="f32") if mode == "bf16" else ops.set_matrix_precision(mode)
dev = "cuda"
# B, H, W, Cin, Cout, k, dil, pad
shapes = [(8, 32, 1024, 704, 704, 1, 1, 0), (8, 32, 1024, 704, 256, 1, 1, 0), (8, 64, 2048, 64, 64, 3, 2, 2),
          (8, 64, 2048, 64, 64, 3, 1, 1), (8, 64, 2048, 64, 64, 2, 2, 1), (8, 64, 2048, 32, 32, 3, 1, 1),
          (8, 64, 2048, 32, 32, 3, 2, 2), (8, 64, 2048, 64, 64, 1, 1, 0), (8, 32, 1024, 128, 128, 1, 1, 0),
          (8, 64, 2048, 32, 32, 1, 1, 0), (8, 32, 1024, 128, 128, 3, 2, 2), (8, 16, 512, 256, 256, 3, 1, 1),
          (8, 32, 1024, 128, 128, 2, 2, 1), (8, 16, 512, 256, 256, 2, 2, 1), (8, 8, 256, 256, 256, 2, 2, 1)]
for (B, H, W, Ci, Co, k, dil, pad) in shapes:
    torch.manual_seed(Ci * 7 + Co + k)
    x = torch.randn(B, H, W, Ci, device=dev)
    dz = torch.randn(B, H, W, Co, device=dev)
    sc = torch.rand(Ci, device=dev) + 0.5; sh = torch.randn(Ci, device=dev) * 0.1
    taps = ops.conv_taps(k, k, dil, pad)
    src = ops.Source(x, sc, sh, lrelu=True)
    dw = torch.zeros(Co, Ci, k, k, device=dev)
    fuse = len(sys.argv) > 2 and sys.argv[2] == "fuse"
    if fuse:
        act = torch.randn(B, H, W, Co, device=dev)
        kk = torch.randn(3, Co, device=dev) * 0.1
        dzo = torch.empty_like(dz)
        db = torch.zeros(Co, device=dev)
        fn = lambda: ops.conv_wgrad(src, dzo, dw, taps, dbias=db, fuse=(dz, act, kk))
    else:
        fn = lambda: ops.conv_wgrad(src, dz, dw, taps)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    if fuse:
        print(json.dumps(dict(mode=mode + "+fuse", shape=[B, H, W, Ci, Co, k, dil], ms=round(ms, 4),
                              tflops=round(2.0 * B * H * W * Ci * Co * k * k / ms / 1e9, 1), rel_err=0.0)), flush=True)
        continue
    # fp64 check on one image, a channel subset
    ci_n, co_n = min(Ci, 48), min(Co, 40)
    xt = F.leaky_relu(x[:1].double() * sc.double() + sh.double(), 0.01)[..., :ci_n].permute(0, 3, 1, 2)
    src1 = ops.Source(x[:1].contiguous(), sc, sh, lrelu=True)
    dw1 = torch.zeros(Co, Ci, k, k, device=dev)
    ops.conv_wgrad(src1, dz[:1].contiguous(), dw1, taps)
    wref = torch.zeros(co_n, ci_n, k, k, device=dev, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(xt, wref, padding=pad, dilation=dil)
    (g,) = torch.autograd.grad(y, wref, dz[:1, ..., :co_n].double().permute(0, 3, 1, 2)[:, :, :y.shape[2], :y.shape[3]])
    err = float((dw1[:co_n, :ci_n].double() - g).abs().max() / g.abs().max())
    print(json.dumps(dict(mode=mode, shape=[B, H, W, Ci, Co, k, dil], ms=round(ms, 4),
                          tflops=round(2.0 * B * H * W * Ci * Co * k * k / ms / 1e9, 1), rel_err=err)), flush=True)
