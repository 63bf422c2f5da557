#!/usr/bin/env python3
"""Four-tap (2x2, dilation 2) convs: the fused schedule (c3d_conv_desc.variant & 16, round 5) against the phased kernel --
bit identity and time (variant 4 = phased, 0 = the library's choice: fused).  usage (GPU box): python tools/bench_x3f4.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarse3d_amd import ops
ops.set_matrix_precision("bf16x3")
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [(8, 64, 2048, 64, 64), (8, 64, 2048, 32, 32), (8, 32, 1024, 128, 128), (8, 16, 512, 256, 256), (8, 8, 256, 256, 256),
          (2, 40, 232, 64, 64), (1, 16, 72, 128, 96)]
taps = ops.conv_taps(2, 2, 2, 1)
for (B, H, W, Ci, Co) in shapes:
    g = torch.Generator(device=dev).manual_seed(Ci + Co)
    x = torch.randn(B, H, W, Ci, device=dev, generator=g)
    w = torch.randn(Co, Ci, 2, 2, device=dev, generator=g) * 0.05
    sc, sh = torch.rand(Ci, device=dev, generator=g) + 0.5, torch.randn(Ci, device=dev, generator=g) * 0.1
    bias = torch.randn(Co, device=dev, generator=g)
    for grad in (False, True):
        wp = ops.pack_weights(w, 1 if grad else 0, c_off=0, c_cnt=Ci if grad else None, kpad=(Co + 15) // 16 * 16 if grad else None) if grad else ops.pack_weights(w, 0)
        if grad:
            srcs, cout, tp, b_ = [ops.Source(torch.randn(B, H, W, Co, device=dev, generator=g))], Ci, ops.negate_taps(taps), None
        else:
            srcs, cout, tp, b_ = [ops.Source(x, sc, sh)], Co, taps, bias
        res = {}
        for var in (4, 0):
            ops.CONV_VARIANT = var
            out = torch.empty(B, H, W, cout, device=dev)
            part = torch.empty(cout, 2, ops.num_mtiles(B, H, W), device=dev)
            fn = lambda: ops.conv_forward(srcs, wp, b_, cout, tp, lrelu=not grad, out=out, stat_partial=None if grad else part, grad=grad)
            ms = timeit(fn)
            torch.cuda.synchronize()
            res[var] = (ms, out.clone(), part.clone())
        ops.CONV_VARIANT = 0
        same = torch.equal(res[4][1], res[0][1]) and (grad or torch.equal(res[4][2], res[0][2]))
        fl = 2.0 * B * H * W * Ci * Co * 4
        print(json.dumps(dict(shape=[B, H, W, Ci, Co], grad=grad, phased_ms=round(res[4][0], 4), fused_ms=round(res[0][0], 4),
                              phased_tf=round(fl / res[4][0] / 1e9, 1), fused_tf=round(fl / res[0][0] / 1e9, 1), bit_identical=same)), flush=True)
