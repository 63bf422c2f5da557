#!/usr/bin/env python3
"""Round 6 gate: Winograd F(2x2, 3x3) on the exact-split engine (csrc/conv_wino.hip) against the
fused nine-tap kernel (conv_x3f, csrc/conv_x3.hip) -- time and error vs float64, forward and input gradient.
Gate: 64 -> 64 3x3 d1 at 8 x 64 x 2048: <= 0.70x the time AND <= 4x the error.  Run on the GPU box."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
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


def ref64(x, sc, sh, w, dil, grad):
    """float64 reference of one launch: forward = LeakyReLU(conv(LeakyReLU? no: affine(x))); gradient = conv_transpose."""
    xa = (x.double() * sc.double() + sh.double()).permute(0, 3, 1, 2)
    if grad:
        y = F.conv_transpose2d(xa, w.double(), padding=dil, dilation=dil)
    else:
        y = F.conv2d(xa, w.double(), padding=dil, dilation=dil)
        y = torch.where(y > 0, y, 0.01 * y)
    return y.permute(0, 2, 3, 1).contiguous()


shapes = [(8, 64, 2048, 64, 64, 1), (8, 64, 2048, 64, 64, 2), (8, 64, 2048, 32, 64, 1), (8, 32, 1024, 128, 128, 2),
          (8, 16, 512, 256, 256, 2), (8, 32, 1024, 160, 64, 1), (8, 8, 256, 256, 256, 1), (8, 64, 2048, 32, 32, 1)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
res = []
for (B, H, W, Ci, Co, dil) in shapes:
    torch.manual_seed(0)
    r = dict(shape=[B, H, W, Ci, Co, dil])
    for grad in (False, True):
        cin, cout = (Co, Ci) if grad else (Ci, Co)         # the launch's input / output channels
        x = torch.randn(B, H, W, cin, device=dev)
        w = torch.randn(Co, Ci, 3, 3, device=dev) * (1.0 / (3.0 * Ci ** 0.5))
        sc = torch.rand(cin, device=dev) + 0.5
        sh = torch.randn(cin, device=dev) * 0.1
        taps = ops.conv_taps(3, 3, dil, dil)
        if grad:
            taps = ops.negate_taps(taps)
        wp = ops.pack_weights(w, 1 if grad else 0)
        kpad = (cin + 15) // 16 * 16
        wu = ops.pack_weights_wino(wp, kpad, cout, taps)
        src = ops.Source(x, sc, sh)
        out_d = torch.empty(B, H, W, cout, device=dev)
        out_w = torch.empty(B, H, W, cout, device=dev)
        part_d = torch.empty(cout, 2, ops.num_mtiles(B, H, W), device=dev)
        part_w = torch.empty(cout, 2, ops.wino_num_tiles(B, H, W, dil), device=dev)
        run_d = lambda: ops.conv_forward([src], wp, None, cout, taps, lrelu=not grad, out=out_d, stat_partial=part_d, grad=grad)
        run_w = lambda: ops.conv_forward([src], wu, None, cout, taps, lrelu=not grad, out=out_w, stat_partial=part_w, grad=grad)
        tag = "dgrad" if grad else "fwd"
        td, tw = [], []
        for _ in range(3):
            td.append(timeit(run_d))
            tw.append(timeit(run_w))
        r[tag + "_direct_ms"], r[tag + "_wino_ms"] = round(min(td), 4), round(min(tw), 4)
        r[tag + "_ratio"] = round(min(tw) / min(td), 3)
        # error vs float64 on a slice of the batch (the float64 conv is slow)
        nb = 1 if H * W >= 64 * 2048 else 2
        ref = ref64(x[:nb], sc, sh, w, dil, grad)
        ed = (out_d[:nb].double() - ref)
        ew = (out_w[:nb].double() - ref)
        scale = float(ref.abs().max())
        r[tag + "_err_direct_rms"] = float(ed.square().mean().sqrt()) / scale
        r[tag + "_err_wino_rms"] = float(ew.square().mean().sqrt()) / scale
        r[tag + "_err_direct_max"] = float(ed.abs().max()) / scale
        r[tag + "_err_wino_max"] = float(ew.abs().max()) / scale
        r[tag + "_err_ratio_rms"] = round(r[tag + "_err_wino_rms"] / r[tag + "_err_direct_rms"], 2)
        r[tag + "_err_ratio_max"] = round(r[tag + "_err_wino_max"] / r[tag + "_err_direct_max"], 2)
        # statistics partials: the same sums
        sd, sw = part_d.double().sum(-1), part_w.double().sum(-1)
        r[tag + "_stat_rel"] = float((sd - sw).abs().max() / sd.abs().max())
    print(json.dumps(r), flush=True)
    res.append(r)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/bench_wino.json", "w"), indent=1)
