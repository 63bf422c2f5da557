#!/usr/bin/env python3
"""Round 6: the narrow 1x1 convs of the exact-split engine -- streaming kernel (csrc/conv_pws.hip) against the staged tile of
conv_bfp (c3d_conv_desc.variant & 32) at the step's shapes; ms, TB/s of the algorithmic bytes, bit-identity.  GPU box."""
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


# (B, H, W, source widths, Cout, affine, stats, accumulate, stat_mul)
shapes = [(8, 64, 2048, [32], 32, False, False, False, False), (8, 64, 2048, [32], 32, True, True, False, False),
          (8, 64, 2048, [32], 64, False, False, False, False), (8, 64, 2048, [64], 64, False, False, True, False),
          (8, 64, 2048, [64], 64, False, True, False, True), (8, 64, 2048, [64], 32, False, False, False, False),
          (8, 64, 2048, [32, 32, 32], 32, True, True, False, False), (8, 32, 1024, [64], 64, False, True, True, True),
          (8, 64, 2048, [32], 20, True, False, False, False), (16, 32, 1024, [32], 32, True, True, False, False),
          (8, 40, 1808, [64], 64, False, False, True, False)]
res = []
for (B, H, W, srcC, Co, aff, stats, acc, smul) in shapes:
    torch.manual_seed(0)
    K = sum(srcC)
    srcs = []
    for c in srcC:
        x = torch.randn(B, H, W, c, device=dev)
        sc = (torch.rand(c, device=dev) + 0.5) if aff else None
        sh = (torch.randn(c, device=dev) * 0.1) if aff else None
        srcs.append(ops.Source(x, sc, sh, lrelu=False))
    w = torch.randn(Co, K, 1, 1, device=dev) / K ** 0.5
    wp = ops.pack_weights(w, 0)
    base = torch.randn(B, H, W, max(Co, 32), device=dev)
    mul = torch.randn(B, H, W, Co, device=dev) if smul else None
    part = torch.empty(Co, 2, ops.num_mtiles(B, H, W), device=dev) if stats else None
    outs = {}
    r = dict(shape=[B, H, W, srcC, Co], affine=aff, stats=stats, accumulate=acc, stat_mul=smul)
    byt = B * H * W * 4.0 * (K + Co * (2 if acc else 1) + (Co if smul else 0))
    for name, var in (("staged", 32), ("stream", 0), ("staged", 32), ("stream", 0)):
        ops.CONV_VARIANT = var
        out = base.clone()
        run = lambda: ops.conv_forward(srcs, wp, None, Co, [(0, 0)], lrelu=not acc, out=out, stat_partial=part, accumulate=acc,
                                       grad=True, stat_mul=mul)
        ms = timeit(run)
        r.setdefault(name + "_ms", []).append(round(ms, 4))
        out = base.clone()
        run()
        torch.cuda.synchronize()
        outs[name] = (out.clone(), part.clone() if part is not None else None)
    ops.CONV_VARIANT = 0
    for name in ("staged", "stream"):
        r[name + "_TBps"] = round(byt / min(r[name + "_ms"]) / 1e9, 2)
    r["ratio"] = round(min(r["stream_ms"]) / min(r["staged_ms"]), 3)
    r["out_bit_identical"] = bool(torch.equal(outs["staged"][0], outs["stream"][0]))
    r["out_max_rel"] = float((outs["staged"][0] - outs["stream"][0]).abs().max() / outs["staged"][0].abs().max())
    if part is not None:
        sa, sb = outs["staged"][1].double().sum(-1), outs["stream"][1].double().sum(-1)
        r["stat_rel"] = float((sa - sb).abs().max() / sa.abs().max())
    print(json.dumps(r), flush=True)
    res.append(r)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/bench_pws.json", "w"), indent=1)
