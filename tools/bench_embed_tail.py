#!/usr/bin/env python3
"""The embedding tail of the training step at the headline shape, kernel by kernel: rows of the upsampled embedding on
demand (c3d_bilinear_rows), the compact gradient scatter (c3d_scatter_rows_compact) and its adjoint
(c3d_bilinear_bwd_rows), next to the dense kernels they replace.  Run on the GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarse3d_amd import ops

dev = "cuda"


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


b, hs, ws, d, H, W = 8, 32, 1024, 256, 64, 2048
n = H * W
g = torch.Generator().manual_seed(0)
low = torch.randn(b, hs, ws, d, generator=g).to(dev)
for distinct in (70, 800):
    tmax, A, tn = 160, 512, 152
    img = (torch.arange(tmax) // 20).to(torch.int32).to(dev)
    # pixel sets of different pairs disjoint: pair p draws from pixels congruent to p mod tmax
    pools = [torch.randperm(n // tmax, generator=g)[:min(distinct, n // tmax)] * tmax + p for p in range(tmax)]
    aidx = torch.stack([pl[torch.randint(0, distinct, (A,), generator=g)] for pl in pools]).to(torch.int32).to(dev)
    t = torch.tensor([tn], device=dev, dtype=torch.int32)
    dx = torch.randn(tmax * A, d, generator=g).to(dev)
    gs = torch.tensor([0.5], device=dev)
    print(f"-- {distinct} distinct labelled pixels per (image, class) pair, {tn * A} anchor rows")
    print("bilinear_rows (l2)      %7.1f us" % timeit(lambda: ops.bilinear_rows(low, H, W, aidx, img=img, a=A, count=t, l2=True)))
    drows, cmap, rm = ops.scatter_rows_compact(dx, img, aidx, t, tmax, A, n, b, gs)
    print("scatter_rows_compact    %7.1f us" % timeit(lambda: ops.scatter_rows_compact(dx, img, aidx, t, tmax, A, n, b, gs)))
    dl = torch.empty_like(low)
    print("bilinear_bwd_rows       %7.1f us" % timeit(lambda: ops.bilinear_bwd_rows(dl, drows, cmap, rm, H, W)))
    if distinct == 70:
        dense = ops.bilinear(low, H, W)
        print("dense: bilinear         %7.1f us" % timeit(lambda: ops.bilinear(low, H, W)))
        print("dense: gather_rows_l2   %7.1f us" % timeit(lambda: ops.gather_rows_l2(dense, img, aidx, t, tmax, A, n)))
        dfeat = torch.zeros(b, H, W, d, device=dev)
        rm2 = torch.zeros_like(rm)
        ops.scatter_add_rows(dx, img, aidx, t, tmax, A, n, dfeat, gs, rowmask=rm2)
        print("dense: zero fill        %7.1f us" % timeit(lambda: dfeat.zero_()))
        print("dense: bilinear_bwd+mask%7.1f us" % timeit(lambda: ops.bilinear_bwd(dl, dfeat, rowmask=rm2)))
        del dense, dfeat
idx = torch.randint(0, b * n, (8192,), generator=g).to(dev)
cnt = torch.tensor([8000], device=dev, dtype=torch.int32)
print("bilinear_rows (8192 labelled rows) %7.1f us" % timeit(lambda: ops.bilinear_rows(low, H, W, idx, count=cnt)))
