#!/usr/bin/env python3
"""Micro-benchmark of the narrow-channel HBM-bound kernels at the headline shape (8x64x2048): first conv 5 -> 32 and its
weight gradient, class softmax forward / backward, entropy statistics.  Prints us per launch and achieved GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarse3d_amd import ops


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = "cuda"
    B, H, W, C = 8, 64, 2048, 20
    n = B * H * W
    x = torch.randn(B, 5, H, W, device=dev)
    w = torch.randn(32, 5, device=dev); bias = torch.randn(32, device=dev)
    dz = torch.randn(B, H, W, 32, device=dev); dw = torch.zeros(32, 5, device=dev)
    logits = torch.randn(B, H, W, 32, device=dev)
    prob = ops.softmax(logits, C)
    dprob = torch.randn_like(prob)
    rows = [("conv_in5", lambda: ops.conv_in5(x, w, bias), n * (20 + 128)),
            ("conv_in5_wgrad", lambda: ops.conv_in5_wgrad(x, dz, dw), n * (20 + 128)),
            ("softmax", lambda: ops.softmax(logits, C), n * (128 + 80)),
            ("softmax_bwd", lambda: ops.softmax_bwd(prob, dprob, logits.shape), n * (80 + 80 + 128)),
            ("entropy_stats", lambda: ops.entropy_stats(prob), n * (80 + 12))]
    feat = torch.randn(B, 32, 1024, 256, device=dev)
    up = torch.empty(B, H, W, 256, device=dev)
    dfeat = torch.randn(B, H, W, 256, device=dev)
    dlow = torch.zeros_like(feat)
    lvl = torch.randn(B, 8, 256, 256, device=dev)
    cat = torch.empty(B, 32, 1024, 704, device=dev)
    dcat = torch.randn(B, 32, 1024, 704, device=dev)
    dlvl = torch.zeros_like(lvl)
    rows += [("bilinear x2 256ch", lambda: ops.bilinear(feat, H, W, dst=up), n * 1024 + feat.numel() * 4),
             ("bilinear_bwd x2 256ch", lambda: ops.bilinear_bwd(dlow, dfeat), n * 1024 + feat.numel() * 4),
             ("bilinear x4 256ch->cat", lambda: ops.bilinear(lvl, 32, 1024, dst=cat, dcoff=448), B * 32 * 1024 * 1024 + lvl.numel() * 4),
             ("bilinear_bwd x4 256ch", lambda: ops.bilinear_bwd(dlvl, dcat, dcoff=448, c=256), B * 32 * 1024 * 1024 + lvl.numel() * 4)]
    xin = torch.randn(B, H, W, 64, device=dev); mk = torch.rand(B, 64, device=dev)
    dpo = torch.randn(B, H // 2, W // 2, 64, device=dev)
    rows += [("maskpool 64ch", lambda: ops.maskpool(xin, mk, True), n * 256 * 1.25),
             ("maskpool_bwd 64ch", lambda: ops.maskpool_bwd(dpo, mk, None, (B, H, W, 64), True), n * 256 * 1.25)]
    lnw, lnb = torch.rand(256, device=dev), torch.randn(256, device=dev)
    rows += [("rownorm_ln_l2 256ch", lambda: ops.rownorm_ln_l2(up.view(n, 256), lnw, lnb), n * 2048)]
    for name, fn, byts in rows:
        us = timeit(fn)
        print(f"{name:16s} {us:8.1f} us  {byts / us / 1e3:8.0f} GB/s")


if __name__ == "__main__":
    main()
