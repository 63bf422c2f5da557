#!/usr/bin/env python3
"""A handful of representative MFMA launches (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarse3d_amd import ops
dev = "cuda"
if len(sys.argv) > 1:
    ops.set_matrix_precision(sys.argv[1])
shapes = [(8, 64, 2048, 32, 32, 3, 1, 1), (8, 64, 2048, 64, 64, 3, 2, 2), (8, 16, 512, 256, 256, 3, 1, 1),
          (8, 32, 1024, 704, 704, 1, 1, 0), (8, 64, 2048, 192, 64, 1, 1, 0)]
for (B, H, W, Ci, Co, k, dil, pad) in shapes:
    x = torch.randn(B, H, W, Ci, device=dev)
    w = torch.randn(Co, Ci, k, k, device=dev) * 0.05
    sc = torch.rand(Ci, device=dev) + 0.5; sh = torch.randn(Ci, device=dev) * 0.1
    taps = ops.conv_taps(k, k, dil, pad)
    wp = ops.pack_weights(w, 0)
    src = ops.Source(x, sc, sh)
    dz = torch.randn(B, H, W, Co, device=dev)
    dw = torch.zeros_like(w)
    for _ in range(3):
        ops.conv_forward([src], wp, None, Co, taps, lrelu=True, stats=True)
        ops.conv_wgrad(src, dz, dw, taps)
torch.cuda.synchronize()
