#!/usr/bin/env python3
"""Weight-gradient launches for rocprofv3 --pmc runs: tools/pmc_wgrad.py [mode] [shape index ...]."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarse3d_amd import ops
dev = "cuda"
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
ops.set_matrix_precision(mode, storage="f32") if mode == "bf16" else ops.set_matrix_precision(mode)
shapes = [(8, 32, 1024, 704, 704, 1, 1, 0), (8, 64, 2048, 64, 64, 3, 2, 2), (8, 64, 2048, 64, 64, 2, 2, 1),
          (8, 64, 2048, 32, 32, 3, 1, 1), (8, 64, 2048, 64, 64, 1, 1, 0)]
sel = [int(v) for v in sys.argv[2:]] or range(len(shapes))
for i in sel:
    B, H, W, Ci, Co, k, dil, pad = shapes[i]
    x = torch.randn(B, H, W, Ci, device=dev)
    dz = torch.randn(B, H, W, Co, device=dev)
    sc = torch.rand(Ci, device=dev) + 0.5; sh = torch.randn(Ci, device=dev) * 0.1
    dw = torch.zeros(Co, Ci, k, k, device=dev)
    for _ in range(3):
        ops.conv_wgrad(ops.Source(x, sc, sh, lrelu=True), dz, dw, ops.conv_taps(k, k, dil, pad))
torch.cuda.synchronize()
