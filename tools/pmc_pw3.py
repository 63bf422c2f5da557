#!/usr/bin/env python3
"""The 704 -> 704 projector GEMM on the three pointwise kernels (for a rocprofv3 --pmc run, see tools/pmc_summary.py):
ops.CONV_VARIANT (c3d_conv_desc.variant): 3 (round 2's phased kernel), 1 (fused, eight waves), 2 (fused, four waves x two workgroups per CU)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarse3d_amd import ops

ops.set_matrix_precision("bf16x3")
dev = "cuda"
for (B, H, W, Ci, Co) in [(8, 32, 1024, 704, 704), (8, 64, 2048, 256, 400), (8, 32, 1024, 128, 384)]:
    x = torch.randn(B, H, W, Ci, device=dev)
    w = torch.randn(Co, Ci, 1, 1, device=dev) * 0.05
    wp = ops.pack_weights(w, 0)
    src = ops.Source(x)
    out = torch.empty(B, H, W, Co, device=dev)
    for fused in ("0", "1", "2"):
        ops.CONV_VARIANT = {"0": 3, "1": 1, "2": 2}.get(str(fused), 0)
        for _ in range(3):
            ops.conv_forward([src], wp, None, Co, [(0, 0)], lrelu=True, out=out, stats=True)
torch.cuda.synchronize()
