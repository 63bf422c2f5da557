#!/usr/bin/env python3
"""Producer / consumer ablation of wgrad_tr (c3d_wgrad_desc.variant & 8: consumer waves idle, & 16: producer waves idle).
Needs a library built with the switches compiled in (they cost the product kernels their counted waits):
    make -C coarse3d_amd/csrc clean && make -C coarse3d_amd/csrc -j8 EXTRA=-DC3D_WGRAD_ABLATE
usage (GPU box): C3D_LIB=<that build> python tools/ablate_wgrad.py [fuse | fuse9]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarse3d_amd import ops
ops.set_matrix_precision("bf16x3")
dev = "cuda"
fuse = len(sys.argv) > 1 and sys.argv[1] == "fuse"
shapes = [(8, 64, 2048, 32, 32, 3, 2, 2), (8, 64, 2048, 32, 32, 3, 1, 1), (8, 64, 2048, 64, 64, 3, 2, 2), (8, 64, 2048, 64, 64, 2, 2, 1),
          (8, 32, 1024, 128, 128, 3, 2, 2), (8, 32, 1024, 704, 256, 1, 1, 0), (8, 64, 2048, 64, 64, 1, 1, 0),
          (8, 64, 2048, 32, 32, 1, 1, 0), (8, 32, 1024, 128, 128, 1, 1, 0)]
LEGS = None
if len(sys.argv) > 1 and sys.argv[1] == "fuse9":
    # round 6: the fused nine-tap launches in both forms -- four + four waves (variant & 256) against eight + eight with the producer
    # waves split by tensor -- each whole, producers alone (& 8) and consumers alone (& 16): which side the new form's gain is on
    fuse = True
    shapes = [(8, 64, 2048, 64, 64, 3, 2, 2), (8, 64, 2048, 64, 64, 3, 1, 1), (8, 64, 2048, 64, 32, 3, 1, 1), (8, 64, 2048, 32, 32, 3, 2, 2),
              (8, 64, 2048, 32, 32, 3, 1, 1), (8, 32, 1024, 128, 128, 3, 2, 2)]
    LEGS = (("old_full", 256), ("old_producers_alone", 256 | 8), ("old_consumers_alone", 256 | 16),
            ("new_full", 0), ("new_producers_alone", 8), ("new_consumers_alone", 16), ("barriers_only", 8 | 16),
            ("new_producers_l2_loads", 8 | 32), ("new_producers_no_lds_stores", 8 | 64), ("new_producers_l2_no_lds", 8 | 32 | 64))
for (B, H, W, Ci, Co, k, dil, pad) in shapes:
    x = torch.randn(B, H, W, Ci, device=dev); dz = torch.randn(B, H, W, Co, device=dev)
    sc = torch.rand(Ci, device=dev) + 0.5; sh = torch.randn(Ci, device=dev) * 0.1
    taps = ops.conv_taps(k, k, dil, pad)
    src = ops.Source(x, sc, sh, lrelu=True)
    dw = torch.zeros(Co, Ci, k, k, device=dev)
    act = torch.randn(B, H, W, Co, device=dev); kk = torch.randn(3, Co, device=dev) * 0.1
    dzo = torch.empty_like(dz); db = torch.zeros(Co, device=dev)
    out = {}
    for name, var in LEGS or (("full", 0), ("full_4_producer_waves", 128), ("consumer_only_4pw", 16 | 128), ("producer_only_4pw", 8 | 128), ("producer_only", 8), ("consumer_only", 16), ("barriers_only", 24),
                      ("prod_only_l2_loads", 8 | 32), ("prod_only_no_lds_stores", 8 | 64), ("prod_only_l2_no_lds", 8 | 32 | 64)):
        ops.WGRAD_VARIANT = var
        fn = (lambda: ops.conv_wgrad(src, dzo, dw, taps, dbias=db, fuse=(dz, act, kk))) if fuse else (lambda: ops.conv_wgrad(src, dz, dw, taps))
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        out[name] = round(e0.elapsed_time(e1) / 10, 4)
    ops.WGRAD_VARIANT = 0
    print(json.dumps(dict(shape=[B, H, W, Ci, Co, k, dil], fuse=fuse, **out)), flush=True)
