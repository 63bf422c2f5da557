#!/usr/bin/env python3
"""Same-process A/B of two c3d_wgrad_desc.variant values per layer shape: tools/ab_wgrad_variant.py <variant A> <variant B> [fuse]
(e.g. 0 128: eight against four producer waves in the 1x1 instances).  Alternates the two, three rounds of ten launches."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarse3d_amd import ops
ops.set_matrix_precision("bf16x3")
dev = "cuda"
va, vb = int(sys.argv[1]), int(sys.argv[2])
fuse = len(sys.argv) > 3 and sys.argv[3] == "fuse"
shapes = [(8, 64, 2048, 64, 64, 1, 1, 0), (8, 64, 2048, 32, 32, 1, 1, 0), (8, 64, 2048, 32, 64, 1, 1, 0), (8, 64, 2048, 64, 32, 1, 1, 0),
          (8, 32, 1024, 128, 128, 1, 1, 0), (8, 32, 1024, 64, 128, 1, 1, 0), (8, 16, 512, 256, 128, 1, 1, 0), (8, 16, 512, 128, 256, 1, 1, 0),
          (8, 32, 1024, 704, 256, 1, 1, 0), (8, 8, 256, 256, 256, 1, 1, 0), (8, 4, 128, 256, 256, 1, 1, 0), (8, 64, 2048, 64, 64, 3, 1, 1)]
if len(sys.argv) > 4 and sys.argv[4] == "nine":      # the nine-tap layers of the step (256 against 0: four + four waves against the role split)
    shapes = [(8, 64, 2048, 64, 64, 3, 2, 2), (8, 64, 2048, 64, 64, 3, 1, 1), (8, 64, 2048, 32, 64, 3, 1, 1), (8, 64, 2048, 32, 32, 3, 1, 1),
              (8, 64, 2048, 32, 32, 3, 2, 2), (8, 64, 2048, 64, 32, 3, 1, 1), (8, 64, 2048, 16, 32, 3, 1, 1), (8, 32, 1024, 128, 128, 3, 2, 2),
              (8, 32, 1024, 160, 64, 3, 1, 1), (8, 32, 1024, 64, 128, 3, 1, 1), (8, 16, 512, 256, 256, 3, 2, 2), (8, 16, 512, 288, 128, 3, 1, 1),
              (8, 8, 256, 256, 256, 3, 2, 2), (8, 4, 128, 256, 256, 3, 1, 1), (2, 40, 1800, 32, 32, 3, 2, 2), (2, 5, 70, 64, 64, 3, 1, 1)]
if len(sys.argv) > 4 and sys.argv[4] == "four":      # the 2x2 layers of the step (256 against 0, as "nine")
    shapes = [(8, 64, 2048, 64, 64, 2, 2, 1), (8, 32, 1024, 128, 128, 2, 2, 1), (8, 16, 512, 256, 256, 2, 2, 1), (8, 8, 256, 256, 256, 2, 2, 1),
              (8, 4, 128, 256, 256, 2, 2, 1), (2, 40, 1800, 64, 64, 2, 2, 1), (2, 5, 70, 64, 64, 2, 2, 1), (8, 64, 2048, 32, 32, 2, 2, 1)]
for (B, H, W, Ci, Co, k, dil, pad) in shapes:
    torch.manual_seed(Ci * 7 + Co + k)
    x = torch.randn(B, H, W, Ci, device=dev); dz = torch.randn(B, H, W, Co, device=dev)
    sc = torch.rand(Ci, device=dev) + 0.5; sh = torch.randn(Ci, device=dev) * 0.1
    taps = ops.conv_taps(k, k, dil, pad)
    src = ops.Source(x, sc, sh, lrelu=True)
    act = torch.randn(B, H, W, Co, device=dev); kk = torch.randn(3, Co, device=dev) * 0.1
    res, outs = {va: [], vb: []}, {}
    for rnd in range(3):
        for v in (va, vb):
            ops.WGRAD_VARIANT = v
            dw = torch.zeros(Co, Ci, k, k, device=dev); dzo = torch.empty_like(dz); db = torch.zeros(Co, device=dev)
            fn = (lambda: ops.conv_wgrad(src, dzo, dw, taps, dbias=db, fuse=(dz, act, kk))) if fuse else (lambda: ops.conv_wgrad(src, dz, dw, taps))
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 10)
            outs[v] = (dw.clone(), dzo.clone() if fuse else None, db.clone() if fuse else None)
    ops.WGRAD_VARIANT = 0
    same = torch.equal(outs[va][0], outs[vb][0]) and (not fuse or (torch.equal(outs[va][1], outs[vb][1]) and torch.equal(outs[va][2], outs[vb][2])))
    print(json.dumps(dict(shape=[B, H, W, Ci, Co, k], fuse=fuse, a_ms=round(min(res[va]), 4), b_ms=round(min(res[vb]), 4),
                          ratio=round(min(res[va]) / min(res[vb]), 3), bit_identical=bool(same))), flush=True)
