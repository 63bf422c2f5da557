#!/usr/bin/env python3
"""Gradient error of the HIP backbone against the float64 oracle, relative to the fp32 oracle's own error, at a
given shape (the criterion of tests/test_gpu_backbone.py at sizes the test suite does not afford)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import torch
import weights as W
from test_gpu_backbone import run_oracle
from coarse3d_amd import ops
from coarse3d_amd.backbone import Backbone
b, h, w, ncls, dataset, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], 77
ENGINE = sys.argv[6] if len(sys.argv) > 6 else "bf16x3"
ops.set_matrix_precision(ENGINE)
dev = "cuda"
st = W.closed_form_state(nclasses=ncls)
x, tr, ev = W.synthetic_batch(b, h, w, ncls, seed, 0.02, gh=8, gw=16)
masks = W.dropout_masks_for(None, b, seed + 1)
g = torch.Generator().manual_seed(seed)
d_prob = torch.randn(b, ncls, h, w, generator=g)
d_feat = torch.randn(b, 256, h, w, generator=g) * 0.05
t0 = time.time()
o32, st32, g32 = run_oracle(st, x, masks, dataset, d_prob, d_feat, torch.float32)
o64, _, g64 = run_oracle(st, x, masks, dataset, d_prob, d_feat, torch.float64)
print("oracles %.0f s" % (time.time() - t0), flush=True)
for six in (("0", "1") if ENGINE == "bf16x3" else ("0",)):
    ops.SIX_FWD_MIN_PIXELS = 0 if six == "1" else 1 << 60      # six plane products in every forward multi-tap conv / eight
    P = {k: v.to(dev).clone() for k, v in st.items()}
    bb = Backbone(P, ncls, dataset)
    out = bb.forward(x.to(dev), True, {k: v.to(dev) for k, v in masks.items()}, True)
    grads = bb.backward(d_prob.permute(0, 2, 3, 1).contiguous().to(dev), d_feat.permute(0, 2, 3, 1).contiguous().to(dev))
    torch.cuda.synchronize()
    e_hip, e_ora, bad = [], [], 0
    for k, ref in g64.items():
        if k == "projector.proj.0.bias":
            continue
        scale = float(ref.abs().max()) + 1e-30
        eh = float((grads[k].cpu().double() - ref).abs().max()) / scale
        eo = float((g32[k].double() - ref).abs().max()) / scale
        e_hip.append(eh); e_ora.append(eo)
        bad += eh > 3 * eo + 1e-4
    prob = out["prob"].permute(0, 3, 1, 2).cpu()
    perr = float((prob - o32["pred_2d"].detach()).abs().max() / o32["pred_2d"].detach().abs().max())
    print(f"{ENGINE} six-product forward taps={six}: median hip {np.median(e_hip):.3e} oracle-fp32 {np.median(e_ora):.3e} ratio {np.median(e_hip)/np.median(e_ora):.2f}; "
          f"max hip {max(e_hip):.2e} oracle {max(e_ora):.2e}; beyond 3x: {bad}/{len(e_hip)}; prob rel err {perr:.2e}", flush=True)
