import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import weights as W
from oracle import coarse3d_oracle as oc
from test_gpu_backbone import run_oracle
from coarse3d_amd.backbone import Backbone
b,h,w,ncls,dataset,seed = 2,64,128,20,"SemanticKitti",77
if len(sys.argv) > 1: h, w = int(sys.argv[1]), int(sys.argv[2])
dev="cuda"
st = W.closed_form_state(nclasses=ncls)
x, tr, ev = W.synthetic_batch(b, h, w, ncls, seed, 0.02, gh=8, gw=16)
masks = W.dropout_masks_for(None, b, seed + 1)
g = torch.Generator().manual_seed(seed)
d_prob = torch.randn(b, ncls, h, w, generator=g)
d_feat = torch.randn(b, 256, h, w, generator=g) * 0.05
if len(sys.argv) > 3 and sys.argv[3] == "nofeat": d_feat = d_feat * 0
P = {k: v.to(dev).clone() for k, v in st.items()}
bb = Backbone(P, ncls, dataset)
out = bb.forward(x.to(dev), True, {k: v.to(dev) for k, v in masks.items()}, True)
grads = bb.backward(d_prob.permute(0, 2, 3, 1).contiguous().to(dev), d_feat.permute(0, 2, 3, 1).contiguous().to(dev))
torch.cuda.synchronize()
o32, st32, g32 = run_oracle(st, x, masks, dataset, d_prob, d_feat, torch.float32)
o64, _, g64 = run_oracle(st, x, masks, dataset, d_prob, d_feat, torch.float64)
for k, ref in g64.items():
    scale = float(ref.abs().max()) + 1e-30
    eh = float((grads[k].cpu().double() - ref).abs().max()) / scale
    eo = float((g32[k].double() - ref).abs().max()) / scale
    flag = "BAD" if eh > 3 * eo + 1e-4 else ""
    print(f"{k:34s} hip {eh:.2e} ora {eo:.2e} {flag}")
