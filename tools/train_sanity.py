#!/usr/bin/env python3
"""Does the HIP path actually train?  150 steps on a learnable synthetic task (the input channels
carry a noisy code of the class map; 2 % weak labels), reporting loss and accuracy on ALL pixels
(the labels the model never saw) for the fp32 path and the opt-in bf16 matrix mode."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarse3d_amd import ops
from coarse3d_amd.pc_processor.models import SalsaNextProto
from coarse3d_amd.trainer import TrainStep

dev = "cuda"
B, H, W, C = 4, 64, 512, 20

def batch(seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    grid = torch.randint(1, C, (B, H // 8, W // 32), generator=g, device=dev)
    ev = grid.repeat_interleave(8, 1).repeat_interleave(32, 2)
    code = torch.stack([torch.sin(ev * (k + 1) * 0.7) for k in range(5)], 1)          # class code in 5 channels
    x = code + 0.5 * torch.randn(B, 5, H, W, generator=g, device=dev)
    tr = ev * (torch.rand(B, H, W, generator=g, device=dev) < 0.02)
    return x, tr.long(), ev.long()

out = {}
for mode in (sys.argv[1:] or ["f32", "bf16"]):
    ops.set_matrix_precision(mode)
    torch.manual_seed(0)
    m = SalsaNextProto(5, C, 20, 0, use_prototype=True).to(dev).train()
    ts = TrainStep(m, C, proto_loss=True, lr=2e-3, n_epochs=100, num_anchor=128)      # C3D_GRAPH=1: the captured step
    log = []
    for s in range(150):
        x, tr, ev = batch(s)
        r = ts.step(x, tr, ev, epoch=10)
        if s % 25 == 0 or s == 149:
            acc = float((r["pred_2d"].argmax(1) == ev).float().mean())
            log.append((s, round(float(r["loss"]), 3), round(float(r["ce"].detach()), 3), round(float(r["contrast"].detach()), 3), round(acc, 3)))
    out[mode] = log
    assert all(torch.isfinite(p).all() for p in m.parameters())
ops.set_matrix_precision("f32")
print(json.dumps({"columns": ["step", "loss", "focal", "contrast", "accuracy on all pixels"], **out}))
