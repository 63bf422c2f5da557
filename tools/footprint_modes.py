#!/usr/bin/env python3
"""Allocator footprint after every step, Python's cyclic collector OFF, for the ways the package is driven: TrainStep launch by
launch / captured, the plain module API (with and without the graphed backbone), eval forwards, the other two backbones, a
1-rank data-parallel wrapper.  A mode whose footprint grows holds its activations in a reference cycle.
usage (GPU box): python tools/footprint_modes.py"""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import coarse3d_amd, torch
import weights as W
from coarse3d_amd.pc_processor.models import RangeNetProto, SalsaNextProto, SqueezeSegV3Proto
from coarse3d_amd.trainer import TrainStep
DEV = "cuda"
b, h, w, ncls = 2, 32, 256, 20
x, tr, ev = (t.to(DEV) for t in W.synthetic_batch(b, h, w, ncls, 5, 0.02, gh=8, gw=16))


def run(name, make):
    torch.manual_seed(3)
    step = make()
    for _ in range(4):
        step()
    gc.collect()
    torch.cuda.synchronize()
    gc.disable()
    try:
        seen = []
        for _ in range(12):
            step()
            torch.cuda.synchronize()
            seen.append(torch.cuda.memory_allocated())
    finally:
        gc.enable()
    grow = (max(seen) - min(seen)) / 2**20
    print(f"{name:46s} {min(seen) / 2**20:9.1f} MiB .. {max(seen) / 2**20:9.1f} MiB  {'FLAT' if grow <= 1.0 else 'GROWS by %.1f MiB' % grow}", flush=True)
    gc.collect()
    torch.cuda.empty_cache()
    return grow <= 1.0


def trainstep(cls, graph, **kw):
    def make():
        m = cls(**kw).to(DEV).train()
        ts = TrainStep(m, ncls, proto_loss=True, lr=1e-3, num_anchor=64, loss_w_contrast=0.5, n_epochs=20, graph=graph, inputs_resident=True)
        return lambda: ts.step(x, tr, ev, epoch=10)
    return make


def module_api(graph_backbone, train=True):
    def make():
        m = SalsaNextProto(5, ncls, 20, 0, use_prototype=True).to(DEV)
        m.train(train)
        m.graph_backbone = graph_backbone
        opt = torch.optim.AdamW(m.parameters(), lr=1e-3)

        def step():
            if not train:
                with torch.no_grad():
                    return m(x)["pred_2d"].sum()
            opt.zero_grad(set_to_none=True)
            out = m(x, label=tr, eval_mask=tr > 0, return_feat=True, proto_loss=True)
            (out["pred_2d"].mean() + out["feat_2d"].mean()).backward()
            opt.step()
        return step
    return make


def forward_only_training_mode():
    m = SalsaNextProto(5, ncls, 20, 0, use_prototype=True).to(DEV).train()
    return lambda: m(x, label=tr, eval_mask=tr > 0, return_feat=True, proto_loss=True)["pred_2d"].sum().item()      # graph dropped, no backward


ok = True
ok &= run("TrainStep, launch by launch (SalsaNext)", trainstep(SalsaNextProto, False, in_channel=5, nclasses=ncls, use_prototype=True))
ok &= run("TrainStep, captured (SalsaNext)", trainstep(SalsaNextProto, True, in_channel=5, nclasses=ncls, use_prototype=True))
ok &= run("module API, launch by launch", module_api(False))
ok &= run("module API, graphed backbone", module_api(True))
ok &= run("module API, eval forward", module_api(False, train=False))
ok &= run("training-mode forward without a backward", forward_only_training_mode)
ok &= run("TrainStep, launch by launch (RangeNet-21)", trainstep(RangeNetProto, False, layers=21, nclasses=ncls, use_prototype=True))
ok &= run("TrainStep, launch by launch (SqueezeSegV3-21)", trainstep(SqueezeSegV3Proto, False, nclasses=ncls, layers=21, use_prototype=True))
print("ok" if ok else "SOME MODE GROWS")
sys.exit(0 if ok else 1)
