#!/usr/bin/env python3
"""Same-box A/B of a module-level flag on the captured training step.
usage: python tools/ab_step.py backbone.FUSE_BN_APPLY [H W classes batch [steps]]
Runs the headline step (TrainStep(graph=True), synthetic batches of bench.py) with the flag False / True, alternating
twice, and prints ms per step of each pass."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coarse3d_amd  # noqa: F401,E402
import torch  # noqa: E402
import bench  # noqa: E402
from coarse3d_amd import trainer  # noqa: E402
from coarse3d_amd.pc_processor.models import SalsaNextProto  # noqa: E402

flag = sys.argv[1]
modname, attr = flag.rsplit(".", 1)
mod = importlib.import_module("coarse3d_amd." + modname)
H, W, C, B = (int(v) for v in (sys.argv[2:6] if len(sys.argv) >= 6 else (64, 2048, 20, 8)))
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
dataset = "SemanticPOSS" if H == 40 else "SemanticKitti"
dev = torch.device("cuda", 0)
batches = [bench.synth_batch(B, H, W, C, 1000 + s, dev, 1e-3) for s in range(8)]
torch.cuda.synchronize()
for rnd in range(2):
    for val in (False, True):
        setattr(mod, attr, val)
        torch.manual_seed(1)
        model = SalsaNextProto(5, C, 20, 0, use_prototype=True, dataset=dataset).to(dev).train()
        ts = trainer.TrainStep(model, C, lr=1e-3, n_epochs=100, temperature=0.07, num_anchor=512, loss_w_contrast=0.1,
                               feature_mean=bench.FEATURE_MEAN, feature_std=bench.FEATURE_STD, proto_loss=True,
                               inputs_resident=True, graph=True)
        for s in range(6):
            res = ts.step(*batches[s % 8], epoch=10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(steps):
            res = ts.step(*batches[s % 8], epoch=10)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / steps
        print(f"{flag}={val}: {el * 1e3:.3f} ms/step ({B / el:.1f} img/s), loss {float(res['loss']):.6f}", flush=True)
        del ts, model
        torch.cuda.empty_cache()
