#!/usr/bin/env python3
"""Run-to-run determinism of the captured training step: 20 steps at the headline shape, per-step loss and per-parameter
checksums written to a JSON file; run it a few times and compare the files (round 4: a key tie at the threshold of the
pseudo-label selection made such runs two-valued).  usage: python tools/determinism_probe.py out.json"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coarse3d_amd, torch, bench
from coarse3d_amd import trainer, backbone
from coarse3d_amd.pc_processor.models import SalsaNextProto
dev = torch.device("cuda", 0)
H, W, C, B = 64, 2048, 20, 8
batches = [bench.synth_batch(B, H, W, C, 1000 + s, dev, 1e-3) for s in range(8)]
torch.manual_seed(1)
model = SalsaNextProto(5, C, 20, 0, use_prototype=True).to(dev).train()
ts = trainer.TrainStep(model, C, lr=1e-3, n_epochs=100, temperature=0.07, num_anchor=512, loss_w_contrast=0.1,
                       feature_mean=bench.FEATURE_MEAN, feature_std=bench.FEATURE_STD, proto_loss=True, inputs_resident=True, graph=True)
log = []
for s in range(20):
    res = ts.step(*batches[s % 8], epoch=10)
    torch.cuda.synchronize()
    rec = {"loss": float(res["loss"])}
    for n, p in model.named_parameters():
        rec["p/" + n] = float(p.detach().double().sum())
        if p.grad is not None:
            rec["g/" + n] = float(p.grad.detach().double().abs().sum())
    log.append(rec)
json.dump(log, open(sys.argv[1], "w"))
print("final", log[-1]["loss"])
