#!/usr/bin/env python3
"""One timed run of the captured headline step (6 warm-up + 20 timed steps): tools/ab_step.py [H W classes batch].
The A/B scripts (tools/ab_lib.sh, tools/ab_env.sh) alternate it under two settings on one box."""

import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coarse3d_amd, torch, bench
from coarse3d_amd import trainer
from coarse3d_amd.pc_processor.models import SalsaNextProto
H, W, C, B = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (64, 2048, 20, 8)))
dev = torch.device("cuda", 0)
batches = [bench.synth_batch(B, H, W, C, 1000 + s, dev, 1e-3) for s in range(8)]
torch.manual_seed(1)
model = SalsaNextProto(5, C, 20, 0, use_prototype=True, dataset="SemanticPOSS" if H == 40 else "SemanticKitti").to(dev).train()
ts = trainer.TrainStep(model, C, lr=1e-3, n_epochs=100, temperature=0.07, num_anchor=512, loss_w_contrast=0.1,
                       feature_mean=bench.FEATURE_MEAN, feature_std=bench.FEATURE_STD, proto_loss=True, inputs_resident=True, graph=True)
for s in range(6):
    res = ts.step(*batches[s % 8], epoch=10)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(20):
    res = ts.step(*batches[s % 8], epoch=10)
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / 20
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("C3D_") and v) or "default build"
tag += f" [late_steps {getattr(ts, 'late_steps', '-')}]"
print(f"{tag}: {el * 1e3:.3f} ms/step ({B / el:.1f} img/s), loss {float(res['loss']):.6f}", flush=True)
