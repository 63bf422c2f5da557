#!/usr/bin/env python3
"""cProfile of the HOST side of the launch-by-launch training step (10 steps, headline shape): where the ~20 ms go."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from coarse3d_amd import ops, trainer
from coarse3d_amd.pc_processor.models import SalsaNextProto

dev = torch.device("cuda", 0)
ops.set_matrix_precision("bf16x3")
torch.manual_seed(1)
model = SalsaNextProto(5, 20, 20, 0, use_prototype=True).to(dev).train()
ts = trainer.TrainStep(model, 20, lr=1e-3, n_epochs=100, temperature=0.07, num_anchor=512, loss_w_ce_2d=1.0, loss_w_lov_2d=1.0,
                       loss_w_contrast=0.1, feature_mean=bench.FEATURE_MEAN, feature_std=bench.FEATURE_STD, proto_loss=True,
                       inputs_resident=True)
HH, WW = (int(os.environ.get("HP_H", "64")), int(os.environ.get("HP_W", "2048")))
batches = [bench.synth_batch(8, HH, WW, 20, 1000 + s, dev, 1e-3) for s in range(4)]
for s in range(4):
    ts.step(*batches[s], epoch=10)
torch.cuda.synchronize()
t0 = time.perf_counter()
host = 0.0
for s in range(10):
    h0 = time.perf_counter()
    ts.step(*batches[s % 4], epoch=10)
    host += time.perf_counter() - h0
torch.cuda.synchronize()
print(f"wall {1e3 * (time.perf_counter() - t0) / 10:.2f} ms/step, host inside step() {1e3 * host / 10:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for s in range(10):
    ts.step(*batches[s % 4], epoch=10)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 28)
