#!/usr/bin/env python3
"""Host-side dispatch time of one training step vs its GPU time (is the step launch-bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from coarse3d_amd.pc_processor.models import SalsaNextProto
from coarse3d_amd.trainer import TrainStep
dev = "cuda"
torch.manual_seed(1)
m = SalsaNextProto(5, 20, 20, 0, use_prototype=True).to(dev).train()
ts = TrainStep(m, 20, lr=1e-3, num_anchor=512, feature_mean=bench.FEATURE_MEAN, feature_std=bench.FEATURE_STD)
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
batches = [bench.synth_batch(bs, 64, 2048, 20, 1000 + s, dev) for s in range(6)]
for s in range(2):
    ts.step(*batches[s], epoch=10)
torch.cuda.synchronize()
host, total = [], []
for s in range(2, 6):
    t0 = time.perf_counter()
    ts.step(*batches[s], epoch=10)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0); total.append(t2 - t0)
print(f"bs={bs}: host dispatch {1e3*sum(host)/len(host):.1f} ms/step, wall {1e3*sum(total)/len(total):.1f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); ts.step(*batches[0], epoch=10); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
