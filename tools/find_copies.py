#!/usr/bin/env python3
"""Where do the device-to-device copies / fills of one training step come from? (torch.profiler, with stacks)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from coarse3d_amd.pc_processor.models import SalsaNextProto
from coarse3d_amd.trainer import TrainStep
dev = "cuda"
torch.manual_seed(1)
m = SalsaNextProto(5, 20, 20, 0, use_prototype=True).to(dev).train()
ts = TrainStep(m, 20, lr=1e-3, num_anchor=512, feature_mean=bench.FEATURE_MEAN, feature_std=bench.FEATURE_STD)
bt = [bench.synth_batch(8, 64, 2048, 20, 1000 + s, dev) for s in range(3)]
for s in range(2):
    ts.step(*bt[s], epoch=10)
torch.cuda.synchronize()
import collections, traceback
from torch.utils._python_dispatch import TorchDispatchMode
agg = collections.Counter()
WANT = ("copy_", "fill_", "zero_", "clone", "zeros", "zeros_like", "new_zeros", "full")
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in WANT:
            st = [f for f in traceback.extract_stack() if "/coarse3d_amd/" in f.filename or f.filename.endswith("bench.py")]
            where = f"{os.path.basename(st[-1].filename)}:{st[-1].lineno}" if st else "autograd/optimizer"
            agg[(name, where)] += 1
        return func(*args, **(kwargs or {}))
with Log():
    ts.step(*bt[2], epoch=10)
    torch.cuda.synchronize()
for (n, s), c in agg.most_common(60):
    print(c, n, s)
