#!/usr/bin/env python3
"""Step time of the headline training step over the first seconds of a process (first GPU process on a fresh box:
is the slow first bench run a clock / power ramp or a property of the process?).  Prints ms per group of 10 steps."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from coarse3d_amd import ops
from coarse3d_amd.pc_processor.models import SalsaNextProto
from coarse3d_amd.trainer import TrainStep

ops.set_matrix_precision("bf16x3")
dev = "cuda"
torch.manual_seed(0)
m = SalsaNextProto(5, 20, 20, 0, use_prototype=True).to(dev).train()
ts = TrainStep(m, 20, lr=1e-3, n_epochs=100, temperature=0.07, num_anchor=512, proto_loss=True)
x, tr, ev = synth_batch(8, 64, 2048, 20, 1, dev)
t_start = time.perf_counter()
for grp in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ts.step(x, tr, ev, epoch=10)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"t={t1 - t_start:6.2f}s  {1e3 * (t1 - t0) / 10:7.2f} ms/step", flush=True)
