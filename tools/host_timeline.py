#!/usr/bin/env python3
"""Where does the HOST spend a training step, and is it ahead of the GPU?  Wraps the step's phases with
perf_counter (no device synchronisation added) and asks the stream at a few points whether it has drained
(stream.query() == True means the GPU is idle, i.e. the host is the bottleneck there)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from coarse3d_amd import ops, trainer
from coarse3d_amd.pc_processor.models import SalsaNextProto
from coarse3d_amd.pc_processor.loss import lovasz_softmax

dev = torch.device("cuda", 0)
ops.set_matrix_precision("bf16x3")
torch.manual_seed(1)
model = SalsaNextProto(5, 20, 20, 0, use_prototype=True).to(dev).train()
ts = trainer.TrainStep(model, 20, lr=1e-3, n_epochs=100, temperature=0.07, num_anchor=512, loss_w_ce_2d=1.0, loss_w_lov_2d=1.0,
                       loss_w_contrast=0.1, feature_mean=bench.FEATURE_MEAN, feature_std=bench.FEATURE_STD, proto_loss=True,
                       inputs_resident=True)
batches = [bench.synth_batch(8, 64, 2048, 20, 1000 + s, dev, 1e-3) for s in range(14)]
torch.cuda.synchronize()
T = {}
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        idle = torch.cuda.current_stream().query()
        t0 = time.perf_counter()
        r = f(*a, **k)
        d = T.setdefault(key, [0.0, 0, 0])
        d[0] += time.perf_counter() - t0; d[1] += 1; d[2] += int(idle)
        return r
    setattr(obj, name, g)
wrap(trainer, "valid_indices", "valid_indices (nonzero sync)")
wrap(ts, "model", "model forward (host)")
wrap(ts.optimizer, "step", "optimizer.step (host)")
wrap(ts.optimizer, "zero_grad", "zero_grad")
wrap(ts, "contrast", "contrast loss fwd (host)")
wrap(trainer.loss_head, "loss_head", "loss head (host)")
wrap(trainer.contrast, "entropy_selection", "entropy selection (host)")
orig_bw = torch.Tensor.backward
def bw(self, *a, **k):
    idle = torch.cuda.current_stream().query()
    t0 = time.perf_counter(); r = orig_bw(self, *a, **k)
    d = T.setdefault("backward (host)", [0.0, 0, 0]); d[0] += time.perf_counter() - t0; d[1] += 1; d[2] += int(idle)
    return r
torch.Tensor.backward = bw
for s in range(4):
    ts.step(*batches[s], epoch=10)
torch.cuda.synchronize()
T.clear()
t0 = time.perf_counter()
host = 0.0
for s in range(4, 14):
    h0 = time.perf_counter()
    ts.step(*batches[s], epoch=10)
    host += time.perf_counter() - h0
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"step {el / 10 * 1e3:.2f} ms wall, host time inside step() {host / 10 * 1e3:.2f} ms")
for k, (t, n, idle) in T.items():
    print(f"  {k:34s} {t / n * 1e3:8.3f} ms/call  x{n / 10:.0f}/step   stream already drained at entry: {idle}/{n}")
