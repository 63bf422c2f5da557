#!/usr/bin/env python3
"""Soak of the captured headline step: N replays (default 3000) over eight synthetic batches, loss finite throughout, allocator
footprint flat (no growth after the capture), parameters finite at the end.  usage: python tools/step_soak.py [N] [eager|graph] [seed]   (eager: the same steps launch by launch -- the losses must be the same numbers)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coarse3d_amd, torch, bench
from coarse3d_amd import trainer
from coarse3d_amd.pc_processor.models import SalsaNextProto
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
graph = not (len(sys.argv) > 2 and sys.argv[2] == "eager")
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda", 0)
batches = [bench.synth_batch(8, 64, 2048, 20, 1000 + s, dev, 1e-3) for s in range(8)]
torch.manual_seed(seed)
model = SalsaNextProto(5, 20, 20, 0, use_prototype=True, dataset="SemanticKitti").to(dev).train()
ts = trainer.TrainStep(model, 20, lr=1e-3, n_epochs=100, temperature=0.07, num_anchor=512, loss_w_contrast=0.1,
                       feature_mean=bench.FEATURE_MEAN, feature_std=bench.FEATURE_STD, proto_loss=True, inputs_resident=True, graph=graph)
for s in range(6):
    res = ts.step(*batches[s % 8], epoch=10)
torch.cuda.synchronize()
mem0, res0 = torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
losses, trace = [], []
t0 = time.perf_counter()
for s in range(n):
    res = ts.step(*batches[s % 8], epoch=10)
    if s % 250 == 249:
        losses.append(round(float(res["loss"]), 4))
    if s % 10 == 9:
        trace.append(res["loss"].detach().clone())
torch.cuda.synchronize()
el = time.perf_counter() - t0
ts.flush()
mem1, res1 = torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
ok = all(l == l and abs(l) < 1e4 for l in losses) and all(bool(torch.isfinite(p).all()) for p in model.parameters())
print(f"{'captured' if graph else 'launch by launch'}: {n} steps in {el:.1f} s ({el / n * 1e3:.3f} ms each), replays counted {ts._replays}, loss every 250: {losses}")
tr = torch.stack(trace).float().cpu()
spikes = [(10 * int(i) + 9, round(float(tr[i]), 3)) for i in torch.nonzero(tr > 1.5 * tr.median()).flatten()[:20]]
print(f"loss every 10 steps: median {float(tr.median()):.4f}, max {float(tr.max()):.4f}; samples above 1.5 x median: {spikes}")
print(f"allocated {mem0 / 2**20:.1f} -> {mem1 / 2**20:.1f} MiB, reserved {res0 / 2**20:.1f} -> {res1 / 2**20:.1f} MiB, finite: {ok}")
assert ok and mem1 <= mem0 + (8 << 20) and res1 <= res0 + (64 << 20)
print("ok")
