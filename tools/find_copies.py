#!/usr/bin/env python3
"""Where do the device-to-device copies of a training step come from?  (rocprofv3 shows ~124 `__amd_rocclr_copyBuffer` launches
per launch-by-launch step; a contiguous same-dtype Tensor.copy_ / clone() is a hipMemcpyAsync.)  One eager step under
torch.profiler with Python stacks; prints every memcpy / aten::copy_ call site with its count.
usage (GPU box): python tools/find_copies.py [graph]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coarse3d_amd  # noqa: F401
import torch
import bench
from coarse3d_amd.pc_processor.models import SalsaNextProto
from coarse3d_amd.trainer import TrainStep

dev = torch.device("cuda", 0)
B, H, W, C = 2, 64, 512, 20
batches = [bench.synth_batch(B, H, W, C, 1000 + s, dev, 1e-3) for s in range(4)]
torch.manual_seed(1)
model = SalsaNextProto(5, C, 20, 0, use_prototype=True).to(dev).train()
use_graph = len(sys.argv) > 1 and sys.argv[1] == "graph"
ts = TrainStep(model, C, lr=1e-3, n_epochs=100, temperature=0.07, num_anchor=512, loss_w_contrast=0.1,
               feature_mean=bench.FEATURE_MEAN, feature_std=bench.FEATURE_STD, proto_loss=True, inputs_resident=True,
               graph=True, graph_warmup=1000)      # the shape-static eager step: the arithmetic a captured step replays
for s in range(3):
    ts.step(*batches[s], epoch=10)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    ts.step(*batches[3], epoch=10)
    torch.cuda.synchronize()
ev = prof.events()
sites = collections.Counter()
kinds = collections.Counter()
for e in ev:
    n = e.name
    if "Memcpy" in n or "memcpy" in n or "Memset" in n or "memset" in n:
        kinds[n] += 1
    if n in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_", "aten::_to_copy"):
        st = [f for f in (e.stack or []) if "coarse3d_amd" in f or "bench.py" in f or "tools/" in f][:3]
        sites[(n, " <- ".join(s.split("/")[-1] for s in st))] += 1
print("device memcpy / memset activities:", dict(kinds))
for (n, st), c in sites.most_common(60):
    print(f"{c:4d}  {n:18s} {st}")
