#!/usr/bin/env python3
"""What a kernel node costs in a replayed hipGraph (a linear chain, as a captured training step is): N dependent launches of
(a) an 8-byte elementwise kernel, (b) a 64 MB elementwise kernel, replayed; microseconds per node, and the same launched eagerly."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import coarse3d_amd  # noqa: F401  (sets the runtime flag graphs need before the first GPU call)
import torch
dev = "cuda"
for name, numel in (("tiny", 2), ("64MB", 16 << 20)):
    x = torch.zeros(numel, device=dev)
    n = 1000 if numel < 1000 else 200
    for _ in range(3):
        x.add_(1.0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            x.add_(1.0)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / 5 / n * 1e6
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        x.add_(1.0)
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {per:.2f} us per node replayed, {e0.elapsed_time(e1) / n * 1e3:.2f} us per launch eager (GPU timeline)", flush=True)
