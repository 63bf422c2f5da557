import sys, os, torch
sys.path.insert(0, "/root/repo")
from coarse3d_amd import ops
dev = "cuda"
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for c, n in ((32, 4096), (64, 4096), (128, 1024), (256, 256), (704, 1024)):
    part = torch.randn(c, 2, n, device=dev).abs()
    g, b_ = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    us = timeit(lambda: ops.bn_finalize_partials(part, float(n * 256), g, b_, rm, rv))
    print(c, n, round(us, 2), "us per bn_finalize_partials (launch-to-launch)")
