import sys, os
sys.path.insert(0, "/root/repo")
import torch
from coarse3d_amd import ops
dev = "cuda"
dt = torch.bfloat16 if len(sys.argv) > 1 and sys.argv[1] == "bf16" else torch.float32
for (B, H, W, C) in [(8, 64, 2048, 32), (8, 64, 2048, 64), (8, 32, 1024, 128), (8, 16, 512, 256), (8, 32, 1024, 704)]:
    dy = torch.randn(B, H, W, C, device=dev).to(dt); a = torch.randn(B, H, W, C, device=dev).to(dt)
    k = torch.randn(3, C, device=dev)
    for name, fn, bytes_per in (("reduce", lambda: ops.bn_bwd_reduce(dy, a, C, 0), 8), ("apply", lambda: ops.bn_bwd_apply(dy, a, C, 0, k), 12)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(name, (B, H, W, C), "%.1f us  %.2f TB/s" % (ms * 1e3, B * H * W * C * bytes_per * (0.5 if dt == torch.bfloat16 else 1.0) / ms / 1e9))
