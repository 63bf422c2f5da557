import os, sys, json
sys.path.insert(0, '/root/repo')
import torch
from coarse3d_amd import ops
ops.set_matrix_precision("bf16x3")
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (B, H, W, Ci, Co, k, dil, pad) in [(8, 64, 2048, 64, 64, 3, 1, 1), (8, 16, 512, 256, 256, 3, 1, 1)]:
    for kind in ("random", "zeros", "random", "zeros"):
        x = (torch.randn if kind == "random" else torch.zeros)(B, H, W, Ci, device=dev)
        w = (torch.randn(Co, Ci, k, k, device=dev) * 0.05) if kind == "random" else torch.zeros(Co, Ci, k, k, device=dev)
        sc = torch.ones(Ci, device=dev); sh = torch.zeros(Ci, device=dev)
        taps = ops.conv_taps(k, k, dil, pad)
        wp = ops.pack_weights(w, 0)
        out = torch.empty(B, H, W, Co, device=dev)
        part = torch.empty(Co, 2, ops.num_mtiles(B, H, W), device=dev)
        ms = timeit(lambda: ops.conv_forward([ops.Source(x, sc, sh)], wp, None, Co, taps, lrelu=True, out=out, stat_partial=part, grad=True))
        fl = 2.0 * B * H * W * Ci * Co * k * k
        print(f"{Ci}->{Co} {H}x{W} {kind:6s}: {ms*1e3:7.1f} us  {fl/ms/1e9:6.1f} TF-equivalent  ({6*fl/ms/1e9:6.0f} bf16-TF/s)", flush=True)
