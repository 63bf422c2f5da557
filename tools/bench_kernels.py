#!/usr/bin/env python3
"""Micro-benchmarks of the MFMA engines at the BASELINE shapes (run on the GPU box)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarse3d_amd import ops

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def main():
    dev = "cuda"
    res = []
    shapes = [  # B,H,W,Cin,Cout,k,dil,pad
        (8, 64, 2048, 32, 32, 3, 1, 1), (8, 64, 2048, 32, 32, 3, 2, 2), (8, 64, 2048, 64, 64, 3, 2, 2),
        (8, 64, 2048, 64, 64, 2, 2, 1), (8, 64, 2048, 192, 64, 1, 1, 0), (8, 32, 1024, 128, 128, 3, 2, 2),
        (8, 16, 512, 256, 256, 3, 1, 1), (8, 8, 256, 256, 256, 3, 2, 2), (8, 4, 128, 256, 256, 3, 1, 1),
        (8, 32, 1024, 704, 704, 1, 1, 0), (8, 32, 1024, 704, 256, 1, 1, 0), (8, 64, 2048, 80, 32, 3, 1, 1),
    ]
    for (B, H, W, Ci, Co, k, dil, pad) in shapes:
        x = torch.randn(B, H, W, Ci, device=dev)
        w = torch.randn(Co, Ci, k, k, device=dev) * 0.05
        sc = torch.rand(Ci, device=dev) + 0.5; sh = torch.randn(Ci, device=dev) * 0.1
        bias = torch.randn(Co, device=dev)
        taps = ops.conv_taps(k, k, dil, pad)
        wp = ops.pack_weights(w, 0)
        src = ops.Source(x, sc, sh)
        out = torch.empty(B, H, W, Co, device=dev)
        part = torch.empty(ops.num_mtiles(B, H, W), Co, 2, device=dev)
        flops = 2.0 * B * H * W * Ci * Co * k * k
        ms = timeit(lambda: ops.conv_forward([src], wp, bias, Co, taps, lrelu=True, out=out, stat_partial=part))
        dz = torch.randn(B, H, W, Co, device=dev)
        dw = torch.zeros_like(w)
        ms_w = timeit(lambda: ops.conv_wgrad(src, dz, dw, taps))
        r = dict(shape=[B, H, W, Ci, Co, k, dil], fwd_ms=round(ms, 4), fwd_tflops=round(flops / ms / 1e9, 2),
                 wgrad_ms=round(ms_w, 4), wgrad_tflops=round(flops / ms_w / 1e9, 2))
        print(json.dumps(r), flush=True)
        res.append(r)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/bench_kernels.json", "w"), indent=1)

if __name__ == "__main__":
    main()
