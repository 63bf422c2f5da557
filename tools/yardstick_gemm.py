#!/usr/bin/env python3
"""Yardstick only (never on the product path): library fp32 GEMM rate at the projector / similarity shapes."""
import torch, json
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
torch.backends.cuda.matmul.allow_tf32 = False
for (m, k, n) in [(262144, 704, 704), (262144, 704, 256), (262144, 256, 704), (1048576, 256, 400), (1048576, 192, 64), (1048576, 32, 32)]:
    a = torch.randn(m, k, device="cuda"); b = torch.randn(k, n, device="cuda"); c = torch.empty(m, n, device="cuda")
    ms = t(lambda: torch.mm(a, b, out=c))
    ms_w = t(lambda: torch.mm(a.t(), c))     # wgrad-shaped: [k,m]x[m,n]
    print(json.dumps(dict(m=m, k=k, n=n, fwd_ms=round(ms, 3), fwd_tf=round(2 * m * k * n / ms / 1e9, 1),
                          wgrad_ms=round(ms_w, 3), wgrad_tf=round(2 * m * k * n / ms_w / 1e9, 1))), flush=True)
