#!/usr/bin/env python3
"""Does handing VRAM back to the driver disturb kernels that run afterwards?  (the slow-replay reading only ever
hits bench.py's captured pass, which follows `del model; torch.cuda.empty_cache()` of the eager pass.)

Times a bandwidth-bound copy (2 x 1 GiB per launch, ~0.5 ms) back to back for a few seconds, three times:
  baseline            nothing freed
  after_free          20 GiB allocated through torch, touched, then released to the driver (empty_cache) right before the loop
  after_child_exit    a child process that allocated and touched 20 GiB has just exited
and prints, per phase, the median launch time and every launch more than 1.5x the median with its offset from the start of the loop.
usage (GPU box): python tools/free_then_stream.py [GiB]"""
import json
import subprocess
import sys
import time

import torch

GIB = int(sys.argv[1]) if len(sys.argv) > 1 else 20
if len(sys.argv) > 2 and sys.argv[2] == "child":
    x = torch.empty(GIB << 30, dtype=torch.uint8, device="cuda")
    x.fill_(1)
    torch.cuda.synchronize()
    sys.exit(0)

dev = "cuda"
src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
dst = torch.empty_like(src)
src.fill_(3)
torch.cuda.synchronize()


def stream_for(seconds, tag):
    n = int(seconds / 0.0005)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(n):
        dst.copy_(src)
        ev[i + 1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    med = sorted(ms)[n // 2]
    off, slow, t = [], 0.0, 0.0
    for m in ms:
        if m > 1.5 * med:
            off.append((round(t, 1), round(m, 3)))
            slow += m - med
        t += m
    print(json.dumps({"phase": tag, "launches": n, "median_ms": round(med, 4), "GBps": round(2 * (1 << 30) / med / 1e6, 0),
                      "wall_s": round(wall, 3), "excess_ms_total": round(slow, 2), "n_slow": len(off),
                      "slow_launches_offset_ms_and_ms": off[:40]}), flush=True)


stream_for(2.0, "baseline")
big = torch.empty(GIB << 30, dtype=torch.uint8, device=dev)
big.fill_(1)
torch.cuda.synchronize()
del big
torch.cuda.empty_cache()
stream_for(4.0, "after_free_%dGiB" % GIB)
stream_for(2.0, "baseline_again")
subprocess.run([sys.executable, __file__, str(GIB), "child"], check=True)
stream_for(6.0, "after_child_exit_%dGiB" % GIB)
