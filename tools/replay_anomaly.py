#!/usr/bin/env python3
"""Diagnostic for the slow-replay reading of DESIGN.md (d) 8: some processes replay the captured headline step 5-10 % slower than
they run it launch by launch.  One process, several timed passes of K steps each over the same batches:

    eager_1 -> captured_A (first K) -> captured_A (next K) -> captured_B (a second TrainStep: its own capture) -> eager_2
    -> captured_A again

so that one line tells a transient from a persistent slowdown, a property of ONE instantiated graph from a property of the process,
and whether the eager path of the same process moved at all.  usage (GPU box): python tools/replay_anomaly.py [K]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
wl = dict(net="salsanext", height=64, width=2048, classes=20, dataset="SemanticKitti", batch=8, matrix_dtype="bf16x3", storage=None)
b = bench.Bench(wl, dev, 0, 1, False)
batches = b.batches(K + 4)


def timed(ts, n=K):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(n):
        ts.step(*batches[4 + s % K], epoch=10)
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 3)


def warm(ts, n=4):
    for s in range(n):
        ts.step(*batches[s], epoch=10)
    torch.cuda.synchronize()


out = {}
_, eager = b.build(False)
warm(eager)
out["eager_1"] = timed(eager)
_, cap_a = b.build(True)
warm(cap_a)                                   # two eager steps, the capture, one replay
out["captured_A_first"] = timed(cap_a)
out["captured_A_next"] = timed(cap_a)
out["mem_after_A"] = [torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20]
_, cap_b = b.build(True)
warm(cap_b)
out["captured_B"] = timed(cap_b)
out["eager_2"] = timed(eager)
out["captured_A_again"] = timed(cap_a)
out["slow"] = bool(out["captured_A_first"] > 1.02 * out["eager_1"])
print(json.dumps(out), flush=True)
