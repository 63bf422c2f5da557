#!/usr/bin/env python3
"""N2 measurement: augment + range-project + loader tensors for a 120k-point scan at 64x2048 on
the device vs the CPU oracle (numpy restatement of the reference loader path) on the same host.
Prints one JSON line: scans/s, the HBM-roofline view of the projection kernels, pixel agreement."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from coarse3d_amd.pc_processor.dataset.preprocess import Augmentor, RangeProjection
from oracle import coarse3d_oracle as oc

g = np.random.Generator(np.random.PCG64(3))
n, w, h = 120_000, 2048, 64
yaw, pitch, r = g.uniform(-np.pi, np.pi, n), np.deg2rad(g.uniform(-25, 3, n)), g.uniform(2, 80, n)
pc = np.stack([r * np.cos(pitch) * np.cos(yaw), r * np.cos(pitch) * np.sin(yaw), r * np.sin(pitch), g.uniform(0, 1, n)], 1).astype(np.float32)
sem = g.integers(0, 20, n)
weak = sem * (g.random(n) < 0.001)
rp = RangeProjection(3, -25, w, h, -180, 180)
dev = "cuda"
pcd0 = torch.from_numpy(pc).to(dev)
semd, weakd = torch.from_numpy(sem).to(dev), torch.from_numpy(weak).to(dev)
aug = (True, False, (1.5, -0.7, 0.2), (2.0, -3.0, 77.0))

def gpu_once():
    p = pcd0.clone()
    Augmentor.apply(p, *aug)
    return rp.project_scan(p, semd, weakd)

out = gpu_once(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
e0.record()
for _ in range(reps):
    gpu_once()
e1.record(); torch.cuda.synchronize()
gpu_ms = e0.elapsed_time(e1) / reps

t0 = time.perf_counter()
cpu_reps = 3
for _ in range(cpu_reps):
    a = oc.augment_points(pc, *aug)
    pr = oc.range_projection(a, 3, -25, -180, 180, w, h)
    lt = oc.loader_tensors(pr, sem, weak)
cpu_ms = (time.perf_counter() - t0) / cpu_reps * 1e3
ux = out["uproj_x_idx"].cpu().numpy(); uy = out["uproj_y_idx"].cpu().numpy()
agree = float(((ux == pr["ux"]) & (uy == pr["uy"])).mean())
# algorithmic bytes: points read twice (augment, project) + written once, per-point outputs, z-buffer, images
alg = n * 16 * 3 + n * 12 + h * w * (8 * 2 + 4 * 9)
print(json.dumps({"metric": "scans/sec, augment + range projection + loader tensors, 120k points -> 64x2048", "gpu_scans_per_s": round(1e3 / gpu_ms, 1),
                  "gpu_ms": round(gpu_ms, 4), "cpu_oracle_scans_per_s": round(1e3 / cpu_ms, 2), "cpu_ms": round(cpu_ms, 2),
                  "algorithmic_MB": round(alg / 1e6, 2), "achieved_GBps": round(alg / gpu_ms / 1e6, 1),
                  "pixel_agreement_with_cpu": agree}))
