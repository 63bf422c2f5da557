#!/usr/bin/env python3
"""Host time per training step, launch by launch vs as ONE replayed hipGraph (TrainStep(graph=True)), with the GPU time
next to it.  usage: python tools/host_graph.py [H W classes batch [matrix-dtype]]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from coarse3d_amd import ops, trainer
from coarse3d_amd.pc_processor.models import SalsaNextProto

H, W, C, B = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (64, 2048, 20, 8)))
mode = sys.argv[5] if len(sys.argv) > 5 else "bf16x3"
dev = torch.device("cuda", 0)
ops.set_matrix_precision(mode)
batches = [bench.synth_batch(B, H, W, C, 1000 + s, dev, 1e-3) for s in range(16)]
for graph in (False, True):
    torch.manual_seed(1)
    model = SalsaNextProto(5, C, 20, 0, use_prototype=True).to(dev).train()
    ts = trainer.TrainStep(model, C, lr=1e-3, n_epochs=100, temperature=0.07, num_anchor=512, loss_w_contrast=0.1,
                           feature_mean=bench.FEATURE_MEAN, feature_std=bench.FEATURE_STD, proto_loss=True,
                           inputs_resident=True, graph=graph)
    for s in range(4):
        ts.step(*batches[s], epoch=10)
    torch.cuda.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for s in range(4, 16):
        h0 = time.perf_counter()
        ts.step(*batches[s], epoch=10)
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{H}x{W} C={C} bs={B} {mode} {'graph replay' if graph else 'launch by launch'}: step {el / 12 * 1e3:.2f} ms wall "
          f"({B * 12 / el:.1f} img/s), host time inside step() {host / 12 * 1e3:.2f} ms", flush=True)
    del ts, model
    torch.cuda.empty_cache()
