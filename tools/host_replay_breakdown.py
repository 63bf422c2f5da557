#!/usr/bin/env python3
"""Host time of the parts of one captured step (TrainStep._graph_step) at the headline shape."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coarse3d_amd  # noqa: F401,E402
import torch  # noqa: E402
import bench  # noqa: E402
from coarse3d_amd import trainer  # noqa: E402
from coarse3d_amd.pc_processor.models import SalsaNextProto  # noqa: E402

H, W, C, B = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (64, 2048, 20, 8)))
dev = torch.device("cuda", 0)
batches = [bench.synth_batch(B, H, W, C, 1000 + s, dev, 1e-3) for s in range(8)]
torch.manual_seed(1)
model = SalsaNextProto(5, C, 20, 0, use_prototype=True).to(dev).train()
ts = trainer.TrainStep(model, C, lr=1e-3, n_epochs=100, temperature=0.07, num_anchor=512, loss_w_contrast=0.1,
                       feature_mean=bench.FEATURE_MEAN, feature_std=bench.FEATURE_STD, proto_loss=True,
                       inputs_resident=True, graph=True)
for s in range(6):
    ts.step(*batches[s], epoch=10)
torch.cuda.synchronize()
ent = next(e for e in ts._graphs.values() if e["graph"] is not None)
acc = {}


def t(name, fn):
    t0 = time.perf_counter()
    fn()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0


N = 12
for s in range(N):
    x, tr, ev = batches[s % 8]
    t("poll", ts._poll_capacity)
    t("copy_in", lambda: (ent["sx"].copy_(x), ent["st"].copy_(tr), ent["se"].copy_(ev)))
    t("replay", ent["graph"].replay)
    t("watch", lambda: ts._watch_capacity(ent["res"]["lov_count"]))
torch.cuda.synchronize()
print({k: round(v / N * 1e3, 3) for k, v in acc.items()}, "ms per step; DEBUG_CLR_GRAPH_PACKET_CAPTURE =",
      os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"))
# replay with the GPU idle (launch cost alone)
torch.cuda.synchronize()
t0 = time.perf_counter()
ent["graph"].replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("one replay on an idle GPU: host", round((t1 - t0) * 1e3, 3), "ms; until done", round((time.perf_counter() - t0) * 1e3, 3), "ms")
