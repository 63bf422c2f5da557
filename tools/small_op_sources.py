"""Where do the step's small torch launches (fills, copies, clones) come from?

Runs one warm training step of the headline workload under a TorchDispatchMode, records every
aten fill/zero/copy/clone-like call with the innermost coarse3d_amd (or bench) frame that made
it, and prints the counts.  GPU box only.  usage: python tools/small_op_sources.py [B H W]
"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

WATCH = ("fill", "zero", "copy", "clone", "full", "ones", "contiguous", "_to_copy", "cat", "index", "arange",
         "mul", "add", "sub", "div", "eq", "ne", "gt", "lt", "where", "sum", "select", "scatter", "gather")


class Tracer(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.counts = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        out = func(*args, **(kwargs or {}))
        dev = None
        for a in list(args) + [out]:
            if isinstance(a, torch.Tensor):
                dev = a.device.type
                break
        if dev == "cuda":
            where = "?"
            for fr in reversed(traceback.extract_stack(limit=30)):
                if "coarse3d_amd" in fr.filename or fr.filename.endswith("bench.py"):
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                    break
            self.counts[(name, where)] += 1
        return out


def main():
    b, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 64, 2048)
    import bench
    from coarse3d_amd import ops
    from coarse3d_amd.pc_processor.models import SalsaNextProto
    from coarse3d_amd.trainer import TrainStep
    dev = torch.device("cuda:0")
    ops.set_matrix_precision("bf16x3")
    torch.manual_seed(1)
    model = SalsaNextProto(5, 20, 20, 0, use_prototype=True).to(dev).train()
    ts = TrainStep(model, 20, lr=1e-3, n_epochs=100, temperature=0.07, num_anchor=512, loss_w_ce_2d=1.0,
                   loss_w_lov_2d=1.0, loss_w_contrast=0.1, feature_mean=bench.FEATURE_MEAN,
                   feature_std=bench.FEATURE_STD, proto_loss=True, inputs_resident=True)
    batches = [bench.synth_batch(b, h, w, 20, 1000 + s, dev, 1e-3) for s in range(3)]
    for s in range(2):
        ts.step(*batches[s], epoch=10)
    torch.cuda.synchronize()
    with Tracer() as tr:
        ts.step(*batches[2], epoch=10)
    torch.cuda.synchronize()
    byname = collections.Counter()
    for (name, where), n in tr.counts.items():
        byname[name] += n
    print("== per aten op ==")
    for name, n in byname.most_common(40):
        print(f"{n:5d}  {name}")
    print("== per (op, site) ==")
    for (name, where), n in tr.counts.most_common(70):
        print(f"{n:5d}  {name:32s} {where}")


if __name__ == "__main__":
    main()
