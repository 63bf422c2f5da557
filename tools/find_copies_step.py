#!/usr/bin/env python3
"""Which Python lines issue the device-to-device copies of one eager training step (rocprofv3 shows them as
__amd_rocclr_copyBuffer): aten::copy_ / clone calls counted by source line.  usage (GPU box): python tools/find_copies_step.py"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coarse3d_amd, torch, bench
from coarse3d_amd import trainer
from coarse3d_amd.pc_processor.models import SalsaNextProto
dev = torch.device("cuda", 0)
batches = [bench.synth_batch(8, 64, 2048, 20, 1000 + s, dev, 1e-3) for s in range(3)]
torch.manual_seed(1)
model = SalsaNextProto(5, 20, 20, 0, use_prototype=True, dataset="SemanticKitti").to(dev).train()
ts = trainer.TrainStep(model, 20, lr=1e-3, n_epochs=100, temperature=0.07, num_anchor=512, loss_w_contrast=0.1,
                       feature_mean=bench.FEATURE_MEAN, feature_std=bench.FEATURE_STD, proto_loss=True, inputs_resident=True, graph=False)
for s in range(2):
    ts.step(*batches[s], epoch=10)
torch.cuda.synchronize()
counts = collections.Counter()
orig_copy, orig_clone = torch.Tensor.copy_, torch.Tensor.clone
def where():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "coarse3d_amd" in fr.filename or fr.filename.endswith("bench.py"):
            return f"{os.path.relpath(fr.filename)}:{fr.lineno} {fr.line.strip()[:90]}"
    return "?"
def copy_(self, src, *a, **k):
    if self.is_cuda and getattr(src, "is_cuda", False):
        counts[("copy_", where())] += 1
    return orig_copy(self, src, *a, **k)
def clone(self, *a, **k):
    if self.is_cuda:
        counts[("clone", where())] += 1
    return orig_clone(self, *a, **k)
torch.Tensor.copy_, torch.Tensor.clone = copy_, clone
ts.step(*batches[2], epoch=10)
torch.cuda.synchronize()
torch.Tensor.copy_, torch.Tensor.clone = orig_copy, orig_clone
for (kind, loc), n in counts.most_common(40):
    print(f"{n:4d} {kind:6s} {loc}")
print("total", sum(counts.values()))
