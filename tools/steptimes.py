import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
from coarse3d_amd import ops, trainer
from coarse3d_amd.pc_processor.models import SalsaNextProto
dev = torch.device("cuda", 0)
torch.manual_seed(1)
model = SalsaNextProto(5, 20, 20, 0, use_prototype=True).to(dev).train()
ts = trainer.TrainStep(model, 20, lr=1e-3, n_epochs=100, temperature=0.07, num_anchor=512, loss_w_contrast=0.1,
                       feature_mean=bench.FEATURE_MEAN, feature_std=bench.FEATURE_STD, proto_loss=True, inputs_resident=True)
batches = [bench.synth_batch(8, 64, 2048, 20, 1000 + s, dev, 1e-3) for s in range(30)]
torch.cuda.synchronize()
out = []
for s in range(30):
    t0 = time.perf_counter()
    ts.step(*batches[s], epoch=10)
    torch.cuda.synchronize()
    out.append(round((time.perf_counter() - t0) * 1e3, 1))
print(os.environ.get("C3D_FUSE_BN_REDUCE"), out)
