"""the RangeNet (Darknet-21) prototype backbone on the HIP engine
(coarse3d_amd/rangenet.py) against the golden vectors captured from the reference RangeNetProto
(tests/golden/rangenet.npz) with closed-form weights and injected Dropout2d masks.
Forward 1e-4 of max|ref|; every parameter gradient through its (sum, sum of squares) checksum and
a few small tensors element-wise, at the noise-calibrated tolerance used for the SalsaNext
backbone (whole-network fp32 gradients are ~1 % noisy)."""
import os

import numpy as np
import pytest
import torch

import weights as W
from _measure import record

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("tag,b,h,w,ncls,dataset", [("kitti", 2, 8, 64, 20, "SemanticKitti"), ("poss", 1, 8, 40, 14, "SemanticPOSS")])
def test_rangenet_backbone_vs_reference_golden(tag, b, h, w, ncls, dataset):
    from coarse3d_amd.rangenet import RangeNetBackbone
    d = np.load(os.path.join(GOLD, "rangenet.npz"))
    st = {k: v.to(DEV) for k, v in W.rangenet_state(nclasses=ncls).items()}
    x, dp, df = W.rangenet_inputs(b, h, w, ncls, w + 24 if dataset == "SemanticPOSS" else None)
    masks = {k: v.to(DEV) for k, v in W.rangenet_masks(b, 3).items()}
    bb = RangeNetBackbone(st, ncls, dataset)
    out = bb.forward(x.to(DEV), True, masks, True)
    pred = out["prob"].permute(0, 3, 1, 2)
    feat = out["feat"].permute(0, 3, 1, 2)
    assert rel(pred, torch.from_numpy(d[f"{tag}/pred_2d"])) < 1e-4
    assert rel(feat[:, ::4, :, ::2], torch.from_numpy(d[f"{tag}/feat_2d_sub"])) < 1e-4
    for k in d.files:
        if k.startswith(f"{tag}/run/"):
            n = k.split("/", 2)[2]
            assert float((st[n].cpu() - torch.from_numpy(d[k])).abs().max()) < 1e-5, n
    grads = bb.backward(dp.permute(0, 2, 3, 1).contiguous().to(DEV), df.permute(0, 2, 3, 1).contiguous().to(DEV))
    names = [str(n) for n in d[f"{tag}/grad_names"]]
    bad = []
    for n in names:
        gd = grads[n].double().cpu()
        sq = float(d[f"{tag}/gsq/{n}"])
        if n.endswith(("upconv.bias", "proj.0.bias")):       # bias in front of BatchNorm: exactly 0 up to noise
            assert float((gd * gd).sum()) < 1e-8, n
            continue
        if abs(float((gd * gd).sum()) - sq) > 7e-3 * sq + 1e-20:        # measured: median 8e-5, max 3.4e-3 (bound 2x)
            bad.append((n, float((gd * gd).sum()), sq))
    worst = max((abs(float((grads[n].double().cpu() ** 2).sum()) - float(d[f"{tag}/gsq/{n}"])) / (float(d[f"{tag}/gsq/{n}"]) + 1e-30)
                 for n in names if not n.endswith(("upconv.bias", "proj.0.bias"))), default=0.0)
    record(f"rangenet/{tag}/grad_sumsq_rel_err_max", worst)
    assert not bad, bad[:5]
    for k in d.files:
        if k.startswith(f"{tag}/grad/"):
            n = k.split("/", 2)[2]
            assert rel(grads[n], torch.from_numpy(d[k])) < 2e-2, n                 # measured <= 6.4e-3


def test_rangenet_module_api_and_training_step():
    """pc_processor.models.RangeNetProto mirror: reference state_dict surface, autograd hand-off
    (same golden as above through ``loss.backward()``), eval mode, and a full TrainStep (prototype
    bank, focal + Lovasz head, pseudo-label selection, contrast loss, AdamW) that lowers its loss."""
    from coarse3d_amd.pc_processor.models import RangeNetProto
    from coarse3d_amd.trainer import TrainStep
    d = np.load(os.path.join(GOLD, "rangenet.npz"))
    b, h, w, ncls = 2, 8, 64, 20
    m = RangeNetProto(layers=21, nclasses=ncls, use_prototype=True, dataset="SemanticKitti")
    st = W.rangenet_state(nclasses=ncls)
    assert set(m.state_dict()) == set(st) and all(tuple(m.state_dict()[k].shape) == tuple(v.shape) for k, v in st.items())
    m.load_state_dict(st)
    m.to(DEV).train()
    m.dropout_masks = {k: v.to(DEV) for k, v in W.rangenet_masks(b, 3).items()}
    x, dp, df = W.rangenet_inputs(b, h, w, ncls)
    out = m(x.to(DEV), return_feat=True)
    assert out["pred_2d"].shape == (b, ncls, h, w) and out["feat_2d"].shape == (b, 256, h, w)
    assert rel(out["pred_2d"], torch.from_numpy(d["kitti/pred_2d"])) < 1e-4
    ((out["pred_2d"] * dp.to(DEV)).sum() + (out["feat_2d"] * df.to(DEV)).sum()).backward()
    for k in ("head.1.weight", "decoder.dec1.upconv.weight", "backbone.conv1.weight"):
        assert rel(dict(m.named_parameters())[k].grad, torch.from_numpy(d[f"kitti/grad/{k}"])) < 2e-2, k
    m.eval()
    with torch.no_grad():
        e1, e2 = m(x.to(DEV))["pred_2d"], m(x.to(DEV))["pred_2d"]
    assert torch.isfinite(e1).all() and torch.equal(e1, e2)
    # a few optimisation steps on one batch
    import bench
    m.train()
    m.dropout_masks = None
    ts = TrainStep(m, ncls, proto_loss=True, lr=2e-3, num_anchor=32, feature_mean=bench.FEATURE_MEAN, feature_std=bench.FEATURE_STD)
    xb, tr, ev = bench.synth_batch(2, 16, 256, ncls, 5, DEV, label_rate=5e-2)
    losses = [float(ts.step(xb, tr, ev, epoch=10)["loss"]) for _ in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
