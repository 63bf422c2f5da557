"""SURVEY 8f N3, second backbone: SqueezeSegV3Proto on the HIP engine (coarse3d_amd/squeezeseg.py)
against golden vectors captured from the reference module (tests/golden/squeezeseg.npz, closed-form
weights, injected Dropout2d masks) and against the CPU oracle.  Forward 1e-4 of max|ref|; every
parameter gradient through its sum-of-squares checksum and several tensors element-wise."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import weights as W
from _measure import record

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CANCELLED = ("attention_x.0.bias", "position_mlp_2.0.bias", "position_mlp_2.3.bias", "upconv.bias", ".conv.bias", "proj.0.bias")


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_sac_kernels_vs_torch():
    """c3d_sac_im2col7 / modulate / modulate_bwd / fold against unfold, conv2d and autograd."""
    from coarse3d_amd import ops
    g = torch.Generator().manual_seed(3)
    b, c, h, w = 2, 8, 6, 20
    feat = torch.randn(b, c, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    xyz = torch.randn(b, 3, h, w, generator=g, dtype=torch.float64)
    att = torch.randn(b, 9 * c, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    sc = torch.rand(9 * c, generator=g, dtype=torch.float64) + 0.5
    sh = torch.randn(9 * c, generator=g, dtype=torch.float64) * 0.3
    wgt = torch.randn(5, 3, 7, 7, generator=g, dtype=torch.float64)
    # im2col: conv7x7(xyz) == xcol @ W.reshape(., 147)
    xyz4 = torch.zeros(b, h, w, 4)
    xyz4[..., :3] = xyz.permute(0, 2, 3, 1).float()
    xcol = ops.sac_im2col7(xyz4.to(DEV)).cpu().double()
    assert xcol.shape == (b, h, w, 160) and float(xcol[..., 147:].abs().max()) == 0
    ref = F.conv2d(xyz, wgt, padding=3).permute(0, 2, 3, 1)
    assert rel(xcol[..., :147] @ wgt.reshape(5, 147).t(), ref) < 1e-6
    # modulate forward / backward
    m_ref = F.unfold(feat, 3, padding=1).view(b, 9 * c, h, w) * torch.sigmoid(att * sc[None, :, None, None] + sh[None, :, None, None])
    nh = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().float().to(DEV)
    m = ops.sac_modulate(nh(feat), nh(att), sc.float().to(DEV), sh.float().to(DEV))
    assert rel(m.permute(0, 3, 1, 2), m_ref) < 1e-5
    dm = torch.randn(m_ref.shape, generator=g, dtype=torch.float64)
    bnout = (att * sc[None, :, None, None] + sh[None, :, None, None]).detach().requires_grad_(True)
    m2 = F.unfold(feat, 3, padding=1).view(b, 9 * c, h, w) * torch.sigmoid(bnout)
    dfeat_ref, dbn_ref = torch.autograd.grad((m2 * dm).sum(), [feat, bnout])
    dmd = nh(dm)
    datt = ops.sac_modulate_bwd(dmd, nh(feat), nh(att), sc.float().to(DEV), sh.float().to(DEV))
    assert rel(datt.permute(0, 3, 1, 2), dbn_ref) < 1e-5
    base = torch.randn(b, h, w, c, generator=g)
    dfeat = ops.sac_fold(dmd, base.clone().to(DEV), True)
    assert rel(dfeat.permute(0, 3, 1, 2) - base.permute(0, 3, 1, 2).to(DEV), dfeat_ref) < 1e-5
    dfeat0 = ops.sac_fold(dmd, torch.full((b, h, w, c), 7.0, device=DEV), False)
    assert rel(dfeat0.permute(0, 3, 1, 2), dfeat_ref) < 1e-5


@pytest.mark.parametrize("tag,b,h,w,ncls", [("kitti", 2, 8, 64, 20), ("poss", 1, 8, 40, 14)])
def test_squeezeseg_backbone_vs_reference_golden(tag, b, h, w, ncls):
    from coarse3d_amd.squeezeseg import SqueezeSegBackbone
    d = np.load(os.path.join(GOLD, "squeezeseg.npz"))
    st = {k: v.to(DEV) for k, v in W.squeezeseg_state(nclasses=ncls).items()}
    x, dp, df = W.rangenet_inputs(b, h, w, ncls)
    masks = {k: v.to(DEV) for k, v in W.squeezeseg_masks(b, 3).items()}
    bb = SqueezeSegBackbone(st, ncls, "SemanticKitti")
    out = bb.forward(x.to(DEV), True, masks, True)
    pred = out["prob"].permute(0, 3, 1, 2)
    feat = out["feat"].permute(0, 3, 1, 2)
    assert record(f"squeezeseg/{tag}/pred_rel_err", rel(pred, torch.from_numpy(d[f"{tag}/pred_2d"]))) < 1e-4
    assert record(f"squeezeseg/{tag}/feat_rel_err", rel(feat[:, ::4, :, ::2], torch.from_numpy(d[f"{tag}/feat_2d_sub"]))) < 1e-4
    for k in d.files:
        if k.startswith(f"{tag}/run/"):
            n = k.split("/", 2)[2]
            assert rel(st[n], torch.from_numpy(d[k])) < 1e-4, n
    grads = bb.backward(dp.permute(0, 2, 3, 1).contiguous().to(DEV), df.permute(0, 2, 3, 1).contiguous().to(DEV))
    names = [str(n) for n in d[f"{tag}/grad_names"]]
    assert sorted(names) == sorted(grads)
    worst, bad = 0.0, []
    for n in names:
        gd = grads[n].double().cpu()
        if n.endswith(CANCELLED):                           # a bias in front of a BatchNorm: zero up to noise
            assert float((gd * gd).sum()) < 1e-6, n
            continue
        sq = float(d[f"{tag}/gsq/{n}"])
        e = abs(float((gd * gd).sum()) - sq) / (sq + 1e-30)
        worst = max(worst, e)
        if e > 2e-2:
            bad.append((n, e))
    record(f"squeezeseg/{tag}/grad_sumsq_rel_err_max", worst)
    assert not bad, bad[:5]
    worst_t = 0.0
    for k in d.files:
        if k.startswith(f"{tag}/grad/"):
            n = k.split("/", 2)[2]
            if n.endswith(CANCELLED):
                continue
            e = rel(grads[n], torch.from_numpy(d[k]))
            worst_t = max(worst_t, e)
            assert e < 2e-2, (n, e)
    record(f"squeezeseg/{tag}/grad_tensor_rel_err_max", worst_t)


def test_squeezeseg_forward_vs_oracle_at_a_larger_size():
    """64 x 512 (eight times the fixture width, all three stride-2 stages with >1 tile): forward
    against the CPU oracle on the same closed-form weights."""
    from coarse3d_amd.squeezeseg import SqueezeSegBackbone
    from oracle import squeezeseg_oracle as so
    b, h, w, ncls = 1, 64, 512, 20
    st = W.squeezeseg_state(nclasses=ncls)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(b, 5, h, w, generator=g)
    masks = W.squeezeseg_masks(b, 5)
    with torch.no_grad():
        ref = so.squeezeseg_forward({k: v.clone() for k, v in st.items()}, x, True, masks, True)
    bb = SqueezeSegBackbone({k: v.to(DEV) for k, v in st.items()}, ncls, "SemanticKitti")
    out = bb.forward(x.to(DEV), True, {k: v.to(DEV) for k, v in masks.items()}, True)
    assert rel(out["prob"].permute(0, 3, 1, 2), ref["pred_2d"]) < 1e-4
    assert rel(out["feat"].permute(0, 3, 1, 2), ref["feat_2d"]) < 1e-4


def test_squeezeseg_module_api_and_training_step():
    """pc_processor.models.SqueezeSegV3Proto mirror: reference state_dict surface, autograd hand-off
    (same golden through ``loss.backward()``), head1..4 untouched, eval mode, and a full TrainStep
    (prototype bank, focal + Lovasz head, pseudo-label selection, contrast loss, AdamW) that lowers
    its loss."""
    from coarse3d_amd.pc_processor.models import SqueezeSegV3Proto
    from coarse3d_amd.trainer import TrainStep
    d = np.load(os.path.join(GOLD, "squeezeseg.npz"))
    b, h, w, ncls = 2, 8, 64, 20
    m = SqueezeSegV3Proto(nclasses=ncls, use_prototype=True)
    st = W.squeezeseg_state(nclasses=ncls)
    assert set(m.state_dict()) == set(st) and all(tuple(m.state_dict()[k].shape) == tuple(v.shape) for k, v in st.items())
    m.load_state_dict(st)
    m.to(DEV).train()
    m.dropout_masks = {k: v.to(DEV) for k, v in W.squeezeseg_masks(b, 3).items()}
    x, dp, df = W.rangenet_inputs(b, h, w, ncls)
    out = m(x.to(DEV), return_feat=True)
    assert out["pred_2d"].shape == (b, ncls, h, w) and out["feat_2d"].shape == (b, 256, h, w)
    assert rel(out["pred_2d"], torch.from_numpy(d["kitti/pred_2d"])) < 1e-4
    ((out["pred_2d"] * dp.to(DEV)).sum() + (out["feat_2d"] * df.to(DEV)).sum()).backward()
    P = dict(m.named_parameters())
    for k in ("head5.1.weight", "decoder.dec1.upconv.weight", "backbone.conv1.weight",
              "backbone.enc1.residual_0.attention_x.0.weight"):
        assert rel(P[k].grad, torch.from_numpy(d[f"kitti/grad/{k}"])) < 2e-2, k
    assert all(P[f"head{k}.1.weight"].grad is None for k in (1, 2, 3, 4))      # as in the reference: unused heads
    m.eval()
    with torch.no_grad():
        e1, e2 = m(x.to(DEV))["pred_2d"], m(x.to(DEV))["pred_2d"]
    assert torch.isfinite(e1).all() and torch.equal(e1, e2)
    import bench
    m.train()
    m.dropout_masks = None
    ts = TrainStep(m, ncls, proto_loss=True, lr=2e-3, num_anchor=32, feature_mean=bench.FEATURE_MEAN, feature_std=bench.FEATURE_STD)
    xb, tr, ev = bench.synth_batch(2, 16, 256, ncls, 5, DEV, label_rate=5e-2)
    losses = [float(ts.step(xb, tr, ev, epoch=10)["loss"]) for _ in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
