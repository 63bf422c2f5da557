"""Stand-alone module mirrors against golden vectors of the reference (tests/golden/heads.npz,
tests/golden/make_golden_round2.py): ``distributed_sinkhorn`` (sinkhorn.py:5-33),
``ProjectionV1.forward`` (projector.py:11-27) and the ``proto_pl`` branch of
``SalsaNextProto.forward`` (salsanext_proto.py:515-518)."""
import os

import numpy as np
import pytest
import torch

import weights as W
from _measure import record

pytestmark = pytest.mark.gpu
DEV = "cuda"
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "heads.npz"))


def t(k):
    return torch.from_numpy(G[k])


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_distributed_sinkhorn_vs_reference():
    from coarse3d_amd.pc_processor.models import distributed_sinkhorn
    q, idx = distributed_sinkhorn(t("sink/out").to(DEV), noise=t("sink/noise"))
    assert q.shape == (300, 20) and idx.shape == (300,)
    agree_idx = record("sinkhorn/argmax_agreement", (idx.cpu() == t("sink/indexs")).float().mean().item())
    agree_q = record("sinkhorn/gumbel_onehot_agreement", (q.cpu().argmax(1) == t("sink/q").argmax(1)).float().mean().item())
    assert agree_idx == 1.0 and agree_q == 1.0           # measured: every one of the 300 rows
    assert torch.equal(q.sum(1).cpu(), torch.ones(300))
    with pytest.raises(ValueError):
        distributed_sinkhorn(t("sink/out").to(DEV), sinkhorn_iterations=5)


def test_projection_v1_forward_vs_reference():
    from coarse3d_amd.pc_processor.models import ProjectionV1
    proj = ProjectionV1(32, 16)
    proj.load_state_dict({k[len("proj/state/"):]: t(k) for k in G.files if k.startswith("proj/state/")})
    proj.to(DEV).train()
    y = proj(t("proj/x").to(DEV))
    assert y.shape == (2, 16, 8, 32)
    assert rel(y, t("proj/y_train")) < 1e-4
    assert rel(proj.proj[1].running_mean, t("proj/run_mean")) < 1e-5
    assert rel(proj.proj[1].running_var, t("proj/run_var")) < 1e-5
    proj.eval()
    assert rel(proj(t("proj/x").to(DEV)), t("proj/y_eval")) < 1e-4


def test_projection_v1_is_trainable_stand_alone():
    """projector.py:11-27 is an ordinary trainable nn.Module in the reference: a caller that trains it outside
    SalsaNextProto must get gradients (the stand-alone forward used to detach silently).  Forward,
    input gradient and all six parameter gradients against torch's own float64 nn.Sequential with the same weights."""
    from coarse3d_amd.pc_processor.models import ProjectionV1
    torch.manual_seed(4)
    proj = ProjectionV1(48, 32)
    ref = torch.nn.Sequential(torch.nn.Conv2d(48, 48, 1), torch.nn.BatchNorm2d(48), torch.nn.LeakyReLU(),
                              torch.nn.Conv2d(48, 32, 1)).double()
    ref.load_state_dict({k[len("proj."):]: v.double() for k, v in proj.state_dict().items()})
    x = torch.randn(2, 48, 8, 64)
    dy = torch.randn(2, 32, 8, 64)
    xr = x.double().requires_grad_(True)
    ref.train()
    (ref(xr) * dy.double()).sum().backward()
    proj.to(DEV).train()
    xd = x.to(DEV).requires_grad_(True)
    y = proj(xd)
    assert y.requires_grad
    (y * dy.to(DEV)).sum().backward()
    assert rel(y, ref(x.double())) < 1e-4
    assert rel(xd.grad, xr.grad) < 1e-4
    for (n, p), (_, pr) in zip(proj.proj.named_parameters(), ref.named_parameters()):
        assert p.grad is not None, n
        if n == "0.bias":       # a bias in front of a BatchNorm has no gradient (the mean is subtracted): both are rounding noise
            assert float(p.grad.abs().max()) < 1e-4 * float(proj.proj[0].weight.grad.abs().max())
        else:
            assert rel(p.grad, pr.grad) < 1e-4, n
    # no gradient requested anywhere: the inference path (no tape), same numbers
    with torch.no_grad():
        y2 = proj(x.to(DEV))
    assert not y2.requires_grad and rel(y2, ref(x.double())) < 1e-4


def test_proto_pl_replaces_the_bank_before_the_update():
    from coarse3d_amd.pc_processor.models import SalsaNextProto
    b, h, w, ncls, seed = 2, 32, 64, 20, 101
    x, tr, ev = W.synthetic_batch(b, h, w, ncls, seed, 0.02, gh=8, gw=16)
    masks = {k: v.to(DEV) for k, v in W.dropout_masks_for(None, b, seed + 1).items()}
    noise = torch.ones(b * h * w, 20)
    flat = tr.reshape(-1)
    for c in range(1, ncls):
        if f"pl/gumbel_{c}" in G.files:
            noise[flat == c] = t(f"pl/gumbel_{c}")
    bank = t("pl/bank").to(DEV)
    for proto_loss in (True, False):
        m = SalsaNextProto(5, ncls, 20, 0, use_prototype=True)
        m.load_state_dict(W.closed_form_state(nclasses=ncls))
        m.to(DEV).train()
        m.dropout_masks, m.gumbel_noise = masks, noise.to(DEV)
        out = m(x.to(DEV), label=tr.to(DEV), eval_mask=(tr > 0).to(DEV), return_feat=True, proto_loss=proto_loss,
                proto_pl=bank)
        if proto_loss:
            assert rel(m.prototypes, t("pl/new_prototypes")) < 1e-4
            assert rel(out["contrast_logits"][::16], t("pl/contrast_logits_sub")) < 1e-4
            got_t, ref_t = out["contrast_target"].cpu(), t("pl/contrast_target")
            agree = record("proto_pl/contrast_target_agreement", (got_t == ref_t).float().mean().item())
            assert agree >= 0.999
            # every differing Sinkhorn target explained: in float64, from the HIP run's own
            # similarity rows, the two best assignment scores of that pixel (sinkhorn.py:5-29) are within 1e-5 of each
            # other -- an argmax between two values closer than the fp32 rounding of exp(sim / 0.05)
            logits = out["contrast_logits"].detach().cpu().reshape(flat.numel(), 20, ncls)
            detail = []
            for pix in torch.nonzero(got_t != ref_t).reshape(-1).tolist():
                cls = int(flat[pix])
                rows = torch.nonzero(flat == cls).reshape(-1)
                q = torch.exp(logits[rows][:, :, cls].double() / 0.05).t()
                n_, k_ = q.shape[1], q.shape[0]
                q = q / q.sum()
                for _ in range(3):
                    q = q / q.sum(dim=1, keepdim=True) / k_
                    q = q / q.sum(dim=0, keepdim=True) / n_
                top = torch.topk((q * n_).t()[int((rows == pix).nonzero()[0])], 2).values
                gap = float((top[0] - top[1]) / top[0])
                detail.append({"pixel": pix, "class": cls, "target_got": int(got_t[pix]), "target_ref": int(ref_t[pix]),
                               "relative_gap_of_the_two_best_scores_float64": gap})
                assert gap < 1e-5, detail[-1]
            record("proto_pl/contrast_target_detail", detail)
        else:
            assert "contrast_logits" not in out
            assert torch.equal(m.prototypes.detach().cpu(), t("pl/replaced_only"))
        assert not m.prototypes.requires_grad and m.prototypes.data_ptr() != bank.data_ptr()



def test_classification_mode_vs_reference_golden():
    """``SalsaNextProto(..., classification=True)``: the reference's ImageNet pre-training mode (salsanext_proto.py:216-231,
    445-447: encoder -> global average pool -> Linear(256, 1000); VERDICT round 4, missing #5).  Golden from the reference
    class itself (tests/golden/make_golden_round5.py) in training mode with injected dropout masks: class scores, the
    head's gradients and encoder gradients at three depths (1e-4 / the noise-calibrated bounds of the other golden tests),
    gradient norms of all 122 tensors that get one, running statistics; the decoder, the segmentation head, the projector
    and the bank get no gradient, as in the reference."""
    import numpy as np
    from coarse3d_amd.pc_processor.models import SalsaNextProto
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "classify.npz"))
    b, h, w, ncls = 2, 32, 128, 20
    x, _, _ = W.synthetic_batch(b, h, w, ncls, 311, 0.02, gh=8, gw=16)
    sd = W.closed_form_state(nclasses=ncls)
    sd.update(W.fc_state())
    m = SalsaNextProto(5, ncls, 20, 0, classification=True)
    m.load_state_dict(sd)
    m.to(DEV).train()
    m.dropout_masks = {k: v.to(DEV) for k, v in W.dropout_masks_for(None, b, 312).items()}
    out = m(x.to(DEV))
    assert tuple(out.shape) == (b, 1000)
    ref = torch.from_numpy(g["cls_out"])
    assert float((out.detach().cpu() - ref).abs().max() / ref.abs().max()) < 1e-4
    wgt = torch.from_numpy(np.random.Generator(np.random.PCG64(313)).standard_normal((b, 1000)).astype(np.float32)).to(DEV)
    (out * wgt).sum().backward()
    got = {k for k, p in m.named_parameters() if p.grad is not None}
    assert got == set(g["with_grad"].tolist())
    for k in ("fc.linear.weight", "fc.linear.bias", "resBlock5.conv5.weight", "resBlock2.conv3.weight", "downCntx.conv1.weight",
              "resBlock3.bn2.weight"):
        gr = dict(m.named_parameters())[k].grad.detach().cpu()
        r = torch.from_numpy(g[f"grad/{k}"])
        gr = gr if gr.numel() <= 20000 else gr.reshape(-1)[::16]
        err = float((gr.reshape(-1) - r.reshape(-1)).abs().max() / (r.abs().max() + 1e-12))
        assert err < (1e-4 if k.startswith("fc") else 8e-2), (k, err)       # (encoder: the whole-network noise bound of the golden step)
    worst = 0.0
    for k, p in m.named_parameters():
        if p.grad is not None:
            worst = max(worst, abs(float(p.grad.norm()) - float(g[f"gnorm/{k}"])) / (float(g[f"gnorm/{k}"]) + 1e-9))
    assert worst < 5e-2, worst
    sdo = m.state_dict()
    for k, key in (("run_mean", "resBlock5.bn4.running_mean"), ("run_var", "resBlock5.bn4.running_var")):
        r = torch.from_numpy(g[k])
        assert float((sdo[key].cpu() - r).abs().max() / r.abs().max()) < 2e-4

