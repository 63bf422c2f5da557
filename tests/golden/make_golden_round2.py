#!/usr/bin/env python3
"""Round-2 golden vectors from the REAL reference (build container only):
    python tests/golden/make_golden_round2.py
  heads.npz   stand-alone ``distributed_sinkhorn`` (pc_processor/models/sinkhorn.py:5-33) with its
              Exp(1) noise recorded, ``ProjectionV1.forward`` (projector.py:11-27) in train and eval
              mode, and ``SalsaNextProto.forward(..., proto_pl=bank)`` (salsanext_proto.py:515-518).
Uses the import shim and the RNG recorder of make_golden.py (importing that module generates
nothing: its gold_* functions only run from its own __main__)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as MG  # noqa: E402
import weights as W  # noqa: E402

R = MG.R


def gold_heads():
    arrs = {}
    g = np.random.Generator(np.random.PCG64(71))
    # ---- distributed_sinkhorn on cosine-like scores, n = 300 pixels, K = 20 prototypes
    out = torch.from_numpy(g.uniform(-1, 1, (300, 20)).astype(np.float32))
    torch.manual_seed(72)
    with MG.Recorder() as rec:
        q, idx = R.distributed_sinkhorn(out.clone())
    assert len(rec.exp) == 1
    arrs.update({"sink/out": out, "sink/noise": rec.exp[0], "sink/q": q, "sink/indexs": idx})
    # ---- ProjectionV1(32 -> 16) stand-alone, train then eval
    proj = R.ProjectionV1(32, 16)
    sd = proj.state_dict()
    for k in sd:
        if sd[k].is_floating_point():
            gen = g
            if k.endswith("running_var") or (k.endswith("weight") and sd[k].dim() == 1):
                sd[k] = torch.from_numpy(gen.uniform(0.5, 1.5, tuple(sd[k].shape)).astype(np.float32))
            else:
                sd[k] = torch.from_numpy(gen.uniform(-0.3, 0.3, tuple(sd[k].shape)).astype(np.float32))
    proj.load_state_dict(sd)
    for k, v in sd.items():
        arrs[f"proj/state/{k}"] = v
    x = torch.from_numpy(g.standard_normal((2, 32, 8, 32)).astype(np.float32))
    proj.train()
    y_train = proj(x)
    arrs.update({"proj/x": x, "proj/y_train": y_train, "proj/run_mean": proj.proj[1].running_mean.clone(),
                 "proj/run_var": proj.proj[1].running_var.clone()})
    proj.eval()
    arrs["proj/y_eval"] = proj(x)
    # ---- SalsaNextProto.forward with proto_pl: the bank is REPLACED by proto_pl before the update
    b, h, w, ncls, seed = 2, 32, 64, 20, 101
    st = W.closed_form_state(nclasses=ncls)
    xin, tr, ev = W.synthetic_batch(b, h, w, ncls, seed, 0.02, gh=8, gw=16)
    masks = W.dropout_masks_for(None, b, seed + 1)
    m = MG.build_ref_model(st, "SemanticKitti", ncls)
    m.train()
    MG.install_dropout_masks(m, masks)
    bank = torch.nn.functional.normalize(torch.from_numpy(g.standard_normal((ncls, 20, 256)).astype(np.float32)), dim=-1)
    torch.manual_seed(seed + 2)
    with MG.Recorder() as rec:
        o = m(xin, label=tr, eval_mask=tr > 0, return_feat=True, proto_loss=True, proto_pl=bank)
    present = [c for c in range(1, ncls) if int((tr == c).sum()) > 0]
    assert len(present) == len(rec.exp)
    for c, e in zip(present, rec.exp):
        arrs[f"pl/gumbel_{c}"] = e
    arrs.update({"pl/bank": bank, "pl/new_prototypes": m.state_dict()["prototypes"],
                 "pl/contrast_target": o["contrast_target"], "pl/contrast_logits_sub": o["contrast_logits"][::16]})
    # without proto_loss the bank is just replaced
    m2 = MG.build_ref_model(st, "SemanticKitti", ncls)
    m2.train()
    MG.install_dropout_masks(m2, masks)
    m2(xin, label=tr, eval_mask=tr > 0, return_feat=True, proto_loss=False, proto_pl=bank)
    arrs["pl/replaced_only"] = m2.state_dict()["prototypes"]
    MG.npz("heads.npz", **arrs)


def gold_squeezeseg():
    """Reference SqueezeSegV3Proto(layers=21) forward + backward with closed-form weights and
    injected Dropout2d masks (second backbone): outputs, running statistics,
    per-tensor gradient checksums and a few small gradients in full."""
    import contextlib
    import importlib
    import io
    m = importlib.import_module("pc_processor.models.squeezesegv3_Proto")
    arrs = {}
    for tag, b, h, w, ncls in (("kitti", 2, 8, 64, 20), ("poss", 1, 8, 40, 14)):
        with contextlib.redirect_stdout(io.StringIO()):
            net = m.SqueezeSegV3Proto(nclasses=ncls, use_prototype=True, layers=21)
        missing = net.load_state_dict(W.squeezeseg_state(nclasses=ncls))
        assert not missing.missing_keys and not missing.unexpected_keys
        net.train()
        masks = W.squeezeseg_masks(b, 3)
        seq = iter(["enc1", "enc2", "enc3", "enc4", "enc5"])
        net.backbone.dropout.forward = lambda x, seq=seq, masks=masks: x * masks[next(seq)][:, :, None, None]
        net.decoder.dropout.forward = lambda x, masks=masks: x * masks["decoder"][:, :, None, None]
        net.head5[0].forward = lambda x, masks=masks: x * masks["head"][:, :, None, None]
        x, dp, df = W.rangenet_inputs(b, h, w, ncls)
        with contextlib.redirect_stdout(io.StringIO()):
            out = net(x, return_feat=True)
        loss = (out["pred_2d"] * dp).sum() + (out["feat_2d"] * df).sum()
        loss.backward()
        arrs.update({f"{tag}/pred_2d": out["pred_2d"].detach(), f"{tag}/feat_2d_sub": out["feat_2d"].detach()[:, ::4, :, ::2]})
        names = []
        for k, p_ in net.named_parameters():
            if p_.grad is None:
                continue
            names.append(k)
            gd = p_.grad.double()
            arrs[f"{tag}/gsum/{k}"] = gd.sum()
            arrs[f"{tag}/gsq/{k}"] = (gd * gd).sum()
        for k in ("backbone.conv1.weight", "head5.1.weight", "head5.1.bias", "decoder.dec1.upconv.weight",
                  "backbone.enc1.residual_0.attention_x.0.weight", "backbone.enc1.residual_0.position_mlp_2.0.weight",
                  "backbone.enc3.residual_1.position_mlp_2.4.weight", "backbone.enc2.conv.weight",
                  "projector.proj.3.bias", "decoder.dec4.conv.bias", "decoder.dec1.residual.conv2.weight"):
            arrs[f"{tag}/grad/{k}"] = dict(net.named_parameters())[k].grad
        arrs[f"{tag}/grad_names"] = np.array(names)
        sd = net.state_dict()
        for k in ("backbone.bn1.running_mean", "backbone.enc3.bn.running_var", "backbone.enc2.residual_0.attention_x.1.running_mean",
                  "backbone.enc5.residual_0.position_mlp_2.4.running_var", "decoder.dec1.residual.bn2.running_mean",
                  "projector.proj.1.running_var"):
            arrs[f"{tag}/run/{k}"] = sd[k]
    MG.npz("squeezeseg.npz", **arrs)


if __name__ == "__main__":
    which = sys.argv[1:] or ["heads", "squeezeseg"]
    if "heads" in which:
        gold_heads()
    if "squeezeseg" in which:
        gold_squeezeseg()
