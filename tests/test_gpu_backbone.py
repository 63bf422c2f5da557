"""HIP backbone (coarse3d_amd.backbone) vs the CPU oracle on the same closed-form weights, inputs
and injected Dropout2d masks.  Forward: 1e-4 of max|ref| (north-star tolerance for fp32 logits).
Backward: the network's fp32 gradients are intrinsically noisy (see tests/test_oracle_golden.py),
so the HIP gradient error against a float64 oracle must stay within 3x the error the fp32 oracle
itself makes against float64 (median over tensors).  Single tensors may exceed that when one
LeakyReLU pre-activation lands on the other side of zero than in the oracle (the derivative jumps
1 <-> 0.01 and a gradient that is a cancelling sum over N pixels moves by ~1/sqrt(N)); at most
15 % of the tensors may do so.  The tight (1e-4) gradient checks are per kernel: test_gpu_conv.py,
test_gpu_ops.py."""
import numpy as np
import pytest
import torch

import weights as W
from _measure import record
from oracle import coarse3d_oracle as oc

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def run_oracle(st, x, masks, dataset, d_prob, d_feat, dtype):
    st = {k: (v.to(dtype).clone() if v.is_floating_point() else v.clone()) for k, v in st.items()}
    names = oc.trainable_names(st)
    for k in names:
        st[k].requires_grad_(True)
    out = oc.backbone_forward(st, x.to(dtype), True, {k: v.to(dtype) for k, v in masks.items()}, True, dataset)
    loss = (out["pred_2d"] * d_prob.to(dtype)).sum() + (out["feat_2d"] * d_feat.to(dtype)).sum()
    grads = torch.autograd.grad(loss, [st[k] for k in names], allow_unused=True)
    return out, st, {k: g for k, g in zip(names, grads) if g is not None}


@pytest.mark.parametrize("b,h,w,ncls,dataset,seed", [
    (2, 32, 64, 20, "SemanticKitti", 101),
    (1, 24, 56, 14, "SemanticPOSS", 201),
    (2, 64, 128, 20, "SemanticKitti", 77),
])
def test_backbone_forward_backward(b, h, w, ncls, dataset, seed):
    from coarse3d_amd.backbone import Backbone
    dev = "cuda"
    st = W.closed_form_state(nclasses=ncls)
    x, tr, ev = W.synthetic_batch(b, h, w, ncls, seed, 0.02, gh=8, gw=16)
    masks = W.dropout_masks_for(None, b, seed + 1)
    g = torch.Generator().manual_seed(seed)
    d_prob = torch.randn(b, ncls, h, w, generator=g)
    d_feat = torch.randn(b, 256, h, w, generator=g) * 0.05

    P = {k: v.to(dev).clone() for k, v in st.items()}
    bb = Backbone(P, ncls, dataset)
    out = bb.forward(x.to(dev), True, {k: v.to(dev) for k, v in masks.items()}, True)
    grads = bb.backward(d_prob.permute(0, 2, 3, 1).contiguous().to(dev),
                        d_feat.permute(0, 2, 3, 1).contiguous().to(dev))
    torch.cuda.synchronize()

    o32, st32, g32 = run_oracle(st, x, masks, dataset, d_prob, d_feat, torch.float32)
    o64, _, g64 = run_oracle(st, x, masks, dataset, d_prob, d_feat, torch.float64)

    prob = out["prob"].permute(0, 3, 1, 2).cpu()
    feat = out["feat"].permute(0, 3, 1, 2).cpu()
    assert rel(prob, o32["pred_2d"].detach()) < 1e-4
    assert rel(feat, o32["feat_2d"].detach()) < 1e-4
    lg = out["logits"][:, :h, :w, :ncls].permute(0, 3, 1, 2).cpu()
    assert rel(lg, o32["logits"].detach()) < 1e-4
    # batch statistics -> running stats
    for k in st32:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert rel(P[k].cpu(), st32[k]) < 2e-4, k
        if k.endswith("num_batches_tracked"):
            assert int(P[k]) == int(st32[k])
    # gradients
    bad = []
    e_hip, e_ora = [], []
    for k, ref in g64.items():
        if k == "projector.proj.0.bias":
            continue
        scale = float(ref.abs().max()) + 1e-30
        eh = float((grads[k].cpu().double() - ref).abs().max()) / scale
        eo = float((g32[k].double() - ref).abs().max()) / scale
        e_hip.append(eh)
        e_ora.append(eo)
        if eh > 3 * eo + 1e-4:
            bad.append((k, eh, eo))
    print(f"median err vs f64: hip {np.median(e_hip):.2e} oracle-fp32 {np.median(e_ora):.2e}; "
          f"max hip {max(e_hip):.2e} oracle {max(e_ora):.2e}")
    tag = f"backbone/{dataset}_{h}x{w}"
    record(f"{tag}/grad_err_vs_f64_median_hip", float(np.median(e_hip)))
    record(f"{tag}/grad_err_vs_f64_median_oracle_fp32", float(np.median(e_ora)))
    record(f"{tag}/grad_err_vs_f64_max_hip", float(max(e_hip)))
    record(f"{tag}/grad_err_vs_f64_max_oracle_fp32", float(max(e_ora)))
    record(f"{tag}/tensors_beyond_3x_oracle_noise", len(bad) / len(g64))
    # measured (round 2, profiles/round2_parity_measured.json): at most 6.8 % of the tensors beyond
    # 3x the fp32 oracle's own error, HIP median <= 1.25x and HIP max <= 1.0x the oracle's
    assert len(bad) <= 0.14 * len(g64), bad[:10]
    assert np.median(e_hip) < 2 * np.median(e_ora) + 1e-5
    assert max(e_hip) < 2 * max(e_ora) + 1e-4


def test_gradient_noise_at_a_realistic_population():
    """The gradient-noise criterion of test_backbone_forward_backward at 2x64x512 (65 536 pixels per
    first-level BatchNorm): the size class of real batches, where the bf16x3 engine runs its forward 3x3 / 2x2
    convolutions with six plane products (ops.SIX_FWD_MIN_PIXELS).  Measured (tools/noise_probe.py): median
    gradient error against float64 0.79x the fp32 oracle's with six products, 0.84x with eight, 1.15x on the
    fp32-MFMA engine; 2-4 of 191 tensors beyond 3x the oracle's error in all three."""
    from coarse3d_amd import ops
    from coarse3d_amd.backbone import Backbone
    b, h, w, ncls, dataset, seed = 2, 64, 512, 20, "SemanticKitti", 77
    assert b * h * w >= ops.SIX_FWD_MIN_PIXELS
    dev = "cuda"
    st = W.closed_form_state(nclasses=ncls)
    x, tr, ev = W.synthetic_batch(b, h, w, ncls, seed, 0.02, gh=8, gw=16)
    masks = W.dropout_masks_for(None, b, seed + 1)
    g = torch.Generator().manual_seed(seed)
    d_prob = torch.randn(b, ncls, h, w, generator=g)
    d_feat = torch.randn(b, 256, h, w, generator=g) * 0.05
    P = {k: v.to(dev).clone() for k, v in st.items()}
    bb = Backbone(P, ncls, dataset)
    out = bb.forward(x.to(dev), True, {k: v.to(dev) for k, v in masks.items()}, True)
    grads = bb.backward(d_prob.permute(0, 2, 3, 1).contiguous().to(dev), d_feat.permute(0, 2, 3, 1).contiguous().to(dev))
    torch.cuda.synchronize()
    o32, _, g32 = run_oracle(st, x, masks, dataset, d_prob, d_feat, torch.float32)
    _, _, g64 = run_oracle(st, x, masks, dataset, d_prob, d_feat, torch.float64)
    assert rel(out["prob"].permute(0, 3, 1, 2).cpu(), o32["pred_2d"].detach()) < 1e-4
    e_hip, e_ora, bad = [], [], 0
    for k, ref in g64.items():
        if k == "projector.proj.0.bias":
            continue
        scale = float(ref.abs().max()) + 1e-30
        eh = float((grads[k].cpu().double() - ref).abs().max()) / scale
        eo = float((g32[k].double() - ref).abs().max()) / scale
        e_hip.append(eh)
        e_ora.append(eo)
        bad += eh > 3 * eo + 1e-4
    tag = f"backbone/{dataset}_{h}x{w}"
    record(f"{tag}/grad_err_vs_f64_median_hip", float(np.median(e_hip)))
    record(f"{tag}/grad_err_vs_f64_median_oracle_fp32", float(np.median(e_ora)))
    record(f"{tag}/tensors_beyond_3x_oracle_noise", bad / len(e_hip))
    assert np.median(e_hip) < 2 * np.median(e_ora) + 1e-5
    assert max(e_hip) < 2 * max(e_ora) + 1e-4
    assert bad <= 0.06 * len(e_hip)
