"""Per-layer backward parity IN CONTEXT: every convolution layer's weight / bias gradient and
every BatchNorm layer's input gradient + parameter gradients of a full SalsaNextProto backward
pass are recomputed in float64 from the HIP pass's OWN activations and output gradients.

The whole-network gradient tests (test_gpu_backbone.py, test_gpu_step.py) compare against an
oracle that ran its own forward: one LeakyReLU pre-activation landing on the other side of zero
changes a derivative from 1 to 0.01 and shows up as percent-level differences that say nothing
about the kernels.  Here both sides see identical activations, so each layer is held to 1e-5 of
max|ref| end to end (measured: 7e-7) -- a wrong wgrad scaling or tap offset in ANY layer fails this test."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import weights as W
from _measure import record

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _nchw64(t):
    return t.detach().double().cpu().permute(0, 3, 1, 2).contiguous()


def _transformed_input(rec, slope):
    xs = []
    for s in rec.srcs:
        v = _nchw64(s.t)
        if s.scale is not None:
            v = v * s.scale.double().cpu()[None, :, None, None] + s.shift.double().cpu()[None, :, None, None]
        if rec.src_lrelu:
            v = F.leaky_relu(v, slope)
        xs.append(v)
    return torch.cat(xs, 1)


def _wgrad64(x, dz, taps, cout):
    """dw[o, c, t] = sum_p dz[p, o] * x[p + tap_t, c] (zero outside the image)."""
    b, cin, h, w = x.shape
    pad = max(max(abs(dy), abs(dx)) for dy, dx in taps)
    xp = F.pad(x, (pad, pad, pad, pad))
    out = torch.empty(cout, cin, len(taps), dtype=torch.float64)
    for t, (dy, dx) in enumerate(taps):
        win = xp[:, :, pad + dy:pad + dy + h, pad + dx:pad + dx + w]
        out[:, :, t] = torch.einsum("bohw,bchw->oc", dz, win)
    return out


@pytest.mark.parametrize("b,h,w,ncls,dataset,seed", [(2, 32, 64, 20, "SemanticKitti", 101), (1, 24, 56, 14, "SemanticPOSS", 201)])
def test_every_layer_gradient_vs_float64_on_the_hip_activations(b, h, w, ncls, dataset, seed):
    from coarse3d_amd.backbone import Backbone
    st = W.closed_form_state(nclasses=ncls)
    x, tr, ev = W.synthetic_batch(b, h, w, ncls, seed, 0.02, gh=8, gw=16)
    masks = W.dropout_masks_for(None, b, seed + 1)
    g = torch.Generator().manual_seed(seed)
    d_prob = torch.randn(b, ncls, h, w, generator=g)
    d_feat = torch.randn(b, 256, h, w, generator=g) * 0.05
    P = {k: v.to(DEV).clone() for k, v in st.items()}
    bb = Backbone(P, ncls, dataset)
    bb.forward(x.to(DEV), True, {k: v.to(DEV) for k, v in masks.items()}, True)
    bb.capture = {}
    cap = bb.capture
    grads = bb.backward(d_prob.permute(0, 2, 3, 1).contiguous().to(DEV), d_feat.permute(0, 2, 3, 1).contiguous().to(DEV))
    torch.cuda.synchronize()
    assert len(cap) == 52                         # every conv but the 5 -> 32 input layer (VALU kernel, test_gpu_ops.py)
    worst = {"wgrad": 0.0, "bias": 0.0, "bn_dz": 0.0, "bn_param": 0.0}

    def rel(a, ref, what, name):
        e = float((a.double().cpu() - ref).abs().max() / (ref.abs().max() + 1e-300))
        worst[what] = max(worst[what], e)
        assert e < 1e-5, (name, what, e)

    for name, (rec, dy, dz) in cap.items():
        slope = rec.slope if rec.slope > 0 else 0.01
        cout = rec.cout
        dz64 = _nchw64(dz)[:, :cout]
        xin = _transformed_input(rec, slope)
        wref = _wgrad64(xin, dz64, rec.taps, cout)
        rel(grads[f"{name}.weight"].reshape(cout, xin.shape[1], -1), wref, "wgrad", name)
        if f"{name}.bias" in grads:
            if rec.mode == 1:
                # a bias directly in front of a BatchNorm is cancelled exactly: sum(dz) = 0 up to the
                # rounding of its summands -- bound the error by their magnitude instead of by ~0
                err = (grads[f"{name}.bias"].double().cpu() - dz64.sum(dim=(0, 2, 3))).abs().max()
                assert float(err / dz64.abs().sum(dim=(0, 2, 3)).max()) < 1e-6, name
            else:
                rel(grads[f"{name}.bias"], dz64.sum(dim=(0, 2, 3)), "bias", name)
        if rec.mode in (0, 1):
            # mode 0: conv -> LeakyReLU -> BatchNorm (SalsaNext blocks): dz = lrelu'(a) * dBN
            # mode 1: conv -> BatchNorm -> LeakyReLU (projector): dy passes the activation first
            a = _nchw64(rec.out.t)[:, :cout]
            gdy = _nchw64(dy)[:, :cout]
            bn = rec.bn
            gamma = P[f"{bn.name}.weight"].double().cpu()
            mean, invstd = bn.mean.double().cpu(), bn.invstd.double().cpu()
            if rec.mode == 1:
                y = a * bn.scale.double().cpu()[None, :, None, None] + bn.shift.double().cpu()[None, :, None, None]
                gdy = gdy * torch.where(y > 0, torch.ones_like(y), torch.full_like(y, slope))
            xhat = (a - mean[None, :, None, None]) * invstd[None, :, None, None]
            n = a.numel() // cout
            dgamma = (gdy * xhat).sum(dim=(0, 2, 3))
            dbeta = gdy.sum(dim=(0, 2, 3))
            da = (gamma * invstd)[None, :, None, None] * (gdy - dbeta[None, :, None, None] / n
                                                          - xhat * dgamma[None, :, None, None] / n)
            if rec.mode == 0:        # a is the post-activation tensor: the derivative is taken at its sign
                da = da * torch.where(a > 0, torch.ones_like(a), torch.full_like(a, slope))
            rel(dz[..., :cout].permute(0, 3, 1, 2), da, "bn_dz", name)
            rel(grads[f"{bn.name}.weight"], dgamma, "bn_param", name)
            rel(grads[f"{bn.name}.bias"], dbeta, "bn_param", name)
    for k, v in worst.items():
        record(f"layer_grads/{dataset}/{k}_max_rel_err", v)
    print(worst)
