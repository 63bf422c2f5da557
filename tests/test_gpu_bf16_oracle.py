"""BASELINE configs[2] (the bf16 engine: bf16 activations in HBM, bf16 MFMA operands, fp32 accumulation / statistics /
master weights) against the REFERENCE side, not against the library's own fp32 engine:

* the forward pass against the fp32 CPU oracle (oracle/coarse3d_oracle.py, pinned to the reference by
  tests/test_oracle_golden.py) at 2 x 64 x 512 with default-style initialisation;
* one whole training step against the second reference-generated golden step (tests/golden/step2.npz, produced by the
  reference modules themselves: tests/golden/make_golden_round4.py) with the recorded randomness injected.

The reference has no bf16 mode: the contract is "the bf16 step tracks the fp32 reference within a STATED bf16
bound".  Every figure is recorded (gpurun_out/parity_measured_bf16_vs_oracle.json, committed as
profiles/round5_parity_measured_bf16.json) and every bound below is <= 2x what was measured on MI355X."""
import json
import os

import numpy as np
import pytest
import torch

import weights as W
from oracle import coarse3d_oracle as oc

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "parity_measured_bf16_vs_oracle.json")


def rec(name, value):
    try:
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        data = json.load(open(OUT)) if os.path.exists(OUT) else {}
        data[name] = value
        json.dump(data, open(OUT, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    return value


class bf16_engine:
    def __enter__(self):
        from coarse3d_amd import ops
        self.prev = ops.matrix_precision_state()
        ops.set_matrix_precision("bf16")            # bf16 storage is this mode's default
        return ops

    def __exit__(self, *a):
        from coarse3d_amd import ops
        ops.set_matrix_precision(*self.prev)


# bound = <= 2x the value measured on MI355X (profiles/round5_parity_measured_bf16.json; measured value in the comment).
# The *_over_yardstick entries are the contract in one number each: the engine's distance from the fp32 oracle divided by the
# distance of the SAME oracle with bf16-rounded convolution operands -- measured 0.99 / 1.11 / 0.99: the HIP bf16 engine is
# as far from the fp32 reference as the reference is from itself under bf16 operands.
FWD_BOUNDS = {"pred_abs_max": 0.26,                    # 0.130 (yardstick 0.117)
              "pred_abs_mean": 8.7e-3,                  # 4.34e-3 (yardstick 4.38e-3)
              "argmax_disagree": 0.40,                  # 0.198 of the pixels of an UNTRAINED net (yardstick 0.201)
              "argmax_disagree_margin_ge_0.05": 1.1e-2,  # 5.2e-3 where the oracle's two best classes are >= 0.05 apart
              "feat_cos_min": 0.974,                    # 0.98685 (lower bound)
              "feat_cos_mean_gap": 9.4e-3,              # 4.66e-3 (yardstick 4.71e-3)
              "running_stat_rel_max": 8.5e-2,           # 4.24e-2
              "pred_abs_mean_over_yardstick": 1.5,      # 0.993
              "pred_abs_max_over_yardstick": 2.0,       # 1.113
              "argmax_disagree_over_yardstick": 1.5}    # 0.987


def test_bf16_forward_against_the_fp32_cpu_oracle():
    from coarse3d_amd.pc_processor.models import SalsaNextProto
    b, h, w, ncls = 2, 64, 512, 20
    st = oc.init_state(nclasses=ncls, seed=3)
    g = torch.Generator().manual_seed(h + w)
    x = torch.randn(b, 5, h, w, generator=g)
    masks = W.dropout_masks_for(None, b, 9)
    ref_st = {k: v.clone() for k, v in st.items()}
    with torch.no_grad():
        ref = oc.backbone_forward(ref_st, x, True, masks, True, "SemanticKitti")
        # Yardstick from the reference side: the SAME fp32 oracle with the operands of every convolution rounded to bf16
        # (products exact, fp32 accumulation -- what bf16 MFMA operands do, and what torch.autocast(bfloat16) would do to
        # the reference's convs).  Its distance from the fp32 oracle is the noise bf16 operands cause in THIS network on
        # THESE weights (an untrained net amplifies rounding noise ~170x: fp32's 6e-8 arrives as 1e-5 at the output);
        # the HIP engine must stay within a stated multiple of it.
        orig_conv = oc._Ctx.conv

        def conv_bf16(self, name, t, dilation=1, padding=0):
            return torch.nn.functional.conv2d(t.bfloat16().float(), self.p[f"{name}.weight"].bfloat16().float(),
                                              self.p[f"{name}.bias"], stride=1, padding=padding, dilation=dilation)
        oc._Ctx.conv = conv_bf16
        try:
            yard = oc.backbone_forward({k: v.clone() for k, v in st.items()}, x, True, masks, True, "SemanticKitti")
        finally:
            oc._Ctx.conv = orig_conv
    with bf16_engine():
        m = SalsaNextProto(5, ncls, 20, 0, use_prototype=False)
        m.load_state_dict(st)
        m.to(DEV).train()
        m.dropout_masks = {k: v.to(DEV) for k, v in masks.items()}
        with torch.no_grad():
            out = m(x.to(DEV), return_feat=True)
        pred, feat = out["pred_2d"].float().cpu(), out["feat_2d"].float().cpu()
        sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    assert pred.shape == ref["pred_2d"].shape and feat.shape == ref["feat_2d"].shape
    d = (pred - ref["pred_2d"]).abs()
    got = {
        "pred_abs_max": float(d.max()),
        "pred_abs_mean": float(d.mean()),
        "argmax_disagree": float((pred.argmax(1) != ref["pred_2d"].argmax(1)).float().mean()),
    }
    # (default-style initialisation: the class probabilities of an untrained net are nearly tied, 0.03 ... 0.08 each --
    #  an argmax flips wherever the two best classes are closer than the bf16 noise.  Where the oracle's own margin between
    #  its two best classes exceeds that noise, the decisions must agree.)
    top2 = ref["pred_2d"].topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    flip = pred.argmax(1) != ref["pred_2d"].argmax(1)
    got["flip_margin_max"] = float(margin[flip].max()) if bool(flip.any()) else 0.0
    for thr in (0.02, 0.05):
        sel = margin >= thr
        got[f"argmax_disagree_margin_ge_{thr}"] = float(flip[sel].float().mean()) if bool(sel.any()) else 0.0
        got[f"pixels_margin_ge_{thr}"] = float(sel.float().mean())
    dy = (yard["pred_2d"] - ref["pred_2d"]).abs()
    got["yardstick_pred_abs_max"] = float(dy.max())
    got["yardstick_pred_abs_mean"] = float(dy.mean())
    got["yardstick_argmax_disagree"] = float((yard["pred_2d"].argmax(1) != ref["pred_2d"].argmax(1)).float().mean())
    got["yardstick_feat_cos_mean_gap"] = float(1.0 - torch.nn.functional.cosine_similarity(yard["feat_2d"], ref["feat_2d"], dim=1).mean())
    got["pred_abs_mean_over_yardstick"] = got["pred_abs_mean"] / got["yardstick_pred_abs_mean"]
    got["pred_abs_max_over_yardstick"] = got["pred_abs_max"] / got["yardstick_pred_abs_max"]
    got["argmax_disagree_over_yardstick"] = got["argmax_disagree"] / max(got["yardstick_argmax_disagree"], 1e-9)
    cos = torch.nn.functional.cosine_similarity(feat, ref["feat_2d"], dim=1)
    got["feat_cos_min"] = float(cos.min())
    got["feat_cos_mean_gap"] = float(1.0 - cos.mean())
    # the BatchNorm running statistics the training forward updated (fp32 sums of fp32 accumulator values on the device)
    worst = 0.0
    for k, v in ref_st.items():
        if k.endswith(("running_mean", "running_var")):
            worst = max(worst, float((sd[k] - v).abs().max() / (v.abs().max() + 1e-12)))
    got["running_stat_rel_max"] = worst
    for k, v in got.items():
        rec(f"forward_2x64x512/{k}", v)
    assert float((pred.sum(1) - 1).abs().max()) < 1e-5
    for k, bound in FWD_BOUNDS.items():
        if bound is None:
            continue
        if k == "feat_cos_min":
            assert got[k] >= bound, (k, got[k], bound)
        else:
            assert got[k] <= bound, (k, got[k], bound)


# second reference-generated golden step (2 x 64 x 512, 512 anchors, epoch 40): bound = <= 2x measured
STEP_BOUNDS = {"ce_rel": 7.1e-3,                        # 3.53e-3
               "lov_rel": 9.4e-4,                       # 4.68e-4
               "contrast_rel": 1.0e-4,                  # 4.5e-5 (a mean over 19 456 anchors: 67 % of the draws land on other
                                                        #          pixels of their class, the loss barely moves)
               "loss_rel": 5.0e-3,                      # 2.46e-3
               "pred_abs_max": 0.25,                    # 0.1235 on the stored sub-grid
               "argmax_disagree_sub": 0.27,             # 0.136 (closed-form weights: near-tied classes)
               "labels_contra_disagree": 0.11,          # 0.0553: pseudo labels follow the argmax
               "mask_contra_disagree": 0.10,            # 0.0493
               "prototypes_rel": 1.7e-3,                # 8.2e-4
               "grad_norm_rel_median": 7.3e-2,          # 3.6e-2 over the 190 parameter tensors
               "grad_norm_rel_max": 0.96}               # 0.48 (one small tensor)


def test_bf16_training_step_against_the_reference_generated_golden_step():
    import test_gpu_step as S
    with bf16_engine():
        g, m, ts, (x, tr, ev), (b, h, w, ncls, A) = S.step2_setup()
        ts.sparse_proto = True
        res = ts.step(x.to(DEV), tr.to(DEV), ev.to(DEV), epoch=40)
        torch.cuda.synchronize()
        got = {k + "_rel": S.rel(res[k], g[k]) for k in ("ce", "lov", "contrast", "loss")}
        got["pred_abs_max"] = float((res["pred_2d"][:, :, ::4, ::16].float().cpu() - g["pred_sub"]).abs().max())
        ps = res["pred_2d"][:, :, ::4, ::16].float().cpu()
        top2 = g["pred_sub"].topk(2, dim=1).values
        margin = top2[:, 0] - top2[:, 1]
        flip = ps.argmax(1) != g["pred_sub"].argmax(1)
        got["argmax_disagree_sub"] = float(flip.float().mean())
        got["flip_margin_max_sub"] = float(margin[flip].max()) if bool(flip.any()) else 0.0
        got["labels_contra_disagree"] = float((res["labels_contra"].cpu() != g["labels_contra"].long()).float().mean())
        got["mask_contra_disagree"] = float((res["mask_contra"].cpu() != g["mask_contra"]).float().mean())
        got["prototypes_rel"] = S.rel(m.prototypes, g["new_prototypes"])
        dbg = ts.contrast.last_debug
        T = g["anchor_idx"].shape[0]
        got["anchor_pairs_equal"] = int(int(dbg["T"]) == T)
        if int(dbg["T"]) == T:
            got["anchor_disagree"] = float((dbg["idx"][:T].cpu().long() != g["anchor_idx"].long()).float().mean())
        errs, cos_heads = [], []
        for k, p in m.named_parameters():
            if f"gnorm/{k}" in g and k != "projector.proj.0.bias" and p.grad is not None:
                errs.append(abs(float(p.grad.norm()) - float(g[f"gnorm/{k}"])) / (float(g[f"gnorm/{k}"]) + 1e-9))
        got["grad_norm_rel_median"] = float(np.median(errs))
        got["grad_norm_rel_max"] = float(max(errs))
    for k, v in got.items():
        rec(f"step2/{k}", v)
    for k in ("ce", "lov", "contrast", "loss"):
        assert torch.isfinite(res[k]).all()
    assert got["anchor_pairs_equal"] == 1            # the same (image, class) pairs reach the sampler
    for k, bound in STEP_BOUNDS.items():
        if bound is not None and k in got:
            assert got[k] <= bound, (k, got[k], bound)
