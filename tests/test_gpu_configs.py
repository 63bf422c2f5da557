"""BASELINE.json configs as parity cases at their real spatial sizes (forward vs the CPU oracle,
1e-4 of max|ref|; module-level API; default-style initialisation):
  configs[1] SemanticKITTI 64x2048, C=20            (B=2 here: the oracle runs on the CPU)
  configs[3] nuScenes      32x1024, C=17            (2x64 bottleneck -> TR=2 tile path)
  configs[4] SemanticPOSS  40x1800 (+8 pad), C=14   (W=1808: partial 32-wide tiles)
  + nuScenes at the reference YAML's own shape 64x2048, C=17 (config_nuscenes.yaml:132-133)
plus size-independent properties of a full step at the benchmark size."""
import pytest
from _measure import record
import torch

import weights as W
from oracle import coarse3d_oracle as oc

pytestmark = pytest.mark.gpu
_PREV = []          # matrix-engine states to restore: the suite may run under C3D_MATRIX=<engine>
DEV = "cuda"


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("b,h,w,ncls,dataset", [
    (2, 64, 2048, 20, "SemanticKitti"),
    (2, 32, 1024, 17, "nuScenes"),
    (1, 40, 1800, 14, "SemanticPOSS"),
    # the reference's OWN nuScenes shape (tasks/weak_segmentation/config_nuscenes.yaml:132-133: 64 x 2048, 16 classes + ignore;
    # BASELINE configs[3] quotes 32 x 1024 -- SURVEY appendix C, Q11: "ship both")
    (1, 64, 2048, 17, "nuScenes"),
])
def test_forward_parity_at_config_sizes(b, h, w, ncls, dataset):
    from coarse3d_amd.pc_processor.models import SalsaNextProto
    st = oc.init_state(nclasses=ncls, seed=3)
    g = torch.Generator().manual_seed(h + w)
    x = torch.randn(b, 5, h, w, generator=g)
    masks = W.dropout_masks_for(None, b, 9)
    m = SalsaNextProto(5, ncls, 20, 0, use_prototype=False, dataset=dataset)
    m.load_state_dict(st)
    m.to(DEV).train()
    m.dropout_masks = {k: v.to(DEV) for k, v in masks.items()}
    with torch.no_grad():
        out = m(x.to(DEV), return_feat=True)
        ref = oc.backbone_forward({k: v.clone() for k, v in st.items()}, x, True, masks, True, dataset)
    assert out["pred_2d"].shape == ref["pred_2d"].shape
    assert rel(out["pred_2d"], ref["pred_2d"]) < 1e-4
    assert rel(out["feat_2d"], ref["feat_2d"]) < 1e-4
    # probabilities sum to one, embedding rows have (interpolated) norm <= 1
    assert float((out["pred_2d"].sum(1) - 1).abs().max()) < 1e-5
    assert float(out["feat_2d"].norm(dim=1).max()) < 1 + 1e-4


def test_full_step_properties_at_benchmark_size():
    """bs=8 64x2048 step (the bench workload): finite losses, every trainable tensor receives a
    finite gradient, the bank stays l2-normalised, anchors respect their class, determinism."""
    from coarse3d_amd import contrast
    from coarse3d_amd.pc_processor.models import SalsaNextProto
    from coarse3d_amd.trainer import TrainStep
    import bench
    b, h, w, ncls = 8, 64, 2048, 20
    torch.manual_seed(1)
    m = SalsaNextProto(5, ncls, 20, 0, use_prototype=True).to(DEV).train()
    ts = TrainStep(m, ncls, proto_loss=True, lr=1e-3, num_anchor=512, feature_mean=bench.FEATURE_MEAN, feature_std=bench.FEATURE_STD)
    x, tr, ev = bench.synth_batch(b, h, w, ncls, 1000, DEV)
    res = ts.step(x, tr, ev, epoch=10)
    for k in ("ce", "lov", "contrast", "loss"):
        assert torch.isfinite(res[k]).all(), k
    for k, p in m.named_parameters():
        if p.requires_grad and not k.startswith(("feat_norm", "mask_norm")):   # unused in the reference too
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
    assert float((m.prototypes.norm(dim=-1) - 1).abs().max()) < 1e-4
    # pseudo labels: only on evaluated pixels, weak labels preserved, roughly ratio * |class mask|
    lab, mask = res["labels_contra"], res["mask_contra"]
    assert bool((lab[ev == 0] == 0).all())
    assert bool((lab[tr > 0] == tr[tr > 0]).all())
    frac = float((lab > 0).float().mean())
    assert 0.01 < frac < 0.2, frac
    # anchor sampler on the same inputs twice -> identical indices (no atomics on the index path)
    prob = torch.softmax(torch.randn(2, ncls, 64, 256, device=DEV), 1)
    feats = torch.randn(2, 256, 64, 256, device=DEV)
    labels = torch.randint(0, ncls, (2, 64, 256), device=DEV)
    u = torch.rand(2 * ncls, 512, dtype=torch.float64)
    perms = torch.stack([torch.randperm(20) for _ in range(ncls - 1)])
    l1, d1 = contrast.contrast_mem_loss(feats, prob, labels, None, m.prototypes.detach(), 0.07, 0.07, 512, 0, u, perms,
                                        return_debug=True)
    l2, d2 = contrast.contrast_mem_loss(feats, prob, labels, None, m.prototypes.detach(), 0.07, 0.07, 512, 0, u, perms,
                                        return_debug=True)
    assert torch.equal(d1["idx"], d2["idx"]) and float(l1) == float(l2)
    T = int(d1["T"])
    img, cls, idx = d1["img"][:T].long(), d1["cls"][:T].long(), d1["idx"][:T].long()
    picked = labels.reshape(2, -1)[img[:, None], idx]
    assert bool((picked == cls[:, None]).all())


@pytest.mark.parametrize("b,h,w,ncls,dataset,rate", [
    (16, 32, 1024, 17, "nuScenes", 1e-3),          # BASELINE configs[3]: bs=16/GPU, small-H path
    (8, 40, 1800, 14, "SemanticPOSS", 1e-4),       # BASELINE configs[4]: 0.01 % weak labels, entropy selection on
])
def test_full_step_properties_at_config_batch_sizes(b, h, w, ncls, dataset, rate):
    """Full training steps of configs[3] / configs[4] at their per-GPU batch sizes: the properties
    that do not depend on the size (finite losses and gradients for every trainable tensor, unit
    prototype bank, pseudo labels only where allowed, weak labels preserved, anchors of their own
    class) + the forward of image 0 of the batch against the CPU oracle run on that batch's own
    statistics is covered at these shapes by test_forward_parity_at_config_sizes."""
    from coarse3d_amd.pc_processor.models import SalsaNextProto
    from coarse3d_amd.trainer import TrainStep
    import bench
    torch.manual_seed(1)
    m = SalsaNextProto(5, ncls, 20, 0, use_prototype=True, dataset=dataset).to(DEV).train()
    ts = TrainStep(m, ncls, proto_loss=True, lr=1e-3, num_anchor=512, feature_mean=bench.FEATURE_MEAN,
                   feature_std=bench.FEATURE_STD, entropy_selection=True)
    losses = []
    for s in range(2):
        x, tr, ev = bench.synth_batch(b, h, w, ncls, 1000 + s, DEV, rate)
        res = ts.step(x, tr, ev, epoch=10)
        for k in ("ce", "lov", "contrast", "loss"):
            assert torch.isfinite(res[k]).all(), (k, s)
        losses.append(float(res["loss"]))
        assert res["pred_2d"].shape == (b, ncls, h, w)
        assert float((res["pred_2d"].sum(1) - 1).abs().max()) < 1e-5
        lab = res["labels_contra"]
        assert bool((lab[ev == 0] == 0).all()) and bool((lab[tr > 0] == tr[tr > 0]).all())
        assert int((lab > 0).sum()) >= int((tr > 0).sum())
    for k, p in m.named_parameters():
        if p.requires_grad and not k.startswith(("feat_norm", "mask_norm")):
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
    assert float((m.prototypes.norm(dim=-1) - 1).abs().max()) < 1e-4
    assert int(m.state_dict()["resBlock1.bn1.num_batches_tracked"]) == 2


def test_bf16_matrix_mode_at_config2_size():
    """BASELINE configs[2] at its real per-GPU size (8 x 64x2048, bf16 matrix operands): the step
    runs, stays finite and tracks the fp32 step of the same seed (probabilities, loss terms)."""
    from coarse3d_amd import ops
    from coarse3d_amd.pc_processor.models import SalsaNextProto
    from coarse3d_amd.trainer import TrainStep
    import bench
    b, h, w, ncls = 8, 64, 2048, 20
    outs = {}
    for kind in ("f32", "bf16"):
        _PREV.append(ops.matrix_precision_state())
        ops.set_matrix_precision(kind)
        try:
            torch.manual_seed(1)
            m = SalsaNextProto(5, ncls, 20, 0, use_prototype=True).to(DEV).train()
            m.dropout_masks = {k: v.to(DEV) for k, v in W.dropout_masks_for(None, b, 11).items()}
            ts = TrainStep(m, ncls, proto_loss=True, lr=1e-3, num_anchor=512, feature_mean=bench.FEATURE_MEAN,
                           feature_std=bench.FEATURE_STD)
            x, tr, ev = bench.synth_batch(b, h, w, ncls, 1000, DEV)
            torch.manual_seed(7)
            res = ts.step(x, tr, ev, epoch=10)
            outs[kind] = {k: res[k].detach().clone() for k in ("pred_2d", "ce", "lov", "contrast", "loss")}
            del m, ts, res
            torch.cuda.empty_cache()
        finally:
            ops.set_matrix_precision(*_PREV.pop())
    for k in ("ce", "lov", "contrast", "loss"):
        assert torch.isfinite(outs["bf16"][k]).all()
    dp = (outs["f32"]["pred_2d"] - outs["bf16"]["pred_2d"]).abs()
    assert float(dp.mean()) < 1e-2 and float(dp.max()) < 0.3, (float(dp.mean()), float(dp.max()))
    for k in ("ce", "lov"):
        assert abs(float(outs["bf16"][k]) - float(outs["f32"][k])) < 3e-2 * abs(float(outs["f32"][k])), k


def test_bf16_matrix_mode_tracks_fp32():
    """BASELINE configs[2] (bf16): opt-in mode with bf16 MFMA operands and fp32 accumulate /
    storage.  Not a parity path (the reference has no bf16 mode): the check is that one training
    step tracks the fp32 step within bf16-level tolerances -- class probabilities within 1e-2
    mean / 0.3 max absolute (bf16 operand ulp 4e-3 through ~50 layers), each loss term within 3 %, gradient
    directions as described below."""
    from coarse3d_amd import ops
    from coarse3d_amd.pc_processor.models import SalsaNextProto
    from coarse3d_amd.trainer import TrainStep
    import bench
    b, h, w, ncls = 2, 64, 512, 20
    outs = {}
    for kind in ("f32", "bf16"):
        _PREV.append(ops.matrix_precision_state())
        ops.set_matrix_precision(kind)
        try:
            torch.manual_seed(1)
            m = SalsaNextProto(5, ncls, 20, 0, use_prototype=True).to(DEV).train()
            m.dropout_masks = {k: v.to(DEV) for k, v in W.dropout_masks_for(None, b, 11).items()}
            ts = TrainStep(m, ncls, proto_loss=True, lr=1e-3, num_anchor=64, feature_mean=bench.FEATURE_MEAN,
                           feature_std=bench.FEATURE_STD, loss_w_contrast=0.1)
            x, tr, ev = bench.synth_batch(b, h, w, ncls, 1000, DEV, label_rate=2e-2)
            torch.manual_seed(7)
            res = ts.step(x, tr, ev, epoch=0)
            outs[kind] = (res, {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
        finally:
            ops.set_matrix_precision(*_PREV.pop())
    r32, g32 = outs["f32"]
    r16, g16 = outs["bf16"]
    dp = (r32["pred_2d"] - r16["pred_2d"]).abs()
    assert float(dp.mean()) < 1e-2 and float(dp.max()) < 0.3, (float(dp.mean()), float(dp.max()))
    for k in ("ce", "lov"):
        assert abs(float(r16[k].detach()) - float(r32[k].detach())) < 3e-2 * abs(float(r32[k].detach())), k
    # gradient directions: the heads (one or two layers from the loss) must agree closely; deep
    # into the backbone the direction decorrelates progressively -- this network amplifies even
    # fp32 rounding noise to ~1 % of the gradient (tests/test_gpu_backbone.py), and bf16 operand
    # noise is 4 orders of magnitude larger -- so there only a clearly positive correlation is
    # required.
    cos = {}
    for k, a in g32.items():
        if a.numel() >= 4096 and float(a.norm()) > 0:
            cos[k] = float((a * g16[k]).sum() / (a.norm() * g16[k].norm() + 1e-30))
    print({k: round(v, 3) for k, v in cos.items()})
    for k, v in cos.items():
        # (heads: 0.9797-0.985 measured across kernel schedules that differ in fp32 accumulation order only -- the statistic
        #  itself moves by 5e-4 when the order of the four-tap convs' k steps changes)
        assert v > (0.97 if k.startswith(("cls_head", "projector.proj.3")) else 0.3), (k, v)


def test_parity_suite_passes_on_the_strict_fp32_engine():
    """The library default is the "bf16x3" engine (every fp32 operand split exactly into three bf16 planes, six or
    eight of nine plane products accumulated in fp32): every test of this suite runs on it and shows up in the
    driver's pass count.  This guard re-runs the ENGINE-DEPENDENT part of the suite -- `-m "gpu and parity"`: every
    oracle / golden / float64 comparison at its unchanged tolerance, the anchor and pseudo-label indices, the per-layer
    float64 gradient check, all three backbones -- on the strict fp32-MFMA engine (C3D_MATRIX=f32,
    v_mfma_f32_32x32x2_f32 == an fmaf chain), so that both engines stay parity-tested.  Round 6:
    it used to re-run EVERYTHING, i.e. also the multi-process exchange tests, bench.py subprocesses, soaks and integer paths
    that never see a matrix engine -- 447 s of a 1 200 s driver limit; tests/conftest.py::_NOT_ENGINE_PARITY lists what stays out."""
    import os
    import subprocess
    import sys
    if os.environ.get("C3D_MATRIX"):
        pytest.skip("already inside an engine-pinned run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, C3D_MATRIX="f32", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests"), "-m", "gpu and parity", "-q", "-x", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=root)
    tail = (r.stdout or "")[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in tail and " failed" not in tail, tail
    import re
    m = re.search(r"(\d+) passed", tail)
    record("suite/strict_fp32_engine_tests_passed", int(m.group(1)))
