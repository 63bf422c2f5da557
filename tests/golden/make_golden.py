#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz from the REAL reference.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py
The reference is imported through tests/golden/_ref_shim.py; its randomness is either
injected (Dropout2d masks) or recorded at the call site (Exp(1) noise of gumbel_softmax and
of multinomial-without-replacement, the float64 uniform stream of multinomial-with-
replacement, randperm) so that the oracle / the HIP path can be replayed on the same draws.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _ref_shim import load_reference  # noqa: E402
import weights as W  # noqa: E402

torch.set_num_threads(8)
R = load_reference()


# ----------------------------------------------------------------------------- recorders
class Recorder:
    """Wraps the RNG entry points the reference uses and records what they drew."""

    def __init__(self):
        self.exp = []         # every Tensor.exponential_ result (gumbel noise)
        self.multi_rep = []   # (weights, uniforms f64, indices)
        self.multi_norep = [] # (weights, k, exp noise, sorted indices)
        self.perms = []
        self._orig = {}

    def __enter__(self):
        rec = self
        self._orig = dict(exp=torch.Tensor.exponential_, multi=torch.multinomial,
                          perm=torch.randperm)
        o_exp, o_multi, o_perm = self._orig["exp"], self._orig["multi"], self._orig["perm"]

        def exponential_(t, *a, **k):
            out = o_exp(t, *a, **k)
            rec.exp.append(out.detach().clone())
            return out

        def multinomial(w, n, replacement=False, **k):
            gen_state = torch.random.get_rng_state()
            torch.Tensor.exponential_ = o_exp          # do not double-record
            idx = o_multi(w, n, replacement=replacement, **k)
            after = torch.random.get_rng_state()
            torch.random.set_rng_state(gen_state)
            if replacement:
                u = torch.rand(n, dtype=torch.float64)
                assert torch.equal(torch.random.get_rng_state(), after), "stream mismatch"
                rec.multi_rep.append((w.detach().clone(), u, idx.clone()))
            else:
                q = o_exp(torch.empty_like(w), 1)
                assert torch.equal(torch.random.get_rng_state(), after), "stream mismatch"
                rec.multi_norep.append((w.detach().clone(), n, q, torch.sort(idx)[0].clone()))
            torch.random.set_rng_state(after)
            torch.Tensor.exponential_ = exponential_
            return idx

        def randperm(n, *a, **k):
            p = o_perm(n, *a, **k)
            rec.perms.append(p.clone())
            return p

        torch.Tensor.exponential_ = exponential_
        torch.multinomial = multinomial
        torch.randperm = randperm
        return self

    def __exit__(self, *a):
        torch.Tensor.exponential_ = self._orig["exp"]
        torch.multinomial = self._orig["multi"]
        torch.randperm = self._orig["perm"]


def install_dropout_masks(model, masks):
    """Replace every Dropout2d.forward by an injected [B,C] multiplier."""
    for name, mod in model.named_modules():
        if isinstance(mod, torch.nn.Dropout2d):
            if name in masks:
                m = masks[name]
                mod.forward = (lambda x, m=m: x * m[:, :, None, None])
            else:
                mod.forward = (lambda x: x)


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path)/1e6:.2f} MB, keys={len(out)}")


# ----------------------------------------------------------------------------- blocks
def gold_blocks():
    out = {}
    g = np.random.Generator(np.random.PCG64(11))
    b, h, w = 2, 16, 64
    # ResContextBlock 5 -> 8
    x = torch.from_numpy(g.standard_normal((b, 5, h, w)).astype(np.float32))
    st = W.block_state("ctx", 5, 8)
    blk = R.ResContextBlock(5, 8)
    blk.load_state_dict({k[4:]: v for k, v in st.items()})
    blk.train()
    out["ctx_x"], out["ctx_y"] = x, blk(x)
    # ResBlock 8 -> 16, pooling + dropout
    x = torch.from_numpy(g.standard_normal((b, 8, h, w)).astype(np.float32))
    st = W.block_state("res", 8, 16)
    mask = torch.from_numpy((g.random((b, 16)) >= 0.2).astype(np.float32) * 1.25)
    blk = R.ResBlock(8, 16, 0.2, pooling=True, drop_out=True)
    blk.load_state_dict({k[4:]: v for k, v in st.items()})
    blk.train()
    blk.dropout.forward = lambda t: t * mask[:, :, None, None]
    yb, ya = blk(x)
    out.update(res_x=x, res_mask=mask, res_pooled=yb, res_skip=ya)
    # ResBlock 8 -> 16, no pooling
    blk = R.ResBlock(8, 16, 0.2, pooling=False, drop_out=True)
    blk.load_state_dict({k[4:]: v for k, v in st.items()})
    blk.train()
    blk.dropout.forward = lambda t: t * mask[:, :, None, None]
    out["res_nopool"] = blk(x)
    # UpBlock 32 -> 8 (in: 32 ch at h/2, skip: 16 ch at h)
    xin = torch.from_numpy(g.standard_normal((b, 32, h // 2, w // 2)).astype(np.float32))
    skip = torch.from_numpy(g.standard_normal((b, 16, h, w)).astype(np.float32))
    st = W.block_state("up", 32, 8)
    m1 = torch.from_numpy((g.random((b, 8)) >= 0.2).astype(np.float32) * 1.25)
    m2 = torch.from_numpy((g.random((b, 24)) >= 0.2).astype(np.float32) * 1.25)
    m3 = torch.from_numpy((g.random((b, 8)) >= 0.2).astype(np.float32) * 1.25)
    blk = R.UpBlock(32, 8, 0.2, drop_out=True)
    blk.load_state_dict({k[4:]: v for k, v in st.items()})
    blk.train()
    blk.dropout1.forward = lambda t: t * m1[:, :, None, None]
    blk.dropout2.forward = lambda t: t * m2[:, :, None, None]
    blk.dropout3.forward = lambda t: t * m3[:, :, None, None]
    out.update(up_x=xin, up_skip=skip, up_m1=m1, up_m2=m2, up_m3=m3, up_y=blk(xin, skip))
    npz("blocks.npz", **out)


# ----------------------------------------------------------------------------- full model
def build_ref_model(state, dataset="SemanticKitti", ncls=20):
    m = R.SalsaNextProto(5, ncls, 20, 0, use_prototype=True, dataset=dataset)
    sd = {k: v.clone() for k, v in state.items()}
    m.load_state_dict(sd)
    return m


def gold_model(tag, b, h, w, ncls, dataset, seed, label_rate):
    st = W.closed_form_state(nclasses=ncls)
    x, tr, ev = W.synthetic_batch(b, h, w, ncls, seed, label_rate, gh=8, gw=16)
    masks = W.dropout_masks_for(None, b, seed + 1)
    m = build_ref_model(st, dataset, ncls)
    m.train()
    install_dropout_masks(m, masks)
    stats = {}

    def hook(name):
        def f(mod, inp, out):
            t = inp[0]
            stats[name] = (t.mean(dim=(0, 2, 3)).detach(), t.var(dim=(0, 2, 3), unbiased=False).detach())
        return f
    for name, mod in m.named_modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.register_forward_hook(hook(name))
    torch.manual_seed(seed + 2)
    with Recorder() as rec:
        out = m(x, label=tr, eval_mask=tr > 0, return_feat=True, proto_loss=True)
    sd = m.state_dict()
    arrs = dict(pred_2d=out["pred_2d"], feat_2d_sub=out["feat_2d"][:, :, ::2, ::4],
                contrast_logits_sub=out["contrast_logits"][::16],
                contrast_target=out["contrast_target"], new_prototypes=sd["prototypes"])
    # gumbel noise per class in execution order (classes present among labelled pixels)
    present = [c for c in range(1, ncls) if int((tr == c).sum()) > 0]
    assert len(present) == len(rec.exp), (len(present), len(rec.exp))
    for c, e in zip(present, rec.exp):
        arrs[f"gumbel_{c}"] = e
    for name, (mu, var) in stats.items():
        arrs[f"bnmean/{name}"] = mu
        arrs[f"bnvar/{name}"] = var
    for k, v in sd.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            arrs[f"run/{k}"] = v
    npz(f"model_{tag}.npz", **arrs)
    return st, x, tr, ev, masks, m, rec


# ----------------------------------------------------------------------------- losses
def gold_contrast():
    g = np.random.Generator(np.random.PCG64(21))
    b, ncls, h, w, d, m_ = 2, 6, 16, 32, 256, 20
    feats = torch.from_numpy(g.standard_normal((b, d, h, w)).astype(np.float32)).requires_grad_(True)
    logits = torch.from_numpy(g.standard_normal((b, ncls, h, w)).astype(np.float32) * 2)
    prob = torch.softmax(logits, 1)
    labels = torch.from_numpy(g.integers(0, ncls, (b, h, w)))
    labels[1][labels[1] == 3] = 0          # class 3 absent in image 1 (ragged case)
    keep = torch.from_numpy(g.random((b, h, w)) < 0.7)
    queue = torch.from_numpy(g.standard_normal((ncls, m_, d)).astype(np.float32))
    crit = R.ContrastMEMLoss(ignore_label=0, temperature=0.07, num_anchor=64)
    torch.manual_seed(5)
    with Recorder() as rec:
        loss = crit(feats=feats, output=prob, labels=labels, keep_mask=keep,
                    proto_queue=queue.unsqueeze(0))
    loss.backward()
    arrs = dict(feats=feats.detach(), prob=prob, labels=labels, keep=keep, queue=queue,
                loss=loss.detach(), grad_feats=feats.grad,
                uniforms=torch.stack([u for _, u, _ in rec.multi_rep]),
                indices=torch.stack([i for _, _, i in rec.multi_rep]),
                perms=torch.stack(rec.perms))
    npz("contrast.npz", **arrs)


def gold_pl_select():
    g = np.random.Generator(np.random.PCG64(31))
    b, ncls, h, w = 2, 6, 16, 64
    logits = torch.from_numpy(g.standard_normal((b, ncls, h, w)).astype(np.float32) * 1.5)
    prob = torch.softmax(logits, 1)
    ev = torch.from_numpy(g.integers(0, ncls, (b, h, w)))
    tr = ev * torch.from_numpy(g.random((b, h, w)) < 0.02)
    tr[1][tr[1] == 2] = 0
    fake_self = types.SimpleNamespace(settings=types.SimpleNamespace(ignore_cls=0, n_classes=ncls))
    ratio = float(np.log(1 + (1 + 10) / 100) / np.log(2) * 0.5)
    torch.manual_seed(9)
    with Recorder() as rec:
        lab, mask = R.entropy_based_selection(fake_self, output=prob, wss_mask=tr > 0,
                                              eval_mask=ev > 0, train_label=tr,
                                              select_ratio=ratio)
    arrs = dict(prob=prob, eval_label=ev, train_label=tr, ratio=np.float64(ratio),
                labels=lab, mask=mask,
                noise=torch.stack([q for _, _, q, _ in rec.multi_norep]),
                ks=np.array([k for _, k, _, _ in rec.multi_norep]))
    npz("pl_select.npz", **arrs)


def gold_losses():
    g = np.random.Generator(np.random.PCG64(41))
    b, ncls, h, w = 2, 6, 8, 32
    prob = torch.softmax(torch.from_numpy(g.standard_normal((b, ncls, h, w)).astype(np.float32)), 1)
    prob.requires_grad_(True)
    tr = torch.from_numpy(g.integers(0, ncls, (b, h, w))) * torch.from_numpy(g.random((b, h, w)) < 0.1)
    alpha = torch.from_numpy(g.uniform(0.2, 1, ncls).astype(np.float32))
    alpha[0] = 0
    focal = R.FocalSoftmaxLoss(ncls, gamma=2, alpha=alpha.numpy(), softmax=False)
    lov = R.Lovasz_softmax(ignore=0, per_image=False, softmax=False)
    lf = focal(prob, tr, mask=tr > 0)
    ll = lov(prob, tr)
    gf, = torch.autograd.grad(lf, prob, retain_graph=True)
    gl, = torch.autograd.grad(ll, prob)
    npz("losses.npz", prob=prob.detach(), train_label=tr, alpha=alpha, focal=lf.detach(),
        lovasz=ll.detach(), grad_focal=gf, grad_lovasz=gl)


def gold_metrics():
    """Per-iteration metrics: argmax + un-projection exactly as trainer.py:713-726 (both dataset
    conventions), accumulated and summarised by the reference IOUEval (iou_eval.py:35-119)."""
    g = np.random.Generator(np.random.PCG64(77))
    arrs = {}
    for tag, ncls, h, w, npts, poss in (("kitti", 20, 16, 64, 1500, False), ("poss", 14, 8, 36, 8 * 36, True)):
        ev = R.IOUEval(ncls, ignore=[0])
        scans = 3
        pred = torch.softmax(torch.from_numpy(g.standard_normal((scans, ncls, h, w)).astype(np.float32)) * 3, 1)
        pred[0, :, 0, 0] = 0.05                       # an exact tie: argmax must return the first class
        argmax_2d = pred.argmax(dim=1)
        arrs[f"{tag}/pred_2d"] = pred
        for ii in range(scans):
            if not poss:                              # trainer.py:718-719
                uy = torch.from_numpy(g.integers(0, h, npts))
                ux = torch.from_numpy(g.integers(0, w, npts))
                labels = torch.from_numpy(g.integers(0, ncls, npts))
                unproj = argmax_2d[ii, uy, ux]
                arrs[f"{tag}/ux{ii}"] = ux
            else:                                     # trainer.py:720-726 (40*1800 -> h*w here)
                nvalid = int(g.integers(npts // 2, npts))
                uy = torch.from_numpy(g.integers(0, h * w, nvalid))
                labels = torch.from_numpy(g.integers(0, ncls, npts))
                temp = argmax_2d[ii, :].reshape(-1)[uy]
                unproj = torch.zeros(npts).long()
                unproj[: temp.shape[0]] = temp
            ev.addBatch(unproj, labels)
            arrs[f"{tag}/uy{ii}"], arrs[f"{tag}/labels{ii}"], arrs[f"{tag}/unproj{ii}"] = uy, labels, unproj
        arrs[f"{tag}/conf"] = ev.conf_matrix
        for name, (mean, per) in (("iou", ev.getIoU()), ("acc", ev.getAcc()), ("recall", ev.getRecall())):
            arrs[f"{tag}/{name}_mean"], arrs[f"{tag}/{name}"] = mean, per
    npz("metrics.npz", **arrs)


def synthetic_scan(seed, n):
    """LiDAR-like scan: points on rays inside (and slightly outside) a 3/-25 degree field of view."""
    g = np.random.Generator(np.random.PCG64(seed))
    yaw = g.uniform(-np.pi, np.pi, n)
    pitch = np.deg2rad(g.uniform(-27.0, 4.0, n))
    rng = g.uniform(2.0, 80.0, n) ** 1.0
    pc = np.stack([rng * np.cos(pitch) * np.cos(yaw), rng * np.cos(pitch) * np.sin(yaw), rng * np.sin(pitch),
                   g.uniform(0, 1, n)], 1).astype(np.float32)
    pc[: n // 50] = pc[n // 50: 2 * (n // 50)]          # exact duplicates: equal depth, same pixel
    sem = g.integers(0, 20, n)
    weak = sem * (g.random(n) < 0.01)
    return pc, sem, weak


def gold_projection():
    """The reference RangeProjection / Augmentor (loaded by file: their package __init__ pulls in
    the dataset dependencies) on a synthetic scan, plus the loader tensors exactly as
    wss_sem_kitti_loader.py:113-164 derives them."""
    import importlib.util
    import random

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    proj_m = load("_ref_projection", "/root/reference/pc_processor/dataset/preprocess/projection.py")
    aug_m = load("_ref_augmentor", "/root/reference/pc_processor/dataset/preprocess/augmentor.py")
    pc, sem, weak = synthetic_scan(5, 12000)
    arrs = {"pc": pc, "sem": sem, "weak": weak}
    # augmentation: seeded host draws, then the reference transforms
    params = aug_m.AugmentParams()
    params.setFlipProb(0.5, 0.5)
    params.setTranslationParams(1.0, -5, 5, 1.0, -3, 3, 1.0, -1, 0)
    params.setRotationParams(1.0, -5, 5, 1.0, -5, 5, 1.0, -180, 180)
    random.seed(11)
    aug = aug_m.Augmentor(params).doAugmentation(pc.copy())
    arrs["aug"] = aug
    for tag, src, w, h in (("raw", pc, 256, 32), ("aug", aug, 2048, 64)):
        rp = proj_m.RangeProjection(fov_up=3, fov_down=-25, fov_left=-180, fov_right=180, proj_w=w, proj_h=h)
        proj_pc, proj_range, proj_idx, proj_mask = rp.doProjection(src.copy())
        arrs.update({f"{tag}/proj_pc": proj_pc, f"{tag}/proj_range": proj_range, f"{tag}/proj_idx": proj_idx,
                     f"{tag}/proj_mask": proj_mask, f"{tag}/ux": rp.cached_data["uproj_x_idx"],
                     f"{tag}/uy": rp.cached_data["uproj_y_idx"], f"{tag}/udepth": rp.cached_data["uproj_depth"]})
        ev = np.zeros(proj_idx.shape, dtype=np.float32)
        ev[proj_idx > -1] = sem[proj_idx[proj_idx > -1]]
        tr = np.zeros(proj_idx.shape, dtype=np.float32)
        tr[proj_idx > -1] = weak[proj_idx[proj_idx > -1]]
        inten = torch.from_numpy(proj_pc[..., 3])
        feat = torch.cat([torch.from_numpy(proj_range).unsqueeze(0), torch.from_numpy(proj_pc[..., :3]).permute(2, 0, 1),
                          (inten.ne(-1).float() * inten).unsqueeze(0)], 0)
        arrs.update({f"{tag}/eval_label": ev, f"{tag}/train_label": tr, f"{tag}/feature": feat})
        if tag == "aug":     # full KITTI width: keep the index-level outputs only (the images follow from them)
            for k in ("proj_pc", "proj_range", "feature", "udepth"):
                del arrs[f"{tag}/{k}"]
    npz("projection.npz", **arrs)


def gold_knn():
    """The reference KNN module (pc_processor/postproc/knn.py, loaded by file) on a synthetic
    scan: smooth range image with invalid pixels, points scattered around their pixels."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_ref_knn", "/root/reference/pc_processor/postproc/knn.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = np.random.Generator(np.random.PCG64(23))
    arrs = {}
    for tag, h, w, npts, ncls, params in (("a", 16, 96, 1200, 20, dict(knn=5, search=5, sigma=1.0, cutoff=1.0)),
                                          ("b", 12, 64, 700, 14, dict(knn=7, search=7, sigma=2.0, cutoff=0.0))):
        yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        rng = (10 + 5 * np.sin(xx / 9.0) + 0.3 * yy + g.normal(0, 0.05, (h, w))).astype(np.float32)
        rng[g.random((h, w)) < 0.1] = -1.0
        lab = ((xx // 8 + yy // 4) % ncls).astype(np.int64)
        py = g.integers(0, h, npts)
        px = g.integers(0, w, npts)
        ur = (np.abs(rng[py, px]) + g.normal(0, 0.2, npts)).astype(np.float32)
        knn = mod.KNN(params, ncls)
        out = knn(torch.from_numpy(rng), torch.from_numpy(ur), torch.from_numpy(lab), torch.from_numpy(px), torch.from_numpy(py))
        arrs.update({f"{tag}/proj_range": rng, f"{tag}/proj_argmax": lab, f"{tag}/px": px, f"{tag}/py": py,
                     f"{tag}/unproj_range": ur, f"{tag}/out": out})
    npz("knn.npz", **arrs)


def gold_rangenet():
    """Reference RangeNetProto(layers=21) forward + backward with closed-form weights and injected
    Dropout2d masks.  Gradients of a 25 M-parameter model are stored as per-tensor
    (sum, sum of squares) plus a few small tensors in full."""
    import contextlib
    import importlib
    import io
    m = importlib.import_module("pc_processor.models.rangenet_proto")
    arrs = {}
    for tag, b, h, w, ncls, dataset in (("kitti", 2, 8, 64, 20, "SemanticKitti"), ("poss", 1, 8, 40, 14, "SemanticPOSS")):
        with contextlib.redirect_stdout(io.StringIO()):
            net = m.RangeNetProto(layers=21, nclasses=ncls, use_prototype=True, dataset=dataset)
        net.load_state_dict(W.rangenet_state(nclasses=ncls))
        net.train()
        masks = W.rangenet_masks(b, 3)
        seq = iter(["enc1", "enc2", "enc3", "enc4", "enc5"])
        net.backbone.dropout.forward = lambda x, seq=seq, masks=masks: x * masks[next(seq)][:, :, None, None]
        net.decoder.dropout.forward = lambda x, masks=masks: x * masks["decoder"][:, :, None, None]
        net.head[0].forward = lambda x, masks=masks: x * masks["head"][:, :, None, None]
        g = torch.Generator().manual_seed(1)
        x = torch.randn(b, 5, h, w, generator=g)
        with contextlib.redirect_stdout(io.StringIO()):
            out = net(x, return_feat=True)
        dp = torch.randn(out["pred_2d"].shape, generator=g)
        df = torch.randn(out["feat_2d"].shape, generator=g) * 0.05
        loss = (out["pred_2d"] * dp).sum() + (out["feat_2d"] * df).sum()
        loss.backward()
        # x, dp, df are regenerated by the tests from the same seeded generator
        arrs.update({f"{tag}/pred_2d": out["pred_2d"].detach(), f"{tag}/feat_2d_sub": out["feat_2d"].detach()[:, ::4, :, ::2]})
        names = []
        for k, p_ in net.named_parameters():
            if p_.grad is None:
                continue
            names.append(k)
            gd = p_.grad.double()
            arrs[f"{tag}/gsum/{k}"] = gd.sum()
            arrs[f"{tag}/gsq/{k}"] = (gd * gd).sum()
        for k in ("backbone.conv1.weight", "head.1.weight", "head.1.bias", "decoder.dec1.upconv.weight",
                  "backbone.enc1.residual_0.bn2.weight", "projector.proj.3.bias", "decoder.dec5.bn.bias"):
            arrs[f"{tag}/grad/{k}"] = dict(net.named_parameters())[k].grad
        arrs[f"{tag}/grad_names"] = np.array(names)
        sd = net.state_dict()
        for k in ("backbone.bn1.running_mean", "backbone.enc5.bn.running_var", "decoder.dec1.residual.bn2.running_mean",
                  "projector.proj.1.running_var"):
            arrs[f"{tag}/run/{k}"] = sd[k]
    npz("rangenet.npz", **arrs)


# ----------------------------------------------------------------------------- full step
def gold_step():
    """One optimisation step of the reference modules, trainer.py:621-704 order, with
    use_prototype / proto_loss switched on (latent in the shipped trainer)."""
    b, h, w, ncls = 2, 64, 128, 20
    st = W.closed_form_state(nclasses=ncls)
    x, tr, ev = W.synthetic_batch(b, h, w, ncls, 77, 0.02, gh=8, gw=16)
    masks = W.dropout_masks_for(None, b, 78)
    m = build_ref_model(st)
    m.train()
    install_dropout_masks(m, masks)
    alpha = torch.ones(ncls)
    alpha[0] = 0
    focal = R.FocalSoftmaxLoss(ncls, gamma=2, alpha=alpha.numpy(), softmax=False)
    lov = R.Lovasz_softmax(ignore=0, per_image=False, softmax=False)
    con = R.ContrastMEMLoss(ignore_label=0, temperature=0.07, num_anchor=64)
    fake_self = types.SimpleNamespace(settings=types.SimpleNamespace(ignore_cls=0, n_classes=ncls))
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    torch.manual_seed(79)
    with Recorder() as rec:
        out = m(x, label=tr, eval_mask=tr > 0, return_feat=True, proto_loss=True)
        n_gumbel = len(rec.exp)
        pred, feat = out["pred_2d"], out["feat_2d"]
        l_ce = focal(pred, tr, mask=tr > 0)
        l_lov = lov(pred, tr)
        ratio = float(np.log(1 + (1 + 10) / 100) / np.log(2) * 0.5)
        with torch.no_grad():
            lab_c, mask_c = R.entropy_based_selection(
                fake_self, output=pred, wss_mask=tr > 0, eval_mask=ev > 0, train_label=tr,
                select_ratio=ratio)
        l_con = con(feats=feat, output=pred, labels=lab_c, keep_mask=mask_c,
                    proto_queue=m.prototypes.detach().unsqueeze(0))
        total = 1.0 * l_ce + 1.0 * l_lov + 0.1 * l_con
    opt.zero_grad()
    total.backward()
    arrs = dict(loss=total.detach(), ce=l_ce.detach(), lov=l_lov.detach(), contrast=l_con.detach(),
                labels_contra=lab_c, mask_contra=mask_c, pred_2d=pred.detach(),
                new_prototypes=m.prototypes.detach(),
                pl_noise=torch.stack([q for _, _, q, _ in rec.multi_norep]),
                uniforms=torch.stack([u for _, u, _ in rec.multi_rep]),
                anchor_idx=torch.stack([i for _, _, i in rec.multi_rep]),
                perms=torch.stack(rec.perms))
    present = [c for c in range(1, ncls) if int((tr == c).sum()) > 0]
    assert len(present) == n_gumbel
    for c, e in zip(present, rec.exp[:n_gumbel]):
        arrs[f"gumbel_{c}"] = e
    for k, p in m.named_parameters():
        if p.grad is None:
            continue
        gr = p.grad.detach()
        arrs[f"gnorm/{k}"] = gr.norm()
        arrs[f"grad/{k}"] = gr if gr.numel() <= 4096 else gr.reshape(-1)[:: max(gr.numel() // 2048, 1)]
    opt.step()
    for k in ("downCntx.conv1.weight", "resBlock3.bn2.bias", "cls_head.weight",
              "projector.proj.3.bias", "upBlock2.conv4.bias"):
        arrs[f"after/{k}"] = dict(m.named_parameters())[k].detach()
    npz("step.npz", **arrs)


def gold_init():
    """Default initialisation of the reference under a fixed seed: per-tensor checksums (the
    drop-in module must consume the RNG in the same order, salsanext_proto.py:284-328)."""
    torch.manual_seed(7)
    m = R.SalsaNextProto(5, 20, 20, 0)
    arrs = {"keys": np.array(list(m.state_dict().keys()))}
    for k, v in m.state_dict().items():
        v = v.double()
        arrs[f"sum/{k}"] = v.sum()
        arrs[f"sq/{k}"] = (v * v).sum()
    npz("init_checksums.npz", **arrs)


if __name__ == "__main__":
    gold_init()
    gold_blocks()
    gold_model("kitti_small", 2, 32, 64, 20, "SemanticKitti", 101, 0.02)
    gold_model("poss_small", 1, 24, 56, 14, "SemanticPOSS", 201, 0.02)
    gold_contrast()
    gold_pl_select()
    gold_losses()
    gold_metrics()
    gold_projection()
    gold_knn()
    gold_rangenet()
    gold_step()
