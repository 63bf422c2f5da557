"""CPU oracle for the COARSE3D training hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch *restatement* (plain PyTorch-CPU fp32 ops + NumPy) of the
algorithm the upstream reference implements with ``torch.nn`` modules.  It exists so the
HIP path can be checked on machines where the reference itself is absent (the GPU box).
It is imported only by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` -- never by the product package ``coarse3d_amd``.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real reference in
the build container and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function below against those vectors.

Reference map (paths relative to /root/reference):
  res_context_block      pc_processor/models/salsanext_proto.py:38-65
  res_block              pc_processor/models/salsanext_proto.py:68-148
  up_block               pc_processor/models/salsanext_proto.py:151-212
  backbone_forward       pc_processor/models/salsanext_proto.py:423-492 (+ projector.py:11-27)
  prototype_similarity   pc_processor/models/salsanext_proto.py:494-510
  sinkhorn_assign        pc_processor/models/sinkhorn.py:5-33
  prototype_learning     pc_processor/models/salsanext_proto.py:337-402
  anchor_weights / sample_anchors / info_nce / contrast_mem_loss
                         pc_processor/loss/contrast_pixel_loss.py:27-195
  entropy_selection      tasks/weak_segmentation/trainer.py:447-518
  focal_loss             pc_processor/loss/focal_softmax.py:30-77
  lovasz_loss            pc_processor/loss/lovasz_softmax.py:56-68,101-160
  normalise_input        tasks/weak_segmentation/trainer.py:599-609
  train_step             tasks/weak_segmentation/trainer.py:621-704
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.01
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
DROP_P = 0.2

# (name, kind) of the 13 Dropout2d sites in execution order (Appendix A of SURVEY.md)
DROPOUT_SITES = (
    "resBlock2.dropout", "resBlock3.dropout", "resBlock4.dropout", "resBlock5.dropout",
    "upBlock1.dropout1", "upBlock1.dropout2", "upBlock1.dropout3",
    "upBlock2.dropout1", "upBlock2.dropout2", "upBlock2.dropout3",
    "upBlock3.dropout1", "upBlock3.dropout2", "upBlock3.dropout3",
)


# --------------------------------------------------------------------------------------
# parameter table
# --------------------------------------------------------------------------------------
def conv_specs(in_channel=5, nclasses=20, base=32, proj_dim=256):
    """Ordered {conv name: (cout, cin, kh, kw)} of the 53 convolutions."""
    s = OrderedDict()

    def ctx(name, cin, cout):
        s[f"{name}.conv1"] = (cout, cin, 1, 1)
        s[f"{name}.conv2"] = (cout, cout, 3, 3)
        s[f"{name}.conv3"] = (cout, cout, 3, 3)

    def res(name, cin, cout):
        s[f"{name}.conv1"] = (cout, cin, 1, 1)
        s[f"{name}.conv2"] = (cout, cin, 3, 3)
        s[f"{name}.conv3"] = (cout, cout, 3, 3)
        s[f"{name}.conv4"] = (cout, cout, 2, 2)
        s[f"{name}.conv5"] = (cout, 3 * cout, 1, 1)

    def up(name, cin, cout):
        s[f"{name}.conv1"] = (cout, cin // 4 + 2 * cout, 3, 3)
        s[f"{name}.conv2"] = (cout, cout, 3, 3)
        s[f"{name}.conv3"] = (cout, cout, 2, 2)
        s[f"{name}.conv4"] = (cout, 3 * cout, 1, 1)

    ctx("downCntx", in_channel, base)
    ctx("downCntx2", base, base)
    ctx("downCntx3", base, base)
    res("resBlock1", base, 2 * base)
    res("resBlock2", 2 * base, 4 * base)
    res("resBlock3", 4 * base, 8 * base)
    res("resBlock4", 8 * base, 8 * base)
    res("resBlock5", 8 * base, 8 * base)
    up("upBlock1", 8 * base, 4 * base)
    up("upBlock2", 4 * base, 4 * base)
    up("upBlock3", 4 * base, 2 * base)
    up("upBlock4", 2 * base, base)
    s["cls_head"] = (nclasses, base, 1, 1)
    s["projector.proj.0"] = (22 * base, 22 * base, 1, 1)
    s["projector.proj.3"] = (proj_dim, 22 * base, 1, 1)
    return s


def bn_specs(base=32):
    """Ordered {bn name: channels} of the 43 BatchNorm2d layers."""
    s = OrderedDict()
    for n in ("downCntx", "downCntx2", "downCntx3"):
        s[f"{n}.bn1"] = base
        s[f"{n}.bn2"] = base
    for n, c in (("resBlock1", 2), ("resBlock2", 4), ("resBlock3", 8), ("resBlock4", 8),
                 ("resBlock5", 8), ("upBlock1", 4), ("upBlock2", 4), ("upBlock3", 2),
                 ("upBlock4", 1)):
        for i in range(1, 5):
            s[f"{n}.bn{i}"] = c * base
    s["projector.proj.1"] = 22 * base
    return s


def init_state(in_channel=5, nclasses=20, sub_proto=20, proj_dim=256, seed=1, base=32):
    """Default-style initialisation (kaiming-uniform convs, unit BN, trunc-normal bank).

    Returns an OrderedDict with the reference's state_dict key names
    (salsanext_proto.py:284-328)."""
    g = torch.Generator().manual_seed(seed)
    st = OrderedDict()
    bns = bn_specs(base)
    for name, (co, ci, kh, kw) in conv_specs(in_channel, nclasses, base, proj_dim).items():
        fan_in = ci * kh * kw
        bound = 1.0 / math.sqrt(fan_in)
        st[f"{name}.weight"] = (torch.rand(co, ci, kh, kw, generator=g) * 2 - 1) * bound
        st[f"{name}.bias"] = (torch.rand(co, generator=g) * 2 - 1) * bound
    for name, c in bns.items():
        st[f"{name}.weight"] = torch.ones(c)
        st[f"{name}.bias"] = torch.zeros(c)
        st[f"{name}.running_mean"] = torch.zeros(c)
        st[f"{name}.running_var"] = torch.ones(c)
        st[f"{name}.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    protos = torch.empty(nclasses, sub_proto, proj_dim)
    torch.nn.init.trunc_normal_(protos, std=0.02, generator=g)
    st["prototypes"] = protos
    st["feat_norm.weight"] = torch.ones(proj_dim)
    st["feat_norm.bias"] = torch.zeros(proj_dim)
    st["mask_norm.weight"] = torch.ones(nclasses)
    st["mask_norm.bias"] = torch.zeros(nclasses)
    return st


def trainable_names(state):
    return [k for k, v in state.items()
            if v.is_floating_point() and k != "prototypes"
            and not k.endswith("running_mean") and not k.endswith("running_var")]


# --------------------------------------------------------------------------------------
# backbone
# --------------------------------------------------------------------------------------
class _Ctx:
    """Carries parameters, mode, injected dropout masks and collects BN batch statistics."""

    def __init__(self, state, train, dropout_masks, update_running=True):
        self.p = state
        self.train = train
        self.masks = dropout_masks
        self.update_running = update_running
        self.bn_stats = OrderedDict()

    def conv(self, name, x, dilation=1, padding=0):
        return F.conv2d(x, self.p[f"{name}.weight"], self.p[f"{name}.bias"],
                        stride=1, padding=padding, dilation=dilation)

    def bn(self, name, x):
        w, b = self.p[f"{name}.weight"], self.p[f"{name}.bias"]
        rm, rv = self.p[f"{name}.running_mean"], self.p[f"{name}.running_var"]
        if not self.train:
            scale = w / torch.sqrt(rv + BN_EPS)
            return x * scale[None, :, None, None] + (b - rm * scale)[None, :, None, None]
        n = x.numel() // x.shape[1]
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        self.bn_stats[name] = (mean.detach().clone(), var.detach().clone())
        if self.update_running:
            with torch.no_grad():
                rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
                rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var.detach() * n / max(n - 1, 1))
                self.p[f"{name}.num_batches_tracked"] += 1
        xhat = (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS)
        return xhat * w[None, :, None, None] + b[None, :, None, None]

    def drop(self, name, x):
        """Dropout2d: whole (b, c) planes zeroed, survivors scaled by 1/(1-p).
        ``self.masks[name]`` is the injected [B, C] multiplier (0 or 1.25)."""
        if not self.train or self.masks is None:
            return x
        m = self.masks[name]
        return x * m[:, :, None, None]


def _act(x):
    return F.leaky_relu(x, LRELU_SLOPE)


def res_context_block(c, name, x):
    short = _act(c.conv(f"{name}.conv1", x))
    a = c.bn(f"{name}.bn1", _act(c.conv(f"{name}.conv2", short, 1, 1)))
    a = c.bn(f"{name}.bn2", _act(c.conv(f"{name}.conv3", a, 2, 2)))
    return short + a


def res_block(c, name, x, pooling=True, drop_out=True):
    short = _act(c.conv(f"{name}.conv1", x))
    r1 = c.bn(f"{name}.bn1", _act(c.conv(f"{name}.conv2", x, 1, 1)))
    r2 = c.bn(f"{name}.bn2", _act(c.conv(f"{name}.conv3", r1, 2, 2)))
    r3 = c.bn(f"{name}.bn3", _act(c.conv(f"{name}.conv4", r2, 2, 1)))
    r = c.bn(f"{name}.bn4", _act(c.conv(f"{name}.conv5", torch.cat((r1, r2, r3), 1))))
    r = short + r
    rb = c.drop(f"{name}.dropout", r) if drop_out else r
    if pooling:
        rb = F.avg_pool2d(rb, kernel_size=3, stride=2, padding=1)
        return rb, r
    return rb


def up_block(c, name, x, skip, drop_out=True):
    u = F.pixel_shuffle(x, 2)
    if drop_out:
        u = c.drop(f"{name}.dropout1", u)
    u = torch.cat((u, skip), 1)
    if drop_out:
        u = c.drop(f"{name}.dropout2", u)
    e1 = c.bn(f"{name}.bn1", _act(c.conv(f"{name}.conv1", u, 1, 1)))
    e2 = c.bn(f"{name}.bn2", _act(c.conv(f"{name}.conv2", e1, 2, 2)))
    e3 = c.bn(f"{name}.bn3", _act(c.conv(f"{name}.conv3", e2, 2, 1)))
    e = c.bn(f"{name}.bn4", _act(c.conv(f"{name}.conv4", torch.cat((e1, e2, e3), 1))))
    if drop_out:
        e = c.drop(f"{name}.dropout3", e)
    return e


def backbone_forward(state, x, train=True, dropout_masks=None, return_feat=True,
                     dataset="SemanticKitti", update_running=True, return_aux=False):
    """x [B,5,H,W] -> dict(pred_2d [B,C,H,W], feat_2d [B,256,H,W], logits).

    POSS: zero-pad by 8 rows/cols before the net and crop the logits
    (salsanext_proto.py:426-430, 457-458)."""
    c = _Ctx(state, train, dropout_masks, update_running)
    if dataset == "SemanticPOSS":
        x = F.pad(x, (0, 8, 0, 8))
    assert x.shape[2] % 16 == 0 and x.shape[3] % 16 == 0
    d = res_context_block(c, "downCntx", x)
    d = res_context_block(c, "downCntx2", d)
    d = res_context_block(c, "downCntx3", d)
    d0c, d0b = res_block(c, "resBlock1", d, True, False)
    d1c, d1b = res_block(c, "resBlock2", d0c)
    d2c, d2b = res_block(c, "resBlock3", d1c)
    d3c, d3b = res_block(c, "resBlock4", d2c)
    d5c = res_block(c, "resBlock5", d3c, pooling=False)
    u4 = up_block(c, "upBlock1", d5c, d3b)
    u3 = up_block(c, "upBlock2", u4, d2b)
    u2 = up_block(c, "upBlock3", u3, d1b)
    u1 = up_block(c, "upBlock4", u2, d0b, drop_out=False)
    logits = c.conv("cls_head", u1)
    if dataset == "SemanticPOSS":
        logits = logits[:, :, :-8, :-8]
    out = {"logits": logits, "pred_2d": F.softmax(logits, dim=1)}
    if return_feat:
        h, w = logits.shape[2] // 2, logits.shape[3] // 2
        feat = torch.cat([F.interpolate(t, size=(h, w), mode="bilinear", align_corners=True)
                          for t in (d0b, d1b, d2b, d3b)], 1)
        z = c.conv("projector.proj.0", feat)
        z = _act(c.bn("projector.proj.1", z))
        emb = c.conv("projector.proj.3", z)
        emb = F.normalize(emb, p=2, dim=1)
        emb = F.interpolate(emb, size=logits.shape[2:], mode="bilinear", align_corners=True)
        out["feat_2d"] = emb
    out["bn_stats"] = c.bn_stats
    if return_aux:
        out["aux"] = {"downCntx3": d, "down0b": d0b, "down1b": d1b, "down2b": d2b,
                      "down3b": d3b, "down5c": d5c, "up4e": u4, "up3e": u3, "up2e": u2,
                      "up1e": u1}
    return out


# --------------------------------------------------------------------------------------
# prototype bank
# --------------------------------------------------------------------------------------
def _l2(x):
    return F.normalize(x, p=2, dim=-1)


def prototype_similarity(state, feat_2d):
    """feat_2d [B,D,H,W] -> (out_feat [N,D], sim [N,M,C], nearest [B,C,H,W], protos_l2).

    The bank is re-normalised in place first (salsanext_proto.py:502)."""
    b, d, h, w = feat_2d.shape
    rows = feat_2d.permute(0, 2, 3, 1).reshape(-1, d)
    rows = F.layer_norm(rows, (d,), state["feat_norm.weight"], state["feat_norm.bias"])
    rows = _l2(rows)
    protos = _l2(state["prototypes"])
    sim = torch.einsum("nd,kmd->nmk", rows, protos)
    nearest = sim.amax(dim=1)
    ncls = nearest.shape[1]
    nearest = F.layer_norm(nearest, (ncls,), state["mask_norm.weight"], state["mask_norm.bias"])
    nearest = nearest.reshape(b, h, w, ncls).permute(0, 3, 1, 2)
    return rows, sim, nearest, protos


def sinkhorn_assign(scores, exp_noise, iters=3, eps=0.05):
    """scores [n,M] -> (q one-hot [n,M] from the Gumbel-hard draw, argmax index [n]).

    ``exp_noise`` [n,M] are the Exp(1) variates F.gumbel_softmax would draw
    (gumbel = -log(noise)); tau = 0.5 (sinkhorn.py:31)."""
    q = torch.exp(scores / eps).t()
    n = q.shape[1]
    k = q.shape[0]
    q = q / q.sum()
    for _ in range(iters):
        q = q / q.sum(dim=1, keepdim=True)
        q = q / k
        q = q / q.sum(dim=0, keepdim=True)
        q = q / n
    q = (q * n).t()
    index = torch.argmax(q, dim=1)
    gumbel = -torch.log(exp_noise)
    soft = F.softmax((q + gumbel) / 0.5, dim=-1)
    hot = torch.zeros_like(q).scatter_(1, soft.argmax(dim=-1, keepdim=True), 1.0)
    hot = hot - soft + soft  # straight-through value, as the reference computes it
    return hot, index


def prototype_learning(protos_l2, rows, nearest, label, sim, exp_noise, ignore_label=0,
                       momentum=0.999, world_mean=None):
    """Per-class masked reduction + EMA into the bank (salsanext_proto.py:337-402).

    protos_l2 [C,M,D] (already l2), rows [N,D], nearest [B,C,H,W], label [N] long,
    sim [N,M,C]; exp_noise: dict class -> [n_c, M] Exp(1) noise (row order = ascending
    pixel index).  Returns (new_protos [C,M,D], logits [N,M*C], target [N] float)."""
    ncls, m, _ = protos_l2.shape
    pred = nearest.argmax(dim=1).reshape(-1)
    hit = label == pred
    logits = sim.reshape(sim.shape[0], -1)
    target = torch.zeros_like(label).float()
    bank = protos_l2.clone()
    for cls in range(ncls):
        if cls == ignore_label:
            continue
        sel = label == cls
        if int(sel.sum()) == 0:
            continue
        noise = exp_noise[cls] if exp_noise is not None else torch.empty(int(sel.sum()), m).exponential_()
        q, index = sinkhorn_assign(sim[sel][:, :, cls], noise)
        keep = hit[sel].float()
        qm = q * keep[:, None]
        f = qm.t() @ (rows[sel] * keep[:, None])
        cnt = qm.sum(dim=0)
        if float(cnt.sum()) > 0:
            f = F.normalize(f, p=2, dim=-1)
            nz = cnt != 0
            bank[cls, nz] = momentum * bank[cls, nz] + (1 - momentum) * f[nz]
        target[sel] = index.float() + m * cls
    bank = _l2(bank)
    if world_mean is not None:  # data-parallel: mean over ranks (salsanext_proto.py:397-400)
        bank = world_mean(bank)
    return bank, logits, target


# --------------------------------------------------------------------------------------
# samplers (bit-exact index contracts)
# --------------------------------------------------------------------------------------
def multinomial_replace(weights, uniforms):
    """== torch.multinomial(weights, len(uniforms), replacement=True) on CPU when
    ``uniforms`` is the float64 stream torch.rand(n, dtype=float64) of the same generator.

    Sequential fp32 running sum, divided by the fp32 total, left-bisect of each draw
    (contrast_pixel_loss.py:114-116 calls it; semantics pinned in tests)."""
    w = np.asarray(weights, dtype=np.float32).reshape(-1)
    nz = np.flatnonzero(w)
    run = np.float32(0)
    cum = np.empty(len(nz), dtype=np.float32)
    vals = w[nz]
    for i in range(len(nz)):          # zeros leave an fp32 running sum unchanged
        run = np.float32(run + vals[i])
        cum[i] = run
    cum = (cum / run).astype(np.float32)
    u = np.asarray(uniforms, dtype=np.float64)
    pos = np.searchsorted(cum.astype(np.float64), u, side="left")
    pos = np.minimum(pos, len(nz) - 1)
    out = nz[pos]
    out[u <= 0.0] = 0
    return out.astype(np.int64)


def multinomial_noreplace_set(weights, k, exp_noise):
    """Index *set* of torch.multinomial(weights, k, replacement=False):
    top-k of weights / Exp(1) noise (fp32 division)."""
    q = (np.asarray(weights, np.float32) / np.asarray(exp_noise, np.float32)).astype(np.float32)
    idx = np.argpartition(-q, k - 1)[:k]
    return np.sort(idx).astype(np.int64)


def pixel_entropy(prob):
    return -(prob * torch.log(prob + 1e-10)).sum(dim=1)


def entropy_selection(prob, wss_mask, eval_mask, train_label, select_ratio, exp_noise,
                      ignore_cls=0):
    """Pseudo-label selection (trainer.py:447-518).

    exp_noise: list of [H*W] fp32 Exp(1) draws, one per (image, class) that reaches the
    multinomial, in (b, ascending class) order.  Returns (labels [B,H,W] long, mask)."""
    bs, ncls, h, w = prob.shape
    weight = torch.exp(-pixel_entropy(prob))
    pseudo = prob.argmax(dim=1)
    pseudo[~eval_mask] = ignore_cls
    chosen = torch.zeros(bs, h * w, dtype=torch.bool)
    it = iter(exp_noise) if exp_noise is not None else None
    ratio32 = np.float32(select_ratio)
    for b in range(bs):
        for cls in torch.unique(train_label[b]).tolist():
            if cls == ignore_cls:
                continue
            cmask = ((pseudo[b] == cls) & eval_mask[b]).reshape(-1)
            cnt = int(cmask.sum())
            if cnt == 0:
                continue
            k = int(np.float32(cnt) * ratio32)
            if k < 1:
                continue
            wc = weight[b].reshape(-1).clone()
            wc[~cmask] = 0
            q = next(it) if it is not None else torch.empty(h * w).exponential_()
            idx = multinomial_noreplace_set(wc.numpy(), k, q.numpy())
            chosen[b, torch.from_numpy(idx)] |= True
    chosen = chosen.reshape(bs, h, w)
    labels = (pseudo * chosen).long()
    labels[wss_mask] = train_label[wss_mask]
    return labels, labels != ignore_cls


# --------------------------------------------------------------------------------------
# contrastive loss
# --------------------------------------------------------------------------------------
def anchor_weights(prob):
    ent = pixel_entropy(prob)
    return torch.exp(-(ent * ent))


def sample_anchors(labels, weights, uniforms, num_anchor, ignore_label=0):
    """labels [B,N] long, weights [B,N] -> (image index [T], class [T], pixel idx [T,A]).

    uniforms: [T_max, A] float64; the t-th present (image, class) pair consumes row t."""
    imgs, clss, idxs = [], [], []
    t = 0
    for b in range(labels.shape[0]):
        for cls in torch.unique(labels[b]).tolist():
            if cls == ignore_label:
                continue
            wc = weights[b].clone()
            wc[labels[b] != cls] = 0
            u = uniforms[t] if uniforms is not None else torch.rand(num_anchor, dtype=torch.float64)
            idx = multinomial_replace(wc.numpy(), np.asarray(u))
            imgs.append(b)
            clss.append(cls)
            idxs.append(torch.from_numpy(idx))
            t += 1
    if t == 0:
        return None, None, None
    return torch.tensor(imgs), torch.tensor(clss), torch.stack(idxs)


def info_nce(anchors, anchor_cls, queue, perms, temperature, base_temperature=0.07):
    """anchors [T,A,D], anchor_cls [T], queue [C,M,D], perms [C-1,M] row orders.

    Class 0 never enters the queue (contrast_pixel_loss.py:139-140)."""
    t_, a_, d_ = anchors.shape
    ncls, m, _ = queue.shape
    if perms is None:
        perms = torch.stack([torch.randperm(m) for _ in range(ncls - 1)])
    bank = torch.cat([queue[c][perms[c - 1]] for c in range(1, ncls)], 0)
    bank_cls = torch.arange(1, ncls).repeat_interleave(m)
    af = anchors.permute(1, 0, 2).reshape(a_ * t_, d_)          # anchor-major order
    acls = anchor_cls.repeat(a_)
    af = F.normalize(af, p=2, dim=-1)
    bank = F.normalize(bank, p=2, dim=-1)
    logits = (af @ bank.t()) / temperature
    logits = logits - logits.max(dim=1, keepdim=True)[0].detach()
    pos = (acls[:, None] == bank_cls[None, :]).float()
    ex = torch.exp(logits)
    neg = (ex * (1 - pos)).sum(1, keepdim=True)
    logp = logits - torch.log(ex + neg + 1e-6)
    mean_pos = (pos * logp).sum(1) / pos.sum(1)
    return (-(temperature / base_temperature) * mean_pos).mean()


def contrast_mem_loss(feats, prob, labels, keep_mask, queue, uniforms, perms,
                      temperature=0.1, num_anchor=50, ignore_label=0, return_idx=False):
    """feats [B,D,H,W], prob [B,C,H,W], labels [B,H,W], keep_mask bool, queue [C,M,D]."""
    labels = labels.clone()
    labels[~keep_mask.bool()] = ignore_label
    b, d, h, w = feats.shape
    wts = anchor_weights(prob).reshape(b, -1)
    rows = feats.permute(0, 2, 3, 1).reshape(b, h * w, d)
    lab = labels.reshape(b, -1)
    img, cls, idx = sample_anchors(lab, wts, uniforms, num_anchor, ignore_label)
    if img is None:
        return None
    anchors = rows[img[:, None], idx]                         # [T,A,D]
    loss = info_nce(anchors, cls, queue, perms, temperature)
    if return_idx:
        return loss, (img, cls, idx)
    return loss


# --------------------------------------------------------------------------------------
# supervised losses (stock ops in the product too; restated for the step oracle)
# --------------------------------------------------------------------------------------
def focal_loss(prob, target, mask, alpha, gamma=2):
    p = prob.permute(0, 2, 3, 1).reshape(-1, prob.shape[1])
    t = target.reshape(-1, 1)
    pt = p.gather(1, t).reshape(-1)
    loss = -(1 - pt).pow(gamma) * pt.clamp(1e-6).log() * alpha.gather(0, t.reshape(-1))
    m = mask.reshape(-1)
    out = (loss * m).sum() / m.sum()
    if torch.isnan(out):
        return torch.tensor(0.0)
    return out


def lovasz_loss(prob, labels, ignore=0):
    ncls = prob.shape[1]
    p = prob.permute(0, 2, 3, 1).reshape(-1, ncls)
    lab = labels.reshape(-1)
    valid = lab != ignore
    p, lab = p[valid], lab[valid]
    if p.numel() == 0:
        return p.sum() * 0.0
    losses = []
    for cls in range(ncls):
        fg = (lab == cls).float()
        if fg.sum() == 0:
            continue
        err = (fg - p[:, cls]).abs()
        err_sorted, perm = torch.sort(err, 0, descending=True)
        fgs = fg[perm]
        total = fgs.sum()
        inter = total - fgs.cumsum(0)
        union = total + (1 - fgs).cumsum(0)
        jac = 1.0 - inter / union
        jac = torch.cat((jac[:1], jac[1:] - jac[:-1]))
        losses.append(torch.dot(err_sorted, jac))
    return sum(losses) / len(losses)


# --------------------------------------------------------------------------------------
# per-iteration metrics
# --------------------------------------------------------------------------------------
def unproject_argmax(pred_2d, uy, ux=None, n_points=None):
    """pred_2d [C,H,W] -> class per point.  trainer.py:713-726: ``argmax(dim=1)`` then either
    ``argmax_2d[ii, uproj_y_idx, uproj_x_idx]`` (SemanticKitti / nuScenes) or, with ``ux=None``
    (SemanticPOSS), ``argmax.reshape(-1)[uproj_y_idx]`` written into the head of a zero vector
    of ``n_points`` entries."""
    am = pred_2d.argmax(dim=0)
    if ux is not None:
        return am[uy.long(), ux.long()]
    temp = am.reshape(-1)[uy.long()]
    out = torch.zeros(n_points, dtype=torch.long)
    out[: temp.shape[0]] = temp
    return out


def confusion_add(conf, pred, label):
    """IOUEval.addBatch (iou_eval.py:35-58): conf[pred][label] += 1 (int64, in place)."""
    idx = pred.reshape(-1).long() * conf.shape[1] + label.reshape(-1).long()
    conf += torch.bincount(idx, minlength=conf.numel()).reshape(conf.shape)
    return conf


def iou_stats(conf, ignore):
    """IOUEval.getStats/getIoU/getAcc/getRecall (iou_eval.py:60-119).  Returns dict of
    (mean over the included classes, per-class vector)."""
    ncls = conf.shape[0]
    include = torch.tensor([n for n in range(ncls) if n not in ignore], dtype=torch.long)
    c = conf.clone().double()
    c[list(ignore)] = 0
    c[:, list(ignore)] = 0
    tp = c.diag()
    fp = c.sum(dim=1) - tp
    fn = c.sum(dim=0) - tp
    out = {}
    for name, den in (("iou", tp + fp + fn + 1e-15), ("acc", tp + fp + 1e-15), ("recall", tp + fn + 1e-15)):
        per = tp / den
        out[name] = ((tp[include] / den[include]).mean(), per)
    return out


# --------------------------------------------------------------------------------------
# scan -> range image
# --------------------------------------------------------------------------------------
def augment_points(pc, flip_x, flip_y, trans, rot_deg):
    """augmentor.py:150-174 on a float32 [n, c] array (returns a new array): sign flips, float32
    translation, then xyz (as float64) times the transposed 'zyx' Euler matrix, cast to float32."""
    import numpy as np
    from scipy.spatial.transform import Rotation
    out = np.array(pc, dtype=np.float32, copy=True)
    if flip_x:
        out[:, 0] = -out[:, 0]
    if flip_y:
        out[:, 1] = -out[:, 1]
    for k in range(3):
        out[:, k] = out[:, k] + np.float32(trans[k])
    roll, pitch, yaw = rot_deg
    m = Rotation.from_euler("zyx", [yaw, pitch, roll], degrees=True).as_matrix()
    out[:, :3] = out[:, :3].astype(np.float64) @ m.T
    return out


def range_projection(pc, fov_up, fov_down, fov_left, fov_right, proj_w, proj_h, depth=None):
    """projection.py:43-115 restated on numpy float32: pixel of every point, closest point per
    pixel (points visited by decreasing depth, later writes win).  Returns dict with proj_pc
    [H,W,c], proj_range, proj_idx, proj_mask, ux, uy, udepth."""
    import numpy as np
    pc = np.asarray(pc, dtype=np.float32)
    up, down = fov_up / 180.0 * np.pi, fov_down / 180.0 * np.pi
    left, right = fov_left / 180.0 * np.pi, fov_right / 180.0 * np.pi
    vert, hori = abs(up) + abs(down), abs(left) + abs(right)
    if depth is None:
        depth = np.sqrt((pc[:, 0] * pc[:, 0] + pc[:, 1] * pc[:, 1]) + pc[:, 2] * pc[:, 2])
    depth = np.asarray(depth, dtype=np.float32)
    yaw = -np.arctan2(pc[:, 1], pc[:, 0])
    pitch = np.arcsin(pc[:, 2] / depth)
    fx = (yaw + np.float32(abs(left))) / np.float32(hori) * np.float32(proj_w)
    fy = (np.float32(1.0) - (pitch + np.float32(abs(down))) / np.float32(vert)) * np.float32(proj_h)
    ux = np.clip(np.floor(fx), 0, proj_w - 1).astype(np.int32)
    uy = np.clip(np.floor(fy), 0, proj_h - 1).astype(np.int32)
    # closest point per pixel; equal depths: the smallest index (the reference's argsort is unstable there)
    order = np.lexsort((np.arange(len(depth)), depth))[::-1]
    proj_idx = np.full((proj_h, proj_w), -1, dtype=np.int32)
    proj_idx[uy[order], ux[order]] = order.astype(np.int32)
    hit = proj_idx >= 0
    proj_range = np.full((proj_h, proj_w), -1, dtype=np.float32)
    proj_range[hit] = depth[proj_idx[hit]]
    proj_pc = np.full((proj_h, proj_w, pc.shape[1]), -1, dtype=np.float32)
    proj_pc[hit] = pc[proj_idx[hit]]
    return dict(proj_pc=proj_pc, proj_range=proj_range, proj_idx=proj_idx, proj_mask=(proj_idx > 0).astype(np.int32),
                ux=ux, uy=uy, udepth=depth)


def loader_tensors(proj, sem_label, weak_label):
    """wss_sem_kitti_loader.py:121-131,147-164: label images from the winning point of each pixel
    (0 where empty) and the 5-channel input (range, x, y, z, intensity with -1 -> 0)."""
    import numpy as np
    idx = proj["proj_idx"]
    hit = idx > -1
    ev = np.zeros(idx.shape, dtype=np.float32)
    tr = np.zeros(idx.shape, dtype=np.float32)
    ev[hit] = sem_label[idx[hit]]
    tr[hit] = weak_label[idx[hit]]
    inten = proj["proj_pc"][..., 3]
    feat = np.concatenate([proj["proj_range"][None], proj["proj_pc"][..., :3].transpose(2, 0, 1),
                           ((inten != -1).astype(np.float32) * inten)[None]], 0)
    return dict(feature=feat, eval_label=ev, train_label=tr)


# --------------------------------------------------------------------------------------
# kNN label clean-up
# --------------------------------------------------------------------------------------
def knn_gaussian(kernel_size, sigma):
    """knn.py:11-35: normalised 2-d gaussian, float32."""
    import math
    c = torch.arange(kernel_size)
    xg = c.repeat(kernel_size).view(kernel_size, kernel_size)
    grid = torch.stack([xg, xg.t()], dim=-1).float()
    mean = (kernel_size - 1) / 2.
    var = sigma ** 2.
    g = (1. / (2. * math.pi * var)) * torch.exp(-torch.sum((grid - mean) ** 2., dim=-1) / (2 * var))
    return g / torch.sum(g)


def knn_vote(proj_range, unproj_range, proj_argmax, px, py, search, knn, sigma, cutoff, nclasses):
    """knn.py:56-142 per point, without the unfolded images: window padded with zeros, invalid
    ranges at infinity, centre replaced by the point's own range, weights 1 - gaussian, the
    ``knn`` smallest weighted differences (earlier window position first on ties) vote."""
    h, w = proj_range.shape
    pad = (search - 1) // 2
    ig = (1 - knn_gaussian(search, sigma)).reshape(-1)
    rp = F.pad(proj_range, (pad, pad, pad, pad), value=0.0)
    ap = F.pad(proj_argmax.long(), (pad, pad, pad, pad), value=0)
    offs = [(k // search, k % search) for k in range(search * search)]
    rng = torch.stack([rp[py + dy, px + dx] for dy, dx in offs], 1)           # [P, S*S]
    lab = torch.stack([ap[py + dy, px + dx] for dy, dx in offs], 1)
    rng = torch.where(rng < 0, torch.full_like(rng, float("inf")), rng)
    rng[:, (search * search - 1) // 2] = unproj_range
    dist = (rng - unproj_range[:, None]).abs() * ig[None, :]
    order = torch.sort(dist, dim=1, stable=True)[1][:, :knn]
    kd = torch.gather(dist, 1, order)
    kl = torch.gather(lab, 1, order)
    if cutoff > 0:
        kl = torch.where(kd > cutoff, torch.full_like(kl, nclasses), kl)
    votes = torch.zeros(px.numel(), nclasses + 1)
    votes.scatter_add_(1, kl, torch.ones_like(kl, dtype=votes.dtype))
    return votes[:, 1:-1].argmax(dim=1) + 1


def normalise_input(x, eval_label, mean, std):
    m = (eval_label > 0).unsqueeze(1).to(x.dtype)
    return (x - mean[None, :, None, None]) / std[None, :, None, None] * m


# --------------------------------------------------------------------------------------
# one full training step (restated trainer.py:599-704 with proto path enabled)
# --------------------------------------------------------------------------------------
def select_ratio_for(epoch, n_epochs):
    return float(np.log(1 + (1 + epoch) / n_epochs) / np.log(2) * 0.5)


def train_step(state, x, train_label, eval_label, rng, *, epoch=10, n_epochs=100,
               temperature=0.07, num_anchor=512, w_ce=1.0, w_lov=1.0, w_contrast=0.1,
               focal_alpha=None, momentum=0.999, dropout_masks=None, mean=None, std=None,
               dataset="SemanticKitti", use_prototype=True):
    """Forward + losses + backward of one step; returns (loss dict, grads dict).

    ``rng`` supplies the injected randomness: keys gumbel (dict cls -> noise), pl_noise
    (list), uniforms [T_max,A] f64, perms [C-1,M]; ``rng=None`` draws everything from the
    global torch generator (used by the CPU baseline timing).  Parameters in ``state`` that need
    gradients must be leaf tensors with requires_grad=True."""
    ncls = state["cls_head.weight"].shape[0]
    if rng is None:
        rng = dict(gumbel=None, pl_noise=None, uniforms=None, perms=None)
    wss = train_label > 0
    evm = eval_label > 0
    if mean is not None:
        x = normalise_input(x, eval_label, mean, std)
    out = backbone_forward(state, x, True, dropout_masks, True, dataset)
    prob, feat = out["pred_2d"], out["feat_2d"]
    info = {}
    if use_prototype:
        with torch.no_grad():
            rows, sim, nearest, pl2 = prototype_similarity(state, feat)
            state["prototypes"].copy_(pl2)
            bank, logits, target = prototype_learning(
                pl2, rows, nearest, train_label.reshape(-1), sim, rng["gumbel"], 0, momentum)
            state["prototypes"] = bank
            info["contrast_target"] = target
    if focal_alpha is None:
        focal_alpha = torch.ones(ncls)
        focal_alpha[0] = 0
    l_ce = focal_loss(prob, train_label, wss, focal_alpha)
    l_lov = lovasz_loss(prob, train_label)
    with torch.no_grad():
        ratio = select_ratio_for(epoch, n_epochs)
        lab_c, mask_c = entropy_selection(prob.detach(), wss, evm, train_label, ratio,
                                          rng["pl_noise"])
    l_con = contrast_mem_loss(feat, prob.detach(), lab_c, mask_c, state["prototypes"].detach(),
                              rng["uniforms"], rng["perms"], temperature, num_anchor)
    total = w_ce * l_ce + w_lov * l_lov + w_contrast * l_con
    names = [k for k in trainable_names(state) if state[k].requires_grad]
    grads = torch.autograd.grad(total, [state[k] for k in names], allow_unused=True)
    info.update(loss=total.detach(), ce=l_ce.detach(), lov=l_lov.detach(),
                contrast=l_con.detach(), labels_contra=lab_c, mask_contra=mask_c,
                pred_2d=prob.detach(), feat_2d=feat.detach())
    return info, OrderedDict(zip(names, grads))


def adamw_update(param, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.01):
    """torch.optim.AdamW single-tensor math (trainer.py:146-151 uses defaults)."""
    param.mul_(1 - lr * weight_decay)
    exp_avg.mul_(betas[0]).add_(grad, alpha=1 - betas[0])
    exp_avg_sq.mul_(betas[1]).addcmul_(grad, grad, value=1 - betas[1])
    bc1 = 1 - betas[0] ** step
    bc2 = 1 - betas[1] ** step
    denom = (exp_avg_sq.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(exp_avg, denom, value=-lr / bc1)
    return param
