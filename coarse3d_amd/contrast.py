"""Pixel-to-prototype contrastive loss and entropy-based pseudo-label selection on HIP kernels.

Reference: pc_processor/loss/contrast_pixel_loss.py:27-195 (``ContrastMEMLoss``) and
tasks/weak_segmentation/trainer.py:447-518 (``entropy_based_selection``).  Everything is
shape-static and free of host synchronisation: absent (image, class) pairs are masked on the
device instead of being skipped by Python loops."""
import numpy as np
import os
import weakref

import torch

from . import ops


def _nhwc_rows(t_nchw_like):
    """[B,C,H,W]-shaped tensor (any strides) -> contiguous [B,H,W,C]."""
    return t_nchw_like.permute(0, 2, 3, 1).contiguous()


def entropy_selection(prob_nhwc, train_label, eval_label, select_ratio, noise=None, ignore_cls=0):
    """prob [B,H,W,C]; labels [B,H,W] int64.  noise: Exp(1) [B,C,H*W] or None (drawn on device).
    Returns (labels [B,H,W] int64, mask [B,H,W] bool)."""
    b, h, w, c = prob_nhwc.shape
    n = h * w
    _, w_pl, amax = ops.entropy_stats(prob_nhwc, want_anchor=False)
    tl = train_label.reshape(b, n).contiguous()
    ev = eval_label.reshape(b, n).contiguous()
    tl_counts = ops.label_hist(tl, c)
    if noise is None:
        noise = torch.empty(b, c, n, device=prob_nhwc.device, dtype=torch.float32).exponential_()
    labels, mask = ops.pl_select(w_pl, amax, ev, tl, noise.contiguous(), tl_counts, b, n, c, ignore_cls,
                                 np.float32(select_ratio))
    return labels.view(b, h, w), mask.view(b, h, w)


# ---- row-sparsity hint for the gradient of the embedding
# d(loss)/d(feats) is a dense [B,H,W,D] tensor (1 GB at the headline shape) with ~10^3 non-zero pixel rows of 10^6.
# The tensor handed to autograd stays dense and complete; next to it the backward publishes a bitmap of the rows it
# wrote.  A consumer that receives THIS tensor object (a view whose base it is) may skip rows whose bit is clear --
# the backbone's bilinear adjoint does (its 1 GB read drops to the marked rows).  One slot, consumed on first use, held
# by weak reference: a gradient that autograd summed with another one, copied, or that comes from elsewhere never
# matches.  C3D_SPARSE_DFEAT=0 switches it off.
SPARSE_HINT_ON = os.environ.get("C3D_SPARSE_DFEAT", "1") != "0"
_row_hint = None


def _publish_row_hint(dfeat, rowmask):
    global _row_hint
    _row_hint = (weakref.ref(dfeat), dfeat._version, rowmask)


def take_row_hint(t):
    """Bitmap of the non-zero pixel rows of ``t`` if ``t`` is (a view of) the gradient the contrast loss produced in
    this backward pass and nothing has written to it since; else None.  Clears the slot."""
    global _row_hint
    hint, _row_hint = _row_hint, None
    if hint is None or t is None:
        return None
    ref, version, rowmask = hint
    src = ref()
    if src is None:
        return None
    base = t._base if t._base is not None else t
    if base is not src or t._version != version or t.data_ptr() != src.data_ptr() or t.numel() != src.numel():
        return None
    return rowmask


class _ContrastFn(torch.autograd.Function):
    """loss = InfoNCE(anchors sampled from feats, prototype queue); d(loss)/d(feats) is a sparse
    scatter-add of at most B*(C-1)*A rows (the reference zero-fills a dense [B,HW,D] tensor per
    (image, class) through SelectBackward)."""

    @staticmethod
    def forward(ctx, feats, state):
        ctx.state = state
        return state["loss"].reshape(())

    @staticmethod
    def backward(ctx, g):
        st = ctx.state
        feat = st["feat"]
        b, h, w, d = feat.shape
        n = h * w
        tmax, a = st["tmax"], st["a"]
        # d(anchor rows, normalised) = dlogits x queue ; then through the l2 normalisation
        da = ops.gemm_rows(st["dlogits"], st["wq_t"], d)
        dx = ops.l2norm_bwd(st["anchors"], st["norm"], da)
        dfeat = torch.zeros(b, h, w, d, device=feat.device, dtype=torch.float32)
        gs = g.reshape(1).to(torch.float32).contiguous()
        # one bit per pixel that received a row: a hint for whoever consumes this (dense, complete) gradient next
        rowmask = torch.zeros((b * n + 31) // 32, device=feat.device, dtype=torch.int32) if SPARSE_HINT_ON else None
        ops.scatter_add_rows(dx, st["img"], st["idx"], st["T"], tmax, a, n, dfeat, gs, rowmask=rowmask)
        if rowmask is not None:
            _publish_row_hint(dfeat, rowmask)
        return dfeat.permute(0, 3, 1, 2), None


def contrast_mem_loss(feats, prob, labels, keep_mask, proto_queue, temperature=0.1, base_temperature=0.07,
                      num_anchor=50, ignore_label=0, uniforms=None, perms=None, return_debug=False):
    """feats [B,D,H,W]-shaped (channels-last memory preferred), prob [B,C,H,W]-shaped, labels
    [B,H,W] int64, keep_mask [B,H,W] bool or None, proto_queue [C,M,D].

    uniforms: float64 [B*C, A] draws (row t feeds the t-th present (image,class) pair, exactly
    the stream torch.multinomial would consume); perms: int64 [C-1, M] queue row orders.
    Returns the 0-dim loss (autograd-connected to ``feats``)."""
    b, d, h, w = feats.shape
    c = prob.shape[1]
    n = h * w
    dev = feats.device
    feat = _nhwc_rows(feats.detach())
    p = _nhwc_rows(prob.detach())
    w_anchor, _, _ = ops.entropy_stats(p, want_pl=False, want_amax=False)
    lab = labels.reshape(b, n).contiguous()
    keep = keep_mask.reshape(b, n).to(torch.uint8).contiguous() if keep_mask is not None else None
    counts, idx = ops.group_compact(lab, c, keep)
    tmax = b * c
    a = num_anchor
    if uniforms is None:
        uniforms = torch.rand(tmax, a, device=dev, dtype=torch.float64)
    else:
        u = torch.zeros(tmax, a, device=dev, dtype=torch.float64)
        u[: uniforms.shape[0]] = uniforms.to(dev)
        uniforms = u
    a_idx, a_img, a_cls, t = ops.anchor_sample(w_anchor, counts, idx, uniforms, b, n, c, a, ignore_label)
    anchors, norm = ops.gather_rows_l2(feat, a_img, a_idx, t, tmax, a, n)
    # queue: classes 1..C-1, rows permuted, l2-normalised, padded to a multiple of 16 rows
    m = proto_queue.shape[1]
    q = proto_queue.detach()[1:]
    if perms is None:
        # one uniformly random row order per class: argsort of iid keys, two launches instead of the
        # ~100 tiny ones of C-1 torch.randperm calls (contrast_loss.py draws them with randperm; the
        # parity tests inject ``perms``)
        perms = torch.rand(c - 1, m, device=dev).argsort(dim=1)
    q = torch.gather(q, 1, perms.to(dev)[:, :, None].expand(-1, -1, d)).reshape((c - 1) * m, d).contiguous()
    qn, _ = ops.l2norm(q, 1e-12, want_norm=False)
    ncols = (c - 1) * m
    ld = (ncols + 15) // 16 * 16
    qpad = torch.zeros(ld, d, device=dev, dtype=torch.float32)
    qpad[:ncols] = qn
    w_q = qpad.view(ld, d, 1, 1)
    logits = ops.gemm_rows(anchors, ops.pack_weights(w_q, 0), ld)          # [tmax*a, ld] cosine
    loss, row_loss = ops.infonce_rows(logits, a_cls, t, tmax, a, m, ncols, temperature, base_temperature)
    state = dict(feat=feat, anchors=anchors, norm=norm, dlogits=logits, wq_t=ops.pack_weights(w_q, 1),
                 img=a_img, idx=a_idx, T=t, tmax=tmax, a=a, loss=loss)
    out = _ContrastFn.apply(feats, state)
    if return_debug:
        return out, dict(idx=a_idx, img=a_img, cls=a_cls, T=t, row_loss=row_loss, weights=w_anchor, counts=counts,
                         candidates=idx, uniforms=uniforms)
    return out
