"""Pixel-to-prototype contrastive loss and entropy-based pseudo-label selection on HIP kernels.

Reference: pc_processor/loss/contrast_pixel_loss.py:27-195 (``ContrastMEMLoss``) and
tasks/weak_segmentation/trainer.py:447-518 (``entropy_based_selection``).  Everything is
shape-static and free of host synchronisation: absent (image, class) pairs are masked on the
device instead of being skipped by Python loops."""
import numpy as np
import weakref

import torch

from . import ops


def _nhwc_rows(t_nchw_like):
    """[B,C,H,W]-shaped tensor (any strides) -> contiguous [B,H,W,C]."""
    return t_nchw_like.permute(0, 2, 3, 1).contiguous()


def entropy_selection(prob_nhwc, train_label, eval_label, select_ratio, noise=None, ignore_cls=0):
    """prob [B,H,W,C]; labels [B,H,W] int64.  noise: Exp(1) [B,C,H*W] or None (drawn on device).
    select_ratio: Python float or an fp32 device scalar (read by the kernel: nothing epoch-dependent in the launch).
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
                                 select_ratio if isinstance(select_ratio, torch.Tensor) else np.float32(select_ratio))
    return labels.view(b, h, w), mask.view(b, h, w)


# ---- row-sparsity hint for the gradient of the embedding
# d(loss)/d(feats) is a dense [B,H,W,D] tensor (1 GB at the headline shape) with ~10^3 non-zero pixel rows of 10^6.
# The tensor handed to autograd stays dense and complete; next to it the backward publishes a bitmap of the rows it
# wrote.  A consumer that receives THIS tensor object (a view whose base it is) may skip rows whose bit is clear --
# the backbone's bilinear adjoint does (its 1 GB read drops to the marked rows).  One slot, consumed on first use, held
# by weak reference: a gradient that autograd summed with another one, copied, or that comes from elsewhere never
# matches.  (Module flag: the tests switch it off to compare.)
SPARSE_HINT_ON = True
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


# ---- the embedding without its last F.interpolate
# salsanext_proto.py:485-490 l2-normalises the projector's output at half resolution and upsamples it to the label
# resolution: ``feat_2d``, 1.07 GB at 8x256x64x2048.  Inside the training step only ROWS of it are read -- the sampled
# anchors of the contrast loss (contrast_pixel_loss.py) and the labelled pixels of prototype_learning -- and its
# gradient has ~10^3 non-zero pixel rows.  A LowResFeat carries the half-resolution tensor (autograd-connected) and the
# target size; consumers that understand it interpolate the rows they need (ops.bilinear_rows: bit-identical to the
# rows of the dense map) and send the gradient back in compact form (ops.scatter_rows_compact +
# ops.bilinear_bwd_rows: bit-identical to the dense adjoint); everybody else calls ``dense()`` and gets the
# reference's tensor.  (Module flag: False makes the model hand out the dense tensor again; the tests compare both.)
LAZY_FEAT_ON = True


class _UpsampleFn(torch.autograd.Function):
    """F.interpolate(low, size, mode="bilinear", align_corners=True) of a channels-last [B,D,h,w] view, in fp32."""

    @staticmethod
    def forward(ctx, low, hd, wd):
        ctx.low_meta = (tuple(low.shape), low.dtype)
        return ops.bilinear(_nhwc_rows(low.detach()), hd, wd, out_dtype=torch.float32).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        (b, d, h, w), dtype = ctx.low_meta
        d_low = torch.empty(b, h, w, d, device=g.device, dtype=dtype)
        gn = g.permute(0, 2, 3, 1)
        hint = take_row_hint(gn if gn.is_contiguous() else None)
        ops.bilinear_bwd(d_low, gn.contiguous().float(), rowmask=hint)
        return d_low.permute(0, 3, 1, 2), None, None


class LowResFeat:
    def __init__(self, low, size):
        self.low = low                       # [B, D, h, w]-shaped view of channels-last memory, requires grad in training
        self.size = (int(size[0]), int(size[1]))
        self._dense = None

    @property
    def shape(self):                          # of the tensor it stands for
        return torch.Size((self.low.shape[0], self.low.shape[1]) + self.size)

    @property
    def device(self):
        return self.low.device

    def low_nhwc(self):
        return _nhwc_rows(self.low.detach())

    def dense(self):
        """The reference's ``feat_2d`` [B, D, H, W] (fp32, autograd-connected); computed once."""
        if self._dense is None:
            self._dense = _UpsampleFn.apply(self.low, *self.size)
        return self._dense

    def detach(self):
        return self.dense().detach()


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
        return _contrast_backward(ctx.state, g).permute(0, 3, 1, 2), None


def _contrast_backward(st, g):
    """d(loss * g) / d(feats) as an NHWC tensor: of the half-resolution embedding (``st["low"]``: lazy form) or dense."""
    b, h, w, d = st["bhwd"]
    n = h * w
    tmax, a = st["tmax"], st["a"]
    # d(anchor rows, normalised) = dlogits x queue ; then through the l2 normalisation
    da = ops.gemm_rows(st["dlogits"], st["wq_t"], d)
    dx = ops.l2norm_bwd(st["anchors"], st["norm"], da)
    gs = g.reshape(1).to(torch.float32).contiguous()
    if st["low"] is not None:
        # gradient of the half-resolution embedding straight from the anchor rows: owner sums in a compact buffer,
        # then the same gather-form adjoint the dense path runs over a zero-filled [B,H,W,D] tensor
        low = st["low"]
        drows, cmap, rowmask = ops.scatter_rows_compact(dx, st["img"], st["idx"], st["T"], tmax, a, n, b, gs)
        d_low = torch.empty(b, low.shape[2], low.shape[3], d, device=dx.device, dtype=low.dtype)
        ops.bilinear_bwd_rows(d_low, drows, cmap, rowmask, h, w)
        return d_low
    dfeat = torch.zeros(b, h, w, d, device=dx.device, dtype=torch.float32)
    # one bit per pixel that received a row: a hint for whoever consumes this (dense, complete) gradient next
    rowmask = torch.zeros((b * n + 31) // 32, device=dx.device, dtype=torch.int32) if SPARSE_HINT_ON else None
    ops.scatter_add_rows(dx, st["img"], st["idx"], st["T"], tmax, a, n, dfeat, gs, rowmask=rowmask)
    if rowmask is not None:
        _publish_row_hint(dfeat, rowmask)
    return dfeat


def contrast_mem_loss(feats, prob, labels, keep_mask, proto_queue, temperature=0.1, base_temperature=0.07,
                      num_anchor=50, ignore_label=0, uniforms=None, perms=None, return_debug=False, explicit_grad_scale=None):
    """feats [B,D,H,W]-shaped (channels-last memory preferred) or a LowResFeat, prob [B,C,H,W]-shaped, labels
    [B,H,W] int64, keep_mask [B,H,W] bool or None, proto_queue [C,M,D].

    uniforms: float64 [B*C, A] draws (row t feeds the t-th present (image,class) pair, exactly
    the stream torch.multinomial would consume); perms: int64 [C-1, M] queue row orders.
    Returns the 0-dim loss (autograd-connected to ``feats``).
    explicit_grad_scale (a LowResFeat only): no autograd node -- returns (loss, d(explicit_grad_scale * loss) / d(feats.low) as an
    NHWC tensor), the same kernels the autograd path runs in its backward (coarse3d_amd.trainer.TrainStep hands that
    gradient to the backbone's backward itself, from a second stream)."""
    b, d, h, w = feats.shape
    c = prob.shape[1]
    n = h * w
    dev = feats.device
    lazy = isinstance(feats, LowResFeat)
    feat = None if lazy else _nhwc_rows(feats.detach())
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
    if lazy:
        anchors, norm = ops.bilinear_rows(feats.low_nhwc(), h, w, a_idx, img=a_img, a=a, count=t, l2=True)
    else:
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
    state = dict(bhwd=(b, h, w, d), low=feats.low if lazy else None, anchors=anchors, norm=norm, dlogits=logits,
                 wq_t=ops.pack_weights(w_q, 1), img=a_img, idx=a_idx, T=t, tmax=tmax, a=a, loss=loss)
    if explicit_grad_scale is not None:
        if not lazy or return_debug:
            raise ValueError("contrast_mem_loss(explicit_grad_scale=...): a LowResFeat embedding, no debug output")
        gs = torch.full((1,), float(explicit_grad_scale), device=dev, dtype=torch.float32)
        return loss.reshape(()), _contrast_backward(state, gs)
    out = _ContrastFn.apply(feats.low if lazy else feats, state)
    if return_debug:
        return out, dict(idx=a_idx, img=a_img, cls=a_cls, T=t, row_loss=row_loss, weights=w_anchor, counts=counts,
                         candidates=idx, uniforms=uniforms)
    return out
