"""Prototype memory-bank path on the HIP kernels (no autograd: the reference runs it under
``requires_grad=False`` parameters and never back-propagates through it).

Reference: pc_processor/models/salsanext_proto.py:494-530 (similarity + LayerNorm) and
:337-402 (prototype_learning with Sinkhorn, pc_processor/models/sinkhorn.py:5-33)."""
import torch

from . import ops


def l2_bank(protos):
    """Row-wise l2 normalisation of the [C, M, D] bank (salsanext_proto.py:502)."""
    out, _ = ops.l2norm(protos.contiguous(), 1e-12, want_norm=False)
    return out


def similarity(feat_nhwc, bank_l2, ln_w, ln_b):
    """feat [B,H,W,D] -> (rows [N,D] = l2(LN(feat)), sim [N, M*C] with column m*C + k)."""
    b, h, w, d = feat_nhwc.shape
    n = b * h * w
    c, m, _ = bank_l2.shape
    rows = ops.rownorm_ln_l2(feat_nhwc.view(n, d), ln_w, ln_b)
    # GEMM weight in "OIHW": output column m*C+k <- prototype (k, m)
    w_oihw = bank_l2.permute(1, 0, 2).reshape(m * c, d, 1, 1).contiguous()
    wp = ops.pack_weights(w_oihw, 0)
    npad = (n + 31) // 32 * 32
    if npad != n:
        rows_p = torch.zeros(npad, d, device=rows.device, dtype=torch.float32)
        rows_p[:n] = rows
        sim = ops.gemm_rows(rows_p, wp, m * c)[:n]
    else:
        sim = ops.gemm_rows(rows, wp, m * c)
    return rows, sim


class LazyOutputs(dict):
    """The forward's output dict with entries that are computed on first access (``lazy[key] = thunk``).  Used for
    ``contrast_logits``: the [N, C*M] similarity map (1.7 GB at 8x64x2048) is part of the module's output surface
    (salsanext_proto.py:529) but nothing inside the training step reads it -- the prototype update only needs the rows
    of the labelled pixels.  Behaves as a dict everywhere else (``in``, ``keys``, ``items``, ``get`` materialise as needed)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.lazy = {}

    def _force(self, key):
        if key in self.lazy:
            super().__setitem__(key, self.lazy.pop(key)())

    def __getitem__(self, key):
        self._force(key)
        return super().__getitem__(key)

    def get(self, key, default=None):
        self._force(key)
        return super().get(key, default)

    def __contains__(self, key):
        return key in self.lazy or super().__contains__(key)

    def keys(self):
        return list(super().keys()) + list(self.lazy)

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return super().__len__() + len(self.lazy)

    def items(self):
        for k in list(self.lazy):
            self._force(k)
        return super().items()

    def values(self):
        for k in list(self.lazy):
            self._force(k)
        return super().values()


def prototype_step(feat_nhwc, P, label, proto_loss, noise=None, momentum=0.999, ignore_label=0,
                   world_mean=None, want_nearest=False, ema_base=None, sums_reduce=None, labelled=None):  # label: [B, H*W] int64
    """One pass of salsanext_proto.py:494-530.

    P: dict with ``prototypes`` [C,M,D], ``feat_norm.*``, ``mask_norm.*``.  label: [N] int64 or
    None.  noise: Exp(1) variates [N, M] indexed by pixel (None -> drawn on device).
    labelled: None, or (idx int64 [cap], count int32 [1]) -- the flat positions of the pixels with label != ignore in a
    fixed-capacity buffer (``loss_head.valid_indices_static``; the caller guarantees count <= cap).  Then LayerNorm + l2
    and the similarity GEMM run on THOSE rows only ("never materialise [N, 400]") and ``contrast_logits``
    is a thunk that computes the full map if somebody asks for it.
    Returns dict(bank_l2, [nearest], [contrast_logits, contrast_target, new_bank])."""
    bank = P["prototypes"]
    c, m, d = bank.shape
    bank_l2 = l2_bank(bank)
    learn = proto_loss and label is not None
    if not (learn or want_nearest):
        # the similarity map is unobservable in this mode (the reference computes and drops it);
        # the only side effect is the in-place renormalisation of the bank
        return {"bank_l2": bank_l2, "nearest": None, "pred": None}
    handle = feat_nhwc if hasattr(feat_nhwc, "low_nhwc") else None        # contrast.LowResFeat: rows on demand
    b = feat_nhwc.shape[0]
    n = feat_nhwc.shape.numel() // d
    sparse = learn and labelled is not None and not want_nearest
    if handle is not None:
        dense = lambda: handle.dense().detach().permute(0, 2, 3, 1).contiguous()   # noqa: E731
        if not sparse:
            feat_nhwc = dense()
    else:
        dense = lambda: feat_nhwc   # noqa: E731
    full_sim = lambda: similarity(dense(), bank_l2, P["feat_norm.weight"], P["feat_norm.bias"])[1]   # noqa: E731
    cmap = None
    if sparse:
        idx, cnt = labelled
        cap = idx.numel()
        ar = torch.arange(cap, device=idx.device)
        ok = ar < cnt
        if handle is not None:
            g = ops.bilinear_rows(handle.low_nhwc(), handle.size[0], handle.size[1], idx, count=cnt)[:cap]   # [cap, D]
        else:
            g = feat_nhwc.view(n, d).index_select(0, torch.where(ok, idx, torch.zeros_like(idx)))       # [cap, D]
        rows = ops.rownorm_ln_l2(g, P["feat_norm.weight"], P["feat_norm.bias"])
        w_oihw = bank_l2.permute(1, 0, 2).reshape(m * c, d, 1, 1).contiguous()
        cpad = (cap + 31) // 32 * 32
        if cpad != cap:
            rp = torch.zeros(cpad, d, device=rows.device, dtype=torch.float32)
            rp[:cap] = rows
            rows = rp
        sim = ops.gemm_rows(rows, ops.pack_weights(w_oihw, 0), m * c)
        # pixel -> compact row; slot n swallows the padding entries.  Zero-filled: a labelled pixel BEYOND the capacity
        # (the caller's overflow check raises for it, a step or two later in a captured run) reads row 0 -- wrong
        # numbers for a step that is about to be refused, never an out-of-range row
        cmap = torch.zeros(n + 1, device=idx.device, dtype=torch.int32)
        cmap.scatter_(0, torch.where(ok, idx, torch.full_like(idx, n)), ar.to(torch.int32))
    else:
        rows, sim = similarity(feat_nhwc, bank_l2, P["feat_norm.weight"], P["feat_norm.bias"])
    # nearest_proto_distance (:506-510) is only ever consumed through its argmax at LABELLED
    # pixels (prototype_learning :340-341): the full [N, C] map is materialised on request only
    nearest = pred = None
    if want_nearest:
        nearest, pred = ops.proto_nearest(sim, m, c, P["mask_norm.weight"], P["mask_norm.bias"], want_nearest=True)
    out = {"bank_l2": bank_l2, "nearest": nearest, "pred": pred}
    if learn:
        lab = label.reshape(b, n // b).contiguous()
        counts, idx_lists = ops.group_compact(lab, c)
        by_row = noise is None and cmap is not None
        if noise is None:
            # (compact rows: variates for those rows only -- the kernel reads noise at labelled pixels and nowhere else)
            noise = torch.empty(rows.shape[0] if by_row else n, m, device=rows.device, dtype=torch.float32).exponential_()
        base = bank_l2 if ema_base is None else ema_base.contiguous()   # proto_pl replaces the bank (:515-518)
        new_bank, target = ops.proto_learn(sim, rows, pred, counts, idx_lists, noise.contiguous(),
                                           base, m, c, ignore_label, momentum, P["mask_norm.weight"],
                                           P["mask_norm.bias"], sums_reduce=sums_reduce, cmap=cmap, noise_by_row=by_row)
        if world_mean is not None and sums_reduce is None:
            # data parallel, reference semantics: mean over ranks of the per-rank updated banks
            # (salsanext_proto.py:397-400); with ``sums_reduce`` the ranks already agree
            new_bank = world_mean(new_bank)
        out.update(contrast_target=target, new_bank=new_bank)
        out["contrast_logits"] = full_sim if sparse else sim        # a thunk in the sparse mode
    return out
