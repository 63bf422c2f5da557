"""Focal + Lovasz loss head on the HIP kernels of csrc/metric_ops.hip.

Reference: pc_processor/loss/focal_softmax.py:30-77 and pc_processor/loss/lovasz_softmax.py:101-176
as the trainer combines them (tasks/weak_segmentation/trainer.py:640-652).  Both losses only touch
the labelled pixels; their gradients are written straight into ONE dense d(pred) buffer (NHWC,
handed to the backbone's softmax backward as a channels-last view) instead of ~40 stock-op
launches and their intermediate [N, C] tensors."""
import torch

from . import ops


class _LossHeadFn(torch.autograd.Function):
    """pred [B,C,H,W]-shaped probabilities -> (focal, lovasz) 0-dim losses."""

    @staticmethod
    def forward(ctx, pred, target, mask_u8, alpha, gamma, idx, want_focal, want_lovasz, count=None):
        ctx.args = (target, mask_u8, alpha, gamma, idx, count)
        f_stats = l_stats = l_grad = None
        if want_focal:
            f_stats = ops.focal_forward(pred, target, mask_u8, alpha, gamma)
        if want_lovasz:
            if count is None:
                l_stats, l_grad = ops.lovasz_forward(pred, target, idx)
            else:          # shape-static form: idx holds a fixed capacity, the number of entries lives on the device
                l_stats, l_grad = ops.lovasz_forward_dyn(pred, target, idx, count)
        ctx.saved = (pred, f_stats, l_stats, l_grad)
        zero = pred.new_zeros(())
        return (f_stats[0] if want_focal else zero), (l_stats[0] if want_lovasz else zero)

    @staticmethod
    def backward(ctx, g_focal, g_lovasz):
        pred, f_stats, l_stats, l_grad = ctx.saved
        target, mask_u8, alpha, gamma, idx, count = ctx.args
        b, c, h, w = pred.shape
        dprob = torch.zeros(b, h, w, c, device=pred.device, dtype=torch.float32)
        if f_stats is not None and g_focal is not None:
            ops.focal_backward(pred, target, mask_u8, alpha, gamma, f_stats, g_focal.reshape(1).float().contiguous(), dprob)
        if l_stats is not None and g_lovasz is not None:
            gs = g_lovasz.reshape(1).float().contiguous()
            if count is None:
                ops.lovasz_backward(l_grad, idx, l_stats, gs, dprob)
            else:
                ops.lovasz_backward_dyn(l_grad, idx, count, l_stats, gs, dprob)
        return dprob.permute(0, 3, 1, 2), None, None, None, None, None, None, None, None


def fused_available(n_labelled):
    """The fused head handles any number of labelled pixels: up to ``ops.lovasz_max_pixels()`` the per-class sort runs in
    LDS, beyond it as a device-wide segmented sort (as long as classes x pixels stays below 2^31)."""
    return n_labelled < (1 << 31) // 64


def loss_head(pred, target, mask, alpha, gamma, idx, want_focal=True, want_lovasz=True, count=None):
    """pred [B,C,H,W]-shaped probabilities, target int64 [B,H,W], mask bool [B,H,W] (focal),
    idx int64 [P] flat positions with target != ignore (Lovasz).  ``count`` (int32 [1] on the device): idx is a
    fixed-capacity buffer whose first ``count`` entries are valid (see ``valid_indices_static``).
    Returns (focal, lovasz)."""
    m = mask.to(torch.uint8).contiguous() if mask is not None else None
    return _LossHeadFn.apply(pred, target.contiguous(), m, alpha, float(gamma), idx, want_focal, want_lovasz, count)


def valid_indices_static(labels, ignore=0):
    """Device-side, shape-static replacement of ``torch.nonzero(labels != ignore)`` (lovasz_softmax.py:140-160 indexes
    with a boolean mask, which costs a host synchronisation): ordered compaction into a buffer of
    ``ops.lovasz_max_pixels()`` entries plus the count as a device scalar.  Returns (idx int64 [cap], count int32 [1]).
    More labelled pixels than the capacity are NOT representable here -- the caller checks the count once per epoch
    (TrainStep does, before it captures a graph)."""
    mask = labels.reshape(-1) != ignore
    cap = min(ops.lovasz_max_pixels(), mask.numel())
    # ordered, shape-static compaction without a host synchronisation (torch's own select primitive; entries past the
    # count are 0 and never read).  ~20 us for 10^6 labels; the two-class c3d_group_compact it replaced took ~190 us
    # on one group
    idx = torch.nonzero_static(mask, size=cap, fill_value=0).reshape(-1)
    return idx, mask.sum(dtype=torch.int32).reshape(1)
