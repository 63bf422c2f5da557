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
    def forward(ctx, pred, target, mask_u8, alpha, gamma, idx, want_focal, want_lovasz):
        ctx.args = (target, mask_u8, alpha, gamma, idx)
        f_stats = l_stats = l_grad = None
        if want_focal:
            f_stats = ops.focal_forward(pred, target, mask_u8, alpha, gamma)
        if want_lovasz:
            l_stats, l_grad = ops.lovasz_forward(pred, target, idx)
        ctx.saved = (pred, f_stats, l_stats, l_grad)
        zero = pred.new_zeros(())
        return (f_stats[0] if want_focal else zero), (l_stats[0] if want_lovasz else zero)

    @staticmethod
    def backward(ctx, g_focal, g_lovasz):
        pred, f_stats, l_stats, l_grad = ctx.saved
        target, mask_u8, alpha, gamma, idx = ctx.args
        b, c, h, w = pred.shape
        dprob = torch.zeros(b, h, w, c, device=pred.device, dtype=torch.float32)
        if f_stats is not None and g_focal is not None:
            ops.focal_backward(pred, target, mask_u8, alpha, gamma, f_stats, g_focal.reshape(1).float().contiguous(), dprob)
        if l_stats is not None and g_lovasz is not None:
            ops.lovasz_backward(l_grad, idx, l_stats, g_lovasz.reshape(1).float().contiguous(), dprob)
        return dprob.permute(0, 3, 1, 2), None, None, None, None, None, None, None


def fused_available(n_labelled):
    return n_labelled <= ops.lovasz_max_pixels()


def loss_head(pred, target, mask, alpha, gamma, idx, want_focal=True, want_lovasz=True):
    """pred [B,C,H,W]-shaped probabilities, target int64 [B,H,W], mask bool [B,H,W] (focal),
    idx int64 [P] flat positions with target != ignore (Lovasz).  Returns (focal, lovasz)."""
    m = mask.to(torch.uint8).contiguous() if mask is not None else None
    return _LossHeadFn.apply(pred, target.contiguous(), m, alpha, float(gamma), idx, want_focal, want_lovasz)
