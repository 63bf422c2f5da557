"""Focal loss on class probabilities (reference pc_processor/loss/focal_softmax.py:7-77).

In-step but outside the north-star hot path: stock PyTorch-ROCm
ops, written sync-free (the NaN guard is a device-side ``where``)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class FocalSoftmaxLoss(nn.Module):
    def __init__(self, n_classes, gamma=1, alpha=0.8, softmax=True):
        super().__init__()
        self.gamma = gamma
        self.n_classes = n_classes
        if isinstance(alpha, (list, np.ndarray)):
            if len(alpha) != n_classes:
                raise AssertionError("len(alpha)!=n_classes: {} vs. {}".format(len(alpha), n_classes))
            self.alpha = torch.as_tensor(np.asarray(alpha), dtype=torch.float32)
        else:
            assert 0 < alpha < 1, "invalid alpha: {}".format(alpha)
            self.alpha = torch.full((n_classes,), 1 - alpha, dtype=torch.float32)
            self.alpha[0] = alpha
        self.softmax = softmax

    def forward(self, x, target, mask=None, pred_log=None):
        if x.dim() > 2:
            pred = x.reshape(x.size(0), x.size(1), -1).transpose(1, 2).reshape(-1, x.size(1))
        else:
            pred = x
        target = target.reshape(-1, 1)
        if self.softmax:
            pred = F.softmax(pred, 1)
        pt = pred.gather(1, target).reshape(-1)
        self.alpha = self.alpha.to(x.device)
        loss = -(1 - pt).pow(self.gamma) * pt.clamp(1e-6).log() * self.alpha.gather(0, target.reshape(-1))
        if mask is None:
            return loss.mean()
        m = mask.reshape(-1).to(loss.dtype)
        den = m.sum()
        # reference: 0/0 -> NaN -> "return torch.tensor(0.0)".  Dividing by max(den, 1) gives the
        # same 0 without a NaN ever entering the autograd graph (a where() would leak NaN grads)
        out = (loss * m).sum() / den.clamp(min=1.0)
        return torch.where(torch.isnan(out.detach()), torch.zeros_like(out), out)
