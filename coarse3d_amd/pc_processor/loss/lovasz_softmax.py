"""Lovasz-softmax loss (reference pc_processor/loss/lovasz_softmax.py:56-68, 101-176; Berman et
al. 2018).  In-step but outside the north-star hot path: stock
PyTorch-ROCm ops.  Unlike the reference's per-class Python loop with a host sync per class,
all classes are evaluated in one batched sort and absent classes are masked on the device."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def lovasz_softmax_flat(probas, labels, classes="present"):
    """probas [P, C], labels [P] -> scalar.  Works class-major ([C, P]) so that the sort and the
    two prefix sums run along the contiguous dimension."""
    if probas.numel() == 0:
        return probas.sum() * 0.0
    p, c = probas.shape
    fg = (labels[None, :] == torch.arange(c, device=labels.device)[:, None]).to(probas.dtype)   # [C, P]
    errors = (fg - probas.t()).abs()
    errors_sorted, perm = torch.sort(errors, dim=1, descending=True)
    fg_sorted = torch.gather(fg, 1, perm)
    gts = fg_sorted.sum(1, keepdim=True)
    inter = gts - fg_sorted.cumsum(1)
    union = gts + (1.0 - fg_sorted).cumsum(1)
    jac = 1.0 - inter / union
    jac = torch.cat((jac[:, :1], jac[:, 1:] - jac[:, :-1]), 1)
    per_class = (errors_sorted * jac).sum(1)
    if classes == "present":
        present = (gts.reshape(-1) > 0).to(probas.dtype)
    elif classes == "all":
        present = torch.ones(c, device=probas.device, dtype=probas.dtype)
    else:
        present = torch.zeros(c, device=probas.device, dtype=probas.dtype)
        present[list(classes)] = 1
    return (per_class * present).sum() / present.sum().clamp(min=1)


def valid_indices(labels, ignore):
    """Flat positions of the pixels with ``labels != ignore`` (the data-dependent size costs one
    host synchronisation).  It depends on the labels only, so a caller that knows the labels
    before the forward pass can compute it early on a side stream and hand it to
    ``Lovasz_softmax.forward(..., valid=...)`` -- the main stream then never drains."""
    return torch.nonzero(labels.reshape(-1) != ignore, as_tuple=False).reshape(-1)


def flatten_probas(probas, labels, ignore=None, valid=None):
    c = probas.shape[1]
    pred = probas.permute(0, 2, 3, 1).reshape(-1, c) if probas.dim() == 4 else probas.transpose(1, 2).reshape(-1, c)
    labels = labels.reshape(-1)
    if ignore is None:
        return pred, labels
    sel = valid_indices(labels, ignore) if valid is None else valid
    return pred[sel], labels[sel]


def lovasz_softmax(probas, labels, classes="present", per_image=False, ignore=None, softmax=False, valid=None):
    if softmax:
        probas = F.softmax(probas, 1)
    if per_image:
        losses = [lovasz_softmax_flat(*flatten_probas(p.unsqueeze(0), l.unsqueeze(0), ignore), classes=classes)
                  for p, l in zip(probas, labels)]
        return sum(losses) / max(len(losses), 1)
    return lovasz_softmax_flat(*flatten_probas(probas, labels, ignore, valid), classes=classes)


class Lovasz_softmax(nn.Module):
    def __init__(self, classes="present", per_image=False, ignore=None, softmax=False):
        super().__init__()
        self.classes, self.per_image, self.ignore, self.softmax = classes, per_image, ignore, softmax

    def forward(self, probas, labels, valid=None):
        return lovasz_softmax(probas, labels, self.classes, self.per_image, self.ignore, self.softmax, valid)
