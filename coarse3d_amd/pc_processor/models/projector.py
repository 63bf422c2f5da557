"""ProjectionV1 (reference pc_processor/models/projector.py:11-27).

conv1x1(C->C) -> BatchNorm -> LeakyReLU -> conv1x1(C->proj_dim); state_dict keys
``proj.0.*, proj.1.*, proj.3.*``.  Inside SalsaNextProto the arithmetic runs in the fused HIP
backbone; called stand-alone this module runs the same HIP kernels on its own input -- forward AND
backward: like the reference's, it is an ordinary trainable ``nn.Module`` (gradients for the input and
for all six parameter tensors through an explicit backward over the engine's conv / BatchNorm ops)."""
from collections import OrderedDict

import torch
import torch.nn as nn

from ... import ops
from ...backbone import Act, Backbone


class _ProjectorFn(torch.autograd.Function):
    """x [B,C,H,W], six parameters -> y [B,proj_dim,H,W]; backward = the engine's explicit conv / BN backward."""

    @staticmethod
    def forward(ctx, module, x, *params):
        names = ("proj.0.weight", "proj.0.bias", "proj.1.weight", "proj.1.bias", "proj.3.weight", "proj.3.bias")
        P = {n: p.detach() for n, p in zip(names, params)}
        bn = module.proj[1]
        P["proj.1.running_mean"], P["proj.1.running_var"] = bn.running_mean, bn.running_var
        bb = Backbone(P)
        bb.train, bb.masks, bb.update_running = module.training, None, True
        bb.tape, bb.bn_seen = OrderedDict(), []
        xin = Act(ops.to_nhwc(x.detach().float()))
        xin.no_grad = not x.requires_grad
        z0 = bb._conv("proj.0", [xin], 1, 1, 0, lrelu=False, bn="proj.1", bn_momentum=bn.momentum)
        y = bb._conv("proj.3", [z0], 1, 1, 0, lrelu=False, src_lrelu=True)
        if module.training and bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
        ctx.bb, ctx.xin, ctx.z0, ctx.names = bb, xin, z0, names
        return ops.from_nhwc(y.t)

    @staticmethod
    def backward(ctx, dy):
        bb, xin, z0 = ctx.bb, ctx.xin, ctx.z0
        if not bb.train:
            raise RuntimeError("ProjectionV1: backward through the eval-mode forward (running statistics) is not supported")
        bb.grads = {n: torch.zeros_like(bb.P[n]) for n in ctx.names}
        bb._conv_backward("proj.3", ops.to_nhwc(dy.float()))
        bb._conv_backward("proj.0", z0.grad)
        bb._join()
        dx = ops.from_nhwc(xin.grad) if xin.grad is not None else None
        ctx.bb = None
        return (None, dx) + tuple(bb.grads[n] for n in ctx.names)


class ProjectionV1(nn.Module):
    def __init__(self, base_channels, proj_dim):
        super().__init__()
        self.proj = nn.Sequential(
            nn.Conv2d(base_channels, base_channels, kernel_size=1),
            nn.BatchNorm2d(base_channels),
            nn.LeakyReLU(),
            nn.Conv2d(base_channels, proj_dim, kernel_size=1),
        )

    def forward(self, x):
        """x [B,C,H,W] -> [B,proj_dim,H,W]."""
        c0, bn, _, c3 = self.proj
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return _ProjectorFn.apply(self, x, c0.weight, c0.bias, bn.weight, bn.bias, c3.weight, c3.bias)
        xs = ops.Source(ops.to_nhwc(x.detach().float()))
        b, h, w, _ = xs.t.shape
        z, part = ops.conv_forward([xs], ops.pack_weights(c0.weight.detach(), 0), c0.bias.detach(),
                                   c0.out_channels, [(0, 0)], stats=self.training)
        if self.training:
            sums = ops.stat_reduce(part, c0.out_channels)
            sc, sh, _, _ = ops.bn_finalize(sums, b * h * w, bn.weight.detach(), bn.bias.detach(),
                                           bn.running_mean, bn.running_var, bn.momentum, bn.eps)
        else:
            sc, sh = ops.bn_eval_affine(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                                        bn.eps)
        y, _ = ops.conv_forward([ops.Source(z, sc, sh, lrelu=True)], ops.pack_weights(c3.weight.detach(), 0),
                                c3.bias.detach(), c3.out_channels, [(0, 0)])
        return ops.from_nhwc(y)
