"""ProjectionV1 parameter container (reference pc_processor/models/projector.py:11-27).

conv1x1(C->C) -> BatchNorm -> LeakyReLU -> conv1x1(C->proj_dim); state_dict keys
``proj.0.*, proj.1.*, proj.3.*``.  Inside SalsaNextProto the arithmetic runs in the fused HIP
backbone; called stand-alone this module runs the same HIP kernels on its own input."""
import torch.nn as nn

from ... import ops


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
        """x [B,C,H,W] -> [B,proj_dim,H,W] (inference-style call: no autograd graph)."""
        c0, bn, _, c3 = self.proj
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
