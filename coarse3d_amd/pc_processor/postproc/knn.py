"""kNN label clean-up on the device (reference pc_processor/postproc/knn.py:11-142; same class
name, ``params`` keys and ``forward`` signature).  The reference materialises the unfolded
[1, S*S, H*W] range and label images and indexes them per point; here one thread per point scans
its S x S window directly (csrc/metric_ops.hip, ``c3d_knn_vote``)."""
import math

import torch
import torch.nn as nn

from ... import ops


def get_gaussian_kernel(kernel_size=3, sigma=2, channels=1):
    """Normalised 2-d gaussian [kernel_size, kernel_size] in float32, evaluated with the same
    torch operations as knn.py:11-35 so that the weights agree to the last bit."""
    coord = torch.arange(kernel_size)
    xg = coord.repeat(kernel_size).view(kernel_size, kernel_size)
    grid = torch.stack([xg, xg.t()], dim=-1).float()
    mean = (kernel_size - 1) / 2.
    variance = sigma ** 2.
    g = (1. / (2. * math.pi * variance)) * torch.exp(-torch.sum((grid - mean) ** 2., dim=-1) / (2 * variance))
    return (g / torch.sum(g)).view(kernel_size, kernel_size)


class KNN(nn.Module):
    def __init__(self, params, nclasses):
        super().__init__()
        self.knn = params["knn"]
        self.search = params["search"]
        self.sigma = params["sigma"]
        self.cutoff = params["cutoff"]
        self.nclasses = nclasses
        if self.search % 2 == 0:
            raise ValueError("Nearest neighbor kernel must be odd number")

    def forward(self, proj_range, unproj_range, proj_argmax, px, py):
        """proj_range [H,W] float32, unproj_range [P], proj_argmax [H,W] integer labels, px/py [P]
        pixel coordinates (one un-batched scan, as the reference).  Returns int64 [P]."""
        if not proj_range.is_cuda:
            raise RuntimeError("KNN runs on the GPU (HIP kernel); there is no CPU fallback")
        inv_gauss = (1 - get_gaussian_kernel(self.search, self.sigma, 1)).reshape(-1).to(proj_range.device)
        return ops.knn_vote(proj_range, proj_argmax, unproj_range, px, py, inv_gauss, self.search, self.knn,
                            self.cutoff, self.nclasses)
