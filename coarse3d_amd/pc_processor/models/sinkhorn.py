"""Sinkhorn-Knopp prototype assignment (reference pc_processor/models/sinkhorn.py:5-33).

The training path runs the fused per-class kernel (``c3d_proto_learn``); this function keeps
the reference's stand-alone signature for callers that want the assignment of one score matrix.
It uses the same kernel with a single class."""
import torch

from ... import ops


def distributed_sinkhorn(out, sinkhorn_iterations=3, epsilon=0.05, noise=None):
    """out [n, K] similarity scores -> (Q one-hot [n, K] from a Gumbel-hard draw, argmax [n]).
    ``noise`` (not in the reference signature): the Exp(1) variates [n, K] that
    F.gumbel_softmax would draw (sinkhorn.py:31), for replaying a recorded reference run."""
    if sinkhorn_iterations != 3 or abs(epsilon - 0.05) > 1e-12:
        raise ValueError("the HIP kernel implements the reference defaults (3 iterations, eps 0.05)")
    n, k = out.shape
    dev = out.device
    sim = out.detach().float().contiguous()                # [n, K*1], class 0 of 1
    rows = torch.zeros(n, 4, device=dev)
    pred = torch.zeros(n, device=dev, dtype=torch.int32)
    counts = torch.tensor([[n]], device=dev, dtype=torch.int32)
    idx = torch.arange(n, device=dev, dtype=torch.int32).view(1, 1, n)
    noise = torch.empty(n, k, device=dev).exponential_() if noise is None else noise.to(dev).float().contiguous()
    bank = torch.zeros(1, k, 4, device=dev)
    _, target = ops.proto_learn(sim, rows, pred, counts, idx, noise, bank, k, 1, -1, 0.999)   # pred given
    index = target.long()
    # Gumbel-hard draw from the same noise, on the Sinkhorn-normalised scores
    e = torch.exp(sim / epsilon)
    u = torch.ones(k, device=dev)
    v = torch.ones(n, device=dev)
    for _ in range(sinkhorn_iterations):
        u = 1.0 / (k * (e * v[:, None]).sum(0))
        v = 1.0 / (e * u[None, :]).sum(1)
    q = e * u[None, :] * v[:, None]
    hot = torch.zeros_like(q).scatter_(1, ((q - torch.log(noise)) / 0.5).argmax(1, keepdim=True), 1.0)
    return hot, index
