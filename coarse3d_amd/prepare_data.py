"""Weak-label generation by random voxel sampling on the GPU.

Mirrors the per-scan body of the reference's offline tool, tasks/prepare_data/
gen_sem_weak_label_rand_grid.py:140-272 (``SemanticData.__getitem__``): a voxel grid of edge
``voxel_size`` (0.06 m) over the scan, ``label_ratio`` x points voxels sampled uniformly among
the voxels whose first point carries a class > 0, and the voxel's class written to every point
inside it (``voxel_propagation``) or to its first point.  The reference spends seconds per scan
in a Python loop over points (``voxel_grid.get_voxel(pt) for pt in scan``) and one full-array
comparison per sampled voxel; here it is a handful of launches (csrc/voxel_ops.hip).

Randomness: the reference draws ``np.random.choice(valid, k, replace=False)`` from NumPy's
global Mersenne twister.  The device takes one float32 priority per point and keeps the k
valid voxels whose first point has the smallest priority -- the same uniform distribution over
k-subsets.  ``sample_idx`` (indices into the sorted unique voxels, what the reference's draw
returns) can be injected instead, which reproduces a recorded reference run exactly."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


def sample_count(n_points, label_ratio):
    """gen_sem_weak_label_rand_grid.py:210-216: number of voxels to label, at least one."""
    return max(int(np.around(n_points * label_ratio)), 1)


def voxel_weak_labels(scan, mapped_label, voxel_size=0.06, label_ratio=0.001, voxel_propagation=True,
                      priority=None, generator=None, return_info=False):
    """scan: CUDA float32 [n, >=3] (x, y, z first), mapped_label: CUDA int32 [n] (0 = ignore).
    priority: float32 [n] or None (drawn with ``generator``).  Returns point_weak_label int32 [n]
    (and, with ``return_info``, the counters + per-point voxel coordinates).

    Raises ValueError like np.random.choice when fewer than the requested number of voxels carry
    a label (:222) and when the scan holds non-finite coordinates."""
    if not (scan.is_cuda and scan.dtype == torch.float32 and scan.dim() == 2 and scan.shape[1] >= 3):
        raise ValueError("scan must be a CUDA float32 tensor [n, >=3]")
    scan = scan.contiguous()
    n = scan.shape[0]
    dev = scan.device
    label = mapped_label.to(dev, torch.int32).contiguous()
    if label.numel() != n:
        raise ValueError("scan and label differ in length")          # assert len(scan) == len(label), :153
    k = sample_count(n, label_ratio)
    if priority is None:
        priority = torch.rand(n, device=dev, generator=generator)
    priority = priority.to(dev, torch.float32).contiguous()
    lib = L.lib()
    nbytes = lib.c3d_voxel_sampler_workspace_bytes(n)
    work = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    weak = torch.empty(n, device=dev, dtype=torch.int32)
    stats = torch.empty(5, device=dev, dtype=torch.int32)
    p2v = torch.empty(n, 3, device=dev, dtype=torch.int32) if return_info else None
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.c3d_voxel_weak_labels(scan.data_ptr(), n, scan.shape[1], label.data_ptr(), float(voxel_size),
                                      priority.data_ptr(), k, int(bool(voxel_propagation)), work.data_ptr(), nbytes,
                                      p2v.data_ptr() if p2v is not None else None, weak.data_ptr(), stats.data_ptr(),
                                      stream), "c3d_voxel_weak_labels")
    bad, num_voxel, n_valid, n_sampled, n_labelled = (int(v) for v in stats.cpu())     # offline tool: one sync per scan
    if bad:
        raise ValueError(f"{bad} points have non-finite coordinates or lie outside the 2^21-voxel grid")
    if n_sampled < k:
        raise ValueError("Cannot take a larger sample than population when 'replace=False' "
                         f"({k} voxels requested, {n_valid} carry a label)")
    if return_info:
        return weak, dict(sample_voxel=k, num_voxel=num_voxel, n_valid=n_valid, num_labelled_pts=n_labelled,
                          point2voxel=p2v)
    return weak


def priorities_for(sample_idx, first_point, n):
    """Priorities that make the device sampler pick exactly the voxels ``sample_idx`` (indices into
    the sorted unique voxels whose first points are ``first_point``): replay of a recorded
    np.random.choice draw."""
    pr = np.full(n, 2.0, dtype=np.float32)
    pr[np.asarray(first_point)[np.asarray(sample_idx)]] = np.linspace(0.0, 1.0, len(sample_idx), dtype=np.float32)
    return pr
