"""CPU oracle of the weak-label voxel sampler  --  TEST INFRASTRUCTURE ONLY.

NumPy restatement of the per-scan body of the reference's offline label generator,
tasks/prepare_data/gen_sem_weak_label_rand_grid.py:190-246 (SemanticData.__getitem__).  Imported
only by tests/; never by the product package.

Parity status: PINNED for everything the reference script computes in NumPy -- tests/golden/
make_golden_weak_label.py runs the reference's own ``__getitem__`` on scan files and stores its
outputs in tests/golden/weak_label.npz, which tests/test_oracle_golden.py replays here.  The
voxel grid itself comes from open3d (``open3d==0.15.2``, requirements.txt:1), a third-party
dependency that is absent from this image and not vendored by the reference: **parity UNPINNED
against open3d binaries**.  ``voxel_coords`` restates open3d's published algorithm --
``VoxelGrid::CreateFromPointCloud`` / ``CreateFromPointCloudWithinBounds``
(cpp/open3d/geometry/VoxelGridFactory.cpp at tag v0.15.2: min_bound = GetMinBound() -
voxel_size * 0.5, origin_ = min_bound, index = int(floor((point - min_bound) / voxel_size)))
and ``VoxelGrid::GetVoxel`` (cpp/open3d/geometry/VoxelGrid.cpp: floor((point - origin_) /
voxel_size_).cast<int>()), double arithmetic on the float32 points (``Vector3dVector``), a TRUE
division per component (Eigen >= 3.3) -- and is checked against a second, scalar-by-scalar C
restatement of the same published functions (oracle/open3d_voxel_rule.c) and against
hand-derived expectations on edge-case vectors (tests/golden/weak_label_edges.npz: points
on voxel faces, negative coordinates, a one-point cloud, float32 points a few ulps from a face
where a reciprocal-multiply implementation lands in the neighbouring voxel).  The golden
generator feeds the SAME rule to the reference script through a stand-in ``open3d`` module.
"""
import numpy as np


def voxel_coords(xyz, voxel_size):
    """open3d VoxelGrid.create_from_point_cloud + get_voxel per point (:190-202)."""
    pts = np.asarray(xyz, dtype=np.float64)
    origin = pts.min(0) - voxel_size * 0.5
    return np.floor((pts - origin) / voxel_size).astype(np.int32)


def sample_count(n_points, label_ratio):
    """:210-216: voxels to label, at least one."""
    return max(int(np.around(n_points * label_ratio)), 1)


def voxel_weak_labels(xyz, mapped_label, voxel_size, n_sample, voxel_propagation=True, rng=None, sample_idx=None):
    """Returns (point_weak_label, info).  ``rng``: a np.random.RandomState (the reference uses the
    global one, :222); ``sample_idx`` overrides the draw (indices into the sorted unique voxels)."""
    point2voxel = voxel_coords(xyz, voxel_size)
    voxels_coord, point2voxel_map, num_pts_in_voxel = np.unique(point2voxel, return_index=True, return_counts=True,
                                                                axis=0)                 # :203-205
    voxel_label = mapped_label[point2voxel_map]                                         # :207
    valid_idxes = np.where(voxel_label > 0)[0]                                          # :219
    if sample_idx is None:
        sample_idx = (rng or np.random).choice(valid_idxes, n_sample, replace=False)    # :222
    voxel_weak_label = np.zeros_like(voxel_label, dtype=mapped_label.dtype)
    point_weak_label = np.zeros_like(mapped_label, dtype=mapped_label.dtype)
    voxel_weak_label[sample_idx] = voxel_label[sample_idx]                              # :225
    # :231-241 -- vectorised: the voxel rank of every point, then a table lookup
    _, inverse = np.unique(point2voxel, return_inverse=True, axis=0)
    inverse = inverse.reshape(-1)
    chosen = np.zeros(len(voxels_coord), dtype=bool)
    chosen[sample_idx] = True
    if voxel_propagation:
        sel = chosen[inverse]
        point_weak_label[sel] = voxel_label[inverse[sel]]
    else:
        first = point2voxel_map[sample_idx]
        point_weak_label[first] = voxel_label[sample_idx]
    return point_weak_label, dict(point2voxel=point2voxel, voxels_coord=voxels_coord, first_point=point2voxel_map,
                                  voxel_label=voxel_label, sample_idx=np.asarray(sample_idx),
                                  num_voxel=len(voxels_coord), n_valid=len(valid_idxes))
