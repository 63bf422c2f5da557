"""Weak-label voxel sampler: the HIP path through the C ABI against the outputs
of the reference script's own ``SemanticData.__getitem__`` (tests/golden/weak_label.npz) and
against the NumPy oracle on a full-size scan.  Integer work: everything bit-exact."""
import os

import numpy as np
import pytest
import torch

from coarse3d_amd import prepare_data as PD
from oracle import weak_label_oracle as wo

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _run(scan, lab, vs, ratio, prop, sample_idx):
    info_o = wo.voxel_weak_labels(scan[:, :3], lab, vs, wo.sample_count(len(scan), ratio), prop, sample_idx=sample_idx)[1]
    pr = PD.priorities_for(sample_idx, info_o["first_point"], len(scan))
    weak, info = PD.voxel_weak_labels(torch.from_numpy(scan).cuda(), torch.from_numpy(lab).cuda(), vs, ratio, prop,
                                      priority=torch.from_numpy(pr), return_info=True)
    return weak.cpu().numpy(), info, info_o


def test_matches_reference_script_outputs():
    g = np.load(os.path.join(GOLD, "weak_label.npz"))
    for tag in "abcd":
        scan, lab = g[f"{tag}.scan"], g[f"{tag}.mapped_label"]
        weak, info, info_o = _run(scan, lab, float(g[f"{tag}.voxel_size"]), float(g[f"{tag}.label_ratio"]),
                                  bool(g[f"{tag}.propagation"]), g[f"{tag}.sample_idx"])
        assert (weak == g[f"{tag}.weak"]).all(), tag
        assert info["sample_voxel"] == int(g[f"{tag}.sample_voxel"])
        assert info["num_labelled_pts"] == int(g[f"{tag}.num_labelled"])
        assert info["num_voxel"] == info_o["num_voxel"] and info["n_valid"] == info_o["n_valid"]
        assert (info["point2voxel"].cpu().numpy() == info_o["point2voxel"]).all()


def test_voxel_rule_edge_cases():
    """The device's voxel indices on the edge-case vectors of the published open3d rule (tests/golden/
    make_golden_weak_label_edges.py; tests/test_oracle_golden.py pins both CPU restatements to the same file): points on
    voxel faces, negative coordinates, a one-point cloud, float32 points a few ulps from a face of the 0.06 m grid."""
    g = np.load(os.path.join(GOLD, "weak_label_edges.npz"))
    for tag in ("exact.a", "exact.b", "exact.c", "exact.d", "near.a"):
        xyz, want, vs = g[f"{tag}.xyz"], g[f"{tag}.want"], float(g[f"{tag}.voxel_size"])
        scan = np.concatenate([xyz, np.zeros((len(xyz), 1), np.float32)], 1)
        lab = np.ones(len(xyz), dtype=np.int32)
        _, info = PD.voxel_weak_labels(torch.from_numpy(scan).cuda(), torch.from_numpy(lab).cuda(), vs, 0.001, True,
                                       generator=torch.Generator(device="cuda").manual_seed(1), return_info=True)
        assert (info["point2voxel"].cpu().numpy() == want).all(), tag
        assert info["num_voxel"] == len(np.unique(want, axis=0))


def test_full_size_scan_vs_oracle_and_properties():
    """120k points, 0.06 m voxels, 0.1 % (the SemanticKITTI setting of the reference, :321-335)."""
    rs = np.random.RandomState(3)
    n = 120_000
    ang, rad = rs.uniform(-np.pi, np.pi, n), np.abs(rs.normal(15, 10, n)) + 1
    scan = np.stack([rad * np.cos(ang), rad * np.sin(ang), rs.normal(-1, 0.5, n), rs.uniform(0, 1, n)], 1).astype(np.float32)
    scan[: n // 3, :3] = np.round(scan[: n // 3, :3] * 8) / 8            # many multi-point voxels
    lab = (((ang + np.pi) / (2 * np.pi) * 40).astype(int) % 20).astype(np.int32)
    k = wo.sample_count(n, 0.001)
    weak_o, info_o = wo.voxel_weak_labels(scan[:, :3], lab, 0.06, k, True, rng=np.random.RandomState(5))
    weak, info, _ = _run(scan, lab, 0.06, 0.001, True, info_o["sample_idx"])
    assert (weak == weak_o).all()
    assert info["num_voxel"] == info_o["num_voxel"]
    # device-drawn priorities: properties of a valid sample
    w2, i2 = PD.voxel_weak_labels(torch.from_numpy(scan).cuda(), torch.from_numpy(lab).cuda(), 0.06, 0.001, True,
                                  generator=torch.Generator(device="cuda").manual_seed(1), return_info=True)
    w2 = w2.cpu().numpy()
    p2v = i2["point2voxel"].cpu().numpy()
    labelled = w2 > 0
    assert ((w2 == lab) | ~labelled).all()                               # a weak label is always the point's own class...
    keys = (p2v[:, 0].astype(np.int64) << 42) | (p2v[:, 1].astype(np.int64) << 21) | p2v[:, 2]
    chosen = np.unique(keys[labelled])
    assert len(chosen) == k                                              # ...exactly k voxels were sampled...
    first = {}
    for i, kk in enumerate(keys):
        first.setdefault(kk, i)
    for kk in chosen:                                                    # ...whole voxels, labelled like their first point
        members = keys == kk
        assert (w2[members] == lab[first[kk]]).all()
    # first-point-only mode labels exactly one point per sampled voxel
    w3, i3 = PD.voxel_weak_labels(torch.from_numpy(scan).cuda(), torch.from_numpy(lab).cuda(), 0.06, 0.001, False,
                                  generator=torch.Generator(device="cuda").manual_seed(1), return_info=True)
    assert int((w3 > 0).sum()) == k == i3["num_labelled_pts"]


def test_error_behaviour_follows_the_reference():
    scan = torch.randn(500, 4).cuda()
    lab = torch.zeros(500, dtype=torch.int32).cuda()
    lab[:3] = 4
    with pytest.raises(ValueError, match="larger sample than population"):      # np.random.choice, :222
        PD.voxel_weak_labels(scan, lab, 0.06, 0.5)
    bad = scan.clone()
    bad[7, 1] = float("nan")
    with pytest.raises(ValueError, match="non-finite"):
        PD.voxel_weak_labels(bad, lab + 1, 0.06, 0.01)
    with pytest.raises(ValueError, match="differ in length"):
        PD.voxel_weak_labels(scan, lab[:10], 0.06, 0.01)
