#!/usr/bin/env python3
"""Golden vectors of the weak-label voxel sampler from the REAL reference script.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden_weak_label.py
Imports tasks/prepare_data/gen_sem_weak_label_rand_grid.py and calls its own
``SemanticData.__getitem__`` on scan / label files written to a temporary directory.  The one
dependency the image lacks is open3d (0.15.2): a stand-in module provides the three VoxelGrid
calls the script makes, restating open3d's published voxelisation rule (see
oracle/weak_label_oracle.py).  Everything after it -- np.unique, the label of the first point,
np.random.choice, the propagation loop -- is the reference's code executing unmodified."""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/tasks/prepare_data/gen_sem_weak_label_rand_grid.py"


def _open3d_standin():
    o3d = types.ModuleType("open3d")

    class PointCloud:
        def __init__(self):
            self.points = None

    class VoxelGrid:
        def __init__(self, origin, voxel_size, voxels):
            self.origin, self.voxel_size, self._voxels = origin, voxel_size, voxels

        @staticmethod
        def create_from_point_cloud(pcd, voxel_size):
            pts = np.asarray(pcd.points, dtype=np.float64)
            origin = pts.min(0) - voxel_size * 0.5
            idx = np.floor((pts - origin) / voxel_size).astype(np.int32)
            return VoxelGrid(origin, voxel_size, np.unique(idx, axis=0))

        def get_voxels(self):
            return list(self._voxels)

        def get_voxel(self, pt):
            return np.floor((np.asarray(pt, dtype=np.float64) - self.origin) / self.voxel_size).astype(np.int32)

    o3d.geometry = types.SimpleNamespace(PointCloud=PointCloud, VoxelGrid=VoxelGrid)
    o3d.utility = types.SimpleNamespace(Vector3dVector=lambda a: np.asarray(a, dtype=np.float64))
    return o3d


def synthetic_scan(seed, n, spread):
    """A lidar-like scan: points on a few noisy surfaces so that voxels hold 1..many points;
    raw SemanticKITTI label ids whose learning_map gives classes 0..19 (0 = ignore)."""
    g = np.random.RandomState(seed)
    ang = g.uniform(-np.pi, np.pi, n)
    rad = np.abs(g.normal(spread, spread / 3, n)) + 0.5
    z = g.normal(-1.0, 0.4, n)
    xyz = np.stack([rad * np.cos(ang), rad * np.sin(ang), z], 1)
    xyz[: n // 4] = np.round(xyz[: n // 4] * 4) / 4          # a quarter of the points share exact coordinates
    scan = np.concatenate([xyz, g.uniform(0, 1, (n, 1))], 1).astype(np.float32)
    return scan


def main():
    sys.modules["open3d"] = _open3d_standin()
    spec = importlib.util.spec_from_file_location("ref_weak", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    cfg_path = "/root/reference/pc_processor/dataset/semantic_kitti/semantic-kitti.yaml"
    cfg = yaml.safe_load(open(cfg_path))
    raw_ids = np.array(sorted(cfg["learning_map"].keys()), dtype=np.int32)
    out = {}
    cases = [("a", 11, 4000, 6.0, 0.06, 0.01, True), ("b", 12, 3000, 2.0, 0.5, 0.02, True),
             ("c", 13, 2500, 2.0, 0.5, 0.02, False), ("d", 14, 1500, 1.0, 1.0, 0.0001, True)]
    with tempfile.TemporaryDirectory() as tmp:
        root, save = os.path.join(tmp, "in"), os.path.join(tmp, "out")
        os.makedirs(os.path.join(root, "00", "velodyne"))
        os.makedirs(os.path.join(root, "00", "labels"))
        for k, (tag, seed, n, spread, vs, ratio, prop) in enumerate(cases):
            scan = synthetic_scan(seed, n, spread)
            g = np.random.RandomState(seed + 100)
            # labels constant over coarse angular sectors (so whole voxels share a class), upper half = instance ids
            sector = ((np.arctan2(scan[:, 1], scan[:, 0]) + np.pi) / (2 * np.pi) * 24).astype(int) % 24
            raw = raw_ids[g.randint(0, len(raw_ids), 24)][sector].astype(np.int32) | (g.randint(0, 50, n).astype(np.int32) << 16)
            scan.tofile(os.path.join(root, "00", "velodyne", f"{k:06d}.bin"))
            raw.tofile(os.path.join(root, "00", "labels", f"{k:06d}.label"))
        args = types.SimpleNamespace(data_config_path=cfg_path, sequences=(0,), dataset_root=root, dataset_save=save,
                                     weak_label_name="0.1", debug=False, voxel_size=0.06, label_ratio=0.001,
                                     voxel_propagation=True)
        ref.args = args                     # the script reads `args.voxel_propagation` as a module global (:233)
        ds = ref.SemanticData(args=args)
        for k, (tag, seed, n, spread, vs, ratio, prop) in enumerate(cases):
            args.voxel_size, args.label_ratio, args.voxel_propagation = vs, ratio, prop
            scan = ds.load_scan(ds.scan_files["00"][k])
            mapped = ds.label_map[ds.load_label(ds.label_files["00"][k])]
            np.random.seed(1000 + k)
            state = np.random.get_state()
            weak, _, _, sample_voxel, num_labelled, *_ = ds[k]
            # replay the draw to record WHICH voxels the reference sampled (same stream position)
            import oracle.weak_label_oracle as wo
            p2v = wo.voxel_coords(scan[:, :3], vs)
            _, first = np.unique(p2v, return_index=True, axis=0)
            valid = np.where(mapped[first] > 0)[0]
            np.random.set_state(state)
            sample_idx = np.random.choice(valid, sample_voxel, replace=False)
            out.update({f"{tag}.scan": scan, f"{tag}.mapped_label": mapped.astype(np.int32),
                        f"{tag}.voxel_size": np.float64(vs), f"{tag}.label_ratio": np.float64(ratio),
                        f"{tag}.propagation": np.int32(prop), f"{tag}.sample_voxel": np.int32(sample_voxel),
                        f"{tag}.sample_idx": sample_idx.astype(np.int64), f"{tag}.seed": np.int64(1000 + k),
                        f"{tag}.weak": np.asarray(weak).astype(np.int32),
                        f"{tag}.num_labelled": np.int64(num_labelled)})
            print(tag, "points", len(scan), "sampled voxels", sample_voxel, "labelled points", int(num_labelled))
    np.savez_compressed(os.path.join(HERE, "weak_label.npz"), **out)


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    main()
