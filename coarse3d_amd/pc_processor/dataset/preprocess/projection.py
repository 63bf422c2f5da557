"""Spherical range projection on the device (reference
pc_processor/dataset/preprocess/projection.py:4-115, same class name / constructor / method).

``doProjection`` takes the scan as a CUDA float32 tensor [n, >=3] and returns CUDA tensors; the
un-projection indices are cached in ``cached_data`` exactly as the reference caches them.
``project_scan`` additionally builds the tensors the weak-label loaders hand to the trainer
(pc_processor/dataset/semantic_kitti/wss_sem_kitti_loader.py:113-170) in the same launch."""
import numpy as np

from .... import ops


class RangeProjection(object):
    """project 3d point cloud to 2d data with range projection"""

    def __init__(self, fov_up=3, fov_down=-25, proj_w=512, proj_h=64, fov_left=-180, fov_right=180):
        assert fov_up >= 0 and fov_down <= 0, \
            "require fov_up >= 0 and fov_down <= 0, while fov_up/fov_down is {}/{}".format(fov_up, fov_down)
        assert fov_right >= 0 and fov_left <= 0, \
            "require fov_right >= 0 and fov_left <= 0, while fov_right/fov_left is {}/{}".format(fov_right, fov_left)
        self.fov_up = fov_up / 180.0 * np.pi
        self.fov_down = fov_down / 180.0 * np.pi
        self.fov_vert = abs(self.fov_up) + abs(self.fov_down)
        self.fov_left = fov_left / 180.0 * np.pi
        self.fov_right = fov_right / 180.0 * np.pi
        self.fov_hori = abs(self.fov_left) + abs(self.fov_right)
        self.proj_w = proj_w
        self.proj_h = proj_h
        self.cached_data = {}

    def _fov(self):
        # the reference mixes float32 arrays with these Python floats: numpy rounds each scalar to
        # float32 before the operation, so the kernel receives the float32 values
        return (float(np.float32(abs(self.fov_left))), float(np.float32(self.fov_hori)),
                float(np.float32(abs(self.fov_down))), float(np.float32(self.fov_vert)))

    def doProjection(self, pointcloud, depth=None):
        """pointcloud: CUDA float32 [n, c>=3].  Returns (proj_pointcloud [H,W,c], proj_range [H,W],
        proj_idx [H,W] int32, proj_mask [H,W] int32), all on the device."""
        out = ops.range_project(pointcloud, depth, self._fov(), self.proj_w, self.proj_h, want_image=True)
        self.cached_data = {"uproj_x_idx": out["ux"], "uproj_y_idx": out["uy"], "uproj_depth": out["udepth"]}
        return out["proj_pc"], out["proj_range"], out["proj_idx"], out["proj_mask"]

    def project_scan(self, pointcloud, sem_label, weak_label, depth=None):
        """One launch for everything wss_sem_kitti_loader.py:113-170 derives from the projection:
        returns dict(feature [5,H,W], train_label [H,W], eval_label [H,W], eval_mask [H,W],
        proj_idx [H,W], uproj_x_idx, uproj_y_idx, uproj_depth [n])."""
        out = ops.range_project(pointcloud, depth, self._fov(), self.proj_w, self.proj_h, want_image=False,
                                sem=sem_label, weak=weak_label)
        self.cached_data = {"uproj_x_idx": out["ux"], "uproj_y_idx": out["uy"], "uproj_depth": out["udepth"]}
        return {"feature": out["feat5"], "train_label": out["train_label"], "eval_label": out["eval_label"],
                "eval_mask": out["proj_mask"], "proj_idx": out["proj_idx"], "uproj_x_idx": out["ux"],
                "uproj_y_idx": out["uy"], "uproj_depth": out["udepth"]}
