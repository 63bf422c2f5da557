/* TEST INFRASTRUCTURE ONLY -- never linked or loaded by the product (coarse3d_amd/).
 *
 * Second, independent restatement of the voxelisation the reference's weak-label generator gets from open3d
 * (tasks/prepare_data/gen_sem_weak_label_rand_grid.py:178-193: `o3d.geometry.VoxelGrid.create_from_point_cloud(pcd,
 * voxel_size)` then `voxel_grid.get_voxel(pt)` per point; requirements.txt:1 pins open3d==0.15.2).  open3d is a
 * third-party dependency that is absent from /root/reference and from this image (no binaries, no sources, no network), so
 * this file restates its PUBLISHED algorithm, scalar by scalar in the order the C++ evaluates it:
 *
 *   open3d v0.15.2, cpp/open3d/geometry/VoxelGridFactory.cpp
 *     VoxelGrid::CreateFromPointCloud(input, voxel_size):
 *         voxel_size3 = (voxel_size, voxel_size, voxel_size)
 *         min_bound   = input.GetMinBound() - voxel_size3 * 0.5        // component-wise minimum of the points
 *         max_bound   = input.GetMaxBound() + voxel_size3 * 0.5
 *         return CreateFromPointCloudWithinBounds(input, voxel_size, min_bound, max_bound)
 *     VoxelGrid::CreateFromPointCloudWithinBounds(...):
 *         output->origin_ = min_bound
 *         for each point:  ref_coord = (point - min_bound) / voxel_size
 *                          voxel_index = (int(floor(ref_coord(0))), int(floor(ref_coord(1))), int(floor(ref_coord(2))))
 *   open3d v0.15.2, cpp/open3d/geometry/VoxelGrid.cpp
 *     VoxelGrid::GetVoxel(point):  voxel_f = (point - origin_) / voxel_size_;  return floor(voxel_f.array()).cast<int>()
 *
 * (function names and files as published; the line numbers inside those files cannot be checked offline and are not
 * quoted.)  Points reach open3d through `o3d.utility.Vector3dVector(xyz)`: float32 scan coordinates widened to double.
 * `Eigen::Vector3d / double` is a true IEEE division per component in Eigen >= 3.3 (scalar_quotient_op; open3d 0.15 builds
 * against Eigen 3.4) -- NOT a multiplication by the reciprocal, which differs in the last bit for points next to a voxel
 * face; tests/golden/weak_label_edges.npz holds such points.
 *
 * Parity status: the restatement is checked against hand-derived expectations on exactly representable cases
 * (tests/test_oracle_golden.py) -- "parity UNPINNED against open3d binaries" until a real open3d run exists. */
#include <math.h>
#include <stdint.h>

/* xyz: n points, `stride` floats apart, x y z first.  out: n * 3 int32 voxel indices; origin_out: 3 doubles. */
int open3d_voxel_indices(const float* xyz, int64_t n, int stride, double voxel_size, int32_t* out, double* origin_out) {
  if (n <= 0 || voxel_size <= 0.0) return 1;
  double mn[3];
  for (int a = 0; a < 3; ++a) mn[a] = (double)xyz[a];
  for (int64_t i = 1; i < n; ++i)
    for (int a = 0; a < 3; ++a) {
      const double v = (double)xyz[i * stride + a];
      if (v < mn[a]) mn[a] = v; /* GetMinBound: Eigen cwiseMin fold over the points */
    }
  double origin[3];
  for (int a = 0; a < 3; ++a) {
    origin[a] = mn[a] - voxel_size * 0.5; /* min_bound = GetMinBound() - voxel_size3 * 0.5 */
    origin_out[a] = origin[a];
  }
  for (int64_t i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) {
      const double ref = ((double)xyz[i * stride + a] - origin[a]) / voxel_size; /* (point - origin_) / voxel_size_ */
      out[i * 3 + a] = (int32_t)floor(ref);
    }
  return 0;
}
