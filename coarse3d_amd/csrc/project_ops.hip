// Spherical (range-image) projection of a LiDAR scan and the point augmentation that precedes it,
// on the device.  Reference: pc_processor/dataset/preprocess/projection.py:43-115
// (RangeProjection.doProjection) and augmentor.py:150-230 (flip / translation / rotation).
//
// The reference orders the points by decreasing depth and scatters them in that order, so the
// CLOSEST point wins each pixel.  Here every point does one 64-bit atomicMin of (depth bits,
// point index) on its pixel -- integer min is order independent, so the result is deterministic;
// among points of exactly equal depth the smallest index wins (the reference's argsort is
// unstable there).  All float32 arithmetic is spelled with explicitly rounded operations in
// numpy's order; arctan2 / arcsin are evaluated in double and rounded once to float32.
#include "common.h"
#include "../../include/coarse3d_hip.h"

namespace {

constexpr unsigned long long EMPTY = ~0ull;

__device__ __forceinline__ float c3d_div_rn(float a, float b) { return (float)((double)a / (double)b); }

__global__ __launch_bounds__(256) void augment_kernel(float* __restrict__ pc, int n, int stride, float sx, float sy,
                                                      float tx, float ty, float tz, const double* __restrict__ rot) {
  // augmentor.py: flipX/flipY (:150-158), translation (:160-165, float32 adds), rotation (:167-174,
  // float32 points x float64 matrix, result cast back to float32)
  double r[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) r[k] = rot[k];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    float* p = pc + (size_t)i * stride;
    const float x = __fadd_rn(sx * p[0], tx), y = __fadd_rn(sy * p[1], ty), z = __fadd_rn(p[2], tz);
    const double dx = x, dy = y, dz = z;
    p[0] = (float)(dx * r[0] + dy * r[1] + dz * r[2]);
    p[1] = (float)(dx * r[3] + dy * r[4] + dz * r[5]);
    p[2] = (float)(dx * r[6] + dy * r[7] + dz * r[8]);
  }
}

__global__ __launch_bounds__(256) void project_points_kernel(const float* __restrict__ pc, int n, int stride,
                                                             const float* __restrict__ depth_in, float fov_left_abs,
                                                             float fov_hori, float fov_down_abs, float fov_vert, int W, int H,
                                                             int32_t* __restrict__ ux, int32_t* __restrict__ uy,
                                                             float* __restrict__ udepth, unsigned long long* __restrict__ zbuf) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float* p = pc + (size_t)i * stride;
    const float x = p[0], y = p[1], z = p[2];
    // np.linalg.norm(pointcloud[:, :3], 2, axis=1): sqrt((x*x + y*y) + z*z) in float32
    // (sqrt and the divisions below go through double: the double result rounded once to
    //  float32 is the correctly rounded float32 result, whatever the fp32 fast-path settings)
    const float ss = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
    const float depth = depth_in ? depth_in[i] : (float)sqrt((double)ss);
    const float yaw = -(float)atan2((double)y, (double)x);
    const float pitch = (float)asin((double)c3d_div_rn(z, depth));
    float fx = __fmul_rn(c3d_div_rn(__fadd_rn(yaw, fov_left_abs), fov_hori), (float)W);
    float fy = __fmul_rn(__fsub_rn(1.0f, c3d_div_rn(__fadd_rn(pitch, fov_down_abs), fov_vert)), (float)H);
    // A point at the origin (or with non-finite coordinates) has no direction: z/depth is NaN.
    // numpy's minimum/maximum propagate the NaN and the int32 cast of it is undefined in the
    // reference (projection.py:74-85); here such a point is reported at pixel (0, 0) and never
    // enters the z-buffer, so it cannot shadow a real return.
    const bool ok = isfinite(fx) && isfinite(fy) && depth >= 0.f;
    fx = ok ? fmaxf(fminf((float)(W - 1), floorf(fx)), 0.f) : 0.f;
    fy = ok ? fmaxf(fminf((float)(H - 1), floorf(fy)), 0.f) : 0.f;
    const int px = (int)fx, py = (int)fy;
    ux[i] = px;
    uy[i] = py;
    udepth[i] = depth;
    const unsigned long long key = ((unsigned long long)__float_as_uint(depth) << 32) | (unsigned)i;
    if (ok) atomicMin(&zbuf[(size_t)py * W + px], key);
  }
}

// per pixel: winner index, range, point columns; optionally the loader tensors
__global__ __launch_bounds__(256) void project_gather_kernel(const unsigned long long* __restrict__ zbuf, const float* __restrict__ pc,
                                                             int stride, int cols, const float* __restrict__ udepth, int HW,
                                                             float* __restrict__ proj_pc, float* __restrict__ proj_range,
                                                             int32_t* __restrict__ proj_idx, int32_t* __restrict__ proj_mask,
                                                             const int64_t* __restrict__ sem, const int64_t* __restrict__ weak,
                                                             float* __restrict__ feat5, float* __restrict__ eval_label,
                                                             float* __restrict__ train_label) {
  for (int q = blockIdx.x * 256 + threadIdx.x; q < HW; q += gridDim.x * 256) {
    const unsigned long long key = zbuf[q];
    const bool hit = key != EMPTY;
    const int idx = hit ? (int)(unsigned)(key & 0xffffffffull) : -1;
    const float* p = hit ? pc + (size_t)idx * stride : nullptr;
    const float rng = hit ? udepth[idx] : -1.f;
    if (proj_idx) proj_idx[q] = idx;
    if (proj_mask) proj_mask[q] = idx > 0;                 // projection.py:113: (proj_idx > 0), sic
    if (proj_range) proj_range[q] = rng;
    if (proj_pc)
      for (int c = 0; c < cols; ++c) proj_pc[(size_t)q * cols + c] = hit ? p[c] : -1.f;
    if (feat5) {                                           // wss_sem_kitti_loader.py:150-164
      feat5[q] = rng;
      feat5[HW + q] = hit ? p[0] : -1.f;
      feat5[2 * HW + q] = hit ? p[1] : -1.f;
      feat5[3 * HW + q] = hit ? p[2] : -1.f;
      const float it = hit ? p[3] : -1.f;
      feat5[4 * HW + q] = it != -1.f ? it : 0.f;           // intensity.ne(-1) * intensity
    }
    if (eval_label) eval_label[q] = hit ? (float)sem[idx] : 0.f;
    if (train_label) train_label[q] = hit ? (float)weak[idx] : 0.f;
  }
}

int grid_for(int n) {
  int b = (n + 255) / 256;
  return b < 1 ? 1 : (b > 4096 ? 4096 : b);
}

}  // namespace

extern "C" int c3d_augment_points(float* pc, int n, int stride, float sx, float sy, float tx, float ty, float tz,
                                  const double* rot_dev, c3d_stream stream) {
  C3D_REQUIRE(stride >= 3, "augment: points need at least x, y, z");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(augment_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, pc, n, stride, sx, sy, tx, ty, tz,
                     rot_dev);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_range_project(const float* pc, int n, int stride, int cols, const float* depth, float fov_left_abs,
                                 float fov_hori, float fov_down_abs, float fov_vert, int W, int H, int32_t* ux, int32_t* uy,
                                 float* udepth, uint64_t* zbuf, float* proj_pc, float* proj_range, int32_t* proj_idx,
                                 int32_t* proj_mask, const int64_t* sem, const int64_t* weak, float* feat5,
                                 float* eval_label, float* train_label, c3d_stream stream) {
  C3D_REQUIRE(stride >= 3 && cols <= stride, "range_project: bad point layout");
  C3D_REQUIRE(!feat5 || cols >= 4, "range_project: the 5-channel feature needs x, y, z, intensity");
  C3D_REQUIRE((!eval_label || sem) && (!train_label || weak), "range_project: labels missing");
  hipStream_t st = (hipStream_t)stream;
  C3D_REQUIRE(hipMemsetAsync(zbuf, 0xff, sizeof(uint64_t) * (size_t)H * W, st) == hipSuccess,
              "range_project: clearing the z-buffer failed");
  if (n > 0) {
    hipLaunchKernelGGL(project_points_kernel, dim3(grid_for(n)), dim3(256), 0, st, pc, n, stride, depth, fov_left_abs, fov_hori,
                       fov_down_abs, fov_vert, W, H, ux, uy, udepth, reinterpret_cast<unsigned long long*>(zbuf));
    C3D_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(project_gather_kernel, dim3(grid_for(H * W)), dim3(256), 0, st,
                     reinterpret_cast<const unsigned long long*>(zbuf), pc, stride, cols, udepth, H * W, proj_pc, proj_range,
                     proj_idx, proj_mask, sem, weak, feat5, eval_label, train_label);
  C3D_CHECK_LAUNCH();
  return 0;
}
