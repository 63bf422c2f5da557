// Weak-label generation by random voxel sampling (SURVEY 8f, N4 second half).
//
// Replaces the per-scan body of tasks/prepare_data/gen_sem_weak_label_rand_grid.py:140-272
// (SemanticData.__getitem__): open3d VoxelGrid.create_from_point_cloud(voxel_size) +
// get_voxel per point (open3d==0.15.2, requirements.txt:1 -- absent from the reference tree;
// its published rule is: origin = min_bound - voxel_size/2, voxel = floor((p - origin) /
// voxel_size) in double), np.unique(rows) with first-occurrence indices, voxel label = label of
// the first point, a uniform sample of `n_sample` voxels among those with label > 0, and
// propagation of the voxel's label to all of its points (or to its first point only).
//
// Device plan (one scan, n ~ 1e5 points; integer work, everything bit-exact):
//   1. one workgroup folds the bounding-box minimum -> origin (double)
//   2. per point: voxel coordinates (double arithmetic), 63-bit key i<<42 | j<<21 | k
//      (row-lexicographic order == np.unique(axis=0) order)
//   3. stable radix sort of (key, point index)   [rocPRIM, the vendor's sort primitive]
//      -> equal keys keep ascending point order, so a segment's head is np.unique's return_index
//   4. head flags + inclusive scan -> voxel rank of every sorted position
//   5. per voxel: first point, label, sampling priority (the caller's random key of the first
//      point; voxels with label <= 0 get +inf)
//   6. stable sort of the priorities; the n_sample smallest are the sample
//      (np.random.choice(valid, n_sample, replace=False) draws a uniform subset: any i.i.d.
//      priorities give the same distribution; the parity test feeds priorities that reproduce
//      the reference's draw)
//   7. per sorted position: write the voxel's label to its points
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "common.h"
#include "../../include/coarse3d_hip.h"

namespace {

constexpr uint32_t KEY_INF = 0xFFFFFFFFu;

struct Workspace {
  double* origin;        // [4]
  int32_t* counters;     // [8]: 0 bad coordinate, 1 num_voxel, 2 n_valid, 3 n_sampled, 4 n_labelled
  uint64_t *k_in, *k_out;
  int32_t *i_in, *i_out;
  int32_t* rank;         // inclusive scan of head flags (1-based voxel rank per sorted position)
  int32_t* head;
  int32_t* vlabel;
  uint32_t *p_in, *p_out;
  int32_t *v_in, *v_out;
  uint8_t* selected;
  void* tmp;
  size_t tmp_bytes;
};

size_t align256(size_t x) { return (x + 255) / 256 * 256; }

size_t sort_tmp_bytes(int n) {
  size_t a = 0, b = 0, c = 0;
  (void)rocprim::radix_sort_pairs(nullptr, a, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const int32_t*)nullptr,
                                  (int32_t*)nullptr, (size_t)n, 0, 63, (hipStream_t)0);
  (void)rocprim::radix_sort_pairs(nullptr, b, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int32_t*)nullptr,
                                  (int32_t*)nullptr, (size_t)n, 0, 32, (hipStream_t)0);
  (void)rocprim::inclusive_scan(nullptr, c, (const int32_t*)nullptr, (int32_t*)nullptr, (size_t)n, rocprim::plus<int32_t>(),
                                (hipStream_t)0);
  size_t m = a > b ? a : b;
  return m > c ? m : c;
}

size_t carve(Workspace& w, char* base, int n) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align256(bytes);
    return p;
  };
  w.origin = (double*)take(4 * sizeof(double));
  w.counters = (int32_t*)take(8 * sizeof(int32_t));
  w.k_in = (uint64_t*)take((size_t)n * 8);
  w.k_out = (uint64_t*)take((size_t)n * 8);
  w.i_in = (int32_t*)take((size_t)n * 4);
  w.i_out = (int32_t*)take((size_t)n * 4);
  w.rank = (int32_t*)take((size_t)n * 4);
  w.head = (int32_t*)take((size_t)n * 4);
  w.vlabel = (int32_t*)take((size_t)n * 4);
  w.p_in = (uint32_t*)take((size_t)n * 4);
  w.p_out = (uint32_t*)take((size_t)n * 4);
  w.v_in = (int32_t*)take((size_t)n * 4);
  w.v_out = (int32_t*)take((size_t)n * 4);
  w.selected = (uint8_t*)take((size_t)n);
  w.tmp_bytes = sort_tmp_bytes(n);
  w.tmp = take(w.tmp_bytes);
  return off;
}

// ---- 1. bounding-box minimum (one workgroup; 3n floats is a few hundred KB)
__global__ __launch_bounds__(1024) void origin_kernel(const float* __restrict__ xyz, int n, int stride, double voxel_size,
                                                      double* __restrict__ origin) {
  __shared__ float red[3][16];
  float m0 = INFINITY, m1 = INFINITY, m2 = INFINITY;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float* p = xyz + (size_t)i * stride;
    m0 = fminf(m0, p[0]);
    m1 = fminf(m1, p[1]);
    m2 = fminf(m2, p[2]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    m0 = fminf(m0, __shfl_xor(m0, o, 64));
    m1 = fminf(m1, __shfl_xor(m1, o, 64));
    m2 = fminf(m2, __shfl_xor(m2, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = m0;
    red[1][threadIdx.x >> 6] = m1;
    red[2][threadIdx.x >> 6] = m2;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    float m = INFINITY;
    for (int w = 0; w < 16; ++w) m = fminf(m, red[threadIdx.x][w]);
    origin[threadIdx.x] = (double)m - voxel_size * 0.5;      // min_bound - voxel_size * 0.5
  }
}

// ---- 2. voxel coordinates and sort keys
__global__ __launch_bounds__(256) void voxel_key_kernel(const float* __restrict__ xyz, int n, int stride, double voxel_size,
                                                        const double* __restrict__ origin, uint64_t* __restrict__ key,
                                                        int32_t* __restrict__ idx, int32_t* __restrict__ point2voxel,
                                                        int32_t* __restrict__ counters) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* p = xyz + (size_t)i * stride;
  int c[3];
  bool ok = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double f = floor(((double)p[a] - origin[a]) / voxel_size);    // Eigen: (point - origin) / voxel_size, floor
    ok = ok && f >= 0.0 && f < 2097152.0;                               // 21 bits per axis (NaN fails both)
    c[a] = ok ? (int)f : 0;
    if (point2voxel) point2voxel[(size_t)i * 3 + a] = c[a];
  }
  if (!ok) atomicAdd(&counters[0], 1);
  key[i] = ((uint64_t)c[0] << 42) | ((uint64_t)c[1] << 21) | (uint64_t)c[2];
  idx[i] = i;
}

// ---- 4. segment heads
__global__ __launch_bounds__(256) void head_kernel(const uint64_t* __restrict__ key, int n, int32_t* __restrict__ head) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s < n) head[s] = (s == 0 || key[s] != key[s - 1]) ? 1 : 0;
}

// ---- 5. per voxel: label of its first point, sampling priority
__global__ __launch_bounds__(256) void voxel_kernel(const int32_t* __restrict__ head, const int32_t* __restrict__ rank,
                                                    const int32_t* __restrict__ sorted_idx, int n,
                                                    const int32_t* __restrict__ label, const float* __restrict__ prio,
                                                    int32_t* __restrict__ vlabel, uint32_t* __restrict__ p_in,
                                                    int32_t* __restrict__ v_in, int32_t* __restrict__ counters) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= n) return;
  if (s == n - 1) counters[1] = rank[s];                 // number of voxels
  if (!head[s]) return;
  const int v = rank[s] - 1;
  const int first = sorted_idx[s];                       // np.unique(..., return_index=True): first occurrence
  const int lab = label[first];
  vlabel[v] = lab;
  uint32_t u = KEY_INF;
  if (lab > 0) {                                         // valid_idxes = np.where(voxel_label > 0)
    const uint32_t b = __float_as_uint(prio[first]);
    u = (b & 0x80000000u) ? ~b : (b | 0x80000000u);      // monotone float -> uint
    if (u == KEY_INF) u = KEY_INF - 1;
    atomicAdd(&counters[2], 1);
  }
  p_in[v] = u;
  v_in[v] = v;
}

// ---- 6. the n_sample valid voxels with the smallest priorities
__global__ __launch_bounds__(256) void select_kernel(const uint32_t* __restrict__ p_sorted, const int32_t* __restrict__ v_sorted,
                                                     int n_sample, uint8_t* __restrict__ selected,
                                                     int32_t* __restrict__ counters) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n_sample) return;
  if (p_sorted[r] != KEY_INF) {
    selected[v_sorted[r]] = 1;
    atomicAdd(&counters[3], 1);
  }
}

// ---- 7. propagate the voxel label to its points
__global__ __launch_bounds__(256) void propagate_kernel(const int32_t* __restrict__ head, const int32_t* __restrict__ rank,
                                                        const int32_t* __restrict__ sorted_idx, int n,
                                                        const uint8_t* __restrict__ selected,
                                                        const int32_t* __restrict__ vlabel, int propagate,
                                                        int32_t* __restrict__ weak, int32_t* __restrict__ counters) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= n) return;
  const int v = rank[s] - 1;
  if (selected[v] && (propagate || head[s])) {
    weak[sorted_idx[s]] = vlabel[v];
    atomicAdd(&counters[4], 1);
  }
}

__global__ void stats_kernel(const int32_t* __restrict__ counters, int32_t* __restrict__ stats) {
  if (threadIdx.x < 5) stats[threadIdx.x] = counters[threadIdx.x];
}

}  // namespace

extern "C" int64_t c3d_voxel_sampler_workspace_bytes(int n) {
  if (n <= 0) return 256;
  Workspace w;
  return (int64_t)carve(w, nullptr, n);
}

extern "C" int c3d_voxel_weak_labels(const float* xyz, int n, int stride, const int32_t* label, double voxel_size,
                                     const float* priority, int n_sample, int propagate, void* workspace,
                                     int64_t workspace_bytes, int32_t* point2voxel, int32_t* weak, int32_t* stats,
                                     c3d_stream stream) {
  C3D_REQUIRE(xyz && label && priority && workspace && weak && stats, "voxel sampler: null pointer");
  C3D_REQUIRE(n > 0 && stride >= 3, "voxel sampler: need at least one point with x, y, z");
  C3D_REQUIRE(voxel_size > 0.0, "voxel sampler: voxel_size must be positive");
  C3D_REQUIRE(n_sample >= 1 && n_sample <= n, "voxel sampler: n_sample must be in 1..n");
  Workspace w;
  const size_t need = carve(w, (char*)workspace, n);
  C3D_REQUIRE((size_t)workspace_bytes >= need, "voxel sampler: workspace too small (c3d_voxel_sampler_workspace_bytes)");
  hipStream_t st = (hipStream_t)stream;
  const int nb = (n + 255) / 256;
  C3D_REQUIRE(hipMemsetAsync(w.counters, 0, 8 * sizeof(int32_t), st) == hipSuccess, "voxel sampler: memset failed");
  C3D_REQUIRE(hipMemsetAsync(w.p_in, 0xff, (size_t)n * 4, st) == hipSuccess, "voxel sampler: memset failed");
  C3D_REQUIRE(hipMemsetAsync(w.v_in, 0, (size_t)n * 4, st) == hipSuccess, "voxel sampler: memset failed");
  C3D_REQUIRE(hipMemsetAsync(w.selected, 0, (size_t)n, st) == hipSuccess, "voxel sampler: memset failed");
  C3D_REQUIRE(hipMemsetAsync(weak, 0, (size_t)n * 4, st) == hipSuccess, "voxel sampler: memset failed");
  hipLaunchKernelGGL(origin_kernel, dim3(1), dim3(1024), 0, st, xyz, n, stride, voxel_size, w.origin);
  C3D_CHECK_LAUNCH();
  hipLaunchKernelGGL(voxel_key_kernel, dim3(nb), dim3(256), 0, st, xyz, n, stride, voxel_size, w.origin, w.k_in, w.i_in,
                     point2voxel, w.counters);
  C3D_CHECK_LAUNCH();
  size_t tb = w.tmp_bytes;
  C3D_REQUIRE(rocprim::radix_sort_pairs(w.tmp, tb, w.k_in, w.k_out, w.i_in, w.i_out, (size_t)n, 0, 63, st) == hipSuccess,
              "voxel sampler: key sort failed");
  hipLaunchKernelGGL(head_kernel, dim3(nb), dim3(256), 0, st, w.k_out, n, w.head);
  C3D_CHECK_LAUNCH();
  tb = w.tmp_bytes;
  C3D_REQUIRE(rocprim::inclusive_scan(w.tmp, tb, w.head, w.rank, (size_t)n, rocprim::plus<int32_t>(), st) == hipSuccess,
              "voxel sampler: scan failed");
  hipLaunchKernelGGL(voxel_kernel, dim3(nb), dim3(256), 0, st, w.head, w.rank, w.i_out, n, label, priority, w.vlabel, w.p_in,
                     w.v_in, w.counters);
  C3D_CHECK_LAUNCH();
  tb = w.tmp_bytes;
  C3D_REQUIRE(rocprim::radix_sort_pairs(w.tmp, tb, w.p_in, w.p_out, w.v_in, w.v_out, (size_t)n, 0, 32, st) == hipSuccess,
              "voxel sampler: priority sort failed");
  hipLaunchKernelGGL(select_kernel, dim3((n_sample + 255) / 256), dim3(256), 0, st, w.p_out, w.v_out, n_sample, w.selected,
                     w.counters);
  C3D_CHECK_LAUNCH();
  hipLaunchKernelGGL(propagate_kernel, dim3(nb), dim3(256), 0, st, w.head, w.rank, w.i_out, n, w.selected, w.vlabel, propagate,
                     weak, w.counters);
  C3D_CHECK_LAUNCH();
  hipLaunchKernelGGL(stats_kernel, dim3(1), dim3(64), 0, st, w.counters, stats);
  C3D_CHECK_LAUNCH();
  return 0;
}
