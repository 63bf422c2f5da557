// Per-iteration segmentation metrics on the device (gfx950): argmax of the class probabilities,
// un-projection from the range image to the points of the scan, confusion-matrix accumulation.
// Reference: tasks/weak_segmentation/trainer.py:713-730 (argmax + unprojection) and
// pc_processor/metrics/iou_eval.py:35-58 (IOUEval.addBatch: conf[pred][gt] += 1, int64).
// Integer work, bit-exact: block-local [C][C] histograms in LDS, one 64-bit atomic per non-zero
// cell per block (integer adds commute, so the result does not depend on scheduling).
#include "common.h"
#include "../../include/coarse3d_hip.h"

namespace {

constexpr int MAXC = 32;

__device__ __forceinline__ void flush_hist(const unsigned* hist, int C, int64_t* conf) {
  for (int i = threadIdx.x; i < C * C; i += blockDim.x)
    if (hist[i]) atomicAdd(reinterpret_cast<unsigned long long*>(conf) + i, (unsigned long long)hist[i]);
}

// pred = argmax_c prob[pix][c] (first maximum, as torch.argmax); pix = uy*W + ux, or uy alone
// when ux == NULL (SemanticPOSS path); points i >= n_valid predict class 0 (trainer.py:722-726)
__global__ __launch_bounds__(256) void unproject_confusion_kernel(const float* __restrict__ prob, int W, int C, int cstride,
                                                                  const int32_t* __restrict__ uy, const int32_t* __restrict__ ux,
                                                                  const int64_t* __restrict__ labels, int64_t n, int64_t n_valid,
                                                                  int64_t npix, int64_t* __restrict__ conf, int32_t* __restrict__ pred_out) {
  __shared__ unsigned hist[MAXC * MAXC];
  for (int i = threadIdx.x; i < C * C; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int best = 0;
    if (i < n_valid) {
      const int64_t pix = ux ? (int64_t)uy[i] * W + ux[i] : (int64_t)uy[i];
      if (pix >= 0 && pix < npix) {
        const float* p = prob + pix * cstride;
        float bv = p[0];
        for (int c = 1; c < C; ++c) {
          const float v = p[c];
          if (v > bv) {
            bv = v;
            best = c;
          }
        }
      }
    }
    if (pred_out) pred_out[i] = best;
    const int64_t g = labels[i];
    if (g >= 0 && g < C) atomicAdd(&hist[best * C + (int)g], 1u);
  }
  __syncthreads();
  flush_hist(hist, C, conf);
}

__global__ __launch_bounds__(256) void confusion_add_kernel(const int64_t* __restrict__ pred, const int64_t* __restrict__ label,
                                                            int64_t n, int C, int64_t* __restrict__ conf) {
  __shared__ unsigned hist[MAXC * MAXC];
  for (int i = threadIdx.x; i < C * C; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = pred[i], g = label[i];
    if (p >= 0 && p < C && g >= 0 && g < C) atomicAdd(&hist[(int)p * C + (int)g], 1u);
  }
  __syncthreads();
  flush_hist(hist, C, conf);
}

int blocks_for(int64_t n) {
  int64_t b = (n + 256 * 8 - 1) / (256 * 8);
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" int c3d_unproject_confusion(const float* prob, int H, int W, int C, int cstride, const int32_t* uy,
                                       const int32_t* ux, const int64_t* labels, int64_t n, int64_t n_valid,
                                       int64_t* conf, int32_t* pred_out, c3d_stream stream) {
  C3D_REQUIRE(C >= 1 && C <= MAXC, "unproject_confusion: 1..32 classes");
  C3D_REQUIRE(n_valid <= n, "unproject_confusion: n_valid must not exceed n");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(unproject_confusion_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, prob, W, C,
                     cstride, uy, ux, labels, n, n_valid, (int64_t)H * W, conf, pred_out);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_confusion_add(const int64_t* pred, const int64_t* label, int64_t n, int C, int64_t* conf,
                                 c3d_stream stream) {
  C3D_REQUIRE(C >= 1 && C <= MAXC, "confusion_add: 1..32 classes");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(confusion_add_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, pred, label, n, C,
                     conf);
  C3D_CHECK_LAUNCH();
  return 0;
}
