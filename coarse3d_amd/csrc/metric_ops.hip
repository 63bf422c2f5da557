// Per-iteration segmentation metrics on the device (gfx950): argmax of the class probabilities,
// un-projection from the range image to the points of the scan, confusion-matrix accumulation.
// Reference: tasks/weak_segmentation/trainer.py:713-730 (argmax + unprojection) and
// pc_processor/metrics/iou_eval.py:35-58 (IOUEval.addBatch: conf[pred][gt] += 1, int64).
// Integer work, bit-exact: block-local [C][C] histograms in LDS, one 64-bit atomic per non-zero
// cell per block (integer adds commute, so the result does not depend on scheduling).
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
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

// ====================================================================== loss head
// Focal loss on probabilities (pc_processor/loss/focal_softmax.py:30-77) and Lovasz-softmax
// (pc_processor/loss/lovasz_softmax.py:101-176, classes='present', per_image=False) on the
// labelled pixels, forward and backward, without the ~40 stock-op launches of the PyTorch path.
namespace {

// ---- focal: loss_i = -(1-pt)^gamma * log(max(pt,1e-6)) * alpha[t] over pixels with mask != 0
//      partial[block] = (sum loss_i, count); deterministic two-stage reduction
__global__ __launch_bounds__(256) void focal_fwd_kernel(const float* __restrict__ prob, int C, int cstride,
                                                        const int64_t* __restrict__ target, const uint8_t* __restrict__ mask,
                                                        const float* __restrict__ alpha, float gamma, int64_t n,
                                                        double* __restrict__ partial) {
  __shared__ double red[2][4];
  double s = 0.0, cnt = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    if (mask && !mask[i]) continue;
    const int64_t t = target[i];
    if (t < 0 || t >= C) continue;
    const float pt = prob[i * cstride + t];
    const float l = -powf(1.f - pt, gamma) * logf(fmaxf(pt, 1e-6f)) * alpha[t];
    s += (double)l;
    cnt += 1.0;
  }
  s = c3d_wave_sum_d(s);
  cnt = c3d_wave_sum_d(cnt);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wv] = s;
    red[1][wv] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    partial[2 * blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

// out[0] = sum / count (0 when no pixel is selected: focal_softmax.py:67-73 returns 0 for NaN), out[1] = count
__global__ void focal_finish_kernel(const double* __restrict__ partial, int nblk, float* __restrict__ out) {
  double s = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 64) {
    s += partial[2 * i];
    c += partial[2 * i + 1];
  }
  s = c3d_wave_sum_d(s);
  c = c3d_wave_sum_d(c);
  if (threadIdx.x == 0) {
    out[0] = c > 0.0 ? (float)(s / c) : 0.f;
    out[1] = (float)c;
  }
}

// dprob[i][t] += g * d(mean loss)/d pt ;  g = *gscale, count = stats[1]
__global__ __launch_bounds__(256) void focal_bwd_kernel(const float* __restrict__ prob, int C, int cstride,
                                                        const int64_t* __restrict__ target, const uint8_t* __restrict__ mask,
                                                        const float* __restrict__ alpha, float gamma, int64_t n,
                                                        const float* __restrict__ stats, const float* __restrict__ gscale,
                                                        float* __restrict__ dprob, int dstride) {
  const float cnt = stats[1];
  if (cnt <= 0.f) return;
  const float g = (gscale ? *gscale : 1.f) / cnt;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    if (mask && !mask[i]) continue;
    const int64_t t = target[i];
    if (t < 0 || t >= C) continue;
    const float pt = prob[i * cstride + t];
    const float om = 1.f - pt;
    // d/dpt [ -(1-pt)^g * log(clamp(pt)) ] = g (1-pt)^(g-1) log(clamp(pt)) - (1-pt)^g / pt * [pt > 1e-6]
    const float dl = gamma * powf(om, gamma - 1.f) * logf(fmaxf(pt, 1e-6f)) - (pt > 1e-6f ? powf(om, gamma) / pt : 0.f);
    dprob[i * dstride + t] += g * dl * alpha[t];
  }
}

// ---- Lovasz: one workgroup per class over the P labelled pixels (P <= CAP, LDS-resident)
// P_dev != nullptr: the number of labelled pixels is read from device memory (clamped to `P`, which is then the
// capacity of idx and the row stride of grad) -- the launch is shape-static and capturable in a hipGraph.
template <int CAP>
__global__ __launch_bounds__(256) void lovasz_class_kernel(const float* __restrict__ prob, int cstride,
                                                           const int64_t* __restrict__ labels, const int64_t* __restrict__ idx,
                                                           int P, const int* __restrict__ P_dev, float* __restrict__ loss_c,
                                                           float* __restrict__ present, float* __restrict__ grad /* [C][stride] */) {
  extern __shared__ float lsm[];
  __shared__ float wsum[4];
  __shared__ double dsum[4];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int gstride = P;
  if (P_dev) P = min(max(*P_dev, 0), P);
  int npow = 1;
  while (npow < P) npow <<= 1;                              // <= CAP (checked by the host)
  float* key = lsm;                                         // [npow] errors, sorted descending
  unsigned* val = reinterpret_cast<unsigned*>(lsm + npow);  // [npow] 2*p + fg
  float fgs = 0.f;
  for (int p = tid; p < npow; p += 256) {
    float e = -1.f;                                       // padding sorts behind every real error (>= 0)
    unsigned v = 0;
    if (p < P) {
      const int64_t i = idx[p];
      const unsigned fg = labels[i] == c;
      e = fabsf((float)fg - prob[i * cstride + c]);
      v = 2u * (unsigned)p + fg;
      fgs += (float)fg;
    }
    key[p] = e;
    val[p] = v;
  }
  fgs = c3d_wave_sum(fgs);
  if (lane == 0) wsum[wv] = fgs;
  __syncthreads();
  const float gts = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  if (gts == 0.f) {                                       // class absent: skipped by classes='present'
    if (tid == 0) {
      loss_c[c] = 0.f;
      present[c] = 0.f;
    }
    for (int p = tid; p < P; p += 256) grad[(size_t)c * gstride + p] = 0.f;
    return;
  }
  // bitonic sort, descending by key
  for (int k = 2; k <= npow; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int i = tid; i < npow; i += 256) {
        const int l = i ^ j;
        if (l > i) {
          const bool desc = (i & k) == 0;
          const float a = key[i], b = key[l];
          if (desc ? (a < b) : (a > b)) {
            key[i] = b;
            key[l] = a;
            const unsigned t = val[i];
            val[i] = val[l];
            val[l] = t;
          }
        }
      }
    }
  }
  __syncthreads();
  // prefix sums of fg over the sorted order: each thread owns a contiguous run of `per` ranks
  const int per = (P + 255) / 256;
  const int r0 = tid * per, r1 = min(P, r0 + per);
  float run = 0.f;
  for (int r = r0; r < r1; ++r) run += (float)(val[r] & 1u);
  float scan = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float v = __shfl_up(scan, o, 64);
    if (lane >= o) scan += v;
  }
  __syncthreads();
  if (lane == 63) wsum[wv] = scan;
  __syncthreads();
  float cum = scan - run;                                 // fg before this thread's run
  for (int k = 0; k < wv; ++k) cum += wsum[k];
  // jaccard gradient (lovasz_softmax.py:56-68) + loss + per-pixel gradient
  float jprev = 0.f;
  if (r0 > 0 && r0 < P) {                                 // jaccard at rank r0-1
    const float inter = gts - cum, uni = gts + ((float)r0 - cum);
    jprev = 1.f - inter / uni;
  }
  double acc = 0.0;
  for (int r = r0; r < r1; ++r) {
    const unsigned v = val[r];
    const float fg = (float)(v & 1u);
    cum += fg;
    const float inter = gts - cum, uni = gts + ((float)(r + 1) - cum);
    const float jac = 1.f - inter / uni;
    const float jd = r == 0 ? jac : jac - jprev;
    jprev = jac;
    const float e = key[r];
    acc += (double)(e * jd);
    // d|fg - p|/dp = -1 (fg = 1), +1 (fg = 0); 0 where the error is exactly 0 (abs'(0) = 0)
    const float sgn = e == 0.f ? 0.f : (fg != 0.f ? -1.f : 1.f);
    grad[(size_t)c * gstride + (v >> 1)] = jd * sgn;
  }
  acc = c3d_wave_sum_d(acc);
  if (lane == 0) dsum[wv] = acc;
  __syncthreads();
  if (tid == 0) {
    loss_c[c] = (float)(dsum[0] + dsum[1] + dsum[2] + dsum[3]);
    present[c] = 1.f;
  }
}

// out[0] = mean over present classes of loss_c (0 if none), out[1] = number of present classes
__global__ void lovasz_finish_kernel(const float* __restrict__ loss_c, const float* __restrict__ present, int C,
                                     float* __restrict__ out) {
  float s = 0.f, n = 0.f;
  for (int c = 0; c < C; ++c) {
    s += loss_c[c] * present[c];
    n += present[c];
  }
  out[0] = n > 0.f ? s / n : 0.f;
  out[1] = n;
}

// dprob[idx[p]][c] += g / npresent * grad[c][p]
__global__ __launch_bounds__(256) void lovasz_bwd_kernel(const float* __restrict__ grad, const int64_t* __restrict__ idx, int P,
                                                         const int* __restrict__ P_dev, int C, const float* __restrict__ stats,
                                                         const float* __restrict__ gscale, float* __restrict__ dprob, int dstride) {
  const float np_ = stats[1];
  if (np_ <= 0.f) return;
  const int gstride = P;
  if (P_dev) P = min(max(*P_dev, 0), P);
  const float g = (gscale ? *gscale : 1.f) / np_;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)P * C; e += (int64_t)gridDim.x * 256) {
    const int p = (int)(e / C), c = (int)(e % C);
    dprob[idx[p] * dstride + c] += g * grad[(size_t)c * gstride + p];
  }
}

}  // namespace

extern "C" int c3d_focal_forward(const float* prob, int C, int cstride, const int64_t* target, const uint8_t* mask,
                                 const float* alpha, float gamma, int64_t n, double* partial, int nblk, float* out,
                                 c3d_stream stream) {
  C3D_REQUIRE(nblk >= 1 && nblk <= 4096, "focal: 1..4096 partial blocks");
  hipLaunchKernelGGL(focal_fwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, prob, C, cstride, target, mask, alpha,
                     gamma, n, partial);
  C3D_CHECK_LAUNCH();
  hipLaunchKernelGGL(focal_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, nblk, out);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_focal_backward(const float* prob, int C, int cstride, const int64_t* target, const uint8_t* mask,
                                  const float* alpha, float gamma, int64_t n, const float* stats, const float* gscale,
                                  float* dprob, int dstride, c3d_stream stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(focal_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, prob, C, cstride, target, mask,
                     alpha, gamma, n, stats, gscale, dprob, dstride);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_lovasz_max_pixels(void) { return 8192; }

static int lovasz_forward_impl(const float* prob, int C, int cstride, const int64_t* labels, const int64_t* idx, int P,
                               const int* P_dev, float* loss_c, float* present, float* grad, float* out, hipStream_t st) {
  C3D_REQUIRE(P >= 0 && P <= 8192, "lovasz: at most 8192 labelled pixels on the fused path");
  C3D_REQUIRE(C >= 1 && C <= 64, "lovasz: 1..64 classes");
  if (P == 0) {
    (void)hipMemsetAsync(loss_c, 0, sizeof(float) * C, st);
    (void)hipMemsetAsync(present, 0, sizeof(float) * C, st);
  } else {
    c3d_opt_in_lds<&lovasz_class_kernel<8192>>(8192 * 8);
    int npow = 1;
    while (npow < P) npow <<= 1;
    hipLaunchKernelGGL(lovasz_class_kernel<8192>, dim3(C), dim3(256), (size_t)npow * 8, st, prob, cstride, labels, idx, P, P_dev,
                       loss_c, present, grad);
    C3D_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(lovasz_finish_kernel, dim3(1), dim3(1), 0, st, loss_c, present, C, out);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_lovasz_forward(const float* prob, int C, int cstride, const int64_t* labels, const int64_t* idx, int P,
                                  float* loss_c, float* present, float* grad, float* out, c3d_stream stream) {
  return lovasz_forward_impl(prob, C, cstride, labels, idx, P, nullptr, loss_c, present, grad, out, (hipStream_t)stream);
}

extern "C" int c3d_lovasz_backward(const float* grad, const int64_t* idx, int P, int C, const float* stats,
                                   const float* gscale, float* dprob, int dstride, c3d_stream stream) {
  if (P <= 0) return 0;
  hipLaunchKernelGGL(lovasz_bwd_kernel, dim3(blocks_for((int64_t)P * C)), dim3(256), 0, (hipStream_t)stream, grad, idx, P, nullptr,
                     C, stats, gscale, dprob, dstride);
  C3D_CHECK_LAUNCH();
  return 0;
}

// ---- Lovasz beyond the LDS capacity (fully supervised batches: up to B*H*W labelled pixels).  Same arithmetic as
//      lovasz_class_kernel; the per-class sort of the P errors is a device-wide SEGMENTED radix sort (rocPRIM, the vendor's
//      sort primitive -- as in csrc/voxel_ops.hip), the prefix sums run over chunks of LV_CHUNK ranks.
namespace {
constexpr int LV_CHUNK = 4096;

// keys[c][p] = |fg - prob[idx[p]][c]|, vals[c][p] = 2p + fg
__global__ __launch_bounds__(256) void lovasz_fill_kernel(const float* __restrict__ prob, int cstride, const int64_t* __restrict__ labels,
                                                          const int64_t* __restrict__ idx, int P, int C, float* __restrict__ keys,
                                                          uint32_t* __restrict__ vals) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)P * C; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e / P), p = (int)(e % P);
    const int64_t i = idx[p];
    const uint32_t fg = labels[i] == c;
    keys[e] = fabsf((float)fg - prob[i * cstride + c]);
    vals[e] = 2u * (uint32_t)p + fg;
  }
}

// cnt[c][chunk] = foreground pixels among the chunk's ranks of the sorted order
__global__ __launch_bounds__(256) void lovasz_count_kernel(const uint32_t* __restrict__ vals, int P, int nchunk, int32_t* __restrict__ cnt) {
  __shared__ int red[4];
  const int c = blockIdx.y, ch = blockIdx.x;
  const int r0 = ch * LV_CHUNK, r1 = min(P, r0 + LV_CHUNK);
  int n = 0;
  for (int r = r0 + threadIdx.x; r < r1; r += 256) n += (int)(vals[(size_t)c * P + r] & 1u);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) cnt[c * nchunk + ch] = red[0] + red[1] + red[2] + red[3];
}

// jaccard gradient of one chunk of ranks: grad[c][p] = jd * sign, part[c][chunk] = sum e * jd
__global__ __launch_bounds__(256) void lovasz_chunk_kernel(const float* __restrict__ keys, const uint32_t* __restrict__ vals, int P,
                                                           int nchunk, const int32_t* __restrict__ cnt, float* __restrict__ grad,
                                                           double* __restrict__ part) {
  __shared__ float wsum[4];
  __shared__ double dsum[4];
  const int c = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int before = 0, total = 0;
  for (int k = 0; k < nchunk; ++k) {
    const int n = cnt[c * nchunk + k];
    total += n;
    if (k < ch) before += n;
  }
  if (total == 0) {                                       // class absent: classes='present' skips it
    const int r0 = ch * LV_CHUNK, r1 = min(P, r0 + LV_CHUNK);
    for (int r = r0 + tid; r < r1; r += 256) grad[(size_t)c * P + (vals[(size_t)c * P + r] >> 1)] = 0.f;
    if (tid == 0) part[c * nchunk + ch] = 0.0;
    return;
  }
  const float gts = (float)total;
  constexpr int PER = LV_CHUNK / 256;                     // contiguous ranks per thread
  const int r0 = ch * LV_CHUNK + tid * PER, r1 = min(P, r0 + PER);
  float run = 0.f;
  for (int r = r0; r < r1; ++r) run += (float)(vals[(size_t)c * P + r] & 1u);
  float scan = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float v = __shfl_up(scan, o, 64);
    if (lane >= o) scan += v;
  }
  if (lane == 63) wsum[wv] = scan;
  __syncthreads();
  float cum = (float)before + scan - run;                 // foreground before this thread's run
  for (int k = 0; k < wv; ++k) cum += wsum[k];
  float jprev = 0.f;
  if (r0 > 0 && r0 < P) {
    const float inter = gts - cum, uni = gts + ((float)r0 - cum);
    jprev = 1.f - inter / uni;
  }
  double acc = 0.0;
  for (int r = r0; r < r1; ++r) {
    const uint32_t v = vals[(size_t)c * P + r];
    const float fg = (float)(v & 1u);
    cum += fg;
    const float inter = gts - cum, uni = gts + ((float)(r + 1) - cum);
    const float jac = 1.f - inter / uni;
    const float jd = r == 0 ? jac : jac - jprev;
    jprev = jac;
    const float e = keys[(size_t)c * P + r];
    acc += (double)(e * jd);
    const float sgn = e == 0.f ? 0.f : (fg != 0.f ? -1.f : 1.f);
    grad[(size_t)c * P + (v >> 1)] = jd * sgn;
  }
  acc = c3d_wave_sum_d(acc);
  if (lane == 0) dsum[wv] = acc;
  __syncthreads();
  if (tid == 0) part[c * nchunk + ch] = dsum[0] + dsum[1] + dsum[2] + dsum[3];
}

__global__ void lovasz_large_finish_kernel(const double* __restrict__ part, const int32_t* __restrict__ cnt, int nchunk, int C,
                                           float* __restrict__ loss_c, float* __restrict__ present, float* __restrict__ out) {
  float s = 0.f, n = 0.f;
  for (int c = 0; c < C; ++c) {
    double l = 0.0;
    int total = 0;
    for (int k = 0; k < nchunk; ++k) {
      l += part[c * nchunk + k];
      total += cnt[c * nchunk + k];
    }
    loss_c[c] = total > 0 ? (float)l : 0.f;
    present[c] = total > 0 ? 1.f : 0.f;
    s += loss_c[c] * present[c];
    n += present[c];
  }
  out[0] = n > 0.f ? s / n : 0.f;
  out[1] = n;
}

struct LovaszWs {
  float *k_in, *k_out;
  uint32_t *v_in, *v_out;
  int32_t *offsets, *cnt;
  double* part;
  void* tmp;
  size_t tmp_bytes;
};
size_t lovasz_align(size_t b) { return (b + 255) / 256 * 256; }
size_t lovasz_sort_tmp(int C, int64_t n) {
  size_t t = 0;
  (void)rocprim::segmented_radix_sort_pairs_desc(nullptr, t, (const float*)nullptr, (float*)nullptr, (const uint32_t*)nullptr,
                                                 (uint32_t*)nullptr, (unsigned)n, (unsigned)C, (const int32_t*)nullptr,
                                                 (const int32_t*)nullptr, 0, 32, (hipStream_t)0);
  return t;
}
size_t lovasz_carve(LovaszWs& w, char* base, int C, int P) {
  const size_t n = (size_t)C * P;
  const int nchunk = (P + LV_CHUNK - 1) / LV_CHUNK;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += lovasz_align(bytes);
    return p;
  };
  w.k_in = (float*)take(n * 4);
  w.k_out = (float*)take(n * 4);
  w.v_in = (uint32_t*)take(n * 4);
  w.v_out = (uint32_t*)take(n * 4);
  w.offsets = (int32_t*)take((size_t)(C + 1) * 4);
  w.cnt = (int32_t*)take((size_t)C * nchunk * 4);
  w.part = (double*)take((size_t)C * nchunk * 8);
  w.tmp_bytes = lovasz_sort_tmp(C, (int64_t)n);
  w.tmp = take(w.tmp_bytes);
  return off;
}
__global__ void lovasz_offsets_kernel(int32_t* offsets, int C, int P) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c <= C) offsets[c] = c * P;
}
}  // namespace

extern "C" int64_t c3d_lovasz_workspace_bytes(int C, int P) {
  if (C < 1 || P < 1) return 0;
  LovaszWs w;
  return (int64_t)lovasz_carve(w, nullptr, C, P);
}

extern "C" int c3d_lovasz_forward_large(const float* prob, int C, int cstride, const int64_t* labels, const int64_t* idx, int P,
                                        float* loss_c, float* present, float* grad, float* out, void* workspace,
                                        int64_t workspace_bytes, c3d_stream stream) {
  C3D_REQUIRE(C >= 1 && C <= 64 && P >= 1, "lovasz_large: 1..64 classes, at least one labelled pixel");
  C3D_REQUIRE((int64_t)C * P < (1ll << 31), "lovasz_large: C * P must stay below 2^31");
  C3D_REQUIRE(workspace != nullptr && workspace_bytes >= c3d_lovasz_workspace_bytes(C, P), "lovasz_large: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  LovaszWs w;
  (void)lovasz_carve(w, (char*)workspace, C, P);
  const int nchunk = (P + LV_CHUNK - 1) / LV_CHUNK;
  hipLaunchKernelGGL(lovasz_offsets_kernel, dim3((C + 64) / 64), dim3(64), 0, st, w.offsets, C, P);
  C3D_CHECK_LAUNCH();
  hipLaunchKernelGGL(lovasz_fill_kernel, dim3(blocks_for((int64_t)P * C)), dim3(256), 0, st, prob, cstride, labels, idx, P, C, w.k_in,
                     w.v_in);
  C3D_CHECK_LAUNCH();
  size_t tb = w.tmp_bytes;
  C3D_REQUIRE(rocprim::segmented_radix_sort_pairs_desc(w.tmp, tb, w.k_in, w.k_out, w.v_in, w.v_out, (unsigned)((size_t)C * P),
                                                       (unsigned)C, w.offsets, w.offsets + 1, 0, 32, st) == hipSuccess,
              "lovasz_large: segmented sort failed");
  hipLaunchKernelGGL(lovasz_count_kernel, dim3(nchunk, C), dim3(256), 0, st, w.v_out, P, nchunk, w.cnt);
  C3D_CHECK_LAUNCH();
  hipLaunchKernelGGL(lovasz_chunk_kernel, dim3(nchunk, C), dim3(256), 0, st, w.k_out, w.v_out, P, nchunk, w.cnt, grad, w.part);
  C3D_CHECK_LAUNCH();
  hipLaunchKernelGGL(lovasz_large_finish_kernel, dim3(1), dim3(1), 0, st, w.part, w.cnt, nchunk, C, loss_c, present, out);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_lovasz_forward_dyn(const float* prob, int C, int cstride, const int64_t* labels, const int64_t* idx,
                                      const int* P_dev, int P_cap, float* loss_c, float* present, float* grad, float* out,
                                      c3d_stream stream) {
  C3D_REQUIRE(P_dev != nullptr && P_cap >= 1, "lovasz_dyn: needs the device count and a capacity");
  return lovasz_forward_impl(prob, C, cstride, labels, idx, P_cap, P_dev, loss_c, present, grad, out, (hipStream_t)stream);
}

extern "C" int c3d_lovasz_backward_dyn(const float* grad, const int64_t* idx, const int* P_dev, int P_cap, int C,
                                       const float* stats, const float* gscale, float* dprob, int dstride, c3d_stream stream) {
  C3D_REQUIRE(P_dev != nullptr && P_cap >= 1, "lovasz_dyn: needs the device count and a capacity");
  hipLaunchKernelGGL(lovasz_bwd_kernel, dim3(blocks_for((int64_t)P_cap * C)), dim3(256), 0, (hipStream_t)stream, grad, idx, P_cap,
                     P_dev, C, stats, gscale, dprob, dstride);
  C3D_CHECK_LAUNCH();
  return 0;
}

// ====================================================================== kNN label clean-up
// pc_processor/postproc/knn.py:36-142 (KNN.forward): every point looks at the S x S window of
// the range image around its pixel, keeps the K neighbours whose range is closest to its own
// (differences weighted by 1 - gaussian(offset), the centre replaced by the point itself), and
// takes the majority label among them (labels 1..C-1; neighbours beyond `cutoff` vote for nothing).
// One thread per point; the S*S <= 49 candidates live in registers/LDS-free local arrays.
namespace {

template <int SMAX>
__global__ __launch_bounds__(256) void knn_vote_kernel(const float* __restrict__ proj_range, const int64_t* __restrict__ proj_argmax,
                                                       int H, int W, const float* __restrict__ unproj_range,
                                                       const int64_t* __restrict__ px, const int64_t* __restrict__ py, int64_t n,
                                                       const float* __restrict__ inv_gauss, int S, int K, float cutoff, int C,
                                                       int64_t* __restrict__ out) {
  const int pad = (S - 1) / 2, SS = S * S, center = (SS - 1) / 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int x = (int)px[i], y = (int)py[i];
    const float r0 = unproj_range[i];
    float bd[SMAX];     // K best distances, ascending
    int bl[SMAX];       // their labels
    int nb = 0;
    for (int k = 0; k < SS; ++k) {
      const int yy = y + k / S - pad, xx = x + k % S - pad;
      const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
      float r = in ? proj_range[(size_t)yy * W + xx] : 0.f;          // F.unfold pads with zeros
      if (r < 0.f) r = INFINITY;                                     // knn.py:88 (invalid pixels)
      if (k == center) r = r0;                                       // knn.py:92
      const float d = fabsf(r - r0) * inv_gauss[k];
      const int lab = in ? (int)proj_argmax[(size_t)yy * W + xx] : 0;
      // insert into the sorted list of the K smallest (earlier candidate wins ties)
      if (nb < K || d < bd[nb - 1]) {
        int j = nb < K ? nb : K - 1;
        while (j > 0 && bd[j - 1] > d) {
          bd[j] = bd[j - 1];
          bl[j] = bl[j - 1];
          --j;
        }
        bd[j] = d;
        bl[j] = lab;
        if (nb < K) ++nb;
      }
    }
    // vote over classes 1..C-1 (knn.py:124-133); first maximum wins
    int best = 1, bestv = -1;
    for (int c = 1; c < C; ++c) {
      int v = 0;
      for (int j = 0; j < nb; ++j) {
        const bool valid = !(cutoff > 0.f && bd[j] > cutoff);
        v += (valid && bl[j] == c) ? 1 : 0;
      }
      if (v > bestv) {
        bestv = v;
        best = c;
      }
    }
    out[i] = best;
  }
}

}  // namespace

extern "C" int c3d_knn_vote(const float* proj_range, const int64_t* proj_argmax, int H, int W, const float* unproj_range,
                            const int64_t* px, const int64_t* py, int64_t n, const float* inv_gauss, int search, int knn,
                            float cutoff, int nclasses, int64_t* out, c3d_stream stream) {
  C3D_REQUIRE(search % 2 == 1 && search >= 1 && search <= 7, "knn: search window must be odd and <= 7");
  C3D_REQUIRE(knn >= 1 && knn <= search * search, "knn: 1 <= knn <= search^2");
  C3D_REQUIRE(nclasses >= 2, "knn: at least two classes");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(knn_vote_kernel<49>, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, proj_range, proj_argmax, H, W,
                     unproj_range, px, py, n, inv_gauss, search, knn, cutoff, nclasses, out);
  C3D_CHECK_LAUNCH();
  return 0;
}
