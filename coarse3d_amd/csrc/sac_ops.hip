// Spatially-adaptive convolution glue of SqueezeSegV3 (SURVEY 8f, N3 second backbone).
//
// Reference: pc_processor/models/squeezesegv3_Proto.py:490-503 (SACBlock.forward)
//     new_feature = unfold(feature, 3, pad 1)            [N, 9C, H, W], channel j = c*9 + tap
//     attention   = sigmoid(BN(conv7x7(xyz)))            [N, 9C, H, W]
//     new_feature = new_feature * attention  -> 1x1 conv (9C -> C) -> ...
// On this engine the 7x7 conv over the 3 coordinate channels is an im2col (147 -> 160 columns,
// built once per resolution level and shared by every SAC block of that level) followed by a
// pointwise MFMA GEMM, and the modulated unfold is one elementwise pass that writes the
// [N,H,W,9C] operand of the 1x1 GEMM.  All tensors NHWC fp32.  HBM-bound, one thread per element.
#include "common.h"
#include "../../include/coarse3d_hip.h"

namespace {

__device__ __forceinline__ size_t gtid() { return (size_t)blockIdx.x * blockDim.x + threadIdx.x; }
__device__ __forceinline__ size_t gstride() { return (size_t)gridDim.x * blockDim.x; }
int nblocks(size_t n) {
  size_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}

// xcol[p][c*49 + ky*7 + kx] = xyz[p + (ky-3, kx-3)][c]  (zero outside the image; columns 147..159 zero)
__global__ void sac_im2col7_kernel(const float* __restrict__ xyz, int B, int H, int W, int xcs, float* __restrict__ xcol) {
  const size_t total = (size_t)B * H * W * 160;
  for (size_t i = gtid(); i < total; i += gstride()) {
    const int col = i % 160;
    size_t p = i / 160;
    const int x = p % W;
    p /= W;
    const int y = p % H;
    const int b = p / H;
    float v = 0.f;
    if (col < 147) {
      const int c = col / 49, r = col % 49;
      const int yy = y + r / 7 - 3, xx = x + r % 7 - 3;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = xyz[((size_t)(b * H + yy) * W + xx) * xcs + c];
    }
    xcol[i] = v;
  }
}

__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + expf(-v)); }

// m[p][j] = feat[p + tap(j % 9)][j / 9] * sigmoid(att[p][j] * scale[j] + shift[j])
__global__ void sac_modulate_kernel(const float* __restrict__ feat, const float* __restrict__ att,
                                    const float* __restrict__ scale, const float* __restrict__ shift, int B, int H, int W,
                                    int C, float* __restrict__ m) {
  const int J = 9 * C;
  const size_t total = (size_t)B * H * W * J;
  for (size_t i = gtid(); i < total; i += gstride()) {
    const int j = i % J;
    size_t p = i / J;
    const int x = p % W;
    p /= W;
    const int y = p % H;
    const int b = p / H;
    const int c = j / 9, k = j % 9;
    const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
    float f = 0.f;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) f = feat[((size_t)(b * H + yy) * W + xx) * C + c];
    m[i] = f * sigmoidf(att[i] * scale[j] + shift[j]);
  }
}

// given dm = d(loss)/d(m):  datt[p][j] = dm * feat_tap * s * (1 - s)   (gradient w.r.t. the BatchNorm
// output in front of the sigmoid);  dm[p][j] <- dm * s  (what sac_fold scatters back to the feature)
__global__ void sac_modulate_bwd_kernel(float* __restrict__ dm, const float* __restrict__ feat, const float* __restrict__ att,
                                        const float* __restrict__ scale, const float* __restrict__ shift, int B, int H,
                                        int W, int C, float* __restrict__ datt) {
  const int J = 9 * C;
  const size_t total = (size_t)B * H * W * J;
  for (size_t i = gtid(); i < total; i += gstride()) {
    const int j = i % J;
    size_t p = i / J;
    const int x = p % W;
    p /= W;
    const int y = p % H;
    const int b = p / H;
    const int c = j / 9, k = j % 9;
    const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
    float f = 0.f;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) f = feat[((size_t)(b * H + yy) * W + xx) * C + c];
    const float s = sigmoidf(att[i] * scale[j] + shift[j]);
    const float g = dm[i];
    datt[i] = g * f * s * (1.f - s);
    dm[i] = g * s;
  }
}

// dfeat[q][c] (+)= sum_k t[q - tap(k)][c*9 + k]   (adjoint of the unfold)
__global__ void sac_fold_kernel(const float* __restrict__ t, int B, int H, int W, int C, int accumulate,
                                float* __restrict__ dfeat) {
  const size_t total = (size_t)B * H * W * C;
  const int J = 9 * C;
  for (size_t i = gtid(); i < total; i += gstride()) {
    const int c = i % C;
    size_t p = i / C;
    const int x = p % W;
    p /= W;
    const int y = p % H;
    const int b = p / H;
    float acc = accumulate ? dfeat[i] : 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int yy = y - (k / 3 - 1), xx = x - (k % 3 - 1);
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) acc += t[((size_t)(b * H + yy) * W + xx) * J + c * 9 + k];
    }
    dfeat[i] = acc;
  }
}

}  // namespace

extern "C" int c3d_sac_im2col7(const float* xyz, int B, int H, int W, int xcs, float* xcol, c3d_stream stream) {
  C3D_REQUIRE(xyz && xcol && B > 0 && H > 0 && W > 0 && xcs >= 3, "sac_im2col7: bad arguments");
  hipLaunchKernelGGL(sac_im2col7_kernel, dim3(nblocks((size_t)B * H * W * 160)), dim3(256), 0, (hipStream_t)stream, xyz, B, H,
                     W, xcs, xcol);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_sac_modulate(const float* feat, const float* att, const float* scale, const float* shift, int B, int H,
                                int W, int C, float* m, c3d_stream stream) {
  C3D_REQUIRE(feat && att && scale && shift && m && C > 0, "sac_modulate: bad arguments");
  hipLaunchKernelGGL(sac_modulate_kernel, dim3(nblocks((size_t)B * H * W * 9 * C)), dim3(256), 0, (hipStream_t)stream, feat,
                     att, scale, shift, B, H, W, C, m);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_sac_modulate_bwd(float* dm, const float* feat, const float* att, const float* scale, const float* shift,
                                    int B, int H, int W, int C, float* datt, c3d_stream stream) {
  C3D_REQUIRE(dm && feat && att && scale && shift && datt && C > 0, "sac_modulate_bwd: bad arguments");
  hipLaunchKernelGGL(sac_modulate_bwd_kernel, dim3(nblocks((size_t)B * H * W * 9 * C)), dim3(256), 0, (hipStream_t)stream, dm,
                     feat, att, scale, shift, B, H, W, C, datt);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_sac_fold(const float* t, int B, int H, int W, int C, int accumulate, float* dfeat, c3d_stream stream) {
  C3D_REQUIRE(t && dfeat && C > 0, "sac_fold: bad arguments");
  hipLaunchKernelGGL(sac_fold_kernel, dim3(nblocks((size_t)B * H * W * C)), dim3(256), 0, (hipStream_t)stream, t, B, H, W, C,
                     accumulate, dfeat);
  C3D_CHECK_LAUNCH();
  return 0;
}
