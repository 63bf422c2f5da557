// Helpers shared by the bf16x3 convolution kernels (conv_x3.hip, conv_pw3.hip).
#pragma once
#include "conv_common.h"

namespace {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// 4 floats -> three bf16 planes (h, m, l), each 4 bf16 packed in 2 dwords; x = h + m + l exactly
__device__ __forceinline__ void split4x3(f32x4 v, u32x2 (&out)[3]) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  f32x4 r = v;
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    bf16x4 h;
#pragma unroll
    for (int q = 0; q < 4; ++q) h[q] = (__bf16)r[q];
    out[p] = __builtin_bit_cast(u32x2, h);
    if (p < 2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) r[q] -= (float)h[q];
    }
  }
}

template <int NP>
__device__ __forceinline__ void split4_planes(f32x4 v, u32x2 (&out)[NP]) {
  if constexpr (NP == 3) {
    split4x3(v, out);
  } else {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 h;
#pragma unroll
    for (int q = 0; q < 4; ++q) h[q] = (__bf16)v[q];
    out[0] = __builtin_bit_cast(u32x2, h);
  }
}

// LDS rows are 32 B (16 channels) with NO padding; the two 16-B halves of row R are swapped when
// bit 3 of R is set, which makes every ds_read_b128 fragment read conflict-free.
// element offset (bf16 units) of channel quad c4 (0..3) inside the 16-channel row R
__device__ __forceinline__ int swz_quad(int R, int c4) { return (((c4 >> 1) ^ ((R >> 3) & 1)) << 3) + ((c4 & 1) << 2); }
// element offset of the 8-channel fragment `half` inside row R
__device__ __forceinline__ int swz_half(int R, int half) { return (half ^ ((R >> 3) & 1)) << 3; }

}  // namespace
