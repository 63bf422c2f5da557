// Shared device/host helpers for the COARSE3D gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Opt a kernel into more than 64 KB of dynamic LDS.  The attribute is PER DEVICE: once per (kernel instance, device),
// lock-free (a race between two host threads only repeats the cheap, idempotent call).
#include <atomic>
template <auto Kernel>
inline void c3d_opt_in_lds(int bytes = 160 * 1024) {
  static std::atomic<uint64_t> seen{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (!(seen.load(std::memory_order_relaxed) & bit)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    seen.fetch_or(bit, std::memory_order_relaxed);
  }
}

#define C3D_LRELU_SLOPE 0.01f
#define C3D_MAX_SRC 3
#define C3D_MAX_TAPS 9

// LeakyReLU as max(v, slope * v): the same value as v > 0 ? v : slope * v for every slope in [0, 1] (the entry points
// refuse others), two VALU instructions instead of three (compare, multiply, select) in every staging loop
__device__ __forceinline__ float c3d_lrelu(float v) { return __builtin_fmaxf(v, C3D_LRELU_SLOPE * v); }
__device__ __forceinline__ float c3d_lrelu(float v, float slope) { return __builtin_fmaxf(v, slope * v); }
// descriptors carry the slope as a float where 0 means the SalsaNext default (0.01)
inline float c3d_slope_or_default(float s) { return s > 0.f ? s : C3D_LRELU_SLOPE; }

// 8 floats -> 8 bf16 (round to nearest even; v_cvt_pk_bf16_f32 on gfx950): one operand of
// v_mfma_f32_32x32x16_bf16
__device__ __forceinline__ bf16x8 c3d_pack_bf16x8(f32x4 a, f32x4 b) {
  bf16x8 r;
  r[0] = (__bf16)a[0]; r[1] = (__bf16)a[1]; r[2] = (__bf16)a[2]; r[3] = (__bf16)a[3];
  r[4] = (__bf16)b[0]; r[5] = (__bf16)b[1]; r[6] = (__bf16)b[2]; r[7] = (__bf16)b[3];
  return r;
}

// ---- activation storage: fp32 or bf16 behind the same (float*) signatures; `i` counts ELEMENTS.
// The flag is uniform over a launch, so the branch costs nothing.
typedef unsigned int c3d_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 c3d_ld4(const float* p, size_t i, bool bf) {
  if (bf) {
    const c3d_u32x2 r = *reinterpret_cast<const c3d_u32x2*>(reinterpret_cast<const unsigned short*>(p) + i);
    f32x4 v;
    v[0] = __uint_as_float(r[0] << 16);
    v[1] = __uint_as_float(r[0] & 0xffff0000u);
    v[2] = __uint_as_float(r[1] << 16);
    v[3] = __uint_as_float(r[1] & 0xffff0000u);
    return v;
  }
  return *reinterpret_cast<const f32x4*>(p + i);
}
__device__ __forceinline__ void c3d_st4(float* p, size_t i, bool bf, f32x4 v) {
  if (bf) {
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    bf16x4_t h;
#pragma unroll
    for (int q = 0; q < 4; ++q) h[q] = (__bf16)v[q];
    *reinterpret_cast<c3d_u32x2*>(reinterpret_cast<unsigned short*>(p) + i) = __builtin_bit_cast(c3d_u32x2, h);
  } else {
    *reinterpret_cast<f32x4*>(p + i) = v;
  }
}
__device__ __forceinline__ float c3d_ld1(const float* p, size_t i, bool bf) {
  if (bf) return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(p)[i] << 16);
  return p[i];
}
__device__ __forceinline__ void c3d_st1(float* p, size_t i, bool bf, float v) {
  if (bf) reinterpret_cast<__bf16*>(p)[i] = (__bf16)v;
  else p[i] = v;
}

// V = 4 or 8 channels per lane.  With bf16 storage only the 8-wide form keeps every lane on a 16-byte access
// (8-byte accesses run at roughly half the instruction rate: the HBM-bound kernels measured SLOWER with bf16
// tensors and 4-wide lanes than with fp32 tensors).
template <int V>
struct c3d_vec {
  float v[V];
  __device__ __forceinline__ c3d_vec& operator+=(const c3d_vec& o) {
#pragma unroll
    for (int q = 0; q < V; ++q) v[q] += o.v[q];
    return *this;
  }
  __device__ __forceinline__ c3d_vec& operator*=(const c3d_vec& o) {
#pragma unroll
    for (int q = 0; q < V; ++q) v[q] *= o.v[q];
    return *this;
  }
  __device__ __forceinline__ c3d_vec& operator*=(float s) {
#pragma unroll
    for (int q = 0; q < V; ++q) v[q] *= s;
    return *this;
  }
};
template <int V>
__device__ __forceinline__ c3d_vec<V> c3d_vzero() {
  c3d_vec<V> r;
#pragma unroll
  for (int q = 0; q < V; ++q) r.v[q] = 0.f;
  return r;
}
template <int V>
__device__ __forceinline__ c3d_vec<V> c3d_vld(const float* p, size_t i, bool bf) {
  c3d_vec<V> r;
  if (V == 8 && bf) {
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t w = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const unsigned short*>(p) + i);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      r.v[2 * q] = __uint_as_float(w[q] << 16);
      r.v[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u);
    }
    return r;
  }
#pragma unroll
  for (int h = 0; h < V / 4; ++h) {
    const f32x4 t = c3d_ld4(p, i + 4 * h, bf);
#pragma unroll
    for (int q = 0; q < 4; ++q) r.v[4 * h + q] = t[q];
  }
  return r;
}
// fp32 per-channel vectors (scale, shift, masks)
template <int V>
__device__ __forceinline__ c3d_vec<V> c3d_vldf(const float* p, size_t i) {
  return c3d_vld<V>(p, i, false);
}
template <int V>
__device__ __forceinline__ void c3d_vst(float* p, size_t i, bool bf, const c3d_vec<V>& x) {
  if (V == 8 && bf) {
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    bf16x8_t h;
#pragma unroll
    for (int q = 0; q < 8; ++q) h[q] = (__bf16)x.v[q];
    *reinterpret_cast<u32x4_t*>(reinterpret_cast<unsigned short*>(p) + i) = __builtin_bit_cast(u32x4_t, h);
    return;
  }
#pragma unroll
  for (int h = 0; h < V / 4; ++h) c3d_st4(p, i + 4 * h, bf, f32x4{x.v[4 * h], x.v[4 * h + 1], x.v[4 * h + 2], x.v[4 * h + 3]});
}

// Streaming form of c3d_vst (nontemporal hint).  Used where a kernel's READS live on L2 hits that its own large output
// would evict: the bilinear upsampling (four taps per output, 1 GB written) ran 467 -> 234 us with it.  Measured
// neutral for the whole step when applied to every glue / BatchNorm store (197.4 vs 196.6 img/s): consumers that
// follow immediately lose as much as the producers gain.
template <int V>
__device__ __forceinline__ void c3d_vst_nt(float* p, size_t i, bool bf, const c3d_vec<V>& x) {
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  if (bf) {
    if constexpr (V == 8) {
      typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
      bf16x8_t h;
#pragma unroll
      for (int q = 0; q < 8; ++q) h[q] = (__bf16)x.v[q];
      __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, h), reinterpret_cast<u32x4_t*>(reinterpret_cast<unsigned short*>(p) + i));
    } else {
      c3d_vst<V>(p, i, bf, x);
    }
    return;
  }
#pragma unroll
  for (int h = 0; h < V / 4; ++h)
    __builtin_nontemporal_store(f32x4{x.v[4 * h], x.v[4 * h + 1], x.v[4 * h + 2], x.v[4 * h + 3]},
                                reinterpret_cast<f32x4*>(p + i + 4 * h));
}

// Bijective XCD-aware remap: consecutive logical tiles land on the same XCD (block b runs on
// XCD b % 8 on MI355X), so neighbouring tiles that share halos / weights hit one L2.
__device__ __forceinline__ int c3d_xcd_remap(int bid, int n) {
  const int nx = 8;
  int q = n / nx, r = n % nx;
  int xcd = bid % nx, idx = bid / nx;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// wave-level sum over 64 lanes
__device__ __forceinline__ float c3d_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double c3d_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// host-side error plumbing (c_api.cpp owns the storage)
extern "C" void c3d_set_error(const char* msg);
#define C3D_CHECK_LAUNCH()                                        \
  do {                                                            \
    hipError_t e_ = hipGetLastError();                            \
    if (e_ != hipSuccess) {                                       \
      c3d_set_error(hipGetErrorString(e_));                       \
      return 1;                                                   \
    }                                                             \
  } while (0)
#define C3D_REQUIRE(cond, msg)    \
  do {                            \
    if (!(cond)) {                \
      c3d_set_error(msg);         \
      return 2;                   \
    }                             \
  } while (0)
