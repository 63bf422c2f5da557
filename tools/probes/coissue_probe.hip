// Do VALU instructions of one wave overlap with MFMAs of another wave on the same SIMD (gfx950)?
// 512-thread blocks, one per CU: waves 0-3 issue MFMAs, waves 4-7 issue VALU work (fma chain, or the
// bf16 3-plane split mix).  mode bit 0: MFMA waves active, bit 1: VALU waves active.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512, 1) void probe(float* out, int mode, int nm, int nv, int kind) {
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x) >> 6;
  if (wave < 4) {
    if (!(mode & 1)) return;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (__bf16)(float)(threadIdx.x + q); b[q] = (__bf16)(float)(q * 3 + 1); }
    for (int it = 0; it < nm; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    if (!(mode & 2)) return;
    float v[8];
    for (int q = 0; q < 8; ++q) v[q] = (float)threadIdx.x * 0.001f + q;
    if (kind == 0) {
      for (int it = 0; it < nv; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = __builtin_fmaf(v[q], 1.0001f, 0.5f);
      }
    } else {   // split mix: cvt_pk, shift, sub
      for (int it = 0; it < nv; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const __bf16 h = (__bf16)v[q];
          v[q] = (v[q] - (float)h) * 256.f + 1.0f;
        }
      }
    }
    float s = 0.f;
    for (int q = 0; q < 8; ++q) s += v[q];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 512 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int nm = 20000;   // x4 MFMAs per wave = 80000 x 32 cycles = 2.56 M cycles
  for (int kind = 0; kind < 2; ++kind)
    for (int nv : {20000, 40000, 80000}) {
      for (int mode = 1; mode <= 3; ++mode) {
        probe<<<256, 512>>>(d, mode, nm, nv, kind);
        (void)hipEventRecord(e0);
        probe<<<256, 512>>>(d, mode, nm, nv, kind);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("kind %d nv %6d mode %d: %.3f ms\n", kind, nv, mode, ms);
      }
    }
  return 0;
}
