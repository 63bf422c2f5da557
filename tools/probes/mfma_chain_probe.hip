// Issue rate of v_mfma_f32_32x32x16_bf16 on one wave per SIMD when consecutive MFMAs accumulate into
// NACC different accumulators (NACC = 1: every MFMA depends on the previous one).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256, 1) void probe(float* out, int n) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int q = 0; q < 8; ++q) { a[q] = (__bf16)(float)(threadIdx.x + q); b[q] = (__bf16)(float)(q * 3 + 1); }
  for (int it = 0; it < n; it += NACC) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(float* d, int n) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  probe<NACC><<<256, 256>>>(d, n);
  (void)hipEventRecord(e0);
  probe<NACC><<<256, 256>>>(d, n);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("NACC %d: %.3f ms for %d MFMAs per wave -> %.1f ns per MFMA\n", NACC, ms, n, ms * 1e6 / n);
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 256 * 4);
  const int n = 96000;
  run<1>(d, n); run<2>(d, n); run<4>(d, n); run<8>(d, n);
  return 0;
}
