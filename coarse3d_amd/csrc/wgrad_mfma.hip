// Convolution weight gradient on fp32 MFMA for gfx950.
//
//   dW[cout][cin][t] = sum_pixels x[p + tap_t][cin] * dz[p][cout]
// (autograd of the nn.Conv2d layers of pc_processor/models/salsanext_proto.py:41-62, 82-132,
// 164-208, 318 and projector.py:18-23).  GEMM view: M = cin, N = cout, K = pixels.  A workgroup
// owns a (cin slice, cout slice) pair and a strip of pixel tiles; it keeps one 32x32 accumulator
// per (tap, cin tile, cout tile) in registers across the whole strip, so the only HBM writes
// are one partial per workgroup, reduced by a second tiny kernel (deterministic, no atomics).
// x is transformed on load exactly like the forward conv (BatchNorm affine of the producer,
// zero padding afterwards).
//
// LDS images are [pixel][channel]; lane l reads channel (l&31) of pixel k + (l>>5): every
// ds_read_b32 is conflict-free and feeds one MFMA operand (A = x, B = dz).
#include "common.h"
#include "../../include/coarse3d_hip.h"

namespace {

struct WgradArgs {
  c3d_src x;
  const float* dz;
  int dz_cstride;
  int B, H, W, Cout;
  int T;
  int dy[C3D_MAX_TAPS];
  int dx[C3D_MAX_TAPS];
  float* partial;
  int tiles_x, tiles_y, ntiles, strips, tiles_per_strip;
  int ci_slices, co_slices;
};

// TMAX taps, wave tile CI_T x CO_T (32x32 each), TRW tile rows (split between wave pairs)
template <int TMAX, int CI_T, int CO_T, int TRW, int HALO>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(WgradArgs a) {
  constexpr int CI = 32 * CI_T;        // cin slice of the workgroup
  constexpr int CO = 64 * CO_T;        // cout slice of the workgroup (2 waves along cout)
  constexpr int TWh = 32 + 2 * HALO;
  constexpr int THh = TRW + 2 * HALO;
  constexpr int RPW = TRW / 2;         // rows per wave
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_x = smem;                   // [THh][TWh][CI]
  float* s_dz = smem + THh * TWh * CI; // [TRW*32][CO]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wc = wave & 1, wr = wave >> 1;

  const int nsl = a.ci_slices * a.co_slices;
  const int logical = c3d_xcd_remap(blockIdx.x, a.strips * nsl);
  const int strip = logical / nsl;
  const int sl = logical % nsl;
  const int ci0 = (sl % a.ci_slices) * CI;
  const int co0 = (sl / a.ci_slices) * CO;

  f32x16 acc[TMAX][CI_T][CO_T];
#pragma unroll
  for (int t = 0; t < TMAX; ++t)
#pragma unroll
    for (int i = 0; i < CI_T; ++i)
#pragma unroll
      for (int j = 0; j < CO_T; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;

  const int t_begin = strip * a.tiles_per_strip;
  const int t_end = min(t_begin + a.tiles_per_strip, a.ntiles);
  for (int mt = t_begin; mt < t_end; ++mt) {
    const int tx = mt % a.tiles_x;
    const int ty = (mt / a.tiles_x) % a.tiles_y;
    const int b = mt / (a.tiles_x * a.tiles_y);
    const int x0 = tx * 32, y0 = ty * TRW;
    __syncthreads();
    // ---- stage x tile (+halo), transformed
    for (int u = tid; u < THh * TWh * (CI / 4); u += 256) {
      const int c4 = u % (CI / 4);
      const int p = u / (CI / 4);
      const int px = p % TWh, py = p / TWh;
      const int gx = x0 + px - HALO, gy = y0 + py - HALO;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const int c = ci0 + c4 * 4;
      if (gx >= 0 && gx < a.W && gy >= 0 && gy < a.H && c < a.x.C) {
        const size_t off = ((size_t)(b * a.H + gy) * a.W + gx) * a.x.cstride + a.x.coff + c;
        v = *reinterpret_cast<const f32x4*>(a.x.ptr + off);
        if (a.x.scale) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(a.x.scale + c);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(a.x.shift + c);
          v = v * sc + sh;
        }
        if (a.x.lrelu) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = c3d_lrelu(v[q]);
        }
      }
      *reinterpret_cast<f32x4*>(s_x + p * CI + c4 * 4) = v;
    }
    // ---- stage dz tile
    for (int u = tid; u < TRW * 32 * (CO / 4); u += 256) {
      const int c4 = u % (CO / 4);
      const int p = u / (CO / 4);
      const int gx = x0 + (p & 31), gy = y0 + (p >> 5);
      const int c = co0 + c4 * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gx < a.W && gy < a.H && c + 3 < a.dz_cstride)
        v = *reinterpret_cast<const f32x4*>(a.dz + ((size_t)(b * a.H + gy) * a.W + gx) * a.dz_cstride + c);
      *reinterpret_cast<f32x4*>(s_dz + p * CO + c4 * 4) = v;
    }
    __syncthreads();
    // ---- K loop over this wave's pixels: k-step = pixel pair (2s + half)
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int row = wr * RPW + rr;
#pragma unroll 4
      for (int s = 0; s < 16; ++s) {
        const int col = 2 * s + half;
        float bv[CO_T];
#pragma unroll
        for (int j = 0; j < CO_T; ++j) bv[j] = s_dz[(row * 32 + col) * CO + (wc * CO_T + j) * 32 + l31];
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
          if (t < a.T) {
            const float* xp = s_x + ((row + HALO + a.dy[t]) * TWh + (col + HALO + a.dx[t])) * CI + l31;
#pragma unroll
            for (int i = 0; i < CI_T; ++i) {
              const float av = xp[i * 32];
#pragma unroll
              for (int j = 0; j < CO_T; ++j)
                acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[j], acc[t][i][j], 0, 0, 0);
            }
          }
        }
      }
    }
  }
  // ---- reduce the two row-halves through LDS, write the workgroup partial
  //      partial layout: [strip][t][cin (slice-local CI)][cout (slice-local CO)] per slice
  __syncthreads();
  float* red = smem;  // [2 (wc)][CI_T*CO_T tiles][16][64]  of one tap at a time
  const size_t slice_floats = (size_t)a.T * CI * CO;
  float* pout = a.partial + ((size_t)(sl * a.strips + strip)) * slice_floats;
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    if (t < a.T) {
      if (wr == 1) {
#pragma unroll
        for (int i = 0; i < CI_T; ++i)
#pragma unroll
          for (int j = 0; j < CO_T; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              red[(((wc * CI_T + i) * CO_T + j) * 16 + r) * 64 + lane] = acc[t][i][j][r];
      }
      __syncthreads();
      if (wr == 0) {
#pragma unroll
        for (int i = 0; i < CI_T; ++i)
#pragma unroll
          for (int j = 0; j < CO_T; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float v = acc[t][i][j][r] + red[(((wc * CI_T + i) * CO_T + j) * 16 + r) * 64 + lane];
              const int ci = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
              const int co = (wc * CO_T + j) * 32 + l31;
              pout[((size_t)t * CI + ci) * CO + co] = v;
            }
      }
      __syncthreads();
    }
  }
}

// dw[cout][cin_off + cin][t] (+)= sum_strips partial
__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int strips, int T,
                                    int CI, int CO, int ci_slices, int co_slices, int Cin_src, int Cout,
                                    int Cin_total, int cin_off, int accumulate) {
  const size_t total = (size_t)Cout * Cin_src * T;
  const size_t slice_floats = (size_t)T * CI * CO;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    // thread index ordered (t, cin, cout) with cout fastest: coalesced partial reads
    const int co = i % Cout;
    size_t r = i / Cout;
    const int ci = r % Cin_src;
    const int t = r / Cin_src;
    const int sl = (co / CO) * ci_slices + (ci / CI);
    const float* p = partial + (size_t)sl * strips * slice_floats + ((size_t)t * CI + (ci % CI)) * CO + (co % CO);
    double s = 0.0;
    for (int k = 0; k < strips; ++k) s += (double)p[(size_t)k * slice_floats];
    float* d = dw + ((size_t)co * Cin_total + cin_off + ci) * T + t;
    *d = accumulate ? (*d + (float)s) : (float)s;
  }
}

struct WgCfg {
  int CI, CO, TRW;
};
WgCfg cfg_for(int T) {
  if (T == 1) return {64, 256, 2};
  if (T <= 4) return {32, 128, 4};
  return {32, 64, 4};
}

void plan(const c3d_wgrad_desc* d, WgradArgs& a) {
  const WgCfg c = cfg_for(d->ntaps);
  a.tiles_x = (d->W + 31) / 32;
  a.tiles_y = (d->H + c.TRW - 1) / c.TRW;
  a.ntiles = d->B * a.tiles_x * a.tiles_y;
  a.ci_slices = (d->x.C + c.CI - 1) / c.CI;
  a.co_slices = (d->Cout + c.CO - 1) / c.CO;
  const int nsl = a.ci_slices * a.co_slices;
  int strips = (1024 + nsl - 1) / nsl;
  if (strips > a.ntiles) strips = a.ntiles;
  if (strips < 1) strips = 1;
  a.tiles_per_strip = (a.ntiles + strips - 1) / strips;
  a.strips = (a.ntiles + a.tiles_per_strip - 1) / a.tiles_per_strip;
}

template <int TMAX, int CI_T, int CO_T, int TRW, int HALO>
int launch_wg(const WgradArgs& a, hipStream_t st) {
  constexpr int CI = 32 * CI_T, CO = 64 * CO_T;
  size_t lds = ((size_t)(TRW + 2 * HALO) * (32 + 2 * HALO) * CI + (size_t)TRW * 32 * CO) * sizeof(float);
  const size_t red = (size_t)2 * CI_T * CO_T * 16 * 64 * sizeof(float);
  if (red > lds) lds = red;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_mfma_kernel<TMAX, CI_T, CO_T, TRW, HALO>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  dim3 grid(a.strips * a.ci_slices * a.co_slices);
  hipLaunchKernelGGL((wgrad_mfma_kernel<TMAX, CI_T, CO_T, TRW, HALO>), grid, dim3(256), lds, st, a);
  C3D_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" int64_t c3d_wgrad_partial_floats(const c3d_wgrad_desc* d) {
  WgradArgs a;
  plan(d, a);
  const WgCfg c = cfg_for(d->ntaps);
  return (int64_t)a.strips * a.ci_slices * a.co_slices * d->ntaps * c.CI * c.CO;
}

extern "C" int c3d_conv_wgrad(const c3d_wgrad_desc* d, c3d_stream stream) {
  C3D_REQUIRE(d->ntaps == 1 || d->ntaps == 4 || d->ntaps == 9, "wgrad: ntaps must be 1, 4 or 9");
  C3D_REQUIRE(d->x.C % 4 == 0, "wgrad: source channels must be a multiple of 4");
  C3D_REQUIRE(d->dz_cstride % 4 == 0 && d->x.cstride % 4 == 0 && d->x.coff % 4 == 0, "wgrad: strides must be multiples of 4");
  WgradArgs a;
  a.x = d->x; a.dz = d->dz; a.dz_cstride = d->dz_cstride;
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cout = d->Cout; a.T = d->ntaps;
  int halo = 0;
  for (int t = 0; t < d->ntaps; ++t) {
    a.dy[t] = d->tap_dy[t];
    a.dx[t] = d->tap_dx[t];
    int m = abs(d->tap_dy[t]) > abs(d->tap_dx[t]) ? abs(d->tap_dy[t]) : abs(d->tap_dx[t]);
    if (m > halo) halo = m;
  }
  C3D_REQUIRE(halo <= 2, "wgrad: tap offsets beyond +-2 are not supported");
  a.partial = d->partial;
  plan(d, a);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (d->ntaps == 1) {
    rc = launch_wg<1, 2, 4, 2, 0>(a, st);
  } else if (d->ntaps == 4) {
    rc = launch_wg<4, 1, 2, 4, 1>(a, st);
  } else {
    rc = (halo <= 1) ? launch_wg<9, 1, 1, 4, 1>(a, st) : launch_wg<9, 1, 1, 4, 2>(a, st);
  }
  if (rc) return rc;
  const WgCfg c = cfg_for(d->ntaps);
  const size_t total = (size_t)d->Cout * d->x.C * d->ntaps;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, a.partial, d->dw, a.strips, d->ntaps,
                     c.CI, c.CO, a.ci_slices, a.co_slices, d->x.C, d->Cout, d->Cin_total, d->cin_off,
                     d->accumulate);
  C3D_CHECK_LAUNCH();
  return 0;
}
