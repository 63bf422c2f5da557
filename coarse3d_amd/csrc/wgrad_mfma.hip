// Convolution weight gradient on fp32 MFMA for gfx950.
//
//   dW[cout][cin][t] = sum_pixels x[p + tap_t][cin] * dz[p][cout]
// (autograd of the nn.Conv2d layers of pc_processor/models/salsanext_proto.py:41-62, 82-132,
// 164-208, 318 and projector.py:18-23).  GEMM view: M = cin, N = cout, K = pixels.  A workgroup
// owns a (cin slice, cout slice) pair and a strip of pixel tiles; it keeps one 32x32 accumulator
// per (tap, cin tile, cout tile) in registers across the whole strip, so the only HBM writes
// are one partial per workgroup, folded by a second small kernel (deterministic, no atomics).
// x is transformed on load exactly like the forward conv (BatchNorm affine of the producer,
// zero padding afterwards).  Pixel tiles are software-pipelined through registers: the global
// loads of tile k+1 are in flight during the MFMA loop of tile k.
//
// The 4 waves of a workgroup are arranged WCI x WCO x WK: WCI/WCO split the cin/cout slice,
// WK splits the pixel rows of the tile (K split, reduced through LDS at the end).  Small channel
// counts (HBM-bound layers) use a 32x32 / 64x64 slice with WK = 4; the big 1x1 GEMMs of the
// projector use 128x256 slices with WK = 1.
//
// LDS images are [pixel][channel]; lane l reads channel (l&31) of pixel k + (l>>5): every
// ds_read_b32 is conflict-free and feeds one MFMA operand (A = x, B = dz).
#include <stdlib.h>
#include <type_traits>
#include "wgrad_common.h"

namespace {

// BF = false: fp32 MFMA (default parity path).  BF = true: operands rounded to bf16 (RNE) when
// read from LDS, v_mfma_f32_32x32x16_bf16 with k = 16 consecutive pixels per step (lane l:
// channel l&31, pixels 8*(l>>5)..+7), fp32 accumulation -- the first-generation "bf16" mode, kept
// behind C3D_WGRAD_TR=0 for A/B measurements; both bf16-pipe modes run wgrad_tr.hip by default.
template <int TMAX, int CI_T, int CO_T, int WCI, int WCO, int TRW, int HALO, bool BF>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(WgradArgs a) {
  constexpr int WK = 4 / (WCI * WCO);
  constexpr int CI = 32 * CI_T * WCI;  // cin slice of the workgroup
  constexpr int CO = 32 * CO_T * WCO;  // cout slice of the workgroup
  constexpr int TWh = 32 + 2 * HALO;
  constexpr int THh = TRW + 2 * HALO;
  constexpr int RPW = TRW / WK;        // tile rows per wave
  static_assert(TRW % WK == 0 && RPW >= 1, "tile rows must split across the K waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_x = smem;                   // [THh][TWh][CI]
  float* s_dz = smem + THh * TWh * CI; // [TRW*32][CO]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wci = wave % WCI, wco = (wave / WCI) % WCO, wk = wave / (WCI * WCO);

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

  // ---- register staging (prefetch) of the next pixel tile while the current one is consumed
  constexpr int X_UNITS = THh * TWh * (CI / 4);
  constexpr int X_PT = (X_UNITS + 255) / 256;
  constexpr int D_UNITS = TRW * 32 * (CO / 4);
  constexpr int D_PT = (D_UNITS + 255) / 256;
  f32x4 px_[X_PT], pd_[D_PT];
  unsigned inb = 0;
  const int xc4 = tid % (CI / 4);          // 256 % (CI/4) == 0: fixed channel quad per thread
  const int xc = ci0 + xc4 * 4;
  const bool xc_ok = xc < a.x.C;
  f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  if (a.x.scale && xc_ok) {
    psc = *reinterpret_cast<const f32x4*>(a.x.scale + xc);
    psh = *reinterpret_cast<const f32x4*>(a.x.shift + xc);
  }
  const bool aff = a.x.scale != nullptr, lr = a.x.lrelu != 0;

  auto load_tile = [&](int mt) {
    const int tx = mt % a.tiles_x;
    const int ty = (mt / a.tiles_x) % a.tiles_y;
    const int b = mt / (a.tiles_x * a.tiles_y);
    const int x0 = tx * 32, y0 = ty * TRW;
    inb = 0;
    // (one branch per tile on the storage type, not one per element: the per-element form cost
    //  spills and ~70 % of the 704x704 weight gradient's time)
    auto load_x = [&](auto bf_tag) {
      constexpr bool XBF = decltype(bf_tag)::value;
#pragma unroll
      for (int i = 0; i < X_PT; ++i) {
        const int u = tid + i * 256;
        px_[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (u < X_UNITS) {
          const int p = u / (CI / 4);
          const int px = p % TWh, py = p / TWh;
          const int gx = x0 + px - HALO, gy = y0 + py - HALO;
          if (gx >= 0 && gx < a.W && gy >= 0 && gy < a.H && xc_ok) {
            px_[i] = c3d_ld4(a.x.ptr, ((size_t)(b * a.H + gy) * a.W + gx) * a.x.cstride + a.x.coff + xc, XBF);
            inb |= 1u << i;
          }
        }
      }
    };
    auto load_dz = [&](auto bf_tag) {
      constexpr bool DBF = decltype(bf_tag)::value;
#pragma unroll
      for (int i = 0; i < D_PT; ++i) {
        const int u = tid + i * 256;
        pd_[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (u < D_UNITS) {
          const int c4 = u % (CO / 4);
          const int p = u / (CO / 4);
          const int gx = x0 + (p & 31), gy = y0 + (p >> 5);
          const int c = co0 + c4 * 4;
          if (gx < a.W && gy < a.H && c + 3 < a.dz_cstride)
            pd_[i] = c3d_ld4(a.dz, ((size_t)(b * a.H + gy) * a.W + gx) * a.dz_cstride + c, DBF);
        }
      }
    };
    if (BF && a.x.bf16) load_x(std::true_type{});
    else load_x(std::false_type{});
    if (BF && a.dz_bf16) load_dz(std::true_type{});
    else load_dz(std::false_type{});
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < X_PT; ++i) {
      const int u = tid + i * 256;
      if (u < X_UNITS) {
        f32x4 v = px_[i];
        if ((inb >> i) & 1u) {
          if (aff) v = v * psc + psh;
          if (lr) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = c3d_lrelu(v[q], a.slope);
          }
        }
        *reinterpret_cast<f32x4*>(s_x + (u / (CI / 4)) * CI + xc4 * 4) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < D_PT; ++i) {
      const int u = tid + i * 256;
      if (u < D_UNITS) *reinterpret_cast<f32x4*>(s_dz + (u / (CO / 4)) * CO + (u % (CO / 4)) * 4) = pd_[i];
    }
  };

  int tapoff[TMAX];                        // LDS offset of each tap (a.T == TMAX by construction)
#pragma unroll
  for (int t = 0; t < TMAX; ++t) tapoff[t] = (a.dy[t] * TWh + a.dx[t]) * CI;

  const int t_begin = strip * a.tiles_per_strip;
  const int t_end = min(t_begin + a.tiles_per_strip, a.ntiles);
  if (t_begin < t_end) load_tile(t_begin);
  for (int mt = t_begin; mt < t_end; ++mt) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (mt + 1 < t_end) load_tile(mt + 1);
    // ---- K loop over this wave's pixels: k-step = pixel pair (2s + half); the operand
    //      fragments of step s+1 are read from LDS while the MFMAs of step s issue
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int row = wk * RPW + rr;
      if constexpr (BF) {
        const float* xr8 = s_x + ((row + HALO) * TWh + HALO + 8 * half) * CI + wci * CI_T * 32 + l31;
        const float* dr8 = s_dz + (row * 32 + 8 * half) * CO + wco * CO_T * 32 + l31;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf16x8 bh[CO_T];
#pragma unroll
          for (int j = 0; j < CO_T; ++j) {
            f32x4 v0, v1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              v0[q] = dr8[(16 * s + q) * CO + j * 32];
              v1[q] = dr8[(16 * s + 4 + q) * CO + j * 32];
            }
            bh[j] = c3d_pack_bf16x8(v0, v1);
          }
#pragma unroll
          for (int t = 0; t < TMAX; ++t)
#pragma unroll
            for (int i = 0; i < CI_T; ++i) {
              f32x4 v0, v1;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                v0[q] = xr8[(16 * s + q) * CI + tapoff[t] + i * 32];
                v1[q] = xr8[(16 * s + 4 + q) * CI + tapoff[t] + i * 32];
              }
              const bf16x8 ac = c3d_pack_bf16x8(v0, v1);
#pragma unroll
              for (int j = 0; j < CO_T; ++j)
                acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac, bh[j], acc[t][i][j], 0, 0, 0);
            }
        }
        continue;
      }
      const float* xr = s_x + ((row + HALO) * TWh + HALO + half) * CI + wci * CI_T * 32 + l31;
      const float* dr = s_dz + (row * 32 + half) * CO + wco * CO_T * 32 + l31;
      float ac[TMAX][CI_T], bc[CO_T];
#pragma unroll
      for (int j = 0; j < CO_T; ++j) bc[j] = dr[j * 32];
#pragma unroll
      for (int t = 0; t < TMAX; ++t)
#pragma unroll
        for (int i = 0; i < CI_T; ++i) ac[t][i] = xr[tapoff[t] + i * 32];
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        float an[TMAX][CI_T], bn[CO_T];
        if (s < 15) {
#pragma unroll
          for (int j = 0; j < CO_T; ++j) bn[j] = dr[(2 * s + 2) * CO + j * 32];
#pragma unroll
          for (int t = 0; t < TMAX; ++t)
#pragma unroll
            for (int i = 0; i < CI_T; ++i) an[t][i] = xr[(2 * s + 2) * CI + tapoff[t] + i * 32];
        }
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
#pragma unroll
          for (int i = 0; i < CI_T; ++i)
#pragma unroll
            for (int j = 0; j < CO_T; ++j)
              acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[t][i], bc[j], acc[t][i][j], 0, 0, 0);
        if (s < 15) {
#pragma unroll
          for (int j = 0; j < CO_T; ++j) bc[j] = bn[j];
#pragma unroll
          for (int t = 0; t < TMAX; ++t)
#pragma unroll
            for (int i = 0; i < CI_T; ++i) ac[t][i] = an[t][i];
        }
      }
    }
  }
  // ---- fold the WK pixel-row groups through LDS, write the workgroup partial
  //      partial layout per (slice, strip): [t][cin (slice-local CI)][cout (slice-local CO)]
  const size_t slice_floats = (size_t)a.T * CI * CO;
  float* pout = a.partial + ((size_t)(sl * a.strips + strip)) * slice_floats;
  float* red = smem;  // [WK-1][WCI*WCO][CI_T*CO_T][16][64], one tap at a time
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    if (t < a.T) {
      if (WK > 1) {
        __syncthreads();
        if (wk > 0) {
#pragma unroll
          for (int i = 0; i < CI_T; ++i)
#pragma unroll
            for (int j = 0; j < CO_T; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r)
                red[(((((wk - 1) * WCO + wco) * WCI + wci) * CI_T + i) * CO_T + j) * 1024 + r * 64 + lane] =
                    acc[t][i][j][r];
        }
        __syncthreads();
      }
      if (wk == 0) {
#pragma unroll
        for (int i = 0; i < CI_T; ++i)
#pragma unroll
          for (int j = 0; j < CO_T; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float v = acc[t][i][j][r];
#pragma unroll
              for (int k = 1; k < WK; ++k)
                v += red[(((((k - 1) * WCO + wco) * WCI + wci) * CI_T + i) * CO_T + j) * 1024 + r * 64 + lane];
              const int ci = (wci * CI_T + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
              const int co = (wco * CO_T + j) * 32 + l31;
              pout[((size_t)t * CI + ci) * CO + co] = v;
            }
      }
    }
  }
}

// dw[cout][cin_off + cin][t] (+)= sum_strips partial.  Block = 256 outputs x 4 strip lanes.
// Blocks beyond `main_blocks` fold the layer's bias-gradient partials (one channel each; c3d_wgrad_desc.bias_partial).
// One fold = one c3d_wgrad_fold record; `blk` is the block's index within the fold.
__device__ __forceinline__ void wgrad_fold_body(const c3d_wgrad_fold& f, int blk, double (*red)[32]) {      // red: 1024 doubles
  const double osc = f.out_scale_dev ? (double)f.out_scale * (double)*f.out_scale_dev : (double)f.out_scale;   // powers of two: exact
  const int main_blocks = f.main_blocks;
  if (blk >= main_blocks) {
    const int c = blk - main_blocks;
    const float* p = f.bias_partial + (size_t)c * 2 * f.bias_n;
    double s = 0.0;
    for (int t = threadIdx.x; t < f.bias_n; t += 256) s += (double)p[t];
    s = c3d_wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) f.dbias[c] = (float)(red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    return;
  }
  const int T = f.T, CI = f.CI, CO = f.CO, ci_slices = f.ci_slices, strips = f.strips;
  const size_t slice_floats = (size_t)T * CI * CO;
  const int nsl = ci_slices * f.co_slices;
  const size_t total = slice_floats * nsl;
  // 256 consecutive outputs per block trip: 64 threads x float4 along the outputs (CO is a multiple of 32, so the four
  // outputs of a thread share tap and input channel), one strip lane per wave; every thread keeps four 16-byte loads in
  // flight.  (Round 4, first form: 32 outputs x 32 strip lanes -- every strip contributed 128 contiguous bytes to a block,
  // strips are 10^5 bytes apart: ~1.3 GB of partials per step at 2.4 TB/s.  A wave now reads 1 KB contiguous per strip.)
  const int o4 = threadIdx.x & 63, lanek = threadIdx.x >> 6;
  double (*red4)[64][4] = reinterpret_cast<double (*)[64][4]>(red);      // [4 strip lanes][64 quads][4]
  for (size_t base = (size_t)blk * 256; base < total; base += (size_t)main_blocks * 256) {
    const size_t e = base + (size_t)o4 * 4;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (e < total) {
      const int sl = e / slice_floats;
      const size_t r = e % slice_floats;
      const int co = (int)(r % CO) + (sl / ci_slices) * CO;
      const int ci = (int)((r / CO) % CI) + (sl % ci_slices) * CI;
      if (ci < f.Cin_src && co < f.Cout) {                 // (co .. co + 3: judged per output below)
        const float* p = f.partial + (size_t)sl * strips * slice_floats + r;
        int k = lanek;
        for (; k + 12 < strips; k += 16) {     // 4 independent 16-byte loads in flight
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + (size_t)k * slice_floats);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + (size_t)(k + 4) * slice_floats);
          const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + (size_t)(k + 8) * slice_floats);
          const f32x4 v3 = *reinterpret_cast<const f32x4*>(p + (size_t)(k + 12) * slice_floats);
          s0 += (double)v0[0] + (double)v1[0] + (double)v2[0] + (double)v3[0];
          s1 += (double)v0[1] + (double)v1[1] + (double)v2[1] + (double)v3[1];
          s2 += (double)v0[2] + (double)v1[2] + (double)v2[2] + (double)v3[2];
          s3 += (double)v0[3] + (double)v1[3] + (double)v2[3] + (double)v3[3];
        }
        for (; k < strips; k += 4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(p + (size_t)k * slice_floats);
          s0 += (double)v[0];
          s1 += (double)v[1];
          s2 += (double)v[2];
          s3 += (double)v[3];
        }
      }
    }
    red4[lanek][o4][0] = s0;
    red4[lanek][o4][1] = s1;
    red4[lanek][o4][2] = s2;
    red4[lanek][o4][3] = s3;
    __syncthreads();
    {
      const int q = threadIdx.x >> 2, j = threadIdx.x & 3;      // output 4 * q + j of the trip: every thread finishes one
      const size_t e2 = base + (size_t)q * 4;
      if (e2 < total) {
        const int sl = e2 / slice_floats;
        const size_t r = e2 % slice_floats;
        const int co2 = (int)(r % CO) + (sl / ci_slices) * CO + j;
        const int ci2 = (int)((r / CO) % CI) + (sl % ci_slices) * CI;
        const int t2 = r / ((size_t)CO * CI);
        if (ci2 < f.Cin_src && co2 < f.Cout) {
          double v = (red4[0][q][j] + red4[1][q][j]) + (red4[2][q][j] + red4[3][q][j]);
          v *= osc;
          float* d = f.dw + ((size_t)co2 * f.Cin_total + f.cin_off + ci2) * T + t2;
          *d = f.accumulate ? (*d + (float)v) : (float)v;
        }
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(c3d_wgrad_fold f) {
  __shared__ double red[32][32];
  wgrad_fold_body(f, blockIdx.x, red);
}

// Up to FOLD_BATCH folds in ONE launch (the records travel as kernel arguments: no table in device memory, nothing to
// upload, capturable as it is).  Round 3 ran 74 fold launches of ~9 us per training step, one behind each weight-gradient
// launch; nothing reads a weight gradient before the optimiser does, so the backward pass now queues them and folds
// them all at its end (coarse3d_amd.ops.WgradFolds).
constexpr int FOLD_BATCH = 32;
struct FoldBatch {
  c3d_wgrad_fold f[FOLD_BATCH];
  int n;
};
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(FoldBatch b) {
  __shared__ double red[32][32];
  int e = 0;
  const int blk = blockIdx.x;
  while (e + 1 < b.n && blk >= b.f[e + 1].block0) ++e;       // block -> fold (b.n <= 32: a short scalar walk)
  wgrad_fold_body(b.f[e], blk - b.f[e].block0, red);
}

// bf16 planes of the operands (wgrad_tr.hip) or 0 = the fp32-MFMA kernel of this file.
int planes_for(const c3d_wgrad_desc* d) {
  if (d->mfma_bf16 == 0) return 0;
  if (d->mfma_bf16 == 4) return 2;        // EXPERIMENT: two fp16 planes
  return d->mfma_bf16 == 2 ? 3 : 1;
}

void plan(const c3d_wgrad_desc* d, WgradArgs& a, WgCfg& c) {
  int halo = 0;
  for (int t = 0; t < d->ntaps && t < 9; ++t) {
    const int m = abs(d->tap_dy[t]) > abs(d->tap_dx[t]) ? abs(d->tap_dy[t]) : abs(d->tap_dx[t]);
    if (m > halo) halo = m;
  }
  c = c3d_wgrad_cfg(d->ntaps, d->x.C, d->Cout, planes_for(d) == 2 ? 3 : planes_for(d), halo);     // (two fp16 planes: the tiles of three)
  // BatchNorm backward on load over 256-cout slices keeps dy AND the stored output of 8 units per thread in flight, twice:
  // wgrad_tr_kernel<3, 1, 2, 4, 2, 2, 1, 0, true> spilled 50 registers (184 scratch instructions, 108 full drains of the loads
  // in flight; 51-76 TF where the unfused instance makes 149-176).  The fused launch takes the 128 x 128 slice instead
  // (x is staged twice, it is the narrow side; no spill).  Round 5.
  if (d->fuse_dy && planes_for(d) == 3 && c.id == 0 && !(d->variant & 4)) c = WgCfg{1, 128, 128, 1};
  a.tiles_x = (d->W + 31) / 32;
  a.tiles_y = (d->H + c.TRW - 1) / c.TRW;
  a.ntiles = d->B * a.tiles_x * a.tiles_y;
  a.ci_slices = (d->x.C + c.CI - 1) / c.CI;
  a.co_slices = (d->Cout + c.CO - 1) / c.CO;
  const int nsl = a.ci_slices * a.co_slices;
  // one resident round: 256 CUs x 2 workgroups (x 1 eight-wave workgroup for the bf16-plane
  // kernels); never exceed it (a 2nd, nearly empty round would double the kernel time)
  int strips = (planes_for(d) ? 256 : 512) / nsl;
  if (strips > a.ntiles) strips = a.ntiles;
  if (strips < 1) strips = 1;
  a.tiles_per_strip = (a.ntiles + strips - 1) / strips;
  a.strips = (a.ntiles + a.tiles_per_strip - 1) / a.tiles_per_strip;
  a.npw = c3d_wgrad_producer_waves(planes_for(d), c.id, d->variant, d->ntaps);
  // nine taps with the BatchNorm backward on load: sixteen waves with the producer waves split by tensor (round 6, ROLES in
  // wgrad_tr.hip); layers with a pre-activation affine, three / six taps and variant & 256 keep the four + four wave form
  if (c.id >= 6 && d->fuse_dy && (a.npw != 8 || d->fuse_pre_scale || (d->variant & 256))) a.npw = 4;
  // unfused 1x1 launches over 128 x 256 slices (the projector's 704-wide layers): cout tiles split across the consumer halves (SPL1)
  if (c.id == 0 && planes_for(d) == 3 && !d->fuse_dy && !(d->variant & (256 | 128))) a.npw = 8;
  // the four-tap launch over 64 x 64 slices alike (taps split 2 + 2; wgrad_tr.hip, SPL4), fused or not
  if (c.id == 8 && planes_for(d) == 3 && d->ntaps == 4 && halo <= 1 && !d->fuse_pre_scale && !(d->variant & (256 | 128))) a.npw = 8;
}

// threads the dz units of a fused launch are dealt to (= entries per strip and channel of fuse_sum, times CO / 4)
int fused_dz_threads(const c3d_wgrad_desc* d, const WgradArgs& a, const WgCfg& c) {
  return (c.id >= 6 && d->fuse_dy && a.npw == 8) ? 256 : 64 * a.npw;      // (id 8 included; unfused launches have no sums)
}

template <int TMAX, int CI_T, int CO_T, int WCI, int WCO, int TRW, int HALO, bool BF>
int launch_wg(const WgradArgs& a, hipStream_t st) {
  constexpr int WK = 4 / (WCI * WCO);
  constexpr int CI = 32 * CI_T * WCI, CO = 32 * CO_T * WCO;
  size_t lds = ((size_t)(TRW + 2 * HALO) * (32 + 2 * HALO) * CI + (size_t)TRW * 32 * CO) * sizeof(float);
  const size_t red = (size_t)(WK - 1) * WCI * WCO * CI_T * CO_T * 1024 * sizeof(float);
  if (red > lds) lds = red;
  c3d_opt_in_lds<&wgrad_mfma_kernel<TMAX, CI_T, CO_T, WCI, WCO, TRW, HALO, BF>>();
  dim3 grid(a.strips * a.ci_slices * a.co_slices);
  hipLaunchKernelGGL((wgrad_mfma_kernel<TMAX, CI_T, CO_T, WCI, WCO, TRW, HALO, BF>), grid, dim3(256), lds, st, a);
  C3D_CHECK_LAUNCH();
  return 0;
}

template <bool BF>
int launch_id(int id, int halo, const WgradArgs& a, hipStream_t st) {
  switch (id) {
    //                     TMAX CI_T CO_T WCI WCO TRW HALO
    case 0: return launch_wg<1, 2, 4, 2, 2, 1, 0, BF>(a, st);
    case 1: return launch_wg<1, 2, 2, 2, 2, 1, 0, BF>(a, st);
    case 2: return launch_wg<1, 2, 2, 1, 1, 4, 0, BF>(a, st);
    case 3: return launch_wg<1, 1, 1, 1, 1, 4, 0, BF>(a, st);
    case 4: return halo <= 1 ? launch_wg<4, 1, 2, 1, 1, 4, 1, BF>(a, st) : launch_wg<4, 1, 2, 1, 1, 4, 2, BF>(a, st);
    case 5: return halo <= 1 ? launch_wg<4, 1, 1, 1, 1, 4, 1, BF>(a, st) : launch_wg<4, 1, 1, 1, 1, 4, 2, BF>(a, st);
    case 6: return halo <= 1 ? launch_wg<9, 1, 1, 1, 2, 2, 1, BF>(a, st) : launch_wg<9, 1, 1, 1, 2, 2, 2, BF>(a, st);
    default: return halo <= 1 ? launch_wg<9, 1, 1, 1, 1, 4, 1, BF>(a, st) : launch_wg<9, 1, 1, 1, 1, 4, 2, BF>(a, st);
  }
}

}  // namespace

// c3d_wgrad_desc.fuse_*: the shapes / engines that have the fused form (wgrad_tr.hip, three planes, fp32 tensors)
static bool fuse_supported(const c3d_wgrad_desc* d) {
  return planes_for(d) == 3 && !d->dz_bf16 && !d->x.bf16 && d->Cout % 4 == 0 && d->dz_cstride % 4 == 0;
}

extern "C" int c3d_wgrad_fused_sum_n(const c3d_wgrad_desc* d) {
  if (!d || !fuse_supported(d)) return 0;
  WgradArgs a;
  WgCfg c;
  plan(d, a, c);
  return a.strips * (fused_dz_threads(d, a, c) / (c.CO / 4));
}

extern "C" int64_t c3d_wgrad_partial_floats(const c3d_wgrad_desc* d) {
  WgradArgs a;
  WgCfg c;
  plan(d, a, c);
  return (int64_t)a.strips * a.ci_slices * a.co_slices * d->ntaps * c.CI * c.CO;
}

extern "C" int c3d_conv_wgrad(const c3d_wgrad_desc* d, c3d_stream stream) {
  C3D_REQUIRE(d != nullptr && d->x.ptr && d->dz && d->dw && d->partial, "wgrad: null pointer");
  // 3 and 6 taps (round 5: RangeNet's strided / transposed convs over column-pair views, coarse3d_amd/rangenet.py) run in the
  // four- / nine-tap instances with the surplus taps at offset zero: their products are computed and never written
  C3D_REQUIRE(d->ntaps == 1 || d->ntaps == 3 || d->ntaps == 4 || d->ntaps == 6 || d->ntaps == 9, "wgrad: ntaps must be 1, 3, 4, 6 or 9");
  C3D_REQUIRE(d->x.C % 4 == 0, "wgrad: source channels must be a multiple of 4");
  C3D_REQUIRE(d->dz_cstride % 4 == 0 && d->x.cstride % 4 == 0 && d->x.coff % 4 == 0, "wgrad: strides must be multiples of 4");
  WgradArgs a;
  a.x = d->x; a.dz = d->dz; a.dz_cstride = d->dz_cstride; a.dz_bf16 = d->dz_bf16;
  C3D_REQUIRE((!d->dz_bf16 && !d->x.bf16) || d->mfma_bf16 == 1, "wgrad: bf16 activation storage needs mfma_bf16 == 1");
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cout = d->Cout; a.T = d->ntaps;
  int halo = 0;
  for (int t = 0; t < 9; ++t) a.dy[t] = a.dx[t] = 0;
  for (int t = 0; t < d->ntaps; ++t) {
    a.dy[t] = d->tap_dy[t];
    a.dx[t] = d->tap_dx[t];
    int m = abs(d->tap_dy[t]) > abs(d->tap_dx[t]) ? abs(d->tap_dy[t]) : abs(d->tap_dx[t]);
    if (m > halo) halo = m;
  }
  C3D_REQUIRE(halo <= 2, "wgrad: tap offsets beyond +-2 are not supported");
  C3D_REQUIRE(d->ntaps != 1 || halo == 0, "wgrad: a single tap must have zero offset");
  a.dz_scale = d->dz_scale;
  C3D_REQUIRE(d->mfma_bf16 != 4 || (d->dz_scale && d->out_scale_dev && !d->dz_bf16 && !d->x.bf16 && d->ntaps > 1),
              "wgrad: mfma_bf16 == 4 (f16x2 experiment) needs dz_scale, out_scale_dev, fp32 tensors and more than one tap");
  a.partial = d->partial;
  a.slope = c3d_slope_or_default(d->lrelu_slope);
  a.variant = d->variant;
  C3D_REQUIRE(a.slope <= 1.f, "wgrad: LeakyReLU slopes above 1 are not supported (the kernels evaluate max(v, slope * v))");
  WgCfg c;
  plan(d, a, c);
  if (d->fuse_dy) {
    C3D_REQUIRE(fuse_supported(d), "wgrad: fuse_dy needs the bf16x3 engine (mfma_bf16 == 2), fp32 tensors and Cout % 4 == 0");
    C3D_REQUIRE(d->fuse_act && d->fuse_sum && d->dz != d->fuse_dy, "wgrad: fuse_dy needs fuse_act, fuse_sum and a dz buffer of its own");
    C3D_REQUIRE((d->fuse_k1 == nullptr) == (d->fuse_k2 == nullptr) && (d->fuse_k1 == nullptr) == (d->fuse_k3 == nullptr),
                "wgrad: fuse_k1 / k2 / k3 come together");
    a.f_dy = d->fuse_dy; a.f_act = d->fuse_act; a.f_k1 = d->fuse_k1; a.f_k2 = d->fuse_k2; a.f_k3 = d->fuse_k3;
    a.f_sum = d->fuse_sum;
    C3D_REQUIRE((d->fuse_pre_scale == nullptr) == (d->fuse_pre_shift == nullptr) && (!d->fuse_pre_scale || d->fuse_k1),
                "wgrad: fuse_pre_scale / fuse_pre_shift come together and need the BatchNorm coefficients");
    a.f_ps = d->fuse_pre_scale; a.f_psh = d->fuse_pre_shift;
    a.f_sum_n = a.strips * (fused_dz_threads(d, a, c) / (c.CO / 4));
  }
  hipStream_t st = (hipStream_t)stream;
  const int planes = planes_for(d);
  const int rc = planes ? c3d_wgrad_launch_tr(planes, c.id, halo, a, st)
                        : (d->mfma_bf16 == 1 ? launch_id<true>(c.id, halo, a, st) : launch_id<false>(c.id, halo, a, st));
  if (rc) return rc;
  const size_t total = (size_t)d->ntaps * c.CI * c.CO * a.ci_slices * a.co_slices;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  const bool bias = d->bias_partial != nullptr;
  C3D_REQUIRE(!bias || (d->dbias != nullptr && d->bias_n > 0), "wgrad: bias_partial needs dbias and bias_n");
  c3d_wgrad_fold f;
  f.partial = a.partial; f.dw = d->dw;
  f.strips = a.strips; f.T = d->ntaps; f.CI = c.CI; f.CO = c.CO; f.ci_slices = a.ci_slices; f.co_slices = a.co_slices;
  f.Cin_src = d->x.C; f.Cout = d->Cout; f.Cin_total = d->Cin_total; f.cin_off = d->cin_off; f.accumulate = d->accumulate;
  f.main_blocks = blocks; f.nblocks = blocks + (bias ? d->Cout : 0); f.block0 = 0;
  f.bias_partial = d->bias_partial; f.dbias = d->dbias; f.bias_n = d->bias_n;
  f.out_scale = d->mfma_bf16 == 4 ? 1.f / 64.f : 1.f;
  f.out_scale_dev = d->mfma_bf16 == 4 ? d->out_scale_dev : nullptr;
  if (d->fold_out) {          // deferred: the caller folds a batch of them with c3d_wgrad_fold_batch
    *d->fold_out = f;
    return 0;
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(f.nblocks), dim3(256), 0, st, f);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_wgrad_fold_batch(const c3d_wgrad_fold* folds, int n, c3d_stream stream) {
  C3D_REQUIRE(folds != nullptr && n >= 0, "wgrad_fold_batch: null table");
  for (int i0 = 0; i0 < n; i0 += FOLD_BATCH) {
    FoldBatch b;
    b.n = n - i0 < FOLD_BATCH ? n - i0 : FOLD_BATCH;
    int blocks = 0;
    for (int i = 0; i < b.n; ++i) {
      b.f[i] = folds[i0 + i];
      C3D_REQUIRE(b.f[i].partial && b.f[i].dw && b.f[i].nblocks > 0, "wgrad_fold_batch: a record was not filled by c3d_conv_wgrad");
      b.f[i].block0 = blocks;
      blocks += b.f[i].nblocks;
    }
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b);
    C3D_CHECK_LAUNCH();
  }
  return 0;
}
