// Shared by the weight-gradient kernels (wgrad_mfma.hip: fp32 MFMA; wgrad_tr.hip: bf16 planes read
// through ds_read_b64_tr_b16): launch arguments, tile configurations and the partial-sum layout.
#pragma once
#include <cstdlib>
#include "common.h"
#include "../../include/coarse3d_hip.h"

struct WgradArgs {
  c3d_src x;
  const float* dz;
  int dz_cstride;
  int dz_bf16;
  int B, H, W, Cout;
  int T;
  int dy[C3D_MAX_TAPS];
  int dx[C3D_MAX_TAPS];
  float* partial;
  int tiles_x, tiles_y, ntiles, strips, tiles_per_strip;
  const float* dz_scale = nullptr;   // f16x2 experiment: per-cout scale of dz on load
  // BatchNorm / LeakyReLU backward on load (c3d_wgrad_desc.fuse_*): dz is then written, not read
  const float* f_dy = nullptr;
  const float* f_act = nullptr;
  const float* f_k1 = nullptr;
  const float* f_k2 = nullptr;
  const float* f_k3 = nullptr;
  float* f_sum = nullptr;
  const float* f_ps = nullptr;       // pre-activation affine of a conv -> BatchNorm -> LeakyReLU layer (mode 1), or NULL
  const float* f_psh = nullptr;
  int f_sum_n = 0;
  int ci_slices, co_slices;
  float slope;
  int variant = 0;                   // c3d_wgrad_desc.variant
  int npw = 4;                       // producer waves of the wgrad_tr workgroup (c3d_wgrad_producer_waves)
};

// A workgroup owns a (CI cin, CO cout) slice and TRW x 32 pixel tiles.
// id: 0..3 = 1x1 {128x256, 128x128, 64x64, 32x32}; 4,5 = 2x2 {CO64, CO32}; 6,7 = 3x3 {CO64, CO32}
struct WgCfg {
  int id, CI, CO, TRW;
};

// planes = 0: fp32 MFMA kernels; 1 / 3: bf16-plane kernels (smaller pixel tiles: three planes of a
// tile must leave room for two workgroups per CU)
inline WgCfg c3d_wgrad_cfg(int T, int Cin, int Cout, int planes, int halo = 1) {
  const bool tr = planes != 0;
  if (T == 1) {
    if (Cin >= 96 && Cout >= 192) return {0, 128, 256, 1};
    if (Cin >= 96 && Cout >= 96) return {1, 128, 128, 1};
    if (Cout > 32) return {2, 64, 64, tr ? 2 : 4};
    return {3, 32, 32, 4};
  }
  // (one plane: a tile costs a third of the LDS and next to no matrix time; the kernel is then bound by
  //  the per-tile synchronisation, so it takes 4-row tiles everywhere)
  // (three planes, >= 64 input channels: two consumer waves split a 64-channel cin slice -- twice the MFMAs per staged
  //  dz element of the 32-channel slice, which left the 2x2 weight gradients at 125-133 TF against 195-210 for the 3x3)
  //  (halo <= 1: with a two-pixel halo the two tile buffers of that slice exceed the 160 KB of LDS)
  if (T <= 4 && planes == 3 && Cout > 32 && Cin % 64 == 0 && halo <= 1) return WgCfg{8, 64, 64, 2};
  if (T <= 4) return Cout > 32 ? WgCfg{4, 32, 64, planes == 3 ? 2 : 4} : WgCfg{5, 32, 32, 4};
  return Cout > 32 ? WgCfg{6, 32, 64, planes == 1 ? 4 : 2} : WgCfg{7, 32, 32, 4};
}

// Producer waves of the wgrad_tr workgroup: eight (two per SIMD, 768 threads) for the three-plane 1x1 instances with small
// accumulators (ids 1-3), four elsewhere.  c3d_wgrad_desc.variant & 128: four everywhere (the bit-identity test, A/B runs).
inline int c3d_wgrad_producer_waves(int planes, int id, int variant, int ntaps = 1) {
  if ((planes != 3 && planes != 1) || (variant & 128)) return 4;
  if (id >= 1 && id <= 3) return 8;      // (one plane: the bf16 engine's 1x1 weight gradients are bound by requests in flight alike)
  // nine taps: with the taps split across eight consumer waves (wgrad_tr.hip, NCW); one plane: the 32-cout instance only
  return (id >= 6 && ntaps == 9 && (planes == 3 || id == 7)) ? 8 : 4;
}

// wgrad_tr.hip
int c3d_wgrad_launch_tr(int planes, int id, int halo, const WgradArgs& a, hipStream_t st);
