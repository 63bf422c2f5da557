// 3x3 / 2x2 convolutions in fp32-class arithmetic on the bf16 matrix pipe ("bf16x3", second
// generation of conv_bfp.hip's NP = 3 mode; c3d_conv_desc.mfma_bf16 == 2, 8-row tiles, > 1 tap).
//
// Arithmetic: every fp32 operand is split exactly into three bf16 planes (x = h + m + l) and
// eight of the nine plane products are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 -- only
// l*l (< 2^-32 |a||b|) is dropped, so every product is exact far below fp32 resolution and the
// result carries fp32 accumulation rounding only (the whole parity suite passes in this
// mode: C3D_MATRIX=bf16x3).  Eight bf16 MFMAs cost half the issue time of the fp32 MFMAs they
// replace; the first-generation kernel could not cash that in because its LDS image (three
// planes of the input tile AND of all nine weight taps, padded rows: 130-145 KB) left ONE
// workgroup per CU -- staging and matrix phases never overlapped.  This kernel:
//   * weights arrive pre-split: c3d_pack_weights(mode | 2) appends the three bf16 planes to the
//     fp32 pack once per step, the kernel copies them global -> LDS with no VALU work;
//   * the weight slab is staged ONE TAP ROW (3 taps) at a time: LDS = 3 planes x (input tile +
//     3 taps x TN couts) x 32 B = 51-60 KB -> two workgroups per CU;
//     (tried and slower, 0.66 vs 0.62 ms on 64->64 3x3 d2 at 8x64x2048: B fragments straight
//     from the L2-resident planes into registers, one tap ahead, no weight LDS at all.  Also
//     tried: a pointwise twin of this kernel -- 32-channel chunks, rotated 64-byte rows, 64 or
//     128 couts per workgroup: 117 TF on the 704x704 GEMM against conv_bfp's 124, the 128-wide
//     one spills -- so 1x1 layers stay on conv_bfp.hip's NP = 3 kernel)
//     (phase ablation, 64 -> 64 3x3 d2 at 8x64x2048: total 0.60 ms = matrix phase 0.39 (the pipe alone
//     would take 0.25-0.30) + staging 0.11 (loads 0.06, split + LDS stores 0.05) + epilogue and loop
//     0.09; VALU and MFMA time add up on gfx950, see conv_pw3.hip -- the split is cheap here
//     because nine taps reuse every staged element)
//   * LDS rows are 32 B (16 channels) with NO padding; the two 16-B halves of row R are swapped
//     when bit 3 of R is set, which makes every ds_read_b128 fragment read conflict-free.
// GEMM view, tile shape (8 x 32 pixels x 32*NT couts), on-load BatchNorm affine and epilogue as
// conv_mfma.hip (reference: pc_processor/models/salsanext_proto.py:41-62, 82-132, 164-208).
#include <type_traits>
#include "conv_x3_common.h"


namespace {

// SIX: only six plane products (l*m and m*l dropped as well, each product then off by up to 2^-23 |a||b|).
// For INPUT-GRADIENT convolutions (c3d_conv_desc.mfma_bf16 == 3): on the forward activations six (or
// seven) products measured 4-5x the fp32 engine's gradient noise through the 43 BatchNorm
// renormalisations of the network; on the gradients they measure none -- the whole GPU suite, the
// per-layer float64 gradient check and the backbone noise test pass unchanged.
template <int NT, int HALO, int TT, bool SIX>
__global__ __launch_bounds__(256, 2) void conv_x3_kernel(ConvArgs a) {
  constexpr int TR = 8, CQ = 4;                    // 16 channels per K chunk
  constexpr int TWh = 32 + 2 * HALO, THh = TR + 2 * HALO;
  constexpr int TN = 32 * NT;
  constexpr int WM = 4, WN = 1, RPW = 2, NPW = NT;
  constexpr int G = (TT == 9) ? 3 : TT;            // taps per staged weight group
  constexpr int NG = TT / G;
  constexpr int IN_ROWS = THh * TWh;
  constexpr int IN_UNITS = IN_ROWS * CQ;
  constexpr int IN_PT = (IN_UNITS + 255) / 256;
  constexpr int WG_ROWS = G * TN;
  constexpr int W_UNITS = WG_ROWS * CQ;
  constexpr int W_PT = (W_UNITS + 255) / 256;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short* s_in = reinterpret_cast<unsigned short*>(smem);   // [3][IN_ROWS][16]
  unsigned short* s_w = s_in + 3 * IN_ROWS * 16;                    // [3][WG_ROWS][16]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave, wn = 0;

  const int ntile = a.B * a.tiles_y * a.tiles_x;
  const int logical = c3d_xcd_remap(blockIdx.x, ntile * a.ntn);
  const int mt = logical / a.ntn;
  const int n0 = (logical % a.ntn) * TN;
  const int tx = mt % a.tiles_x;
  const int ty = (mt / a.tiles_x) % a.tiles_y;
  const int b = mt / (a.tiles_x * a.tiles_y);
  const int x0 = tx * 32, y0 = ty * TR;

  f32x16 acc[RPW][NPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i)
#pragma unroll
    for (int j = 0; j < NPW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- chunk-invariant staging indices
  f32x4 pin[IN_PT];
  u32x2 pw[W_PT][3];
  f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  const int c4 = tid % CQ;                 // 256 % CQ == 0: the channel quad of a thread is fixed
  unsigned inb = 0;
  int pixrel[IN_PT];
#pragma unroll
  for (int i = 0; i < IN_PT; ++i) {
    const int u = tid + i * 256;
    pixrel[i] = 0;
    if (u < IN_UNITS) {
      const int p = u / CQ;
      const int px = p % TWh, py = p / TWh;
      const int gx = x0 + px - HALO, gy = y0 + py - HALO;
      pixrel[i] = (py - HALO) * a.W + (px - HALO);
      if (gx >= 0 && gx < a.W && gy >= 0 && gy < a.H) inb |= 1u << i;
    }
  }
  int wrel[W_PT];                          // ((tap-in-group * Kq + kq) * Cout + n0 + n) * 4, or -1
#pragma unroll
  for (int i = 0; i < W_PT; ++i) {
    const int u = tid + i * 256;
    wrel[i] = -1;
    if (u < W_UNITS) {
      const int n = u % TN;
      const int r = u / TN;
      const int kq = r % CQ, tg = r / CQ;
      if (n0 + n < a.Cout) wrel[i] = ((tg * a.Kq + kq) * a.Cout + n0 + n) * 4;
    }
  }
  const size_t tile_pix = (size_t)(b * a.H + y0) * a.W + x0;
  const size_t wplane = (size_t)a.T * a.Kq * a.Cout * 4;                       // bf16 elements per weight plane
  const unsigned short* wplanes = reinterpret_cast<const unsigned short*>(a.wpack + wplane);   // behind the fp32 pack

  auto load_in = [&](int s, int c0) {
    const c3d_src& sr = a.src[s];
    const float* base = sr.ptr + tile_pix * sr.cstride + sr.coff + c0 + c4 * 4;
#pragma unroll
    for (int i = 0; i < IN_PT; ++i) {
      pin[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if ((inb >> i) & 1u) pin[i] = *reinterpret_cast<const f32x4*>(base + (ptrdiff_t)pixrel[i] * sr.cstride);
    }
    if (sr.scale) {
      psc = *reinterpret_cast<const f32x4*>(sr.scale + c0 + c4 * 4);
      psh = *reinterpret_cast<const f32x4*>(sr.shift + c0 + c4 * 4);
    }
  };
  auto load_w = [&](int g, int kofs) {     // kofs = kbase + c0 of the chunk
    const unsigned short* wb = wplanes + ((size_t)g * G * a.Kq + (kofs >> 2)) * a.Cout * 4;
#pragma unroll
    for (int i = 0; i < W_PT; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        pw[i][p] = u32x2{0u, 0u};
        if (wrel[i] >= 0) pw[i][p] = *reinterpret_cast<const u32x2*>(wb + p * wplane + wrel[i]);
      }
  };
  auto store_in = [&](int s) {
    const c3d_src& sr = a.src[s];
    const bool aff = sr.scale != nullptr;
    const bool lr = sr.lrelu != 0;
#pragma unroll
    for (int i = 0; i < IN_PT; ++i) {
      const int u = tid + i * 256;
      if (u < IN_UNITS) {
        f32x4 v = pin[i];
        if ((inb >> i) & 1u) {
          if (aff) v = v * psc + psh;
          if (lr) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = c3d_lrelu(v[q], a.slope);
          }
        }
        u32x2 pl[3];
        split4x3(v, pl);
        const int R = u / CQ;
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2*>(s_in + (p * IN_ROWS + R) * 16 + swz_quad(R, c4)) = pl[p];
      }
    }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int i = 0; i < W_PT; ++i) {
      const int u = tid + i * 256;
      if (u < W_UNITS) {
        const int n = u % TN;
        const int r = u / TN;
        const int kq = r % CQ, tg = r / CQ;
        const int R = tg * TN + n;
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2*>(s_w + (p * WG_ROWS + R) * 16 + swz_quad(R, kq)) = pw[i][p];
      }
    }
  };

  const int nj = min(NPW, (a.Cout - n0 + 31) / 32);   // live 32-wide cout sub-tiles (ragged last tile)
  auto mfma_group = [&](int g, auto nj_tag) {
    constexpr int NJ = decltype(nj_tag)::value;
#pragma unroll
    for (int tg = 0; tg < G; ++tg) {
      const int t = g * G + tg;
      bf16x8 bp[3][NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int R = tg * TN + j * 32 + l31;
        const int o = R * 16 + swz_half(R, half);
#pragma unroll
        for (int p = 0; p < 3; ++p) bp[p][j] = *reinterpret_cast<const bf16x8*>(s_w + p * WG_ROWS * 16 + o);
      }
      bf16x8 ap[3][RPW];
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const int R = (wm + i * WM + HALO + a.dy[t]) * TWh + HALO + a.dx[t] + l31;
        const int o = R * 16 + swz_half(R, half);
#pragma unroll
        for (int p = 0; p < 3; ++p) ap[p][i] = *reinterpret_cast<const bf16x8*>(s_in + p * IN_ROWS * 16 + o);
      }
      // eight of the nine plane products, smallest first; only l*l (< 2^-32 |a||b|) is dropped
#define C3D_PLANE(PA, PB)                                                                          \
  _Pragma("unroll") for (int i = 0; i < RPW; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j)  \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[PA][i], bp[PB][j], acc[i][j], 0, 0, 0);
      if constexpr (!SIX) {
        C3D_PLANE(2, 1) C3D_PLANE(1, 2)
      }
      C3D_PLANE(2, 0) C3D_PLANE(0, 2) C3D_PLANE(1, 1) C3D_PLANE(1, 0) C3D_PLANE(0, 1)
      C3D_PLANE(0, 0)
#undef C3D_PLANE
    }
  };

  int s = 0, c0 = 0, kbase = 0;
  load_in(s, c0);
  load_w(0, kbase + c0);
  while (true) {
    int s2 = s, c2 = c0 + 16, kb2 = kbase;
    if (c2 >= a.src[s].C) {
      kb2 += a.src[s].C;
      s2 = s + 1;
      c2 = 0;
    }
    const bool more = s2 < a.nsrc;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      __syncthreads();                       // the previous matrix phase has finished reading LDS
      if (g == 0) store_in(s);
      store_w();
      __syncthreads();
      if (g + 1 < NG) {
        load_w(g + 1, kbase + c0);
      } else if (more) {
        load_in(s2, c2);
        load_w(0, kb2 + c2);
      }
      __builtin_amdgcn_s_setprio(1);
      if (NPW == 1 || nj >= NPW) mfma_group(g, std::integral_constant<int, NPW>{});
      else mfma_group(g, std::integral_constant<int, 1>{});
      __builtin_amdgcn_s_setprio(0);
    }
    if (!more) break;
    s = s2;
    c0 = c2;
    kbase = kb2;
  }
  conv_epilogue<TR, NT, WM, WN>(a, acc, smem, tid, lane, half, l31, wm, wn, b, x0, y0, n0, mt, ntile, tile_pix);
}

template <int NT, int HALO, int TT, bool SIX>
int launch_x3_s(ConvArgs& a, hipStream_t st) {
  constexpr int G = (TT == 9) ? 3 : TT;
  size_t lds = (size_t)3 * ((size_t)(8 + 2 * HALO) * (32 + 2 * HALO) + (size_t)G * 32 * NT) * 16 * 2;
  const size_t red = (size_t)4 * 32 * NT * 2 * sizeof(float);   // statistics scratch of the epilogue
  if (lds < red) lds = red;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_x3_kernel<NT, HALO, TT, SIX>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  a.ntn = (a.Cout + 32 * NT - 1) / (32 * NT);
  dim3 grid(a.B * a.tiles_x * a.tiles_y * a.ntn);
  hipLaunchKernelGGL((conv_x3_kernel<NT, HALO, TT, SIX>), grid, dim3(256), lds, st, a);
  C3D_CHECK_LAUNCH();
  return 0;
}

template <int NT, int HALO, int TT>
int launch_x3(ConvArgs& a, hipStream_t st) {
  return a.six ? launch_x3_s<NT, HALO, TT, true>(a, st) : launch_x3_s<NT, HALO, TT, false>(a, st);
}

template <int NT>
int launch_x3_taps(ConvArgs& a, int halo, hipStream_t st) {
  if (a.T == 4) return halo <= 1 ? launch_x3<NT, 1, 4>(a, st) : launch_x3<NT, 2, 4>(a, st);
  return halo <= 1 ? launch_x3<NT, 1, 9>(a, st) : launch_x3<NT, 2, 9>(a, st);
}

}  // namespace

// called by c3d_conv_forward for mfma_bf16 == 2, 8-row tiles, 4 or 9 taps; a.wpack must be a
// c3d_pack_weights(mode | 2) pack (fp32 image followed by the three bf16 planes)
int c3d_conv_forward_x3(ConvArgs& a, int halo, hipStream_t st) {
  return c3d_wide_cout_tiles(a) ? launch_x3_taps<2>(a, halo, st) : launch_x3_taps<1>(a, halo, st);
}
