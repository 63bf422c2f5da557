// Implicit-GEMM convolution on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16) for gfx950, with
// fp32 tensors in HBM and fp32 accumulation.  Opt-in precision modes of c3d_conv_forward
// (c3d_conv_desc.mfma_bf16); the fp32-MFMA kernel in conv_mfma.hip stays the default path.
//
//   NP = 1  "bf16":   every operand is rounded once to bf16 (RNE) while it is staged in LDS.
//   NP = 3  "bf16x3": every fp32 operand is split EXACTLY into three bf16 planes
//                     x = h + m + l  (h = RNE8(x), m = RNE8(x-h), l = RNE8(x-h-m); both
//                     residuals are exact in fp32, 3 x 8 significand bits cover all 24) and eight
//                     of the nine plane products are accumulated, smallest first:
//                     l*m + m*l + l*h + h*l + m*m + m*h + h*m + h*h.  Only l*l (< 2^-32 |a||b|) is
//                     dropped, so each product a*b is exact far below fp32 resolution and the
//                     result carries fp32 accumulation rounding only -- at 8/16 of the fp32-MFMA
//                     issue time.  (Six products -- without l*m, m*l -- measured ~4x the gradient
//                     noise of the fp32 engine on the whole network: tests/test_gpu_backbone.py.)
//
// Same GEMM view, tile shapes, on-load BatchNorm affine and epilogue as conv_mfma.hip.  The split
// happens ONCE per staged element (not per fragment read): LDS holds NP bf16 images of the
// input tile and of the weight slab, rows padded to CK+8 bf16 (48 / 80 B: conflict-free
// ds_read_b128); a fragment read is one b128 per plane = one MFMA operand (lane l: 8 consecutive
// k of row l&31, k-group l>>5).
#include "conv_common.h"

namespace {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

// 4 floats -> NP x (4 bf16 packed in 2 dwords)
// EXPERIMENT (NP == 2, "f16x2"): two fp16 planes, x = H + L to 2^-22 |x| (worst case), three products H*H' + H*L' + L*H'
template <int NP>
__device__ __forceinline__ void split4(f32x4 v, u32x2 (&out)[NP]) {
  if constexpr (NP == 2) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    f32x4 r = v;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      f16x4 h;
#pragma unroll
      for (int q = 0; q < 4; ++q) h[q] = (_Float16)r[q];
      out[p] = __builtin_bit_cast(u32x2, h);
      if (p == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] -= (float)h[q];
      }
    }
    return;
  }
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  f32x4 r = v;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    bf16x4 h;
#pragma unroll
    for (int q = 0; q < 4; ++q) h[q] = (__bf16)r[q];
    out[p] = __builtin_bit_cast(u32x2, h);
    if (p + 1 < NP) {
#pragma unroll
      for (int q = 0; q < 4; ++q) r[q] -= (float)h[q];
    }
  }
}

// BFS (round 4, NP = 1 with every source a bf16 tensor): the chunk in flight stays the 8 bytes per unit it is loaded as (widened
// when it is stored to LDS), which makes room for DEEPER K chunks -- 64 channels for 1x1 convs, 32 for the four-tap ones:
// with 16 / 32-channel chunks a pixel contributes 32 / 64 bytes to a load instruction and this kernel took as long over bf16
// tensors as over fp32 ones (requests, not bytes: 704 -> 64 at 8 x 32 x 1024 even 248 vs 166 us).
// SM (round 5, BFS only): the instance with the BatchNorm-backward epilogue (ConvArgs::stat_mul) -- a kernel of its own, two
// workgroups per CU: compiled into the common instance the epilogue's multiplier tile pushed it past its register cap (303
// spilled registers at three workgroups per CU) for EVERY launch, with or without stat_mul.
template <int TR, int NT, int CK, int HALO, int TT, int NP, bool BFS = false, bool SM = false>
__global__ __launch_bounds__(256, (!SM && NP == 1 && !(TT >= 6 && NT == 2 && TR == 8) && !(BFS && NT == 2 && (CK == 64 || (TT == 4 && CK == 32)))) ? 3 : 2)
    void conv_bfp_kernel(ConvArgs a) {
  static_assert(!BFS || NP == 1, "raw bf16 staging belongs to the one-plane engine");
  static_assert(!SM || BFS, "the separate stat_mul instance exists for the raw-bf16 kernels");
  using PinT = std::conditional_t<BFS, u32x2, f32x4>;
  constexpr int CSB = CK + 8;            // bf16 elements per LDS row
  constexpr int TWh = 32 + 2 * HALO;
  constexpr int THh = TR + 2 * HALO;
  constexpr int TN = 32 * NT;
  constexpr int WM = (TR >= 4) ? 4 : TR;
  constexpr int WN = 4 / WM;
  constexpr int RPW = TR / WM;
  constexpr int NPW = NT / WN;
  static_assert(NT % WN == 0, "NT must split across waves");
  static_assert(NT / WN <= 2, "the ragged-tile path assumes at most two cout sub-tiles per wave");
  static_assert(CK % 16 == 0, "K chunk must be a multiple of the MFMA K (16)");
  constexpr int CQ = CK / 4;
  constexpr int IN_ROWS = THh * TWh;
  constexpr int W_ROWS = TT * TN;
  constexpr int IN_UNITS = IN_ROWS * CQ;
  constexpr int IN_PT = (IN_UNITS + 255) / 256;
  constexpr int W_UNITS = W_ROWS * CQ;
  constexpr int W_PT = (W_UNITS + 255) / 256;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short* s_in = reinterpret_cast<unsigned short*>(smem);   // [NP][IN_ROWS][CSB]
  unsigned short* s_w = s_in + NP * IN_ROWS * CSB;                  // [NP][W_ROWS][CSB]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave % WM, wn = wave / WM;

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

  // ---- register staging (prefetch) state, as in conv_mfma.hip
  PinT pin[IN_PT];
  f32x4 pw[W_PT];
  f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  const int c4 = tid % CQ;
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
  int wrel[W_PT];
#pragma unroll
  for (int i = 0; i < W_PT; ++i) {
    const int u = tid + i * 256;
    wrel[i] = -1;
    if (u < W_UNITS) {
      const int n = u % TN;
      const int r = u / TN;
      const int kq = r % CQ, t = r / CQ;
      if (n0 + n < a.Cout) wrel[i] = (t * a.Kq + kq) * a.Cout + n0 + n;
    }
  }
  const size_t tile_pix = (size_t)(b * a.H + y0) * a.W + x0;

  auto load_chunk = [&](int s, int c0, int kbase) {
    const c3d_src& sr = a.src[s];
    if constexpr (BFS) {             // every source is bf16: the 8 bytes stay as they are until store_chunk
      const unsigned short* base = reinterpret_cast<const unsigned short*>(sr.ptr) + tile_pix * sr.cstride + sr.coff + c0 + c4 * 4;
#pragma unroll
      for (int i = 0; i < IN_PT; ++i) {
        pin[i] = u32x2{0u, 0u};
        if ((inb >> i) & 1u) pin[i] = __builtin_bit_cast(u32x2, *reinterpret_cast<const c3d_u32x2*>(base + (ptrdiff_t)pixrel[i] * sr.cstride));
      }
    } else if (NP == 1 && sr.bf16) {        // bf16 activation storage: 8-byte loads, widened in registers
      if constexpr (!BFS) {
        const unsigned short* base = reinterpret_cast<const unsigned short*>(sr.ptr) + tile_pix * sr.cstride + sr.coff + c0 + c4 * 4;
#pragma unroll
        for (int i = 0; i < IN_PT; ++i) {
          pin[i] = f32x4{0.f, 0.f, 0.f, 0.f};
          if ((inb >> i) & 1u) {
            const c3d_u32x2 r = *reinterpret_cast<const c3d_u32x2*>(base + (ptrdiff_t)pixrel[i] * sr.cstride);
            pin[i] = f32x4{__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16),
                           __uint_as_float(r[1] & 0xffff0000u)};
          }
        }
      }
    } else {
      if constexpr (!BFS) {
        const float* base = sr.ptr + tile_pix * sr.cstride + sr.coff + c0 + c4 * 4;
#pragma unroll
        for (int i = 0; i < IN_PT; ++i) {
          pin[i] = f32x4{0.f, 0.f, 0.f, 0.f};
          if ((inb >> i) & 1u) pin[i] = *reinterpret_cast<const f32x4*>(base + (ptrdiff_t)pixrel[i] * sr.cstride);
        }
      }
    }
    if (sr.scale) {
      psc = *reinterpret_cast<const f32x4*>(sr.scale + c0 + c4 * 4);
      psh = *reinterpret_cast<const f32x4*>(sr.shift + c0 + c4 * 4);
    }
    const float* wbase = a.wpack + (size_t)((kbase + c0) >> 2) * a.Cout * 4;
#pragma unroll
    for (int i = 0; i < W_PT; ++i) {
      pw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (wrel[i] >= 0) pw[i] = *reinterpret_cast<const f32x4*>(wbase + (size_t)wrel[i] * 4);
    }
  };

  auto store_chunk = [&](int s) {
    const c3d_src& sr = a.src[s];
    const bool aff = sr.scale != nullptr;
    const bool lr = sr.lrelu != 0;
#pragma unroll
    for (int i = 0; i < IN_PT; ++i) {
      const int u = tid + i * 256;
      if (u < IN_UNITS) {
        f32x4 v;
        if constexpr (BFS) {
          v = f32x4{__uint_as_float(pin[i][0] << 16), __uint_as_float(pin[i][0] & 0xffff0000u), __uint_as_float(pin[i][1] << 16),
                    __uint_as_float(pin[i][1] & 0xffff0000u)};
        } else {
          v = pin[i];
        }
        if ((inb >> i) & 1u) {
          if (aff) v = v * psc + psh;
          if (lr) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = c3d_lrelu(v[q], a.slope);
          }
        }
        if constexpr (NP == 2) v = v * 64.f;            // f16x2: keep the low plane out of fp16's subnormals (exact)
        u32x2 pl[NP];
        split4<NP>(v, pl);
#pragma unroll
        for (int p = 0; p < NP; ++p)
          *reinterpret_cast<u32x2*>(s_in + (p * IN_ROWS + u / CQ) * CSB + c4 * 4) = pl[p];
      }
    }
#pragma unroll
    for (int i = 0; i < W_PT; ++i) {
      const int u = tid + i * 256;
      if (u < W_UNITS) {
        const int n = u % TN;
        const int r = u / TN;
        const int kq = r % CQ, t = r / CQ;
        u32x2 pl[NP];
        split4<NP>(NP == 2 ? pw[i] * 1024.f : pw[i], pl);
#pragma unroll
        for (int p = 0; p < NP; ++p)
          *reinterpret_cast<u32x2*>(s_w + (p * W_ROWS + t * TN + n) * CSB + kq * 4) = pl[p];
      }
    }
  };

  const int nj = min(NPW, (a.Cout - n0 - wn * NPW * 32 + 31) / 32);   // live 32-wide cout sub-tiles (conv_mfma.hip)
  int s = 0, c0 = 0, kbase = 0;
  bool mul_dma = false;
  load_chunk(s, c0, kbase);
  while (true) {
    __syncthreads();
    store_chunk(s);
    __syncthreads();
    int s2 = s, c2 = c0 + CK, kb2 = kbase;
    if (c2 >= a.src[s].C) {
      kb2 += a.src[s].C;
      s2 = s + 1;
      c2 = 0;
    }
    const bool more = s2 < a.nsrc;
    if (more) load_chunk(s2, c2, kb2);
    if constexpr (SM) if (!more) mul_dma = conv_mul_dma_issue<TR, TN, 256>(a, smem, tid, x0, y0, n0, tile_pix);      // (round 6: under the last chunk)
    __builtin_amdgcn_s_setprio(1);
    // NJ = live 32-wide cout sub-tiles of this wave, a compile-time constant per code path: a
    // per-MFMA predicate (as the fp32 kernel uses) broke the bf16 MFMA schedule (68 vs 53 ms/step)
    auto mfma_phase = [&](auto nj_tag) {
      constexpr int NJ = decltype(nj_tag)::value;
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const int tap_off = ((HALO + a.dy[t]) * TWh + (HALO + a.dx[t]) + l31) * CSB + half * 8;
        const unsigned short* wb = s_w + (t * TN + wn * NPW * 32 + l31) * CSB + half * 8;
#pragma unroll
        for (int kk = 0; kk < CK / 16; ++kk) {
          bf16x8 bp[NP][NJ];
#pragma unroll
          for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              bp[p][j] = *reinterpret_cast<const bf16x8*>(wb + (p * W_ROWS + j * 32) * CSB + kk * 16);
          bf16x8 ap[NP][RPW];
#pragma unroll
          for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int i = 0; i < RPW; ++i)
              ap[p][i] = *reinterpret_cast<const bf16x8*>(s_in + (p * IN_ROWS + (wm + i * WM) * TWh) * CSB + tap_off + kk * 16);
#define C3D_PLANE(PA, PB)                                                                          \
  _Pragma("unroll") for (int i = 0; i < RPW; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j)  \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[PA][i], bp[PB][j], acc[i][j], 0, 0, 0);
#define C3D_PLANE_H(PA, PB)                                                                        \
  _Pragma("unroll") for (int i = 0; i < RPW; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j)  \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, ap[PA][i]),   \
                                                          __builtin_bit_cast(f16x8_t, bp[PB][j]), acc[i][j], 0, 0, 0);
          if constexpr (NP == 2) {
            C3D_PLANE_H(1, 0) C3D_PLANE_H(0, 1) C3D_PLANE_H(0, 0)
          } else if constexpr (NP == 3) {
            // eight of the nine plane products, smallest first; only l*l (< 2^-32 |a||b|) is dropped:
            // every a*b is then exact to 2^-32, i.e. the result carries fp32 ACCUMULATION rounding only
            C3D_PLANE(2, 1) C3D_PLANE(1, 2) C3D_PLANE(2, 0) C3D_PLANE(0, 2) C3D_PLANE(1, 1) C3D_PLANE(1, 0) C3D_PLANE(0, 1)
            C3D_PLANE(0, 0)
          } else {
            C3D_PLANE(0, 0)
          }
#undef C3D_PLANE
#undef C3D_PLANE_H
        }
      }
    };
    if (NPW == 1 || nj >= NPW) mfma_phase(std::integral_constant<int, NPW>{});
    else mfma_phase(std::integral_constant<int, 1>{});
    __builtin_amdgcn_s_setprio(0);
    if (!more) break;
    s = s2;
    c0 = c2;
    kbase = kb2;
  }
  conv_epilogue<TR, NT, WM, WN, NP == 1, false, 256, false, (NP >= 2 || SM)>(a, acc, smem, tid, lane, half, l31, wm, wn, b, x0, y0, n0, mt, ntile,
                                                                     tile_pix, mul_dma);
}

template <int TR, int NT, int CK, int HALO, int TT, int NP, bool BFS = false>
int launch_bfp(ConvArgs& a, hipStream_t st) {
  constexpr int CSB = CK + 8;
  size_t lds = (size_t)NP * ((size_t)(TR + 2 * HALO) * (32 + 2 * HALO) + (size_t)TT * 32 * NT) * CSB * 2;
  const size_t red = (size_t)4 * 32 * NT * 2 * sizeof(float);   // statistics scratch of the epilogue
  if (lds < red) lds = red;
  a.ntn = (a.Cout + 32 * NT - 1) / (32 * NT);
  dim3 grid(a.B * a.tiles_x * a.tiles_y * a.ntn);
  if constexpr (BFS) {
    if (a.stat_mul && a.stat_partial) {      // BatchNorm-backward sums in the epilogue: the instance that has it
      if (lds < (size_t)TR * 32 * (32 * NT + 8) * 2) lds = (size_t)TR * 32 * (32 * NT + 8) * 2;      // the multiplier tile of the epilogue
      if (!(a.variant & 64) && (lds + 15) / 16 * 16 + (size_t)TR * 32 * (32 * NT) * 2 <= 80 * 1024) {
        // round 6: a region of its own behind the K loop's buffers, filled by LDS-DMA under the last chunk (two workgroups per CU kept)
        lds = (lds + 15) / 16 * 16;
        a.mul_lds_off = (unsigned)lds;
        lds += (size_t)TR * 32 * (32 * NT) * 2;
      }
      a.lds_bytes = (unsigned)lds;
      c3d_opt_in_lds<&conv_bfp_kernel<TR, NT, CK, HALO, TT, NP, BFS, true>>();
      hipLaunchKernelGGL((conv_bfp_kernel<TR, NT, CK, HALO, TT, NP, BFS, true>), grid, dim3(256), lds, st, a);
      C3D_CHECK_LAUNCH();
      return 0;
    }
  }
  c3d_opt_in_lds<&conv_bfp_kernel<TR, NT, CK, HALO, TT, NP, BFS>>();
  hipLaunchKernelGGL((conv_bfp_kernel<TR, NT, CK, HALO, TT, NP, BFS>), grid, dim3(256), lds, st, a);
  C3D_CHECK_LAUNCH();
  return 0;
}

template <int TR, int NT, int NP>
int launch_bfp_taps(ConvArgs& a, int halo, hipStream_t st) {
  if (a.T == 1) return launch_bfp<TR, NT, 16, 0, 1, NP>(a, st);
  if (a.T == 4) return halo <= 1 ? launch_bfp<TR, NT, 16, 1, 4, NP>(a, st) : launch_bfp<TR, NT, 16, 2, 4, NP>(a, st);
  if constexpr (NP != 2) {      // (3 / 6 taps, halo 1: RangeNet's strided / transposed convs over column-pair views)
    if (a.T == 3) return launch_bfp<TR, NT, 16, 1, 3, NP>(a, st);
    if (a.T == 6) return launch_bfp<TR, NT, 16, 1, 6, NP>(a, st);
  }
  if (halo <= 1) return launch_bfp<TR, NT, 16, 1, 9, NP>(a, st);
  return launch_bfp<TR, NT, 16, 2, 9, NP>(a, st);
}

// one plane, 8-row tiles, every source a bf16 tensor (see BFS at the kernel): 1x1 convs in chunks of 64 channels (32 where a
// source's width is not a multiple of 64), four-tap convs in chunks of 32 (16 likewise).  c3d_conv_desc.variant & 8: the
// plain instantiations (tests).  Returns -1 when the launch has no such form.
template <int NT>
int launch_bfp_bf16_sources(ConvArgs& a, int halo, hipStream_t st) {
  bool all_bf = !(a.variant & 8), c32 = true, c64 = true;
  for (int s = 0; s < a.nsrc; ++s) {
    all_bf = all_bf && a.src[s].bf16 != 0;
    c32 = c32 && a.src[s].C % 32 == 0;
    c64 = c64 && a.src[s].C % 64 == 0;
  }
  if (!all_bf) return -1;
  if (a.T == 1) {
    if (c64) return launch_bfp<8, NT, 64, 0, 1, 1, true>(a, st);
    if (c32) return launch_bfp<8, NT, 32, 0, 1, 1, true>(a, st);
    return launch_bfp<8, NT, 16, 0, 1, 1, true>(a, st);
  }
  if (a.T == 4) {
    if (c32) return halo <= 1 ? launch_bfp<8, NT, 32, 1, 4, 1, true>(a, st) : launch_bfp<8, NT, 32, 2, 4, 1, true>(a, st);
    return halo <= 1 ? launch_bfp<8, NT, 16, 1, 4, 1, true>(a, st) : launch_bfp<8, NT, 16, 2, 4, 1, true>(a, st);
  }
  return -1;
}

template <int NP>
int dispatch_bfp(ConvArgs& a, int tr, int halo, bool k32, hipStream_t st) {
  const bool wide = c3d_wide_cout_tiles(a);
  if constexpr (NP == 1) {
    if (tr == 8) {
      const int rc = wide ? launch_bfp_bf16_sources<2>(a, halo, st) : launch_bfp_bf16_sources<1>(a, halo, st);
      if (rc >= 0) return rc;
    }
  }
  if (tr == 8 && a.T == 1 && k32) return wide ? launch_bfp<8, 2, 32, 0, 1, NP>(a, st) : launch_bfp<8, 1, 32, 0, 1, NP>(a, st);
  if (tr == 8) return wide ? launch_bfp_taps<8, 2, NP>(a, halo, st) : launch_bfp_taps<8, 1, NP>(a, halo, st);
  if (tr == 4) return wide ? launch_bfp_taps<4, 2, NP>(a, halo, st) : launch_bfp_taps<4, 1, NP>(a, halo, st);
  return launch_bfp_taps<2, 2, NP>(a, halo, st);
}

}  // namespace

// called by c3d_conv_forward (conv_mfma.hip) for mfma_bf16 = 1 (planes = 1) or 2 (planes = 3)
int c3d_conv_forward_bfp(ConvArgs& a, int planes, int tr, int halo, bool k32, hipStream_t st) {
  if (planes == 2) {     // EXPERIMENT (mfma_bf16 == 4)
    a.acc_scale = 1.f / 65536.f;                                   // operands are staged times 2^6 and 2^10
    return dispatch_bfp<2>(a, tr, halo, k32, st);
  }
  return planes == 3 ? dispatch_bfp<3>(a, tr, halo, k32, st) : dispatch_bfp<1>(a, tr, halo, k32, st);
}
