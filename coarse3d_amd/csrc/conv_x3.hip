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
//     0.09 -- the phases of this kernel run in lockstep and its staging was packed-f32 VALU, so they
//     add up; the fused kernel below deals the staging into the MFMA stream instead)
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
      // (the three 2^-16-class products m*m, l*h, h*l in this order: the fused kernel below wants a tap to open with
      //  planes other than the ones the previous tap closed with, and the two kernels stay bit-identical)
      if constexpr (!SIX) {
        C3D_PLANE(2, 1) C3D_PLANE(1, 2) C3D_PLANE(2, 0) C3D_PLANE(0, 2) C3D_PLANE(1, 1)
      } else {
        C3D_PLANE(1, 1) C3D_PLANE(2, 0) C3D_PLANE(0, 2)
      }
      C3D_PLANE(1, 0) C3D_PLANE(0, 1)
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
  conv_epilogue<TR, NT, WM, WN, false, false, 256, false, true>(a, acc, smem, tid, lane, half, l31, wm, wn, b, x0, y0, n0, mt, ntile, tile_pix);
}

// ---------------------------------------------------------------------------------------------------
// Round 3: the same tile, LDS image, arithmetic and product order with the staging dealt into the MFMA
// stream (profiles/round3_coissue_probe.md; the method of conv_pw3.hip's fused kernel).  What changes against
// conv_x3_kernel above:
//   * the split of the NEXT chunk's input tile (affine, LeakyReLU, three planes: ~25 scalar VALU per unit)
//     runs between the MFMAs of the current chunk and leaves the planes in registers; only their LDS stores
//     (3 x IN_PT ds_write_b64 per thread) remain between two barriers, once per chunk -- the input tile stays
//     single-buffered (two workgroups per CU);
//   * the weight slab is double-buffered (+18 KB): the next tap row's planes are stored while the current one is
//     multiplied, so a tap row costs one barrier instead of two (4 per chunk instead of 6);
//   * the fragments of the next tap are read while the current tap is multiplied (each plane right after
//     its last product; the B plane that closes one tap and opens the next is double-buffered), so the LDS
//     latency is exposed once per tap row instead of once per tap;
//   * every load is a buffer load (descriptor + K offset in SGPRs, one 32-bit voffset per unit; pixels outside
//     the image and couts beyond Cout carry an out-of-range voffset and read zeros): the next chunk's input is
//     requested in the first tap row and converted in the later ones (a tap row's time of lead, and the raw
//     values and the planes never fill their registers at the same time), weights one tap row ahead;
//   * no packed-f32 VALU (NOPK in the Makefile).
// Outputs are bit-identical to conv_x3_kernel (tests/test_gpu_conv.py::test_fused_multitap_kernel_...).
__device__ float c3d_x3_unit_affine[32] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f,
                                           0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

template <int I, int N, class F>
__device__ __forceinline__ void c3d_x3_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    c3d_x3_static_for<I + 1, N>(f);
  }
}

// plane products of one tap in issue order (A plane, B plane), smallest first
template <bool SIX>
struct c3d_x3_products {
  static constexpr int N = SIX ? 6 : 8;
  static constexpr int pa(int q) {
    constexpr int six[6] = {1, 2, 0, 1, 0, 0}, eight[8] = {2, 1, 2, 0, 1, 1, 0, 0};
    return SIX ? six[q] : eight[q];
  }
  static constexpr int pb(int q) {
    constexpr int six[6] = {1, 0, 2, 0, 1, 0}, eight[8] = {1, 2, 0, 2, 1, 0, 1, 0};
    return SIX ? six[q] : eight[q];
  }
  static constexpr int last_a(int p) { int l = -1; for (int q = 0; q < N; ++q) if (pa(q) == p) l = q; return l; }
  static constexpr int last_b(int p) { int l = -1; for (int q = 0; q < N; ++q) if (pb(q) == p) l = q; return l; }
};

// EXPERIMENT (NPL = 2, "f16x2"): two fp16 planes per operand, three products -- L*H', H*H', H*L' (no plane both opens and
// closes a tap); DESIGN.md round-4 list, profiles/round3_f16x2_probe.txt
struct c3d_x3_products_f16 {
  static constexpr int N = 3;
  static constexpr int pa(int q) { constexpr int t[3] = {1, 0, 0}; return t[q]; }
  static constexpr int pb(int q) { constexpr int t[3] = {0, 0, 1}; return t[q]; }
  static constexpr int last_a(int p) { int l = -1; for (int q = 0; q < N; ++q) if (pa(q) == p) l = q; return l; }
  static constexpr int last_b(int p) { int l = -1; for (int q = 0; q < N; ++q) if (pb(q) == p) l = q; return l; }
};

// NPL = 1 (round 4, the `bf16` mixed-precision engine, BASELINE configs[2]): one bf16 plane per operand -- the activations are
// rounded once while they are staged (BFS: read from bf16 tensors, 8 bytes per unit), the weights are plane 0 of the pack --
// and ONE product per tap; the fragments of the next tap are then read into a second register set while the current tap is
// multiplied.  Same tile, LDS image and accumulation order as conv_bfp_kernel<8, NT, 16, HALO, 9, 1>: bit-identical outputs
// (tests/test_gpu_bf16_storage.py).
struct c3d_x3_products_one {
  static constexpr int N = 1;
  static constexpr int pa(int) { return 0; }
  static constexpr int pb(int) { return 0; }
  static constexpr int last_a(int) { return 0; }
  static constexpr int last_b(int) { return 0; }
};

// SM (round 5, the one-plane kernel over bf16 tensors): the instance with the BatchNorm-backward epilogue (ConvArgs::stat_mul);
// the common instance stays without it (34 spilled registers otherwise, for every launch)
// taps per staged weight group of the fused kernel: nine taps = three rows of three, four = two rows of two; round 5: six (the
// stride-(1, 2) conv of RangeNet over a column-pair view: three rows of two) and three (its transposed conv: three "rows" of one)
constexpr int c3d_x3f_group(int tt) { return tt == 9 ? 3 : ((tt == 4 || tt == 6) ? 2 : (tt == 3 ? 1 : tt)); }

// PLAIN (round 6, the three-plane engine): every source of the launch comes without BatchNorm affine and without LeakyReLU -- the
// input-gradient launches, whose source is dz.  The on-load transform is then the identity (fma(x, 1, 0), max(v, 1 * v)) and
// the zero padding needs no mask (an out-of-range buffer load returns 0 and there is no shift to undo): the XF atoms -- four
// of the twelve VALU operations per staged value -- and the scale / shift loads are left out.  With 32 couts per workgroup
// the fused schedule deals ~3 VALU + ~1.2 LDS operations into every MFMA gap, the edge of what hides there (guide: <= 5):
// these launches ran at 0.37-0.43 matrix-pipe busy against 0.62-0.65 for the 64-cout ones.  Same values, same order: bit-identical.
template <int NT, int HALO, int TT, bool SIX, int NPL = 3, bool BFS = false, bool SM = false, bool PLAIN = false>
__global__ __launch_bounds__(256, 2) void conv_x3f_kernel(ConvArgs a) {
  static_assert(!BFS || NPL == 1, "bf16 sources belong to the one-plane engine");
  static_assert(!PLAIN || (NPL == 3 && !BFS), "the transform-free instance exists for the three-plane engine over fp32 tensors");
  static_assert(!SM || BFS, "the separate stat_mul instance exists for the one-plane kernel over bf16 tensors");
  constexpr bool F16 = NPL == 2;
  constexpr unsigned EB = BFS ? 2u : 4u;           // bytes per stored activation element
  typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
  constexpr int TR = 8, CQ = 4;                    // 16 channels per K chunk
  constexpr int TWh = 32 + 2 * HALO, THh = TR + 2 * HALO;
  constexpr int TN = 32 * NT;
  constexpr int WM = 4, WN = 1, RPW = 2, NPW = NT;
  constexpr int G = c3d_x3f_group(TT);            // taps per staged weight group (tap row)
  constexpr int NG = TT / G;
  constexpr int IN_ROWS = THh * TWh;
  constexpr int IN_UNITS = IN_ROWS * CQ;
  constexpr int IN_PT = (IN_UNITS + 255) / 256;
  constexpr int WG_ROWS = G * TN;
  constexpr int W_UNITS = WG_ROWS * CQ;
  constexpr int W_PT = (W_UNITS + 255) / 256;
  using PR = std::conditional_t<F16, c3d_x3_products_f16, std::conditional_t<NPL == 1, c3d_x3_products_one, c3d_x3_products<SIX>>>;
  constexpr int NQ = PR::N;

  // LDS rows are padded to whole staging units (64 rows per unit index): unit i of a thread is row tid/4 + 64 i,
  // always -- lanes past the end of the tile / slab read zeros (out-of-range voffset) and store them into pad rows
  // nobody reads, so the staging needs no predicates, no clamps and no per-unit address registers
  constexpr int IN_ROWS_P = IN_PT * 64, WG_ROWS_P = W_PT * 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short* s_in = reinterpret_cast<unsigned short*>(smem);   // [NPL][IN_ROWS_P][16]
  unsigned short* s_w0 = s_in + NPL * IN_ROWS_P * 16;               // 2 x [NPL][WG_ROWS_P][16]
  constexpr int WBUF = NPL * WG_ROWS_P * 16;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  const size_t tile_pix = (size_t)(b * a.H + y0) * a.W + x0;

  f32x16 acc[RPW][NPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i)
#pragma unroll
    for (int j = 0; j < NPW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- load cursors.  Units past the end of the tile / slab repeat the last one (same value to the same LDS
  //      address); pixels outside the image and couts beyond Cout get an out-of-range voffset (reads as zero).
  constexpr unsigned OOB = 0xfffffff0u;
  const int c4 = tid % CQ;
  const int r0 = tid / CQ;                 // LDS row of unit 0; unit i: r0 + 64 i (same swizzle: 64 keeps bit 3)
  unsigned inb = 0;                        // bit i: unit i is a pixel inside the image
#pragma unroll
  for (int i = 0; i < IN_PT; ++i) {
    const int p = r0 + 64 * i;
    const int px = p % TWh, py = p / TWh;
    const int gx = x0 + px - HALO, gy = y0 + py - HALO;
    if (p < IN_ROWS && gx >= 0 && gx < a.W && gy >= 0 && gy < a.H) inb |= 1u << i;
  }
  const unsigned wplane_b = (unsigned)a.T * a.Kq * a.Cout * 8;                  // bytes per bf16 weight plane
  // (F16: the fp32 image at the head of the pack, 16 bytes per unit, split in the kernel -- no fp16 planes in the pack yet)
  constexpr unsigned WU = F16 ? 16u : 8u;                                       // bytes per weight unit in global memory
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.wpack) + (F16 ? (size_t)0 : (size_t)a.T * a.Kq * a.Cout * 4), 0, F16 ? 2 * wplane_b : 3 * wplane_b,
      0x00020000);
  // weight unit i: u = tid + 256 i -> cout n = u % TN, row r = u / TN = kq + 4 * tap-in-row; slab row tg * TN + n
  unsigned vw[W_PT];
#pragma unroll
  for (int i = 0; i < W_PT; ++i) {
    const int u = tid + i * 256;
    const int n = u % TN;
    const int r = u / TN;
    const int kq = r % CQ, tg = r / CQ;
    vw[i] = (u < W_UNITS && n0 + n < a.Cout) ? (unsigned)(((tg * a.Kq + kq) * a.Cout + n0 + n) * WU) : OOB;
  }
  // LDS byte offsets of unit 0 (plane 0): input row r0, weight row (tid / TN / 4) * TN + tid % TN; unit i adds a constant
  const int wR0 = (tid / TN / CQ) * TN + tid % TN, wkq0 = (tid / TN) % CQ;
  static_assert(256 % TN == 0 && (256 / TN) % CQ == 0 || TN == 64 || TN == 32, "weight units advance by whole tap rows");
  constexpr int W_ROW_STEP = (256 / TN / CQ) * TN;     // slab rows between consecutive units of a thread
  __amdgpu_buffer_rsrc_t rs_in, rs_sc, rs_sh;
  unsigned vin[IN_PT];
  const int vaff = c4 * 16;
  int lstep = 0;
  float lslope = 1.f;
  int ls = 0, lc0 = 0, lC = 0, lk = 0;     // input cursor: source, channel inside it, its width, chunk index
  const unsigned img_bytes_per_c = (unsigned)a.B * a.H * a.W * EB;
  auto open_src = [&](int s) {
    const c3d_src& sr = a.src[s];
    rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sr.ptr), 0, img_bytes_per_c * sr.cstride, 0x00020000);
#pragma unroll
    for (int i = 0; i < IN_PT; ++i) {
      const int p = r0 + 64 * i;
      const int prel = (p / TWh - HALO) * a.W + (p % TWh - HALO);
      vin[i] = ((inb >> i) & 1u) ? (unsigned)(((tile_pix + prel) * sr.cstride + sr.coff + c4 * 4) * EB) : OOB;
    }
    if (sr.scale) {
      rs_sc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sr.scale), 0, 0x7fffffff, 0x00020000);
      rs_sh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sr.shift), 0, 0x7fffffff, 0x00020000);
      lstep = 64;
    } else {
      rs_sc = __builtin_amdgcn_make_buffer_rsrc(c3d_x3_unit_affine, 0, 0x7fffffff, 0x00020000);
      rs_sh = __builtin_amdgcn_make_buffer_rsrc(c3d_x3_unit_affine + 16, 0, 0x7fffffff, 0x00020000);
      lstep = 0;
    }
    lslope = sr.lrelu ? a.slope : 1.f;
    lC = sr.C;
    lc0 = 0;
  };
  open_src(0);
  const int nchunks = a.Kq / 4;
  auto advance = [&]() {                   // input cursor -> next chunk; stays on the last one
    if (lk + 1 < nchunks) {
      ++lk;
      lc0 += 16;
      if (lc0 >= lC) open_src(++ls);
    }
  };
  int wq = 0;                              // weight cursor: (chunk, tap row) pairs in multiplication order
  const int nwq = nchunks * NG;

  // ---- registers in flight
  f32x4 pin[BFS ? 1 : IN_PT];              // raw input of the chunk after next (after its atoms ran: of the one after)
  u32x2 pinb[BFS ? IN_PT : 1];             // BFS: the same, four bf16 per unit
  u32x2 npl[IN_PT][NPL];                   // the next chunk's planes, waiting for the store phase
  u32x2 pw[W_PT][NPL];
  f32x4 pwf[F16 ? W_PT : 1];               // F16: the raw fp32 weight units in flight
  f32x4 psc, psh;
  float pslope;
  auto load_in = [&](int i) {
    if constexpr (BFS) pinb[i] = __builtin_amdgcn_raw_buffer_load_b64(rs_in, vin[i], lc0 * 2, 0);
    else pin[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, vin[i], lc0 * 4, 0));
  };
  auto load_aff = [&]() {
    const int so = (lstep >> 6) * lc0 * 4;
    psc = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_sc, vaff, so, 0));
    psh = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_sh, vaff, so, 0));
    if constexpr (F16) {                   // activations staged times 2^6 (LeakyReLU is positively homogeneous: exact)
      psc = psc * 64.f;
      psh = psh * 64.f;
    }
    pslope = lslope;
  };
  auto load_w = [&](int i) {               // weight unit i of pair wq (clamped to the last pair)
    const int q = min(wq, nwq - 1);
    const int chunk = q / NG, g = q % NG;
    const int so = (g * G * a.Kq + chunk * 4) * a.Cout * WU;
    if constexpr (F16) {
      pwf[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, vw[i], so, 0));
    } else {
#pragma unroll
      for (int p = 0; p < NPL; ++p) pw[i][p] = __builtin_amdgcn_raw_buffer_load_b64(rs_w, vw[i], so + p * wplane_b, 0);
    }
  };

  // ---- atoms of the NEXT chunk's input, two lists: LOAD (the units' raw values + scale / shift, issued early in the
  //      current chunk's first tap row) and CONV (per unit: XF e0, XF e1 -- affine, LeakyReLU, zero outside the image --
  //      then two SPLIT halves per plane), which run in the later tap rows, a tap row's time after the loads: raw values
  //      and planes never fill their registers together.  (Four taps as ONE tap row per chunk: 96 MFMAs are not enough
  //      lead for an HBM load -- 0.21 -> 0.26 ms on 64 -> 64 2x2 -- and reloading a unit right after its XF atoms, one
  //      chunk earlier, spills at 64 couts per workgroup.  Round 5: four taps run as two rows of two, G = 2.)
  static_assert(NG > 1, "the fused schedule needs more than one tap row per chunk");
  f32x4 sv;
  constexpr int LOAD_ATOMS = IN_PT + (PLAIN ? 0 : 1);            // + scale / shift
  constexpr int XFA = PLAIN ? 0 : 2;               // XF atoms per unit
  constexpr int CA_U = XFA + 2 * NPL;              // atoms per input unit: XF e0, XF e1, then two SPLIT halves per plane
  constexpr int CONV_ATOMS = IN_PT * CA_U;
  constexpr int WA_U = NPL + 1;                    // atoms per weight unit: one store per plane, then the next load
  constexpr int W_ATOMS = W_PT * WA_U;
  // two fp16 planes of two floats: H = RNE11(x), L = RNE11(x - H)
  auto split_f16 = [](float x0, float x1, float& r0, float& r1) {
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    f16x2_t h;
    h[0] = (_Float16)x0;
    h[1] = (_Float16)x1;
    r0 = x0 - (float)h[0];
    r1 = x1 - (float)h[1];
    return __builtin_bit_cast(unsigned, h);
  };
  auto load_atom = [&](auto k_tag) {
    constexpr int k = decltype(k_tag)::value;
    if constexpr (k == IN_PT) load_aff();
    else load_in(k);
  };
  auto conv_atom = [&](auto k_tag) {
    constexpr int k = decltype(k_tag)::value;
    constexpr int i = k / CA_U, r = k % CA_U;
    if constexpr (PLAIN && r == 0) sv = pin[i];      // the raw unit IS the staged value
    if constexpr (r < XFA) {
      // zero padding AFTER the transform, as a bit mask: written as `in ? f(v) : 0.f` the compiler built a branch per pair
      // of elements (s_and_saveexec / s_xor / s_andn2_saveexec / s_or around three VALU instructions each) in the
      // middle of the MFMA stream
      const unsigned keep = 0u - ((inb >> i) & 1u);
#pragma unroll
      for (int q = 2 * r; q < 2 * r + 2; ++q) {
        float x;
        if constexpr (BFS) x = __uint_as_float((q & 1) ? (pinb[i][q >> 1] & 0xffff0000u) : (pinb[i][q >> 1] << 16));
        else x = pin[i][q];
        const float v = __builtin_fmaf(x, psc[q], psh[q]);
        sv[q] = __uint_as_float(__float_as_uint(__builtin_fmaxf(v, v * pslope)) & keep);
      }
    } else {
      constexpr int p = (r - XFA) / 2, e = (r - XFA) % 2;
      if constexpr (F16) {
        float r0, r1;
        npl[i][p][e] = split_f16(sv[2 * e], sv[2 * e + 1], r0, r1);
        if constexpr (p == 0) {
          sv[2 * e] = r0;
          sv[2 * e + 1] = r1;
        }
        return;
      }
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
      bf16x2 h;
      h[0] = (__bf16)sv[2 * e];
      h[1] = (__bf16)sv[2 * e + 1];
      const unsigned pk = __builtin_bit_cast(unsigned, h);
      npl[i][p][e] = pk;
      if constexpr (p + 1 < NPL) {
        sv[2 * e] -= __uint_as_float(pk << 16);
        sv[2 * e + 1] -= __uint_as_float(pk & 0xffff0000u);
      }
    }
  };
  auto w_atom = [&](auto k_tag, unsigned short* s_w) {
    constexpr int k = decltype(k_tag)::value;
    constexpr int i = k / WA_U, r = k % WA_U;
    if constexpr (r < NPL) {
      if constexpr (F16 && r == 0) {           // weights staged times 2^10, split here (the pack carries bf16 planes only)
        const f32x4 wv = pwf[i] * 1024.f;
        float r0, r1, r2, r3, d0, d1;
        pw[i][0][0] = split_f16(wv[0], wv[1], r0, r1);
        pw[i][0][1] = split_f16(wv[2], wv[3], r2, r3);
        pw[i][1][0] = split_f16(r0, r1, d0, d1);
        pw[i][1][1] = split_f16(r2, r3, d0, d1);
      }
      *reinterpret_cast<u32x2*>(s_w + (r * WG_ROWS_P + wR0 + i * W_ROW_STEP) * 16 + swz_quad(wR0, wkq0)) = pw[i][r];
    } else {
      load_w(i);
    }
  };
  auto store_in = [&]() {
#pragma unroll
    for (int i = 0; i < IN_PT; ++i)
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        *reinterpret_cast<u32x2*>(s_in + (p * IN_ROWS_P + r0 + 64 * i) * 16 + swz_quad(r0, c4)) = npl[i][p];
  };

  const int nj = min(NPW, (a.Cout - n0 + 31) / 32);   // live 32-wide cout sub-tiles (ragged last tile)

  // one tap row (G taps) of the current chunk out of slab buffer `wb`, with the weight atoms of the next pair
  // (into the other buffer) and input atoms [IA0, IA1) of the next chunk dealt between the products
  auto group = [&](auto g_tag, int wb, auto nj_tag, auto last_tag) {
    constexpr int g = decltype(g_tag)::value;
    constexpr int NJ = decltype(nj_tag)::value;
    constexpr bool LAST = decltype(last_tag)::value;  // last chunk: there is no next chunk to load and convert
    // atoms of this tap row, in issue order: [LOAD (first tap row only)] [weights of the next pair] [CONV share]
    constexpr int NL = (g == 0 && !LAST) ? LOAD_ATOMS : 0;
    constexpr int CG = NG - 1;                        // tap rows that convert: all but the first
    constexpr int CA0 = (g == 0 || LAST) ? 0 : ((g - 1) * CONV_ATOMS) / CG;
    constexpr int CA1 = (g == 0 || LAST) ? 0 : (g * CONV_ATOMS) / CG;
    constexpr int NA = NL + W_ATOMS + (CA1 - CA0);
    constexpr int NS = G * NQ;                        // slots = products
    const unsigned short* s_w = s_w0 + wb * WBUF;
    unsigned short* d_w = s_w0 + (wb ^ 1) * WBUF;
    constexpr int NFB = NQ == 1 ? 2 : NPL;            // fragment register sets: one per plane; one plane: two, by tap parity
    bf16x8 ap[NFB][RPW];
    bf16x8 bq[NFB][NJ];
    // (no plane both closes a tap and opens the next one -- static_assert below -- so one register set per plane)
    static_assert(NQ == 1 || (PR::last_a(PR::pa(0)) != NQ - 1 && PR::last_b(PR::pb(0)) != NQ - 1),
                  "a tap must not open with the plane it closed with");
    // (`fresh` is zero, but opaque to the compiler and redefined per tap row: the 2 x 9 fragment row addresses are then
    //  recomputed next to their reads -- a few VALU in the MFMA shadow -- instead of being hoisted out of the chunk loop
    //  into 18 registers, which pushed the 64-cout dilation-2 kernel into scratch; a kernel with scratch gets ONE
    //  workgroup per CU on this GPU whatever its LDS and register budget says: matrix pipe busy 0.36 instead of 0.71)
    int fresh = 0;
    asm volatile("" : "+v"(fresh));
    // plane p of tap tg into register set p (one plane: set tg & 1)
    auto read_a = [&](int tg, int p) {
      const int t = g * G + tg;
      const int rs = NQ == 1 ? (tg & 1) : p;
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const int R = (wm + i * WM + HALO + a.dy[t]) * TWh + HALO + a.dx[t] + l31 + fresh;
        ap[rs][i] = *reinterpret_cast<const bf16x8*>(s_in + p * IN_ROWS_P * 16 + R * 16 + swz_half(R, half));
      }
    };
    auto read_b = [&](int tg, int p) {
      const int rs = NQ == 1 ? (tg & 1) : p;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int R = tg * TN + j * 32 + l31;
        bq[rs][j] = *reinterpret_cast<const bf16x8*>(s_w + p * WG_ROWS_P * 16 + R * 16 + swz_half(R, half));
      }
    };
    // first tap: all six fragments, in the order the products need them
    c3d_x3_static_for<0, NQ>([&](auto q_tag) {
      constexpr int q = decltype(q_tag)::value;
      bool first_a = true, first_b = true;
      for (int r = 0; r < q; ++r) {
        if (PR::pa(r) == PR::pa(q)) first_a = false;
        if (PR::pb(r) == PR::pb(q)) first_b = false;
      }
      if (first_a) read_a(0, PR::pa(q));
      if (first_b) read_b(0, PR::pb(q));
    });
    __builtin_amdgcn_sched_barrier(0);
    c3d_x3_static_for<0, NS>([&](auto s_tag) {
      constexpr int s = decltype(s_tag)::value, tg = s / NQ, q = s % NQ;
      constexpr int PA = NQ == 1 ? (tg & 1) : PR::pa(q), PB = NQ == 1 ? (tg & 1) : PR::pb(q);      // register sets
      // one plane: the next tap's fragments are requested BEFORE this tap's products (they land in the other register set)
      if constexpr (NQ == 1 && tg + 1 < G) {
        read_a(tg + 1, 0);
        read_b(tg + 1, 0);
      }
#pragma unroll
      for (int i = 0; i < RPW; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if constexpr (F16)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, ap[PA][i]), __builtin_bit_cast(f16x8_t, bq[PB][j]),
                                                               acc[i][j], 0, 0, 0);
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[PA][i], bq[PB][j], acc[i][j], 0, 0, 0);
        }
      // fragments of the next tap: each plane right after its last product of this tap
      if constexpr (NQ > 1 && tg + 1 < G) {
        if constexpr (PR::last_a(PA) == q) read_a(tg + 1, PA);
        if constexpr (PR::last_b(PB) == q) read_b(tg + 1, PB);
      }
      c3d_x3_static_for<(s * NA) / NS, ((s + 1) * NA) / NS>([&](auto k_tag) {
        constexpr int k = decltype(k_tag)::value;
        if constexpr (k < NL) load_atom(k_tag);
        else if constexpr (k < NL + W_ATOMS) w_atom(std::integral_constant<int, k - NL>{}, d_w);
        else conv_atom(std::integral_constant<int, CA0 + k - NL - W_ATOMS>{});
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ---- prologue: chunk 0 / pair 0 into LDS, pair 1 into registers; the input cursor then points at chunk 1
  c3d_x3_static_for<0, LOAD_ATOMS>([&](auto k) { load_atom(k); });
#pragma unroll
  for (int i = 0; i < W_PT; ++i) load_w(i);
  ++wq;
  c3d_x3_static_for<0, CONV_ATOMS>([&](auto k) { conv_atom(k); });
  c3d_x3_static_for<0, W_ATOMS>([&](auto k) { w_atom(k, s_w0); });   // pair 0 -> buffer 0; requests pair 1
  store_in();
  advance();
  ++wq;
  __syncthreads();

  bool mul_dma = false;
  auto k_loop = [&](auto nj_tag) {
    int wb = 0;
    for (int c = 0; c + 1 < nchunks; ++c) {
      c3d_x3_static_for<0, NG>([&](auto g_tag) {
        group(g_tag, wb, nj_tag, std::false_type{});   // multiplies tap row g of chunk c; stores pair +1 into the other slab
        ++wq;                          // buffer and requests pair +2; first row: requests chunk c+1, later rows: convert it
        wb ^= 1;
        __syncthreads();               // the other slab buffer is complete; everyone is done with this one
      });
      store_in();                      // chunk c+1's planes (the barrier above: every wave is done reading chunk c)
      advance();
      __syncthreads();
    }
    // last chunk: weight atoms only (round 3: its input atoms used to re-load and re-convert the last chunk -- half of
    // all staging work of a 32-channel layer)
    if constexpr (SM) mul_dma = conv_mul_dma_issue<TR, TN, 256>(a, smem, tid, x0, y0, n0, tile_pix);      // (round 6: under this chunk)
    c3d_x3_static_for<0, NG>([&](auto g_tag) {
      group(g_tag, wb, nj_tag, std::true_type{});
      ++wq;
      wb ^= 1;
      __syncthreads();
    });
  };
  if (NPW == 1 || nj >= NPW) k_loop(std::integral_constant<int, NPW>{});
  else k_loop(std::integral_constant<int, 1>{});
  conv_epilogue<TR, NT, WM, WN, NPL == 1, false, 256, false, (NPL >= 2 || SM)>(a, acc, smem, tid, lane, half, l31, wm, wn, b, x0, y0, n0, mt, ntile,
                                                                       tile_pix, mul_dma);
}

template <int NT, int HALO, int TT, bool SIX, int NPL = 3, bool BFS = false>
int launch_x3f_s(ConvArgs& a, hipStream_t st) {
  constexpr int G = c3d_x3f_group(TT);
  constexpr int IN_PT = ((8 + 2 * HALO) * (32 + 2 * HALO) * 4 + 255) / 256, W_PT = (G * 32 * NT * 4 + 255) / 256;
  size_t lds = (size_t)NPL * (IN_PT * 64 + 2 * W_PT * 64) * 16 * 2;      // rows padded to whole staging units
  const size_t red = (size_t)4 * 32 * NT * 2 * sizeof(float);   // statistics scratch of the epilogue
  if (lds < red) lds = red;
  a.ntn = (a.Cout + 32 * NT - 1) / (32 * NT);
  dim3 grid(a.B * a.tiles_x * a.tiles_y * a.ntn);
  if (NPL == 2) a.acc_scale = 1.f / 65536.f;           // operands staged times 2^6 and 2^10
  if constexpr (BFS) {
    if (a.stat_mul && a.stat_partial) {      // BatchNorm-backward sums in the epilogue: the instance that has it
      if (lds < (size_t)8 * 32 * (32 * NT + 8) * 2) lds = (size_t)8 * 32 * (32 * NT + 8) * 2;      // the multiplier tile of the epilogue
      if (!(a.variant & 64)) {      // round 6: a region of its own behind the K loop's buffers, filled by LDS-DMA under the last chunk
        lds = (lds + 15) / 16 * 16;
        a.mul_lds_off = (unsigned)lds;
        lds += (size_t)8 * 32 * (32 * NT) * 2;
      }
      a.lds_bytes = (unsigned)lds;
      c3d_opt_in_lds<&conv_x3f_kernel<NT, HALO, TT, SIX, NPL, BFS, true>>();
      hipLaunchKernelGGL((conv_x3f_kernel<NT, HALO, TT, SIX, NPL, BFS, true>), grid, dim3(256), lds, st, a);
      C3D_CHECK_LAUNCH();
      return 0;
    }
  }
  if constexpr (NPL == 3 && SIX && !BFS) {
    // input-gradient launches (six products) whose sources carry no on-load transform: the transform-free instance
    // (variant & 128 keeps the general one: bit-identity test)
    bool plain = !(a.variant & 128);
    for (int s = 0; s < a.nsrc; ++s) plain = plain && a.src[s].scale == nullptr && a.src[s].lrelu == 0;
    if (plain) {
      c3d_opt_in_lds<&conv_x3f_kernel<NT, HALO, TT, SIX, NPL, BFS, false, true>>();
      hipLaunchKernelGGL((conv_x3f_kernel<NT, HALO, TT, SIX, NPL, BFS, false, true>), grid, dim3(256), lds, st, a);
      C3D_CHECK_LAUNCH();
      return 0;
    }
  }
  c3d_opt_in_lds<&conv_x3f_kernel<NT, HALO, TT, SIX, NPL, BFS>>();
  hipLaunchKernelGGL((conv_x3f_kernel<NT, HALO, TT, SIX, NPL, BFS>), grid, dim3(256), lds, st, a);
  C3D_CHECK_LAUNCH();
  return 0;
}

template <int NT, int HALO, int TT, bool SIX>
int launch_x3_s(ConvArgs& a, hipStream_t st) {
  constexpr int G = (TT == 9) ? 3 : TT;
  size_t lds = (size_t)3 * ((size_t)(8 + 2 * HALO) * (32 + 2 * HALO) + (size_t)G * 32 * NT) * 16 * 2;
  const size_t red = (size_t)4 * 32 * NT * 2 * sizeof(float);   // statistics scratch of the epilogue
  if (lds < red) lds = red;
  c3d_opt_in_lds<&conv_x3_kernel<NT, HALO, TT, SIX>>();
  a.ntn = (a.Cout + 32 * NT - 1) / (32 * NT);
  dim3 grid(a.B * a.tiles_x * a.tiles_y * a.ntn);
  hipLaunchKernelGGL((conv_x3_kernel<NT, HALO, TT, SIX>), grid, dim3(256), lds, st, a);
  C3D_CHECK_LAUNCH();
  return 0;
}

template <int NT, int HALO, int TT>
int launch_x3(ConvArgs& a, hipStream_t st) {
  if constexpr (TT == 9) {
    // the fused kernel; c3d_conv_desc.variant & 4: round 2's phased kernel (the bit-identity test compares them)
    if (a.f16x2) return launch_x3f_s<NT, HALO, TT, true, 2>(a, st);           // EXPERIMENT: two fp16 planes, three products
    if (!(a.variant & 4)) return a.six ? launch_x3f_s<NT, HALO, TT, true>(a, st) : launch_x3f_s<NT, HALO, TT, false>(a, st);
  }
  if constexpr (TT == 4) {
    // Round 5: the fused schedule for the four-tap convs too, as TWO tap rows of two taps per chunk (round 3 tried one row
    // of four: no lead for the loads, 0.21 -> 0.26 ms, and left them on the phased kernel).  The next chunk's input is
    // requested in the first row and converted in the second, 48 MFMAs later.  Measured at 8 x 64 x 2048 / 32 x 1024 /
    // 16 x 512 / 8 x 256: 64 -> 64 0.268 -> 0.228 ms, 128 -> 128 0.194 -> 0.173, 256 -> 256 0.172 -> 0.163 and 0.069 -> 0.056,
    // input gradients alike; bit-identical (tools/bench_x3f4.py, tests/test_gpu_conv.py).  variant & 4: the phased kernel.
    if (!a.f16x2 && !(a.variant & 4)) return a.six ? launch_x3f_s<NT, HALO, TT, true>(a, st) : launch_x3f_s<NT, HALO, TT, false>(a, st);
  }
  return a.six ? launch_x3_s<NT, HALO, TT, true>(a, st) : launch_x3_s<NT, HALO, TT, false>(a, st);
}

template <int NT>
int launch_x3_taps(ConvArgs& a, int halo, hipStream_t st) {
  if (a.T == 6) return a.six ? launch_x3f_s<NT, 1, 6, true>(a, st) : launch_x3f_s<NT, 1, 6, false>(a, st);      // (halo 1: c3d_conv_forward)
  if (a.T == 3) return a.six ? launch_x3f_s<NT, 1, 3, true>(a, st) : launch_x3f_s<NT, 1, 3, false>(a, st);
  if (a.T == 4) return halo <= 1 ? launch_x3<NT, 1, 4>(a, st) : launch_x3<NT, 2, 4>(a, st);
  return halo <= 1 ? launch_x3<NT, 1, 9>(a, st) : launch_x3<NT, 2, 9>(a, st);
}

}  // namespace

// called by c3d_conv_forward for mfma_bf16 == 2, 8-row tiles, 4 or 9 taps; a.wpack must be a
// c3d_pack_weights(mode | 2) pack (fp32 image followed by the three bf16 planes)
int c3d_conv_forward_x3(ConvArgs& a, int halo, hipStream_t st) {
  if (a.one_plane) {      // the bf16 engine's nine-tap convs over bf16 tensors (c3d_conv_forward checked both)
    if (c3d_wide_cout_tiles(a)) return halo <= 1 ? launch_x3f_s<2, 1, 9, true, 1, true>(a, st) : launch_x3f_s<2, 2, 9, true, 1, true>(a, st);
    return halo <= 1 ? launch_x3f_s<1, 1, 9, true, 1, true>(a, st) : launch_x3f_s<1, 2, 9, true, 1, true>(a, st);
  }
  return c3d_wide_cout_tiles(a) ? launch_x3_taps<2>(a, halo, st) : launch_x3_taps<1>(a, halo, st);
}
