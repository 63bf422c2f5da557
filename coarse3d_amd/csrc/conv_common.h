// Pieces shared by the fp32-MFMA and the bf16-plane convolution kernels.
#pragma once
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "../../include/coarse3d_hip.h"

struct ConvArgs {
  c3d_src src[C3D_MAX_SRC];
  int nsrc;
  int B, H, W, Cout;
  int T;
  int dy[C3D_MAX_TAPS];
  int dx[C3D_MAX_TAPS];
  const float* wpack;
  const float* bias;
  int epi_lrelu;
  float* out;
  int out_cstride, out_coff, accumulate;
  float* stat_partial;
  int tiles_x, tiles_y, Kq;  // Kq = padded K / 4 (rows of the packed weight per tap)
  int ntn;                   // number of cout tiles
  float slope;               // LeakyReLU slope of the on-load and epilogue activations
  int out_bf16;              // out is bf16 (conv_bfp NP = 1 only)
  int six;                   // bf16x3: six plane products instead of eight (input-gradient convs, mfma_bf16 == 3)
  const float* stat_mul;     // NULL, or the tensor whose product with the stored values replaces v*v in stat_partial
  int stat_mul_cs;
  int stat_mul_bf16 = 0;     // stat_mul holds bf16 (the bf16 engine): the sums then use the values as stored (rounded to bf16)
  int variant = 0;         // c3d_conv_desc.variant (schedule selector of the bit-identity tests)
  bool one_plane = false;  // conv_x3.hip: the bf16 engine's fused nine-tap kernel (one plane, bf16 tensors)
  bool f16x2 = false;      // EXPERIMENT (mfma_bf16 == 4): two fp16 planes / three products where a kernel has the variant
  const float* acc_scale_dev = nullptr;   // times this device scalar, if any (per-tensor gradient exponent)
  unsigned mul_lds_off = 0;  // byte offset of a DEDICATED multiplier-tile region behind the K loop's LDS (0 = none): the kernels that have
                           // it fetch the tile with LDS-DMA loads under their last K chunk (conv_mul_dma_issue, round 6)
  unsigned lds_bytes = 0;  // dynamic LDS of THIS launch, set by the launchers whose kernels stage the epilogue's multiplier tile there
                           // (conv_epilogue only stages when the tile fits: 0 = never, the per-element loads take over)
  float acc_scale = 1.f;   // the accumulators are multiplied by this before bias / activation (1: fma(acc, 1, bias) == acc + bias);
                           // the f16x2 experiment stages its operands times 2^6 / 2^10 and hands 2^-16 back here
};

// 64-cout tiles (two 32-wide sub-tiles per workgroup) unless the grid would then cover too few CUs -- the 8 x 256 and
// 4 x 128 levels of the encoder have 64 / 32 pixel tiles per batch of 8 -- in which case 32-cout tiles double the
// workgroups.  Mirrored by ops._wide_cout_tiles().
inline bool c3d_wide_cout_tiles(const ConvArgs& a) {
  if (a.Cout <= 32) return false;
  constexpr int min_wg = 192;
  return a.B * a.tiles_x * a.tiles_y * ((a.Cout + 63) / 64) >= min_wg;
}

// bf16-plane kernels (conv_bfp.hip): planes = 1 (bf16) or 3 (bf16x3)
int c3d_conv_forward_bfp(ConvArgs& a, int planes, int tr, int halo, bool k32, hipStream_t st);
// second-generation bf16x3 engine for 8-row tiles with 4 or 9 taps (conv_x3.hip); needs a mode | 2 pack
int c3d_conv_forward_x3(ConvArgs& a, int halo, hipStream_t st);
// Winograd F(2x2, 3x3) on the exact-split engine (conv_wino.hip; c3d_conv_desc.variant & 16): nine taps on the 3 x 3 grid of
// dilation `dil` (1 or 2); a.wpack is a c3d_pack_weights_wino pack
int c3d_conv_forward_wino(ConvArgs& a, int dil, hipStream_t st);
// narrow pointwise convs of the exact-split engine as a streaming kernel (conv_pws.hip, round 6): Cout <= 64, K <= 192, six products
bool c3d_conv_pws_takes(const ConvArgs& a);
int c3d_conv_forward_pws(ConvArgs& a, hipStream_t st);
// wide pointwise engine on the bf16 pipe, 8-row tiles, Cout > 64 (conv_pw3.hip; planes = 1 or 3); needs a mode | 2 pack
int c3d_conv_forward_pw3(ConvArgs& a, int planes, bool wide, hipStream_t st);   // wide: 256-cout tiles, else 128

namespace {

template <typename T>
struct c3d_type_tag {
  using type = T;
};

// Round 6 (the bf16 engine's BatchNorm-backward sums in the input-gradient epilogue, ConvArgs::stat_mul over bf16 tensors): round 5
// copied the multiplier tile into LDS at the START of the epilogue -- 32 KB of loads per workgroup with nothing left to hide
// them under: the launches that carried the epilogue lost more (79 -> 130 us) than the separate reduce pass costs (24 us).
// LDS-DMA (global_load_lds_dwordx4: 1 KB per wave instruction straight into LDS, no registers) lets the tile be REQUESTED while
// the last K chunk is still being multiplied and merely waited for in the epilogue.  The region is the launch's own
// (ConvArgs::mul_lds_off), [TR * 32 pixels][TN channels] bf16, unpadded: a wave instruction covers 64 consecutive 16-byte units.
// Returns whether the tile was requested (workgroup-uniform).
template <int TR, int TN, int NTHR>
__device__ __forceinline__ bool conv_mul_dma_issue(const ConvArgs& a, float* smem, int tid, int x0, int y0, int n0, size_t tile_pix) {
  const bool ok = a.mul_lds_off != 0 && a.stat_mul != nullptr && a.stat_partial != nullptr && a.stat_mul_bf16 != 0 &&
                  x0 + 32 <= a.W && y0 + TR <= a.H && n0 + TN <= a.Cout && (a.stat_mul_cs & 7) == 0;
  if (!ok) return false;
  constexpr int UPP = TN / 8;                        // 16-byte units per pixel
  constexpr int UNITS = TR * 32 * UPP;
  static_assert(UNITS % NTHR == 0, "whole wave instructions");
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const unsigned short* mp = reinterpret_cast<const unsigned short*>(a.stat_mul);
  char* lbase = reinterpret_cast<char*>(smem) + a.mul_lds_off;
#pragma unroll
  for (int k = 0; k < UNITS / NTHR; ++k) {
    const int u0 = k * NTHR + wave * 64;             // wave-uniform: the LDS address of a DMA load is M0 + 16 * lane
    const int u = u0 + lane;
    const int p = u / UPP, cu = u % UPP;
    const unsigned short* g = mp + (tile_pix + (size_t)(p >> 5) * a.W + (p & 31)) * a.stat_mul_cs + n0 + cu * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(lbase + (size_t)u0 * 16), 16, 0, 0);
  }
  return true;
}

// Epilogue of one workgroup tile: bias, LeakyReLU, (accumulating) store, per-tile channel
// statistics [C][2][ntile].  acc[i][j] is the 32x32 MFMA accumulator of tile row wm + i*WM and
// cout tile wn*NPW + j (lane l: cout l&31, pixels (r&3) + 8*(r>>2) + 4*(l>>5)).
// ILV: cout sub-tile j of wave column wn is j*WN + wn (interleaved) instead of wn*NPW + j;
// NTHR: threads of the workgroup.  SUBFAST: the straight-line path is chosen per 32-wide cout
// sub-tile (a ragged last cout tile keeps it for its live sub-tiles); otherwise per workgroup tile
// (fewer code paths -- the 4-wave kernels sit at their register limits).
// STATMUL: the kernel honours ConvArgs::stat_mul (BatchNorm-backward sums in the epilogue).  The bf16x3 engine's kernels
// compile it in and, since round 5, the bf16 engine's kernels over bf16 tensors (conv_x3f one plane, conv_pw1, conv_bfp with
// raw bf16 staging): the 16 extra registers of the multiplier tile spill in the register-capped fp32 / widening bf16 kernels.
template <int TR, int NT, int WM, int WN, bool BF16_OUT = false, bool ILV = false, int NTHR = 256, bool SUBFAST = false,
          bool STATMUL = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[TR / WM][NT / WN], float* smem, int tid,
                                              int lane, int half, int l31, int wm, int wn, int b, int x0, int y0,
                                              int n0, int mt, int ntile, size_t tile_pix, bool mul_dma = false) {
  constexpr int RPW = TR / WM;
  constexpr int NPW = NT / WN;
  constexpr int TN = 32 * NT;
  float s1[NPW], s2v[NPW];
  const bool full_pix = (x0 + 32 <= a.W) && (y0 + TR <= a.H) && (SUBFAST || n0 + 32 * NT <= a.Cout);
  const int ocs = a.out_cstride;
  const size_t obase_i = tile_pix * a.out_cstride + a.out_coff + n0 + l31;   // element index, + per-lane cout
  const bool obf = BF16_OUT && a.out_bf16 != 0;      // engines that never store bf16 compile that path out
  const float* mulp = (STATMUL && a.stat_partial) ? a.stat_mul : nullptr;
  const bool mbf = STATMUL && BF16_OUT && a.stat_mul_bf16 != 0;      // bf16 multiplier tensor; the sums use the values as stored
  const float asc = a.acc_scale_dev ? a.acc_scale * *a.acc_scale_dev : a.acc_scale;
  // bf16 multiplier tile through LDS (round 5): read per accumulator element it is sixteen 2-byte loads per lane and sub-tile,
  // and the bf16 engine's kernels are bound by requests in flight -- the launches that carried the epilogue more than
  // doubled (conv_x3f<2,2,9,..> 79 -> 180 us).  The workgroup copies the tile's [TR * 32 pixels][TN channels] with 16-byte
  // loads into the (now dead) staging buffers and the lanes pick their elements from there.
  int MROW = TN + 8;                                 // bf16 per LDS row: the two half-waves of a read (4 rows apart) hit different banks
  unsigned short* s_mul = reinterpret_cast<unsigned short*>(smem);
  bool stage_mul = false;
  if constexpr (STATMUL && BF16_OUT) if (mul_dma) {
    // the tile was requested under the last K chunk (conv_mul_dma_issue): wait for it, nothing to copy
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    s_mul = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(smem) + a.mul_lds_off);
    MROW = TN;
    stage_mul = true;
  }
  if constexpr (STATMUL && BF16_OUT) if (!mul_dma) {
    // (the tile lands in the K loop's dynamic LDS: only if THIS launch allocated enough of it -- the launcher says how much)
    stage_mul = mulp != nullptr && mbf && full_pix && n0 + TN <= a.Cout && (a.stat_mul_cs & 7) == 0 &&
                a.lds_bytes >= (unsigned)(TR * 32 * MROW * 2);     // workgroup-uniform
    if (stage_mul) {
      __syncthreads();                               // every wave is done with the K loop's buffers
      constexpr int UPP = TN / 8;                    // 16-byte units per pixel
      typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
      for (int u = tid; u < TR * 32 * UPP; u += NTHR) {
        const int p = u / UPP, cu = u % UPP;
        const size_t g = (tile_pix + (size_t)(p >> 5) * a.W + (p & 31)) * a.stat_mul_cs + n0 + cu * 8;
        *reinterpret_cast<u32x4_t*>(s_mul + p * MROW + cu * 8) =
            *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const unsigned short*>(mulp) + g);
      }
      __syncthreads();
    }
  }
  // one 32-wide cout sub-tile, any position: per-element predicates
  auto slow_sub = [&](int j) {
    const int co = n0 + (ILV ? j * WN + wn : wn * NPW + j) * 32 + l31;
    const bool cok = co < a.Cout;
    const float bias = (a.bias && cok) ? a.bias[co] : 0.f;
    s1[j] = 0.f;
    s2v[j] = 0.f;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int gy = y0 + wm + i * WM;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gx = x0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = __builtin_fmaf(acc[i][j][r], asc, bias);
        if (a.epi_lrelu) v = c3d_lrelu(v, a.slope);
        if (cok && gy < a.H && gx < a.W) {
          const size_t o = ((size_t)(b * a.H + gy) * a.W + gx) * a.out_cstride + a.out_coff + co;
          if (a.accumulate) v += c3d_ld1(a.out, o, obf);
          c3d_st1(a.out, o, obf, v);
          if (mulp && obf) v = (float)(__bf16)v;
          s1[j] += v;
          s2v[j] += v * (mulp ? c3d_ld1(mulp, ((size_t)(b * a.H + gy) * a.W + gx) * a.stat_mul_cs + co, mbf) : v);
        }
        // (edge tiles only: keep the scheduler from batching four elements' loads and predicates -- registers, see fast_epilogue)
        if constexpr (STATMUL && BF16_OUT) if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // tiles whose pixels are all inside the image (the vast majority): straight-line code, no
  // per-element predicates, for every cout sub-tile that is completely live; a partially live
  // sub-tile (Cout % 32 != 0) takes slow_sub, dead ones (ragged last cout tile) are skipped.
  // OT = float or __bf16 (bf16 activation storage, values rounded RNE on store; the statistics
  // below use the fp32 values)
  auto fast_epilogue = [&](auto accumulate_tag, auto type_tag) {
    constexpr bool ACC = decltype(accumulate_tag)::value;
    using OT = typename decltype(type_tag)::type;
    OT* obase = reinterpret_cast<OT*>(a.out) + obase_i;
    if constexpr (std::is_same_v<OT, __bf16> && !ACC && !ILV && (NT / WN) > 1) if (!mulp || stage_mul) {
      // bf16 output, several ADJACENT cout sub-tiles per wave: a sub-tile is 64 bytes of a pixel, half a cache line, and a
      // launch that mostly writes (192 -> 704) took as long as over fp32 tensors.  With the sub-tiles innermost the wave's
      // consecutive stores cover NPW x 64 contiguous bytes of the same pixel.
      bool allfull = true;
#pragma unroll
      for (int j = 0; j < NPW; ++j)
        if (a.Cout - n0 - (wn * NPW + j) * 32 < 32) allfull = false;
      if (allfull) {
        float bj[NPW];
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
          bj[j] = a.bias ? a.bias[n0 + (wn * NPW + j) * 32 + l31] : 0.f;
          s1[j] = 0.f;
          s2v[j] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
          OT* orow = obase + (ptrdiff_t)((wm + i * WM) * a.W + 4 * half) * ocs + wn * NPW * 32;
          // (round 6: with the multiplier tile in LDS the BatchNorm-backward sums ride on this store order too -- the per-sub-tile
          //  order below writes 64-byte halves of a line: the 64-cout launches that carried the sums took 120 us instead of 78)
          const unsigned short* mr = s_mul + ((wm + i * WM) * 32 + 4 * half) * MROW + wn * NPW * 32 + l31;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int j = 0; j < NPW; ++j) {
              float v = __builtin_fmaf(acc[i][j][r], asc, bj[j]);
              if (a.epi_lrelu) v = c3d_lrelu(v, a.slope);
              __builtin_nontemporal_store((OT)v, &orow[(ptrdiff_t)((r & 3) + 8 * (r >> 2)) * ocs + j * 32]);
              if constexpr (STATMUL) {
                if (mulp) {
                  v = (float)(OT)v;                                      // the value as stored
                  s1[j] += v;
                  s2v[j] += v * __uint_as_float((unsigned)mr[((r & 3) + 8 * (r >> 2)) * MROW + j * 32] << 16);
                } else {
                  s1[j] += v;
                  s2v[j] += v * v;
                }
              } else {
                s1[j] += v;
                s2v[j] += v * v;
              }
            }
            if constexpr (STATMUL) if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
          }
        }
        return;
      }
    }
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      const int cl = (ILV ? j * WN + wn : wn * NPW + j) * 32;
      if constexpr (SUBFAST) {
        const int rem = a.Cout - n0 - cl;          // wave-uniform
        if (rem < 32) {
          s1[j] = 0.f;
          s2v[j] = 0.f;
          if (rem > 0) slow_sub(j);
          continue;
        }
      }
      const float bias = a.bias ? a.bias[n0 + cl + l31] : 0.f;
      s1[j] = 0.f;
      s2v[j] = 0.f;
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        OT* orow = obase + (ptrdiff_t)((wm + i * WM) * a.W + 4 * half) * ocs + cl;
        float old[16];
        if constexpr (ACC) {
#pragma unroll
          for (int r = 0; r < 16; ++r) old[r] = (float)orow[(ptrdiff_t)((r & 3) + 8 * (r >> 2)) * ocs];
        }
        // BatchNorm-backward sums (stat_mul): the multiplier tile is requested with the old values, one round trip
        float mul[STATMUL ? 16 : 1];
        if constexpr (STATMUL) if (mulp) {
          if (stage_mul) {
            const unsigned short* mr = s_mul + ((wm + i * WM) * 32 + 4 * half) * MROW + cl + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) mul[r] = __uint_as_float((unsigned)mr[((r & 3) + 8 * (r >> 2)) * MROW] << 16);
          } else {
            const size_t mbase = (tile_pix + (size_t)((wm + i * WM) * a.W + 4 * half)) * a.stat_mul_cs + n0 + cl + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) mul[r] = c3d_ld1(mulp, mbase + (size_t)((r & 3) + 8 * (r >> 2)) * a.stat_mul_cs, mbf);
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = __builtin_fmaf(acc[i][j][r], asc, bias);
          if (a.epi_lrelu) v = c3d_lrelu(v, a.slope);
          if constexpr (ACC) v += old[r];
          // fresh outputs leave with the nontemporal hint: the tile's halo / weight re-reads live on L2 hits and the
          // output stream is never re-read by this kernel (same-box A/B of the step: 197.9 -> 199.2 img/s)
          if constexpr (!ACC) __builtin_nontemporal_store((OT)v, &orow[(ptrdiff_t)((r & 3) + 8 * (r >> 2)) * ocs]);
          else orow[(ptrdiff_t)((r & 3) + 8 * (r >> 2)) * ocs] = (OT)v;
          if constexpr (STATMUL && std::is_same_v<OT, __bf16>) if (mulp) v = (float)(OT)v;     // the value as stored
          s1[j] += v;
          if constexpr (STATMUL) s2v[j] += v * (mulp ? mul[r] : v);
          else s2v[j] += v * v;
        }
        // (one (row pair, sub-tile) at a time: without the fence the scheduler hoists the multiplier reads of every sub-tile to the
        //  top of the epilogue -- 64 registers on top of the accumulators in the 64-cout kernels, which spilled 36-67 of them
        //  INTO THE K LOOP's allocation: tools/obj_resources.py, round 6)
        if constexpr (STATMUL && BF16_OUT) __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  if (full_pix && !obf) {
    if (!a.accumulate) fast_epilogue(std::false_type{}, c3d_type_tag<float>{});
    else fast_epilogue(std::true_type{}, c3d_type_tag<float>{});
  } else if (full_pix) {
    if constexpr (BF16_OUT) {
      if (!a.accumulate) fast_epilogue(std::false_type{}, c3d_type_tag<__bf16>{});
      else fast_epilogue(std::true_type{}, c3d_type_tag<__bf16>{});
    }
  } else {
#pragma unroll
    for (int j = 0; j < NPW; ++j) slow_sub(j);
  }
  if (a.stat_partial) {
    __syncthreads();
    float* red = smem;  // [WM][TN][2]
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      float t1 = s1[j] + __shfl_xor(s1[j], 32, 64);
      float t2 = s2v[j] + __shfl_xor(s2v[j], 32, 64);
      if (half == 0) {
        const int n = (ILV ? j * WN + wn : wn * NPW + j) * 32 + l31;
        red[(wm * TN + n) * 2 + 0] = t1;
        red[(wm * TN + n) * 2 + 1] = t2;
      }
    }
    __syncthreads();
    for (int n = tid; n < TN; n += NTHR) {
      if (n0 + n < a.Cout) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          t1 += red[(w * TN + n) * 2 + 0];
          t2 += red[(w * TN + n) * 2 + 1];
        }
        float* sp = a.stat_partial + (size_t)(n0 + n) * 2 * ntile + mt;   // [C][2][ntile]
        sp[0] = t1;
        sp[ntile] = t2;
      }
    }
  }
}

}  // namespace
