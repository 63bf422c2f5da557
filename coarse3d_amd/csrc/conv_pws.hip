// Narrow pointwise convolutions (1x1, Cout <= 64, K <= 96) of the exact-split engine as a STREAMING kernel.
//
// Round 6 (conv_bfp_kernel<8,{1,2},32,0,1,3> -- 26 launches, 2.3 ms of the step at 0.50 of
// the HBM peak and 0.10-0.26 matrix-pipe busy, bound by neither for three rounds).  These launches (the 32 / 64-channel
// 1x1 convs at full and half resolution and their input gradients: salsanext_proto.py:41-44 conv1 of ResContextBlock,
// :82-85 / :117-119 conv1 / conv5 of ResBlock, :203-205 conv4 of UpBlock, :318 the head) move 8-16 bytes per FLOP-pair of the
// matrix pipe: they are a memory stream with a small GEMM attached.  conv_bfp treats them as a GEMM: a workgroup stages a
// 256-pixel tile into three LDS planes, barriers, multiplies, stores -- phases in lockstep, two workgroups per CU, ~64 KB of
// loads in flight per CU.  Here every WAVE is its own stream:
//   * roles swapped: A = weights (rows = couts), B = pixels (columns).  A lane then owns ONE pixel: it loads that pixel's
//     channels from global memory straight into the B-fragment layout (8 consecutive channels = two 16-byte loads), applies
//     the BatchNorm affine / LeakyReLU, splits into the three bf16 planes in registers -- no LDS image of the activations, no
//     barrier in the loop -- and receives 4 x 4 CONSECUTIVE couts of its pixel in the accumulator: 16-byte stores (the
//     pixel-row layout of the other kernels gives 4-byte ones);
//   * the weights (<= 72 KB of bf16 planes) are split once per workgroup into LDS and read as A fragments;
//   * a wave walks the eight 32-pixel rows of one 8 x 32 tile with the NEXT row's loads in flight (two register sets of one
//     row: K registers), 8-12 waves per CU;
//   * statistics (sum, sum of squares / BatchNorm-backward sums with ConvArgs::stat_mul) accumulate per lane over the tile's
//     rows and are reduced across the 32 pixel lanes once per tile: the [C][2][ntile] partial layout of the other kernels.
// Same arithmetic as conv_bfp's NP = 3 kernel with six plane products (c3d_conv_desc.mfma_bf16 == 3; six is what every
// launch of these layers runs at the step's sizes), another summation order inside the statistics.
#include <type_traits>
#include "conv_x3_common.h"

int c3d_conv_forward_pws(ConvArgs& a, hipStream_t st);
bool c3d_conv_pws_takes(const ConvArgs& a);

namespace {

template <int NT, int KS, bool STATS>
__global__ __launch_bounds__(256, 2) void conv_pws_kernel(ConvArgs a) {
  constexpr int TN = 32 * NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short* s_w = reinterpret_cast<unsigned short*>(smem);                  // [KS][3][TN][16] bf16, rows swizzled
  float* s_aff = smem + (KS * 3 * TN * 16) / 2;                                   // [KS * 16][2]: scale, shift
  float* s_bias = s_aff + KS * 16 * 2;                                            // [TN]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;

  // ---- once per workgroup: weights -> three planes in LDS; the on-load affine of every input channel; the bias
  {
    const float* wp = a.wpack;                                                    // fp32 image [Kq][Cout][4]
    for (int u = tid; u < KS * 4 * TN; u += 256) {                                // unit: (k quad kq, cout n)
      const int n = u % TN, kq = u / TN;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n < a.Cout) v = *reinterpret_cast<const f32x4*>(wp + ((size_t)kq * a.Cout + n) * 4);
      u32x2 pl[3];
      split4x3(v, pl);
      const int s = kq >> 2, c4 = kq & 3;
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2*>(s_w + ((s * 3 + p) * TN + n) * 16 + swz_quad(n, c4)) = pl[p];
    }
    int k0 = 0;
    for (int si = 0; si < a.nsrc; ++si) {
      const c3d_src& sr = a.src[si];
      for (int c = tid; c < sr.C; c += 256) {
        s_aff[(k0 + c) * 2 + 0] = sr.scale ? sr.scale[c] : 1.f;
        s_aff[(k0 + c) * 2 + 1] = sr.scale ? sr.shift[c] : 0.f;
      }
      k0 += sr.C;
    }
    for (int n = tid; n < TN; n += 256) s_bias[n] = (a.bias && n < a.Cout) ? a.bias[n] : 0.f;
  }
  __syncthreads();

  // ---- per k-step: source pointer and LeakyReLU slope -- wave-uniform cursors, unrolled (SGPRs)
  const float* kptr[KS];
  int kcs[KS];
  float kslope[KS];
  {
    int si = 0, c0 = 0;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const c3d_src& sr = a.src[si];
      kptr[s] = sr.ptr + sr.coff + c0;
      kcs[s] = sr.cstride;
      kslope[s] = sr.lrelu ? a.slope : 1.f;
      c0 += 16;
      if (c0 >= sr.C && si + 1 < a.nsrc) {
        ++si;
        c0 = 0;
      }
    }
  }

  const int ntile = a.B * a.tiles_y * a.tiles_x;
  const int mt = c3d_xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;                 // one 8 x 32 tile per wave
  if (mt >= ntile) return;                                                        // (no barrier below)
  const int tx = mt % a.tiles_x;
  const int ty = (mt / a.tiles_x) % a.tiles_y;
  const int b = mt / (a.tiles_x * a.tiles_y);
  const int x = tx * 32 + l31, y0 = ty * 8;
  const bool xok = x < a.W;
  const int rows = min(8, a.H - y0);
  const size_t pix0 = (size_t)(b * a.H + y0) * a.W + x;                           // this lane's pixel in row 0 of the tile

  // two register sets of one row each: row r + 1 is requested BEFORE row r is multiplied (a row's worth of time + its epilogue
  // of lead; 2-5 waves per SIMD do the rest)
  f32x4 rawA[KS][2], rawB[KS][2];
  auto load_row = [&](f32x4 (&raw)[KS][2], int r) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float* p = kptr[s] + (pix0 + (size_t)r * a.W) * kcs[s] + 8 * half;
      raw[s][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      raw[s][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (xok) {
        raw[s][0] = *reinterpret_cast<const f32x4*>(p);
        raw[s][1] = *reinterpret_cast<const f32x4*>(p + 4);
      }
    }
  };

  constexpr int NS = STATS ? NT : 1;
  f32x4 s1[NS][4], s2[NS][4];
#pragma unroll
  for (int j = 0; j < NS; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) s1[j][g] = s2[j][g] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* mulp = STATS ? a.stat_mul : nullptr;

  auto process_row = [&](f32x4 (&raw)[KS][2], int r) {
    // (`fresh` is zero, but opaque and redefined per row: the weight fragments and the affine are then re-read from LDS next to
    //  their use instead of being hoisted out of the row loop into KS x (12 NT + 16) registers)
    int fresh = 0;
    asm volatile("" : "+v"(fresh));
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      // this lane's 8 channels of its pixel: affine, LeakyReLU, zero outside the image, three planes
      const float* ap = s_aff + (16 * s + 8 * half) * 2 + fresh;
      const f32x4 sc0 = *reinterpret_cast<const f32x4*>(ap);                     // (scale, shift) pairs
      const f32x4 sc1 = *reinterpret_cast<const f32x4*>(ap + 4);
      const f32x4 sc2 = *reinterpret_cast<const f32x4*>(ap + 8);
      const f32x4 sc3 = *reinterpret_cast<const f32x4*>(ap + 12);
      const float scl[8] = {sc0[0], sc0[2], sc1[0], sc1[2], sc2[0], sc2[2], sc3[0], sc3[2]};
      const float shf[8] = {sc0[1], sc0[3], sc1[1], sc1[3], sc2[1], sc2[3], sc3[1], sc3[3]};
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float t = __builtin_fmaf(raw[s][q >> 2][q & 3], scl[q], shf[q]);
        v[q] = xok ? __builtin_fmaxf(t, t * kslope[s]) : 0.f;
      }
      bf16x8 xp[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const __bf16 h = (__bf16)v[q];
          xp[p][q] = h;
          v[q] -= (float)h;
        }
      }
      // six plane products (activation plane, weight plane), smallest first: the order of conv_bfp / conv_x3
      constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
      bf16x8 wf[3][NT];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int n = 32 * j + l31;
          wf[p][j] = *reinterpret_cast<const bf16x8*>(s_w + ((s * 3 + p) * TN + n) * 16 + swz_half(n, half) + fresh);
        }
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[PB[q]][j], xp[PA[q]], acc[j], 0, 0, 0);
    }
    // ---- epilogue of the row: acc[j][4 g + e] = cout 32 j + 8 g + 4 half + e of pixel (y0 + r, x)
    if (xok) {
      const size_t pix = pix0 + (size_t)r * a.W;
      float* op = a.out + pix * a.out_cstride + a.out_coff;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = 32 * j + 8 * g + 4 * half;
          if (c < a.Cout) {
            const f32x4 bias4 = *reinterpret_cast<const f32x4*>(s_bias + c + fresh);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float t = acc[j][4 * g + e] + bias4[e];
              if (a.epi_lrelu) t = c3d_lrelu(t, a.slope);
              o[e] = t;
            }
            if (a.accumulate) {
              o = o + *reinterpret_cast<const f32x4*>(op + c);
              *reinterpret_cast<f32x4*>(op + c) = o;
            } else {
              __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(op + c));
            }
            if constexpr (STATS) {
              f32x4 m = o;
              if (mulp) m = *reinterpret_cast<const f32x4*>(mulp + pix * a.stat_mul_cs + c);
              s1[j][g] = s1[j][g] + o;
              s2[j][g] = s2[j][g] + o * m;
            }
          }
        }
    }
  };

  load_row(rawA, 0);
  for (int r = 0; r < rows; r += 2) {
    if (r + 1 < rows) load_row(rawB, r + 1);
    process_row(rawA, r);
    if (r + 1 >= rows) break;
    if (r + 2 < rows) load_row(rawA, r + 2);
    process_row(rawB, r + 1);
  }
  if constexpr (STATS) {
    // Reduce 16 NT (x 2) per-lane sums over the 32 pixel lanes of each half (lanes l and l + 32 hold different couts) by
    // butterfly WITH HALVING: at distance d a lane keeps the half of its values that its bit d selects and adds the partner's
    // copy of them -- 16 + 8 + 4 + 2 + 1 = 31 exchanges per 32 values instead of 5 x 32; after the last step each lane holds
    // the total of ONE value, the one its lane bits spell (`idx`).
    float v1[16 * NT], v2[16 * NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v1[16 * j + 4 * g + e] = s1[j][g][e];
          v2[16 * j + 4 * g + e] = s2[j][g][e];
        }
    int idx = 0;                                            // value index this lane ends up with
    auto halve = [&](auto n_tag, int d) {
      constexpr int n = decltype(n_tag)::value;
      const bool up = (l31 & d) != 0;
#pragma unroll
      for (int i = 0; i < n / 2; ++i) {
        const float k1 = up ? v1[i + n / 2] : v1[i], g1 = up ? v1[i] : v1[i + n / 2];
        const float k2 = up ? v2[i + n / 2] : v2[i], g2 = up ? v2[i] : v2[i + n / 2];
        v1[i] = k1 + __shfl_xor(g1, d, 64);
        v2[i] = k2 + __shfl_xor(g2, d, 64);
      }
      if (up) idx += n / 2;
    };
    if constexpr (NT == 2) {
      halve(std::integral_constant<int, 32>{}, 16);
      halve(std::integral_constant<int, 16>{}, 8);
      halve(std::integral_constant<int, 8>{}, 4);
      halve(std::integral_constant<int, 4>{}, 2);
      halve(std::integral_constant<int, 2>{}, 1);
    } else {
      halve(std::integral_constant<int, 16>{}, 16);
      halve(std::integral_constant<int, 8>{}, 8);
      halve(std::integral_constant<int, 4>{}, 4);
      halve(std::integral_constant<int, 2>{}, 2);
      v1[0] += __shfl_xor(v1[0], 1, 64);
      v2[0] += __shfl_xor(v2[0], 1, 64);
    }
    // value index i = 16 j + 4 g + e  ->  cout 32 j + 8 g + 4 half + e
    const int c = 32 * (idx >> 4) + 8 * ((idx >> 2) & 3) + 4 * half + (idx & 3);
    if ((NT == 2 || (l31 & 1) == 0) && c < a.Cout) {
      float* sp = a.stat_partial + (size_t)c * 2 * ntile + mt;                   // [C][2][ntile]
      sp[0] = v1[0];
      sp[ntile] = v2[0];
    }
  }
}

template <int NT, int KS, bool STATS>
int launch_pws_s(ConvArgs& a, hipStream_t st) {
  constexpr int TN = 32 * NT;
  const size_t lds = (size_t)KS * 3 * TN * 16 * 2 + ((size_t)KS * 16 * 2 + TN) * sizeof(float);
  c3d_opt_in_lds<&conv_pws_kernel<NT, KS, STATS>>();
  const int ntile = a.B * a.tiles_x * a.tiles_y;
  dim3 grid((ntile + 3) / 4);
  hipLaunchKernelGGL((conv_pws_kernel<NT, KS, STATS>), grid, dim3(256), lds, st, a);
  C3D_CHECK_LAUNCH();
  return 0;
}

}  // namespace

// What c3d_conv_forward hands over (it has checked: exact-split engine, six plane products, 8-row tiles, one tap).
// MEASURED (tools/bench_pws.py, profiles/round6_pws.md, one MI355X, 8 x 64 x 2048, ms staged -> streaming):
//   32 -> 32 plain                 0.054 -> 0.047 (5.0 -> 5.8 TB/s)      32 -> 20 (the head: ragged couts)   0.087 -> 0.048
//   32 -> 32 affine + statistics   0.056 -> 0.065                         32 -> 64                            0.077 -> 0.109
//   64 -> 64 accumulate            0.160 -> 0.239                         64 -> 32                            0.074 -> 0.089
//   3 x 32 -> 32 affine + stats    0.115 -> 0.128                         64 -> 64 + BatchNorm-backward sums  0.160 -> 0.208
// i.e. the stream wins where a lane's 128 bytes of one pixel are ONE cache line and nothing but the store follows (K = 32, no
// statistics: 4-5 waves per SIMD), and loses everywhere else: with 64 input channels a wave's load instruction touches 32
// lines for 32 bytes each (the staged kernel reads whole lines), and the per-lane statistics cost 32-64 registers = a wave
// per SIMD.  conv_bfp's staged tile already moves 4.6-5.5 TB/s on these layers in isolation.  So only that corner is
// dispatched here -- the 32-channel 1x1 convs without statistics and the class head, ~0.1 ms of the step -- and the other
// instances are not even built.
bool c3d_conv_pws_takes(const ConvArgs& a) {
  if (a.T != 1 || a.Kq != 8 || a.Cout > 32 || a.stat_partial) return false;
  // (a ragged last quad -- the 20 / 17 / 14-class head -- stores exact zeros into the pad couts of its 32-channel buffer)
  if (a.out_cstride % 4 || a.out_coff % 4 || (a.Cout + 3) / 4 * 4 + a.out_coff > a.out_cstride) return false;
  if (a.Cout % 4 && a.accumulate) return false;
  if (a.acc_scale_dev || a.acc_scale != 1.f || a.out_bf16) return false;
  for (int s = 0; s < a.nsrc; ++s)
    if (a.src[s].bf16 || a.src[s].cstride % 4 || a.src[s].coff % 4) return false;
  return true;
}

int c3d_conv_forward_pws(ConvArgs& a, hipStream_t st) { return launch_pws_s<1, 2, false>(a, st); }
