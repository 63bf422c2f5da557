// 3x3 convolutions (dilation 1 or 2) as Winograd F(2x2, 3x3) on the exact-split bf16 engine ("bf16x3").
//
// Round 6 gate experiment -- MEASURED AND NOT TAKEN: opt-in only (c3d_conv_desc.variant & 16), see
// "Gate result" below.  The nine-tap kernels of conv_x3.hip sit at the matrix pipe's
// sustained rate: six bf16 MFMAs per fp32 multiply-add is what exactness costs, so the lever that is left is FEWER
// multiplies -- 16 instead of 36 per 2x2 output tile (2.25x).  What cuDNN does under the reference's nn.Conv2d
// (pc_processor/models/salsanext_proto.py:41-62, 82-132, 164-208) is the same re-association of fp32 arithmetic.
//
//   V = B^T d B   (input tile 4x4 -> 16 "frequencies"; additions only, in fp32, THEN the exact three-plane split)
//   U = G g G^T   (weights; once per step by c3d_pack_weights_wino: float64 arithmetic, one rounding to fp32, exact split)
//   M = sum_cin U (.) V   (16 independent [tiles x Cin] x [Cin x Cout] GEMMs on v_mfma_f32_32x32x16_bf16, six / eight plane
//                          products each, fp32 accumulate -- the arithmetic of conv_x3.hip)
//   Y = A^T M A   (after the K reduction, fp32)
// Dilation 2: the four (row parity, column parity) sub-grids of the image are independent dilation-1 problems.
//
// Workgroup: 4 x 32 output pixels of one sub-grid (2 x 16 Winograd tiles = one 32-row MFMA block) x 64 couts, 256 threads.
// Wave w owns frequency ROW w (four frequencies) for all tiles and couts: 4 x 2 accumulator tiles = 128 registers, every
// V fragment and every U fragment is read by exactly one wave -- V goes through LDS only to change layout (lane = (tile,
// channel quad) when it is produced, lane = (tile, 8 channels) when it is multiplied), U comes straight from the
// L2-resident pack into registers, one frequency ahead.  Per 16-channel chunk:
//   A  raw input tile (6 x 34 pixels) -> BatchNorm affine, LeakyReLU, zero padding -> fp32 LDS image (each element once)
//   B  thread = (tile, channel quad, upper / lower frequency rows): 12 LDS reads, 64 additions, 32 values split into
//      three planes, 24 ds_write_b64
//   C  48 MFMAs per wave
// with two barriers; two workgroups per CU (65-74 KB LDS each) overlap one's A / B with the other's C.
//
// Gate result (profiles/round6_wino_gate.md; tools/bench_wino.py; gate: <= 0.70x conv_x3f's time AND <= 4x its error on
// 64 -> 64 3x3 d1 at 8 x 64 x 2048, forward + input gradient):
//   time   0.374-0.389 ms against 0.347-0.364 (forward 1.07x, input gradient 1.09x; d2: 1.05x); 0.96-1.00x at 128 / 256
//          channels (32 x 1024, 16 x 512), 0.78x / 0.92x only at 256 channels on 8 x 256; 1.3-1.9x at 32 channels.  FAILS.
//   error  0.57x conv_x3f's rms error vs float64 (0.26-0.36x of its max): sixteen 64-term sums combined by a 9-term output
//          transform round LESS than one 576-term sum.  passes -- accuracy was never the problem.
//   why    ablations of this kernel (each switch a uniform branch, so only indicative): transform + split phases alone
//          0.17 ms, matrix phase with its U stream 0.26 (0.19 without the stream), epilogue + output traffic 0.11 -- three
//          co-equal costs, each about the whole budget of the gate (0.245 ms), that overlap only as far as two workgroups per
//          CU happen to interleave.  The arithmetic behind it: the input transform and the exact split cost ~130 VALU
//          operations per (tile, channel) whatever the schedule (32 per output pixel-channel against 12 in the direct
//          kernel) and buy 0.375 MFMA per (tile, channel) at 64 couts -- 5.3 VALU per MFMA, the edge of what hides in an MFMA's
//          shadow -- while U has no reuse inside a 32-tile workgroup (512 bytes per MFMA and wave from L2: 8.3 TB/s at
//          this kernel's rate, the CU's 64 B/clk L1 fill port at twice its rate), a 64-tile workgroup needs 98 KB of V per
//          16-channel chunk (no double buffer in 160 KB) and 128 couts per wave need 256 accumulator registers.  Every
//          way out trades one wall for another; none of them is a scheduling exercise on this kernel.
// Kept in the tree, tested (tests/test_gpu_conv.py), off: it is the measurement.
#include <type_traits>
#include "conv_x3_common.h"

int c3d_conv_forward_wino(ConvArgs& a, int dil, hipStream_t st);

namespace {

constexpr int WINO_SD_ROW = 2 * 17 * 16;                 // floats per input row of the fp32 image: [parity][17 columns][16 ch]
constexpr int WINO_SD_FLOATS = 6 * WINO_SD_ROW;          // 6 rows
constexpr int WINO_SV_BF16 = 16 * 3 * 32 * 16;           // [freq][plane][tile][16 ch]

// plane products per frequency in issue order (A plane, B plane), smallest first -- the sequences of conv_x3.hip
template <bool SIX>
struct wino_products {
  static constexpr int N = SIX ? 6 : 8;
  static constexpr int pa(int q) {
    constexpr int six[6] = {1, 2, 0, 1, 0, 0}, eight[8] = {2, 1, 2, 0, 1, 1, 0, 0};
    return SIX ? six[q] : eight[q];
  }
  static constexpr int pb(int q) {
    constexpr int six[6] = {1, 0, 2, 0, 1, 0}, eight[8] = {1, 2, 0, 2, 1, 0, 1, 0};
    return SIX ? six[q] : eight[q];
  }
};

template <int NT, int DIL, bool SIX>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(ConvArgs a) {
  constexpr int TN = 32 * NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_d = smem;
  unsigned short* s_V = reinterpret_cast<unsigned short*>(smem + WINO_SD_FLOATS);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;

  const int per_img = DIL * DIL * a.tiles_y * a.tiles_x;
  const int ntile = a.B * per_img;
  const int logical = c3d_xcd_remap(blockIdx.x, ntile * a.ntn);
  const int mt = logical / a.ntn;
  const int n0 = (logical % a.ntn) * TN;
  const int tx = mt % a.tiles_x;
  const int ty = (mt / a.tiles_x) % a.tiles_y;
  const int cls = (mt / (a.tiles_x * a.tiles_y)) % (DIL * DIL);
  const int b = mt / per_img;
  const int py = cls / DIL, px = cls % DIL;
  const int ys0 = ty * 4, xs0 = tx * 32;                  // sub-grid coordinates of the tile's first output pixel

  f32x16 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- phase A units: (pixel of the 6 x 34 input window, channel quad)
  unsigned inb = 0;
  int grel[4], soff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = tid + 256 * i;
    const int p = u >> 2, q = u & 3;
    const int row = p / 34, col = p % 34;
    const int ys = ys0 - 1 + row, xs = xs0 - 1 + col;
    const int y = py + DIL * ys, x = px + DIL * xs;
    grel[i] = 0;
    soff[i] = ((row * 2 + (col & 1)) * 17 + (col >> 1)) * 16 + q * 4;
    if (u < 6 * 34 * 4) {
      if (ys >= 0 && xs >= 0 && y < a.H && x < a.W) {
        inb |= 1u << i;
        grel[i] = (b * a.H + y) * a.W + x;
      }
    } else {
      soff[i] = -1;
    }
  }
  const int q4 = tid & 3;

  f32x4 raw[4];
  f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  bool paff = false, plr = false;
  auto load_raw = [&](int s, int c0) {
    const c3d_src& sr = a.src[s];
    const float* base = sr.ptr + sr.coff + c0 + q4 * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      raw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if ((inb >> i) & 1u) raw[i] = *reinterpret_cast<const f32x4*>(base + (size_t)grel[i] * sr.cstride);
    }
    paff = sr.scale != nullptr;
    plr = sr.lrelu != 0;
    if (paff) {
      psc = *reinterpret_cast<const f32x4*>(sr.scale + c0 + q4 * 4);
      psh = *reinterpret_cast<const f32x4*>(sr.shift + c0 + q4 * 4);
    }
  };
  auto phase_a = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (soff[i] >= 0) {
        f32x4 v = raw[i];
        if ((inb >> i) & 1u) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float t = v[q];
            if (paff) t = __builtin_fmaf(t, psc[q], psh[q]);
            if (plr) t = c3d_lrelu(t, a.slope);
            v[q] = t;
          }
        }
        *reinterpret_cast<f32x4*>(s_d + soff[i]) = v;
      }
    }
  };

  // ---- phase B: thread = (hh: frequency rows {0,1} / {2,3}; tile; channel quad)
  const int hh = wave >> 1;                                // wave-uniform
  const int tile = (tid & 127) >> 2;
  const int trr = tile >> 4, tj = tile & 15;
  auto phase_b = [&]() {
    // frequency row i = 2 hh + rr of B^T d is a combination of TWO window rows: 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
    // (one row pair at a time: 32 registers of d in flight instead of 48 -- the kernel sits at its register cap)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int i = 2 * hh + rr;                           // wave-uniform
      const int ra = 2 * trr + (i == 0 ? 0 : (i == 2 ? 2 : 1));
      const int rb = 2 * trr + (i == 3 ? 3 : (i == 2 ? 1 : 2));
      const float sgn = (i == 1) ? 1.f : -1.f;
      f32x4 T[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int col = ((k & 1) * 17 + tj + (k >> 1)) * 16 + q4 * 4;
        const f32x4 da = *reinterpret_cast<const f32x4*>(s_d + ra * WINO_SD_ROW + col);
        const f32x4 db = *reinterpret_cast<const f32x4*>(s_d + rb * WINO_SD_ROW + col);
#pragma unroll
        for (int q = 0; q < 4; ++q) T[k][q] = __builtin_fmaf(sgn, db[q], da[q]);       // da +- db, exact (one rounding, as an add)
      }
      f32x4 V[4];
      V[0] = T[0] - T[2];
      V[1] = T[1] + T[2];
      V[2] = T[2] - T[1];
      V[3] = T[1] - T[3];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int f = i * 4 + jj;
        u32x2 pl[3];
        split4x3(V[jj], pl);
#pragma unroll
        for (int p = 0; p < 3; ++p)
          *reinterpret_cast<u32x2*>(s_V + ((f * 3 + p) * 32 + tile) * 16 + swz_quad(tile, q4)) = pl[p];
      }
    }
  };

  // ---- phase C: wave = frequency row; U fragments straight from the pack, one frequency ahead
  const int Npad = a.ntn * TN;
  const unsigned short* U = reinterpret_cast<const unsigned short*>(a.wpack);
  bf16x8 bfr[2][3][NT];
  auto load_b = [&](int buf, int chunk, int jf) {
    const int f = wave * 4 + jf;
    const unsigned short* ub = U + ((size_t)(chunk * 16 + f) * 3 * Npad + n0 + l31) * 16 + half * 8;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int j = 0; j < NT; ++j) bfr[buf][p][j] = *reinterpret_cast<const bf16x8*>(ub + ((size_t)p * Npad + j * 32) * 16);
  };
  using PR = wino_products<SIX>;
  auto mfma_freq = [&](int buf, int jf) {
    const int f = wave * 4 + jf;
    bf16x8 ap[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) ap[p] = *reinterpret_cast<const bf16x8*>(s_V + ((f * 3 + p) * 32 + l31) * 16 + swz_half(l31, half));
#pragma unroll
    for (int q = 0; q < PR::N; ++q)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[jf][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[PR::pa(q)], bfr[buf][PR::pb(q)][j], acc[jf][j], 0, 0, 0);
  };

  // ---- K loop
  int s = 0, c0 = 0, kbase = 0, chunk = 0;
  load_raw(s, c0);
  load_b(0, 0, 0);
  phase_a();
  __syncthreads();
  phase_b();
  __syncthreads();
  while (true) {
    int s2 = s, c2 = c0 + 16, kb2 = kbase;
    if (c2 >= a.src[s].C) {
      kb2 += a.src[s].C;
      s2 = s + 1;
      c2 = 0;
    }
    const bool more = s2 < a.nsrc;
    if (more) load_raw(s2, c2);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int jf = 0; jf < 4; ++jf) {
      if (jf < 3) load_b((jf + 1) & 1, chunk, jf + 1);
      else if (more) load_b(0, chunk + 1, 0);
      mfma_freq(jf & 1, jf);
    }
    __builtin_amdgcn_s_setprio(0);
    if (!more) break;
    phase_a();                                             // (the fp32 image is not read by phase C)
    __syncthreads();                                       // every wave is done with this chunk's V; the image is complete
    phase_b();
    __syncthreads();
    s = s2;
    c0 = c2;
    kbase = kb2;
    ++chunk;
  }

  // ---- output transform.  Right factor (over the wave's four frequencies) in registers, left factor through LDS.
  __syncthreads();
  float* s_Y = smem;                                       // [freq row 4][x parity 2][tile 32][TN]
  float* s_red = smem + 4 * 2 * 32 * TN;                   // [16][TN][2]
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
      const float p0 = acc[0][j][r] + acc[1][j][r] + acc[2][j][r];
      const float p1 = acc[1][j][r] - acc[2][j][r] - acc[3][j][r];
      s_Y[((wave * 2 + 0) * 32 + m) * TN + j * 32 + l31] = p0;
      s_Y[((wave * 2 + 1) * 32 + m) * TN + j * 32 + l31] = p1;
    }
  __syncthreads();
  constexpr int CQN = TN / 4;                              // cout quads
  constexpr int UNITS = 32 * 2 * CQN;
  static_assert(256 % CQN == 0 && UNITS % 256 == 0, "a thread keeps its cout quad");
  const int cq = tid % CQN;
  const int nq = n0 + cq * 4;
  const bool cok = nq < a.Cout;                            // (Cout is a multiple of 4 in every caller: checked by the launcher)
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (a.bias && cok) bias4 = *reinterpret_cast<const f32x4*>(a.bias + nq);
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2v = {0.f, 0.f, 0.f, 0.f};
  const float* mulp = a.stat_partial ? a.stat_mul : nullptr;
#pragma unroll
  for (int k = 0; k < UNITS / 256; ++k) {
    const int u = tid + 256 * k;
    const int rest = u / CQN;
    const int xp = rest & 1, m = rest >> 1;
    f32x4 P[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) P[i] = *reinterpret_cast<const f32x4*>(s_Y + ((i * 2 + xp) * 32 + m) * TN + cq * 4);
    const int xs = xs0 + 2 * (m & 15) + xp;
    const int x = px + DIL * xs;
#pragma unroll
    for (int yp = 0; yp < 2; ++yp) {
      const f32x4 Yv = yp == 0 ? (P[0] + P[1] + P[2]) : (P[1] - P[2] - P[3]);
      const int ys = ys0 + 2 * (m >> 4) + yp;
      const int y = py + DIL * ys;
      if (cok && y < a.H && x < a.W) {
        const size_t pix = (size_t)(b * a.H + y) * a.W + x;
        float* op = a.out + pix * a.out_cstride + a.out_coff + nq;
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float t = Yv[q] + bias4[q];
          if (a.epi_lrelu) t = c3d_lrelu(t, a.slope);
          v[q] = t;
        }
        if (a.accumulate) v = v + *reinterpret_cast<const f32x4*>(op);
        if (a.accumulate) *reinterpret_cast<f32x4*>(op) = v;
        else __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(op));
        if (a.stat_partial) {
          f32x4 mul = v;
          if (mulp) mul = *reinterpret_cast<const f32x4*>(mulp + pix * a.stat_mul_cs + nq);
          s1 = s1 + v;
          s2v = s2v + v * mul;
        }
      }
    }
  }
  if (a.stat_partial) {
    const int g = tid / CQN;                               // 256 / CQN groups
    constexpr int NG = 256 / CQN;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      s_red[(g * TN + cq * 4 + q) * 2 + 0] = s1[q];
      s_red[(g * TN + cq * 4 + q) * 2 + 1] = s2v[q];
    }
    __syncthreads();
    for (int n = tid; n < TN; n += 256) {
      if (n0 + n < a.Cout) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int gg = 0; gg < NG; ++gg) {
          t1 += s_red[(gg * TN + n) * 2 + 0];
          t2 += s_red[(gg * TN + n) * 2 + 1];
        }
        float* sp = a.stat_partial + (size_t)(n0 + n) * 2 * ntile + mt;   // [C][2][ntile]
        sp[0] = t1;
        sp[ntile] = t2;
      }
    }
  }
}

template <int NT, int DIL>
int launch_wino(ConvArgs& a, hipStream_t st) {
  constexpr int TN = 32 * NT;
  size_t lds = (size_t)WINO_SD_FLOATS * 4 + (size_t)WINO_SV_BF16 * 2;
  const size_t epi = ((size_t)4 * 2 * 32 * TN + (size_t)(256 / (TN / 4)) * TN * 2) * 4;
  if (lds < epi) lds = epi;
  const int Hs = (a.H + DIL - 1) / DIL, Ws = (a.W + DIL - 1) / DIL;
  a.tiles_x = (Ws + 31) / 32;
  a.tiles_y = (Hs + 3) / 4;
  a.ntn = (a.Cout + TN - 1) / TN;
  dim3 grid(a.B * DIL * DIL * a.tiles_x * a.tiles_y * a.ntn);
  if (a.six) {
    c3d_opt_in_lds<&conv_wino_kernel<NT, DIL, true>>();
    hipLaunchKernelGGL((conv_wino_kernel<NT, DIL, true>), grid, dim3(256), lds, st, a);
  } else {
    c3d_opt_in_lds<&conv_wino_kernel<NT, DIL, false>>();
    hipLaunchKernelGGL((conv_wino_kernel<NT, DIL, false>), grid, dim3(256), lds, st, a);
  }
  C3D_CHECK_LAUNCH();
  return 0;
}

// ---- weight transform: U = G g G^T from the fp32 image of a c3d_pack_weights pack ([tap][Kq][N][4]); float64 arithmetic,
//      one rounding to fp32, exact three-plane split.  dst: bf16 [chunk][freq 16][plane 3][Npad][16 ch].
struct WinoPackArgs {
  const float* pack;
  unsigned short* dst;
  int Kq, N, Npad;
  int tmap[9];          // tap index of kernel position (ky, kx), or -1
};

__global__ void wino_pack_kernel(WinoPackArgs w) {
  const size_t total = (size_t)(w.Kq / 4) * 16 * w.Npad * 16;
  const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = i & 15;
    size_t r = i >> 4;
    const int n = r % w.Npad;
    r /= w.Npad;
    const int f = r & 15;
    const int chunk = r >> 4;
    const int fi = f >> 2, fj = f & 3;
    double u = 0.0;
    if (n < w.N) {
      const int ch = chunk * 16 + k;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int t = w.tmap[ky * 3 + kx];
          if (t >= 0) u += G[fi][ky] * G[fj][kx] * (double)w.pack[(((size_t)t * w.Kq + (ch >> 2)) * w.N + n) * 4 + (ch & 3)];
        }
    }
    float rem = (float)u;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const __bf16 h = (__bf16)rem;
      w.dst[((((size_t)chunk * 16 + f) * 3 + p) * w.Npad + n) * 16 + k] = __builtin_bit_cast(unsigned short, h);
      rem -= (float)h;
    }
  }
}

}  // namespace

int c3d_conv_forward_wino(ConvArgs& a, int dil, hipStream_t st) {
  if (dil == 1) return launch_wino<2, 1>(a, st);
  return launch_wino<2, 2>(a, st);
}

extern "C" int c3d_conv_wino_num_tiles(int B, int H, int W, int dil) {
  if (dil != 1 && dil != 2) return 0;
  const int Hs = (H + dil - 1) / dil, Ws = (W + dil - 1) / dil;
  return B * dil * dil * ((Hs + 3) / 4) * ((Ws + 31) / 32);
}

extern "C" int64_t c3d_wino_pack_bytes(int Kpad, int N) {
  if (Kpad <= 0 || N <= 0 || Kpad % 16) return 0;
  const int64_t Npad = (N + 63) / 64 * 64;
  return (int64_t)(Kpad / 16) * 16 * 3 * Npad * 16 * 2;
}

extern "C" int c3d_pack_weights_wino(const float* pack_f32, int Kpad, int N, const int32_t* tap_dy, const int32_t* tap_dx, int dil,
                                     void* dst, c3d_stream stream) {
  C3D_REQUIRE(pack_f32 && dst && tap_dy && tap_dx, "pack_wino: null pointer");
  C3D_REQUIRE(Kpad > 0 && Kpad % 16 == 0 && N > 0, "pack_wino: Kpad must be a positive multiple of 16");
  C3D_REQUIRE(dil == 1 || dil == 2, "pack_wino: dilation 1 or 2");
  WinoPackArgs w;
  w.pack = pack_f32;
  w.dst = static_cast<unsigned short*>(dst);
  w.Kq = Kpad / 4;
  w.N = N;
  w.Npad = (N + 63) / 64 * 64;
  for (int i = 0; i < 9; ++i) w.tmap[i] = -1;
  for (int t = 0; t < 9; ++t) {
    C3D_REQUIRE(tap_dy[t] % dil == 0 && tap_dx[t] % dil == 0, "pack_wino: tap offsets must be multiples of the dilation");
    const int ky = tap_dy[t] / dil + 1, kx = tap_dx[t] / dil + 1;
    C3D_REQUIRE(ky >= 0 && ky < 3 && kx >= 0 && kx < 3 && w.tmap[ky * 3 + kx] < 0, "pack_wino: the nine taps must be the 3 x 3 grid");
    w.tmap[ky * 3 + kx] = t;
  }
  const size_t total = (size_t)(Kpad / 16) * 16 * w.Npad * 16;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wino_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w);
  C3D_CHECK_LAUNCH();
  return 0;
}
