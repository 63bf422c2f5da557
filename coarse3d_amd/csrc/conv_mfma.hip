// Implicit-GEMM convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32) for gfx950.
//
// Replaces nn.Conv2d + LeakyReLU (+ the batch statistics of the BatchNorm2d that follows) of
// the SalsaNext blocks (reference pc_processor/models/salsanext_proto.py:41-62, 82-132,
// 164-208, 318 and projector.py:18-23).  The same kernel computes input gradients when given
// transposed weights and negated tap offsets.
//
// GEMM view: M = pixels (a TR x 32 spatial tile per workgroup), N = Cout (32*NT per
// workgroup), K = taps x Cin.  Per Cin chunk of CK channels the input tile (+halo) and the
// weight slab of every tap are staged in LDS; the BatchNorm affine of the PRODUCER layer is
// applied while staging (zero padding afterwards), so normalised tensors never exist in HBM.
// Staging is software-pipelined through registers: the global loads of chunk k+1 are issued
// before the MFMA loop of chunk k and written to LDS after it (one LDS buffer, loads in flight
// across the whole compute phase).  LDS rows are padded to CK+4 floats: ds_read_b128 /
// ds_write_b128 conflict-free.
//   A fragment: lane l reads 4 consecutive k of pixel (l&31), k-half (l>>5)
//   B fragment: lane l reads 4 consecutive k of cout  (l&31), same k-half
// so one b128 read per operand feeds 4 MFMAs.  Accumulator (32x32): lane holds cout l&31 for
// 16 pixels -> stores are 128 B contiguous per pixel.
// Grid: 1-D, cout tile fastest, XCD-remapped: the workgroups that share one input tile (and
// neighbouring tiles that share halos) run on the same XCD and hit its L2.
#include <type_traits>
#include <cstdlib>
#include "conv_common.h"

namespace {


template <int TR, int NT, int CK, int HALO, int TT>
__global__ __launch_bounds__(256, (NT >= 4 || (TT >= 6 && NT == 2 && TR == 8)) ? 2 : 3) void conv_mfma_kernel(ConvArgs a) {
  constexpr int CS = CK + 4;
  constexpr int TWh = 32 + 2 * HALO;
  constexpr int THh = TR + 2 * HALO;
  constexpr int TN = 32 * NT;
  constexpr int WM = (TR >= 4) ? 4 : TR;  // waves along tile rows
  constexpr int WN = 4 / WM;              // waves along cout tiles
  constexpr int RPW = TR / WM;
  constexpr int NPW = NT / WN;
  static_assert(NT % WN == 0, "NT must split across waves");
  constexpr int CQ = CK / 4;
  constexpr int IN_UNITS = THh * TWh * CQ;
  constexpr int IN_PT = (IN_UNITS + 255) / 256;
  constexpr int W_UNITS = TT * TN * CQ;
  constexpr int W_PT = (W_UNITS + 255) / 256;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;                    // [THh][TWh][CS]
  float* s_w = smem + THh * TWh * CS;    // [TT][TN][CS]

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

  // ---- register staging (prefetch) state
  f32x4 pin[IN_PT], pw[W_PT];
  f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  const int c4 = tid % CQ;               // 256 % CQ == 0: the channel quad of a thread is fixed
  // chunk-invariant staging indices, computed once: pixel offset of each unit relative to the
  // tile origin (32-bit), in-image mask, and the weight unit's (row, column) offset
  unsigned inb = 0;                      // bit i: unit i of this thread lies inside the image
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
  int wrel[W_PT];                        // (t*Kq + kq)*Cout + n  of each weight unit, or -1
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
  const size_t tile_pix = (size_t)(b * a.H + y0) * a.W + x0;   // wave-uniform

  auto load_chunk = [&](int s, int c0, int kbase) {
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
        f32x4 v = pin[i];
        if ((inb >> i) & 1u) {
          if (aff) v = v * psc + psh;
          if (lr) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = c3d_lrelu(v[q], a.slope);
          }
        }
        *reinterpret_cast<f32x4*>(s_in + (u / CQ) * CS + c4 * 4) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < W_PT; ++i) {
      const int u = tid + i * 256;
      if (u < W_UNITS) {
        const int n = u % TN;
        const int r = u / TN;
        const int kq = r % CQ, t = r / CQ;
        *reinterpret_cast<f32x4*>(s_w + (t * TN + n) * CS + kq * 4) = pw[i];
      }
    }
  };

  // 32-wide cout sub-tiles of this wave that exist: a ragged last tile (Cout = 400 = 12.5 x 32,
  // 80, 160, 288) skips the MFMAs of its dead sub-tiles instead of multiplying zero columns
  const int nj = min(NPW, (a.Cout - n0 - wn * NPW * 32 + 31) / 32);
  int s = 0, c0 = 0, kbase = 0;
  load_chunk(s, c0, kbase);
  while (true) {
    __syncthreads();      // previous chunk fully consumed
    store_chunk(s);
    __syncthreads();
    // issue the next chunk's global loads; they stay in flight during the MFMA loop
    int s2 = s, c2 = c0 + CK, kb2 = kbase;
    if (c2 >= a.src[s].C) {
      kb2 += a.src[s].C;
      s2 = s + 1;
      c2 = 0;
    }
    const bool more = s2 < a.nsrc;
    if (more) load_chunk(s2, c2, kb2);
    // ---- MFMA over taps x k (raised wave priority: the co-resident workgroup on this CU is
    //      usually in its staging phase and must not steal issue slots from the matrix pipe)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      const int tap_off = ((HALO + a.dy[t]) * TWh + (HALO + a.dx[t]) + l31) * CS + half * 4;
      const float* wb = s_w + (t * TN + wn * NPW * 32 + l31) * CS + half * 4;
#pragma unroll
      for (int kk = 0; kk < CK / 8; ++kk) {
        f32x4 av[RPW], bv[NPW];
#pragma unroll
        for (int i = 0; i < RPW; ++i)
          av[i] = *reinterpret_cast<const f32x4*>(s_in + (wm + i * WM) * TWh * CS + tap_off + kk * 8);
#pragma unroll
        for (int j = 0; j < NPW; ++j)
          bv[j] = *reinterpret_cast<const f32x4*>(wb + j * 32 * CS + kk * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int j = 0; j < NPW; ++j)
              if (j < nj) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][q], bv[j][q], acc[i][j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    if (!more) break;
    s = s2;
    c0 = c2;
    kbase = kb2;
  }

  conv_epilogue<TR, NT, WM, WN>(a, acc, smem, tid, lane, half, l31, wm, wn, b, x0, y0, n0, mt, ntile, tile_pix);
}

template <int TR, int NT, int CK, int HALO, int TT>
int launch_cfg(ConvArgs& a, hipStream_t st) {
  constexpr int CS = CK + 4;
  const size_t lds = ((size_t)(TR + 2 * HALO) * (32 + 2 * HALO) + (size_t)TT * 32 * NT) * CS * sizeof(float);
  c3d_opt_in_lds<&conv_mfma_kernel<TR, NT, CK, HALO, TT>>();
  a.ntn = (a.Cout + 32 * NT - 1) / (32 * NT);
  dim3 grid(a.B * a.tiles_x * a.tiles_y * a.ntn);
  hipLaunchKernelGGL((conv_mfma_kernel<TR, NT, CK, HALO, TT>), grid, dim3(256), lds, st, a);
  C3D_CHECK_LAUNCH();
  return 0;
}

// halo / tap-count dispatch: 1 tap (pointwise), 4 taps (2x2 dilated, halo 1), 9 taps (halo 1|2)
template <int TR, int NT>
int launch_taps(ConvArgs& a, int halo, hipStream_t st) {
  if (a.T == 1) return launch_cfg<TR, NT, 16, 0, 1>(a, st);
  if (a.T == 4) return halo <= 1 ? launch_cfg<TR, NT, 16, 1, 4>(a, st) : launch_cfg<TR, NT, 16, 2, 4>(a, st);
  if (a.T == 3) return launch_cfg<TR, NT, 16, 1, 3>(a, st);      // (3 / 6 taps: halo 1, c3d_conv_forward checked)
  if (a.T == 6) return launch_cfg<TR, NT, 16, 1, 6>(a, st);
  if (halo <= 1) return launch_cfg<TR, NT, 16, 1, 9>(a, st);
  return launch_cfg<TR, NT, 16, 2, 9>(a, st);
}

}  // namespace

// Tile rows: 8 wherever that pads H by at most a third, else the candidate (4, 2) that pads H the least, larger tiles on
// ties.  Round 3: the fused bf16x3 kernels (conv_x3f, conv_pw3f) exist for 8-row tiles only and run 1.5-5x faster per
// pixel than the narrow-tile kernels (the 256-channel 1x1 convs of the SemanticPOSS pyramid's 12-row level took 169 us on
// 4-row tiles -- 34 TF), which buys back 33 % padded rows at H = 12 and H = 6 many times over (rounds 1-2 picked 4 / 2
// there to save the padding).  A function of the image height only, so that every conv over the same [B,H,W] produces
// the same number of statistic partials.
static int c3d_tile_rows(int H) {
  if (((H + 7) / 8 * 8) * 3 <= H * 4) return 8;
  int best = 4, best_pad = (H + 3) / 4 * 4;
  if ((H + 1) / 2 * 2 < best_pad) best = 2;
  return best;
}

extern "C" int c3d_conv_num_mtiles(int B, int H, int W) {
  const int tr = c3d_tile_rows(H);
  return B * ((H + tr - 1) / tr) * ((W + 31) / 32);
}

// smallest grid the wide pointwise kernel is launched with (see c3d_conv_forward); mirrored by ops._pw3_tile()
static int c3d_pw3_min_workgroups() {
  static const int v = getenv("C3D_PW3_FILL") ? atoi(getenv("C3D_PW3_FILL")) : 128;      // (experiments)
  return v;
}

// Whether the kernel a descriptor selects compiles the BatchNorm-backward epilogue in (conv_common.h, STATMUL).  Mirrors the
// dispatch of c3d_conv_forward below and of conv_bfp.hip / conv_pw3.hip for the bf16 engine.
static bool c3d_stat_mul_kernel(const c3d_conv_desc* d) {
  if (d->mfma_bf16 >= 2) return !d->out_bf16 && !d->stat_mul_bf16;       // every kernel of the bf16x3 engine (4: the f16x2 experiment)
  const int tr = c3d_tile_rows(d->H);
  if (d->mfma_bf16 != 1 || !d->out_bf16 || !d->stat_mul_bf16 || tr != 8) return false;
  for (int s = 0; s < d->nsrc; ++s)
    if (!d->src[s].bf16) return false;
  if (d->ntaps == 9) return d->wpack_planes && !(d->variant & 4);                            // conv_x3f, one plane
  if (d->ntaps == 1 && d->Cout > 64 && d->wpack_planes) {
    const int px_tiles = d->B * ((d->W + 31) / 32) * ((d->H + tr - 1) / tr);
    bool wide = d->Cout > 128;
    if (wide && px_tiles * ((d->Cout + 255) / 256) < c3d_pw3_min_workgroups()) wide = false;
    if (wide) return false;                                                                  // conv_pw1<8>: no such instance
    if (px_tiles * ((d->Cout + 127) / 128) >= c3d_pw3_min_workgroups()) return (d->variant & 3) != 3;            // conv_pw1<4>
  }
  return !(d->variant & 8);                                                                  // conv_bfp, raw bf16 staging
}

extern "C" int c3d_conv_stat_mul_supported(const c3d_conv_desc* d) {
  if (!d || !d->stat_mul || !d->stat_partial || d->nsrc < 1 || d->nsrc > C3D_MAX_SRC || d->H <= 0) return 0;
  if (d->ntaps != 1 && d->ntaps != 3 && d->ntaps != 4 && d->ntaps != 6 && d->ntaps != 9) return 0;
  if ((d->ntaps == 3 || d->ntaps == 6) && d->mfma_bf16 < 2) return 0;      // (the generic kernels; bf16x3: every kernel has the epilogue)
  return c3d_stat_mul_kernel(d) ? 1 : 0;
}

extern "C" int c3d_conv_forward(const c3d_conv_desc* d, c3d_stream stream) {
  C3D_REQUIRE(d != nullptr, "conv: null descriptor");
  C3D_REQUIRE(d->nsrc >= 1 && d->nsrc <= C3D_MAX_SRC, "conv: nsrc must be 1..3");
  C3D_REQUIRE(d->wpack && d->out, "conv: wpack and out must not be null");
  C3D_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cout > 0, "conv: empty problem");
  // 3 and 6 taps (round 5): the column taps of RangeNet's transposed conv and the 3 x 2 taps of its stride-(1, 2) conv over
  // column-pair views (coarse3d_amd/rangenet.py) -- the generic kernels of every engine, offsets within +-1
  C3D_REQUIRE(d->ntaps == 1 || d->ntaps == 3 || d->ntaps == 4 || d->ntaps == 6 || d->ntaps == 9, "conv: ntaps must be 1, 3, 4, 6 or 9");
  const bool odd_taps = d->ntaps == 3 || d->ntaps == 6;
  ConvArgs a;
  int K = 0, halo = 0;
  for (int s = 0; s < d->nsrc; ++s) {
    C3D_REQUIRE(d->src[s].ptr != nullptr, "conv: null source pointer");
    C3D_REQUIRE((d->src[s].scale == nullptr) == (d->src[s].shift == nullptr), "conv: scale and shift come together");
    C3D_REQUIRE(d->src[s].C % 16 == 0 && d->src[s].C > 0, "conv: source channels must be a multiple of 16");
    C3D_REQUIRE(d->src[s].cstride % 4 == 0 && d->src[s].coff % 4 == 0, "conv: source stride/offset must be multiples of 4");
    a.src[s] = d->src[s];
    K += d->src[s].C;
  }
  for (int t = 0; t < d->ntaps; ++t) {
    a.dy[t] = d->tap_dy[t];
    a.dx[t] = d->tap_dx[t];
    int m = abs(d->tap_dy[t]) > abs(d->tap_dx[t]) ? abs(d->tap_dy[t]) : abs(d->tap_dx[t]);
    if (m > halo) halo = m;
  }
  C3D_REQUIRE(halo <= 2, "conv: tap offsets beyond +-2 are not supported");
  C3D_REQUIRE(d->ntaps != 1 || halo == 0, "conv: a single tap must have zero offset");
  C3D_REQUIRE(!odd_taps || (halo <= 1 && d->mfma_bf16 != 4), "conv: 3 / 6 taps: offsets within +-1, not in the f16x2 experiment");
  a.nsrc = d->nsrc;
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cout = d->Cout;
  a.T = d->ntaps;
  a.wpack = d->wpack; a.bias = d->bias; a.epi_lrelu = d->epi_lrelu;
  a.out = d->out; a.out_cstride = d->out_cstride; a.out_coff = d->out_coff;
  a.accumulate = d->accumulate;
  a.stat_partial = d->stat_partial;
  a.slope = c3d_slope_or_default(d->lrelu_slope);
  C3D_REQUIRE(a.slope <= 1.f, "conv: LeakyReLU slopes above 1 are not supported (the kernels evaluate max(v, slope * v))");
  a.out_bf16 = d->out_bf16;
  a.stat_mul = d->stat_mul;
  a.stat_mul_cs = d->stat_mul_cstride;
  a.acc_scale_dev = d->acc_scale_dev;
  a.variant = d->variant;
  C3D_REQUIRE(!d->acc_scale_dev || d->mfma_bf16 == 4, "conv: acc_scale_dev belongs to the f16x2 experiment (mfma_bf16 == 4)");
  a.stat_mul_bf16 = d->stat_mul_bf16;
  C3D_REQUIRE(!d->stat_mul || (d->stat_partial && d->stat_mul_cstride >= d->Cout && c3d_stat_mul_kernel(d)),
              "conv: stat_mul needs stat_partial, a channel stride >= Cout and a kernel with the epilogue (c3d_conv_stat_mul_supported: "
              "the bf16x3 engine over fp32 tensors, or the bf16 engine's 8-row-tile kernels over bf16 tensors)");
  {
    bool any_bf = d->out_bf16 != 0;
    for (int s = 0; s < d->nsrc; ++s) any_bf = any_bf || d->src[s].bf16 != 0;
    C3D_REQUIRE(!any_bf || d->mfma_bf16 == 1, "conv: bf16 activation storage needs mfma_bf16 == 1");
    C3D_REQUIRE(!d->out_bf16 || (d->out_cstride % 4 == 0 && d->out_coff % 4 == 0), "conv: bf16 out stride/offset must be multiples of 4");
  }
  const int tr = c3d_tile_rows(d->H);
  a.tiles_x = (d->W + 31) / 32;
  a.tiles_y = (d->H + tr - 1) / tr;
  a.Kq = K / 4;
  a.ntn = 1;
  hipStream_t st = (hipStream_t)stream;
  if (d->mfma_bf16) {      // opt-in precision modes on the bf16 matrix pipe (conv_bfp.hip)
    C3D_REQUIRE(d->mfma_bf16 >= 1 && d->mfma_bf16 <= 4, "conv: mfma_bf16 must be 0, 1, 2, 3 or 4");
    if (d->mfma_bf16 == 4) {     // EXPERIMENT: two fp16 planes, three products, generic kernel (forward convs; DESIGN.md round-4 list)
      bool k32f = tr == 8 && d->ntaps == 1;
      for (int s = 0; s < d->nsrc; ++s) k32f = k32f && (d->src[s].C % 32 == 0);
      if (tr == 8 && d->ntaps == 9 && d->wpack_planes) {     // the fused nine-tap kernel has the variant
        a.f16x2 = true;
        return c3d_conv_forward_x3(a, halo, st);
      }
      return c3d_conv_forward_bfp(a, 2, tr, halo, k32f, st);
    }
    bool k32 = true;
    for (int s = 0; s < d->nsrc; ++s) k32 = k32 && (d->src[s].C % 32 == 0);
    const bool x3 = d->mfma_bf16 >= 2;      // 3 = the exact-split engine with six plane products (input gradients)
    a.six = d->mfma_bf16 == 3;
    if (d->variant & 16) {
      // Winograd F(2x2, 3x3) (conv_wino.hip, round 6): the caller packed the weights with c3d_pack_weights_wino and sized
      // stat_partial with c3d_conv_wino_num_tiles
      C3D_REQUIRE(x3 && d->ntaps == 9 && (halo == 1 || halo == 2), "conv: the Winograd variant takes nine-tap convs of the bf16x3 engine");
      C3D_REQUIRE(d->Cout % 4 == 0 && d->out_cstride % 4 == 0 && d->out_coff % 4 == 0 && (!d->stat_mul || d->stat_mul_cstride % 4 == 0),
                  "conv (Winograd): output channels, stride and offset must be multiples of 4");
      for (int t = 0; t < 9; ++t)
        C3D_REQUIRE(d->tap_dy[t] % halo == 0 && d->tap_dx[t] % halo == 0, "conv (Winograd): the taps must be the 3 x 3 grid of one dilation");
      return c3d_conv_forward_wino(a, halo, st);
    }
    // (3 / 6 taps: the fused multi-tap kernel of the bf16x3 engine on 8-row tiles -- variant & 4: the generic kernel, as everywhere else)
    if (odd_taps && !(x3 && tr == 8 && d->wpack_planes && !(d->variant & 4))) return c3d_conv_forward_bfp(a, x3 ? 3 : 1, tr, halo, false, st);
    if (!x3 && tr == 8 && d->ntaps == 9 && d->wpack_planes && !(d->variant & 4)) {
      // bf16 engine, nine taps, every source a bf16 tensor: the fused kernel with one plane (conv_x3.hip); variant & 4 keeps
      // the phased conv_bfp kernel (bit-identity test)
      bool all_bf = true;
      for (int s = 0; s < d->nsrc; ++s) all_bf = all_bf && d->src[s].bf16 != 0;
      if (all_bf) {
        a.one_plane = true;
        return c3d_conv_forward_x3(a, halo, st);
      }
    }
    if (x3 && tr == 8 && d->ntaps > 1) {
      C3D_REQUIRE(d->wpack_planes, "conv: multi-tap bf16x3 convs need a c3d_pack_weights(mode | 2) pack (wpack_planes = 1)");
      return c3d_conv_forward_x3(a, halo, st);
    }
    if (tr == 8 && d->ntaps == 1 && d->Cout > 64 && d->wpack_planes) {
      // eight-wave workgroups of 256 x 256 (or 128) outputs -- unless that leaves most of the 256 CUs idle (the 8 x 256
      // and 4 x 128 levels of the encoder have 64 / 32 pixel tiles per batch): then narrower tiles, down to conv_bfp's
      // 64-wide four-wave workgroups
      static const int fill = c3d_pw3_min_workgroups();
      const int px_tiles = d->B * a.tiles_x * a.tiles_y;
      bool wide = d->Cout > 128;
      if (wide && px_tiles * ((d->Cout + 255) / 256) < fill) wide = false;
      if (wide || px_tiles * ((d->Cout + 127) / 128) >= fill) return c3d_conv_forward_pw3(a, x3 ? 3 : 1, wide, st);
    }
    // narrow 1x1 convs (Cout <= 64) of the exact-split engine with six plane products: the streaming kernel (conv_pws.hip, round 6);
    // variant & 32 keeps conv_bfp's staged tile (A/B runs, tests)
    if (x3 && a.six && tr == 8 && d->ntaps == 1 && !(d->variant & 32) && c3d_conv_pws_takes(a)) return c3d_conv_forward_pws(a, st);
    return c3d_conv_forward_bfp(a, x3 ? 3 : 1, tr, halo, k32, st);
  }
  if (tr == 8 && d->ntaps == 1) {
    // pointwise convs are plain GEMMs: deeper K chunk (32) and up to 128 output channels per
    // workgroup so that each barrier pair covers 128 MFMAs per wave
    bool k32 = true;
    for (int s = 0; s < d->nsrc; ++s) k32 = k32 && (d->src[s].C % 32 == 0);
    if (k32) {
      // 128-wide cout tiles unless 64-wide ones waste fewer padded columns (704 -> 11 x 64
      // instead of 6 x 128, 400 -> 7 x 64 instead of 4 x 128); measured equal MFMA efficiency
      const int pad128 = (d->Cout + 127) / 128 * 128, pad64 = (d->Cout + 63) / 64 * 64;
      if (d->Cout > 64 && pad128 <= pad64) return launch_cfg<8, 4, 32, 0, 1>(a, st);
      if (d->Cout > 32) return launch_cfg<8, 2, 32, 0, 1>(a, st);
      return launch_cfg<8, 1, 32, 0, 1>(a, st);
    }
  }
  const bool wide = d->Cout > 32;
  if (tr == 8) return wide ? launch_taps<8, 2>(a, halo, st) : launch_taps<8, 1>(a, halo, st);
  if (tr == 4) return wide ? launch_taps<4, 2>(a, halo, st) : launch_taps<4, 1>(a, halo, st);
  return launch_taps<2, 2>(a, halo, st);
}

// ------------------------------------------------------------------ weight repack
namespace {
// value of packed element i = (t, kq, n, j) -- shared by the single and the batched kernel
__device__ __forceinline__ float pack_value(const float* __restrict__ w, size_t i, int N, int K, int Cin, int T, int mode,
                                            int c_off, int Kpad) {
  const int j = i & 3;
  size_t r = i >> 2;
  const int n = r % N;
  r /= N;
  const int kq = r % (Kpad / 4);
  const int t = r / (Kpad / 4);
  const int k = kq * 4 + j;
  float v = 0.f;
  if (k < K) {
    if ((mode & 1) == 0) v = w[((size_t)n * Cin + c_off + k) * T + t];
    else v = w[((size_t)k * Cin + c_off + n) * T + t];
  }
  return v;
}

// mode & 2: behind the fp32 image, the exact 3-way bf16 split of every value (h = RNE8(v), m = RNE8(v - h),
// l = RNE8(v - h - m)) as three bf16 images of the same [t][kq][n][4] shape -- the weights are split ONCE
// per step here instead of by every workgroup that stages them (conv_x3.hip)
__device__ __forceinline__ void pack_store(float* __restrict__ dst, size_t i, size_t total, int mode, float v) {
  dst[i] = v;
  if (mode & 2) {
    __bf16* d = reinterpret_cast<__bf16*>(dst + total);
    float r = v;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const __bf16 h = (__bf16)r;
      d[(size_t)p * total + i] = h;
      r -= (float)h;
    }
  }
}

__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin,
                                    int T, int mode, int c_off, int c_cnt, int Kpad) {
  // dst[t][kq][n][j]
  const int N = (mode & 1) == 0 ? Cout : c_cnt;
  const int K = (mode & 1) == 0 ? c_cnt : Cout;
  const size_t total = (size_t)T * (Kpad / 4) * N * 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    pack_store(dst, i, total, mode, pack_value(w, i, N, K, Cin, T, mode, c_off, Kpad));
}
}  // namespace

extern "C" int c3d_pack_weights(const float* w_oihw, float* dst, int Cout, int Cin, int T, int mode,
                                int c_off, int c_cnt, int Kpad, c3d_stream stream) {
  C3D_REQUIRE(w_oihw && dst, "pack: null pointer");
  C3D_REQUIRE(Kpad % 16 == 0, "pack: Kpad must be a multiple of 16");
  C3D_REQUIRE(mode >= 0 && mode <= 3, "pack: mode must be 0..3");
  const int N = (mode & 1) == 0 ? Cout : c_cnt;
  const size_t total = (size_t)T * (Kpad / 4) * N * 4;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, dst, Cout,
                     Cin, T, mode, c_off, c_cnt, Kpad);
  C3D_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ batched repack
// One launch for every layer of the model: blockIdx.y selects the table entry.
namespace {
__global__ void pack_weights_batch_kernel(const c3d_pack_entry* __restrict__ table) {
  const c3d_pack_entry e = table[blockIdx.y];
  const int N = (e.mode & 1) == 0 ? e.Cout : e.c_cnt;
  const int K = (e.mode & 1) == 0 ? e.c_cnt : e.Cout;
  const size_t total = (size_t)e.T * (e.Kpad / 4) * N * 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    pack_store(e.dst, i, total, e.mode, pack_value(e.src, i, N, K, e.Cin, e.T, e.mode, e.c_off, e.Kpad));
}
}  // namespace

extern "C" int c3d_pack_weights_batch(const c3d_pack_entry* table_dev, int n, c3d_stream stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(pack_weights_batch_kernel, dim3(32, n), dim3(256), 0, (hipStream_t)stream, table_dev);
  C3D_CHECK_LAUNCH();
  return 0;
}
