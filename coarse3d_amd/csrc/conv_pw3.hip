// Wide pointwise (1x1) convolution in fp32-class arithmetic on the bf16 matrix pipe ("bf16x3":
// c3d_conv_desc.mfma_bf16 == 2, 8-row tiles, one tap, Cout > 64, plane-carrying weight pack).
//
// Why a separate kernel.  The exact 3-way bf16 split of the bf16x3 engine costs ~6 VALU instructions per staged
// element.  (Round 2 believed that VALU and MFMA time simply add up on gfx950 and therefore only tried to pay the split
// fewer times; round 3's probe -- profiles/round3_coissue_probe.md -- shows that plain VALU hides under the MFMAs and
// that PACKED f32 VALU, which hipcc had made of the staging code, does not: see conv_pw3f_kernel below, which
// the bf16x3 mode runs.  This phased kernel stays for the "bf16" mode and as the bit-identity reference.)
// conv_bfp.hip's NP = 3 kernel splits a
// 256-pixel input tile once per 64 output channels (11 times for the 704 -> 704 projector GEMM)
// and splits the weights in every workgroup as well.  This kernel
//   * computes 256 pixels x 256 (NT = 8) or 128 (NT = 4) output channels per workgroup: the input
//     tile is split once per 256 / 128 couts -- 4x / 2x fewer split instructions per MFMA;
//   * takes the weights pre-split (c3d_pack_weights(mode | 2) appends the three bf16 planes to
//     the fp32 pack once per step): global -> LDS copies, no VALU work;
//   * runs 8 waves (4 pixel-row groups x 2 cout groups, 2 x NT/2 MFMA tiles each) on a
//     double-buffered LDS image (K chunk = 16 channels = one MFMA K step, 3 planes x (256 + 32*NT)
//     rows x 32 B per buffer, unpadded swizzled rows as in conv_x3.hip): one barrier per chunk, the
//     global loads of chunk c+2 are in flight while chunk c is multiplied;
//   * deals the 32-wide cout sub-tiles to the two cout wave groups alternately, so a ragged last
//     tile (704 = 256 + 256 + 192) keeps both groups equally busy and issues no MFMA on dead
//     sub-tiles.
// Arithmetic (NP = 3): SIX of the nine plane products -- h*h, h*m, m*h, m*m, h*l, l*h; each product is then
// off by up to 2^-23 |a||b|.  The multi-tap forward convolutions need eight (six or seven measured 4-5x
// the fp32 engine's gradient noise through the network's BatchNorm renormalisations, conv_x3.hip);
// for the layers this kernel serves -- wide 1x1 convs, the projector, the prototype similarity, and
// every input gradient -- six were measured to change nothing: the backbone noise test, the per-layer
// float64 gradient check, the bit-exact index tests and the whole GPU suite pass on this engine
// unchanged (DESIGN.md "Where the plane products matter").  On-load BatchNorm affine and epilogue as
// conv_x3.hip (reference: pc_processor/models/salsanext_proto.py:41-62, 164-208, projector.py:18-23).
#include "conv_x3_common.h"

namespace {

template <int I, int N, class F>
__device__ __forceinline__ void c3d_pw_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    c3d_pw_static_for<I + 1, N>(f);
  }
}


// NP = 3: bf16x3 (exact split, eight plane products).  NP = 1: "bf16" mode -- operands rounded to
// bf16 (the h plane of the pack IS the RNE-rounded weight), one product, fp32 or bf16 tensors.
template <int NT, int NP>
__global__ __launch_bounds__(512, 1) void conv_pw3_kernel(ConvArgs a) {
  constexpr int TR = 8, CQ = 4;                    // 16 channels per K chunk
  constexpr int TN = 32 * NT;
  constexpr int WM = 4, WN = 2, RPW = 2, NPW = NT / WN;
  constexpr int IN_ROWS = TR * 32;
  constexpr int IN_PT = IN_ROWS * CQ / 512;        // 2
  constexpr int W_PT = TN * CQ / 512;              // 2 (NT = 8) or 1 (NT = 4)
  constexpr int BUF = NP * (IN_ROWS + TN) * 16;    // bf16 elements per LDS buffer
  static_assert(IN_ROWS * CQ % 512 == 0 && TN * CQ % 512 == 0, "staging units must tile the workgroup");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short* s_base = reinterpret_cast<unsigned short*>(smem);   // 2 x { [3][IN_ROWS][16], [3][TN][16] }

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

  // ---- staging state.  A load cursor walks the K chunks of the concatenated sources; everything
  //      the loads need lives in registers as ready-made per-thread pointers that advance by one
  //      chunk per step (a.src[s] is indexed only when the cursor enters a source): the first
  //      version re-derived its addresses from a.src[s] per chunk and spent 0.44 of 1.96 ms of the
  //      704x704 layer issuing eight loads per chunk.  Every load is unconditional -- pixels beyond
  //      the image and couts beyond Cout are CLAMPED to valid ones (their products only reach
  //      outputs the epilogue masks) -- so the waits are counted (s_waitcnt vmcnt(N)), not vmcnt(0).
  //      (Tried on top: the input tile two chunks ahead in a second register set -- no change,
  //      1.79 ms either way on the 704x704 layer; weight fragments of sub-tile j+1 read ahead of the
  //      MFMAs of sub-tile j -- kept, neutral.  Phase ablation of that layer: matrix phase 1.1 ms,
  //      staging 0.6 ms, epilogue + loop 0.24 ms, and the three simply add up.  In the one-plane mode:
  //      two / four chunks per barrier -- 0.71 -> 0.64 ms on that layer, 303.9 -> 304.7 img/s for the
  //      step, not kept.)
  f32x4 pin[IN_PT];
  u32x2 pw[W_PT][NP];
  f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  bool paff = false, plr = false;          // on-load transform of the chunk held in pin
  const int c4 = tid % CQ;                 // the channel quad of a thread is fixed
  const size_t tile_pix = (size_t)(b * a.H + y0) * a.W + x0;
  int pixrel[IN_PT];
#pragma unroll
  for (int i = 0; i < IN_PT; ++i) {
    const int p = tid / CQ + i * (512 / CQ);
    const int gx = min(x0 + (p & 31), a.W - 1), gy = min(y0 + (p >> 5), a.H - 1);
    pixrel[i] = (gy - y0) * a.W + (gx - x0);
  }
  const size_t wplane = (size_t)a.Kq * a.Cout * 4;                             // bf16 elements per weight plane (T = 1)
  const unsigned short* lw[W_PT];          // plane 0 of this thread's weight units at the cursor's K offset
#pragma unroll
  for (int i = 0; i < W_PT; ++i) {
    const int u = tid + i * 512;
    const int n = min(n0 + u % TN, a.Cout - 1), kq = u / TN;
    lw[i] = reinterpret_cast<const unsigned short*>(a.wpack + wplane) + (size_t)(kq * a.Cout + n) * 4;   // behind the fp32 pack
  }
  const char* lin[IN_PT];                  // byte pointers: a source is fp32 or (NP = 1 only) bf16
  const float *lsc = nullptr, *lsh = nullptr;
  int ls = 0, lc0 = 0, lC = 0;
  bool llr = false, lbf = false;
  auto open_src = [&](int s) {
    const c3d_src& sr = a.src[s];
    lbf = NP == 1 && sr.bf16 != 0;
#pragma unroll
    for (int i = 0; i < IN_PT; ++i)
      lin[i] = reinterpret_cast<const char*>(sr.ptr) + ((tile_pix + pixrel[i]) * sr.cstride + sr.coff + c4 * 4) * (lbf ? 2 : 4);
    lsc = sr.scale ? sr.scale + c4 * 4 : nullptr;
    lsh = sr.scale ? sr.shift + c4 * 4 : nullptr;
    llr = sr.lrelu != 0;
    lC = sr.C;
    lc0 = 0;
  };
  open_src(0);
  auto load_chunk = [&]() {                // loads the cursor's chunk, then moves the cursor on
    if (lbf) {
#pragma unroll
      for (int i = 0; i < IN_PT; ++i) {
        const c3d_u32x2 r = *reinterpret_cast<const c3d_u32x2*>(lin[i]);
        pin[i] = f32x4{__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16),
                       __uint_as_float(r[1] & 0xffff0000u)};
      }
    } else {
#pragma unroll
      for (int i = 0; i < IN_PT; ++i) pin[i] = *reinterpret_cast<const f32x4*>(lin[i]);
    }
    paff = lsc != nullptr;
    plr = llr;
    if (paff) {
      psc = *reinterpret_cast<const f32x4*>(lsc);
      psh = *reinterpret_cast<const f32x4*>(lsh);
    }
#pragma unroll
    for (int i = 0; i < W_PT; ++i)
#pragma unroll
      for (int p = 0; p < NP; ++p) pw[i][p] = *reinterpret_cast<const u32x2*>(lw[i] + p * wplane);
#pragma unroll
    for (int i = 0; i < W_PT; ++i) lw[i] += (size_t)a.Cout * 16;
    lc0 += 16;
    if (lc0 >= lC) {
      if (++ls < a.nsrc) open_src(ls);
    } else {
#pragma unroll
      for (int i = 0; i < IN_PT; ++i) lin[i] += lbf ? 32 : 64;
      if (paff) {
        lsc += 16;
        lsh += 16;
      }
    }
  };
  auto store_chunk = [&](int buf) {
    unsigned short* s_in = s_base + buf * BUF;
    unsigned short* s_w = s_in + NP * IN_ROWS * 16;
#pragma unroll
    for (int i = 0; i < IN_PT; ++i) {
      f32x4 v = pin[i];
      if (paff) v = v * psc + psh;
      if (plr) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = c3d_lrelu(v[q], a.slope);
      }
      u32x2 pl[NP];
      split4_planes<NP>(v, pl);
      const int R = tid / CQ + i * (512 / CQ);
#pragma unroll
      for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(s_in + (p * IN_ROWS + R) * 16 + swz_quad(R, c4)) = pl[p];
    }
#pragma unroll
    for (int i = 0; i < W_PT; ++i) {
      const int u = tid + i * 512;
      const int R = u % TN, kq = u / TN;
#pragma unroll
      for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(s_w + (p * TN + R) * 16 + swz_quad(R, kq)) = pw[i][p];
    }
  };

  // live 32-wide cout sub-tiles of this tile, dealt alternately to the two cout wave groups
  const int live = min(NT, (a.Cout - n0 + 31) / 32);
  const int nj = (live - wn + WN - 1) / WN;
  auto mfma_chunk = [&](int buf, auto nj_tag) {
    constexpr int NJ = decltype(nj_tag)::value;
    const unsigned short* s_in = s_base + buf * BUF;
    const unsigned short* s_w = s_in + NP * IN_ROWS * 16;
    bf16x8 ap[NP][RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int R = (wm + i * WM) * 32 + l31;
      const int o = R * 16 + swz_half(R, half);
#pragma unroll
      for (int p = 0; p < NP; ++p) ap[p][i] = *reinterpret_cast<const bf16x8*>(s_in + p * IN_ROWS * 16 + o);
    }
    // the weight fragments of sub-tile j+1 are read while the 16 MFMAs of sub-tile j issue
    bf16x8 bp[2][NP];
    auto load_b = [&](int j) {
      const int R = (j * WN + wn) * 32 + l31;
      const int o = R * 16 + swz_half(R, half);
#pragma unroll
      for (int p = 0; p < NP; ++p) bp[j & 1][p] = *reinterpret_cast<const bf16x8*>(s_w + p * TN * 16 + o);
    };
    if (NJ > 0) load_b(0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (j + 1 < NJ) load_b(j + 1);
      __builtin_amdgcn_sched_barrier(0);
      // six of the nine plane products, smallest first (see the file header)
#define C3D_PLANE(PA, PB) \
  _Pragma("unroll") for (int i = 0; i < RPW; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[PA][i], bp[j & 1][PB], acc[i][j], 0, 0, 0);
      if constexpr (NP == 3) {
        C3D_PLANE(2, 0) C3D_PLANE(0, 2) C3D_PLANE(1, 1) C3D_PLANE(1, 0) C3D_PLANE(0, 1)
      }
      C3D_PLANE(0, 0)
#undef C3D_PLANE
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  const int nchunks = a.Kq / 4;            // K / 16
  load_chunk();
  store_chunk(0);
  if (nchunks > 1) load_chunk();
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int cur = c & 1;
    if (c + 1 < nchunks) {
      store_chunk(cur ^ 1);                // chunk c+1, loaded while chunk c-1 was multiplied
      if (c + 2 < nchunks) load_chunk();
    }
    if (nj >= NPW) mfma_chunk(cur, std::integral_constant<int, NPW>{});
    else if (NPW > 3 && nj == 3) mfma_chunk(cur, std::integral_constant<int, (NPW > 3 ? 3 : 1)>{});
    else if (NPW > 2 && nj == 2) mfma_chunk(cur, std::integral_constant<int, (NPW > 2 ? 2 : 1)>{});
    else if (nj == 1) mfma_chunk(cur, std::integral_constant<int, 1>{});
    __syncthreads();                       // the other buffer is complete, this one is free again
  }
  conv_epilogue<TR, NT, WM, WN, NP == 1, true, 512, true, NP == 3>(a, acc, smem, tid, lane, half, l31, wm, wn, b, x0, y0, n0, mt, ntile,
                                                  tile_pix);
}

// ---------------------------------------------------------------------------------------------------
// Round 4, the `bf16` engine over bf16 tensors (BASELINE configs[2]): conv_pw3_kernel<NT, 1>'s tile, LDS image, rounding
// points and accumulation order (outputs bit-identical, tests/test_gpu_bf16_storage.py) with FOUR K chunks in flight.
// One plane makes a chunk 8 MFMAs per wave (0.1 us) against 8 KB of input and 8 KB of weights per workgroup, and there is
// one workgroup per CU: with one chunk in flight every chunk cost a memory latency (192 -> 704 at 8 x 32 x 1024: 12 chunks and
// 26 us per workgroup, 320 us for 470 MB).  The raw units are cheap to keep (8 bytes: two registers), so chunks c + 1 ...
// c + 4 are in flight while chunk c is multiplied: a ring of four register sets by chunk index, every load a buffer load
// whose offset goes out of range past the last chunk (no traffic, no condition around a load: the waits stay counted), the
// trip count rounded up to a multiple of four (a chunk of zeros adds nothing).  The on-load transform reads scale / shift /
// LeakyReLU slope of its four channels from a table in LDS (one formula for every source: fma(x, scale, shift), then
// max(v, v * slope) with slope 1 where there is no activation -- the values the phased kernel computes).
// SM (round 5): the instance with the BatchNorm-backward epilogue (ConvArgs::stat_mul), launched when a launch asks for it
template <int NT, bool SM = false>
__global__ __launch_bounds__(512, 1) void conv_pw1_kernel(ConvArgs a) {
  constexpr int TR = 8, CQ = 4;                    // 16 channels per K chunk
  constexpr int TN = 32 * NT;
  constexpr int WM = 4, WN = 2, RPW = 2, NPW = NT / WN;
  constexpr int IN_ROWS = TR * 32;
  constexpr int IN_PT = IN_ROWS * CQ / 512;        // 2
  constexpr int W_PT = TN * CQ / 512;              // 2 (NT = 8) or 1 (NT = 4)
  constexpr int BUF = (IN_ROWS + TN) * 16;         // bf16 elements per LDS buffer
  constexpr int DEPTH = 4;                         // chunks in flight (register sets)

  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short* s_base = reinterpret_cast<unsigned short*>(smem);   // 2 x { [IN_ROWS][16], [TN][16] }, then the affine table
  const int K = a.Kq * 4;
  float* s_aff = reinterpret_cast<float*>(s_base + 2 * BUF);          // [3][K]: scale, shift, slope per input channel

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

  // ---- the affine table
  {
    int k0 = 0;
    for (int s = 0; s < a.nsrc; ++s) {
      const c3d_src& sr = a.src[s];
      for (int k = tid; k < sr.C; k += 512) {
        s_aff[k0 + k] = sr.scale ? sr.scale[k] : 1.f;
        s_aff[K + k0 + k] = sr.scale ? sr.shift[k] : 0.f;
        s_aff[2 * K + k0 + k] = sr.lrelu ? a.slope : 1.f;
      }
      k0 += sr.C;
    }
  }

  // ---- load cursor: buffer loads, one 32-bit offset per unit (pixels beyond the image and couts beyond Cout are CLAMPED
  //      to valid ones as in the phased kernel: their products only reach outputs the epilogue masks)
  const int c4 = tid % CQ;
  const size_t tile_pix = (size_t)(b * a.H + y0) * a.W + x0;
  int pixrel[IN_PT];
#pragma unroll
  for (int i = 0; i < IN_PT; ++i) {
    const int p = tid / CQ + i * (512 / CQ);
    const int gx = min(x0 + (p & 31), a.W - 1), gy = min(y0 + (p >> 5), a.H - 1);
    pixrel[i] = (gy - y0) * a.W + (gx - x0);
  }
  const unsigned wplane_b = (unsigned)a.Kq * a.Cout * 8;                        // bytes of a bf16 weight plane (T = 1)
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.wpack) + (size_t)a.Kq * a.Cout * 4, 0, wplane_b, 0x00020000);     // plane 0, behind the fp32 image
  unsigned vw[W_PT];
#pragma unroll
  for (int i = 0; i < W_PT; ++i) {
    const int u = tid + i * 512;
    const int n = min(n0 + u % TN, a.Cout - 1), kq = u / TN;
    vw[i] = (unsigned)((kq * a.Cout + n) * 8);
  }
  __amdgpu_buffer_rsrc_t rs_in;
  unsigned vin[IN_PT];
  int ls = 0, lc0 = 0, lC = 0, lk = 0;
  unsigned lpast = 0;                      // all ones once the cursor is past the last chunk: loads then read nothing
  const unsigned img_bytes_per_c = (unsigned)a.B * a.H * a.W * 2;
  auto open_src = [&](int s) {
    const c3d_src& sr = a.src[s];
    rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sr.ptr), 0, img_bytes_per_c * sr.cstride, 0x00020000);
#pragma unroll
    for (int i = 0; i < IN_PT; ++i) vin[i] = (unsigned)(((tile_pix + pixrel[i]) * sr.cstride + sr.coff + c4 * 4) * 2);
    lC = sr.C;
    lc0 = 0;
  };
  open_src(0);
  const int nchunks = a.Kq / 4;            // K / 16
  u32x2 pin[DEPTH][IN_PT], pw[DEPTH][W_PT];
  auto load_chunk = [&](auto slot_tag) __attribute__((always_inline)) {      // the cursor's chunk into set `slot`, then the cursor moves on
    constexpr int S = decltype(slot_tag)::value;
#pragma unroll
    for (int i = 0; i < IN_PT; ++i) pin[S][i] = __builtin_amdgcn_raw_buffer_load_b64(rs_in, vin[i] | lpast, lc0 * 2, 0);
#pragma unroll
    for (int i = 0; i < W_PT; ++i) pw[S][i] = __builtin_amdgcn_raw_buffer_load_b64(rs_w, vw[i] | lpast, lk * a.Cout * 32, 0);
    if (lk + 1 < nchunks) {
      ++lk;
      lc0 += 16;
      if (lc0 >= lC) open_src(++ls);
    } else {
      lpast = 0xffffffffu;
    }
  };
  int sk = 0;                              // first channel of the chunk the next store_chunk writes
  auto store_chunk = [&](int buf, auto slot_tag) __attribute__((always_inline)) {
    constexpr int S = decltype(slot_tag)::value;
    unsigned short* s_in = s_base + buf * BUF;
    unsigned short* s_w = s_in + IN_ROWS * 16;
    const int ka = min(sk, K - 16) + c4 * 4;                 // (chunks of the rounded-up trip count: any valid row of the table)
    const f32x4 sc = *reinterpret_cast<const f32x4*>(s_aff + ka);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(s_aff + K + ka);
    const f32x4 sl = *reinterpret_cast<const f32x4*>(s_aff + 2 * K + ka);
    sk += 16;
#pragma unroll
    for (int i = 0; i < IN_PT; ++i) {
      const u32x2 r = pin[S][i];
      f32x4 v = f32x4{__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16),
                      __uint_as_float(r[1] & 0xffff0000u)};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float t = __builtin_fmaf(v[q], sc[q], sh[q]);
        v[q] = __builtin_fmaxf(t, t * sl[q]);
      }
      u32x2 pl[1];
      split4_planes<1>(v, pl);
      const int R = tid / CQ + i * (512 / CQ);
      *reinterpret_cast<u32x2*>(s_in + R * 16 + swz_quad(R, c4)) = pl[0];
    }
#pragma unroll
    for (int i = 0; i < W_PT; ++i) {
      const int u = tid + i * 512;
      const int R = u % TN, kq = u / TN;
      *reinterpret_cast<u32x2*>(s_w + R * 16 + swz_quad(R, kq)) = pw[S][i];
    }
  };

  // live 32-wide cout sub-tiles of this tile: wave group wn owns sub-tiles wn * NPW ... (adjacent in memory: the epilogue's
  // consecutive stores then cover NPW x 64 contiguous bytes of a pixel; the phased kernel deals them alternately)
  const int live = min(NT, (a.Cout - n0 + 31) / 32);
  const int nj = max(0, min(NPW, live - wn * NPW));
  auto mfma_chunk = [&](int buf, auto nj_tag) {
    constexpr int NJ = decltype(nj_tag)::value;
    const unsigned short* s_in = s_base + buf * BUF;
    const unsigned short* s_w = s_in + IN_ROWS * 16;
    bf16x8 ap[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int R = (wm + i * WM) * 32 + l31;
      ap[i] = *reinterpret_cast<const bf16x8*>(s_in + R * 16 + swz_half(R, half));
    }
    bf16x8 bp[2];
    auto load_b = [&](int j) {
      const int R = (wn * NPW + j) * 32 + l31;
      bp[j & 1] = *reinterpret_cast<const bf16x8*>(s_w + R * 16 + swz_half(R, half));
    };
    if (NJ > 0) load_b(0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (j + 1 < NJ) load_b(j + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < RPW; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i], bp[j & 1], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  const bool mul_dma = false;      // (see launch_pw1: no LDS-DMA prefetch in this kernel)
  // ---- prologue: chunks 0 .. DEPTH - 1 requested, chunk 0 staged, chunk DEPTH requested into its set
  c3d_pw_static_for<0, DEPTH>([&](auto d_tag) { load_chunk(d_tag); });
  __syncthreads();                         // the affine table is complete
  store_chunk(0, std::integral_constant<int, 0>{});
  load_chunk(std::integral_constant<int, 0>{});
  __syncthreads();
  for (int c0 = 0; c0 < nchunks; c0 += DEPTH) {
    c3d_pw_static_for<0, DEPTH>([&](auto j_tag) {
      constexpr int J = decltype(j_tag)::value;           // chunk c0 + J: LDS buffer J & 1 (DEPTH is even), register set J
      constexpr int cur = J & 1;
      using next_set = std::integral_constant<int, (J + 1) % DEPTH>;
      store_chunk(cur ^ 1, next_set{});    // chunk c0 + J + 1 (zeros past the end), requested DEPTH chunks ago
      load_chunk(next_set{});              // chunk c0 + J + 1 + DEPTH
      if (nj >= NPW) mfma_chunk(cur, std::integral_constant<int, NPW>{});
      else if (NPW > 3 && nj == 3) mfma_chunk(cur, std::integral_constant<int, (NPW > 3 ? 3 : 1)>{});
      else if (NPW > 2 && nj == 2) mfma_chunk(cur, std::integral_constant<int, (NPW > 2 ? 2 : 1)>{});
      else if (nj == 1) mfma_chunk(cur, std::integral_constant<int, 1>{});
      __syncthreads();                     // the other buffer is complete, this one is free again
    });
  }
  conv_epilogue<TR, NT, WM, WN, true, false, 512, true, SM>(a, acc, smem, tid, lane, half, l31, wm, wn, b, x0, y0, n0, mt, ntile, tile_pix, mul_dma);
}

// ---------------------------------------------------------------------------------------------------
// Round 3: the same tile, arithmetic and LDS image, with the staging dealt INTO the MFMA stream.
//
// profiles/round3_coissue_probe.md: one wave hides <= 5 plain VALU / one ds_write_b64 / two ds_read_b128 per
// v_mfma_f32_32x32x16_bf16 gap at no cost (32.3 -> 35 cycles per MFMA); what does not hide is packed-f32 VALU
// (v_pk_fma/mul/add_f32: +18 cycles for one per gap) -- which is what hipcc makes of every float4 expression,
// and what round 2's "VALU and MFMA do not overlap" measured.  conv_pw3_kernel above runs its phases in
// lockstep (all eight waves split + store chunk c+1, then all eight multiply chunk c, one barrier per
// chunk), so matrix 1.10 ms + staging 0.60 ms + 0.24 ms simply add up on the 704 -> 704 layer.
// Here every wave's chunk iteration is ONE basic block: the 6 * NJ MFMA pairs of chunk c, and between
// consecutive pairs a slice ("atoms") of the work that stages chunk c+1 and requests chunk c+2:
//     a(i,e)   BatchNorm affine + LeakyReLU of two channels of input unit i         6 VALU (scalar f32 only)
//     h/m/l    one bf16 plane of that pair + the exact residual                     5 / 5 / 1 VALU
//     ds_write_b64 of a finished plane of a unit / of a pre-split weight unit        1 LDS store
//     the global loads of chunk c+2 (as soon as the registers they land in are free)
// pinned in source order by __builtin_amdgcn_sched_barrier(0) after every pair (this file is compiled
// without packed-f32 ops, see NOPK in the Makefile).  Everything inside the block is unconditional: a source
// without BatchNorm loads scale = 1 / shift = 0 from a constant with a zero per-chunk step, a source
// without LeakyReLU uses slope 1 (max(v, 1 * v) = v), the cursor is clamped at the last chunk.  The values
// that reach LDS, the product order and the epilogue are those of conv_pw3_kernel: outputs are bit-identical
// (tests/test_gpu_conv.py::test_fused_pointwise_kernel_is_bit_identical_to_the_phased_one).
// scale (16 x 1) and shift (16 x 0) of a source without BatchNorm: every thread reads its channel quad of a 16-channel
// chunk, the per-chunk step is zero
__device__ float c3d_unit_affine[32] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f,
                                        0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

template <int I, int N, class F>
__device__ __forceinline__ void c3d_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    c3d_static_for<I + 1, N>(f);
  }
}

// The staging work of one chunk as a sequence of atoms (kind, unit, plane, element pair), in issue order:
// per input unit  XF e0, XF e1 (affine + LeakyReLU), LIN (reload the unit's registers with chunk c+2), then per
// plane two SPLIT atoms and the unit's ds_write_b64; the weight units (three stores + their reload) are dealt
// between the input units so that the LDS stores are spread evenly; the scale / shift reload comes last.
enum { A_XF, A_LIN, A_SPLIT, A_STIN, A_STW, A_LW, A_LAFF };
struct c3d_atom_seq {
  int n;
  int kind[96], u[96], p[96], e[96];
};
constexpr c3d_atom_seq c3d_make_atoms(int in_pt, int w_pt) {
  c3d_atom_seq s{};
  int n = 0, wdone = 0;
  for (int i = 0; i < in_pt; ++i) {
    for (int e = 0; e < 2; ++e) { s.kind[n] = A_XF; s.u[n] = i; s.e[n] = e; ++n; }
    s.kind[n] = A_LIN; s.u[n] = i; ++n;
    for (int p = 0; p < 3; ++p) {
      for (int e = 0; e < 2; ++e) { s.kind[n] = A_SPLIT; s.u[n] = i; s.p[n] = p; s.e[n] = e; ++n; }
      s.kind[n] = A_STIN; s.u[n] = i; s.p[n] = p; ++n;
    }
    if (wdone < w_pt) {
      for (int p = 0; p < 3; ++p) { s.kind[n] = A_STW; s.u[n] = wdone; s.p[n] = p; ++n; }
      s.kind[n] = A_LW; s.u[n] = wdone; ++n;
      ++wdone;
    }
  }
  for (; wdone < w_pt; ++wdone) {
    for (int p = 0; p < 3; ++p) { s.kind[n] = A_STW; s.u[n] = wdone; s.p[n] = p; ++n; }
    s.kind[n] = A_LW; s.u[n] = wdone; ++n;
  }
  s.kind[n] = A_LAFF; ++n;
  s.n = n;
  return s;
}

// WN = 2: eight waves (4 pixel-row groups x 2 cout groups), one workgroup per CU -- round 2's geometry.
// WN = 1: four waves per workgroup and TWO workgroups per CU: the barrier bubbles, the prologue and the
//         epilogue of one workgroup (0.5 of the 704 -> 704 layer's time on the eight-wave form: the matrix pipe
//         is busy 0.55-0.6 of the time) run under the other workgroup's MFMAs.
template <int NT, int WN>
__global__ __launch_bounds__(256 * WN, WN == 1 ? 2 : 1) void conv_pw3f_kernel(ConvArgs a) {
  constexpr int NP = 3, TR = 8, CQ = 4;
  constexpr int TN = 32 * NT;
  constexpr int WM = 4, RPW = 2, NPW = NT / WN;
  constexpr int NTHR = 64 * WM * WN;
  constexpr int IN_ROWS = TR * 32;
  constexpr int IN_PT = IN_ROWS * CQ / NTHR;       // input units (4 channels of a pixel) per thread
  constexpr int W_PT = TN * CQ / NTHR;             // weight units (4 channels of a cout) per thread
  constexpr int BUF = NP * (IN_ROWS + TN) * 16;    // bf16 elements per LDS buffer
  static_assert(IN_ROWS * CQ % NTHR == 0 && TN * CQ % NTHR == 0 && NT % WN == 0, "staging units must tile the workgroup");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short* s_base = reinterpret_cast<unsigned short*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: the
  const int half = lane >> 5, l31 = lane & 31;                                   // sub-tile dispatch below must be
  const int wm = wave % WM, wn = wave / WM;                                      // scalar branches

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

  // ---- load cursor.  Every load is a buffer load: the descriptor (base of the tile in the current source / of
  //      the weight planes) and the K offset (soffset) live in SGPRs, the per-thread part is a 32-bit voffset that
  //      only changes when the cursor enters another source -- no per-chunk pointer arithmetic in VGPRs.
  const int c4 = tid % CQ;
  const size_t tile_pix = (size_t)(b * a.H + y0) * a.W + x0;
  const unsigned wplane_b = (unsigned)a.Kq * a.Cout * 8;                         // bytes per bf16 weight plane (T = 1)
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.wpack) + (size_t)a.Kq * a.Cout * 4, 0, 0x7fffffff, 0x00020000);   // planes follow the fp32 pack
  int vw[W_PT];
#pragma unroll
  for (int i = 0; i < W_PT; ++i) {
    const int u = tid + i * NTHR;
    const int n = min(n0 + u % TN, a.Cout - 1), kq = u / TN;
    vw[i] = (kq * a.Cout + n) * 8;
  }
  __amdgpu_buffer_rsrc_t rs_in, rs_sc, rs_sh;
  int vin[IN_PT];
  const int vaff = c4 * 16;
  int lstep = 0;                           // per-chunk step of the scale / shift offset (bytes; 0: the unit constant)
  float lslope = 1.f;
  int ls = 0, lc0 = 0, lC = 0, lk = 0;     // source, channel inside it, its width, global chunk index
  auto open_src = [&](int s) {
    const c3d_src& sr = a.src[s];
    rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sr.ptr) + tile_pix * sr.cstride + sr.coff, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int i = 0; i < IN_PT; ++i) {
      const int p = tid / CQ + i * (NTHR / CQ);
      const int gx = min(x0 + (p & 31), a.W - 1), gy = min(y0 + (p >> 5), a.H - 1);
      vin[i] = (((gy - y0) * a.W + (gx - x0)) * sr.cstride + c4 * 4) * 4;
    }
    if (sr.scale) {
      rs_sc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sr.scale), 0, 0x7fffffff, 0x00020000);
      rs_sh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sr.shift), 0, 0x7fffffff, 0x00020000);
      lstep = 64;
    } else {
      rs_sc = __builtin_amdgcn_make_buffer_rsrc(c3d_unit_affine, 0, 0x7fffffff, 0x00020000);
      rs_sh = __builtin_amdgcn_make_buffer_rsrc(c3d_unit_affine + 16, 0, 0x7fffffff, 0x00020000);
      lstep = 0;
    }
    lslope = sr.lrelu ? a.slope : 1.f;
    lC = sr.C;
    lc0 = 0;
  };
  open_src(0);
  const int nchunks = a.Kq / 4;            // K / 16
  auto advance = [&]() {                   // cursor -> next chunk; stays on the last one (clamped re-loads)
    if (lk + 1 < nchunks) {
      ++lk;
      lc0 += 16;
      if (lc0 >= lC) open_src(++ls);
    }
  };

  // ---- registers of the chunks in flight
  f32x4 pin[IN_PT];
  u32x2 pw[W_PT][NP];
  f32x4 psc, psh;
  float pslope;
  auto load_in = [&](int i) { pin[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, vin[i], lc0 * 4, 0)); };
  auto load_aff = [&]() {
    const int so = (lstep >> 6) * lc0 * 4;
    psc = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_sc, vaff, so, 0));
    psh = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_sh, vaff, so, 0));
    pslope = lslope;
  };
  auto load_w = [&](int i) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
      pw[i][p] = __builtin_amdgcn_raw_buffer_load_b64(rs_w, vw[i], lk * a.Cout * 32 + p * wplane_b, 0);
  };

  constexpr c3d_atom_seq SEQ = c3d_make_atoms(IN_PT, W_PT);
  constexpr int NATOM = SEQ.n;
  f32x4 sv;                                // the unit being split: transformed values, then the residuals
  unsigned pl[2];                          // the plane being formed (4 bf16 of the unit)
  auto atom = [&](auto k_tag, unsigned short* s_in, unsigned short* s_w) {
    constexpr int k = decltype(k_tag)::value;
    constexpr int kind = SEQ.kind[k], i = SEQ.u[k], p = SEQ.p[k], e = SEQ.e[k];
    if constexpr (kind == A_XF) {
#pragma unroll
      for (int q = 2 * e; q < 2 * e + 2; ++q) {
        const float v = __builtin_fmaf(pin[i][q], psc[q], psh[q]);
        sv[q] = __builtin_fmaxf(v, v * pslope);
      }
    } else if constexpr (kind == A_LIN) {
      load_in(i);
    } else if constexpr (kind == A_LAFF) {
      load_aff();
    } else if constexpr (kind == A_STW) {
      const int u = tid + i * NTHR;
      const int R = u % TN, kq = u / TN;
      *reinterpret_cast<u32x2*>(s_w + (p * TN + R) * 16 + swz_quad(R, kq)) = pw[i][p];
    } else if constexpr (kind == A_LW) {
      load_w(i);
    } else if constexpr (kind == A_SPLIT) {
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
      bf16x2 h;
      h[0] = (__bf16)sv[2 * e];
      h[1] = (__bf16)sv[2 * e + 1];
      const unsigned pk = __builtin_bit_cast(unsigned, h);     // one v_cvt_pk_bf16_f32
      pl[e] = pk;
      if constexpr (p < 2) {                                   // exact residuals: x - bf16(x) is representable
        sv[2 * e] -= __uint_as_float(pk << 16);
        sv[2 * e + 1] -= __uint_as_float(pk & 0xffff0000u);
      }
    } else {   // A_STIN
      const int R = tid / CQ + i * (NTHR / CQ);
      *reinterpret_cast<u32x2*>(s_in + (p * IN_ROWS + R) * 16 + swz_quad(R, c4)) = u32x2{pl[0], pl[1]};
    }
  };

  // live 32-wide cout sub-tiles of this tile, dealt alternately to the cout wave groups
  const int live = min(NT, (a.Cout - n0 + 31) / 32);
  const int nj = (live - wn + WN - 1) / WN;

  // one chunk: the MFMAs of buffer `cur` with (STAGE) the atoms of the next chunk into the other buffer
  auto chunk = [&](int cur, auto nj_tag, auto stage_tag) {
    constexpr int NJ = decltype(nj_tag)::value;
    constexpr bool STAGE = decltype(stage_tag)::value;
    const unsigned short* s_in = s_base + cur * BUF;
    const unsigned short* s_w = s_in + NP * IN_ROWS * 16;
    unsigned short* d_in = s_base + (cur ^ 1) * BUF;
    unsigned short* d_w = d_in + NP * IN_ROWS * 16;
    if constexpr (NJ == 0) {
      if constexpr (STAGE) c3d_static_for<0, NATOM>([&](auto k) { atom(k, d_in, d_w); });
    } else {
      // six of the nine plane products, smallest first: (A plane, B plane)
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
      constexpr int NS = 6 * NJ;                    // MFMA pairs = slots
      bf16x8 ap[NP][RPW];
      // weight fragments: plane 0 (first and last product of a sub-tile) double-buffered, planes 1 and 2 re-read
      // right after their last product of sub-tile j for sub-tile j+1
      bf16x8 b0[2], b1, b2;
      // (`fresh` is zero but opaque to the compiler, redefined per chunk: fragment addresses are recomputed next to
      //  their reads instead of being hoisted into registers -- the four-wave form otherwise needs scratch, and a
      //  kernel with scratch gets one workgroup per CU on this GPU)
      int fresh = 0;
      asm volatile("" : "+v"(fresh));
      auto read_a = [&](int p) {
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
          const int R = (wm + i * WM) * 32 + l31 + fresh;
          ap[p][i] = *reinterpret_cast<const bf16x8*>(s_in + p * IN_ROWS * 16 + R * 16 + swz_half(R, half));
        }
      };
      auto read_b = [&](int j, int p) {
        const int R = (j * WN + wn) * 32 + l31 + fresh;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(s_w + p * TN * 16 + R * 16 + swz_half(R, half));
        if (p == 0) b0[j & 1] = v;
        else if (p == 1) b1 = v;
        else b2 = v;
      };
      // fragments in the order the products need them
      read_a(2); read_b(0, 0); read_a(0); read_b(0, 2); read_a(1); read_b(0, 1);
      __builtin_amdgcn_sched_barrier(0);
      c3d_static_for<0, NS>([&](auto s_tag) {
        constexpr int s = decltype(s_tag)::value, j = s / 6, q = s % 6;
        if constexpr (j + 1 < NJ) {
          if constexpr (q == 0) read_b(j + 1, 0);
          if constexpr (q == 2) read_b(j + 1, 2);      // plane 2's only product of sub-tile j was q = 1
          if constexpr (q == 5) read_b(j + 1, 1);      // plane 1's last product is q = 4
        }
        const bf16x8 bq = PB[q] == 0 ? b0[j & 1] : (PB[q] == 1 ? b1 : b2);
#pragma unroll
        for (int i = 0; i < RPW; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[PA[q]][i], bq, acc[i][j], 0, 0, 0);
        if constexpr (STAGE) c3d_static_for<(s * NATOM) / NS, ((s + 1) * NATOM) / NS>([&](auto k) { atom(k, d_in, d_w); });
        __builtin_amdgcn_sched_barrier(0);
      });
    }
  };

  // prologue: chunk 0 into registers, then through the atoms into buffer 0 (which also request chunk 1)
#pragma unroll
  for (int i = 0; i < IN_PT; ++i) load_in(i);
  load_aff();
#pragma unroll
  for (int i = 0; i < W_PT; ++i) load_w(i);
  advance();
  c3d_static_for<0, NATOM>([&](auto k) { atom(k, s_base, s_base + NP * IN_ROWS * 16); });
  advance();
  __syncthreads();
  // the number of live sub-tiles is fixed for the launch: one copy of the K loop per value, so that each loop
  // body is a single basic block with its own register allocation
  auto k_loop = [&](auto nj_tag) {
    for (int c = 0; c + 1 < nchunks; ++c) {
      chunk(c & 1, nj_tag, std::true_type{});      // multiplies chunk c, stages chunk c+1, requests chunk c+2
      advance();
      __syncthreads();                             // the other buffer is complete, this one is free again
    }
    chunk((nchunks - 1) & 1, nj_tag, std::false_type{});
    __syncthreads();                               // the epilogue reuses the LDS
  };
  if (nj >= NPW) k_loop(std::integral_constant<int, NPW>{});
  else if (NPW > 3 && nj == 3) k_loop(std::integral_constant<int, (NPW > 3 ? 3 : 1)>{});
  else if (NPW > 2 && nj == 2) k_loop(std::integral_constant<int, (NPW > 2 ? 2 : 1)>{});
  else if (nj == 1) k_loop(std::integral_constant<int, 1>{});
  else k_loop(std::integral_constant<int, 0>{});
  conv_epilogue<TR, NT, WM, WN, false, true, NTHR, true, true>(a, acc, smem, tid, lane, half, l31, wm, wn, b, x0, y0, n0, mt, ntile,
                                                               tile_pix);
}

template <int NT, int WN>
int launch_pw3f(ConvArgs& a, hipStream_t st) {
  constexpr int NP = 3;
  size_t lds = (size_t)2 * NP * (8 * 32 + 32 * NT) * 16 * 2;
  const size_t red = (size_t)4 * 32 * NT * 2 * sizeof(float);   // statistics scratch of the epilogue
  if (lds < red) lds = red;
  c3d_opt_in_lds<&conv_pw3f_kernel<NT, WN>>();
  a.ntn = (a.Cout + 32 * NT - 1) / (32 * NT);
  dim3 grid(a.B * a.tiles_x * a.tiles_y * a.ntn);
  hipLaunchKernelGGL((conv_pw3f_kernel<NT, WN>), grid, dim3(256 * WN), lds, st, a);
  C3D_CHECK_LAUNCH();
  return 0;
}

template <int NT>
int launch_pw1(ConvArgs& a, hipStream_t st) {
  size_t lds = (size_t)2 * (8 * 32 + 32 * NT) * 16 * 2 + (size_t)3 * a.Kq * 4 * sizeof(float);    // two buffers + the affine table
  const size_t red = (size_t)4 * 32 * NT * 2 * sizeof(float);   // statistics scratch of the epilogue
  if (lds < red) lds = red;
  a.ntn = (a.Cout + 32 * NT - 1) / (32 * NT);
  dim3 grid(a.B * a.tiles_x * a.tiles_y * a.ntn);
  // BatchNorm-backward sums in the epilogue: the instance that has it (128-cout tiles only: with 256 the multiplier tiles of
  // four sub-tiles per wave end up in scratch, 4 326 scratch instructions -- c3d_conv_stat_mul_supported() answers 0 there)
  if constexpr (NT == 4) if (a.stat_mul && a.stat_partial) {
    if (lds < (size_t)8 * 32 * (32 * NT + 8) * 2) lds = (size_t)8 * 32 * (32 * NT + 8) * 2;      // the multiplier tile of the epilogue
    // (round 6: the LDS-DMA form of the multiplier tile -- conv_mul_dma_issue, which conv_x3f and conv_bfp use -- measured 34 -> 73 us
    //  here: the only place to request it without a branch around the ring's counted loads is the prologue, and 32 KB of in-order
    //  DMA in front of the first chunks delays the whole ring.  This kernel keeps the copy at the start of its epilogue.)
    a.lds_bytes = (unsigned)lds;
    c3d_opt_in_lds<&conv_pw1_kernel<NT, true>>();
    hipLaunchKernelGGL((conv_pw1_kernel<NT, true>), grid, dim3(512), lds, st, a);
    C3D_CHECK_LAUNCH();
    return 0;
  }
  c3d_opt_in_lds<&conv_pw1_kernel<NT>>();
  hipLaunchKernelGGL((conv_pw1_kernel<NT>), grid, dim3(512), lds, st, a);
  C3D_CHECK_LAUNCH();
  return 0;
}

template <int NT, int NP>
int launch_pw3(ConvArgs& a, hipStream_t st) {
  size_t lds = (size_t)2 * NP * (8 * 32 + 32 * NT) * 16 * 2;
  const size_t red = (size_t)4 * 32 * NT * 2 * sizeof(float);   // statistics scratch of the epilogue
  if (lds < red) lds = red;
  c3d_opt_in_lds<&conv_pw3_kernel<NT, NP>>();
  a.ntn = (a.Cout + 32 * NT - 1) / (32 * NT);
  dim3 grid(a.B * a.tiles_x * a.tiles_y * a.ntn);
  hipLaunchKernelGGL((conv_pw3_kernel<NT, NP>), grid, dim3(512), lds, st, a);
  C3D_CHECK_LAUNCH();
  return 0;
}

}  // namespace

// called by c3d_conv_forward for mfma_bf16 == 1 / 2 (planes = 1 / 3), 8-row tiles, one tap, Cout > 64;
// a.wpack must be a c3d_pack_weights(mode | 2) pack (fp32 image followed by the three bf16 planes)
int c3d_conv_forward_pw3(ConvArgs& a, int planes, bool wide, hipStream_t st) {
  if (planes == 3) {
    // Geometry of the fused kernel (mirrored by ops._pw3_kernel_name): eight waves x 256 / 128 couts per workgroup;
    // with a short K (<= 256 channels: the prologue, the epilogue and the barrier bubbles are a large share of a
    // workgroup's life) and couts that tile by 128, four waves x 128 couts with TWO workgroups per CU, which run those
    // under each other's MFMAs (128 -> 384 at 8x32x1024: 0.181 -> 0.159 ms; 704 -> 704: 1.16 vs 1.22, so long K stays
    // on eight waves).  c3d_conv_desc.variant & 3: 0 = this choice, 1 / 2 = force eight / four waves, 3 = round 2's
    // phased kernel (the bit-identity tests compare all of them).
    const int mode = a.variant & 3;
    const bool four = mode == 2 || (mode != 1 && a.Kq * 4 <= 256 && a.Cout % 128 == 0);
    if (mode != 3) {
      if (four) return launch_pw3f<4, 1>(a, st);
      return wide ? launch_pw3f<8, 2>(a, st) : launch_pw3f<4, 2>(a, st);
    }
    return wide ? launch_pw3<8, 3>(a, st) : launch_pw3<4, 3>(a, st);
  }
  // one plane: over bf16 tensors the kernel with four chunks in flight (c3d_conv_desc.variant & 3 == 3: the phased one)
  bool all_bf = (a.variant & 3) != 3;
  for (int s = 0; s < a.nsrc; ++s) all_bf = all_bf && a.src[s].bf16 != 0;
  if (all_bf) return wide ? launch_pw1<8>(a, st) : launch_pw1<4>(a, st);
  return wide ? launch_pw3<8, 1>(a, st) : launch_pw3<4, 1>(a, st);
}
