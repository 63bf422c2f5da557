// Convolution weight gradient on the bf16 matrix pipe for gfx950, operands read through the LDS
// transpose read (ds_read_b64_tr_b16).
//
//   dW[cout][cin][t] = sum_pixels x[p + tap_t][cin] * dz[p][cout]        (GEMM: M = cin, N = cout, K = pixels)
// (autograd of the nn.Conv2d layers of pc_processor/models/salsanext_proto.py:41-62, 82-132,
// 164-208, 318 and projector.py:18-23; same workgroup decomposition, partial layout and fold as
// wgrad_mfma.hip.)
//
// v_mfma_f32_32x32x16_bf16 wants 8 CONSECUTIVE K values (pixels) per lane, but the activations
// are stored [pixel][channel]: a plain LDS read would gather 8 rows.  ds_read_b64_tr_b16 does the
// 4x4 transpose in the LDS pipe: a 16-lane group reads 4 pixel rows x 16 channels (lane s points
// at 4 channels of pixel s/4) and lane i receives channel i of those 4 pixels -- two such reads
// make one MFMA operand.  The LDS image therefore stays [pixel][channel] (bf16), tap shifts are
// whole-row offsets (no alignment problem), and staging writes are plain 8-byte stores.
//
// NP = 3 ("bf16x3"): every fp32 operand is split exactly into three bf16 planes when the tile is
// staged (once per element per workgroup) and SIX of the nine plane products are accumulated in
// fp32: h*h, h*m, m*h, m*m, h*l, l*h.  The convolution kernels keep eight (l*m and m*l too) because
// with six each product is off by up to 2^-23 |a||b| and that measured ~4x the fp32 engine's
// gradient noise through the network; a weight gradient is a sum over 10^4..10^6 pixels whose fp32
// accumulation error dwarfs those terms -- measured against float64 on twelve layer shapes
// (tools/bench_wgrad.py) the six- and eight-product kernels agree to every printed digit but one
// (7.63e-7 vs 7.67e-7 of max|dW| on the 704x704 layer; 5.05e-7, 4.05e-7, 4.98e-7 ... identical),
// the per-layer float64 gradient check and the whole GPU suite pass unchanged, and the kernels
// are 13-15 % faster.  Six bf16 MFMAs cost 3/8 of the issue time of the 8 fp32 MFMAs (32x32x2)
// they replace.  NP = 1 ("bf16"): operands rounded to bf16 (RNE) at staging, one product.
//
// Bank conflicts: one tr read touches 4 pixel rows x 64 B per 32-lane pass.  Rows are unpadded
// (32*NS channels, NS = 1, 2, 4 or 8 segments of 64 B); segment s of pixel row R is stored at
// segment s ^ f(R) with f = 0 / (R>>1)&1 / R&3 for NS = 1 / 2 / >= 4, which puts any 4 consecutive
// rows in 4 different 16-bank groups (SQ_LDS_BANK_CONFLICT = 0 on every instance).
//
// Workgroup = 8 waves, one per CU: waves 0-3 (one per SIMD) only issue transposed reads + MFMAs on
// the current tile buffer, waves 4-7 stage the next tile into the other LDS buffer and keep the
// loads of TWO further tiles in flight in registers (one tile per CU in flight left the kernel
// bound by memory latency).  One barrier per tile.  Measured on the 704x704 1x1 layer at
// 8x32x1024 (bf16x3): fp32-MFMA kernel 2.77 ms, this kernel 2.26 ms; consumer waves alone 1.57 ms,
// producer waves alone 1.06 ms, and round 2 read the sum as "VALU and MFMA of one SIMD do not overlap".  Round 3
// (profiles/round3_coissue_probe.md): plain VALU of the producer wave overlaps with the consumer wave's MFMAs
// completely; what took the matrix pipe's issue slots were the PACKED f32 ops hipcc made of the staging code (5 136
// v_pk_* in this file).  The file is now compiled without them (NOPK in the Makefile: -0.55 ms of weight-gradient
// time per step) and the consumer stages are software-pipelined.  (dz kept as pre-split planes in HBM was built and
// measured in round 2: bit-identical, 1 % slower for the step -- three 8-byte loads per unit instead of one 16-byte
// load cost what the saved split instructions gain; DESIGN.md.)
#include <type_traits>
#include "wgrad_common.h"

namespace {

// lean register sets from this many tiles per column on (launch_tr)
constexpr int C3D_WGRAD_LEAN_MIN_TILES_Y = 8;

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

// EXPERIMENT (NP == 2): two fp16 planes, H = RNE11(x), L = RNE11(x - H); three products (DESIGN.md round-4 list)
template <int NP>
__device__ __forceinline__ void split_planes(f32x4 v, u32x2 (&out)[NP]) {
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
  // Packed conversions and bit expansion (round 5): element-wise (__bf16) casts cost 14 v_cvt_pk_bf16_f32 + 10 shifts + 8
  // subtractions per unit (one conversion per element plus the packing); this form 6 + 10 + 8.  Same RNE values, same bits.
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x4 r = v;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const unsigned w0 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r[0], r[1]}, bf16x2));
    const unsigned w1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r[2], r[3]}, bf16x2));
    out[p] = u32x2{w0, w1};
    if (p + 1 < NP) {
      r[0] -= __uint_as_float(w0 << 16);
      r[1] -= __uint_as_float(w0 & 0xffff0000u);
      r[2] -= __uint_as_float(w1 << 16);
      r[3] -= __uint_as_float(w1 & 0xffff0000u);
    }
  }
}

// element offset (bf16 units) of channel c of pixel row R in a [rows][32*NS] image
template <int NS>
__device__ __forceinline__ int tr_swz(int R, int c) {
  const int f = NS == 1 ? 0 : (NS == 2 ? ((R >> 1) & 1) : (R & 3));
  return R * (32 * NS) + (((c >> 5) ^ f) << 5) + (c & 31);
}

__device__ __forceinline__ s16x4 tr_read(const unsigned short* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

// The same from an LDS BYTE address split into a wave-uniform part (SALU arithmetic) and a per-lane constant: one v_add per
// fragment instead of the multiply / shift / or chain of tr_swz on a per-lane row index (round 5)
__device__ __forceinline__ s16x4 tr_read_u(unsigned byte_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(size_t)byte_addr);
}
template <int NS>
__device__ __forceinline__ bf16x8 tr_frag_u(unsigned ubytes, unsigned lane_bytes) {
  const unsigned a0 = ubytes + lane_bytes;
  const s16x4 lo = tr_read_u(a0);
  const s16x4 hi = tr_read_u(a0 + 4u * 32u * NS * 2u);          // four pixel rows on (same swizzle class: rows R and R + 4)
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// one MFMA operand: channel (lane) x 8 pixels = two transposed reads 4 pixel rows apart
template <int NS>
__device__ __forceinline__ bf16x8 tr_frag(const unsigned short* plane, int R, int c) {
  const s16x4 lo = tr_read(plane + tr_swz<NS>(R, c));
  const s16x4 hi = tr_read(plane + tr_swz<NS>(R + 4, c));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// 4 channels at (uniform element offset ubase) + (per-lane element offset off): the split keeps the
// address a scalar base plus a 32-bit lane offset (global_load ... v, s[base]) -- no 64-bit lane math.
// (readfirstlane pins the uniform part in SGPRs: without it the optimiser re-associates the sum
//  and hoists one loop-invariant 64-bit lane address per staged unit out of the tile loop -- 24
//  VGPRs, which spill.)
__device__ __forceinline__ const char* c3d_uniform_ptr(const void* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}
template <bool BF>
__device__ __forceinline__ f32x4 c3d_ld4u(const float* p, size_t ubase, unsigned off) {
  if constexpr (BF) {
    const char* b = c3d_uniform_ptr(reinterpret_cast<const unsigned short*>(p) + ubase);
    // (explicit global address space: the integer round trip above would otherwise leave a generic
    //  pointer and a flat_load, which counts on lgkmcnt and stalls every LDS wait behind the prefetch)
    const c3d_u32x2 r = *(const __attribute__((address_space(1))) c3d_u32x2*)(b + (size_t)(off * 2u));
    f32x4 v;
    v[0] = __uint_as_float(r[0] << 16);
    v[1] = __uint_as_float(r[0] & 0xffff0000u);
    v[2] = __uint_as_float(r[1] << 16);
    v[3] = __uint_as_float(r[1] & 0xffff0000u);
    return v;
  } else {
    const char* b = c3d_uniform_ptr(p + ubase);
    return *(const __attribute__((address_space(1))) f32x4*)(b + (size_t)(off * 4u));
  }
}

template <int I, int N, class F>
__device__ __forceinline__ void c3d_wg_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    c3d_wg_static_for<I + 1, N>(f);
  }
}

// FA ("fused apply", round 4): the BatchNorm / LeakyReLU backward of the layer runs while dz is staged -- the producer
// waves read dy and the layer's stored output instead of dz, form dz = LeakyReLU'(act) * (k1 * dy + k2 * act + k3), write
// it out for the input-gradient convolution that follows (workgroups of cin slice 0 only) and keep its per-channel sums
// (the bias gradient).  The separate c3d_bn_bwd_apply pass (three tensor passes at HBM speed, the largest kernel of the
// round-3 step) disappears; the arithmetic is the same fmaf chain, dz is bit-identical.
// RAW (round 4, the bf16 engine over bf16 tensors): the tiles in flight travel as the 8 bytes per unit they are loaded as -- half
// the registers of the widened values -- and FOUR tiles are kept in flight instead of two.  A tile of this mode is 8-16 KB of
// loads; with two in flight per workgroup the weight gradients of this engine took exactly as long as on fp32 tensors (122 us
// for the 64 -> 64 1x1 layer at 8 x 64 x 2048 either way: 2.2 vs 4.4 TB/s), bound by requests in flight, not by bytes.
// LEAN (round 5, kernels with a halo): the register sets in flight carry only the TRW NEW rows of a window.  Round 4's sets were
// sized for a whole window (TRW + 2 * HALO rows) although only the first tile of a column needs one -- every other tile issued
// the surplus loads as dummies ("every unit issues its load").  With the BatchNorm backward on load (FA: dy AND the stored
// output in flight) that put the producer waves past the 256 registers a 512-thread workgroup has: the fused instances
// spilled (up to 553 scratch instructions), and a scratch reload is the YOUNGEST memory operation when its value is needed,
// so every one of them was an s_waitcnt vmcnt(0) that drained both tiles in flight (80-356 such waits per instance against 3
// in the unfused ones; the fused 32 -> 32 3x3 weight gradient ran at 110 TF against 197 unfused).  The 2 * HALO rows that
// only a column's first tile needs are now loaded inside store_tile of that tile, into the registers its x units have just
// left, and converted after the dz units (whose conversion hides most of their latency).
// NPW (round 5, the 1x1 instances of the three-plane engine): EIGHT producer waves, two per SIMD, in a 768-thread workgroup.  A
// 1x1 weight gradient has next to no matrix work per staged byte (matrix pipe busy 0.05-0.16): its time is the producer waves'
// -- one wave per SIMD issuing a dependent VALU chain per staged unit with nothing to fill its stalls, and two tiles of
// loads in flight per CU.  Its accumulators are small (16-64 registers), so the 168 registers of a twelve-wave workgroup
// are enough; a wave stages half the units per tile and the CU keeps twice the waves' worth of loads in flight.
// NCW = 8 (round 5, the nine-tap instances of the three-plane engine; with NPW = 8: a 1024-thread workgroup, four waves per SIMD
// at 128 registers): the TAPS are split across two consumer waves per SIMD so that a second producer wave per SIMD fits.  The
// ablation (tools/ablate_wgrad.py) says these kernels are bound by their producer waves -- one per SIMD, a dependent VALU
// chain per staged unit, nothing to fill its stalls -- while a consumer wave holds 9 x 16 accumulator registers and leaves
// no room for more waves.  Consumer wave (w, half h) of SIMD w: taps 0-4 (h = 0) or 5-8 (h = 1) on every K step of w's pixel
// share -- five accumulators = 80 registers.  The two halves share the SIMD's matrix pipe, so that five against four taps is
// no imbalance of the pipe (half 0 runs its last sixth alone).  One tap per stage: the six products of a tap are a dependent
// MFMA chain, and the two consumer waves of a SIMD interleave theirs (a chain alone issues at 41 cycles per MFMA, the matrix
// pipe takes one every 32).  Every tap sees its K steps and products in the order of the four-wave form: the same bits.
template <int NP, int TMAX, int CI_T, int CO_T, int WCI, int WCO, int TRW, int HALO, bool FA = false, bool RAW = false, bool LEAN = false,
          int NPW = 4, int NCW = 4>
__global__ __launch_bounds__(64 * (NCW + NPW), 1) void wgrad_tr_kernel(WgradArgs a) {
  constexpr int NPT = 64 * NPW;        // producer threads
  constexpr int NCT = 64 * NCW;        // consumer threads
  constexpr bool SPL = NCW == 8;       // taps split across two consumer waves per SIMD
  // ROLES (round 6, the nine-tap instances and the four-tap 64 x 64 one with the BatchNorm backward on load): the producer waves split by TENSOR -- waves
  // 0 .. NPW/2 - 1 stage the x units of a tile, the others its dz units (dy and the stored output in flight, the fmaf chain, the
  // dz store, the bias sums).  With both tensors in flight twice a producer wave did not fit the 128 registers of a sixteen-wave
  // workgroup (9-45 spilled registers: the fused launches stayed on the four + four wave form, one producer wave per SIMD with
  // nothing to fill its stalls).  Either role alone does.  XT / DT: the threads a tile's x / dz units are dealt to -- with
  // ROLES each is the four + four form's 256, so every unit sits with the thread index it had there: same dz, same bias sums.
  constexpr bool ROLES = FA && SPL;
  constexpr int XT = ROLES ? NPT / 2 : NPT;
  constexpr int DT = ROLES ? NPT - XT : NPT;
  // SPL4 (round 6): the same split for the four-tap instance over 64 x 64 slices -- two taps per half, each wave its cin tile and
  // BOTH cout tiles (4 accumulators = 64 registers instead of 128) -- so that its fused launches can take the ROLES form too
  constexpr bool SPL4 = SPL && TMAX == 4;
  // SPL1 (round 6): the 1x1 instance over 128 x 256 slices (the projector's 704-wide layers) -- the COUT tiles of a wave split
  // across the halves: half th owns cout tiles 2 th, 2 th + 1 of the wave's four, both cin tiles (4 accumulators = 64 registers
  // instead of 128).  Its producers stage ~12 units per thread per 96 MFMAs of a consumer wave, twice the nine-tap instances'
  // ratio: one wave per SIMD runs that chain at ~8 cycles per VALU instruction (ablation, profiles/round6_wgrad_roles.md).
  constexpr bool SPL1 = SPL && TMAX == 1;
  static_assert(NCW == 4 || (NCW == 8 && NPW == 8 && TMAX == 9 && CI_T == 1 && CO_T == 1 && WCI == 1 && (NP == 3 || NP == 1)) ||
                    (NCW == 8 && NPW == 8 && TMAX == 4 && CI_T == 1 && CO_T == 2 && WCI == 2 && WCO == 1 && NP == 3 && !RAW) ||
                    (NCW == 8 && NPW == 8 && TMAX == 1 && CI_T == 2 && CO_T == 4 && WCI == 2 && WCO == 2 && NP == 3 && !RAW && !FA),
                "split consumers: the nine-tap instances of the three-plane and one-plane engines, the four-tap 64 x 64 one; eight producer waves");
  constexpr int TLS = (SPL && !SPL1) ? (TMAX + 1) / 2 : 0;      // taps of a consumer half = barrier pairs of the final fold (SPL1: WK = 1, no fold)
  static_assert(!RAW || (NP == 1 && !FA), "raw bf16 stages belong to the one-plane engine without the fused apply");
  constexpr int DEPTH = RAW ? 4 : 2;   // tiles the producer waves keep in flight (register sets)
  using SU = std::conditional_t<RAW, u32x2, f32x4>;   // a staged unit in flight: four bf16 as loaded, or four floats
  constexpr int WK = 4 / (WCI * WCO);
  constexpr int CI = 32 * CI_T * WCI;  // cin slice of the workgroup
  constexpr int CO = 32 * CO_T * WCO;  // cout slice of the workgroup
  constexpr int NSX = CI / 32, NSD = CO / 32;
  constexpr int TWh = 32 + 2 * HALO;
  constexpr int THh = TRW + 2 * HALO;
  constexpr int XROWS = THh * TWh, DROWS = TRW * 32;
  constexpr int KS = TRW * 2;          // 16-pixel K steps per tile
  constexpr int KPW = KS / WK;         // per wave
  static_assert(KS % WK == 0 && KPW >= 1, "K steps must split across the K waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // two tile buffers, each [NP][XROWS][CI] x planes followed by [NP][DROWS][CO] dz planes
  constexpr int BUF = NP * (XROWS * CI + DROWS * CO);
  unsigned short* s_base = reinterpret_cast<unsigned short*>(smem);
  // ROLLING x WINDOW (round 4, kernels with a halo).  The tiles of a strip walk DOWN the image, so the input window of a tile
  // (TRW + 2 * HALO rows) shares 2 * HALO rows with the window of the tile before it -- with TRW = 2 and HALO = 2 two thirds
  // of every window was loaded, transformed, split into planes and written to LDS again (the producer waves' work, which
  // is what bounds this kernel).  The x image in LDS is now a ring of RING = 2 * (TRW + 2 * HALO) pixel rows -- the memory of
  // the two tile buffers -- in which a window is THh consecutive slots (mod RING): a tile below its predecessor adds its
  // TRW new rows behind the predecessor's window, a tile that starts a column (or the strip) takes the THh slots behind
  // it.  Either way the slots written for tile n + 1 are disjoint from the window the consumers read for tile n.  The dz
  // tiles keep their two buffers.  LDS layout: [NP][RING * TWh][CI] x planes, then 2 x [NP][DROWS][CO] dz planes.
  constexpr bool RINGX = HALO > 0;
  constexpr int RING = 2 * THh;
  constexpr int XPLANE = RINGX ? RING * TWh * CI : XROWS * CI;     // bf16 elements per x plane
  constexpr int NEWROWS_UNITS = TRW * TWh * (CI / 4);               // staging units of the TRW new rows of a window
  constexpr bool LEANX = LEAN && RINGX;
  static_assert(!LEAN || (HALO > 0 && NP == 3 && !RAW), "the lean register sets belong to the three-plane kernels with a rolling window");
  constexpr int EXTRA_UNITS = 2 * HALO * TWh * (CI / 4);            // LEANX: the rows only a column's first tile stages

  // Eight waves, two roles: waves 0-3 (one per SIMD) only issue transposed reads and MFMAs on
  // the current tile buffer; waves 4-7 stage the NEXT tile meanwhile (global loads -> on-load
  // transform -> plane split -> LDS) into the other buffer.  One barrier per tile.  With all
  // waves doing both jobs in turn (the first version of this kernel) the staging and matrix
  // phases of the two resident workgroups fell into lock-step and simply added up
  // (704x704 layer: 0.96 ms staging + 1.50 ms matrix phase -> 2.16 ms).
  const bool producer = __builtin_amdgcn_readfirstlane((int)threadIdx.x) >= NCT;   // wave-uniform, in an SGPR
  // (wave-uniform values pinned in SGPRs, round 5: the compiler cannot prove that threadIdx.x >> 6 or the result of an
  //  integer division of blockIdx-derived numbers is uniform, so the consumer waves computed every fragment's ring slot --
  //  add, compare, select, a quarter-rate v_mul_lo_u32 by the window width -- per lane: 218 VALU instructions per 108 MFMAs
  //  of the 32 x 32 nine-tap instance, in a kernel whose MFMA and VALU streams share the SIMD's issue port -- PMC: 39 % of
  //  the wave cycles were issue stalls.)
  // (consumer threads 0 .. NCT - 1, producer threads 0 .. NPT - 1)
  const bool xrole = !ROLES || __builtin_amdgcn_readfirstlane((int)threadIdx.x) < NCT + XT;          // (producers; wave-uniform)
  const int tid = producer ? (int)threadIdx.x - NCT - (xrole ? 0 : XT) : (int)threadIdx.x, lane = tid & 63;   // index within the role
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = SPL ? (wave8 & 3) : wave8;       // SPL: consumer wave (wave, th); producers never use it
  const int th = SPL ? ((wave8 >> 2) & 1) : 0;
  const int half = lane >> 5, l31 = lane & 31;
  const int wci = wave % WCI, wco = (wave / WCI) % WCO, wk = wave / (WCI * WCO);
  // transposed-read source of this lane: group g = lane>>4 covers channels 16*(g&1).. of pixels 8*(g>>1)..
  const int lp = 8 * (lane >> 5) + ((lane & 15) >> 2);
  const int lc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  const int nsl = a.ci_slices * a.co_slices;
  const int logical = __builtin_amdgcn_readfirstlane(c3d_xcd_remap(blockIdx.x, a.strips * nsl));
  const int strip = __builtin_amdgcn_readfirstlane(logical / nsl);
  const int sl = logical - strip * nsl;
  const int ci0 = (sl % a.ci_slices) * CI;
  const int co0 = (sl / a.ci_slices) * CO;

  // ---- register staging (prefetch) of the next pixel tile while the current one is consumed
  constexpr int X_UNITS = LEANX ? NEWROWS_UNITS : XROWS * (CI / 4);      // x units of a register set in flight
  constexpr int X_PT = (X_UNITS + XT - 1) / XT;
  constexpr int E_PT = LEANX ? (EXTRA_UNITS + XT - 1) / XT : 0;
  constexpr int D_UNITS = DROWS * (CO / 4);
  constexpr int D_PT = (D_UNITS + DT - 1) / DT;
  // one register set per tile in flight; the producers keep TWO tiles of loads outstanding (one
  // tile per CU in flight left the kernel bound by memory latency)
  struct Stage {
    SU px[X_PT], pd[D_PT];
    f32x4 pa[FA ? D_PT : 1];   // FA: the layer's stored output at the dz units (pd then holds dy)
    bool allin;     // interior tile, full channel slices: every unit of the tile is inside the image (uniform) -- no masking
    unsigned inb;   // units of px that came from inside the image (the others are zero padding)
    unsigned dmask; // same for pd
    int dt;         // FA: element offset of the tile's first pixel in its image (uniform), for the dz store
    size_t dimg;    // FA: element offset of the image (uniform)
    int xbase;      // RINGX: first ring slot of the tile's window (uniform)
    bool xfresh;    // RINGX: the whole window is staged (first tile of a column / of the strip); else its TRW new rows
    int ex0, ey0;   // LEANX: image column / row of the window's first pixel (uniform; may be negative)
    size_t ximg;    // LEANX: element offset of the image in x (uniform)
  };
  const int xc4 = tid % (CI / 4);          // XT % (CI/4) == 0: fixed channel quad per thread
  const int xc = ci0 + xc4 * 4;
  const bool xc_ok = xc < a.x.C;
  f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  if (a.x.scale && xc_ok) {
    psc = *reinterpret_cast<const f32x4*>(a.x.scale + xc);
    psh = *reinterpret_cast<const f32x4*>(a.x.shift + xc);
  }
  bool aff = a.x.scale != nullptr;
  const bool lr = a.x.lrelu != 0;
  f32x4 dsc = {1.f, 1.f, 1.f, 1.f};          // NP == 2: the gradient's per-tensor exponent, applied while dz is staged
  if constexpr (NP == 2) {
    psc = psc * 64.f;                          // x staged times 2^6 (LeakyReLU is positively homogeneous: exact)
    psh = psh * 64.f;
    aff = true;
  }

  // Per-thread constant element offsets of every staged unit relative to the tile's first (halo)
  // pixel: interior tiles (the vast majority) load with a uniform base + these offsets, no
  // per-unit index arithmetic or bounds tests.  Without a halo a pass of 256 threads advances by
  // a uniform pixel offset, so one offset per thread is enough.
  const int xp0 = tid / (CI / 4);
  constexpr int XSTEP = XT / (CI / 4);    // pixels per pass of the producer waves
  const int dc4 = tid % (CO / 4), dp0 = tid / (CO / 4);
  constexpr int DSTEP = DT / (CO / 4);
  static_assert(XSTEP % 32 == 0 || 32 % XSTEP == 0, "pass size must tile the 32-pixel rows");
  static_assert(DSTEP % 32 == 0 || 32 % DSTEP == 0, "pass size must tile the 32-pixel rows");
  const int dc = co0 + dc4 * 4;
  const bool dc_ok = dc + 3 < a.dz_cstride;
  if constexpr (NP == 2) {
    if (a.dz_scale && dc + 3 < a.Cout) dsc = *reinterpret_cast<const f32x4*>(a.dz_scale + dc);
  }
  // FA: BatchNorm-backward coefficients of this thread's four channels (channels beyond Cout give dz = 0), running sums of dz
  f32x4 fk1 = {0.f, 0.f, 0.f, 0.f}, fk2 = fk1, fk3 = fk1, fsum = fk1, fps = fk1, fpsh = fk1;
  // (SPL: the launcher keeps layers with a pre-activation affine on the four-wave form -- eight registers the 128 do not have)
  const bool fpre = FA && !SPL && a.f_ps != nullptr;  // conv -> BatchNorm -> LeakyReLU layer: activation derivative at BN(act) (uniform)
  const bool fwrite = FA && ci0 == 0;                 // one cin slice per cout slice writes dz and the sums (uniform)
  if constexpr (FA) {
    if (dc + 3 < a.Cout) {
      if (a.f_k1) {
        fk1 = *reinterpret_cast<const f32x4*>(a.f_k1 + dc);
        fk2 = *reinterpret_cast<const f32x4*>(a.f_k2 + dc);
        fk3 = *reinterpret_cast<const f32x4*>(a.f_k3 + dc);
        if (fpre) {
          fps = *reinterpret_cast<const f32x4*>(a.f_ps + dc);
          fpsh = *reinterpret_cast<const f32x4*>(a.f_psh + dc);
        }
      } else {
        fk1 = f32x4{1.f, 1.f, 1.f, 1.f};
      }
    }
  }
  // SPL (1024 threads, 128 registers): the per-channel constants of the staging code -- the on-load affine of x, the three
  // BatchNorm-backward coefficients -- live in LDS behind the tile buffers and are read where a phase of store_tile needs
  // them (20 registers that are not held across the tile loop; every thread of a channel quad reads the same 16 bytes)
  constexpr int CONST_OFF = 2 * NP * ((TRW + 2 * HALO) * (32 + 2 * HALO) * CI + TRW * 32 * CO) * 2;     // bytes: both tile buffers
  float* s_const = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + CONST_OFF);                  // [2][CI], then [3][CO]
  if constexpr (SPL) {
    if (producer) {
      if (tid < CI / 4) {
        *reinterpret_cast<f32x4*>(s_const + tid * 4) = psc;
        *reinterpret_cast<f32x4*>(s_const + CI + tid * 4) = psh;
      }
      if (FA && tid < CO / 4) {
        *reinterpret_cast<f32x4*>(s_const + 2 * CI + tid * 4) = fk1;
        *reinterpret_cast<f32x4*>(s_const + 2 * CI + CO + tid * 4) = fk2;
        *reinterpret_cast<f32x4*>(s_const + 2 * CI + 2 * CO + tid * 4) = fk3;
      }
    }
    __syncthreads();
  }
  auto x_affine = [&](f32x4& sc, f32x4& sh) __attribute__((always_inline)) {
    if constexpr (SPL) {
      sc = *reinterpret_cast<const f32x4*>(s_const + xc4 * 4);
      sh = *reinterpret_cast<const f32x4*>(s_const + CI + xc4 * 4);
    } else {
      sc = psc;
      sh = psh;
    }
  };
  unsigned xoff[HALO > 0 ? X_PT : 1];
  if constexpr (HALO > 0) {
#pragma unroll
    for (int i = 0; i < X_PT; ++i) {
      const int pp = xp0 + i * XSTEP;
      xoff[i] = (unsigned)(((pp / TWh) * a.W + pp % TWh) * a.x.cstride + a.x.coff + xc);
    }
  } else {
    xoff[0] = (unsigned)(((xp0 / 32) * a.W + xp0 % 32) * a.x.cstride + a.x.coff + xc);
  }
  const unsigned doff0 = (unsigned)(((dp0 / 32) * a.W + dp0 % 32) * a.dz_cstride + dc);
  const unsigned x_all = xc_ok ? ((X_UNITS % XT == 0 || tid < X_UNITS % XT) ? ((1u << X_PT) - 1u) : ((1u << (X_PT - 1)) - 1u)) : 0u;
  unsigned x_new = 0;        // RINGX: this thread's units that lie in the first TRW rows of the staged rectangle
#pragma unroll
  for (int i = 0; i < X_PT; ++i)
    if (xp0 + i * XSTEP < TRW * TWh) x_new |= 1u << i;

  // coordinates of the next tile to load (producer state, advanced by load_tile)
  int ltx = 0, lty = 0, lb = 0;
  int lxbase = 0;        // RINGX: ring slot of the window of the tile loaded last
  bool lfirst = true;    // RINGX: nothing loaded yet
  // Tiles are walked DOWN the image first (tile index = (image, column of tiles, row)): with a halo the next tile of a
  // strip re-reads 2 * HALO of its TRW + 2 * HALO input rows, and they are the rows the workgroup fetched a tile ago (L2
  // hits); walking along x the vertical neighbour came tiles_x tiles later, from HBM again.  (The order of the tiles only
  // changes the order of the fp32 pixel sums.)
  const bool ymajor = HALO > 0;
  auto seek_tile = [&](int mt) {
    if (ymajor) {
      lty = mt % a.tiles_y;
      ltx = (mt / a.tiles_y) % a.tiles_x;
    } else {
      ltx = mt % a.tiles_x;
      lty = (mt / a.tiles_x) % a.tiles_y;
    }
    lb = mt / (a.tiles_x * a.tiles_y);
  };
  // (always_inline: with four register sets there are ~18 call sites, and a call that is not inlined puts the sets in scratch)
  // every thread's channel quad inside the tensors (uniform): with an interior tile on top no staged unit needs a mask
  const bool full_slices = (ci0 + CI <= a.x.C) && (co0 + CO <= a.dz_cstride);
  // Ablation switches (tools/ablate_wgrad.py; c3d_wgrad_desc.variant & 8: consumer waves idle, & 16: producer waves idle, & 32:
  // every load an L2 hit, & 64: no LDS stores).  ONLY in a build with -DC3D_WGRAD_ABLATE: a uniform branch around the
  // loads of load_tile is enough for the compiler to give up its counted vmcnt waits (6-11 full drains per instance instead
  // of 3) -- measured in the round-5 step: every 1x1 instance 8-26 % slower with the switches compiled in.
#ifdef C3D_WGRAD_ABLATE
  const bool abl_l2 = (a.variant & 32) != 0, abl_nolds = (a.variant & 64) != 0, abl_prod = (a.variant & 16) != 0;
  const bool abl_cons = (a.variant & 8) != 0;
#else
  constexpr bool abl_l2 = false, abl_nolds = false, abl_prod = false, abl_cons = false;
#endif
  // role_tag: 0 = both tensors (every producer wave stages its share of both), 1 = the x units only, 2 = the dz units only (ROLES).
  // A compile-time tag, not a branch: a uniform branch around the loads costs the counted waits (the trap described above).
  auto load_tile = [&](Stage& sg, auto role_tag) __attribute__((always_inline)) {
    constexpr bool DO_X = decltype(role_tag)::value != 2, DO_D = decltype(role_tag)::value != 1;
    if (abl_prod) return;
    const int x0 = ltx * 32, y0 = lty * TRW, b = lb;
    // RINGX: a tile right below its predecessor in the strip stages only its TRW new rows (image rows y0 + HALO ...)
    const bool fresh = !RINGX || lfirst || lty == 0;
    if constexpr (RINGX) {
      lxbase = lfirst ? 0 : (lxbase + (fresh ? THh : TRW)) % RING;
      lfirst = false;
      sg.xbase = lxbase;
      sg.xfresh = fresh;
    }
    const int yx = (fresh && !LEANX) ? y0 - HALO : y0 + HALO;        // image row of the first x row of the register set
    if constexpr (LEANX) {
      sg.ex0 = x0 - HALO;
      sg.ey0 = y0 - HALO;
    }
    if (ymajor) {
      if (++lty == a.tiles_y) {
        lty = 0;
        if (++ltx == a.tiles_x) {
          ltx = 0;
          ++lb;
        }
      }
    } else if (++ltx == a.tiles_x) {
      ltx = 0;
      if (++lty == a.tiles_y) {
        lty = 0;
        ++lb;
      }
    }
    const bool interior = x0 >= HALO && x0 + 32 + HALO <= a.W && y0 >= HALO && y0 + TRW + HALO <= a.H;   // uniform
    // validity of this thread's units: interior tiles (the vast majority) need no pixel tests
    unsigned xmask = (fresh || LEANX) ? x_all : (x_all & x_new), dmask = dc_ok ? ((1u << D_PT) - 1u) : 0u;
    if (!interior) {
      unsigned xm = 0;
#pragma unroll
      for (int i = 0; i < X_PT; ++i) {
        const int pp = xp0 + i * XSTEP;
        const int gx = x0 + pp % TWh - HALO, gy = yx + pp / TWh;
        if (gx >= 0 && gx < a.W && gy >= 0 && gy < a.H) xm |= 1u << i;
      }
      xmask &= xm;
      unsigned dm = 0;
#pragma unroll
      for (int i = 0; i < D_PT; ++i) {
        const int pp = dp0 + i * DSTEP;
        if (x0 + (pp & 31) < a.W && y0 + (pp >> 5) < a.H) dm |= 1u << i;
      }
      dmask &= dm;
    }
    sg.inb = xmask;
    sg.dmask = dmask;
    sg.allin = interior && full_slices;
    // EVERY unit issues its load (invalid ones read element 0 of the image and are zeroed in
    // store_tile): with a fixed number of loads per tile the compiler can wait for the older of
    // the two tiles in flight with s_waitcnt vmcnt(N); loads under per-unit branches made it
    // fall back to vmcnt(0), which serialised the two tiles.
    const size_t ximg = (size_t)b * a.H * a.W * a.x.cstride, dimg = (size_t)b * a.H * a.W * a.dz_cstride;   // uniform
    if constexpr (LEANX) sg.ximg = ximg;
    const int xt = (yx * a.W + (x0 - HALO)) * a.x.cstride;     // uniform; < 0 only where masked
    const int dt = (y0 * a.W + x0) * a.dz_cstride;
    auto load_x = [&](auto bf_tag) {
      constexpr bool XBF = decltype(bf_tag)::value;
      // (RINGX: a tile below its predecessor needs its TRW new rows only, but EVERY unit still issues a load -- the
      //  others read element 0 of the image, an L2 hit: requesting them under a uniform branch measured 3-5 % slower,
      //  the fixed load count is what lets the waits in front of store_tile be counted ones)
#pragma unroll
      for (int i = 0; i < X_PT; ++i) {
        int off;
        if constexpr (HALO > 0) off = xt + (int)xoff[i];
        else off = xt + (((i * XSTEP) / 32) * a.W + (i * XSTEP) % 32) * a.x.cstride + (int)xoff[0];
        const unsigned o2 = ((xmask >> i) & 1u) ? (unsigned)off : 0u;
        if constexpr (RAW) {
          const char* bp = c3d_uniform_ptr(reinterpret_cast<const unsigned short*>(a.x.ptr) + ximg);
          sg.px[i] = __builtin_bit_cast(u32x2, *(const __attribute__((address_space(1))) c3d_u32x2*)(bp + (size_t)(o2 * 2u)));
        } else {
          sg.px[i] = c3d_ld4u<XBF>(a.x.ptr, ximg, abl_l2 ? 0u : o2);
        }
      }
    };
    auto load_dz = [&](auto bf_tag) {
      constexpr bool DBF = decltype(bf_tag)::value;
      if constexpr (FA) {
        sg.dt = dt;
        sg.dimg = dimg;
      }
#pragma unroll
      for (int i = 0; i < D_PT; ++i) {
        const int off = dt + (((i * DSTEP) / 32) * a.W + (i * DSTEP) % 32) * a.dz_cstride + (int)doff0;
        if constexpr (RAW) {
          const char* bp = c3d_uniform_ptr(reinterpret_cast<const unsigned short*>(a.dz) + dimg);
          const unsigned o2 = ((dmask >> i) & 1u) ? (unsigned)off : 0u;
          sg.pd[i] = __builtin_bit_cast(u32x2, *(const __attribute__((address_space(1))) c3d_u32x2*)(bp + (size_t)(o2 * 2u)));
        } else if constexpr (FA) {
          const unsigned o2 = ((dmask >> i) & 1u) ? (unsigned)off : 0u;
          sg.pd[i] = c3d_ld4u<false>(a.f_dy, dimg, o2);
          sg.pa[i] = c3d_ld4u<false>(a.f_act, dimg, o2);
        } else if constexpr (DBF) {
          // bf16 gradient, one plane: the four values go to LDS as they are -- the raw 8 bytes travel in the first two
          // lanes of the register set (no widening here, no rounding back in store_tile: 8 of the ~10 VALU instructions
          // a dz unit cost the producer waves, which bound this mode)
          const char* bp = c3d_uniform_ptr(reinterpret_cast<const unsigned short*>(a.dz) + dimg);
          const unsigned o2 = ((dmask >> i) & 1u) ? (unsigned)off : 0u;
          const c3d_u32x2 r = *(const __attribute__((address_space(1))) c3d_u32x2*)(bp + (size_t)(o2 * 2u));
          sg.pd[i] = f32x4{__uint_as_float(r[0]), __uint_as_float(r[1]), 0.f, 0.f};
        } else {
          sg.pd[i] = c3d_ld4u<DBF>(a.dz, dimg, (((dmask >> i) & 1u) && !abl_l2) ? (unsigned)off : 0u);
        }
      }
    };
    if constexpr (RAW) {
      load_x(std::true_type{});
      load_dz(std::true_type{});
    } else {
      if constexpr (DO_X) {
        if (NP == 1 && a.x.bf16) load_x(std::true_type{});
        else load_x(std::false_type{});
      }
      if constexpr (DO_D) {
        if (NP == 1 && a.dz_bf16) load_dz(std::true_type{});
        else load_dz(std::false_type{});
      }
    }
  };
  auto store_tile = [&](int buf, const Stage& sg, auto role_tag) __attribute__((always_inline)) {
    constexpr bool DO_X = decltype(role_tag)::value != 2, DO_D = decltype(role_tag)::value != 1;
    if (abl_prod) return;
    unsigned short* s_x = RINGX ? s_base : s_base + buf * BUF;
    unsigned short* s_dz = RINGX ? s_base + NP * XPLANE + buf * (NP * DROWS * CO) : s_x + NP * XROWS * CI;
    // RINGX: first ring slot the staged rows go to (a whole window, or the TRW rows behind the rows kept from the tile above)
    const int slot0 = RINGX ? ((sg.xfresh && !LEANX) ? sg.xbase : sg.xbase + 2 * HALO) : 0;
    const int xunits = (RINGX && !sg.xfresh) ? NEWROWS_UNITS : X_UNITS;      // uniform
    // LEANX: the 2 * HALO rows above the new ones (a column's first tile only), in rounds of X_PT units per thread
    f32x4 pe[LEANX ? X_PT : 1];
    unsigned emask = 0;
    auto extra_load = [&](int j0) __attribute__((always_inline)) {
      if constexpr (LEANX) {
        emask = 0;
#pragma unroll
        for (int j = 0; j < X_PT; ++j) {
          const int u = tid + (j0 + j) * XT;
          const int pp = u / (CI / 4);
          const int r = pp / TWh, col = pp - r * TWh;
          const int gx = sg.ex0 + col, gy = sg.ey0 + r;
          const bool in = (j0 + j) < E_PT && u < EXTRA_UNITS && xc_ok && gx >= 0 && gx < a.W && gy >= 0 && gy < a.H;
          if (in) emask |= 1u << j;
          pe[j] = c3d_ld4u<false>(a.x.ptr, sg.ximg, in ? (unsigned)((gy * a.W + gx) * a.x.cstride + a.x.coff + xc) : 0u);
        }
      }
    };
    auto extra_store = [&](int j0) __attribute__((always_inline)) {
      if constexpr (LEANX) {
        f32x4 esc, esh;
        x_affine(esc, esh);
#pragma unroll
        for (int j = 0; j < X_PT; ++j) {
          const int u = tid + (j0 + j) * XT;
          if ((j0 + j) < E_PT && u < EXTRA_UNITS) {
            f32x4 v = pe[j];
            if (aff) v = v * esc + esh;
            if (lr) {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = c3d_lrelu(v[q], a.slope);
            }
            if (!((emask >> j) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
            u32x2 pl[NP];
            split_planes<NP>(v, pl);
            const int pp = u / (CI / 4);
            const int r = pp / TWh;
            int slot = sg.xbase + r;
            slot = slot >= RING ? slot - RING : slot;
            const int o = tr_swz<NSX>(slot * TWh + (pp - r * TWh), xc4 * 4);
            if (!abl_nolds) {
#pragma unroll
              for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(s_x + p * XPLANE + o) = pl[p];
            }
          }
        }
      }
    };
    if constexpr (DO_X) {
    f32x4 xsc, xsh;
    x_affine(xsc, xsh);
#pragma unroll
    for (int i = 0; i < X_PT; ++i) {
      const int u = tid + i * XT;
      if (u < xunits) {
        f32x4 v;
        if constexpr (RAW) {
          v = f32x4{__uint_as_float(sg.px[i][0] << 16), __uint_as_float(sg.px[i][0] & 0xffff0000u), __uint_as_float(sg.px[i][1] << 16),
                    __uint_as_float(sg.px[i][1] & 0xffff0000u)};
        } else {
          v = sg.px[i];
        }
        if (aff) v = v * xsc + xsh;
        if (lr) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = c3d_lrelu(v[q], a.slope);
        }
        if (!sg.allin && !((sg.inb >> i) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x2 pl[NP];
        split_planes<NP>(v, pl);
        int R = u / (CI / 4);                 // pixel of the staged rectangle: row R / TWh, column R % TWh
        if constexpr (RINGX) {
          const int r = R / TWh;
          int slot = slot0 + r;
          slot = slot >= RING ? slot - RING : slot;
          R = slot * TWh + (R - r * TWh);
        }
        const int o = tr_swz<NSX>(R, xc4 * 4);
        if (!abl_nolds) {
#pragma unroll
          for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(s_x + p * XPLANE + o) = pl[p];
        }
      }
    }
    if constexpr (LEANX) {
      if (sg.xfresh) extra_load(0);          // into the registers the x units above have left; converted behind the dz units
    }
    }
    if constexpr (DO_D) {
    f32x4 k1 = fk1, k2 = fk2, k3 = fk3;
    if constexpr (SPL && FA) {
      k1 = *reinterpret_cast<const f32x4*>(s_const + 2 * CI + dc4 * 4);
      k2 = *reinterpret_cast<const f32x4*>(s_const + 2 * CI + CO + dc4 * 4);
      k3 = *reinterpret_cast<const f32x4*>(s_const + 2 * CI + 2 * CO + dc4 * 4);
    }
#pragma unroll
    for (int i = 0; i < D_PT; ++i) {
      const int u = tid + i * DT;
      if (u < D_UNITS) {
        u32x2 pl[NP];
        if constexpr (RAW) {
          const bool in = (sg.dmask >> i) & 1u;
          pl[0] = u32x2{in ? sg.pd[i][0] : 0u, in ? sg.pd[i][1] : 0u};
        } else if constexpr (FA) {
          const bool in = sg.allin || ((sg.dmask >> i) & 1u);
          f32x4 t;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            // the arithmetic of bn_bwd_kernel<true> (bn_ops.hip), same operation order: bit-identical dz
            float dyq = sg.pd[i][q];
            if (fpre) dyq *= (fmaf(sg.pa[i][q], fps[q], fpsh[q]) > 0.f) ? 1.f : a.slope;
            float da = fmaf(k2[q], sg.pa[i][q], fmaf(k1[q], dyq, k3[q]));
            if (!fpre) da *= (sg.pa[i][q] > 0.f) ? 1.f : a.slope;
            t[q] = in ? da : 0.f;
          }
          fsum += t;
          if (fwrite && in) {
            const unsigned off = (unsigned)(sg.dt + (((i * DSTEP) / 32) * a.W + (i * DSTEP) % 32) * a.dz_cstride + (int)doff0);
            float* ob = const_cast<float*>(a.dz) + sg.dimg;
            *reinterpret_cast<f32x4*>(ob + off) = t;
          }
          split_planes<NP>(t, pl);
        } else {
          if (NP == 1 && a.dz_bf16) {       // raw bf16 in the first two lanes (load_dz)
            const bool in = (sg.dmask >> i) & 1u;
            pl[0] = u32x2{in ? __float_as_uint(sg.pd[i][0]) : 0u, in ? __float_as_uint(sg.pd[i][1]) : 0u};
          } else {
            split_planes<NP>((sg.allin || ((sg.dmask >> i) & 1u)) ? (NP == 2 ? sg.pd[i] * dsc : sg.pd[i]) : f32x4{0.f, 0.f, 0.f, 0.f}, pl);
          }
        }
        const int o = tr_swz<NSD>(u / (CO / 4), (u % (CO / 4)) * 4);
        if (!abl_nolds) {
#pragma unroll
          for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(s_dz + p * DROWS * CO + o) = pl[p];
        }
      }
    }
    }
    if constexpr (LEANX && DO_X) {
      if (sg.xfresh) {
        extra_store(0);
#pragma unroll
        for (int j0 = X_PT; j0 < E_PT; j0 += X_PT) {     // (TRW < 2 * HALO: further rounds, load and convert back to back)
          extra_load(j0);
          extra_store(j0);
        }
      }
    }
  };

  int tapoff[TMAX];                        // pixel-row offset of each tap (a.T == TMAX by construction)
#pragma unroll
  for (int t = 0; t < TMAX; ++t) tapoff[t] = a.dy[t] * TWh + a.dx[t];
  int cxbase = 0;                          // RINGX: ring slot of the current tile's window (consumer waves; same walk as load_tile)

  const int t_begin = __builtin_amdgcn_readfirstlane(strip * a.tiles_per_strip);
  const int t_end = __builtin_amdgcn_readfirstlane(min(t_begin + a.tiles_per_strip, a.ntiles));
  // ---- producer waves: stage tile mt+1 while the consumers work on tile mt; they take part in
  //      every barrier of the consumer path below (tile loop + K-split fold) and nothing else
  if (producer) {
   auto produce = [&](auto role_tag) __attribute__((always_inline)) {
   if constexpr (DEPTH == 2) {
    Stage sa, sb;      // tile n of the strip travels through set / LDS buffer (n - t_begin) & 1
    int mt = t_begin;
    if (t_end - t_begin >= 5) {
      // long strips: prologue and steady state carry no condition around a load, so the waits in
      // front of store_tile are s_waitcnt vmcnt(<loads of the younger tile>), not vmcnt(0)
      seek_tile(t_begin);
      load_tile(sa, role_tag);
      load_tile(sb, role_tag);
      store_tile(0, sa, role_tag);
      load_tile(sa, role_tag);
      __syncthreads();
      while (mt + 4 < t_end) {      // two tiles per trip
        store_tile(1, sb, role_tag);
        load_tile(sb, role_tag);
        __syncthreads();
        store_tile(0, sa, role_tag);
        load_tile(sa, role_tag);
        __syncthreads();
        mt += 2;
      }
    } else {
      if (t_begin < t_end) {
        seek_tile(t_begin);
        load_tile(sa, role_tag);
        if (t_begin + 1 < t_end) load_tile(sb, role_tag);
        store_tile(0, sa, role_tag);
        if (t_begin + 2 < t_end) load_tile(sa, role_tag);
      }
      __syncthreads();
    }
    for (; mt < t_end;) {
      // consumers on buffer 0: tile mt+1 -> buffer 1, then tile mt+3 takes its registers
      if (mt + 1 < t_end) {
        store_tile(1, sb, role_tag);
        if (mt + 3 < t_end) load_tile(sb, role_tag);
      }
      __syncthreads();
      if (++mt >= t_end) break;
      if (mt + 1 < t_end) {
        store_tile(0, sa, role_tag);
        if (mt + 3 < t_end) load_tile(sa, role_tag);
      }
      __syncthreads();
      ++mt;
    }
   } else {
    // DEPTH register sets: tile n of the strip (relative index) travels through set n % DEPTH and LDS buffer n & 1; while the
    // consumers work on tile r, tiles r + 1 ... r + DEPTH are loaded or in flight.  Same barrier count as above: one in front
    // of the tile loop, one per tile.
    static_assert(DEPTH % 2 == 0, "the LDS buffer of a tile is its index's parity");
    Stage st[DEPTH];
    int mt = t_begin;
    if (t_end - t_begin > 2 * DEPTH) {
      // long strips: no condition around a load (counted waits, see above)
      seek_tile(t_begin);
      c3d_wg_static_for<0, DEPTH>([&](auto k_tag) { load_tile(st[decltype(k_tag)::value], role_tag); });
      store_tile(0, st[0], role_tag);
      load_tile(st[0], role_tag);
      __syncthreads();
      while (mt + 2 * DEPTH < t_end) {      // DEPTH tiles per trip; the last load of a trip is tile mt + 2 * DEPTH
        c3d_wg_static_for<1, DEPTH + 1>([&](auto j_tag) {
          constexpr int J = decltype(j_tag)::value;
          store_tile(J & 1, st[J % DEPTH], role_tag);
          load_tile(st[J % DEPTH], role_tag);
          __syncthreads();
        });
        mt += DEPTH;
      }
    } else {
      if (t_begin < t_end) {
        seek_tile(t_begin);
        load_tile(st[0], role_tag);
        c3d_wg_static_for<1, DEPTH>([&](auto k_tag) {
          constexpr int K = decltype(k_tag)::value;
          if (t_begin + K < t_end) load_tile(st[K], role_tag);
        });
        store_tile(0, st[0], role_tag);
        if (t_begin + DEPTH < t_end) load_tile(st[0], role_tag);
      }
      __syncthreads();
    }
    // tail (mt - t_begin is a multiple of DEPTH here): the same steps with every store / load under its test
    while (mt < t_end) {
      c3d_wg_static_for<1, DEPTH + 1>([&](auto j_tag) {
        constexpr int J = decltype(j_tag)::value;
        if (mt < t_end) {
          if (mt + 1 < t_end) {
            store_tile(J & 1, st[J % DEPTH], role_tag);
            if (mt + 1 + DEPTH < t_end) load_tile(st[J % DEPTH], role_tag);
          }
          __syncthreads();
          ++mt;
        }
      });
    }
   }
   };
   if constexpr (ROLES) {
     if (xrole) produce(std::integral_constant<int, 1>{});
     else produce(std::integral_constant<int, 2>{});
   } else {
     produce(std::integral_constant<int, 0>{});
   }
    if constexpr (FA) {
      // per-thread sums of dz over the strip: [Cout][2][strips * DSTEP], row 0 (the launch's fold adds them in fp64)
      if (fwrite && (!ROLES || !xrole)) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (dc + q < a.Cout) a.f_sum[((size_t)(dc + q) * 2) * a.f_sum_n + strip * DSTEP + dp0] = fsum[q];
      }
    }
    if constexpr (SPL) {
      for (int t = 0; t < TLS; ++t) {
        __syncthreads();
        __syncthreads();
      }
    } else if (WK > 1) {
      for (int t = 0; t < a.T; ++t) {
        __syncthreads();
        __syncthreads();
      }
    }
    return;
  }

  if constexpr (SPL1) {
    // ---- consumer waves of the 1x1 128 x 256 slice, cout tiles split: wave (wci, wco, half th) owns cin tiles 2 wci, 2 wci + 1 and
    //      cout tiles 4 wco + 2 th, 4 wco + 2 th + 1.  Per accumulator: K steps ascending, the six products in the order of the
    //      four-wave form -- the same bits.  WK = 1: no fold, the accumulators go straight to the partial.
    static_assert(WK == 1 && HALO == 0, "one K group, no halo");
    constexpr int JT = 2;                    // cout tiles of a half
    f32x16 acc[CI_T][JT];
#pragma unroll
    for (int i = 0; i < CI_T; ++i)
#pragma unroll
      for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    constexpr int NQ = 6;
    constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
    auto last_use = [](const int (&pl)[6], int plane) constexpr {
      int l = -1;
      for (int q = 0; q < NQ; ++q)
        if (pl[q] == plane) l = q;
      return l;
    };
    constexpr int NSTAGE = KPW;              // stage = K step
    unsigned lane_a[CI_T], lane_b[JT];
#pragma unroll
    for (int i = 0; i < CI_T; ++i) lane_a[i] = 2u * (unsigned)tr_swz<NSX>(lp, (wci * CI_T + i) * 32 + lc);
#pragma unroll
    for (int j = 0; j < JT; ++j) lane_b[j] = 2u * (unsigned)tr_swz<NSD>(lp, (wco * CO_T + th * JT + j) * 32 + lc);
    __syncthreads();
    for (int mt = t_begin; mt < t_end; ++mt) {
      const int cur = (mt - t_begin) & 1;
      const unsigned short* s_x = s_base + cur * BUF;
      const unsigned short* s_dz = s_x + NP * XROWS * CI;
      const unsigned lds_x = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned short*)s_x;
      const unsigned lds_d = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned short*)s_dz;
      bf16x8 ap[CI_T][NP], bp[JT][NP];
      auto read_a = [&](int st, int p) {
        const int ks = st;                   // (wk = 0)
        const int U = (ks >> 1) * 32 + (ks & 1) * 16;
        const unsigned ub = lds_x + 2u * (unsigned)(p * XPLANE + U * (32 * NSX));
#pragma unroll
        for (int i = 0; i < CI_T; ++i) ap[i][p] = tr_frag_u<NSX>(ub, lane_a[i]);
      };
      auto read_b = [&](int st, int p) {
        const int ks = st;
        const int U = (ks >> 1) * 32 + (ks & 1) * 16;
        const unsigned ub = lds_d + 2u * (unsigned)(p * DROWS * CO + U * (32 * NSD));
#pragma unroll
        for (int j = 0; j < JT; ++j) bp[j][p] = tr_frag_u<NSD>(ub, lane_b[j]);
      };
      if (!abl_cons) {
      c3d_wg_static_for<0, NQ>([&](auto q_tag) {
        constexpr int q = decltype(q_tag)::value;
        bool fa = true, fb = true;
        for (int r = 0; r < q; ++r) {
          if (PA[r] == PA[q]) fa = false;
          if (PB[r] == PB[q]) fb = false;
        }
        if (fa) read_a(0, PA[q]);
        if (fb) read_b(0, PB[q]);
      });
      __builtin_amdgcn_sched_barrier(0);
      c3d_wg_static_for<0, NSTAGE * NQ>([&](auto s_tag) {
        constexpr int sq = decltype(s_tag)::value, st = sq / NQ, q = sq % NQ;
#pragma unroll
        for (int i = 0; i < CI_T; ++i)
#pragma unroll
          for (int j = 0; j < JT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][PA[q]], bp[j][PB[q]], acc[i][j], 0, 0, 0);
        if constexpr (st + 1 < NSTAGE) {
          if constexpr (last_use(PA, PA[q]) == q) read_a(st + 1, PA[q]);
          if constexpr (last_use(PB, PB[q]) == q) read_b(st + 1, PB[q]);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      }
      __syncthreads();   // next buffer written, this one no longer read
    }
    const size_t slice_floats = (size_t)a.T * CI * CO;
    float* pout = a.partial + ((size_t)(sl * a.strips + strip)) * slice_floats;
#pragma unroll
    for (int i = 0; i < CI_T; ++i)
#pragma unroll
      for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ci = (wci * CI_T + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const int co = (wco * CO_T + th * JT + j) * 32 + l31;
          pout[(size_t)ci * CO + co] = acc[i][j][r];
        }
    return;
  } else if constexpr (SPL4) {
    // ---- consumer waves of the four-tap 64 x 64 slice, taps split: half th owns taps 2 th, 2 th + 1; wave (wci, wk) its 32 cin
    //      and both 32-cout tiles.  Per accumulator the K steps ascend and the six products come in the order of the four-wave
    //      form (stage = K step x local tap; the two cout tiles of a stage are independent MFMA chains): the same bits.
    constexpr int TL = 2;
    f32x16 acc[TL][CO_T];
#pragma unroll
    for (int t = 0; t < TL; ++t)
#pragma unroll
      for (int j = 0; j < CO_T; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
    int ldy[TL], ldx[TL];
#pragma unroll
    for (int t = 0; t < TL; ++t) {
      ldy[t] = th ? a.dy[2 + t] : a.dy[t];
      ldx[t] = th ? a.dx[2 + t] : a.dx[t];
    }
    constexpr int NQ = 6;
    constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};      // as the four-wave form: smallest products first
    auto last_use = [](const int (&pl)[6], int plane) constexpr {
      int l = -1;
      for (int q = 0; q < NQ; ++q)
        if (pl[q] == plane) l = q;
      return l;
    };
    constexpr int NSTAGE = KPW * TL;
    const unsigned lane_a = 2u * (unsigned)tr_swz<NSX>(lp, wci * 32 + lc);
    unsigned lane_b[CO_T];
#pragma unroll
    for (int j = 0; j < CO_T; ++j) lane_b[j] = 2u * (unsigned)tr_swz<NSD>(lp, j * 32 + lc);
    // (64-channel rows are swizzled by the pixel row: the per-lane part of a fragment address is constant only where the
    //  uniform row offset is a multiple of 4 -- always for dz; for x the tap shifts are not, so x fragments take tr_frag)
    __syncthreads();
    for (int mt = t_begin; mt < t_end; ++mt) {
      const int cur = (mt - t_begin) & 1;
      if constexpr (RINGX) {
        if (mt != t_begin) cxbase = __builtin_amdgcn_readfirstlane((cxbase + ((mt % a.tiles_y == 0) ? THh : TRW)) % RING);
      }
      const unsigned short* s_x = RINGX ? s_base : s_base + cur * BUF;
      const unsigned short* s_dz = RINGX ? s_base + NP * XPLANE + cur * (NP * DROWS * CO) : s_x + NP * XROWS * CI;
      const unsigned lds_d = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned short*)s_dz;
      bf16x8 ap[NP], bp[CO_T][NP];
      auto read_a = [&](int st, int p) {
        const int ks = wk * KPW + st / TL, tl = st % TL;
        int Ux;
        if constexpr (RINGX) {
          int slot = cxbase + (ks >> 1) + HALO + ldy[tl];
          slot = slot >= RING ? slot - RING : slot;
          Ux = slot * TWh + HALO + (ks & 1) * 16 + ldx[tl];
        } else {
          Ux = ((ks >> 1) + HALO + ldy[tl]) * TWh + HALO + (ks & 1) * 16 + ldx[tl];
        }
        ap[p] = tr_frag<NSX>(s_x + p * XPLANE, Ux + lp, wci * 32 + lc);
      };
      auto read_b = [&](int st, int p) {
        const int ks = wk * KPW + st / TL;
        const int Ud = (ks >> 1) * 32 + (ks & 1) * 16;
        const unsigned ub = lds_d + 2u * (unsigned)(p * DROWS * CO + Ud * (32 * NSD));
#pragma unroll
        for (int j = 0; j < CO_T; ++j) bp[j][p] = tr_frag_u<NSD>(ub, lane_b[j]);
      };
      (void)lane_a;
      if (!abl_cons) {
      c3d_wg_static_for<0, NQ>([&](auto q_tag) {
        constexpr int q = decltype(q_tag)::value;
        bool fa = true, fb = true;
        for (int r = 0; r < q; ++r) {
          if (PA[r] == PA[q]) fa = false;
          if (PB[r] == PB[q]) fb = false;
        }
        if (fa) read_a(0, PA[q]);
        if (fb) read_b(0, PB[q]);
      });
      __builtin_amdgcn_sched_barrier(0);
      c3d_wg_static_for<0, NSTAGE * NQ>([&](auto s_tag) {
        constexpr int sq = decltype(s_tag)::value, st = sq / NQ, q = sq % NQ, tl = st % TL;
#pragma unroll
        for (int j = 0; j < CO_T; ++j) acc[tl][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[PA[q]], bp[j][PB[q]], acc[tl][j], 0, 0, 0);
        if constexpr (st + 1 < NSTAGE) {
          constexpr int n1 = st + 1;
          if constexpr (last_use(PA, PA[q]) == q) read_a(n1, PA[q]);
          if constexpr (n1 % TL == 0 && last_use(PB, PB[q]) == q) read_b(n1, PB[q]);      // the K step changes
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      }
      __syncthreads();   // next buffer written, this one no longer read
    }
    // ---- fold the WK pixel groups of every tap through LDS (rank order, as the four-wave form); same partial layout
    const size_t slice_floats = (size_t)a.T * CI * CO;
    float* pout = a.partial + ((size_t)(sl * a.strips + strip)) * slice_floats;
    float* red = smem;   // [consumer wave 0..7][cout tile][16][64]
#pragma unroll
    for (int tl = 0; tl < TL; ++tl) {
      const int t = 2 * th + tl;
      __syncthreads();
      if (wk > 0) {
#pragma unroll
        for (int j = 0; j < CO_T; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((wave8 * CO_T + j) * 16 + r) * 64 + lane] = acc[tl][j][r];
      }
      __syncthreads();
      if (wk == 0 && t < a.T) {
#pragma unroll
        for (int j = 0; j < CO_T; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[tl][j][r];
#pragma unroll
            for (int k = 1; k < WK; ++k) v += red[(((th * 4 + k * WCI + wci) * CO_T + j) * 16 + r) * 64 + lane];
            const int ci = wci * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int co = j * 32 + l31;
            pout[((size_t)t * CI + ci) * CO + co] = v;
          }
      }
    }
    return;
  } else if constexpr (SPL) {
    // ---- consumer waves, taps split (see NCW at the head of the kernel): half 0 owns taps 0-4, half 1 taps 5-8
    constexpr int TL = 5;
    f32x16 acc[TL];
#pragma unroll
    for (int t = 0; t < TL; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    int ldy[TL], ldx[TL];                    // (uniform: two scalar loads and a select each; half 1 never uses its fifth)
#pragma unroll
    for (int t = 0; t < TL; ++t) {
      ldy[t] = (th && t < 4) ? a.dy[5 + t] : a.dy[t];
      ldx[t] = (th && t < 4) ? a.dx[5 + t] : a.dx[t];
    }
    constexpr int NQ = NP == 3 ? 6 : 1;      // (one plane: the one product)
    constexpr int PA[6] = {NP == 3 ? 1 : 0, NP == 3 ? 2 : 0, 0, NP == 3 ? 1 : 0, 0, 0};      // as below: smallest products first
    constexpr int PB[6] = {NP == 3 ? 1 : 0, 0, NP == 3 ? 2 : 0, 0, NP == 3 ? 1 : 0, 0};
    auto last_use = [](const int (&pl)[6], int plane) constexpr {
      int l = -1;
      for (int q = 0; q < NQ; ++q)
        if (pl[q] == plane) l = q;
      return l;
    };
    constexpr int NSTAGE = KPW * TL;         // stage = (K step, local tap); half 1 skips the fifth tap of every K step
    const unsigned lane_a = 2u * (unsigned)tr_swz<NSX>(lp, lc);
    const unsigned lane_b = 2u * (unsigned)tr_swz<NSD>(lp, wco * 32 + lc);
    __syncthreads();
    for (int mt = t_begin; mt < t_end; ++mt) {
      const int cur = (mt - t_begin) & 1;
      if constexpr (RINGX) {
        if (mt != t_begin) cxbase = __builtin_amdgcn_readfirstlane((cxbase + ((mt % a.tiles_y == 0) ? THh : TRW)) % RING);
      }
      const unsigned short* s_x = RINGX ? s_base : s_base + cur * BUF;
      const unsigned short* s_dz = RINGX ? s_base + NP * XPLANE + cur * (NP * DROWS * CO) : s_x + NP * XROWS * CI;
      const unsigned lds_x = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned short*)s_x;
      const unsigned lds_d = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned short*)s_dz;
      bf16x8 ap[NP], bp[NP];
      auto read_a = [&](int st, int p) {
        const int ks = wk * KPW + st / TL, tl = st % TL;
        int Ux;
        if constexpr (RINGX) {
          int slot = cxbase + (ks >> 1) + HALO + ldy[tl];
          slot = slot >= RING ? slot - RING : slot;
          Ux = slot * TWh + HALO + (ks & 1) * 16 + ldx[tl];
        } else {
          Ux = ((ks >> 1) + HALO + ldy[tl]) * TWh + HALO + (ks & 1) * 16 + ldx[tl];
        }
        ap[p] = tr_frag_u<NSX>(lds_x + 2u * (unsigned)(p * XPLANE + Ux * (32 * NSX)), lane_a);
      };
      auto read_b = [&](int st, int p) {
        const int ks = wk * KPW + st / TL;
        const int Ud = (ks >> 1) * 32 + (ks & 1) * 16;
        bp[p] = tr_frag_u<NSD>(lds_d + 2u * (unsigned)(p * DROWS * CO + Ud * (32 * NSD)), lane_b);
      };
      c3d_wg_static_for<0, NQ>([&](auto q_tag) {
        constexpr int q = decltype(q_tag)::value;
        bool fa = true, fb = true;
        for (int r = 0; r < q; ++r) {
          if (PA[r] == PA[q]) fa = false;
          if (PB[r] == PB[q]) fb = false;
        }
        if (fa) read_a(0, PA[q]);
        if (fb) read_b(0, PB[q]);
      });
      __builtin_amdgcn_sched_barrier(0);
      // The same accumulation order per tap as the four-wave form (K steps ascending, the six products in this order): the
      // same bits.  Operand planes of the next stage are read right after their last product of this one.
      if (!abl_cons)       // (constant false outside a -DC3D_WGRAD_ABLATE build)
      c3d_wg_static_for<0, NSTAGE * NQ>([&](auto s_tag) {
        constexpr int sq = decltype(s_tag)::value, st = sq / NQ, q = sq % NQ, tl = st % TL;
        if (tl < 4 || th == 0) {             // (uniform; a real branch for the fifth tap only)
          acc[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[PA[q]], bp[PB[q]], acc[tl], 0, 0, 0);
          if constexpr (st + 1 < NSTAGE) {
            constexpr int n1 = st + 1;
            if constexpr (n1 % TL == 4) {    // the fifth tap comes next: half 0 runs it, half 1 goes on to the next K step
              if constexpr (last_use(PA, PA[q]) == q) {
                if (!th) read_a(n1, PA[q]);
                else if constexpr (st + 2 < NSTAGE) read_a(st + 2, PA[q]);
              }
              if constexpr (last_use(PB, PB[q]) == q && st + 2 < NSTAGE) {
                if (th) read_b(st + 2, PB[q]);
              }
            } else {
              if constexpr (last_use(PA, PA[q]) == q) read_a(n1, PA[q]);
              if constexpr (n1 % TL == 0 && last_use(PB, PB[q]) == q) read_b(n1, PB[q]);      // the K step changes
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      __syncthreads();   // next buffer written, this one no longer read
    }
    // ---- fold the WK pixel groups of every tap through LDS (rank order, as below); partial layout as below
    const size_t slice_floats = (size_t)a.T * CI * CO;
    float* pout = a.partial + ((size_t)(sl * a.strips + strip)) * slice_floats;
    float* red = smem;   // [consumer wave 0..7][16][64]
#pragma unroll
    for (int tl = 0; tl < TL; ++tl) {
      const int t = th ? 5 + tl : tl;        // (half 1, tl = 4: no such tap -- it only keeps the barriers)
      __syncthreads();
      if (wk > 0 && t < 9) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave8 * 16 + r) * 64 + lane] = acc[tl][r];
      }
      __syncthreads();
      if (wk == 0 && t < 9) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[tl][r];
#pragma unroll
          for (int k = 1; k < WK; ++k) v += red[((th * 4 + k * WCO + wco) * 16 + r) * 64 + lane];
          const int ci = (r & 3) + 8 * (r >> 2) + 4 * half;
          const int co = wco * 32 + l31;
          pout[((size_t)t * CI + ci) * CO + co] = v;
        }
      }
    }
    return;
  }

  // ---- consumer waves
  // (round 5: s_setprio 1 / 3 for these waves measured nothing on fifteen layer shapes, fused and plain -- the consumers do not
  //  lose issue slots to the producer waves; what they wait for is the producers' tile)
  // (tried: 32-wide channel tiles dealt to the waves interleaved, dead tiles of a ragged last slice skipped under
  //  wave-uniform tests -- 704 = 5.5 x 128 = 2.75 x 256 leaves 16 % dead MFMAs.  The scalar branches around the
  //  MFMAs cost every instance more than the dead work: 704x704 layer 1.97 -> 2.11 ms, the step 197 -> 191 img/s.)
  f32x16 acc[TMAX][CI_T][CO_T];
#pragma unroll
  for (int t = 0; t < TMAX; ++t)
#pragma unroll
    for (int i = 0; i < CI_T; ++i)
#pragma unroll
      for (int j = 0; j < CO_T; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;

  __syncthreads();
  for (int mt = t_begin; mt < t_end; ++mt) {
    const int cur = (mt - t_begin) & 1;
    if constexpr (RINGX) {
      if (mt != t_begin) cxbase = __builtin_amdgcn_readfirstlane((cxbase + ((mt % a.tiles_y == 0) ? THh : TRW)) % RING);
    }
    {
    const unsigned short* s_x = RINGX ? s_base : s_base + cur * BUF;
    const unsigned short* s_dz = RINGX ? s_base + NP * XPLANE + cur * (NP * DROWS * CO) : s_x + NP * XROWS * CI;
    // Consecutive MFMAs must write DIFFERENT accumulators (a chain of dependent v_mfma_f32_32x32x16_bf16 issues at
    // 20.7 ns per instruction, four interleaved chains at 15.8: tools/probes/mfma_chain_probe.hip), so a STAGE = TG taps
    // x JG cout tiles x all cin tiles shares each plane pair: 3-4 accumulators in rotation.
    //
    // Round 3: the stages of a tile (K steps x cout-tile groups x tap groups) run as one software-pipelined stream.
    // Every operand plane of stage s+1 is read right after its LAST product of stage s (the product order opens a stage
    // with planes other than the ones the previous stage closed with), pinned by sched_barrier: the transposed reads
    // of a stage fly under the MFMAs of the previous one, and their latency is exposed once per tile instead of once
    // per stage -- with 18-24 MFMAs per stage that was a quarter of the consumer waves' time (round 2 tried the same
    // prefetch with all of a stage's fragments in a second register set: no gain, because the producer waves' packed-f32
    // VALU was taking the matrix pipe's issue slots, profiles/round3_coissue_probe.md; this file is now compiled
    // without packed ops).
    constexpr int JG = CO_T >= 2 ? 2 : 1;                                   // cout tiles per stage
    constexpr int TG = TMAX == 1 ? 1 : (TMAX % 3 == 0 ? 3 : (4 / (JG * CI_T) > 1 ? 4 / (JG * CI_T) : 1));   // taps per stage
    static_assert(TMAX % TG == 0 && CO_T % JG == 0, "tap / cout-tile groups must divide the loops");
    constexpr int NJ0 = CO_T / JG, NT0 = TMAX / TG;
    constexpr int NSTAGE = KPW * NJ0 * NT0;
    // plane products in issue order (A plane, B plane): NP = 3: six of nine, smallest class first (m*m, l*h, h*l:
    // 2^-16; m*h, h*m: 2^-8; h*h); NP = 1: the one product
    constexpr int NQ = NP == 3 ? 6 : (NP == 2 ? 3 : 1);
    // (NP == 2: L*H', H*H', H*L' -- no plane both closes a stage and opens the next)
    constexpr int PA[6] = {NP >= 2 ? 1 : 0, NP == 3 ? 2 : 0, 0, 1, 0, 0}, PB[6] = {NP == 3 ? 1 : 0, 0, NP == 3 ? 2 : 1, 0, 1, 0};
    auto last_use = [](const int (&pl)[6], int plane) constexpr {
      int l = -1;
      for (int q = 0; q < NQ; ++q)
        if (pl[q] == plane) l = q;
      return l;
    };
    bf16x8 ap[TG][CI_T][NP];
    bf16x8 bp[JG][NP];
    // Fragment addresses as (uniform byte offset) + (per-lane constant).  The row index of a fragment is U + lp with U uniform;
    // where U is a multiple of 4 (no halo: U = row * 32 + 16 * (ks & 1); always for dz) or the rows are unswizzled (one
    // 64-byte segment per row: the 32-channel cin slices of every nine-tap instance) the swizzled offset of lane (lp, c)
    // does not depend on U, and the per-tile / per-tap part is scalar arithmetic.
    constexpr bool A_SPLIT = HALO == 0 || NSX == 1;
    const unsigned lds_x = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned short*)s_x;
    const unsigned lds_d = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned short*)s_dz;
    unsigned lane_a[CI_T], lane_b[CO_T];
#pragma unroll
    for (int i = 0; i < CI_T; ++i) lane_a[i] = 2u * (unsigned)tr_swz<NSX>(lp, (wci * CI_T + i) * 32 + lc);
#pragma unroll
    for (int j = 0; j < CO_T; ++j) lane_b[j] = 2u * (unsigned)tr_swz<NSD>(lp, (wco * CO_T + j) * 32 + lc);
    // (uniform parts of the fragments' pixel rows; the lane adds lp)
    auto stage_rows = [&](int st, int& Ud, int& Ux0) {
      const int kk = st / (NJ0 * NT0);
      const int ks = wk * KPW + kk;
      const int row = ks >> 1, px0 = (ks & 1) * 16;
      Ud = row * 32 + px0;
      Ux0 = (row + HALO) * TWh + HALO + px0;
    };
    auto read_a = [&](int st, int p) {
      int Ud, Ux0;
      stage_rows(st, Ud, Ux0);
      const int t0 = (st % NT0) * TG;
#pragma unroll
      for (int tg = 0; tg < TG; ++tg) {
        int Ux = Ux0 + tapoff[t0 + tg];
        if constexpr (RINGX) {
          // window row of this tap's fragment -> ring slot (uniform arithmetic; the 8 pixels of a fragment share a row)
          const int kk = st / (NJ0 * NT0), ks = wk * KPW + kk;
          int slot = cxbase + (ks >> 1) + HALO + a.dy[t0 + tg];
          slot = slot >= RING ? slot - RING : slot;
          Ux = slot * TWh + HALO + (ks & 1) * 16 + a.dx[t0 + tg];
        }
        if constexpr (A_SPLIT) {
          const unsigned ub = lds_x + 2u * (unsigned)(p * XPLANE + Ux * (32 * NSX));      // uniform
#pragma unroll
          for (int i = 0; i < CI_T; ++i) ap[tg][i][p] = tr_frag_u<NSX>(ub, lane_a[i]);
        } else {
#pragma unroll
          for (int i = 0; i < CI_T; ++i)
            ap[tg][i][p] = tr_frag<NSX>(s_x + p * XPLANE, Ux + lp, (wci * CI_T + i) * 32 + lc);
        }
      }
    };
    auto read_b = [&](int st, int p) {
      int Ud, Ux0;
      stage_rows(st, Ud, Ux0);
      const int j0 = ((st / NT0) % NJ0) * JG;
      const unsigned ub = lds_d + 2u * (unsigned)(p * DROWS * CO + Ud * (32 * NSD));        // uniform
#pragma unroll
      for (int jg = 0; jg < JG; ++jg) bp[jg][p] = tr_frag_u<NSD>(ub, lane_b[j0 + jg]);
    };
    if (!abl_cons) {
    // stage 0: every plane, in the order the products need them
    c3d_wg_static_for<0, NQ>([&](auto q_tag) {
      constexpr int q = decltype(q_tag)::value;
      bool fa = true, fb = true;
      for (int r = 0; r < q; ++r) {
        if (PA[r] == PA[q]) fa = false;
        if (PB[r] == PB[q]) fb = false;
      }
      if (fa) read_a(0, PA[q]);
      if (fb) read_b(0, PB[q]);
    });
    __builtin_amdgcn_sched_barrier(0);
    c3d_wg_static_for<0, NSTAGE * NQ>([&](auto s_tag) {
      constexpr int sq = decltype(s_tag)::value, st = sq / NQ, q = sq % NQ;
      constexpr int t0 = (st % NT0) * TG, j0 = ((st / NT0) % NJ0) * JG;
#pragma unroll
      for (int tg = 0; tg < TG; ++tg)
#pragma unroll
        for (int i = 0; i < CI_T; ++i)
#pragma unroll
          for (int jg = 0; jg < JG; ++jg)
            acc[t0 + tg][i][j0 + jg] =
                NP == 2 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, ap[tg][i][PA[q]]),
                                                                 __builtin_bit_cast(f16x8_t, bp[jg][PB[q]]), acc[t0 + tg][i][j0 + jg], 0, 0, 0)
                        : __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[tg][i][PA[q]], bp[jg][PB[q]], acc[t0 + tg][i][j0 + jg], 0, 0, 0);
      if constexpr (st + 1 < NSTAGE) {
        constexpr int n = st + 1;
        // A changes with the tap group or the K step, B with the cout group or the K step
        constexpr bool a_new = (n % NT0) != (st % NT0) || n / (NJ0 * NT0) != st / (NJ0 * NT0);
        constexpr bool b_new = ((n / NT0) % NJ0) != ((st / NT0) % NJ0) || n / (NJ0 * NT0) != st / (NJ0 * NT0);
        if constexpr (a_new && last_use(PA, PA[q]) == q) read_a(n, PA[q]);
        if constexpr (b_new && last_use(PB, PB[q]) == q) read_b(n, PB[q]);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    }
    }
    __syncthreads();   // next buffer written, this one no longer read
  }
  // ---- fold the WK pixel groups through LDS, write the workgroup partial
  //      partial layout per (slice, strip): [t][cin (slice-local CI)][cout (slice-local CO)]
  const size_t slice_floats = (size_t)a.T * CI * CO;
  float* pout = a.partial + ((size_t)(sl * a.strips + strip)) * slice_floats;
  float* red = smem;  // [WK-1][WCI*WCO][CI_T*CO_T][16][64], one tap at a time (the tile loop ended on a barrier)
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

template <int NP, int TMAX, int CI_T, int CO_T, int WCI, int WCO, int TRW, int HALO, bool LEAN_FA = false, int NPW = 4, int NCW = 4>
int launch_tr(const WgradArgs& a, hipStream_t st) {
  constexpr int NPT = 64 * NPW;
  static_assert(NPW == 4 || (((NP == 3 || NP == 1) && HALO == 0 && !LEAN_FA && NCW == 4) || ((NP == 3 || NP == 1) && TMAX == 9 && NCW == 8) ||
                             (NP == 3 && TMAX == 4 && NCW == 8) || (NP == 3 && TMAX == 1 && NCW == 8)),
                "eight producer waves: the three-plane 1x1 instances, the nine-tap ones with split consumers, the fused four-tap 64 x 64 one");
  constexpr int WK = 4 / (WCI * WCO);
  constexpr int CI = 32 * CI_T * WCI, CO = 32 * CO_T * WCO;
  size_t lds = 2 * (size_t)NP * ((size_t)(TRW + 2 * HALO) * (32 + 2 * HALO) * CI + (size_t)TRW * 32 * CO) * 2;   // two tile buffers
  const size_t red = NCW == 8 ? (size_t)8 * CO_T * 1024 * sizeof(float) : (size_t)(WK - 1) * WCI * WCO * CI_T * CO_T * 1024 * sizeof(float);
  if (NCW == 8) lds += (size_t)(2 * CI + 3 * CO) * sizeof(float);      // the staging constants behind the tile buffers
  if (red > lds) lds = red;
  dim3 grid(a.strips * a.ci_slices * a.co_slices);
  if constexpr (NP == 3 && NCW == 4) {      // (split consumers: unfused launches only, see launch_tr_id)
    if (a.f_dy) {       // BatchNorm / LeakyReLU backward on load
      if (a.f_sum_n != a.strips * (NPT / (CO / 4))) {
        c3d_set_error("wgrad: fuse_sum was not sized with c3d_wgrad_fused_sum_n()");
        return 1;
      }
      if constexpr (LEAN_FA && HALO > 0) {
        // Lean register sets for the fused instances that spilled with whole-window sets (tools/obj_resources.py: 50-553 scratch
        // instructions, 45-356 full drains): a column's first tile loads its upper rows inside its own staging step, which
        // exposes part of a memory latency once per column -- taken where a column has several tiles.  Measured (round 5,
        // 8 x 64 x 2048, fused): 32 -> 32 3x3 d2 0.179 -> 0.158 ms, 64 -> 64 2x2 0.312 -> 0.287, 128 -> 128 2x2 at half
        // resolution 0.263 -> 0.244; the instances that did not spill lose 3-7 % in this form and keep the old one.
        // (c3d_wgrad_desc.variant & 3: 1 / 2 force one form -- tests/test_gpu_conv.py holds the two to the same bits.)
        const int force = a.variant & 3;
        if (force ? force == 2 : a.tiles_y >= C3D_WGRAD_LEAN_MIN_TILES_Y) {
          c3d_opt_in_lds<&wgrad_tr_kernel<NP, TMAX, CI_T, CO_T, WCI, WCO, TRW, HALO, true, false, true, NPW, NCW>>();
          hipLaunchKernelGGL((wgrad_tr_kernel<NP, TMAX, CI_T, CO_T, WCI, WCO, TRW, HALO, true, false, true, NPW, NCW>), grid, dim3(64 * (NCW + NPW)), lds, st, a);
          C3D_CHECK_LAUNCH();
          return 0;
        }
      }
      c3d_opt_in_lds<&wgrad_tr_kernel<NP, TMAX, CI_T, CO_T, WCI, WCO, TRW, HALO, true, false, false, NPW, NCW>>();
      hipLaunchKernelGGL((wgrad_tr_kernel<NP, TMAX, CI_T, CO_T, WCI, WCO, TRW, HALO, true, false, false, NPW, NCW>), grid, dim3(64 * (NCW + NPW)), lds, st, a);
      C3D_CHECK_LAUNCH();
      return 0;
    }
  }
  if constexpr (NP == 3 && NCW == 8 && TMAX != 1) {
    if (a.f_dy) {       // BatchNorm / LeakyReLU backward on load, producer waves split by tensor (ROLES at the kernel)
      if (a.f_sum_n != a.strips * ((NPT / 2) / (CO / 4))) {
        c3d_set_error("wgrad: fuse_sum was not sized with c3d_wgrad_fused_sum_n()");
        return 1;
      }
      // (LEAN_FA: the 32-cout instance with a two-pixel halo -- nine whole-window x units per thread in flight twice spill
      //  37 registers inside the x waves' tile loop; with the lean sets none)
      constexpr bool LN = LEAN_FA && HALO > 0;
      c3d_opt_in_lds<&wgrad_tr_kernel<NP, TMAX, CI_T, CO_T, WCI, WCO, TRW, HALO, true, false, LN, NPW, NCW>>();
      hipLaunchKernelGGL((wgrad_tr_kernel<NP, TMAX, CI_T, CO_T, WCI, WCO, TRW, HALO, true, false, LN, NPW, NCW>), grid, dim3(64 * (NCW + NPW)), lds, st, a);
      C3D_CHECK_LAUNCH();
      return 0;
    }
  }
  if constexpr (NP == 1) {
    if (a.x.bf16 && a.dz_bf16 && !a.f_dy) {       // bf16 tensors on both sides: raw stages, four tiles in flight
      c3d_opt_in_lds<&wgrad_tr_kernel<NP, TMAX, CI_T, CO_T, WCI, WCO, TRW, HALO, false, true, false, NPW, NCW>>();
      hipLaunchKernelGGL((wgrad_tr_kernel<NP, TMAX, CI_T, CO_T, WCI, WCO, TRW, HALO, false, true, false, NPW, NCW>), grid, dim3(64 * (NCW + NPW)), lds, st, a);
      C3D_CHECK_LAUNCH();
      return 0;
    }
  }
  c3d_opt_in_lds<&wgrad_tr_kernel<NP, TMAX, CI_T, CO_T, WCI, WCO, TRW, HALO, false, false, false, NPW, NCW>>();
  hipLaunchKernelGGL((wgrad_tr_kernel<NP, TMAX, CI_T, CO_T, WCI, WCO, TRW, HALO, false, false, false, NPW, NCW>), grid, dim3(64 * (NCW + NPW)), lds, st, a);
  C3D_CHECK_LAUNCH();
  return 0;
}

// tile rows (TRW) must match c3d_wgrad_cfg(..., planes != 0)
template <int NP>
int launch_tr_id(int id, int halo, const WgradArgs& a, hipStream_t st) {
  switch (id) {
    //                         TMAX CI_T CO_T WCI WCO TRW HALO
    case 0:
      // (round 6: unfused launches of the three-plane engine on sixteen waves, cout tiles split across the consumer halves; plan())
      if constexpr (NP == 3) if (a.npw == 8 && !a.f_dy) return launch_tr<NP, 1, 2, 4, 2, 2, 1, 0, false, 8, 8>(a, st);
      return launch_tr<NP, 1, 2, 4, 2, 2, 1, 0>(a, st);
    case 1:
      if constexpr (NP == 3 || NP == 1) if (a.npw == 8) return launch_tr<NP, 1, 2, 2, 2, 2, 1, 0, false, 8>(a, st);
      return launch_tr<NP, 1, 2, 2, 2, 2, 1, 0>(a, st);
    case 2:
      if constexpr (NP == 3 || NP == 1) if (a.npw == 8) return launch_tr<NP, 1, 2, 2, 1, 1, 2, 0, false, 8>(a, st);
      return launch_tr<NP, 1, 2, 2, 1, 1, 2, 0>(a, st);
    case 3:
      if constexpr (NP == 3 || NP == 1) if (a.npw == 8) return launch_tr<NP, 1, 1, 1, 1, 1, 4, 0, false, 8>(a, st);
      return launch_tr<NP, 1, 1, 1, 1, 1, 4, 0>(a, st);
    case 4: return halo <= 1 ? launch_tr<NP, 4, 1, 2, 1, 1, (NP == 1 ? 4 : 2), 1>(a, st) : launch_tr<NP, 4, 1, 2, 1, 1, (NP == 1 ? 4 : 2), 2>(a, st);
    case 5: return halo <= 1 ? launch_tr<NP, 4, 1, 1, 1, 1, 4, 1>(a, st) : launch_tr<NP, 4, 1, 1, 1, 1, 4, 2, NP == 3>(a, st);
    case 8:
      // (halo <= 1 only: c3d_wgrad_cfg never hands this slice a two-pixel halo -- its two tile buffers would exceed the LDS; the
      //  instances that existed for it until round 5 were dead code, and the library's worst spillers)
      if constexpr (NP >= 2) {
        // (round 6: sixteen waves, taps split 2 + 2; a fused launch with its producer waves split by tensor, an unfused one -- a
        //  layer's second and later sources, the second-stream mode, direct API use; NOT the data-parallel step, which has run the
        //  same fused launches as the plain step since round 5 put it on one stream -- with eight producer waves of both tensors;
        //  plan() hands them npw = 8)
        if constexpr (NP == 3) if (halo <= 1 && a.npw == 8 && a.T == 4) return launch_tr<NP, 4, 1, 2, 2, 1, 2, 1, true, 8, 8>(a, st);
        if (halo <= 1) return launch_tr<NP, 4, 1, 2, 2, 1, 2, 1, NP == 3>(a, st);
        c3d_set_error("wgrad: the 64 x 64 four-tap slice has no two-pixel-halo form (c3d_wgrad_cfg)");
        return 1;
      } else {
        return -1;
      }
    case 6:
      // Nine taps, unfused (through round 4 every weight gradient of a data-parallel step; since round 5's one-stream form that step runs
      // the fused launches of the plain one -- unfused are a layer's second and later sources and the second-stream mode): taps split
      // across eight consumer waves + eight producer waves (NCW at the kernel).  Measured at 8 x 64 x 2048 / 32 x 1024 / 16 x 512:
      // 64 -> 64 0.438 -> 0.403 ms, 32 -> 32 0.125 -> 0.117, 128 -> 128 0.439 -> 0.417, 256 -> 256 0.405 -> 0.365.  NOT the launches
      // with the BatchNorm backward on load: dy, the stored output and x in flight twice do not fit the 128 registers of a
      // sixteen-wave workgroup even with the per-channel constants in LDS (9-45 spilled registers, every reload a full drain of
      // the loads in flight: 64 -> 64 d2 0.467 -> 0.554 ms); those keep the four + four wave form.
      // (one plane: its four-row tiles of 64 couts with four raw tiles in flight spill at 128 registers -- the four + four wave form)
      // (round 6: the fused launches too, with the producer waves split by tensor -- plan() hands them npw = 8 where the layer has
      //  no pre-activation affine)
      if constexpr (NP == 3) if (a.npw == 8 && a.T == 9) return halo <= 1 ? launch_tr<NP, 9, 1, 1, 1, 2, 2, 1, false, 8, 8>(a, st) : launch_tr<NP, 9, 1, 1, 1, 2, 2, 2, false, 8, 8>(a, st);
      return halo <= 1 ? launch_tr<NP, 9, 1, 1, 1, 2, (NP == 1 ? 4 : 2), 1>(a, st) : launch_tr<NP, 9, 1, 1, 1, 2, (NP == 1 ? 4 : 2), 2>(a, st);
    default:
      if constexpr (NP == 3 || NP == 1) if (a.npw == 8 && a.T == 9 && (NP == 3 || !a.f_dy)) return halo <= 1 ? launch_tr<NP, 9, 1, 1, 1, 1, 4, 1, false, 8, 8>(a, st) : launch_tr<NP, 9, 1, 1, 1, 1, 4, 2, NP == 3, 8, 8>(a, st);
      return halo <= 1 ? launch_tr<NP, 9, 1, 1, 1, 1, 4, 1>(a, st) : launch_tr<NP, 9, 1, 1, 1, 1, 4, 2, NP == 3>(a, st);
  }
}

}  // namespace

int c3d_wgrad_launch_tr(int planes, int id, int halo, const WgradArgs& a, hipStream_t st) {
  if (planes == 2) return launch_tr_id<2>(id, halo, a, st);      // EXPERIMENT
  return planes == 3 ? launch_tr_id<3>(id, halo, a, st) : launch_tr_id<1>(id, halo, a, st);
}
