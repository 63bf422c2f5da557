// HBM-bound glue of the SalsaNext blocks (gfx950): everything between the convolutions.
// All tensors NHWC, fp32 or -- per activation pointer, bit i of the trailing `bf16_mask` argument -- bf16
// (BASELINE configs[2] storage); kernels are 4-channel-vectorised and grid-strided, arithmetic is fp32.
// Reference (pc_processor/models/salsanext_proto.py unless noted):
//   input_norm            tasks/weak_segmentation/trainer.py:599-609
//   conv_in5              downCntx.conv1 (5 -> 32, 1x1) + LeakyReLU, :41-42,53-54
//   affine_add            shortcut + BN(a)   :64, :133
//   avgpool3s2            Dropout2d mask + AvgPool2d(3, stride 2, pad 1)  :108-109,135-142
//   pixshuf_cat           PixelShuffle(2) + Dropout2d + cat(skip) + Dropout2d  :185-191
//   softmax / crop        :456-460
//   bilinear              F.interpolate(mode="bilinear", align_corners=True)  :470-490
//   l2norm                F.normalize(dim=channel)  :485
#include "common.h"
#include "../../include/coarse3d_hip.h"

namespace {

constexpr int GS_BLOCKS = 2048;

__device__ __forceinline__ size_t gtid() { return (size_t)blockIdx.x * blockDim.x + threadIdx.x; }
__device__ __forceinline__ size_t gstride() { return (size_t)gridDim.x * blockDim.x; }

inline int nblocks(size_t n, int per = 256) {
  size_t b = (n + per - 1) / per;
  return (int)(b > GS_BLOCKS ? GS_BLOCKS : (b < 1 ? 1 : b));
}

// ---------------------------------------------------------------- input normalisation (T1)
__global__ void input_norm_kernel(const float* __restrict__ x, const int64_t* __restrict__ eval_label,
                                  const float* __restrict__ mean, const float* __restrict__ stdv, int B, int Cn,
                                  int HW, float* __restrict__ out) {
  const size_t total = (size_t)B * Cn * HW;
  for (size_t i = gtid(); i < total; i += gstride()) {
    const int p = i % HW;
    const int c = (i / HW) % Cn;
    const int b = i / ((size_t)HW * Cn);
    const float m = eval_label[(size_t)b * HW + p] > 0 ? 1.f : 0.f;
    out[i] = (x[i] - mean[c]) / stdv[c] * m;
  }
}

// ---------------------------------------------------------------- first conv 5 -> 32 (VALU)
// x NCHW [B,Cn,H,W] (Cn <= 8), w [32][Cn], out NHWC [B,HW,32]
// HBM-bound (20 B in, 128 B out per pixel): eight lanes per pixel, four couts each -- every lane's store is one
// 16-byte access and a wave writes 1 KB of consecutive bytes; the weights of the four couts live in registers.
// (The first version computed 32 couts per lane and transposed through LDS with 4-byte stores: 179 us for the
// 155 MB of the 8x64x2048 launch, 0.87 TB/s.)
__global__ __launch_bounds__(256) void conv_in5_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, int Cn, int HW, int total_pix,
                                                       float* __restrict__ out, int bf) {
  const int tid = threadIdx.x, q = tid & 7;
  float wr[4][8], br[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    br[k] = bias[q * 4 + k];
#pragma unroll
    for (int c = 0; c < 8; ++c) wr[k][c] = c < Cn ? w[(q * 4 + k) * Cn + c] : 0.f;
  }
  const int step = gridDim.x * 32;
#pragma unroll 2
  for (int p = blockIdx.x * 32 + (tid >> 3); p < total_pix; p += step) {
    const int b = p / HW, hw = p - b * HW;
    const float* xp = x + (size_t)b * Cn * HW + hw;
    float xv[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) xv[c] = c < Cn ? xp[(size_t)c * HW] : 0.f;
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float acc = br[k];
#pragma unroll
      for (int c = 0; c < 8; ++c) acc = fmaf(wr[k][c], xv[c], acc);     // c >= Cn: fmaf(0, 0, acc) == acc
      o[k] = c3d_lrelu(acc);
    }
    c3d_st4(out, (size_t)p * 32 + q * 4, bf & 1, o);
  }
}

// dW partial: [nblk][32][8].  Eight lanes per pixel (16-byte dz loads, a wave reads 1 KB of consecutive bytes), 4 x 8
// accumulators per lane, folded over the 32 pixel lanes of the workgroup at the end.
// (First version: 4-byte dz loads, one pixel pair per wave instruction: 245 us per launch, 0.64 TB/s.)
__global__ __launch_bounds__(256) void conv_in5_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                             int Cn, int HW, int total_pix, int pix_per_block,
                                                             float* __restrict__ partial, int bf) {
  __shared__ float red[4][32][8];
  const int tid = threadIdx.x, q = tid & 7, wave = tid >> 6;
  float acc[4][8];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[k][c] = 0.f;
  const int p0 = blockIdx.x * pix_per_block;
  const int p1 = min(p0 + pix_per_block, total_pix);
#pragma unroll 2
  for (int p = p0 + (tid >> 3); p < p1; p += 32) {
    const int b = p / HW, hw = p - b * HW;
    const f32x4 g = c3d_ld4(dz, (size_t)p * 32 + q * 4, bf & 1);
    const float* xp = x + (size_t)b * Cn * HW + hw;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float xv = c < Cn ? xp[(size_t)c * HW] : 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k][c] = fmaf(g[k], xv, acc[k][c]);
    }
  }
  // the 8 pixel lanes of a wave (lane bits 3..5), then the 4 waves
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float v = acc[k][c];
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      acc[k][c] = v;
    }
  if ((tid & 63) < 8) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int c = 0; c < 8; ++c) red[wave][q * 4 + k][c] = acc[k][c];
  }
  __syncthreads();
  {
    const int co = tid >> 3, c = tid & 7;
    partial[((size_t)blockIdx.x * 32 + co) * 8 + c] = (red[0][co][c] + red[1][co][c]) + (red[2][co][c] + red[3][co][c]);
  }
}

// out[co][c] = sum_blocks partial[blk][co][c]  (fp64 fold, one wave per output)
__global__ __launch_bounds__(256) void conv_in5_wgrad_reduce_kernel(const float* __restrict__ partial, int nblk, int Cn,
                                                                    float* __restrict__ dw) {
  const int lane = threadIdx.x & 63;
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= 32 * Cn) return;
  const int co = i / Cn, c = i % Cn;
  double s = 0.0;
  for (int k = lane; k < nblk; k += 64) s += (double)partial[((size_t)k * 32 + co) * 8 + c];
  s = c3d_wave_sum_d(s);
  if (lane == 0) dw[i] = (float)s;
}

// ---------------------------------------------------------------- out = x + act(a*scale + shift)
// act = identity (slope 0) or LeakyReLU(slope): the residual add of SalsaNext's blocks
// (salsanext_proto.py:64,133) and of RangeNet's BasicBlock (rangenet_proto.py:52-63)
template <int V>
__global__ void affine_add_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                  const float* __restrict__ scale, const float* __restrict__ shift, size_t npix, int C,
                                  float slope, float* __restrict__ out, int bf) {
  const int Q = C / V;
  const size_t total = npix * Q;
  for (size_t i = gtid(); i < total; i += gstride()) {
    const int c = (i % Q) * V;
    c3d_vec<V> v = c3d_vld<V>(a, i * V, bf & 2);
    if (scale) {
      const c3d_vec<V> sc = c3d_vldf<V>(scale, c), sh = c3d_vldf<V>(shift, c);
#pragma unroll
      for (int q = 0; q < V; ++q) v.v[q] = v.v[q] * sc.v[q] + sh.v[q];
    }
    if (slope > 0.f) {
#pragma unroll
      for (int q = 0; q < V; ++q) v.v[q] = c3d_lrelu(v.v[q], slope);
    }
    if (x) v += c3d_vld<V>(x, i * V, bf & 1);
    c3d_vst<V>(out, i * V, bf & 4, v);
  }
}

// ---------------------------------------------------------------- column resampling for strided / transposed convs
// up = 0: out[b][y][xo][c] = in[b][y][2*xo][c]            (W -> W/2: the stride-(1,2) convs of
//         rangenet_proto.py:194-203 are computed at stride 1 and keep the even columns)
// up = 1: out[b][y][2*xi][c] = in[b][y][xi][c], odd columns 0   (W -> 2W: zero insertion in front of
//         the 1x4 conv that realises ConvTranspose2d([1,4], stride [1,2], padding [0,1]), :328-336)
// Each is the other's adjoint, so the same kernel serves the backward passes.
__global__ void cols_resample_kernel(const float* __restrict__ in, size_t rows, int Win, int C, int up,
                                     float* __restrict__ out) {
  const int Q = C >> 2;
  const int Wout = up ? Win * 2 : Win / 2;
  const size_t total = rows * Wout * Q;
  for (size_t i = gtid(); i < total; i += gstride()) {
    const int q = i % Q;
    const size_t p = i / Q;
    const int xo = p % Wout;
    const size_t r = p / Wout;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (!up) v = *reinterpret_cast<const f32x4*>(in + ((r * Win + 2 * (size_t)xo) * Q + q) * 4);
    else if ((xo & 1) == 0) v = *reinterpret_cast<const f32x4*>(in + ((r * Win + (xo >> 1)) * Q + q) * 4);
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
  }
}

// x [B][Cn][H*W] (NCHW) -> out [B][H*W][Cp] (NHWC), channels Cn..Cp-1 zero: the 5-channel range image as a
// 16-channel MFMA operand (first 3x3 conv of rangenet_proto.py:141-143)
__global__ void nchw_to_nhwc_pad_kernel(const float* __restrict__ x, int B, int Cn, size_t HW, int Cp,
                                        float* __restrict__ out) {
  const size_t total = (size_t)B * HW * Cp;
  for (size_t i = gtid(); i < total; i += gstride()) {
    const int c = i % Cp;
    const size_t p = (i / Cp) % HW;
    const size_t b = i / ((size_t)Cp * HW);
    out[i] = c < Cn ? x[(b * Cn + c) * HW + p] : 0.f;
  }
}

// y (+)= alpha * x  (flat)
template <int V>
__global__ void axpy_kernel(const float* __restrict__ x, float alpha, size_t nv, float* __restrict__ y, int accumulate,
                            int bf) {
  for (size_t i = gtid(); i < nv; i += gstride()) {
    c3d_vec<V> v = c3d_vld<V>(x, i * V, bf & 1);
    v *= alpha;
    if (accumulate) v += c3d_vld<V>(y, i * V, bf & 2);
    c3d_vst<V>(y, i * V, bf & 2, v);
  }
}

// ---------------------------------------------------------------- mask (+ 3x3 stride-2 average pool)
// in [B,H,W,C], mask [B,C] or null; pool: out [B,Ho,Wo,C]; no pool: out = in*mask
template <int V>
__global__ void maskpool_kernel(const float* __restrict__ in, const float* __restrict__ mask, int B, int H, int W, int C,
                                int pool, int Ho, int Wo, float* __restrict__ out, int bf) {
  const int Q = C / V;
  const size_t total = (size_t)B * Ho * Wo * Q;
  for (size_t i = gtid(); i < total; i += gstride()) {
    const int c = (i % Q) * V;
    size_t r = i / Q;
    const int xo = r % Wo;
    r /= Wo;
    const int yo = r % Ho;
    const int b = r / Ho;
    c3d_vec<V> acc = c3d_vzero<V>();
    if (pool) {
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const int y = 2 * yo + dy, x = 2 * xo + dx;
          if (y >= 0 && y < H && x >= 0 && x < W) acc += c3d_vld<V>(in, ((size_t)(b * H + y) * W + x) * C + c, bf & 1);
        }
      acc *= (1.f / 9.f);
    } else {
      acc = c3d_vld<V>(in, ((size_t)(b * H + yo) * W + xo) * C + c, bf & 1);
    }
    if (mask) acc *= c3d_vldf<V>(mask, (size_t)b * C + c);
    c3d_vst<V>(out, i * V, bf & 2, acc);
  }
}

// d_in = (extra ? extra : 0) + mask * poolT(d_out)
// (tried at the end of round 2: 32-bit index splits -- no effect, the kernel is not VALU-bound; nontemporal output store --
//  130 -> 82 us in isolation at 8x64x2048x64, but neutral for the step, 209.5 vs 209.3 img/s: the consumer pays)
template <int V>
__global__ void maskpool_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ mask,
                                    const float* __restrict__ extra, int B, int H, int W, int C, int pool, int Ho,
                                    int Wo, float* __restrict__ din, int bf) {
  const int Q = C / V;
  const size_t total = (size_t)B * H * W * Q;
  for (size_t i = gtid(); i < total; i += gstride()) {
    const int c = (i % Q) * V;
    size_t r = i / Q;
    const int x = r % W;
    r /= W;
    const int y = r % H;
    const int b = r / H;
    c3d_vec<V> acc = c3d_vzero<V>();
    if (pool) {
      const int ya = (y & 1) ? (y - 1) / 2 : y / 2, yb = (y & 1) ? (y + 1) / 2 : -1;
      const int xa = (x & 1) ? (x - 1) / 2 : x / 2, xb = (x & 1) ? (x + 1) / 2 : -1;
      const int ys[2] = {ya, yb}, xs[2] = {xa, xb};
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          const int yo = ys[u], xo = xs[v];
          if (yo >= 0 && yo < Ho && xo >= 0 && xo < Wo)
            acc += c3d_vld<V>(dout, ((size_t)(b * Ho + yo) * Wo + xo) * C + c, bf & 1);
        }
      acc *= (1.f / 9.f);
    } else {
      acc = c3d_vld<V>(dout, i * V, bf & 1);
    }
    if (mask) acc *= c3d_vldf<V>(mask, (size_t)b * C + c);
    if (extra) acc += c3d_vld<V>(extra, i * V, bf & 2);
    c3d_vst<V>(din, i * V, bf & 4, acc);
  }
}

// ---------------------------------------------------------------- PixelShuffle(2) + masks + cat
struct PsArgs {
  const float* xa; const float* sc; const float* sh;  // [B,Hs,Ws,Cx], affine [Cx] or null
  const float* m3;  // [B,Cx] or null (dropout3 of the producer block)
  const float* m1;  // [B,Cu] or null
  const float* m2;  // [B,Cu+Cs] or null
  const float* skip;  // [B,H,W,Cs]
  int B, Hs, Ws, Cx, Cs;
  float* out;  // [B,2Hs,2Ws,Cu+Cs]
  int bf;      // bit 0: xa, bit 1: skip, bit 2: out are bf16
};

__global__ void pixshuf_kernel(PsArgs p) {
  const int Cu = p.Cx >> 2, Ct = Cu + p.Cs, H = 2 * p.Hs, W = 2 * p.Ws;
  const size_t total = (size_t)p.B * p.Hs * p.Ws * Cu;
  for (size_t i = gtid(); i < total; i += gstride()) {
    const int c = i % Cu;
    size_t r = i / Cu;
    const int xs = r % p.Ws;
    r /= p.Ws;
    const int ys = r % p.Hs;
    const int b = r / p.Hs;
    f32x4 v = c3d_ld4(p.xa, ((size_t)(b * p.Hs + ys) * p.Ws + xs) * p.Cx + c * 4, p.bf & 1);
    if (p.sc) v = v * *reinterpret_cast<const f32x4*>(p.sc + c * 4) + *reinterpret_cast<const f32x4*>(p.sh + c * 4);
    if (p.m3) v *= *reinterpret_cast<const f32x4*>(p.m3 + (size_t)b * p.Cx + c * 4);
    float m = 1.f;
    if (p.m1) m *= p.m1[(size_t)b * Cu + c];
    if (p.m2) m *= p.m2[(size_t)b * Ct + c];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int y = 2 * ys + (k >> 1), x = 2 * xs + (k & 1);
      c3d_st1(p.out, ((size_t)(b * H + y) * W + x) * Ct + c, p.bf & 4, v[k] * m);
    }
  }
}

template <int V>
__global__ void catskip_kernel(PsArgs p) {
  const int Cu = p.Cx >> 2, Ct = Cu + p.Cs, H = 2 * p.Hs, W = 2 * p.Ws;
  const int Q = p.Cs / V;
  const size_t total = (size_t)p.B * H * W * Q;
  for (size_t i = gtid(); i < total; i += gstride()) {
    const int c = (i % Q) * V;
    const size_t pix = i / Q;
    const int b = pix / ((size_t)H * W);
    c3d_vec<V> v = c3d_vld<V>(p.skip, pix * p.Cs + c, p.bf & 2);
    if (p.m2) v *= c3d_vldf<V>(p.m2, (size_t)b * Ct + Cu + c);
    c3d_vst<V>(p.out, pix * Ct + Cu + c, p.bf & 4, v);
  }
}

struct PsBwdArgs {
  const float* dout;  // [B,H,W,Cu+Cs]
  const float* m3; const float* m1; const float* m2;
  int B, Hs, Ws, Cx, Cs;
  float* dxa;    // [B,Hs,Ws,Cx]  gradient w.r.t. the (affine-transformed) producer output
  float* dskip;  // [B,H,W,Cs]
  int skip_accumulate;
  int bf;        // bit 0: dout, bit 1: dxa, bit 2: dskip are bf16
};

__global__ void pixshuf_bwd_kernel(PsBwdArgs p) {
  const int Cu = p.Cx >> 2, Ct = Cu + p.Cs, H = 2 * p.Hs, W = 2 * p.Ws;
  const size_t total = (size_t)p.B * p.Hs * p.Ws * Cu;
  for (size_t i = gtid(); i < total; i += gstride()) {
    const int c = i % Cu;
    size_t r = i / Cu;
    const int xs = r % p.Ws;
    r /= p.Ws;
    const int ys = r % p.Hs;
    const int b = r / p.Hs;
    float m = 1.f;
    if (p.m1) m *= p.m1[(size_t)b * Cu + c];
    if (p.m2) m *= p.m2[(size_t)b * Ct + c];
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int y = 2 * ys + (k >> 1), x = 2 * xs + (k & 1);
      v[k] = c3d_ld1(p.dout, ((size_t)(b * H + y) * W + x) * Ct + c, p.bf & 1) * m;
    }
    if (p.m3) v *= *reinterpret_cast<const f32x4*>(p.m3 + (size_t)b * p.Cx + c * 4);
    c3d_st4(p.dxa, ((size_t)(b * p.Hs + ys) * p.Ws + xs) * p.Cx + c * 4, p.bf & 2, v);
  }
}

template <int V>
__global__ void catskip_bwd_kernel(PsBwdArgs p) {
  const int Cu = p.Cx >> 2, Ct = Cu + p.Cs, H = 2 * p.Hs, W = 2 * p.Ws;
  const int Q = p.Cs / V;
  const size_t total = (size_t)p.B * H * W * Q;
  for (size_t i = gtid(); i < total; i += gstride()) {
    const int c = (i % Q) * V;
    const size_t pix = i / Q;
    const int b = pix / ((size_t)H * W);
    c3d_vec<V> v = c3d_vld<V>(p.dout, pix * Ct + Cu + c, p.bf & 1);
    if (p.m2) v *= c3d_vldf<V>(p.m2, (size_t)b * Ct + Cu + c);
    if (p.skip_accumulate) v += c3d_vld<V>(p.dskip, pix * p.Cs + c, p.bf & 4);
    c3d_vst<V>(p.dskip, pix * p.Cs + c, p.bf & 4, v);
  }
}

// ---------------------------------------------------------------- channel softmax (+crop)
// One pixel per lane, its C <= 32 channels in registers (16-byte accesses when C and cs are multiples of 4).  Sums
// over the classes keep the pairing of a 32-lane xor butterfly over the zero-padded classes (16, 8, 4, 2, 1): the
// first version of these kernels ran 32 lanes per pixel with ten dependent cross-lane shuffles per pixel and was
// bound by their latency (119 / 110 us per 8x64x2048 launch, 1.8 / 2.7 TB/s); these produce the same bits.
__device__ __forceinline__ float butterfly32_sum(float (&h)[32]) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1)
#pragma unroll
    for (int c = 0; c < o; ++c) h[c] = __fadd_rn(h[c], h[c + o]);
  return h[0];
}
template <bool V4>
__device__ __forceinline__ void ld_row32(const float* __restrict__ p, int C, float fill, float (&v)[32]) {
  if constexpr (V4) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (q * 4 < C) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p + q * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[q * 4 + k] = t[k];
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[q * 4 + k] = fill;
      }
    }
  } else {
#pragma unroll
    for (int c = 0; c < 32; ++c) v[c] = c < C ? p[c] : fill;
  }
}
template <bool V4>
__device__ __forceinline__ void st_row32(float* __restrict__ p, int C, const float (&v)[32]) {
  if constexpr (V4) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (q * 4 < C) *reinterpret_cast<f32x4*>(p + q * 4) = f32x4{v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]};
  } else {
#pragma unroll
    for (int c = 0; c < 32; ++c)
      if (c < C) p[c] = v[c];
  }
}

// logits [B,H,W,cs] (first C channels used) -> prob [B,Ho,Wo,C]
template <bool V4>
__global__ __launch_bounds__(256) void softmax_kernel(const float* __restrict__ logits, int B, int H, int W, int cs,
                                                      int C, int Ho, int Wo, float* __restrict__ prob) {
  const int total = B * Ho * Wo;
  for (int i = (int)gtid(); i < total; i += (int)gstride()) {
    const int x = i % Wo;
    const int y = (i / Wo) % Ho;
    const int b = i / (Wo * Ho);
    float v[32], e[32];
    ld_row32<V4>(logits + ((size_t)(b * H + y) * W + x) * cs, C, -INFINITY, v);
    float mx = v[0];
#pragma unroll
    for (int c = 1; c < 32; ++c) mx = fmaxf(mx, v[c]);
#pragma unroll
    for (int c = 0; c < 32; ++c) e[c] = c < C ? expf(v[c] - mx) : 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) v[c] = e[c];
    const float s = butterfly32_sum(v);
#pragma unroll
    for (int c = 0; c < 32; ++c) e[c] = e[c] / s;
    st_row32<V4>(prob + (size_t)i * C, C, e);
  }
}

// dlogits [B,H,W,cs] = p * (dp - sum(p*dp)) inside the crop, 0 elsewhere (incl. pad channels)
template <bool V4>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ prob,
                                                          const float* __restrict__ dprob, int B, int H, int W, int cs,
                                                          int C, int Ho, int Wo, float* __restrict__ dlogits) {
  const int total = B * H * W;
  for (int i = (int)gtid(); i < total; i += (int)gstride()) {
    const int x = i % W;
    const int y = (i / W) % H;
    const int b = i / (W * H);
    float out[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) out[c] = 0.f;
    if (y < Ho && x < Wo) {
      const size_t j = ((size_t)(b * Ho + y) * Wo + x) * C;
      float p[32], g[32], t[32];
      ld_row32<V4>(prob + j, C, 0.f, p);
      ld_row32<V4>(dprob + j, C, 0.f, g);
#pragma unroll
      for (int c = 0; c < 32; ++c) t[c] = __fmul_rn(p[c], g[c]);
      const float dot = butterfly32_sum(t);
#pragma unroll
      for (int c = 0; c < 32; ++c) out[c] = p[c] * (g[c] - dot);
    }
    st_row32<V4>(dlogits + (size_t)i * cs, cs, out);
  }
}

// ---------------------------------------------------------------- bilinear, align_corners=True
struct BlArgs {
  const float* src; int Hs, Ws, scs, scoff;
  float* dst; int Hd, Wd, dcs, dcoff;
  int B, C;
  float ry, rx;
  int bf;        // bit 0: src, bit 1: dst are bf16
};

__device__ __forceinline__ void bl_coords(int d, float ratio, int n, int& i0, int& i1, float& l1) {
  const float r = ratio * (float)d;
  i0 = (int)r;
  if (i0 > n - 1) i0 = n - 1;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  l1 = fminf(fmaxf(r - (float)i0, 0.f), 1.f);
}

// the four-tap mix of one output element; bilinear_kernel and bilinear_rows_kernel share it so that a row gathered
// straight from the source equals the row of the materialised map bit for bit
template <int V>
__device__ __forceinline__ c3d_vec<V> bl_mix(const c3d_vec<V>& v00, const c3d_vec<V>& v01, const c3d_vec<V>& v10,
                                             const c3d_vec<V>& v11, float lx, float ly) {
  c3d_vec<V> o;
#pragma unroll
  for (int q = 0; q < V; ++q) {
    const float top = v00.v[q] * (1.f - lx) + v01.v[q] * lx;
    const float bot = v10.v[q] * (1.f - lx) + v11.v[q] * lx;
    o.v[q] = top * (1.f - ly) + bot * ly;
  }
  return o;
}

// Work item of a workgroup = 2048 consecutive (pixel, channel-group) elements of ONE destination row: the row's
// image, y coordinates and base addresses are workgroup-uniform (scalar unit), a lane only divides its element index
// by the channel groups per pixel (a shift when that is a power of two).  (First version: a flat grid-stride loop with
// four 64-bit div/mod per element -- the 32x1024 -> 64x2048 upsampling of the 256-channel embedding ran 538 us for
// 1.34 GB, VALU-bound.)
// One work item per workgroup, consecutive items on the same XCD (c3d_xcd_remap): the source rows an XCD's L2 holds are
// then read by that XCD only.  The forward kernel's stores are nontemporal (c3d_vst_nt): its four taps per output live on
// L2 hits, which its own output stream otherwise evicts.
constexpr int BL_IT = 8;      // forward: elements per lane
constexpr int BL_IT_BWD = 1;  // backward: an element is tens of loads already, and small sources need the parallelism
template <int V>
__global__ __launch_bounds__(256) void bilinear_kernel(BlArgs p, int chunks_per_row, int q_shift) {
  const int Q = p.C / V;
  const int row_elems = p.Wd * Q;
  {
    const int item = c3d_xcd_remap(blockIdx.x, gridDim.x);
    const int row = item / chunks_per_row, chunk = item - row * chunks_per_row;
    const int b = row / p.Hd, yd = row - b * p.Hd;
    int y0, y1;
    float ly;
    bl_coords(yd, p.ry, p.Hs, y0, y1, ly);
    const size_t s0 = ((size_t)b * p.Hs + y0) * p.Ws * p.scs + p.scoff;
    const size_t s1 = ((size_t)b * p.Hs + y1) * p.Ws * p.scs + p.scoff;
    const size_t d0 = ((size_t)b * p.Hd + yd) * p.Wd * p.dcs + p.dcoff;
#pragma unroll 4
    for (int k = 0; k < BL_IT; ++k) {
      const int e = (chunk * BL_IT + k) * 256 + threadIdx.x;
      if (e < row_elems) {
        const int xd = q_shift >= 0 ? e >> q_shift : e / Q;
        const int c = (e - xd * Q) * V;
        int x0, x1;
        float lx;
        bl_coords(xd, p.rx, p.Ws, x0, x1, lx);
        const c3d_vec<V> v00 = c3d_vld<V>(p.src, s0 + (size_t)(x0 * p.scs + c), p.bf & 1);
        const c3d_vec<V> v01 = c3d_vld<V>(p.src, s0 + (size_t)(x1 * p.scs + c), p.bf & 1);
        const c3d_vec<V> v10 = c3d_vld<V>(p.src, s1 + (size_t)(x0 * p.scs + c), p.bf & 1);
        const c3d_vec<V> v11 = c3d_vld<V>(p.src, s1 + (size_t)(x1 * p.scs + c), p.bf & 1);
        const c3d_vec<V> o = bl_mix<V>(v00, v01, v10, v11, lx, ly);
        c3d_vst_nt<V>(p.dst, d0 + (size_t)(xd * p.dcs + c), p.bf & 2, o);
      }
    }
  }
}

// out[r][:] = bilinear(src)[pixel of row r][:] for a LIST of destination pixels -- the rows a consumer needs of a map that
// is never materialised (the 64x2048 embedding, 1.07 GB at bs=8: only ~10^3 anchor rows and <= 8192 labelled rows of
// it are ever read inside the training step; salsanext_proto.py:488-490 followed by contrast_pixel_loss.py's and
// prototype_learning's row selections).  Row r is destination pixel idx[r] of image img[r / A] (img != null) or the flat
// pixel idx[r] = b * Hd * Wd + pixel (img == null).  Rows with r / A >= *count are zero.  l2: rows leave l2-normalised,
// norm[r] = their length (rows past count: 1) -- the arithmetic of gather_l2_kernel.  One wave per row.
template <class IDX>
__global__ __launch_bounds__(256) void bilinear_rows_kernel(BlArgs p, const int32_t* __restrict__ img,
                                                            const IDX* __restrict__ idx, int A,
                                                            const int32_t* __restrict__ count, int R, int l2, float eps,
                                                            float* __restrict__ out, float* __restrict__ norm) {
  constexpr int V = 4;
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  const int live = count ? *count : R;
  const int n = p.Hd * p.Wd;
  for (size_t r = wave; r < (size_t)R; r += nw) {
    if ((int)(r / A) >= live) {
      for (int c = lane * V; c < p.C; c += 64 * V) c3d_vst<V>(out, r * p.C + c, false, c3d_vzero<V>());
      if (norm && lane == 0) norm[r] = 1.f;
      continue;
    }
    int b, pix;
    if (img) {
      b = img[r / A];
      pix = (int)idx[r];
    } else {
      const long long flat = (long long)idx[r];
      b = (int)(flat / n);
      pix = (int)(flat - (long long)b * n);
    }
    const int yd = pix / p.Wd, xd = pix - yd * p.Wd;
    int y0, y1, x0, x1;
    float ly, lx;
    bl_coords(yd, p.ry, p.Hs, y0, y1, ly);
    bl_coords(xd, p.rx, p.Ws, x0, x1, lx);
    const size_t s0 = ((size_t)b * p.Hs + y0) * p.Ws * p.scs + p.scoff;
    const size_t s1 = ((size_t)b * p.Hs + y1) * p.Ws * p.scs + p.scoff;
    auto row_quad = [&](int c) {
      const c3d_vec<V> v00 = c3d_vld<V>(p.src, s0 + (size_t)(x0 * p.scs + c), p.bf & 1);
      const c3d_vec<V> v01 = c3d_vld<V>(p.src, s0 + (size_t)(x1 * p.scs + c), p.bf & 1);
      const c3d_vec<V> v10 = c3d_vld<V>(p.src, s1 + (size_t)(x0 * p.scs + c), p.bf & 1);
      const c3d_vec<V> v11 = c3d_vld<V>(p.src, s1 + (size_t)(x1 * p.scs + c), p.bf & 1);
      return bl_mix<V>(v00, v01, v10, v11, lx, ly);
    };
    float inv = 1.f;
    if (p.C <= 64 * V) {
      // one quad per lane (the 256-channel embedding): interpolate once, keep it for the norm and the store
      const int c = lane * V;
      c3d_vec<V> v = c < p.C ? row_quad(c) : c3d_vzero<V>();
      if (l2) {
        const float nr = sqrtf(c3d_wave_sum(v.v[0] * v.v[0] + v.v[1] * v.v[1] + v.v[2] * v.v[2] + v.v[3] * v.v[3]));
        inv = 1.f / fmaxf(nr, eps);
        if (norm && lane == 0) norm[r] = nr;
        v *= inv;
      }
      if (c < p.C) c3d_vst<V>(out, r * p.C + c, false, v);
      continue;
    }
    if (l2) {
      float s = 0.f;
      for (int c = lane * V; c < p.C; c += 64 * V) {
        const c3d_vec<V> v = row_quad(c);
        s += v.v[0] * v.v[0] + v.v[1] * v.v[1] + v.v[2] * v.v[2] + v.v[3] * v.v[3];
      }
      const float nr = sqrtf(c3d_wave_sum(s));
      inv = 1.f / fmaxf(nr, eps);
      if (norm && lane == 0) norm[r] = nr;
    }
    for (int c = lane * V; c < p.C; c += 64 * V) {
      c3d_vec<V> v = row_quad(c);
      if (l2) v *= inv;
      c3d_vst<V>(out, r * p.C + c, false, v);
    }
  }
}

// dst = bilinear(src) + bilinear(src2): two sources of different sizes, the same channel count, fp32, added in
// one pass (the projector's 1x1 conv commutes with the resampling of its inputs: the shares of the two low-resolution
// skips are multiplied at THEIR resolution and meet the full-resolution result here; coarse3d_amd/backbone.py).
// Same work decomposition as bilinear_kernel.
struct Bl2Args {
  BlArgs a;              // src / Hs / Ws / scs, dst geometry, ratios of source 1
  const float* src2; int Hs2, Ws2, scs2;
  float ry2, rx2;
};
// Work order (round 4): channel blocks of qb quads OUTERMOST -- item = (channel block, destination row, chunk of the row's
// pixels).  Row-major over all 704 channels a destination row needs 2.9 + 1.4 MB of source rows, more than an XCD's 4 MB of
// L2 together with the 2.9 MB it writes: the sources (230 MB) came from HBM five times (1.9 GB per launch for 0.97 GB of
// tensors).  With 64-channel blocks the rows two consecutive items share are 0.4 MB.
__global__ __launch_bounds__(256) void bilinear_sum2_kernel(Bl2Args q, int chunks_per_row, int qb, int qb_shift) {
  constexpr int V = 4;
  const BlArgs& p = q.a;
  const int row_elems = p.Wd * qb;                 // elements (channel quads) of one destination row inside a channel block
  const int item = c3d_xcd_remap(blockIdx.x, gridDim.x);
  const int nrows = p.B * p.Hd;
  const int cb = item / (nrows * chunks_per_row);
  const int rem = item - cb * (nrows * chunks_per_row);
  const int row = rem / chunks_per_row, chunk = rem - row * chunks_per_row;
  const int b = row / p.Hd, yd = row - b * p.Hd;
  int y0, y1, u0, u1;
  float ly, lu;
  bl_coords(yd, p.ry, p.Hs, y0, y1, ly);
  bl_coords(yd, q.ry2, q.Hs2, u0, u1, lu);
  const size_t s0 = ((size_t)b * p.Hs + y0) * p.Ws * p.scs, s1 = ((size_t)b * p.Hs + y1) * p.Ws * p.scs;
  const size_t t0 = ((size_t)b * q.Hs2 + u0) * q.Ws2 * q.scs2, t1 = ((size_t)b * q.Hs2 + u1) * q.Ws2 * q.scs2;
  const size_t d0 = ((size_t)b * p.Hd + yd) * p.Wd * p.dcs;
#pragma unroll 4
  for (int k = 0; k < BL_IT; ++k) {
    const int e = (chunk * BL_IT + k) * 256 + threadIdx.x;
    if (e < row_elems) {
      const int xd = qb_shift >= 0 ? e >> qb_shift : e / qb;
      const int c = (cb * qb + (e - xd * qb)) * V;
      int x0, x1, w0, w1;
      float lx, lw;
      bl_coords(xd, p.rx, p.Ws, x0, x1, lx);
      bl_coords(xd, q.rx2, q.Ws2, w0, w1, lw);
      const c3d_vec<V> a00 = c3d_vldf<V>(p.src, s0 + (size_t)(x0 * p.scs + c)), a01 = c3d_vldf<V>(p.src, s0 + (size_t)(x1 * p.scs + c));
      const c3d_vec<V> a10 = c3d_vldf<V>(p.src, s1 + (size_t)(x0 * p.scs + c)), a11 = c3d_vldf<V>(p.src, s1 + (size_t)(x1 * p.scs + c));
      const c3d_vec<V> b00 = c3d_vldf<V>(q.src2, t0 + (size_t)(w0 * q.scs2 + c)), b01 = c3d_vldf<V>(q.src2, t0 + (size_t)(w1 * q.scs2 + c));
      const c3d_vec<V> b10 = c3d_vldf<V>(q.src2, t1 + (size_t)(w0 * q.scs2 + c)), b11 = c3d_vldf<V>(q.src2, t1 + (size_t)(w1 * q.scs2 + c));
      c3d_vec<V> o;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float atop = a00.v[j] * (1.f - lx) + a01.v[j] * lx, abot = a10.v[j] * (1.f - lx) + a11.v[j] * lx;
        const float btop = b00.v[j] * (1.f - lw) + b01.v[j] * lw, bbot = b10.v[j] * (1.f - lw) + b11.v[j] * lw;
        o.v[j] = (atop * (1.f - ly) + abot * ly) + (btop * (1.f - lu) + bbot * lu);
      }
      c3d_vst<V>(p.dst, d0 + (size_t)(xd * p.dcs + c), (p.bf & 2) != 0, o);
    }
  }
}

// Range of destination indices d whose interpolation touches source index s:
// floor(ratio*d) in {s-1, s}  <=>  (s-1)/ratio <= d < (s+1)/ratio   (+-1 slack, exact test inside)
__device__ __forceinline__ void bl_dst_range(int s, float ratio, int nd, int& lo, int& hi) {
  if (ratio <= 0.f) {
    lo = 0;
    hi = nd - 1;
    return;
  }
  lo = (int)floorf((float)(s - 1) / ratio) - 1;
  hi = (int)ceilf((float)(s + 1) / ratio) + 1;
  if (lo < 0) lo = 0;
  if (hi > nd - 1) hi = nd - 1;
}

// weight with which destination index d reads source index s along one axis
__device__ __forceinline__ float bl_weight(int d, int s, float ratio, int n) {
  int i0, i1;
  float l1;
  bl_coords(d, ratio, n, i0, i1, l1);
  float w = 0.f;
  if (i0 == s) w += 1.f - l1;
  if (i1 == s) w += l1;
  return w;
}

// d_src (+)= bilinear^T(d_dst) as a GATHER over the destination pixels that read each source
// pixel: deterministic, no atomics.  src = d_src (written), dst = d_dst (read).
// Same work decomposition as the forward kernel over SOURCE rows: the candidate destination rows and their weights
// are workgroup-uniform, a lane evaluates the weights of its (up to eight cached) candidate columns once instead of
// once per candidate row.  Summation order (rows outer, columns inner, both ascending) as in the first version.
constexpr int BL_NX = 8;
template <int V>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(BlArgs p, int accumulate, int chunks_per_row, int q_shift,
                                                           const uint32_t* __restrict__ rowmask) {
  const int Q = p.C / V;
  const int row_elems = p.Ws * Q;
  float* dsrc = const_cast<float*>(p.src);
  {
    // (items ordered in column bands so that the destination rows two source rows share stay L2-resident: measured
    //  neutral, 394 -> 383 us on the 1 GB embedding gradient; loads batched ahead of the adds: slower, 471 us; round 4:
    //  64-channel blocks outermost, which is worth 15 % in bilinear_sum2_kernel: neutral to worse here, 263 -> 283 us)
    const int item = c3d_xcd_remap(blockIdx.x, gridDim.x);
    const int row = item / chunks_per_row, chunk = item - row * chunks_per_row;
    const int b = row / p.Hs, ys = row - b * p.Hs;
    int ylo, yhi;
    bl_dst_range(ys, p.ry, p.Hd, ylo, yhi);
    const size_t o0 = ((size_t)b * p.Hs + ys) * p.Ws * p.scs + p.scoff;
    for (int k = 0; k < BL_IT_BWD; ++k) {
      const int e = (chunk * BL_IT_BWD + k) * 256 + threadIdx.x;
      if (e < row_elems) {
        const int xs = q_shift >= 0 ? e >> q_shift : e / Q;
        const int c = (e - xs * Q) * V;
        int xlo, xhi;
        bl_dst_range(xs, p.rx, p.Wd, xlo, xhi);
        float wxr[BL_NX];
#pragma unroll
        for (int j = 0; j < BL_NX; ++j) wxr[j] = xlo + j <= xhi ? bl_weight(xlo + j, xs, p.rx, p.Ws) : 0.f;
        c3d_vec<V> acc = c3d_vzero<V>();
        for (int yd = ylo; yd <= yhi; ++yd) {
          const float wy = bl_weight(yd, ys, p.ry, p.Hs);
          if (wy == 0.f) continue;
          const size_t g0 = ((size_t)b * p.Hd + yd) * p.Wd * p.dcs + p.dcoff;
          const unsigned m0 = (unsigned)((b * p.Hd + yd) * p.Wd);      // destination pixel index of column 0
#pragma unroll
          for (int j = 0; j < BL_NX; ++j) {
            if (wxr[j] != 0.f && (!rowmask || ((rowmask[(m0 + xlo + j) >> 5] >> ((m0 + xlo + j) & 31)) & 1u))) {
              c3d_vec<V> g = c3d_vld<V>(p.dst, g0 + (size_t)((xlo + j) * p.dcs + c), p.bf & 2);
              g *= (wy * wxr[j]);
              acc += g;
            }
          }
          for (int xd = xlo + BL_NX; xd <= xhi; ++xd) {
            const float wx = bl_weight(xd, xs, p.rx, p.Ws);
            if (wx == 0.f) continue;
            if (rowmask && !((rowmask[(m0 + xd) >> 5] >> ((m0 + xd) & 31)) & 1u)) continue;
            c3d_vec<V> g = c3d_vld<V>(p.dst, g0 + (size_t)(xd * p.dcs + c), p.bf & 2);
            g *= (wy * wx);
            acc += g;
          }
        }
        const size_t o = o0 + (size_t)(xs * p.scs + c);
        if (accumulate) acc += c3d_vld<V>(dsrc, o, p.bf & 1);
        c3d_vst<V>(dsrc, o, p.bf & 1, acc);
      }
    }
  }
}

// The same adjoint for a destination gradient in COMPACT form (drows / cmap / rowmask of c3d_scatter_rows_compact):
// ~10^4 of 10^6 destination pixels carry a row.  One WAVE per source pixel, so everything that decides which
// destination pixels are read is wave-uniform: the lanes test one candidate destination pixel each (bitmap bit, both
// weights, cmap slot -- all loads in flight together), a ballot gives the contributing ones, and the wave walks them in
// order; the common case -- nothing in the support -- is one round of loads and a zero store.  (Scalar loads of the
// bitmap words row by row, each waited for in turn: 136 us.)
// Summation order and arithmetic as in bilinear_bwd_kernel over the equivalent zero-filled dense gradient (rows outer,
// columns inner, ascending; g * (wy * wx) added): bit-identical.  (The generic kernel with the bitmap as a filter, one
// lane per (pixel, channel quad) each testing its own bits: 228 us for the 8x32x1024x256 embedding gradient.)
__global__ __launch_bounds__(256) void bilinear_bwd_rows_kernel(BlArgs p, int accumulate, const uint32_t* __restrict__ rowmask,
                                                                const int32_t* __restrict__ cmap, int nsrc) {
  constexpr int V = 4;
  const int lane = threadIdx.x & 63;
  float* dsrc = const_cast<float*>(p.src);
  const int wv = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  if (wv >= nsrc) return;
  const int xs = wv % p.Ws;
  const int rowi = wv / p.Ws;
  const int b = rowi / p.Hs, ys = rowi - b * p.Hs;
  int ylo, yhi, xlo, xhi;
  bl_dst_range(ys, p.ry, p.Hd, ylo, yhi);
  bl_dst_range(xs, p.rx, p.Wd, xlo, xhi);
  const size_t o0 = (((size_t)b * p.Hs + ys) * p.Ws + xs) * p.scs + p.scoff;
  // candidate destination pixels (row-major over [ylo, yhi] x [xlo, xhi]: the dense kernel's summation order), 64 per
  // round, one per lane: bit test, weights and the cmap lookup of all of them are in flight together
  const int nx = xhi - xlo + 1, nc = (yhi - ylo + 1) * nx;
  for (int c0 = 0; c0 < p.C; c0 += 64 * V) {
    const int c = c0 + lane * V;
    c3d_vec<V> acc = c3d_vzero<V>();
    for (int base = 0; base < nc; base += 64) {
      const int cand = base + lane;
      const int yd = ylo + cand / nx, xd = xlo + cand % nx;
      float w = 0.f;
      int slot = 0;
      bool live = false;
      if (cand < nc) {
        const float wy = bl_weight(yd, ys, p.ry, p.Hs), wx = bl_weight(xd, xs, p.rx, p.Ws);
        const unsigned pix = (unsigned)((b * p.Hd + yd) * p.Wd + xd);
        live = wy != 0.f && wx != 0.f && ((rowmask[pix >> 5] >> (pix & 31)) & 1u);
        if (live) {
          slot = cmap[pix];
          w = wy * wx;
        }
      }
      unsigned long long m = __ballot(live);
      while (m) {
        const int j = __builtin_ctzll(m);
        m &= m - 1;
        const int sj = __builtin_amdgcn_readlane(slot, j);
        const float wj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w), j));
        if (c < p.C) {
          c3d_vec<V> g = c3d_vld<V>(p.dst, (size_t)sj * p.dcs + c, false);
          g *= wj;
          acc += g;
        }
      }
    }
    if (c < p.C) {
      const size_t o = o0 + c;
      if (accumulate) acc += c3d_vld<V>(dsrc, o, p.bf & 1);
      c3d_vst<V>(dsrc, o, p.bf & 1, acc);
    }
  }
}

// ---------------------------------------------------------------- row-wise L2 normalise (one wave per row)
template <int V>
__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ x, size_t n, int C, float eps,
                                                     float* __restrict__ y, float* __restrict__ norm, int bf) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t r = wave; r < n; r += nw) {
    float s = 0.f;
    for (int c = lane * V; c < C; c += 64 * V) {
      const c3d_vec<V> v = c3d_vld<V>(x, r * C + c, bf & 1);
#pragma unroll
      for (int q = 0; q < V; ++q) s += v.v[q] * v.v[q];
    }
    s = c3d_wave_sum(s);
    const float nr = sqrtf(s);
    const float inv = 1.f / fmaxf(nr, eps);
    for (int c = lane * V; c < C; c += 64 * V) {
      c3d_vec<V> v = c3d_vld<V>(x, r * C + c, bf & 1);
      v *= inv;
      c3d_vst<V>(y, r * C + c, bf & 2, v);
    }
    if (norm && lane == 0) norm[r] = nr;
  }
}

// dx = (dy - y * sum(y*dy)) / max(norm, eps)
template <int V>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ norm,
                                                         const float* __restrict__ dy, size_t n, int C, float eps,
                                                         float* __restrict__ dx, int bf) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t r = wave; r < n; r += nw) {
    float s = 0.f;
    for (int c = lane * V; c < C; c += 64 * V) {
      const c3d_vec<V> a = c3d_vld<V>(y, r * C + c, bf & 1), g = c3d_vld<V>(dy, r * C + c, bf & 2);
#pragma unroll
      for (int q = 0; q < V; ++q) s += a.v[q] * g.v[q];
    }
    s = c3d_wave_sum(s);
    const float inv = 1.f / fmaxf(norm[r], eps);
    for (int c = lane * V; c < C; c += 64 * V) {
      const c3d_vec<V> a = c3d_vld<V>(y, r * C + c, bf & 1), g = c3d_vld<V>(dy, r * C + c, bf & 2);
      c3d_vec<V> o;
#pragma unroll
      for (int q = 0; q < V; ++q) o.v[q] = (g.v[q] - a.v[q] * s) * inv;
      c3d_vst<V>(dx, r * C + c, bf & 4, o);
    }
  }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int c3d_input_norm(const float* x, const int64_t* eval_label, const float* mean, const float* stdv, int B,
                              int Cn, int HW, float* out, c3d_stream stream) {
  hipLaunchKernelGGL(input_norm_kernel, dim3(nblocks((size_t)B * Cn * HW)), dim3(256), 0, ST, x, eval_label, mean, stdv,
                     B, Cn, HW, out);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_conv_in5(const float* x_nchw, const float* w, const float* bias, int B, int Cn, int HW, float* out, int bf16_mask,
                            c3d_stream stream) {
  C3D_REQUIRE(Cn <= 8, "conv_in5: at most 8 input channels");
  const int total = B * HW;
  int nb = (total + 31) / 32;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(conv_in5_kernel, dim3(nb), dim3(256), 0, ST, x_nchw, w, bias, Cn, HW, total, out, bf16_mask);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_conv_in5_wgrad(const float* x_nchw, const float* dz, int B, int Cn, int HW, float* partial,
                                  float* dw, int bf16_mask, c3d_stream stream) {
  C3D_REQUIRE(Cn <= 8, "conv_in5: at most 8 input channels");
  const int total = B * HW;
  int nb = (total + 511) / 512;
  if (nb > 1024) nb = 1024;
  const int ppb = (total + nb - 1) / nb;
  hipLaunchKernelGGL(conv_in5_wgrad_kernel, dim3(nb), dim3(256), 0, ST, x_nchw, dz, Cn, HW, total, ppb, partial, bf16_mask);
  C3D_CHECK_LAUNCH();
  hipLaunchKernelGGL(conv_in5_wgrad_reduce_kernel, dim3((32 * Cn * 64 + 255) / 256), dim3(256), 0, ST, partial, nb, Cn, dw);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_affine_add(const float* x, const float* a, const float* scale, const float* shift, int64_t npix,
                              int C, float lrelu_slope, float* out, int bf16_mask, c3d_stream stream) {
  C3D_REQUIRE(lrelu_slope <= 1.f, "affine_add: LeakyReLU slopes above 1 are not supported (max(v, slope * v))");
  C3D_REQUIRE(C % 4 == 0, "affine_add: C must be a multiple of 4");
  if (bf16_mask && C % 8 == 0)
    hipLaunchKernelGGL(affine_add_kernel<8>, dim3(nblocks((size_t)npix * C / 8)), dim3(256), 0, ST, x, a, scale, shift,
                       (size_t)npix, C, lrelu_slope, out, bf16_mask);
  else
    hipLaunchKernelGGL(affine_add_kernel<4>, dim3(nblocks((size_t)npix * C / 4)), dim3(256), 0, ST, x, a, scale, shift,
                       (size_t)npix, C, lrelu_slope, out, bf16_mask);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_cols_resample(const float* in, int64_t rows, int Win, int C, int up, float* out, c3d_stream stream) {
  C3D_REQUIRE(C % 4 == 0, "cols_resample: C must be a multiple of 4");
  C3D_REQUIRE(up || Win % 2 == 0, "cols_resample: an even width is needed to drop every other column");
  const size_t total = (size_t)rows * (up ? Win * 2 : Win / 2) * (C / 4);
  if (total == 0) return 0;
  hipLaunchKernelGGL(cols_resample_kernel, dim3(nblocks(total)), dim3(256), 0, ST, in, (size_t)rows, Win, C, up, out);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_nchw_to_nhwc_pad(const float* x, int B, int Cn, int64_t HW, int Cp, float* out, c3d_stream stream) {
  C3D_REQUIRE(Cp >= Cn, "nchw_to_nhwc_pad: Cp must not be smaller than Cn");
  hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel, dim3(nblocks((size_t)B * HW * Cp)), dim3(256), 0, ST, x, B, Cn, (size_t)HW, Cp,
                     out);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_axpy(const float* x, float alpha, int64_t n, float* y, int accumulate, int bf16_mask, c3d_stream stream) {
  C3D_REQUIRE(n % 4 == 0, "axpy: n must be a multiple of 4");
  if (bf16_mask && n % 8 == 0)
    hipLaunchKernelGGL(axpy_kernel<8>, dim3(nblocks((size_t)n / 8)), dim3(256), 0, ST, x, alpha, (size_t)n / 8, y, accumulate, bf16_mask);
  else
    hipLaunchKernelGGL(axpy_kernel<4>, dim3(nblocks((size_t)n / 4)), dim3(256), 0, ST, x, alpha, (size_t)n / 4, y, accumulate, bf16_mask);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_maskpool(const float* in, const float* mask, int B, int H, int W, int C, int pool, float* out, int bf16_mask,
                            c3d_stream stream) {
  C3D_REQUIRE(C % 4 == 0, "maskpool: C must be a multiple of 4");
  const int Ho = pool ? (H + 1) / 2 : H, Wo = pool ? (W + 1) / 2 : W;
  if (bf16_mask && C % 8 == 0)
    hipLaunchKernelGGL(maskpool_kernel<8>, dim3(nblocks((size_t)B * Ho * Wo * C / 8)), dim3(256), 0, ST, in, mask, B, H, W, C,
                       pool, Ho, Wo, out, bf16_mask);
  else
    hipLaunchKernelGGL(maskpool_kernel<4>, dim3(nblocks((size_t)B * Ho * Wo * C / 4)), dim3(256), 0, ST, in, mask, B, H, W, C,
                       pool, Ho, Wo, out, bf16_mask);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_maskpool_bwd(const float* dout, const float* mask, const float* extra, int B, int H, int W, int C,
                                int pool, float* din, int bf16_mask, c3d_stream stream) {
  C3D_REQUIRE(C % 4 == 0, "maskpool: C must be a multiple of 4");
  const int Ho = pool ? (H + 1) / 2 : H, Wo = pool ? (W + 1) / 2 : W;
  if (bf16_mask && C % 8 == 0)
    hipLaunchKernelGGL(maskpool_bwd_kernel<8>, dim3(nblocks((size_t)B * H * W * C / 8)), dim3(256), 0, ST, dout, mask, extra,
                       B, H, W, C, pool, Ho, Wo, din, bf16_mask);
  else
    hipLaunchKernelGGL(maskpool_bwd_kernel<4>, dim3(nblocks((size_t)B * H * W * C / 4)), dim3(256), 0, ST, dout, mask, extra,
                       B, H, W, C, pool, Ho, Wo, din, bf16_mask);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_pixshuf_cat(const float* xa, const float* sc, const float* sh, const float* m3, const float* m1,
                               const float* m2, const float* skip, int B, int Hs, int Ws, int Cx, int Cs, float* out, int bf16_mask,
                               c3d_stream stream) {
  C3D_REQUIRE(Cx % 16 == 0 && Cs % 4 == 0, "pixshuf_cat: Cx %% 16 and Cs %% 4 required");
  PsArgs p{xa, sc, sh, m3, m1, m2, skip, B, Hs, Ws, Cx, Cs, out, bf16_mask};
  hipLaunchKernelGGL(pixshuf_kernel, dim3(nblocks((size_t)B * Hs * Ws * (Cx / 4))), dim3(256), 0, ST, p);
  C3D_CHECK_LAUNCH();
  if (Cs == 0) return 0;      // PixelShuffle only: the consumer reads the skip tensor as a second source (no copy)
  C3D_REQUIRE(skip != nullptr, "pixshuf_cat: Cs > 0 needs the skip tensor");
  if (bf16_mask && Cs % 8 == 0 && (Cx / 4) % 8 == 0)
    hipLaunchKernelGGL(catskip_kernel<8>, dim3(nblocks((size_t)B * Hs * Ws * 4 * (Cs / 8))), dim3(256), 0, ST, p);
  else
    hipLaunchKernelGGL(catskip_kernel<4>, dim3(nblocks((size_t)B * Hs * Ws * 4 * (Cs / 4))), dim3(256), 0, ST, p);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_pixshuf_cat_bwd(const float* dout, const float* m3, const float* m1, const float* m2, int B, int Hs,
                                   int Ws, int Cx, int Cs, float* dxa, float* dskip, int skip_accumulate, int bf16_mask,
                                   c3d_stream stream) {
  PsBwdArgs p{dout, m3, m1, m2, B, Hs, Ws, Cx, Cs, dxa, dskip, skip_accumulate, bf16_mask};
  hipLaunchKernelGGL(pixshuf_bwd_kernel, dim3(nblocks((size_t)B * Hs * Ws * (Cx / 4))), dim3(256), 0, ST, p);
  C3D_CHECK_LAUNCH();
  if (Cs == 0) return 0;
  if (bf16_mask && Cs % 8 == 0 && (Cx / 4) % 8 == 0)
    hipLaunchKernelGGL(catskip_bwd_kernel<8>, dim3(nblocks((size_t)B * Hs * Ws * 4 * (Cs / 8))), dim3(256), 0, ST, p);
  else
    hipLaunchKernelGGL(catskip_bwd_kernel<4>, dim3(nblocks((size_t)B * Hs * Ws * 4 * (Cs / 4))), dim3(256), 0, ST, p);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_softmax(const float* logits, int B, int H, int W, int cs, int C, int Ho, int Wo, float* prob,
                           c3d_stream stream) {
  C3D_REQUIRE(C <= 32 && C <= cs, "softmax: at most 32 classes");
  C3D_REQUIRE((int64_t)B * H * W < (1ll << 31), "softmax: more than 2^31 pixels");
  if (C % 4 == 0 && cs % 4 == 0)
    hipLaunchKernelGGL(softmax_kernel<true>, dim3(nblocks((size_t)B * Ho * Wo)), dim3(256), 0, ST, logits, B, H, W, cs, C,
                       Ho, Wo, prob);
  else
    hipLaunchKernelGGL(softmax_kernel<false>, dim3(nblocks((size_t)B * Ho * Wo)), dim3(256), 0, ST, logits, B, H, W, cs, C,
                       Ho, Wo, prob);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_softmax_bwd(const float* prob, const float* dprob, int B, int H, int W, int cs, int C, int Ho,
                               int Wo, float* dlogits, c3d_stream stream) {
  C3D_REQUIRE(C <= 32 && cs <= 32, "softmax_bwd: at most 32 classes / padded channels");
  C3D_REQUIRE((int64_t)B * H * W < (1ll << 31), "softmax_bwd: more than 2^31 pixels");
  if (C % 4 == 0 && cs % 4 == 0)
    hipLaunchKernelGGL(softmax_bwd_kernel<true>, dim3(nblocks((size_t)B * H * W)), dim3(256), 0, ST, prob, dprob, B, H, W,
                       cs, C, Ho, Wo, dlogits);
  else
    hipLaunchKernelGGL(softmax_bwd_kernel<false>, dim3(nblocks((size_t)B * H * W)), dim3(256), 0, ST, prob, dprob, B, H, W,
                       cs, C, Ho, Wo, dlogits);
  C3D_CHECK_LAUNCH();
  return 0;
}

static BlArgs bl_args(const float* src, int Hs, int Ws, int scs, int scoff, float* dst, int Hd, int Wd, int dcs,
                      int dcoff, int B, int C) {
  BlArgs p;
  p.src = src; p.Hs = Hs; p.Ws = Ws; p.scs = scs; p.scoff = scoff;
  p.dst = dst; p.Hd = Hd; p.Wd = Wd; p.dcs = dcs; p.dcoff = dcoff;
  p.B = B; p.C = C; p.bf = 0;
  p.ry = Hd > 1 ? (float)(Hs - 1) / (float)(Hd - 1) : 0.f;
  p.rx = Wd > 1 ? (float)(Ws - 1) / (float)(Wd - 1) : 0.f;
  return p;
}

static inline int bl_chunks(int row_elems, int it) { return (row_elems + 256 * it - 1) / (256 * it); }
static inline int bl_shift(int q) { return (q & (q - 1)) == 0 ? __builtin_ctz(q) : -1; }

extern "C" int c3d_bilinear(const float* src, int Hs, int Ws, int scs, int scoff, float* dst, int Hd, int Wd, int dcs,
                            int dcoff, int B, int C, int bf16_mask, c3d_stream stream) {
  C3D_REQUIRE(C % 4 == 0 && scs % 4 == 0 && dcs % 4 == 0 && scoff % 4 == 0 && dcoff % 4 == 0,
              "bilinear: channel counts/strides must be multiples of 4");
  BlArgs p = bl_args(src, Hs, Ws, scs, scoff, dst, Hd, Wd, dcs, dcoff, B, C);
  p.bf = bf16_mask;
  C3D_REQUIRE((int64_t)Wd * dcs < (1ll << 31) && (int64_t)Ws * scs < (1ll << 31), "bilinear: a row exceeds 2^31 elements");
  const int V = (bf16_mask && C % 8 == 0 && scs % 8 == 0 && dcs % 8 == 0 && scoff % 8 == 0 && dcoff % 8 == 0) ? 8 : 4;
  const int chunks = bl_chunks(Wd * (C / V), BL_IT);
  C3D_REQUIRE((int64_t)B * Hd * chunks < (1ll << 31), "bilinear: grid too large");
  const int grid = B * Hd * chunks;
  if (V == 8)
    hipLaunchKernelGGL(bilinear_kernel<8>, dim3(grid), dim3(256), 0, ST, p, chunks, bl_shift(C / 8));
  else
    hipLaunchKernelGGL(bilinear_kernel<4>, dim3(grid), dim3(256), 0, ST, p, chunks, bl_shift(C / 4));
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_bilinear_sum2(const float* src1, int Hs1, int Ws1, const float* src2, int Hs2, int Ws2, float* dst, int Hd,
                                 int Wd, int B, int C, int dst_bf16, c3d_stream stream) {
  C3D_REQUIRE(src1 && src2 && dst && C % 4 == 0, "bilinear_sum2: two sources and C % 4 == 0 required");
  Bl2Args q;
  q.a = bl_args(src1, Hs1, Ws1, C, 0, dst, Hd, Wd, C, 0, B, C);
  q.a.bf = dst_bf16 ? 2 : 0;          // the sources are fp32 (low-resolution conv outputs); dst may be a bf16 activation
  q.src2 = src2; q.Hs2 = Hs2; q.Ws2 = Ws2; q.scs2 = C;
  q.ry2 = Hd > 1 ? (float)(Hs2 - 1) / (float)(Hd - 1) : 0.f;
  q.rx2 = Wd > 1 ? (float)(Ws2 - 1) / (float)(Wd - 1) : 0.f;
  C3D_REQUIRE((int64_t)Wd * C < (1ll << 31), "bilinear_sum2: a row exceeds 2^31 elements");
  // (eight channels per thread for a bf16 result -- 16-byte stores -- measured slower: 399 vs 333 us at 8 x 32 x 1024 x 704)
  const int Q = C / 4;
  const int qb = (Q > 16 && Q % 16 == 0) ? 16 : Q;            // quads per channel block (64 channels), or no blocking
  const int chunks = bl_chunks(Wd * qb, BL_IT);
  C3D_REQUIRE((int64_t)(Q / qb) * B * Hd * chunks < (1ll << 31), "bilinear_sum2: grid too large");
  hipLaunchKernelGGL(bilinear_sum2_kernel, dim3((Q / qb) * B * Hd * chunks), dim3(256), 0, ST, q, chunks, qb, bl_shift(qb));
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_bilinear_bwd(float* dsrc, int Hs, int Ws, int scs, int scoff, const float* ddst, int Hd, int Wd,
                                int dcs, int dcoff, int B, int C, int accumulate, int bf16_mask, const uint32_t* ddst_rowmask,
                                c3d_stream stream) {
  C3D_REQUIRE(C % 4 == 0 && scs % 4 == 0 && dcs % 4 == 0 && scoff % 4 == 0 && dcoff % 4 == 0,
              "bilinear_bwd: channel counts/strides must be multiples of 4");
  BlArgs p = bl_args(dsrc, Hs, Ws, scs, scoff, const_cast<float*>(ddst), Hd, Wd, dcs, dcoff, B, C);
  p.bf = bf16_mask;
  C3D_REQUIRE((int64_t)Wd * dcs < (1ll << 31) && (int64_t)Ws * scs < (1ll << 31), "bilinear_bwd: a row exceeds 2^31 elements");
  const int V = (bf16_mask && C % 8 == 0 && scs % 8 == 0 && dcs % 8 == 0 && scoff % 8 == 0 && dcoff % 8 == 0) ? 8 : 4;
  const int chunks = bl_chunks(Ws * (C / V), BL_IT_BWD);
  C3D_REQUIRE((int64_t)B * Hs * chunks < (1ll << 31), "bilinear_bwd: grid too large");
  C3D_REQUIRE(!ddst_rowmask || (int64_t)B * Hd * Wd < (1ll << 32), "bilinear_bwd: row mask needs < 2^32 destination pixels");
  const int grid = B * Hs * chunks;
  if (V == 8)
    hipLaunchKernelGGL(bilinear_bwd_kernel<8>, dim3(grid), dim3(256), 0, ST, p, accumulate, chunks, bl_shift(C / 8), ddst_rowmask);
  else
    hipLaunchKernelGGL(bilinear_bwd_kernel<4>, dim3(grid), dim3(256), 0, ST, p, accumulate, chunks, bl_shift(C / 4), ddst_rowmask);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_bilinear_bwd_rows(float* dsrc, int Hs, int Ws, int scs, int scoff, const float* drows, const int32_t* cmap,
                                     const uint32_t* rowmask, int Hd, int Wd, int B, int C, int accumulate, int dsrc_bf16,
                                     c3d_stream stream) {
  C3D_REQUIRE(drows && cmap && rowmask, "bilinear_bwd_rows: drows, cmap and rowmask are required");
  C3D_REQUIRE(C % 4 == 0 && scs % 4 == 0 && scoff % 4 == 0, "bilinear_bwd_rows: channel counts/strides must be multiples of 4");
  BlArgs p = bl_args(dsrc, Hs, Ws, scs, scoff, const_cast<float*>(drows), Hd, Wd, C, 0, B, C);
  p.bf = dsrc_bf16 ? 1 : 0;
  C3D_REQUIRE((int64_t)Ws * scs < (1ll << 31) && (int64_t)B * Hd * Wd < (1ll << 31), "bilinear_bwd_rows: sizes exceed 2^31");
  const int64_t nsrc = (int64_t)B * Hs * Ws;
  C3D_REQUIRE(nsrc < (1ll << 31) && (nsrc + 3) / 4 < (1ll << 31), "bilinear_bwd_rows: grid too large");
  hipLaunchKernelGGL(bilinear_bwd_rows_kernel, dim3((unsigned)((nsrc + 3) / 4)), dim3(256), 0, ST, p, accumulate, rowmask, cmap,
                     (int)nsrc);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_bilinear_rows(const float* src, int Hs, int Ws, int scs, int scoff, int src_bf16, int Hd, int Wd, int B,
                                 int C, const int32_t* img, const void* idx, int idx64, int A, const int32_t* count, int R,
                                 int l2, float eps, float* out, float* norm, c3d_stream stream) {
  C3D_REQUIRE(src && idx && out && C % 4 == 0 && scs % 4 == 0 && scoff % 4 == 0 && A >= 1,
              "bilinear_rows: src, idx, out required; channel counts/strides multiples of 4; A >= 1");
  C3D_REQUIRE(!(img && idx64), "bilinear_rows: (img, idx) pairs use int32 indices");
  if (R <= 0) return 0;
  BlArgs p = bl_args(src, Hs, Ws, scs, scoff, nullptr, Hd, Wd, C, 0, B, C);
  p.bf = src_bf16 ? 1 : 0;
  const int grid = nblocks((size_t)R, 4);
  if (idx64)
    hipLaunchKernelGGL(bilinear_rows_kernel<int64_t>, dim3(grid), dim3(256), 0, ST, p, img, (const int64_t*)idx, A, count, R, l2,
                       eps, out, norm);
  else
    hipLaunchKernelGGL(bilinear_rows_kernel<int32_t>, dim3(grid), dim3(256), 0, ST, p, img, (const int32_t*)idx, A, count, R, l2,
                       eps, out, norm);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_l2norm(const float* x, int64_t n, int C, float eps, float* y, float* norm, int bf16_mask, c3d_stream stream) {
  C3D_REQUIRE(C % 4 == 0, "l2norm: C must be a multiple of 4");
  if (bf16_mask && C % 8 == 0)
    hipLaunchKernelGGL(l2norm_kernel<8>, dim3(nblocks((size_t)n, 4)), dim3(256), 0, ST, x, (size_t)n, C, eps, y, norm, bf16_mask);
  else
    hipLaunchKernelGGL(l2norm_kernel<4>, dim3(nblocks((size_t)n, 4)), dim3(256), 0, ST, x, (size_t)n, C, eps, y, norm, bf16_mask);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_l2norm_bwd(const float* y, const float* norm, const float* dy, int64_t n, int C, float eps,
                              float* dx, int bf16_mask, c3d_stream stream) {
  C3D_REQUIRE(C % 4 == 0, "l2norm: C must be a multiple of 4");
  if (bf16_mask && C % 8 == 0)
    hipLaunchKernelGGL(l2norm_bwd_kernel<8>, dim3(nblocks((size_t)n, 4)), dim3(256), 0, ST, y, norm, dy, (size_t)n, C, eps, dx, bf16_mask);
  else
    hipLaunchKernelGGL(l2norm_bwd_kernel<4>, dim3(nblocks((size_t)n, 4)), dim3(256), 0, ST, y, norm, dy, (size_t)n, C, eps, dx, bf16_mask);
  C3D_CHECK_LAUNCH();
  return 0;
}
