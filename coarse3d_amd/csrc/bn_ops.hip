// Train-mode BatchNorm2d pieces around the conv engine (gfx950, HBM-bound reductions).
//
// Forward: the conv epilogue leaves per-tile (sum, sumsq) partials; stat_reduce folds them in
// fp64, bn_finalize turns them into the per-channel affine (scale, shift) that CONSUMERS apply
// on load, and updates running statistics (momentum 0.1, unbiased variance) exactly like
// nn.BatchNorm2d in pc_processor/models/salsanext_proto.py:46,50,89-105,168-180 and
// projector.py:20.  Between stat_reduce and bn_finalize the fp64 sums can be all-reduced across
// ranks (SyncBatchNorm, tasks/weak_segmentation/trainer.py:54).
//
// Backward (autograd of LeakyReLU -> BatchNorm, or BatchNorm -> LeakyReLU for the projector):
//   bn_bwd_reduce : per-channel sum(dy), sum(dy * a)            (partials, then stat_reduce)
//   bn_bwd_coeffs : k1,k2,k3 with  da = k1*dy + k2*a + k3, plus dgamma, dbeta
//   bn_bwd_apply  : dz = LeakyReLU'(.) * da   and per-channel sum(dz) partials (bias gradient)
#include "common.h"
#include "../../include/coarse3d_hip.h"

namespace {

// partial [C][2][n] fp32  ->  sums [C][2] fp64.  One block per channel, contiguous reads.
__global__ __launch_bounds__(256) void stat_reduce_kernel(const float* __restrict__ partial, int n, int C,
                                                         double* __restrict__ sums, double* __restrict__ sums_copy = nullptr) {
  __shared__ double red[2][4];
  const int c = blockIdx.x;
  const float* p = partial + (size_t)c * 2 * n;
  double s1 = 0.0, s2 = 0.0;
  // (eight iterations' loads in flight before their additions: the loop was one L2 round trip per iteration -- 16 of them at
  //  4 096 partials, most of the kernel's 6 us; the additions keep their order: the same bits)
#pragma unroll 8
  for (int t = threadIdx.x; t < n; t += 256) {
    s1 += (double)p[t];
    s2 += (double)p[n + t];
  }
  s1 = c3d_wave_sum_d(s1);
  s2 = c3d_wave_sum_d(s2);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s1;
    red[1][threadIdx.x >> 6] = s2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    sums[c * 2 + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    sums[c * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    if (sums_copy) {        // the rank-local copy a data-parallel BatchNorm backward keeps next to the all-reduced sums
      sums_copy[c * 2 + 0] = sums[c * 2 + 0];
      sums_copy[c * 2 + 1] = sums[c * 2 + 1];
    }
  }
}

// block-wide fold of one channel's partials [2][n] -> (s1, s2) in fp64, result valid in thread 0
__device__ __forceinline__ void fold_channel(const float* __restrict__ p, int n, double& s1, double& s2) {
  __shared__ double red[2][4];
  s1 = 0.0;
  s2 = 0.0;
  // (eight iterations' loads in flight before their additions: the loop was one L2 round trip per iteration -- 16 of them at
  //  4 096 partials, most of the kernel's 6 us; the additions keep their order: the same bits)
#pragma unroll 8
  for (int t = threadIdx.x; t < n; t += 256) {
    s1 += (double)p[t];
    s2 += (double)p[n + t];
  }
  s1 = c3d_wave_sum_d(s1);
  s2 = c3d_wave_sum_d(s2);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s1;
    red[1][threadIdx.x >> 6] = s2;
  }
  __syncthreads();
  s1 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  s2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
}

// single-rank fast paths: fold the partials and finish in ONE launch (one block per channel)
__global__ __launch_bounds__(256) void bn_finalize_partials_kernel(
    const float* __restrict__ partial, int n, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* running_mean, float* running_var, float momentum, float eps,
    float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ save_mean,
    float* __restrict__ save_invstd) {
  const int c = blockIdx.x;
  double s1, s2;
  fold_channel(partial + (size_t)c * 2 * n, n, s1, s2);
  if (threadIdx.x != 0) return;
  const double mean = s1 / count;
  double var = s2 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  save_mean[c] = (float)mean;
  save_invstd[c] = invstd;
  if (running_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_coeffs_partials_kernel(
    const float* __restrict__ partial, int n, double count, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma, float* __restrict__ k1,
    float* __restrict__ k2, float* __restrict__ k3, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x;
  double sdy, sdya;
  fold_channel(partial + (size_t)c * 2 * n, n, sdy, sdya);
  if (threadIdx.x != 0) return;
  const double mu = mean[c], is = invstd[c], g = gamma[c];
  const double sdyx = is * (sdya - mu * sdy);
  const double kk2 = -g * is * is * sdyx / count;
  k1[c] = (float)(g * is);
  k2[c] = (float)kk2;
  k3[c] = (float)(-g * is * sdy / count - kk2 * mu);
  dgamma[c] = (float)sdyx;
  dbeta[c] = (float)sdy;
}

__global__ __launch_bounds__(256) void bias_from_partials_kernel(const float* __restrict__ partial, int n,
                                                                 float* __restrict__ out, int accumulate) {
  const int c = blockIdx.x;
  double s1, s2;
  fold_channel(partial + (size_t)c * 2 * n, n, s1, s2);
  if (threadIdx.x == 0) out[c] = accumulate ? out[c] + (float)s1 : (float)s1;
}

__global__ void bn_finalize_kernel(const double* __restrict__ sums, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* running_mean, float* running_var,
                                   float momentum, float eps, int C, float* __restrict__ scale,
                                   float* __restrict__ shift, float* __restrict__ save_mean,
                                   float* __restrict__ save_invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mean = sums[c * 2] / count;
  double var = sums[c * 2 + 1] / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  save_mean[c] = (float)mean;
  save_invstd[c] = invstd;
  if (running_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                      int C, float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(rv[c] + eps);
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}

struct BwdArgs {
  const float* dy; int dy_cs;
  const float* a; int a_cs;
  int npix, C, mode;
  const float* pre_scale; const float* pre_shift;   // mode 1: y = a*pre_scale + pre_shift
  const float* k1; const float* k2; const float* k3;
  float* dz; int dz_cs;
  float* partial;            // [C][2][gridDim.x]
  int pix_per_block;
  float slope;
  int bf;                    // bit 0: dy, bit 1: a, bit 2: dz are bf16
  unsigned* gmax = nullptr;  // apply: atomicMax of the bits of max |dz| over the whole tensor (order-free, exact)
};

// thread = (pixel lane, channel quad); block walks a contiguous pixel chunk
template <bool APPLY, int V>
__global__ __launch_bounds__(256) void bn_bwd_kernel(BwdArgs p) {
  extern __shared__ float red[];  // [PL][C][2]
  const int Q = p.C / V;
  const int PL = max(256 / Q, 1);
  const int tid = threadIdx.x;
  const int pl = tid / Q, c = (tid % Q) * V;
  const bool active = tid < PL * Q;
  const int p0 = blockIdx.x * p.pix_per_block;
  const int p1 = min(p0 + p.pix_per_block, p.npix);
  c3d_vec<V> s1 = c3d_vzero<V>(), s2 = c3d_vzero<V>();
  if (active) {
    c3d_vec<V> ps, psh, k1, k2, k3;
#pragma unroll
    for (int q = 0; q < V; ++q) {
      ps.v[q] = 1.f;
      psh.v[q] = 0.f;
      k1.v[q] = 1.f;
      k2.v[q] = 0.f;
      k3.v[q] = 0.f;
    }
    if (p.mode == 1) {
      ps = c3d_vldf<V>(p.pre_scale, c);
      psh = c3d_vldf<V>(p.pre_shift, c);
    }
    if (APPLY && p.mode < 2) {
      k1 = c3d_vldf<V>(p.k1, c);
      k2 = c3d_vldf<V>(p.k2, c);
      k3 = c3d_vldf<V>(p.k3, c);
    }
    // UN pixels per trip: all 2*UN loads are issued before the first use (memory-level parallelism)
    constexpr int UN = V == 8 ? 2 : 4;
    auto body = [&](c3d_vec<V> dy, const c3d_vec<V>& a, int i) {
      if (p.mode == 1) {
#pragma unroll
        for (int q = 0; q < V; ++q) dy.v[q] *= (fmaf(a.v[q], ps.v[q], psh.v[q]) > 0.f) ? 1.f : p.slope;
      }
      if (!APPLY) {
#pragma unroll
        for (int q = 0; q < V; ++q) {
          s1.v[q] += dy.v[q];
          s2.v[q] += dy.v[q] * a.v[q];
        }
      } else {
        c3d_vec<V> dz;
#pragma unroll
        for (int q = 0; q < V; ++q) {
          // (explicit operation order: the weight-gradient kernel that applies this on load, wgrad_tr.hip, produces
          //  the same bits)
          float da = fmaf(k2.v[q], a.v[q], fmaf(k1.v[q], dy.v[q], k3.v[q]));
          if (p.mode == 0 || p.mode == 2) da *= (a.v[q] > 0.f) ? 1.f : p.slope;
          dz.v[q] = da;
        }
        c3d_vst<V>(p.dz, (size_t)i * p.dz_cs + c, p.bf & 4, dz);
        s1 += dz;
#pragma unroll
        for (int q = 0; q < V; ++q) s2.v[q] = fmaxf(s2.v[q], fabsf(dz.v[q]));     // row 1 of the apply partials: max |dz|
      }
    };
    int i = p0 + pl;
    for (; i + (UN - 1) * PL < p1; i += UN * PL) {
      c3d_vec<V> dyv[UN], av[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        dyv[u] = c3d_vld<V>(p.dy, (size_t)(i + u * PL) * p.dy_cs + c, p.bf & 1);
        av[u] = c3d_vld<V>(p.a, (size_t)(i + u * PL) * p.a_cs + c, p.bf & 2);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) body(dyv[u], av[u], i + u * PL);
    }
    for (; i < p1; i += PL)
      body(c3d_vld<V>(p.dy, (size_t)i * p.dy_cs + c, p.bf & 1), c3d_vld<V>(p.a, (size_t)i * p.a_cs + c, p.bf & 2), i);
#pragma unroll
    for (int q = 0; q < V; ++q) {
      red[(pl * p.C + c + q) * 2 + 0] = s1.v[q];
      red[(pl * p.C + c + q) * 2 + 1] = s2.v[q];
    }
  }
  __syncthreads();
  for (int ch = tid; ch < p.C; ch += 256) {
    float t1 = 0.f, t2 = 0.f;
    for (int k = 0; k < PL; ++k) {
      t1 += red[(k * p.C + ch) * 2 + 0];
      if (APPLY) t2 = fmaxf(t2, red[(k * p.C + ch) * 2 + 1]);      // (max |dz|: c3d_grad_exponent reads it)
      else t2 += red[(k * p.C + ch) * 2 + 1];
    }
    float* o = p.partial + (size_t)ch * 2 * gridDim.x + blockIdx.x;   // [C][2][nblk]
    o[0] = t1;
    o[gridDim.x] = t2;
    if (APPLY && p.gmax && t2 > 0.f) atomicMax(p.gmax, __float_as_uint(t2));     // positive floats order like their bits
  }
}

__global__ void bn_bwd_coeffs_kernel(const double* __restrict__ sums, const double* __restrict__ sums_param,
                                     double count, const float* __restrict__ mean,
                                     const float* __restrict__ invstd, const float* __restrict__ gamma, int C,
                                     float* __restrict__ k1, float* __restrict__ k2, float* __restrict__ k3,
                                     float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double sdy = sums[c * 2], sdya = sums[c * 2 + 1];
  const double mu = mean[c], is = invstd[c], g = gamma[c];
  const double sdyx = is * (sdya - mu * sdy);   // sum(dy * xhat)
  const double kk1 = g * is;
  const double kk2 = -g * is * is * sdyx / count;
  const double kk3 = -g * is * sdy / count - kk2 * mu;
  k1[c] = (float)kk1;
  k2[c] = (float)kk2;
  k3[c] = (float)kk3;
  // parameter gradients come from THIS rank's sums (data parallel averages them afterwards,
  // as torch.nn.SyncBatchNorm does); the input-gradient coefficients use the global sums
  const double ldy = sums_param[c * 2], ldya = sums_param[c * 2 + 1];
  dgamma[c] = (float)(is * (ldya - mu * ldy));
  dbeta[c] = (float)ldy;
}

// column `col` of fp64 sums [C][2] -> fp32 vector (bias gradients)
__global__ void sums_to_f32_kernel(const double* __restrict__ sums, int C, int col, float* __restrict__ out,
                                   int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float v = (float)sums[c * 2 + col];
  out[c] = accumulate ? out[c] + v : v;
}

}  // namespace

extern "C" int c3d_stat_reduce(const float* partial, int n, int C, double* sums, c3d_stream stream) {
  hipLaunchKernelGGL(stat_reduce_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partial, n, C, sums, (double*)nullptr);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_stat_reduce2(const float* partial, int n, int C, double* sums, double* sums_copy, c3d_stream stream) {
  hipLaunchKernelGGL(stat_reduce_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partial, n, C, sums, sums_copy);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_bn_finalize(const double* sums, double count, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, float eps, int C,
                               float* scale, float* shift, float* save_mean, float* save_invstd,
                               c3d_stream stream) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, count,
                     gamma, beta, running_mean, running_var, momentum, eps, C, scale, shift, save_mean,
                     save_invstd);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, int C, float* scale, float* shift,
                                  c3d_stream stream) {
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, eps, C, scale, shift);
  C3D_CHECK_LAUNCH();
  return 0;
}

// Blocks of >= 32 pixels, at most 1024 of them.  (256-pixel blocks left the 4 096- and 16 384-pixel maps of the
// deep levels on 16 / 64 blocks: 17-20 us per launch for 12-50 MB of traffic, a latency-bound serial walk.)
extern "C" int c3d_bn_bwd_num_blocks(int npix) {
  int nb = (npix + 31) / 32;
  return nb > 1024 ? 1024 : (nb < 1 ? 1 : nb);
}

static int bn_bwd_launch(bool apply, const float* dy, int dy_cs, const float* a, int a_cs, int npix, int C, int mode,
                         const float* pre_scale, const float* pre_shift, const float* k1, const float* k2,
                         const float* k3, float* dz, int dz_cs, float* partial, float slope, int bf, hipStream_t st,
                         unsigned* gmax = nullptr) {
  C3D_REQUIRE(C % 4 == 0 && C <= 1024, "bn_bwd: C must be a multiple of 4 and <= 1024");
  C3D_REQUIRE(dy_cs % 4 == 0 && a_cs % 4 == 0 && (!apply || dz_cs % 4 == 0), "bn_bwd: strides must be multiples of 4");
  BwdArgs p;
  p.dy = dy; p.dy_cs = dy_cs; p.a = a; p.a_cs = a_cs; p.npix = npix; p.C = C; p.mode = mode;
  p.pre_scale = pre_scale; p.pre_shift = pre_shift; p.k1 = k1; p.k2 = k2; p.k3 = k3;
  p.dz = dz; p.dz_cs = dz_cs; p.partial = partial;
  p.slope = c3d_slope_or_default(slope);
  p.bf = bf;
  p.gmax = gmax;
  const int nb = c3d_bn_bwd_num_blocks(npix);
  p.pix_per_block = (npix + nb - 1) / nb;
  // 8 channels per lane when any tensor is bf16 (16-byte accesses), 4 otherwise (the fp32 path as before)
  const bool wide = bf != 0 && C % 8 == 0 && dy_cs % 8 == 0 && a_cs % 8 == 0 && (!apply || dz_cs % 8 == 0);
  const int Q = C / (wide ? 8 : 4);
  const int PL = 256 / Q > 0 ? 256 / Q : 1;
  const size_t lds = (size_t)PL * C * 2 * sizeof(float);
  if (wide) {
    if (apply) hipLaunchKernelGGL((bn_bwd_kernel<true, 8>), dim3(nb), dim3(256), lds, st, p);
    else hipLaunchKernelGGL((bn_bwd_kernel<false, 8>), dim3(nb), dim3(256), lds, st, p);
  } else {
    if (apply) hipLaunchKernelGGL((bn_bwd_kernel<true, 4>), dim3(nb), dim3(256), lds, st, p);
    else hipLaunchKernelGGL((bn_bwd_kernel<false, 4>), dim3(nb), dim3(256), lds, st, p);
  }
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_bn_bwd_reduce(const float* dy, int dy_cs, const float* a, int a_cs, int npix, int C, int mode,
                                 const float* pre_scale, const float* pre_shift, float* partial,
                                 float lrelu_slope, int bf16_mask, c3d_stream stream) {
  return bn_bwd_launch(false, dy, dy_cs, a, a_cs, npix, C, mode, pre_scale, pre_shift, nullptr, nullptr, nullptr,
                       nullptr, 0, partial, lrelu_slope, bf16_mask, (hipStream_t)stream);
}

extern "C" int c3d_bn_bwd_coeffs(const double* sums, const double* sums_param, double count, const float* mean,
                                 const float* invstd,
                                 const float* gamma, int C, float* k1, float* k2, float* k3, float* dgamma,
                                 float* dbeta, c3d_stream stream) {
  hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums,
                     sums_param ? sums_param : sums, count, mean, invstd, gamma, C, k1, k2, k3, dgamma, dbeta);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_bn_bwd_apply(const float* dy, int dy_cs, const float* a, int a_cs, int npix, int C, int mode,
                                const float* pre_scale, const float* pre_shift, const float* k1, const float* k2,
                                const float* k3, float* dz, int dz_cs, float* partial, float lrelu_slope, int bf16_mask,
                                c3d_stream stream) {
  return bn_bwd_launch(true, dy, dy_cs, a, a_cs, npix, C, mode, pre_scale, pre_shift, k1, k2, k3, dz, dz_cs, partial,
                       lrelu_slope, bf16_mask, (hipStream_t)stream);
}

// scale_out[0 .. scale_len) = 2^s, *inv_out = 2^-s with s = target_log2 - ceil(log2(max |dz|)), the maximum taken from row 1
// of c3d_bn_bwd_apply's partials ([C][2][n]); an all-zero gradient gives s = 0.  One workgroup.
__global__ __launch_bounds__(256) void grad_exponent_kernel(const float* __restrict__ partial, int n, int C, int target_log2,
                                                            float* __restrict__ scale_out, int scale_len,
                                                            float* __restrict__ inv_out) {
  __shared__ float red[256];
  float m = 0.f;
  for (int i = threadIdx.x; i < C * n; i += 256) {
    const int c = i / n, k = i - c * n;
    m = fmaxf(m, partial[((size_t)c * 2 + 1) * n + k]);
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  m = red[0];
  int s = 0;
  if (m > 0.f && m < INFINITY) {
    int e;
    const float f = frexpf(m, &e);                 // m = f * 2^e, f in [0.5, 1): ceil(log2 m) = e, or e - 1 for m = 2^(e-1)
    s = target_log2 - (f == 0.5f ? e - 1 : e);
  }
  const float up = ldexpf(1.f, s), down = ldexpf(1.f, -s);
  for (int i = threadIdx.x; i < scale_len; i += 256) scale_out[i] = up;
  if (threadIdx.x == 0) *inv_out = down;
}

extern "C" int c3d_grad_exponent(const float* partial, int n, int C, int target_log2, float* scale_out, int scale_len,
                                 float* inv_out, c3d_stream stream) {
  C3D_REQUIRE(partial && scale_out && inv_out && n > 0 && C > 0 && scale_len > 0, "grad_exponent: null pointer or empty size");
  hipLaunchKernelGGL(grad_exponent_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n, C, target_log2, scale_out,
                     scale_len, inv_out);
  C3D_CHECK_LAUNCH();
  return 0;
}

// c3d_bn_bwd_apply that also folds max |dz| of the whole tensor into *gmax (bits of a non-negative float, pre-zeroed by the
// caller, atomicMax: exact and order-free) -- the per-tensor exponent of the f16x2 gradient experiment without a reduction
// launch of its own
extern "C" int c3d_bn_bwd_apply_gmax(const float* dy, int dy_cs, const float* a, int a_cs, int npix, int C, int mode,
                                     const float* pre_scale, const float* pre_shift, const float* k1, const float* k2,
                                     const float* k3, float* dz, int dz_cs, float* partial, float lrelu_slope, int bf16_mask,
                                     uint32_t* gmax, c3d_stream stream) {
  return bn_bwd_launch(true, dy, dy_cs, a, a_cs, npix, C, mode, pre_scale, pre_shift, k1, k2, k3, dz, dz_cs, partial,
                       lrelu_slope, bf16_mask, (hipStream_t)stream, gmax);
}

__global__ void grad_exponent_max_kernel(const unsigned* __restrict__ gmax, int target_log2, float* __restrict__ scale_out,
                                         int scale_len, float* __restrict__ inv_out) {
  const float m = __uint_as_float(*gmax);
  int s = 0;
  if (m > 0.f && m < INFINITY) {
    int e;
    const float f = frexpf(m, &e);
    s = target_log2 - (f == 0.5f ? e - 1 : e);
  }
  const float up = ldexpf(1.f, s), down = ldexpf(1.f, -s);
  for (int i = threadIdx.x; i < scale_len; i += 256) scale_out[i] = up;
  if (threadIdx.x == 0) *inv_out = down;
}

extern "C" int c3d_grad_exponent_max(const uint32_t* gmax, int target_log2, float* scale_out, int scale_len, float* inv_out,
                                     c3d_stream stream) {
  C3D_REQUIRE(gmax && scale_out && inv_out && scale_len > 0, "grad_exponent_max: null pointer or empty size");
  hipLaunchKernelGGL(grad_exponent_max_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, gmax, target_log2, scale_out,
                     scale_len, inv_out);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_sums_to_f32(const double* sums, int C, int col, float* out, int accumulate, c3d_stream stream) {
  hipLaunchKernelGGL(sums_to_f32_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, C, col, out,
                     accumulate);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_bn_finalize_partials(const float* partial, int n, double count, const float* gamma,
                                        const float* beta, float* running_mean, float* running_var, float momentum,
                                        float eps, int C, float* scale, float* shift, float* save_mean,
                                        float* save_invstd, c3d_stream stream) {
  hipLaunchKernelGGL(bn_finalize_partials_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partial, n, count, gamma,
                     beta, running_mean, running_var, momentum, eps, scale, shift, save_mean, save_invstd);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_bn_bwd_coeffs_partials(const float* partial, int n, double count, const float* mean,
                                          const float* invstd, const float* gamma, int C, float* k1, float* k2,
                                          float* k3, float* dgamma, float* dbeta, c3d_stream stream) {
  hipLaunchKernelGGL(bn_bwd_coeffs_partials_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partial, n, count,
                     mean, invstd, gamma, k1, k2, k3, dgamma, dbeta);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_bias_from_partials(const float* partial, int n, int C, float* out, int accumulate,
                                      c3d_stream stream) {
  hipLaunchKernelGGL(bias_from_partials_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partial, n, out,
                     accumulate);
  C3D_CHECK_LAUNCH();
  return 0;
}
