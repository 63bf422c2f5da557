// Prototype memory bank pipeline (gfx950): pixel-to-prototype similarity, Sinkhorn assignment,
// per-class masked feature reduction, EMA update.
// Reference: pc_processor/models/salsanext_proto.py:494-510 (similarity), :337-402
// (prototype_learning), pc_processor/models/sinkhorn.py:5-33.
// The [N,256] x [256, M*C] similarity GEMM itself runs on the MFMA conv engine (1x1 "conv").
#include "common.h"
#include "../../include/coarse3d_hip.h"

namespace {

// out = l2_normalize(LayerNorm(x))  per row; one wave per row (salsanext_proto.py:497-501)
// (tried: next row prefetched into registers + nontemporal stores -- 87 VGPRs, 5 waves per SIMD: 431 -> 463 us)
__global__ __launch_bounds__(256) void rownorm_kernel(const float* __restrict__ x, size_t n, int C,
                                                      const float* __restrict__ w, const float* __restrict__ b,
                                                      float ln_eps, float l2_eps, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t r = wave; r < n; r += nw) {
    float v[16];  // up to C = 1024
    int cnt = 0;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(x + r * C + c);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[cnt * 4 + q] = t[q];
        s += t[q];
      }
      ++cnt;
    }
    const float mean = c3d_wave_sum(s) / (float)C;
    float ss = 0.f;
    for (int i = 0; i < cnt * 4; ++i) {
      const float d = v[i] - mean;
      ss += d * d;
    }
    const float rstd = rsqrtf(c3d_wave_sum(ss) / (float)C + ln_eps);
    float nn = 0.f;
    int i = 0;
    for (int c = lane * 4; c < C; c += 256) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float y = (v[i * 4 + q] - mean) * rstd * w[c + q] + b[c + q];
        v[i * 4 + q] = y;
        nn += y * y;
      }
      ++i;
    }
    const float inv = 1.f / fmaxf(sqrtf(c3d_wave_sum(nn)), l2_eps);
    i = 0;
    for (int c = lane * 4; c < C; c += 256) {
      f32x4 t;
#pragma unroll
      for (int q = 0; q < 4; ++q) t[q] = v[i * 4 + q] * inv;
      *reinterpret_cast<f32x4*>(out + r * C + c) = t;
      ++i;
    }
  }
}

// sim [N][M*C] (column m*C+k) -> nearest[N][C] = LayerNorm_C(max_m sim), pred[N] = argmax_k
// one wave per row, row staged in LDS  (salsanext_proto.py:506-507, :340)
__global__ __launch_bounds__(256) void proto_nearest_kernel(const float* __restrict__ sim, size_t n, int M, int C,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            float eps, float* __restrict__ nearest,
                                                            int32_t* __restrict__ pred) {
  extern __shared__ float srow[];  // [4][M*C]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int MC = M * C;
  float* row = srow + wv * MC;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t r = wave; r < n; r += nw) {
    for (int j = lane; j < MC; j += 64) row[j] = sim[r * MC + j];
    __builtin_amdgcn_wave_barrier();
    float mx = -INFINITY;
    if (lane < C)
      for (int m = 0; m < M; ++m) mx = fmaxf(mx, row[m * C + lane]);
    const float val = lane < C ? mx : 0.f;
    const float mean = c3d_wave_sum(val) / (float)C;
    const float d = lane < C ? (mx - mean) : 0.f;
    const float rstd = rsqrtf(c3d_wave_sum(d * d) / (float)C + eps);
    float y = lane < C ? d * rstd * w[lane] + b[lane] : -INFINITY;
    if (nearest && lane < C) nearest[r * C + lane] = y;
    // argmax over lanes (first index on ties)
    float best = y;
    int bi = lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) {
        best = ov;
        bi = oi;
      }
    }
    if (lane == 0) pred[r] = bi;
    __builtin_amdgcn_wave_barrier();
  }
}

// Ordered compaction: for group g and class c list the positions i (ascending) with
// labels[g][i] == c.  idx [groups][ncls][n], counts [groups][ncls].
// Two passes over SEG segments of each label row, so that groups*ncls*SEG workgroups share the
// work: (1) per-(group, segment) class histogram (all classes in one read of the labels);
// (2) each (class, group, segment) workgroup derives its output offset from the histograms of the
// earlier segments and compacts its own segment, 4096 labels per iteration (16 consecutive per
// thread, fetched coalesced through LDS), one wave scan + one barrier pair per iteration.
constexpr int CSEG = 8;

__global__ __launch_bounds__(1024) void compact_hist_kernel(const int64_t* __restrict__ labels,
                                                           const uint8_t* __restrict__ keep, int n, int ncls, int seg_len,
                                                           int32_t* __restrict__ seg_counts) {   // [groups][CSEG][ncls]
  __shared__ int h[64];
  const int sg = blockIdx.x, g = blockIdx.y;
  if (threadIdx.x < 64) h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t* lab = labels + (size_t)g * n;
  const uint8_t* kp = keep ? keep + (size_t)g * n : nullptr;
  const int e = min(n, (sg + 1) * seg_len);
  for (int i = sg * seg_len + threadIdx.x; i < e; i += 1024) {      // 1024 threads: 16 trips per 16 384-label segment, not 64
    int64_t l = lab[i];
    if (kp && !kp[i]) l = 0;
    if (l >= 0 && l < ncls) atomicAdd(&h[(int)l], 1);
  }
  __syncthreads();
  if (threadIdx.x < ncls) seg_counts[((size_t)g * CSEG + sg) * ncls + threadIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(256) void group_compact_kernel(const int64_t* __restrict__ labels,
                                                            const uint8_t* __restrict__ keep, int n, int ncls, int seg_len,
                                                            const int32_t* __restrict__ seg_counts,
                                                            int32_t* __restrict__ counts, int32_t* __restrict__ idx) {
  constexpr int PT = 16;
  __shared__ int wtot[2][4];
  __shared__ unsigned char flag[256 * PT];
  const int c = blockIdx.x, g = blockIdx.y, sg = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t* lab = labels + (size_t)g * n;
  const uint8_t* kp = keep ? keep + (size_t)g * n : nullptr;
  int32_t* out = idx + ((size_t)g * ncls + c) * n;
  int base = 0, total = 0;
  for (int k = 0; k < CSEG; ++k) {
    const int v = seg_counts[((size_t)g * CSEG + k) * ncls + c];
    if (k < sg) base += v;
    total += v;
  }
  if (sg == 0 && tid == 0) counts[g * ncls + c] = total;
  const int e = min(n, (sg + 1) * seg_len);
  int buf = 0;
  for (int i0 = sg * seg_len; i0 < e; i0 += 256 * PT, buf ^= 1) {
#pragma unroll
    for (int q = 0; q < PT; ++q) {           // coalesced fetch -> per-label flag byte
      const int i = i0 + q * 256 + tid;
      unsigned char fl = 0;
      if (i < e) {
        int64_t l = lab[i];
        if (kp && !kp[i]) l = 0;
        fl = (l == c);
      }
      flag[q * 256 + tid] = fl;
    }
    __syncthreads();
    unsigned f = 0;
#pragma unroll
    for (int q = 0; q < PT; ++q) f |= (unsigned)flag[tid * PT + q] << q;
    const int cnt = __popc(f);
    int scan = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(scan, o, 64);
      if (lane >= o) scan += v;
    }
    if (lane == 63) wtot[buf][wv] = scan;
    __syncthreads();
    int off = base + scan - cnt;
    for (int k = 0; k < wv; ++k) off += wtot[buf][k];
    const int i = i0 + tid * PT;
    while (f) {
      const int q = __ffs(f) - 1;
      out[off++] = i + q;
      f &= f - 1;
    }
    base += wtot[buf][0] + wtot[buf][1] + wtot[buf][2] + wtot[buf][3];
  }
}

// counts[g][c] = #{i : labels[g][i] == c}   (class presence for the pseudo-label selection)
__global__ __launch_bounds__(256) void label_hist_kernel(const int64_t* __restrict__ labels, int n, int ncls,
                                                         int32_t* __restrict__ counts) {
  __shared__ int h[64];
  const int g = blockIdx.y;
  if (threadIdx.x < 64) h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t* lab = labels + (size_t)g * n;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int64_t l = lab[i];
    if (l > 0 && l < ncls) atomicAdd(&h[(int)l], 1);     // class 0 (ignore) is never queried
  }
  __syncthreads();
  if (threadIdx.x < ncls && h[threadIdx.x]) atomicAdd(&counts[g * ncls + threadIdx.x], h[threadIdx.x]);
}

struct LearnArgs {
  const float* sim;      // [N][M*C]
  const float* feat;     // [N][D]  (LayerNorm + l2 rows)
  const int32_t* pred;   // [N] argmax class of the nearest-prototype map, or NULL: computed here
  const float* ln_w; const float* ln_b; float ln_eps;   // mask_norm LayerNorm (used when pred == NULL)
  const int32_t* counts; // [B][ncls]   per-image ordered lists from c3d_group_compact
  const int32_t* idx;    // [B][ncls][n]
  int32_t* rows;         // [ncls][N] scratch: flat (image-major) row list of each class
  int B, n;
  const float* noise;    // [N][M] Exp(1) variates (gumbel = -log), indexed by pixel
  const float* protos;   // [ncls][M][D] l2-normalised bank (input)
  float* protos_out;     // [ncls][M][D]
  float* target;         // [N]  (pre-zeroed)
  int32_t* assign;       // [N] scratch
  int N, M, C, D, ignore;
  float momentum;
  float* fsum;           // NULL, or [C][M][D+1]: write the masked feature sums + counts, skip the EMA
  const int32_t* cmap;   // NULL, or [N]: row of a (labelled) pixel in COMPACT sim / feat
  int noise_by_row;      // noise is [rows][M], indexed like sim / feat (compact rows) instead of by pixel
};

// EMA of the class's prototypes with the l2-normalised feature sums, then the final l2
// normalisation (salsanext_proto.py:376-395, :402); one wave per prototype row.  f [M][D] (LDS,
// overwritten), cnt [M], tot = sum(cnt).
__device__ __forceinline__ void proto_ema_rows(float* f, const float* cnt, float tot, bool any, const float* pin,
                                               float* pout, int M, int D, float momentum, int lane, int wv, int nwaves) {
  for (int m = wv; m < M; m += nwaves) {
    float ss = 0.f;
    for (int d = lane; d < D; d += 64) ss += f[m * D + d] * f[m * D + d];
    const float inv = 1.f / fmaxf(sqrtf(c3d_wave_sum(ss)), 1e-12f);
    const bool upd = any && tot > 0.f && cnt[m] != 0.f;
    float nn = 0.f;
    for (int d = lane; d < D; d += 64) {
      float v = pin[m * D + d];
      if (upd) v = momentum * v + (1.f - momentum) * (f[m * D + d] * inv);
      f[m * D + d] = v;
      nn += v * v;
    }
    const float inv2 = 1.f / fmaxf(sqrtf(c3d_wave_sum(nn)), 1e-12f);
    for (int d = lane; d < D; d += 64) pout[m * D + d] = f[m * D + d] * inv2;
  }
}

// 32-lane group helpers (a wave holds two independent groups; xor offsets < 32 stay inside one)
__device__ __forceinline__ float group32_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// argmax with the smallest index on ties (torch.argmax), result in every lane of the group
__device__ __forceinline__ int group32_argmax(float v, int idx) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ov > v || (ov == v && oi < idx)) {
      v = ov;
      idx = oi;
    }
  }
  return idx;
}

// one workgroup of 1024 threads per class; 32 labelled pixels in flight, 32 lanes (sub-prototype m, or class k for
// the nearest-prototype vote) per pixel.  (The whole kernel is a chain of dependent gathers over the class's
// labelled pixels -- ~55 per class at 0.1 % labels -- so its time is trips x memory latency: 8 pixels in flight
// on 256 threads took 91 us.)
constexpr int LEARN_THREADS = 1024;
constexpr int LEARN_SUBS = LEARN_THREADS / 32;
__global__ __launch_bounds__(LEARN_THREADS) void proto_learn_kernel(LearnArgs a) {
  extern __shared__ float sm[];
  double* red = reinterpret_cast<double*>(sm);          // [LEARN_SUBS][32] slot partials
  float* u = sm + 2 * LEARN_SUBS * 32;                  // [32] row scalings
  int* s_r = reinterpret_cast<int*>(u + 32);            // [256] staged pixel rows
  int* s_m = s_r + 256;                                 // [256] staged assignments
  float* f = u + 32 + 512;                              // [M][D]
  float* cnt = f + a.M * a.D;                           // [M]
  const int c = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int sub = tid >> 5, m = tid & 31;
  const int M = a.M, D = a.D, MC = a.M * a.C;
  int32_t* rows = a.rows + (size_t)c * a.N;
  int nc = 0;
  if (c != a.ignore) {
    for (int b = 0; b < a.B; ++b) {
      const int cb = a.counts[b * a.C + c];
      const int32_t* src = a.idx + ((size_t)b * a.C + c) * a.n;
      for (int i = tid; i < cb; i += LEARN_THREADS) rows[nc + i] = src[i] + b * a.n;     // flat pixel ids
      nc += cb;
    }
    __threadfence_block();
    __syncthreads();
  }
  const float* pin = a.protos + (size_t)c * M * D;
  float* pout = a.protos_out + (size_t)c * M * D;

  if (nc > 0) {
    // ---- Sinkhorn scalings: Q[i][m] = E[i][m] * u[m] * v[i],  E = exp(sim/0.05)
    if (tid < 32) u[tid] = 1.f;
    __syncthreads();
    for (int it = 0; it < 3; ++it) {
      double acc = 0.0;
      const float um = u[m];
      for (int i = sub; i < nc; i += LEARN_SUBS) {
        const float* s = a.sim + (size_t)(a.cmap ? a.cmap[rows[i]] : rows[i]) * MC + c;
        const float e = m < M ? expf(s[m * a.C] / 0.05f) : 0.f;
        const float colsum = group32_sum(e * um);
        // v[i] of the previous column step (it == 0: uniform, cancels in the row step)
        const float v = it == 0 ? 1.f : 1.f / colsum;
        acc += (double)(e * v);
      }
      red[sub * 32 + m] = acc;
      __syncthreads();
      if (tid < 32) {
        double t = 0.0;
        for (int k = 0; k < LEARN_SUBS; ++k) t += red[k * 32 + tid];
        // row step: u[m] = 1 / (K * sum_i E[i][m] v[i])  (independent of the previous u)
        u[tid] = tid < M ? (float)(1.0 / ((double)M * t)) : 0.f;
      }
      __syncthreads();
    }
    // ---- assignment per labelled pixel
    const float um = u[m];
    for (int i = sub; i < nc; i += LEARN_SUBS) {
      const int r = rows[i];
      const int cr = a.cmap ? a.cmap[r] : r;           // row of pixel r in sim / feat
      const float* s = a.sim + (size_t)cr * MC + c;
      const float e = m < M ? expf(s[m * a.C] / 0.05f) * um : 0.f;
      const float colsum = group32_sum(e);
      const float qq = m < M ? e / colsum : -INFINITY;
      const int best = group32_argmax(qq, m);
      const float hh = m < M ? (qq - logf(a.noise[(size_t)(a.noise_by_row ? cr : r) * M + m])) / 0.5f : -INFINITY;
      const int hot = group32_argmax(hh, m);
      int pred_r;
      if (a.pred) {
        pred_r = a.pred[r];
      } else {
        // nearest-prototype class of this labelled pixel, computed here instead of for all N
        // pixels: argmax_k LayerNorm_C(max_m sim[r][m][k])   (salsanext_proto.py:506-507, :340)
        const float* row = a.sim + (size_t)cr * MC;
        const int k = m;                    // lane = class
        float mx = -INFINITY;
        if (k < a.C)
          for (int mm = 0; mm < M; ++mm) mx = fmaxf(mx, row[mm * a.C + k]);
        const float mean = group32_sum(k < a.C ? mx : 0.f) / (float)a.C;
        const float dv = k < a.C ? mx - mean : 0.f;
        const float rstd = rsqrtf(group32_sum(dv * dv) / (float)a.C + a.ln_eps);
        const float y = k < a.C ? dv * rstd * a.ln_w[k] + a.ln_b[k] : -INFINITY;
        pred_r = group32_argmax(y, k);
      }
      if (m == 0) {
        a.target[r] = (float)(best + M * c);
        a.assign[r] = (pred_r == c) ? hot : -1;
      }
    }
  }
  // ---- masked reduction f[m][:] = sum feat rows, cnt[m]
  for (int j = tid; j < M * D; j += LEARN_THREADS) f[j] = 0.f;
  if (tid < M) cnt[tid] = 0.f;
  __threadfence_block();
  __syncthreads();
  for (int i0 = 0; i0 < nc; i0 += 256) {
    const int cn = min(256, nc - i0);
    if (tid < cn) {
      const int r = rows[i0 + tid];
      s_r[tid] = a.cmap ? a.cmap[r] : r;
      s_m[tid] = a.assign[r];
    }
    __syncthreads();
    for (int d = tid; d < D; d += LEARN_THREADS) {
      for (int ii = 0; ii < cn; ii += 8) {      // 8 feature rows in flight per thread
        float v[8];
        int mm[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          mm[q] = ii + q < cn ? s_m[ii + q] : -1;
          v[q] = mm[q] >= 0 ? a.feat[(size_t)s_r[ii + q] * D + d] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (mm[q] >= 0) f[mm[q] * D + d] += v[q];
      }
    }
    if (tid == 0)
      for (int ii = 0; ii < cn; ++ii)
        if (s_m[ii] >= 0) cnt[s_m[ii]] += 1.f;
    __syncthreads();
  }
  if (a.fsum) {          // data parallel "prototype sums" mode: hand the sums to the exchange, EMA later
    float* o = a.fsum + (size_t)c * M * (D + 1);
    for (int j = tid; j < M * D; j += LEARN_THREADS) o[(j / D) * (D + 1) + j % D] = f[j];
    if (tid < M) o[tid * (D + 1) + D] = cnt[tid];
    return;
  }
  float tot = 0.f;
  for (int m = 0; m < M; ++m) tot += cnt[m];
  proto_ema_rows(f, cnt, tot, nc > 0, pin, pout, M, D, a.momentum, lane, wv, LEARN_THREADS / 64);
}

// EMA from (all-reduced) sums: fsum [C][M][D+1] -> protos_out; one workgroup per class
__global__ __launch_bounds__(256) void proto_ema_kernel(const float* __restrict__ fsum, const float* __restrict__ protos,
                                                        float* __restrict__ protos_out, int M, int D, int ignore, float momentum) {
  extern __shared__ float sm[];
  float* f = sm;              // [M][D]
  float* cnt = sm + M * D;    // [M]
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* src = fsum + (size_t)c * M * (D + 1);
  for (int j = tid; j < M * D; j += 256) f[j] = src[(j / D) * (D + 1) + j % D];
  if (tid < M) cnt[tid] = src[tid * (D + 1) + D];
  __syncthreads();
  float tot = 0.f;
  for (int m = 0; m < M; ++m) tot += cnt[m];
  proto_ema_rows(f, cnt, tot, c != ignore, protos + (size_t)c * M * D, protos_out + (size_t)c * M * D, M, D, momentum, lane, wv, 4);
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int c3d_rownorm_ln_l2(const float* x, int64_t n, int C, const float* ln_w, const float* ln_b, float ln_eps,
                                 float l2_eps, float* out, c3d_stream stream) {
  C3D_REQUIRE(C % 4 == 0 && C <= 1024, "rownorm: C must be a multiple of 4 and <= 1024");
  size_t nb = ((size_t)n + 3) / 4;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(rownorm_kernel, dim3((int)nb), dim3(256), 0, ST, x, (size_t)n, C, ln_w, ln_b, ln_eps, l2_eps, out);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_proto_nearest(const float* sim, int64_t n, int M, int C, const float* ln_w, const float* ln_b,
                                 float eps, float* nearest, int32_t* pred, c3d_stream stream) {
  C3D_REQUIRE(C <= 64, "proto_nearest: at most 64 classes");
  size_t nb = ((size_t)n + 3) / 4;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(proto_nearest_kernel, dim3((int)nb), dim3(256), (size_t)4 * M * C * sizeof(float), ST, sim,
                     (size_t)n, M, C, ln_w, ln_b, eps, nearest, pred);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_group_compact(const int64_t* labels, const uint8_t* keep, int groups, int n, int ncls,
                                 int32_t* counts, int32_t* idx, int32_t* seg_scratch, c3d_stream stream) {
  C3D_REQUIRE(ncls <= 64, "group_compact: at most 64 classes");
  C3D_REQUIRE(seg_scratch != nullptr, "group_compact: scratch of groups*8*ncls int32 required");
  int seg_len = ((n + CSEG - 1) / CSEG + 4095) / 4096 * 4096;      // whole 4096-label iterations per segment
  hipLaunchKernelGGL(compact_hist_kernel, dim3(CSEG, groups), dim3(1024), 0, ST, labels, keep, n, ncls, seg_len, seg_scratch);
  C3D_CHECK_LAUNCH();
  hipLaunchKernelGGL(group_compact_kernel, dim3(ncls, groups, CSEG), dim3(256), 0, ST, labels, keep, n, ncls, seg_len,
                     seg_scratch, counts, idx);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_label_hist(const int64_t* labels, int groups, int n, int ncls, int32_t* counts, c3d_stream stream) {
  C3D_REQUIRE(ncls <= 64, "label_hist: at most 64 classes");
  (void)hipMemsetAsync(counts, 0, sizeof(int32_t) * groups * ncls, ST);
  int bx = (n + 256 * 16 - 1) / (256 * 16);
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(label_hist_kernel, dim3(bx, groups), dim3(256), 0, ST, labels, n, ncls, counts);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_proto_learn(const float* sim, const float* feat, const int32_t* pred, const float* ln_w,
                               const float* ln_b, float ln_eps, const int32_t* counts,
                               const int32_t* idx, int32_t* rows, const float* noise, const float* protos,
                               float* protos_out, float* target, int32_t* assign, int B, int n, int M, int C, int D,
                               int ignore_label, float momentum, float* fsum, const int32_t* cmap, int noise_by_row,
                               c3d_stream stream) {
  C3D_REQUIRE(M <= 32, "proto_learn: at most 32 sub-prototypes per class");
  C3D_REQUIRE(cmap == nullptr || pred == nullptr, "proto_learn: a precomputed argmax map is indexed by pixel; not with compact rows");
  const int N = B * n;
  LearnArgs a{sim, feat, pred, ln_w, ln_b, ln_eps, counts, idx, rows, B, n, noise, protos, protos_out, target, assign, N, M, C, D,
              ignore_label, momentum, fsum, cmap, noise_by_row};
  const size_t lds = (2 * LEARN_SUBS * 32 + 32 + 512 + (size_t)M * D + M) * sizeof(float);
  hipLaunchKernelGGL(proto_learn_kernel, dim3(C), dim3(LEARN_THREADS), lds, ST, a);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_proto_ema(const float* fsum, const float* protos, float* protos_out, int M, int C, int D,
                             int ignore_label, float momentum, c3d_stream stream) {
  C3D_REQUIRE(M <= 32, "proto_ema: at most 32 sub-prototypes per class");
  hipLaunchKernelGGL(proto_ema_kernel, dim3(C), dim3(256), ((size_t)M * D + M) * sizeof(float), ST, fsum, protos, protos_out,
                     M, D, ignore_label, momentum);
  C3D_CHECK_LAUNCH();
  return 0;
}
