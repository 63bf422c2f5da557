// Entropy-weighted pseudo-label selection, bit-exact anchor sampling and the pixel-to-prototype
// InfoNCE loss (gfx950).
// Reference: pc_processor/loss/contrast_pixel_loss.py:27-195 (ContrastMEMLoss),
// tasks/weak_segmentation/trainer.py:447-518 (entropy_based_selection).
//
// Index contracts (pinned against torch.multinomial in tests/test_oracle_golden.py):
//  * with replacement    : sequential fp32 running sum of the class weights, divided by the fp32
//    total, left-bisect of each float64 uniform draw;
//  * without replacement : top-k of weight / Exp(1) noise (fp32 division).
// Both samplers run one workgroup per (image, class) pair, all pairs concurrently, with no host
// synchronisation: absent pairs simply produce no work.
#include "common.h"
#include "../../include/coarse3d_hip.h"

namespace {

// ---------------------------------------------------------------- entropy weights + argmax
// prob [N][C] -> w_anchor = exp(-H^2), w_pl = exp(-H), amax (contrast_pixel_loss.py:46-49,
// trainer.py:459-466)
// One pixel per lane, the C probabilities in registers (16-byte loads when C % 4 == 0: a wave reads 64*C*4 consecutive
// bytes).  The entropy sum keeps the pairing of a 32-lane xor butterfly (16, 8, 4, 2, 1 over the zero-padded classes) --
// the first version of this kernel ran 32 lanes per pixel with fifteen dependent cross-lane shuffles per pixel and
// was bound by their latency (105 us per 8x64x2048 launch, 0.86 TB/s); this one produces the same bits.
// (The reference sums the classes in order; a tree differs by rounding only.)
template <bool V4>
__global__ __launch_bounds__(256) void entropy_stats_kernel(const float* __restrict__ prob, size_t n, int C,
                                                            float* __restrict__ w_anchor, float* __restrict__ w_pl,
                                                            int32_t* __restrict__ amax) {
  const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t tstride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = t0; i < n; i += tstride) {
    float v[32];
    if constexpr (V4) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (q * 4 < C) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(prob + i * C + q * 4);
#pragma unroll
          for (int k = 0; k < 4; ++k) v[q * 4 + k] = t[k];
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[q * 4 + k] = 0.f;
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < 32; ++c) v[c] = c < C ? prob[i * C + c] : 0.f;
    }
    float best = -INFINITY;
    int bi = 0;
    float h[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      if (c < C) {
        h[c] = __fmul_rn(v[c], logf(v[c] + 1e-10f));
        if (v[c] > best) {          // ties keep the lowest class, as the butterfly's (value, index) order did
          best = v[c];
          bi = c;
        }
      } else {
        h[c] = 0.f;
      }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1)
#pragma unroll
      for (int c = 0; c < o; ++c) h[c] = __fadd_rn(h[c], h[c + o]);
    const float e = -h[0];
    if (w_anchor) w_anchor[i] = expf(-(e * e));
    if (w_pl) w_pl[i] = expf(-e);
    if (amax) amax[i] = bi;
  }
}

// ---------------------------------------------------------------- pseudo-label selection (T3)
// Members of pair (b, c): pixels with amax == c and eval_label > 0, for classes c present in the
// weak labels of image b.  Three light passes bucket the members' keys (bits of w / noise) per
// pair, then one workgroup per pair radix-selects the k-th largest key of its bucket.
struct PlArgs {
  const float* w_pl;           // [B][n]
  const int32_t* amax;         // [B][n]
  const int64_t* eval_label;   // [B][n]
  const float* noise;          // [B][C][n] Exp(1)
  const int32_t* tl_counts;    // [B][C] number of weak labels of class c in image b
  int n, C, ignore;
  float ratio;
  const float* ratio_dev;      // NULL, or the ratio as a device scalar (a captured step must not bake the epoch in)
  int32_t* cnt;                // [B][C] members per pair
  int32_t* cursor;             // [B][C] fill cursor
  uint32_t* keys;              // [B][n] bucketed keys (bucket of (b,c) starts at offset[b][c])
  int32_t* pix;                // [B][n] pixel index of each key
  uint8_t* chosen;             // [B][n], pre-zeroed
};

__device__ __forceinline__ int pl_class(const PlArgs& a, int b, int i) {
  const int c = a.amax[(size_t)b * a.n + i];
  if (c == a.ignore || a.eval_label[(size_t)b * a.n + i] <= 0 || a.tl_counts[b * a.C + c] == 0) return -1;
  return c;
}

// grid (chunks, B): FILL == false counts members, FILL == true writes keys into the buckets
template <bool FILL>
__global__ __launch_bounds__(256) void pl_bucket_kernel(PlArgs a) {
  __shared__ int lcnt[64], lbase[64], loff[64];
  const int b = blockIdx.y, tid = threadIdx.x;
  if (tid < 64) lcnt[tid] = 0;
  __syncthreads();
  const int per = (a.n + gridDim.x - 1) / gridDim.x;
  const int i0 = blockIdx.x * per, i1 = min(i0 + per, a.n);
  for (int i = i0 + tid; i < i1; i += 256) {
    const int c = pl_class(a, b, i);
    if (c >= 0) atomicAdd(&lcnt[c], 1);
  }
  __syncthreads();
  if (!FILL) {
    if (tid < a.C && lcnt[tid]) atomicAdd(&a.cnt[b * a.C + tid], lcnt[tid]);
    return;
  }
  if (tid < a.C) {
    int off = 0;                                    // bucket start = prefix of the pair counts
    for (int c = 0; c < tid; ++c) off += a.cnt[b * a.C + c];
    loff[tid] = off;
    lbase[tid] = lcnt[tid] ? atomicAdd(&a.cursor[b * a.C + tid], lcnt[tid]) : 0;
    lcnt[tid] = 0;
  }
  __syncthreads();
  for (int i = i0 + tid; i < i1; i += 256) {
    const int c = pl_class(a, b, i);
    if (c >= 0) {
      const int pos = loff[c] + lbase[c] + atomicAdd(&lcnt[c], 1);
      const float q = a.w_pl[(size_t)b * a.n + i] / a.noise[((size_t)b * a.C + c) * a.n + i];
      a.keys[(size_t)b * a.n + pos] = __float_as_uint(q);
      a.pix[(size_t)b * a.n + pos] = i;
    }
  }
}

// grid (C, B).  Radix select (4 x 8 bits) of the k-th largest key of the pair's bucket.  1024 threads per pair and a
// parallel suffix scan over the 256 bins (five passes over ~7 000 keys on 256 threads plus four serial bin walks by
// thread 0 took 80 us).
constexpr int PL_THREADS = 1024;
constexpr int PL_MAX_TIES = 1024;
__global__ __launch_bounds__(PL_THREADS) void pl_select_kernel(PlArgs a) {
  __shared__ int hist[256];
  __shared__ int suf[2][256];
  __shared__ unsigned int s_prefix, s_k, s_take, s_nties;
  __shared__ int tie_pix[PL_MAX_TIES];
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int cnt = a.cnt[b * a.C + c];
  if (cnt == 0) return;
  const int k = (int)((float)cnt * (a.ratio_dev ? *a.ratio_dev : a.ratio));
  if (k < 1) return;
  int off = 0;
  for (int cc = 0; cc < c; ++cc) off += a.cnt[b * a.C + cc];
  const uint32_t* keys = a.keys + (size_t)b * a.n + off;
  const int32_t* pix = a.pix + (size_t)b * a.n + off;
  if (tid == 0) {
    s_prefix = 0;
    s_k = (unsigned)k;
  }
  __syncthreads();
  for (int level = 3; level >= 0; --level) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix;
    const unsigned kk = s_k;
    const unsigned mask_hi = level == 3 ? 0u : (0xFFFFFFFFu << ((level + 1) * 8));
    for (int i = tid; i < cnt; i += PL_THREADS) {
      const unsigned key = keys[i];
      if ((key & mask_hi) == prefix) atomicAdd(&hist[(key >> (level * 8)) & 255], 1);
    }
    __syncthreads();
    // inclusive suffix sums S[bin] = sum_{j >= bin} hist[j]; the k-th largest key lies in the highest bin with
    // S[bin] >= kk (bin 0 if none, as the serial walk from bin 255 down did)
    if (tid < 256) suf[0][tid] = hist[tid];
    __syncthreads();
    int cur = 0;
    for (int o = 1; o < 256; o <<= 1) {
      if (tid < 256) {
        int x = suf[cur][tid];
        if (tid + o < 256) x += suf[cur][tid + o];
        suf[cur ^ 1][tid] = x;
      }
      __syncthreads();
      cur ^= 1;
    }
    if (tid < 256) {
      const unsigned S = (unsigned)suf[cur][tid];
      const unsigned Sn = tid < 255 ? (unsigned)suf[cur][tid + 1] : 0u;
      if ((S >= kk && Sn < kk) || (tid == 0 && S < kk)) {
        s_prefix = prefix | ((unsigned)tid << (level * 8));
        s_k = kk - Sn;  // rank of the k-th element inside this bin
      }
    }
    __syncthreads();
  }
  // s_prefix = key of the k-th largest; s_k = how many elements equal to it must be taken
  const unsigned thr = s_prefix;
  if (tid == 0) {
    s_take = s_k;
    s_nties = 0;
  }
  __syncthreads();
  // Keys equal to the threshold.  With float32 keys and 10^4..10^5 members per pair two members DO share a key now and
  // then (round 4: one such tie at the threshold in a 26-step run made the run two-valued -- the bucket order, and with it
  // an atomic first-come rule, depends on the order in which workgroups filled the bucket).  The tied members are
  // collected and the ones with the SMALLEST PIXEL INDEX are taken: any choice is a valid draw of the reference's
  // multinomial (trainer.py:447-518), this one is reproducible.
  for (int i = tid; i < cnt; i += PL_THREADS) {
    const unsigned key = keys[i];
    if (key > thr) a.chosen[(size_t)b * a.n + pix[i]] = 1;
    else if (key == thr) {
      const unsigned p = atomicAdd(&s_nties, 1u);
      if (p < PL_MAX_TIES) tie_pix[p] = pix[i];
    }
  }
  __syncthreads();
  const int take = (int)s_take;
  if (s_nties > (unsigned)PL_MAX_TIES) {
    // More equal keys than the list holds (degenerate weights: saturated-entropy or zero-weight pixels with k reaching into
    // them).  Same rule, without the list: bisection on the pixel index for the smallest P with `take` tied members at
    // pixel <= P (pixel indices are distinct, so exactly `take` members qualify) -- ~log2(n) passes over the bucket, only
    // ever in this case.  (Round 4 took the first PL_MAX_TIES collected: arrival order of an atomic, and fewer than k
    // pixels when take > PL_MAX_TIES -- ADVICE round 4.)
    int lo = 0, hi = a.n - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      __syncthreads();
      if (tid == 0) s_nties = 0;
      __syncthreads();
      unsigned local = 0;
      for (int i = tid; i < cnt; i += PL_THREADS) local += (keys[i] == thr && pix[i] <= mid) ? 1u : 0u;
      if (local) atomicAdd(&s_nties, local);
      __syncthreads();
      if ((int)s_nties >= take) hi = mid;
      else lo = mid + 1;
    }
    for (int i = tid; i < cnt; i += PL_THREADS)
      if (keys[i] == thr && pix[i] <= lo) a.chosen[(size_t)b * a.n + pix[i]] = 1;
    return;
  }
  const int nt = (int)s_nties;
  for (int e = tid; e < nt; e += PL_THREADS) {
    const int me = tie_pix[e];
    int rank = 0;
    for (int j = 0; j < nt; ++j) rank += tie_pix[j] < me;
    if (rank < take) a.chosen[(size_t)b * a.n + me] = 1;
  }
}

// labels_out = chosen ? amax : 0, weak labels override; mask_out = label != ignore (trainer.py:510-516)
__global__ void pl_finalize_kernel(const int32_t* __restrict__ amax, const uint8_t* __restrict__ chosen,
                                   const int64_t* __restrict__ eval_label, const int64_t* __restrict__ train_label,
                                   size_t n, int ignore, int64_t* __restrict__ labels_out, uint8_t* __restrict__ mask_out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int64_t l = (chosen[i] && eval_label[i] > 0) ? (int64_t)amax[i] : 0;
    if (train_label[i] > 0) l = train_label[i];
    labels_out[i] = l;
    mask_out[i] = l != ignore;
  }
}

// ---------------------------------------------------------------- anchor sampling (L2)
// slot[b*C+c] = rank of the pair among present ones in (b, c) order, or -1; T = number present
__global__ __launch_bounds__(256) void pair_slots_kernel(const int32_t* __restrict__ counts, int npairs, int C, int ignore,
                                                         int32_t* __restrict__ slot, int32_t* __restrict__ T) {
  // one block; block-wide inclusive scan of the "present" flags, 256 pairs per round (a single thread walking the
  // B*C pairs took 16 us)
  __shared__ int sc[256];
  const int tid = threadIdx.x;
  int base = 0;
  for (int p0 = 0; p0 < npairs; p0 += 256) {
    const int p = p0 + tid;
    const bool present = p < npairs && (p % C) != ignore && counts[p] > 0;
    sc[tid] = present ? 1 : 0;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      int x = sc[tid];
      if (tid >= o) x += sc[tid - o];
      __syncthreads();
      sc[tid] = x;
      __syncthreads();
    }
    if (p < npairs) slot[p] = present ? base + sc[tid] - 1 : -1;
    base += sc[255];
    __syncthreads();
  }
  if (tid == 0) *T = base;
}

struct SampleArgs {
  const float* weights;    // [B][n]
  const int32_t* counts;   // [B][C]
  const int32_t* idx;      // [B][C][n] ordered pixel lists
  const int32_t* slot;     // [B][C]
  const double* uniforms;  // [Tmax][A]
  float* cum;              // [B][C][n] scratch
  int n, C, A;
  int32_t* anchor_idx;     // [Tmax][A]
  int32_t* anchor_img;     // [Tmax]
  int32_t* anchor_cls;     // [Tmax]
};

// grid (C, B)
__global__ __launch_bounds__(256) void anchor_sample_kernel(SampleArgs a) {
  __shared__ float chunk[2048];
  __shared__ float s_run;
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int pair = b * a.C + c;
  const int t = a.slot[pair];
  if (t < 0) return;
  const int nc = a.counts[pair];
  const int32_t* rows = a.idx + (size_t)pair * a.n;
  const float* w = a.weights + (size_t)b * a.n;
  float* cum = a.cum + (size_t)pair * a.n;
  if (tid == 0) s_run = 0.f;
  __syncthreads();
  // sequential fp32 running sum in list order (== the CPU kernel's order over the whole image,
  // because zero weights leave an fp32 running sum unchanged)
  for (int i0 = 0; i0 < nc; i0 += 2048) {
    const int m = min(2048, nc - i0);
    for (int j = tid; j < m; j += 256) chunk[j] = w[rows[i0 + j]];
    __syncthreads();
    if (tid == 0) {
      float run = s_run;
      for (int j = 0; j < m; ++j) {
        run = run + chunk[j];
        chunk[j] = run;
      }
      s_run = run;
    }
    __syncthreads();
    for (int j = tid; j < m; j += 256) cum[i0 + j] = chunk[j];
    __syncthreads();
  }
  const float total = s_run;
  for (int j = tid; j < nc; j += 256) cum[j] = cum[j] / total;
  __threadfence_block();
  __syncthreads();
  for (int s = tid; s < a.A; s += 256) {
    const double u = a.uniforms[(size_t)t * a.A + s];
    int lo = 0, hi = nc;
    while (hi - lo > 0) {
      const int mid = lo + (hi - lo) / 2;
      if ((double)cum[mid] < u) lo = mid + 1;
      else hi = mid;
    }
    if (lo > nc - 1) lo = nc - 1;
    int pix = rows[lo];
    if (u <= 0.0) pix = 0;
    a.anchor_idx[(size_t)t * a.A + s] = pix;
  }
  if (tid == 0) {
    a.anchor_img[t] = b;
    a.anchor_cls[t] = c;
  }
}

// ---------------------------------------------------------------- anchor gather / scatter
// out[t*A + s][:] = l2_normalize(feat[img[t]][idx[t][s]][:]); rows with t >= T are zero
__global__ __launch_bounds__(256) void gather_l2_kernel(const float* __restrict__ feat, const int32_t* __restrict__ img,
                                                        const int32_t* __restrict__ idx, const int32_t* __restrict__ T,
                                                        int Tmax, int A, int n, int D, float eps,
                                                        float* __restrict__ out, float* __restrict__ norm) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  const int Tn = *T;
  for (size_t r = wave; r < (size_t)Tmax * A; r += nw) {
    const int t = r / A;
    if (t >= Tn) {
      for (int d = lane * 4; d < D; d += 256) *reinterpret_cast<f32x4*>(out + r * D + d) = f32x4{0.f, 0.f, 0.f, 0.f};
      if (lane == 0) norm[r] = 1.f;
      continue;
    }
    const float* src = feat + ((size_t)img[t] * n + idx[r]) * D;
    float s = 0.f;
    for (int d = lane * 4; d < D; d += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + d);
      s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    const float nr = sqrtf(c3d_wave_sum(s));
    const float inv = 1.f / fmaxf(nr, eps);
    for (int d = lane * 4; d < D; d += 256)
      *reinterpret_cast<f32x4*>(out + r * D + d) = *reinterpret_cast<const f32x4*>(src + d) * inv;
    if (lane == 0) norm[r] = nr;
  }
}

// dfeat[img[t]][idx[t][s]][:] += g * sum over the rows s' of pair t with idx[t][s'] == idx[t][s] of dx[t*A+s'][:].
// Anchors repeat (sampling with replacement), but only INSIDE a pair t: a pixel carries one label, so the pixel sets
// of different (image, class) pairs are disjoint.  One wave per row; the wave of the FIRST occurrence of a pixel in its
// pair sums all of that pixel's rows in ascending s and does one plain read-modify-write -- no atomics, and the
// result does not depend on the order in which waves run (bit-reproducible d feat).
// Work item = (row, chunk of 64 * VW channels) per wave (VW = 4 when D % 4 == 0: the ownership test, the expensive part,
// runs once per 256 channels; with VW = 1 -- 64 channels per item -- the compact scatter took 139 us at the headline shape).  With weak labels a pair has a handful of labelled pixels and hundreds
// of anchors, so an owner sums ~A/pixels rows: their loads are issued eight at a time (independent), the adds stay in
// row order.  (One wave per whole row with one load in flight per duplicate: 425 us at the headline shape.)
template <int VW>
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ dx, const int32_t* __restrict__ img,
                                                           const int32_t* __restrict__ idx, const int32_t* __restrict__ T,
                                                           int Tmax, int A, int n, int D, const float* __restrict__ gscale,
                                                           float* __restrict__ dfeat, uint32_t* __restrict__ rowmask,
                                                           int32_t* __restrict__ cmap) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  const int Tn = *T;
  const float g = gscale ? *gscale : 1.f;
  typedef float vecw __attribute__((ext_vector_type(VW)));
  const int DC = (D + 64 * VW - 1) / (64 * VW);
  for (size_t wi = wave; wi < (size_t)Tn * A * DC; wi += nw) {
    const size_t r = wi / DC;
    const int d = ((int)(wi - r * DC) * 64 + lane) * VW;
    const int t = r / A, s = (int)(r - (size_t)t * A);
    const int32_t* grp = idx + (size_t)t * A;
    const int mine = grp[s];
    const float* src = dx + (size_t)t * A * D + d;
    vecw acc = 0.f;
    // sum the rows q[0..8) (ascending, -1 = none): loads first, adds in row order
    auto add_rows = [&](unsigned long long& m, int base) {
      int q[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        q[k] = m ? base + __builtin_ctzll(m) : -1;
        m &= m - 1;          // 0 stays 0
      }
      vecw x[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = (q[k] >= 0 && d < D) ? *reinterpret_cast<const vecw*>(src + (size_t)q[k] * D) : vecw(0.f);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += x[k];      // absent rows add 0
    };
    if (A <= 512) {
      // the pair's whole index row in one round of loads: ownership test and duplicate masks come from registers
      // (chunk-by-chunk scans, a dependent load each, cost 243 us once pseudo-labels make most anchors distinct)
      int v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int j = k * 64 + lane;
        v[k] = j < A ? grp[j] : -1;
      }
      bool hit = false;
#pragma unroll
      for (int k = 0; k < 8; ++k) hit |= (k * 64 + lane < s) && v[k] == mine;
      if (__ballot(hit)) continue;          // an earlier row of the pair with the same pixel owns the sum
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        unsigned long long m = __ballot(k * 64 + lane >= s && v[k] == mine);
        while (m) add_rows(m, k * 64);
      }
    } else {
      bool owner = true;
      for (int base = 0; base < s && owner; base += 512) {
        bool hit = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int j = base + k * 64 + lane;
          hit |= (j < s ? grp[j] : -1) == mine;
        }
        if (__ballot(hit)) owner = false;
      }
      if (!owner) continue;
      for (int base = s & ~63; base < A; base += 64) {
        const int j = base + lane;
        unsigned long long m = __ballot(j >= s && j < A && grp[j] == mine);
        while (m) add_rows(m, base);
      }
    }
    if (d < D) {
      if (cmap) {
        // compact form: the owner's own row slot holds the pixel's sum (what a zero-filled dense gradient would hold
        // there: 0 + g * acc), cmap[pixel] says which slot; pixels without a set rowmask bit have no cmap entry
        *reinterpret_cast<vecw*>(dfeat + r * D + d) = vecw(0.f) + g * acc;
      } else {
        vecw* dst = reinterpret_cast<vecw*>(dfeat + ((size_t)img[t] * n + mine) * D + d);
        *dst += g * acc;
      }
    }
    if (d == 0 && (rowmask || cmap)) {       // one lane per written row (an OR: order-free)
      const size_t pix = (size_t)img[t] * n + mine;
      if (cmap) cmap[pix] = (int32_t)r;
      if (rowmask) atomicOr(&rowmask[pix >> 5], 1u << (pix & 31));
    }
  }
}

// ---------------------------------------------------------------- InfoNCE rows (L3)
// logits [R][ld] = a_i . q_j (cosine); columns j < (C-1)*M valid, column class = 1 + j / M.
// Writes d(loss)/d(logits) in place and per-row losses.  One wave per row.
__global__ __launch_bounds__(256) void infonce_rows_kernel(float* __restrict__ logits, int ld,
                                                           const int32_t* __restrict__ row_cls,
                                                           const int32_t* __restrict__ T, int Tmax, int A, int M,
                                                           int ncols, float temp, float base_temp,
                                                           float* __restrict__ row_loss) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  const int Tn = *T;
  for (size_t r = wave; r < (size_t)Tmax * A; r += nw) {
    float* row = logits + r * ld;
    const int t = r / A;
    if (t >= Tn) {
      for (int j = lane; j < ld; j += 64) row[j] = 0.f;
      if (lane == 0) row_loss[r] = 0.f;
      continue;
    }
    const int cls = row_cls[t];
    float x[8];
    float mx = -INFINITY;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int j = lane + q * 64;
      x[q] = j < ncols ? row[j] / temp : -INFINITY;
      mx = fmaxf(mx, x[q]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float neg = 0.f;
    float ex[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int j = lane + q * 64;
      x[q] -= mx;
      ex[q] = j < ncols ? expf(x[q]) : 0.f;
      const bool pos = j < ncols && (1 + j / M) == cls;
      if (!pos) neg += ex[q];
    }
    neg = c3d_wave_sum(neg);
    float sum_logp = 0.f, sum_inv = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int j = lane + q * 64;
      const bool pos = j < ncols && (1 + j / M) == cls;
      if (pos) {
        const float den = ex[q] + neg + 1e-6f;
        sum_logp += x[q] - logf(den);
        sum_inv += 1.f / den;
      }
    }
    sum_logp = c3d_wave_sum(sum_logp);
    sum_inv = c3d_wave_sum(sum_inv);
    const float coef = -(temp / base_temp) / (float)M;   // loss_row = coef * sum_pos logp
    const float scale = coef / temp / (float)(Tn * A);    // d(mean loss)/d(logit) = scale * g
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int j = lane + q * 64;
      if (j < ld) {
        float g = 0.f;
        if (j < ncols) {
          const bool pos = (1 + j / M) == cls;
          g = pos ? 1.f - ex[q] / (ex[q] + neg + 1e-6f) : -ex[q] * sum_inv;
        }
        row[j] = g * scale;
      }
    }
    if (lane == 0) row_loss[r] = coef * sum_logp;
  }
}

// loss = sum(row_loss) / (T*A)   (single block of 1024 threads, fp64 fold; four independent loads per trip -- a
// 256-thread dependent walk over the 77 824 rows took 70 us)
__global__ __launch_bounds__(1024) void infonce_reduce_kernel(const float* __restrict__ row_loss,
                                                              const int32_t* __restrict__ T, int A,
                                                              float* __restrict__ loss) {
  __shared__ double red[16];
  const int Tn = *T;
  const int n = Tn * A;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int i = threadIdx.x;
  for (; i + 3 * 1024 < n; i += 4 * 1024) {
    const float a = row_loss[i], b = row_loss[i + 1024], c = row_loss[i + 2048], d = row_loss[i + 3072];
    s0 += (double)a;
    s1 += (double)b;
    s2 += (double)c;
    s3 += (double)d;
  }
  for (; i < n; i += 1024) s0 += (double)row_loss[i];
  double s = c3d_wave_sum_d((s0 + s1) + (s2 + s3));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < 16; ++k) t += red[k];
    *loss = Tn > 0 ? (float)(t / ((double)Tn * A)) : 0.f;
  }
}

}  // namespace

#define ST ((hipStream_t)stream)
static inline int nb_for(size_t n, int per) {
  size_t b = (n + per - 1) / per;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

extern "C" int c3d_entropy_stats(const float* prob, int64_t n, int C, float* w_anchor, float* w_pl, int32_t* amax,
                                 c3d_stream stream) {
  C3D_REQUIRE(C <= 32, "entropy_stats: at most 32 classes");
  if (C % 4 == 0)
    hipLaunchKernelGGL(entropy_stats_kernel<true>, dim3(nb_for((size_t)n, 256)), dim3(256), 0, ST, prob, (size_t)n, C,
                       w_anchor, w_pl, amax);
  else
    hipLaunchKernelGGL(entropy_stats_kernel<false>, dim3(nb_for((size_t)n, 256)), dim3(256), 0, ST, prob, (size_t)n, C,
                       w_anchor, w_pl, amax);
  C3D_CHECK_LAUNCH();
  return 0;
}

static int pl_select_impl(const float* w_pl, const int32_t* amax, const int64_t* eval_label,
                             const int64_t* train_label, const float* noise, const int32_t* tl_counts, int B, int n,
                             int C, int ignore_label, float ratio, const float* ratio_dev, int32_t* scratch, uint8_t* chosen,
                             int64_t* labels_out, uint8_t* mask_out, c3d_stream stream) {
  C3D_REQUIRE(C <= 64, "pl_select: at most 64 classes");
  // scratch: cnt [B*C] | cursor [B*C] | keys [B*n] | pix [B*n]
  int32_t* cnt = scratch;
  int32_t* cursor = scratch + (size_t)B * C;
  uint32_t* keys = reinterpret_cast<uint32_t*>(scratch + (size_t)2 * B * C);
  int32_t* pix = scratch + (size_t)2 * B * C + (size_t)B * n;
  (void)hipMemsetAsync(scratch, 0, sizeof(int32_t) * 2 * B * C, ST);
  PlArgs a{w_pl, amax, eval_label, noise, tl_counts, n, C, ignore_label, ratio, ratio_dev, cnt, cursor, keys, pix, chosen};
  int chunks = (n + 4095) / 4096;
  if (chunks < 1) chunks = 1;
  hipLaunchKernelGGL(pl_bucket_kernel<false>, dim3(chunks, B), dim3(256), 0, ST, a);
  C3D_CHECK_LAUNCH();
  hipLaunchKernelGGL(pl_bucket_kernel<true>, dim3(chunks, B), dim3(256), 0, ST, a);
  C3D_CHECK_LAUNCH();
  hipLaunchKernelGGL(pl_select_kernel, dim3(C, B), dim3(PL_THREADS), 0, ST, a);
  C3D_CHECK_LAUNCH();
  hipLaunchKernelGGL(pl_finalize_kernel, dim3(nb_for((size_t)B * n, 256)), dim3(256), 0, ST, amax, chosen, eval_label,
                     train_label, (size_t)B * n, ignore_label, labels_out, mask_out);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_pl_select(const float* w_pl, const int32_t* amax, const int64_t* eval_label,
                             const int64_t* train_label, const float* noise, const int32_t* tl_counts, int B, int n,
                             int C, int ignore_label, float ratio, int32_t* scratch, uint8_t* chosen,
                             int64_t* labels_out, uint8_t* mask_out, c3d_stream stream) {
  return pl_select_impl(w_pl, amax, eval_label, train_label, noise, tl_counts, B, n, C, ignore_label, ratio, nullptr,
                        scratch, chosen, labels_out, mask_out, stream);
}

extern "C" int c3d_pl_select_dev(const float* w_pl, const int32_t* amax, const int64_t* eval_label,
                                 const int64_t* train_label, const float* noise, const int32_t* tl_counts, int B, int n,
                                 int C, int ignore_label, const float* ratio_dev, int32_t* scratch, uint8_t* chosen,
                                 int64_t* labels_out, uint8_t* mask_out, c3d_stream stream) {
  C3D_REQUIRE(ratio_dev != nullptr, "pl_select_dev: the ratio must be a device scalar");
  return pl_select_impl(w_pl, amax, eval_label, train_label, noise, tl_counts, B, n, C, ignore_label, 0.f, ratio_dev,
                        scratch, chosen, labels_out, mask_out, stream);
}

extern "C" int c3d_anchor_sample(const float* weights, const int32_t* counts, const int32_t* idx,
                                 const double* uniforms, int B, int n, int C, int A, int ignore_label, int32_t* slot,
                                 float* cum, int32_t* anchor_idx, int32_t* anchor_img, int32_t* anchor_cls, int32_t* T,
                                 c3d_stream stream) {
  hipLaunchKernelGGL(pair_slots_kernel, dim3(1), dim3(256), 0, ST, counts, B * C, C, ignore_label, slot, T);
  C3D_CHECK_LAUNCH();
  SampleArgs a{weights, counts, idx, slot, uniforms, cum, n, C, A, anchor_idx, anchor_img, anchor_cls};
  hipLaunchKernelGGL(anchor_sample_kernel, dim3(C, B), dim3(256), 0, ST, a);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_gather_rows_l2(const float* feat, const int32_t* img, const int32_t* idx, const int32_t* T, int Tmax,
                                  int A, int n, int D, float eps, float* out, float* norm, c3d_stream stream) {
  C3D_REQUIRE(D % 4 == 0, "gather: D must be a multiple of 4");
  hipLaunchKernelGGL(gather_l2_kernel, dim3(nb_for((size_t)Tmax * A, 4)), dim3(256), 0, ST, feat, img, idx, T, Tmax, A, n,
                     D, eps, out, norm);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_scatter_add_rows(const float* dx, const int32_t* img, const int32_t* idx, const int32_t* T, int Tmax,
                                    int A, int n, int D, const float* gscale, float* dfeat, uint32_t* rowmask, c3d_stream stream) {
  if (D % 4 == 0)
    hipLaunchKernelGGL(scatter_rows_kernel<4>, dim3(nb_for((size_t)Tmax * A, 4)), dim3(256), 0, ST, dx, img, idx, T, Tmax, A,
                       n, D, gscale, dfeat, rowmask, (int32_t*)nullptr);
  else
    hipLaunchKernelGGL(scatter_rows_kernel<1>, dim3(nb_for((size_t)Tmax * A, 4)), dim3(256), 0, ST, dx, img, idx, T, Tmax, A,
                       n, D, gscale, dfeat, rowmask, (int32_t*)nullptr);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_scatter_rows_compact(const float* dx, const int32_t* img, const int32_t* idx, const int32_t* T, int Tmax,
                                        int A, int n, int D, const float* gscale, float* drows, int32_t* cmap,
                                        uint32_t* rowmask, c3d_stream stream) {
  C3D_REQUIRE(drows && cmap && rowmask, "scatter_rows_compact: drows, cmap and rowmask are required");
  if (D % 4 == 0)
    hipLaunchKernelGGL(scatter_rows_kernel<4>, dim3(nb_for((size_t)Tmax * A, 4)), dim3(256), 0, ST, dx, img, idx, T, Tmax, A,
                       n, D, gscale, drows, rowmask, cmap);
  else
    hipLaunchKernelGGL(scatter_rows_kernel<1>, dim3(nb_for((size_t)Tmax * A, 4)), dim3(256), 0, ST, dx, img, idx, T, Tmax, A,
                       n, D, gscale, drows, rowmask, cmap);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_infonce_rows(float* logits, int ld, const int32_t* row_cls, const int32_t* T, int Tmax, int A, int M,
                                int ncols, float temperature, float base_temperature, float* row_loss, float* loss,
                                c3d_stream stream) {
  C3D_REQUIRE(ld <= 512, "infonce: at most 512 queue columns");
  hipLaunchKernelGGL(infonce_rows_kernel, dim3(nb_for((size_t)Tmax * A, 4)), dim3(256), 0, ST, logits, ld, row_cls, T,
                     Tmax, A, M, ncols, temperature, base_temperature, row_loss);
  C3D_CHECK_LAUNCH();
  hipLaunchKernelGGL(infonce_reduce_kernel, dim3(1), dim3(1024), 0, ST, row_loss, T, A, loss);
  C3D_CHECK_LAUNCH();
  return 0;
}
