// SyncBatchNorm statistics exchange between the ranks of ONE node through IPC-mapped peer memory.
//
// Replaces the dist.all_reduce of torch.nn.SyncBatchNorm's forward / backward (the reference wraps its model with
// torch.nn.SyncBatchNorm.convert_sync_batchnorm, tasks/weak_segmentation/trainer.py:54) for the 43 + 43 tiny fp64 vectors
// of a training step (<= 704 x 2 doubles each).  Through torch.distributed every one of them is a RCCL launch of its own
// (~17 us of launch cost in a 1-rank group before any xGMI latency; 84 blocking exchanges per step).  Here one small
// kernel per exchange: every rank WRITES its vector into its slot of every peer's mailbox (device memory of the peer,
// mapped here with hipIpcOpenMemHandle -- xGMI stores), publishes a sequence number behind it, waits for the sequence
// numbers of all peers in its own mailbox and sums the slots in rank order -- the same order on every rank, so the result
// is bit-identical everywhere and run to run.  One process per GPU; gradients and the prototype bank stay on RCCL.
//
// Protocol (per rank: mailbox = header | flags[2][W] | slots[2][W][CAP] doubles; W = C3D_PEER_MAX_RANKS):
//   seq = ++calls (a counter in the rank's OWN mailbox header: device memory, so a captured hipGraph keeps counting; every
//   rank makes the same sequence of calls, as with any collective); parity = seq & 1.
//   1. for every rank p (itself included): peer[p].slots[parity][me][0 .. n) = buf           (system-scope stores)
//   2. system-scope release fence, then peer[p].flags[parity][me] = seq                       (one lane per peer)
//   3. wait until own.flags[parity][q] == seq for every rank q (relaxed system-scope polls with s_sleep, BOUNDED: after
//      ~C3D_PEER_TIMEOUT_S seconds the status word is set and the kernel gives up -- a missing peer must not hang the GPU)
//   4. system-scope acquire fence, buf[i] = sum_q own.slots[parity][q][i] in rank order
// Two parities suffice: a rank can only reach call seq + 2 after it has seen every peer's seq + 1, which the peer
// publishes from a kernel that runs after its call-seq kernel has finished reading (stream order).
// The mailbox is fine-grained device memory (hipDeviceMallocFinegrained: what RCCL itself uses for its flags), zeroed once.
#include <string.h>
#include "common.h"
#include "../../include/coarse3d_hip.h"

namespace {

constexpr int PEER_HEADER_BYTES = 256;
constexpr int PEER_W = C3D_PEER_MAX_RANKS;

struct PeerArgs {
  unsigned char* box[PEER_W];     // mailbox of rank p as mapped in this process (box[rank] = own)
  int rank, world, cap;
  int fences;                     // 0: all ranks on this device -- write-through payload + drained flag suffice (guide G16, R1)
  double* buf;
  int n;
  unsigned long long timeout_ticks;     // of the 100 MHz wall clock
};

__device__ __forceinline__ unsigned long long* peer_flags(unsigned char* box, int parity, int r) {
  return reinterpret_cast<unsigned long long*>(box + PEER_HEADER_BYTES) + parity * PEER_W + r;
}
__device__ __forceinline__ double* peer_slot(unsigned char* box, int cap, int parity, int r) {
  return reinterpret_cast<double*>(box + PEER_HEADER_BYTES + 2 * PEER_W * sizeof(unsigned long long)) + ((size_t)parity * PEER_W + r) * cap;
}

// The exchange, by one 256-thread block: a.buf[0 .. n) becomes the rank-ordered sum over ranks.  COHERENT: a.buf was written by
// other blocks of this launch (the fused kernels below) -- read it past this CU's L1.  Returns false if a peer never arrived
// (status word set, a.buf untouched from the protocol's side).
template <bool COHERENT>
__device__ __forceinline__ bool peer_exchange_block(const PeerArgs& a) {
  __shared__ unsigned long long s_seq;
  __shared__ int s_fail;
  const int tid = threadIdx.x;
  unsigned char* own = a.box[a.rank];
  unsigned long long* calls = reinterpret_cast<unsigned long long*>(own);
  unsigned int* status = reinterpret_cast<unsigned int*>(own + 8);
  if (tid == 0) {
    s_seq = *calls + 1ull;
    s_fail = (int)*status;
  }
  __syncthreads();
  // an earlier exchange gave up: the ranks' sequence numbers no longer agree and every further wait would run into its
  // timeout too -- do nothing (the host raises at its next check; a step with 86 exchanges costs ONE timeout, not 86)
  if (s_fail) return false;
  const unsigned long long seq = s_seq;
  const int parity = (int)(seq & 1ull);
  // 1. my vector into my slot of every mailbox
  for (int p = 0; p < a.world; ++p) {
    double* dst = peer_slot(a.box[p], a.cap, parity, a.rank);
    for (int i = tid; i < a.n; i += 256) {
      const double v = COHERENT ? __hip_atomic_load(a.buf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : a.buf[i];
      __hip_atomic_store(dst + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  // 2. publish: the stores above must be visible system-wide before the sequence number is.  They are write-through
  //    (system-scope sc0 sc1 stores) and every storing wave drains them; across DEVICES a system-scope release fence is
  //    issued on top (the cross-GPU path has not been run on hardware yet: the conservative form).  On one device the
  //    fence is what an exchange costs (buffer_wbl2 + buffer_inv: 11.5 us per exchange in a 1-rank group, 4 without).
  if (a.fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid < a.world) __hip_atomic_store(peer_flags(a.box[tid], parity, a.rank), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // 3. wait for everybody's vector (bounded)
  if (tid < a.world) {
    const unsigned long long* f = peer_flags(own, parity, tid);
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - t0 > a.timeout_ticks) {
        s_fail = 1;
        break;
      }
    }
  }
  __syncthreads();
  if (a.fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");       // (the slot reads below bypass L1 / L2 themselves)
  if (s_fail) {          // a peer never arrived: say so (the host raises at its next check) and leave buf as it is
    if (tid == 0) {
      *status = 1u;
      *calls = seq;
    }
    return false;
  }
  // 4. the sum, in rank order
  for (int i = tid; i < a.n; i += 256) {
    double s = 0.0;
    for (int q = 0; q < a.world; ++q) s += __hip_atomic_load(peer_slot(own, a.cap, parity, q) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    a.buf[i] = s;
  }
  if (tid == 0) *calls = seq;
  __syncthreads();       // (the fused kernels read a.buf across threads next)
  return true;
}

// A failed exchange must not pass for a sum: the collective it replaces would block or raise, a kernel can
// only poison.  NaN in every value the caller reads from this exchange; the host learns of it from the status word
// (c3d_peer_status / c3d_peer_status_to).
__device__ __forceinline__ double peer_nan() { return __longlong_as_double(0x7ff8000000000000ll); }

__global__ __launch_bounds__(256) void peer_allreduce_kernel(PeerArgs a) {
  if (peer_exchange_block<false>(a)) return;
  for (int i = threadIdx.x; i < a.n; i += 256) a.buf[i] = peer_nan();
}

__global__ void peer_status_kernel(const unsigned char* own, float* dst) {
  *dst = __hip_atomic_load(reinterpret_cast<const unsigned int*>(own + 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) ? 1.f : 0.f;
}

// block-wide fold of one channel's partials [2][n] -> (s1, s2) in fp64 (as fold_channel in bn_ops.hip: the same order, the same bits)
__device__ __forceinline__ void peer_fold_channel(const float* __restrict__ p, int n, double& s1, double& s2) {
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

// One block per channel folds; the block that draws the last ticket of the launch is alone from then on.
//
// READ THIS BEFORE "FIXING" OR TRUSTING THE HAND-OVER BELOW.  Two different hand-overs live in these fused kernels:
//   (1) block -> last block, INSIDE this device (this function + peer_publish): fence-free, and independent of
//       c3d_peer_desc.one_device -- the ranks' placement has nothing to do with it, it never leaves the GPU;
//   (2) rank -> rank, the exchange proper (peer_exchange_block): it keeps its system-scope release / acquire fences whenever
//       the ranks sit on different devices (PeerArgs::fences), and only that part crosses xGMI.
// Why (1) needs no fence although the blocks of one launch run on different XCDs (each with its own, non-coherent L2):
//   * the producer side: every sum is published with a system-scope relaxed atomic store (sc0 sc1: write-through, it does not
//     stay in the storing XCD's L2), by thread 0, which then waits for its own stores to have left (s_waitcnt vmcnt(0)) BEFORE
//     it draws its ticket -- so when ticket t is visible, the sums of the block that drew it are in memory;
//   * the ticket itself is an agent-scope atomic read-modify-write, performed at the device-coherent level (not in an XCD's L2),
//     so the block that reads gridDim.x - 1 has been ordered behind every other block's draw;
//   * the consumer side: the last block reads the sums with system-scope atomic loads (peer_exchange_block<COHERENT = true>,
//     f.local in the backward kernel), which bypass its CU's L1 and its XCD's L2 -- it cannot see a stale cached line.
// Formally this is still a data race in the HIP memory model (relaxed operations, no release / acquire pair): it relies on
// the write-through / cache-bypassing behaviour of system-scope accesses to fine-grained memory on gfx950.  Measured: bit-identical
// to the three-launch path over 3 M exchanges (tools/peer_soak.py) and in the judged suite (tests/test_gpu_dp.py), on ONE
// device with one and two processes.  The fenced alternative is correct by the model and was measured too: an agent-scope
// release / acquire is a write-back / invalidate of an XCD's whole L2, 23 us a launch -- more than the three launches these
// kernels replace -- so the choice on a failure of (1) is C3D_PEER_FUSED_BN=0 (three launches), not a fence here.
__device__ __forceinline__ void peer_publish(double* dst, double v) { __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ bool peer_last_block(unsigned int* ticket) {
  __shared__ int s_last;
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this block's sums (thread 0 wrote them) before its ticket
    const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = t == gridDim.x - 1;
    if (s_last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (everybody has drawn)
  }
  __syncthreads();
  return s_last != 0;
}

struct PeerBnFwd {
  const float* partial; int npart; int C; double count;
  const float *gamma, *beta; float *rm, *rv; float momentum, eps;
  float *scale, *shift, *mean, *invstd;
  unsigned int* ticket;
};

// SyncBatchNorm forward statistics in ONE launch (data parallel): fold the conv epilogue's partials, exchange the fp64 sums
// through the mailboxes, finish (scale, shift, mean, invstd, running statistics) -- stat_reduce + the exchange kernel +
// bn_finalize of the three-launch path, the same arithmetic in the same order.  a.buf: scratch [C][2] doubles.
__global__ __launch_bounds__(256) void peer_bn_forward_kernel(PeerArgs a, PeerBnFwd f) {
  const int c = blockIdx.x;
  double s1, s2;
  peer_fold_channel(f.partial + (size_t)c * 2 * f.npart, f.npart, s1, s2);
  if (threadIdx.x == 0) {
    peer_publish(a.buf + c * 2 + 0, s1);
    peer_publish(a.buf + c * 2 + 1, s2);
  }
  if (!peer_last_block(f.ticket)) return;
  if (!peer_exchange_block<true>(a)) {
    const float nanf_ = __uint_as_float(0x7fc00000u);
    for (int ch = threadIdx.x; ch < f.C; ch += 256) f.scale[ch] = f.shift[ch] = f.mean[ch] = f.invstd[ch] = nanf_;
    return;
  }
  for (int ch = threadIdx.x; ch < f.C; ch += 256) {        // bn_finalize_kernel (bn_ops.hip)
    const double mean = a.buf[ch * 2] / f.count;
    double var = a.buf[ch * 2 + 1] / f.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
    const float sc = f.gamma[ch] * invstd;
    f.scale[ch] = sc;
    f.shift[ch] = f.beta[ch] - (float)mean * sc;
    f.mean[ch] = (float)mean;
    f.invstd[ch] = invstd;
    if (f.rm) {
      const double unbiased = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
      f.rm[ch] = (1.f - f.momentum) * f.rm[ch] + f.momentum * (float)mean;
      f.rv[ch] = (1.f - f.momentum) * f.rv[ch] + f.momentum * (float)unbiased;
    }
  }
}

struct PeerBnBwd {
  const float* partial; int npart; int C; double count;
  const float *mean, *invstd, *gamma;
  float *k1, *k2, *k3, *dgamma, *dbeta;
  double* local;          // scratch [C][2]: this rank's sums (the parameter gradients stay rank-local, as torch.nn.SyncBatchNorm's)
  unsigned int* ticket;
};

// ... and the backward sums (sum dy, sum dy * a): stat_reduce2 + the exchange kernel + bn_bwd_coeffs in one launch
__global__ __launch_bounds__(256) void peer_bn_backward_kernel(PeerArgs a, PeerBnBwd f) {
  const int c = blockIdx.x;
  double s1, s2;
  peer_fold_channel(f.partial + (size_t)c * 2 * f.npart, f.npart, s1, s2);
  if (threadIdx.x == 0) {
    peer_publish(a.buf + c * 2 + 0, s1);
    peer_publish(a.buf + c * 2 + 1, s2);
    peer_publish(f.local + c * 2 + 0, s1);
    peer_publish(f.local + c * 2 + 1, s2);
  }
  if (!peer_last_block(f.ticket)) return;
  if (!peer_exchange_block<true>(a)) {
    const float nanf_ = __uint_as_float(0x7fc00000u);
    for (int ch = threadIdx.x; ch < f.C; ch += 256) f.k1[ch] = f.k2[ch] = f.k3[ch] = f.dgamma[ch] = f.dbeta[ch] = nanf_;
    return;
  }
  for (int ch = threadIdx.x; ch < f.C; ch += 256) {        // bn_bwd_coeffs_kernel (bn_ops.hip)
    const double sdy = a.buf[ch * 2], sdya = a.buf[ch * 2 + 1];
    const double mu = f.mean[ch], is = f.invstd[ch], g = f.gamma[ch];
    const double sdyx = is * (sdya - mu * sdy);
    const double kk1 = g * is;
    const double kk2 = -g * is * is * sdyx / f.count;
    const double kk3 = -g * is * sdy / f.count - kk2 * mu;
    f.k1[ch] = (float)kk1;
    f.k2[ch] = (float)kk2;
    f.k3[ch] = (float)kk3;
    const double ldy = __hip_atomic_load(f.local + ch * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const double ldya = __hip_atomic_load(f.local + ch * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    f.dgamma[ch] = (float)(is * (ldya - mu * ldy));
    f.dbeta[ch] = (float)ldy;
  }
}

}  // namespace

extern "C" int c3d_peer_desc_bytes(void) { return (int)sizeof(c3d_peer_desc); }

extern "C" int64_t c3d_peer_mailbox_bytes(int cap_doubles) {
  if (cap_doubles <= 0) return 0;
  return (int64_t)PEER_HEADER_BYTES + 2 * PEER_W * (int64_t)sizeof(unsigned long long) + 2ll * PEER_W * cap_doubles * (int64_t)sizeof(double);
}

extern "C" int c3d_peer_alloc(int64_t bytes, void** ptr, void* handle64) {
  C3D_REQUIRE(ptr && handle64 && bytes > 0, "peer_alloc: null pointer or empty mailbox");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C ABI carries IPC handles as 64 bytes");
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    c3d_set_error(hipGetErrorString(e));
    return 1;
  }
  e = hipMemset(p, 0, (size_t)bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  hipIpcMemHandle_t h;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) {
    c3d_set_error(hipGetErrorString(e));
    (void)hipFree(p);
    return 1;
  }
  memcpy(handle64, &h, 64);
  *ptr = p;
  return 0;
}

extern "C" int c3d_peer_open(const void* handle64, void** ptr) {
  C3D_REQUIRE(ptr && handle64, "peer_open: null pointer");
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  void* p = nullptr;
  const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) {
    c3d_set_error(hipGetErrorString(e));
    return 1;
  }
  *ptr = p;
  return 0;
}

extern "C" int c3d_peer_close(void* ptr) {
  if (ptr && hipIpcCloseMemHandle(ptr) != hipSuccess) {
    c3d_set_error("peer_close: hipIpcCloseMemHandle failed");
    return 1;
  }
  return 0;
}

extern "C" int c3d_peer_free(void* ptr) {
  if (ptr && hipFree(ptr) != hipSuccess) {
    c3d_set_error("peer_free: hipFree failed");
    return 1;
  }
  return 0;
}

static int peer_args(const c3d_peer_desc* d, double* buf, int n, PeerArgs& a) {
  C3D_REQUIRE(d != nullptr && buf != nullptr, "peer exchange: null pointer");
  C3D_REQUIRE(d->world >= 1 && d->world <= PEER_W && d->rank >= 0 && d->rank < d->world, "peer exchange: bad rank / world");
  C3D_REQUIRE(n >= 0 && n <= d->cap_doubles, "peer exchange: the vector does not fit the mailbox slot");
  for (int p = 0; p < PEER_W; ++p) a.box[p] = p < d->world ? static_cast<unsigned char*>(d->mailbox[p]) : nullptr;
  for (int p = 0; p < d->world; ++p) C3D_REQUIRE(a.box[p] != nullptr, "peer exchange: a peer mailbox is not mapped");
  a.rank = d->rank; a.world = d->world; a.cap = d->cap_doubles; a.buf = buf; a.n = n;
  a.fences = (d->one_device || d->world == 1) ? 0 : 1;
  const double secs = d->timeout_s > 0.f ? d->timeout_s : 600.f;
  a.timeout_ticks = (unsigned long long)(secs * 100e6);
  return 0;
}

extern "C" int c3d_peer_allreduce_f64(const c3d_peer_desc* d, double* buf, int n, c3d_stream stream) {
  PeerArgs a;
  if (peer_args(d, buf, n, a)) return 1;
  if (n == 0) return 0;
  hipLaunchKernelGGL(peer_allreduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_peer_bn_finalize_partials(const c3d_peer_desc* d, const float* partial, int n, double count, const float* gamma,
                                             const float* beta, float* running_mean, float* running_var, float momentum, float eps, int C,
                                             float* scale, float* shift, float* save_mean, float* save_invstd, double* scratch,
                                             uint32_t* ticket, c3d_stream stream) {
  C3D_REQUIRE(partial && gamma && beta && scale && shift && save_mean && save_invstd && scratch && ticket && C > 0 && n > 0,
              "peer_bn_finalize_partials: null pointer or empty problem");
  PeerArgs a;
  if (peer_args(d, scratch, 2 * C, a)) return 1;
  PeerBnFwd f{partial, n, C, count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, save_mean, save_invstd, ticket};
  hipLaunchKernelGGL(peer_bn_forward_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, a, f);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_peer_bn_bwd_coeffs_partials(const c3d_peer_desc* d, const float* partial, int n, double count, const float* mean,
                                               const float* invstd, const float* gamma, int C, float* k1, float* k2, float* k3,
                                               float* dgamma, float* dbeta, double* scratch, uint32_t* ticket, c3d_stream stream) {
  C3D_REQUIRE(partial && mean && invstd && gamma && k1 && k2 && k3 && dgamma && dbeta && scratch && ticket && C > 0 && n > 0,
              "peer_bn_bwd_coeffs_partials: null pointer or empty problem");
  PeerArgs a;
  if (peer_args(d, scratch, 2 * C, a)) return 1;
  PeerBnBwd f{partial, n, C, count, mean, invstd, gamma, k1, k2, k3, dgamma, dbeta, scratch + 2 * C, ticket};
  hipLaunchKernelGGL(peer_bn_backward_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, a, f);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_peer_status_to(const c3d_peer_desc* d, float* dst, c3d_stream stream) {
  C3D_REQUIRE(d != nullptr && dst != nullptr, "peer_status_to: null pointer");
  C3D_REQUIRE(d->rank >= 0 && d->rank < PEER_W && d->mailbox[d->rank] != nullptr, "peer_status_to: the own mailbox is not allocated");
  hipLaunchKernelGGL(peer_status_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream,
                     static_cast<const unsigned char*>(d->mailbox[d->rank]), dst);
  C3D_CHECK_LAUNCH();
  return 0;
}

// status word of the rank's own mailbox (host read: synchronises the calling thread with the device).  0 = fine;
// 1 = an exchange gave up waiting for a peer (its result was NOT a sum)
extern "C" int c3d_peer_status(const c3d_peer_desc* d, int32_t* status_out, int64_t* calls_out) {
  C3D_REQUIRE(d != nullptr && status_out != nullptr, "peer_status: null pointer");
  unsigned long long hdr[2] = {0, 0};
  const hipError_t e = hipMemcpy(hdr, d->mailbox[d->rank], sizeof(hdr), hipMemcpyDeviceToHost);
  if (e != hipSuccess) {
    c3d_set_error(hipGetErrorString(e));
    return 1;
  }
  *status_out = (int32_t)(hdr[1] & 0xffffffffull);
  if (calls_out) *calls_out = (int64_t)hdr[0];
  return 0;
}
