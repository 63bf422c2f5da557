/*
 * coarse3d_hip.h -- C ABI of libcoarse3d_hip.so (gfx950 / MI355X).
 *
 * The upstream COARSE3D reference has no native code and no FFI: its "plugin boundary" is the
 * Python nn.Module API of pc_processor.  This C ABI is the layer the
 * Python mirror of that API (coarse3d_amd/pc_processor) binds with ctypes; every entry point
 * cites the reference code whose arithmetic it replaces.  Paths are relative to the reference
 * root (/root/reference).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 / int32 / int64 / fp64 data as stated;
 *   - activations are NHWC ("channels-last") fp32: element (b,y,x,c) of a tensor with channel
 *     stride S lives at ((b*H + y)*W + x)*S + c;
 *   - `stream` is a hipStream_t (0 = default stream); calls are asynchronous;
 *   - return value 0 = launched; non-zero = refused, see c3d_last_error().
 */
#ifndef COARSE3D_HIP_H
#define COARSE3D_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* c3d_stream;

const char* c3d_last_error(void);
int c3d_version(void);
/* out5 = sizeof(c3d_src), sizeof(c3d_conv_desc), sizeof(c3d_wgrad_desc), sizeof(c3d_pack_entry), sizeof(c3d_wgrad_fold):
 * lets a foreign-language binding check its mirror of the struct layouts (host call, no GPU) */
int c3d_abi_sizes(int32_t* out5);
/* number of workgroups conv kernels will use for an [B,H,W] image: size of stat partial bufs */
int c3d_conv_num_mtiles(int B, int H, int W);

/* bf16 activation storage (BASELINE configs[2], with mfma_bf16 == 1): the entry points below that
 * take a trailing `bf16_mask` read / write the activation pointers whose bit is set as bf16 instead
 * of fp32 (same NHWC indexing; arithmetic, statistics, masks, affines stay fp32).  Bit i = the i-th
 * ACTIVATION pointer of the signature in declaration order:
 *   c3d_conv_in5: out | c3d_conv_in5_wgrad: dz | c3d_affine_add: x, a, out | c3d_axpy: x, y |
 *   c3d_maskpool: in, out | c3d_maskpool_bwd: dout, extra, din | c3d_pixshuf_cat: xa, skip, out |
 *   c3d_pixshuf_cat_bwd: dout, dxa, dskip | c3d_bilinear: src, dst | c3d_bilinear_bwd: dsrc, ddst |
 *   c3d_l2norm: x, y | c3d_l2norm_bwd: y, dy, dx | c3d_bn_bwd_reduce: dy, a | c3d_bn_bwd_apply: dy, a, dz
 * (the convolution / weight-gradient descriptors carry per-tensor flags: c3d_src.bf16, out_bf16, dz_bf16). */

/* ------------------------------------------------------------------ convolution engine */

/* One channel-concatenated input of a convolution, transformed on load:
 *   v = ptr[..., coff + c] * scale[c] + shift[c]   (scale/shift NULL = identity)
 *   v = lrelu ? LeakyReLU(v, 0.01) : v ;  zero outside the image (padding is applied AFTER
 * the transform, as in conv(BN(x)) of salsanext_proto.py:56-62).                            */
typedef struct {
  const float* ptr;
  const float* scale;
  const float* shift;
  int32_t C;        /* channels taken from this source (multiple of 16)  */
  int32_t cstride;  /* channel stride of the underlying tensor           */
  int32_t coff;     /* first channel                                     */
  int32_t lrelu;
  int32_t bf16;     /* 0: ptr is fp32; 1: ptr is bf16 (same NHWC indexing, 2-byte elements) -- bf16
                       activation storage, only with mfma_bf16 == 1 (BASELINE configs[2])         */
  int32_t reserved;
} c3d_src;

typedef struct {
  c3d_src src[3];
  int32_t nsrc;
  int32_t B, H, W;
  int32_t Cout;
  int32_t ntaps;            /* 1, 4 or 9; 3 or 6 (offsets within +-1): strided / transposed convs over column-pair views
                             * (rangenet_proto.py:192-200, 328-334; coarse3d_amd/rangenet.py)      */
  int32_t tap_dy[9];        /* input offset of tap t relative to the output pixel        */
  int32_t tap_dx[9];
  const float* wpack;       /* packed by c3d_pack_weights: [tap][K/4][Cout][4]           */
  const float* bias;        /* [Cout] or NULL                                            */
  int32_t epi_lrelu;        /* LeakyReLU(0.01) after bias                                */
  float* out;               /* NHWC, channel stride out_cstride, first channel out_coff  */
  int32_t out_cstride;
  int32_t out_coff;
  int32_t accumulate;       /* out += result (gradient accumulation)                     */
  float* stat_partial;      /* NULL or [Cout][2][c3d_conv_num_mtiles]: per-tile sum,sumsq */
  float lrelu_slope;        /* slope of the on-load (src.lrelu) and epilogue LeakyReLU; 0 = 0.01
                               (SalsaNext), RangeNet uses 0.1 (rangenet_proto.py:45)           */
  int32_t mfma_bf16;        /* 0: fp32 MFMA (the parity path).  1: operands rounded to bf16
                               (RNE) in LDS->register reads, v_mfma_f32_32x32x16_bf16, fp32
                               accumulate/storage -- opt-in mixed precision (BASELINE config 2).
                               2: every fp32 operand split exactly into three bf16 planes, eight of
                               the nine plane products accumulated in fp32 (fp32-class result on the
                               bf16 pipe); multi-tap convs then need a c3d_pack_weights(mode | 2) pack
                               (wpack_planes = 1).
                               3: as 2 with SIX plane products (l*m and m*l dropped too) -- for INPUT-GRADIENT
                               convolutions (transposed weights, negated taps): measured to leave every
                               gradient's error against float64 unchanged, while on the forward activations
                               six products cost 4-5x the fp32 engine's noise (csrc/conv_x3.hip).
                               4 (EXPERIMENT, round 3): two fp16 planes (x = H + L to 2^-22 |x| worst case), three products
                               H*H' + H*L' + L*H', operands staged times 2^6 / 2^10 against fp16's range, generic
                               kernel; forward convolutions only (profiles/round3_f16x2_probe.txt)            */
  int32_t out_bf16;         /* 1: `out` is bf16 (values rounded RNE on store, accumulate reads bf16);
                               the statistics partials are taken from the fp32 values; mfma_bf16 == 1 only */
  int32_t wpack_planes;     /* 1: wpack came from c3d_pack_weights(mode | 2): the three bf16 planes of the
                               weights follow the fp32 image.  Required by the multi-tap bf16x3 kernels;
                               1x1 convs with Cout > 64 then take the wide kernel of conv_pw3.hip       */
  const float* stat_mul;    /* NULL: stat_partial holds (sum v, sum v*v) of the stored values v.  Else an NHWC tensor of the
                               output's shape (channel stride stat_mul_cstride, first channel 0): stat_partial holds
                               (sum v, sum v * stat_mul) -- the two sums a BatchNorm BACKWARD needs when `out` is the
                               gradient at a BatchNorm's output and stat_mul that BatchNorm's input: the LAST
                               input-gradient launch that contributes to such a gradient (accumulate = 1 included:
                               v is the final value) takes them in its epilogue and spares c3d_bn_bwd_reduce's two
                               tensor reads                                                                        */
  int32_t stat_mul_cstride;
  int32_t variant;          /* 0 = the library picks the kernel schedule.  Test / A-B selector of kernels that compute
                               the same bits with another schedule (tests/test_gpu_conv.py compares them):
                               bits 0-1, 1x1 convs with Cout > 64 on the bf16x3 engine (conv_pw3.hip): 1 = fused kernel
                               with eight waves, 2 = with four waves, 3 = round 2's phased kernel;
                               bit 2, nine-tap convs on the bf16x3 engine (conv_x3.hip) and -- over bf16 tensors -- on the
                               bf16 engine: the phased kernel (round 2's conv_x3_kernel / conv_bfp_kernel) instead of the
                               fused one;
                               bit 3, the bf16 engine's 1x1 / four-tap convs with <= 64 outputs over bf16 tensors
                               (conv_bfp.hip): 16 / 32-channel K chunks widened at load instead of the deeper chunks of
                               raw bf16;
                               round 6 -- bit 4 (16): nine-tap convs of the bf16x3 engine as Winograd F(2x2, 3x3) (conv_wino.hip;
                               wpack is then a c3d_pack_weights_wino pack, stat_partial sized by c3d_conv_wino_num_tiles): the
                               gate experiment, NOT the same bits (another association of the same fp32-class arithmetic),
                               measured slower, off;  bit 5 (32): the staged tile (conv_bfp) where the streaming kernel of the
                               32-channel 1x1 convs would run (conv_pws.hip; agrees to 2e-7 of max: another product order);
                               bit 6 (64): the bf16 engine's BatchNorm-backward epilogue copies its multiplier tile at the
                               start of the epilogue instead of fetching it by LDS-DMA under the last K chunk (same bits);
                               bit 7 (128): input-gradient launches of the fused multi-tap kernel keep the general instance
                               instead of the transform-free one (same bits)                                          */
  const float* acc_scale_dev; /* EXPERIMENT (mfma_bf16 == 4): NULL, or a device scalar the accumulators are multiplied
                               with before bias / activation -- the inverse of a per-tensor gradient exponent
                               (c3d_grad_exponent)                                                            */
  int32_t stat_mul_bf16;    /* 1: stat_mul is a bf16 tensor (round 5: the bf16 engine, mfma_bf16 == 1 with out_bf16 -- the
                               sums are then taken from the values AS STORED, i.e. rounded to bf16: what the separate
                               c3d_bn_bwd_reduce pass would read).  c3d_conv_stat_mul_supported() says whether the kernel
                               this descriptor selects has the epilogue                                         */
  int32_t reserved;
} c3d_conv_desc;
/* 1: a launch of this descriptor (all fields filled in, stat_mul set) takes the BatchNorm-backward sums in its epilogue;
 * 0: the selected kernel has no such epilogue (run c3d_bn_bwd_reduce); host call */
int c3d_conv_stat_mul_supported(const c3d_conv_desc* d);

/* y = epilogue(conv(transform(cat(src)))) as an implicit GEMM on fp32 MFMA.
 * Replaces nn.Conv2d (+LeakyReLU, + the batch statistics of the following BatchNorm2d) in
 * ResContextBlock / ResBlock / UpBlock / cls_head / ProjectionV1
 * (pc_processor/models/salsanext_proto.py:41-62,82-132,164-208,318; projector.py:18-23),
 * and, with transposed weights, their input gradients.                                      */
int c3d_conv_forward(const c3d_conv_desc* d, c3d_stream stream);

/* Winograd F(2x2, 3x3) for the nine-tap convolutions (and input gradients) of the exact-split engine, dilation 1 or 2
 * (salsanext_proto.py:46-52, 87-105, 172-186: every 3x3 Conv2d of ResContextBlock / ResBlock / UpBlock; csrc/conv_wino.hip).
 * c3d_conv_forward takes it with c3d_conv_desc.variant & 16, mfma_bf16 == 2 / 3, ntaps == 9 on the 3 x 3 grid of one dilation:
 * wpack is then a c3d_pack_weights_wino pack and stat_partial holds [Cout][2][c3d_conv_wino_num_tiles] partials (4 x 32-pixel
 * tiles per dilation sub-grid instead of 8 x 32).  The weight transform U = G g G^T is taken from the fp32 image of an ordinary
 * c3d_pack_weights pack (either mode: forward or transposed) + the launch's tap offsets, in float64, rounded once to fp32 and
 * split exactly into three bf16 planes: dst [Kpad / 16][16][3][roundup(N, 64)][16] bf16, c3d_wino_pack_bytes bytes. */
int c3d_conv_wino_num_tiles(int B, int H, int W, int dil);
int64_t c3d_wino_pack_bytes(int Kpad, int N);
int c3d_pack_weights_wino(const float* pack_f32, int Kpad, int N, const int32_t* tap_dy, const int32_t* tap_dx, int dil,
                          void* dst, c3d_stream stream);

/* Weight repack from the reference's OIHW layout [Cout][Cin][T] (T = kh*kw):
 *   mode 0 (forward):  dst[t][k/4][n][k%4] = W[n][k0 + k][t],  k < K=Cin_cnt,  n < Cout
 *   mode 1 (dgrad):    dst[t][k/4][n][k%4] = W[k][n0 + n][t],  k < Cout,       n < N=Cin_cnt
 * K is zero-padded to Kpad (multiple of 16).
 *   mode | 2: dst (2.5x the floats) additionally receives, behind the fp32 image, three bf16
 *   images of the same shape holding the exact split w = h + m + l (h = RNE8(w), m = RNE8(w - h),
 *   l = RNE8(w - h - m)): what c3d_conv_forward reads with mfma_bf16 == 2 for multi-tap convs.   */
int c3d_pack_weights(const float* w_oihw, float* dst, int Cout, int Cin, int T, int mode,
                     int c_off, int c_cnt, int Kpad, c3d_stream stream);
/* The same for a whole model in ONE launch: `table_dev` is a device array of n entries.       */
typedef struct {
  const float* src;   /* OIHW weight                                                          */
  float* dst;         /* packed destination                                                   */
  int32_t Cout, Cin, T, mode, c_off, c_cnt, Kpad, reserved;
} c3d_pack_entry;
int c3d_pack_weights_batch(const c3d_pack_entry* table_dev, int n, c3d_stream stream);

/* One pending fold of weight-gradient partials (strips -> dw), filled by c3d_conv_wgrad when c3d_wgrad_desc.fold_out is
 * set and executed, many per launch, by c3d_wgrad_fold_batch.  HOST memory.                                          */
typedef struct {
  const float* partial;
  float* dw;
  int32_t strips, T, CI, CO, ci_slices, co_slices, Cin_src, Cout, Cin_total, cin_off, accumulate;
  int32_t main_blocks, nblocks, block0;
  const float* bias_partial;
  float* dbias;
  int32_t bias_n;
  float out_scale;
  const float* out_scale_dev;
} c3d_wgrad_fold;

/* dW[cout][cin_off + cin][t] (OIHW, full Cin_total) = sum_pixels dz[p][cout] * x[p + tap t][cin]
 * x is given as ONE transformed source (call once per source of a concatenated input).
 * Replaces autograd's conv weight gradient for the layers listed above.
 * `partial` is scratch of c3d_wgrad_partial_floats() floats -- ask with the descriptor filled in
 * completely: the tiling (and the scratch size) depends on mfma_bf16, shapes and taps.
 * mfma_bf16 == 0: fp32 MFMA (wgrad_mfma.hip).  1 / 2: bf16 matrix pipe with the operands read
 * through the LDS transpose read (wgrad_tr.hip) -- 1: operands rounded to bf16, 2: exact 3-plane
 * split, six of the nine plane products (a sum over >= 10^4 pixels: against float64 the result is
 * as accurate as with eight, its error is the fp32 accumulation's; measured in wgrad_tr.hip).   */
typedef struct {
  c3d_src x;                /* input of the forward conv (same transform as forward)      */
  const float* dz;          /* NHWC [B,H,W,Cout] gradient w.r.t. the conv output (pre-act) */
  int32_t dz_cstride;
  int32_t B, H, W, Cout;   /* Cout = rows of dw; dz channels beyond it must be zero       */
  int32_t ntaps;            /* 1, 3, 4, 6 or 9 (as in c3d_conv_desc)                       */
  int32_t tap_dy[9];
  int32_t tap_dx[9];
  int32_t Cin_total;        /* Cin of the OIHW weight tensor                              */
  int32_t cin_off;          /* where this source's channels start inside Cin_total        */
  float* dw;                /* OIHW gradient [Cout][Cin_total][T]                         */
  int32_t accumulate;
  float* partial;
  int32_t mfma_bf16;        /* as in c3d_conv_desc                                         */
  float lrelu_slope;        /* as in c3d_conv_desc                                         */
  int32_t dz_bf16;          /* 1: dz is bf16 (x.bf16 says the same for x); mfma_bf16 == 1 only */
  int32_t variant;          /* schedule selector of the bit-identity tests / A-B runs: 0 = the library's choice.  Fused launches
                             * (fuse_dy) of the kernels with a rolling window: 1 = whole-window register sets (round 4), 2 = lean
                             * register sets (round 5) -- the same bits either way.  Bit 2 (4): a fused 1x1 launch keeps the
                             * 128 x 256 slice of the unfused one (another summation order).  Bit 7 (128): four producer waves
                             * in the 1x1 instances of the three-plane engine that run eight (round 5) -- the same bits.  Bit 8
                             * (256): fused nine-tap launches, and the fused four-tap one over 64 x 64 slices, on four + four
                             * waves instead of eight + eight with the producer waves split by tensor (round 6) -- the same
                             * bits (dw, dz and the bias sums).  Bits 7 and 8 alike keep the unfused four-tap 64 x 64 and the
                             * unfused 1x1 128 x 256 launches on four + four waves (round 6: sixteen, consumers split)      */
  /* optional: fold the layer's bias-gradient partials in the same launch that folds the weight-gradient strips
   * (saves one tiny launch per conv layer): dbias[c] = sum_k bias_partial[c][0][k], k < bias_n, the
   * [Cout][2][bias_n] partials c3d_bn_bwd_apply wrote (what c3d_bias_from_partials computes).  NULL = off. */
  const float* bias_partial;
  float* dbias;
  int32_t bias_n;
  int32_t reserved2;
  /* EXPERIMENT (mfma_bf16 == 4, two fp16 planes / three products): dz is multiplied with dz_scale[cout] (2^s,
   * c3d_grad_exponent_max) while it is staged, x with 2^6; the fold multiplies 2^-6 * *out_scale_dev (2^-s) back */
  const float* dz_scale;
  const float* out_scale_dev;
  /* BatchNorm / LeakyReLU backward ON LOAD (round 4; mfma_bf16 == 2, fp32 tensors).  fuse_dy != NULL: `dz` is an OUTPUT.
   * The launch reads the gradient at the layer's consumer-visible output (fuse_dy) and the layer's stored output
   * (fuse_act = LeakyReLU(conv + bias), pre-BatchNorm), forms
   *     dz = LeakyReLU'(act) * (k1[c] * dy + k2[c] * act + k3[c])        (k1 == NULL: dz = LeakyReLU'(act) * dy)
   * or, with fuse_pre_scale / fuse_pre_shift (a conv -> BatchNorm -> LeakyReLU layer: the derivative of the activation is
   * taken at BN(act) and multiplies dy first, c3d_bn_bwd_apply's mode 1),
   *     dz = k1[c] * (LeakyReLU'(act * pre_scale[c] + pre_shift[c]) * dy) + k2[c] * act + k3[c]
   * while it stages its pixel tiles -- what c3d_bn_bwd_apply computes in a pass of its own (modes 0, 1 and 2) -- uses it as
   * the weight gradient's operand, WRITES it to `dz` (the input-gradient convolution that follows reads it) and leaves the
   * per-channel sums of dz in fuse_sum [Cout][2][c3d_wgrad_fused_sum_n()] (row 0; the bias gradient: pass it as
   * bias_partial with bias_n = that n).  All three tensors share dz_cstride; `dz` must not alias fuse_dy.
   * Reference: autograd of BatchNorm2d + LeakyReLU in salsanext_proto.py:56-62,117-140,185-205.                        */
  const float* fuse_dy;
  const float* fuse_act;
  const float* fuse_k1;
  const float* fuse_k2;
  const float* fuse_k3;
  float* fuse_sum;
  const float* fuse_pre_scale;
  const float* fuse_pre_shift;
  /* NULL: the call folds its strips into dw right away (one more tiny launch).  Else a HOST record that receives the
   * pending fold; `partial` (and bias_partial) must then stay valid until c3d_wgrad_fold_batch has run on the stream.
   * Two deferred folds of one batch must not accumulate into the same dw elements.                                  */
  c3d_wgrad_fold* fold_out;
} c3d_wgrad_desc;
int64_t c3d_wgrad_partial_floats(const c3d_wgrad_desc* d);
/* entries per channel of fuse_sum for this descriptor (0: this shape / engine has no fused form -- run c3d_bn_bwd_apply) */
int c3d_wgrad_fused_sum_n(const c3d_wgrad_desc* d);
/* folds[0 .. n) (HOST array of records filled by c3d_conv_wgrad) in ceil(n / 32) launches; nothing in the training step reads
 * a weight gradient before the optimiser, so one call at the end of the backward pass replaces ~70 tiny launches     */
int c3d_wgrad_fold_batch(const c3d_wgrad_fold* folds, int n, c3d_stream stream);
int c3d_conv_wgrad(const c3d_wgrad_desc* d, c3d_stream stream);

/* ------------------------------------------------------------------ BatchNorm2d (train mode)
 * nn.BatchNorm2d of salsanext_proto.py:46,50,89-105,168-180 and projector.py:20, split so that
 * the fp64 sums can be all-reduced across ranks in between (SyncBatchNorm, trainer.py:54).   */

/* partial [C][2][n] fp32 (per-tile sum, sumsq) -> sums [C][2] fp64 */
int c3d_stat_reduce(const float* partial, int n, int C, double* sums, c3d_stream stream);
/* the same, written twice: sums is about to be all-reduced in place, sums_copy stays rank-local (SyncBatchNorm backward:
 * dgamma / dbeta come from the local sums, the input gradient from the global ones)                     */
int c3d_stat_reduce2(const float* partial, int n, int C, double* sums, double* sums_copy, c3d_stream stream);
/* sums + count -> consumer-side affine scale=gamma*invstd, shift=beta-mean*scale; saves
 * mean/invstd for backward; updates running stats (momentum, unbiased var) when non-NULL.   */
int c3d_bn_finalize(const double* sums, double count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps, int C,
                    float* scale, float* shift, float* save_mean, float* save_invstd,
                    c3d_stream stream);
/* eval mode: affine from running statistics */
int c3d_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, int C, float* scale, float* shift,
                       c3d_stream stream);
/* Backward of  a -> BN -> (consumers)  where a = LeakyReLU(conv) [mode 0], of
 * a -> BN -> LeakyReLU [mode 1, projector; pre_scale/pre_shift = the forward affine],
 * of a = LeakyReLU(conv) without BN [mode 2], or identity [mode 3]; dy/a/dz are [npix][cs].
 *   reduce : partial [C][2][c3d_bn_bwd_num_blocks] = (sum dy, sum dy*a)
 *   coeffs : da = k1*dy + k2*a + k3 ; dgamma = sum(dy*xhat) ; dbeta = sum(dy)
 *   apply  : dz = act'(.) * da ; partial[C][0][..] = sum dz  (conv bias gradient)            */
int c3d_bn_bwd_num_blocks(int npix);
int c3d_bn_bwd_reduce(const float* dy, int dy_cs, const float* a, int a_cs, int npix, int C,
                      int mode, const float* pre_scale, const float* pre_shift, float* partial,
                      float lrelu_slope /* 0 = 0.01 */, int bf16_mask, c3d_stream stream);
/* sums = (all-reduced) statistics for k1..k3; sums_param = this rank's own statistics for
 * dgamma/dbeta (NULL = same as sums): SyncBatchNorm semantics under data parallelism          */
int c3d_bn_bwd_coeffs(const double* sums, const double* sums_param, double count,
                      const float* mean, const float* invstd, const float* gamma, int C,
                      float* k1, float* k2, float* k3, float* dgamma, float* dbeta,
                      c3d_stream stream);
int c3d_bn_bwd_apply(const float* dy, int dy_cs, const float* a, int a_cs, int npix, int C,
                     int mode, const float* pre_scale, const float* pre_shift, const float* k1,
                     const float* k2, const float* k3, float* dz, int dz_cs, float* partial,
                     float lrelu_slope /* 0 = 0.01 */, int bf16_mask, c3d_stream stream);
/* Single-rank fast paths: fold the partials [C][2][n] and finish in one launch (no all-reduce
 * hook in between): = c3d_stat_reduce + c3d_bn_finalize / + c3d_bn_bwd_coeffs / + column 0.    */
int c3d_bn_finalize_partials(const float* partial, int n, double count, const float* gamma,
                             const float* beta, float* running_mean, float* running_var,
                             float momentum, float eps, int C, float* scale, float* shift,
                             float* save_mean, float* save_invstd, c3d_stream stream);
int c3d_bn_bwd_coeffs_partials(const float* partial, int n, double count, const float* mean,
                               const float* invstd, const float* gamma, int C, float* k1,
                               float* k2, float* k3, float* dgamma, float* dbeta,
                               c3d_stream stream);
int c3d_bias_from_partials(const float* partial, int n, int C, float* out, int accumulate,
                           c3d_stream stream);
/* out[c] (+)= (float) sums[c][col] */
int c3d_sums_to_f32(const double* sums, int C, int col, float* out, int accumulate,
                    c3d_stream stream);

/* ------------------------------------------------------------------ block glue (HBM-bound) */

/* x = (x - mean[c]) / std[c] * (eval_label > 0)   NCHW  (trainer.py:599-609)                */
int c3d_input_norm(const float* x, const int64_t* eval_label, const float* mean,
                   const float* stdv, int B, int Cn, int HW, float* out, c3d_stream stream);
/* downCntx.conv1: 1x1 conv Cn(<=8) -> 32 + LeakyReLU, NCHW in, NHWC out (salsanext_proto.py:41,53-54) */
int c3d_conv_in5(const float* x_nchw, const float* w, const float* bias, int B, int Cn, int HW,
                 float* out, int bf16_mask, c3d_stream stream);
/* its weight gradient dw[32][Cn]; partial = scratch of 1024*32*8 floats                      */
int c3d_conv_in5_wgrad(const float* x_nchw, const float* dz, int B, int Cn, int HW,
                       float* partial, float* dw, int bf16_mask, c3d_stream stream);
/* out = x + act(a*scale + shift)  (x, scale may be NULL; act = LeakyReLU(lrelu_slope), or the
 * identity when lrelu_slope == 0): residual adds of salsanext_proto.py:64,133 and of RangeNet's
 * BasicBlock (rangenet_proto.py:52-63)                                                       */
int c3d_affine_add(const float* x, const float* a, const float* scale, const float* shift,
                   int64_t npix, int C, float lrelu_slope, float* out, int bf16_mask, c3d_stream stream);
/* NHWC [rows][Win][C] column resampling: up = 0 keeps the even columns (W -> W/2: a stride-(1,2)
 * conv = stride-1 conv + this, rangenet_proto.py:194-203); up = 1 inserts a zero column after
 * every column (W -> 2W: ConvTranspose2d([1,4],[1,2],[0,1]) = this + a 4-tap conv, :328-336).
 * The two are adjoint: each is the other's backward.                                         */
int c3d_cols_resample(const float* in, int64_t rows, int Win, int C, int up, float* out,
                      c3d_stream stream);
/* x NCHW [B][Cn][HW] -> NHWC [B][HW][Cp] with zero channels Cn..Cp-1 (5-channel input as a
 * 16-channel MFMA operand)                                                                   */
int c3d_nchw_to_nhwc_pad(const float* x, int B, int Cn, int64_t HW, int Cp, float* out,
                         c3d_stream stream);
/* y (+)= alpha*x, flat                                                                      */
int c3d_axpy(const float* x, float alpha, int64_t n, float* y, int accumulate, int bf16_mask, c3d_stream stream);
/* Dropout2d multiplier mask[B,C] (NULL = none) then AvgPool2d(3,2,1) if pool (:108-109,135-142) */
int c3d_maskpool(const float* in, const float* mask, int B, int H, int W, int C, int pool,
                 float* out, int bf16_mask, c3d_stream stream);
/* din = extra + mask * pool^T(dout)   (extra may be NULL)                                   */
int c3d_maskpool_bwd(const float* dout, const float* mask, const float* extra, int B, int H,
                     int W, int C, int pool, float* din, int bf16_mask, c3d_stream stream);
/* out = cat(PixelShuffle2((xa*sc+sh)*m3)*m1, skip) * m2   (:185-191; masks may be NULL)     */
int c3d_pixshuf_cat(const float* xa, const float* sc, const float* sh, const float* m3,
                    const float* m1, const float* m2, const float* skip, int B, int Hs, int Ws,
                    int Cx, int Cs, float* out, int bf16_mask, c3d_stream stream);
int c3d_pixshuf_cat_bwd(const float* dout, const float* m3, const float* m1, const float* m2,
                        int B, int Hs, int Ws, int Cx, int Cs, float* dxa, float* dskip,
                        int skip_accumulate, int bf16_mask, c3d_stream stream);
/* softmax over the first C of cs channels, cropped to [Ho,Wo] (:456-460)                    */
int c3d_softmax(const float* logits, int B, int H, int W, int cs, int C, int Ho, int Wo,
                float* prob, c3d_stream stream);
int c3d_softmax_bwd(const float* prob, const float* dprob, int B, int H, int W, int cs, int C,
                    int Ho, int Wo, float* dlogits, c3d_stream stream);
/* F.interpolate(bilinear, align_corners=True) between channel slices of NHWC tensors (:470-490) */
int c3d_bilinear(const float* src, int Hs, int Ws, int scs, int scoff, float* dst, int Hd,
                 int Wd, int dcs, int dcoff, int B, int C, int bf16_mask, c3d_stream stream);
/* dst[B,Hd,Wd,C] = bilinear(src1[B,Hs1,Ws1,C]) + bilinear(src2[B,Hs2,Ws2,C]) (align_corners=True, fp32 sources, dense
 * channels; dst_bf16 != 0: dst is a bf16 activation tensor).  A 1x1 convolution commutes with the resampling of its input: ProjectionV1's first conv over the
 * concatenation of four resampled skips (salsanext_proto.py:466-483, projector.py:18) is evaluated per skip at the
 * skip's own resolution where that has fewer pixels, and the low-resolution results meet here.            */
int c3d_bilinear_sum2(const float* src1, int Hs1, int Ws1, const float* src2, int Hs2, int Ws2,
                      float* dst, int Hd, int Wd, int B, int C, int dst_bf16, c3d_stream stream);
/* dsrc (+)= bilinear^T(ddst): deterministic gather over the destination pixels that read each
 * source pixel (no atomics); accumulate != 0 adds to the existing dsrc.  ddst_rowmask (may be NULL): one bit per
 * destination pixel (bit p & 31 of word p >> 5, p = (b*Hd + y)*Wd + x); pixels whose bit is clear are known to hold
 * zeros and are not read (the contrast loss' gradient touches ~10^3 of 10^6 pixels: c3d_scatter_add_rows writes the mask) */
int c3d_bilinear_bwd(float* dsrc, int Hs, int Ws, int scs, int scoff, const float* ddst, int Hd,
                     int Wd, int dcs, int dcoff, int B, int C, int accumulate, int bf16_mask, const uint32_t* ddst_rowmask, c3d_stream stream);
/* The same adjoint for a destination gradient kept in COMPACT form: drows [rows][C] fp32 + cmap[pixel] = row of that
 * pixel (valid where the rowmask bit is set; c3d_scatter_rows_compact writes all three).  Same summation order as
 * c3d_bilinear_bwd over the equivalent dense, zero-filled ddst: bit-identical result, without the dense tensor
 * (1.07 GB for the 8x64x2048x256 embedding of salsanext_proto.py:488-490).                                        */
int c3d_bilinear_bwd_rows(float* dsrc, int Hs, int Ws, int scs, int scoff, const float* drows, const int32_t* cmap,
                          const uint32_t* rowmask, int Hd, int Wd, int B, int C, int accumulate, int dsrc_bf16,
                          c3d_stream stream);
/* out[r][0..C) = row of F.interpolate(src, (Hd, Wd), bilinear, align_corners=True) at ONE destination pixel per r --
 * the rows contrast_pixel_loss.py (anchors) and prototype_learning (labelled pixels, salsanext_proto.py:494-527) read
 * of the upsampled embedding, without materialising it; bit-identical to the rows of c3d_bilinear's output.
 * Pixel of row r: image img[r / A], pixel idx[r] (img != NULL, int32 idx) or flat b*Hd*Wd + pixel = idx[r]
 * (img == NULL; idx64 != 0: int64 idx).  Rows with r / A >= *count (count may be NULL) are zero.  l2 != 0: rows are
 * l2-normalised like c3d_gather_rows_l2 and norm[r] (may be NULL) receives their length.                          */
int c3d_bilinear_rows(const float* src, int Hs, int Ws, int scs, int scoff, int src_bf16, int Hd, int Wd, int B,
                      int C, const int32_t* img, const void* idx, int idx64, int A, const int32_t* count, int R,
                      int l2, float eps, float* out, float* norm, c3d_stream stream);
/* F.normalize(p=2) over rows of [n][C] (:485; eps 1e-12); norm may be NULL                  */
int c3d_l2norm(const float* x, int64_t n, int C, float eps, float* y, float* norm, int bf16_mask,
               c3d_stream stream);
int c3d_l2norm_bwd(const float* y, const float* norm, const float* dy, int64_t n, int C,
                   float eps, float* dx, int bf16_mask, c3d_stream stream);

/* ------------------------------------------------------------------ prototype memory bank
 * salsanext_proto.py:494-510 (similarity) and :337-402 (prototype_learning), sinkhorn.py:5-33.
 * The [N,D] x [D, M*C] similarity GEMM runs on c3d_conv_forward (1x1).                       */

/* out = l2_normalize(LayerNorm(x)) per row of [n][C]   (:497-501)                            */
int c3d_rownorm_ln_l2(const float* x, int64_t n, int C, const float* ln_w, const float* ln_b,
                      float ln_eps, float l2_eps, float* out, c3d_stream stream);
/* sim [n][M*C] (column m*C+k) -> nearest [n][C] = LayerNorm(max_m sim) (may be NULL) and
 * pred[n] = argmax_k nearest   (:506-507, :340)                                              */
int c3d_proto_nearest(const float* sim, int64_t n, int M, int C, const float* ln_w,
                      const float* ln_b, float eps, float* nearest, int32_t* pred,
                      c3d_stream stream);
/* Ordered lists: idx[g][c][0..counts[g][c]) = ascending positions i with labels[g][i]==c
 * (labels masked to 0 where keep[g][i]==0, keep may be NULL).  idx is [groups][ncls][n];
 * seg_scratch = groups*8*ncls int32 (per-segment class histograms).                          */
int c3d_group_compact(const int64_t* labels, const uint8_t* keep, int groups, int n, int ncls,
                      int32_t* counts, int32_t* idx, int32_t* seg_scratch, c3d_stream stream);
/* counts[g][c] = number of labels equal to c (c >= 1) in group g                             */
int c3d_label_hist(const int64_t* labels, int groups, int n, int ncls, int32_t* counts,
                   c3d_stream stream);
/* Per class: Sinkhorn (3 iters, eps .05) on sim[rows,:,c], argmax -> target, Gumbel-hard
 * one-hot from Exp(1) `noise` [B*n][M] (indexed by pixel), masked feature sums, EMA into the
 * bank and final l2.  counts [B][C] / idx [B][C][n] from c3d_group_compact(groups=B) on the
 * labels of the B images; rows [C][B*n] and assign [B*n] are scratch; target must be zeroed.
 * pred = c3d_proto_nearest's argmax, or NULL to evaluate it (with the mask_norm LayerNorm
 * ln_w/ln_b/ln_eps) for the labelled pixels only.
 * fsum = NULL: the bank is updated by the call (protos_out).  fsum = [C][M][D+1]: the masked
 * feature sums (and, in column D, the assignment counts) are written there instead and the bank
 * is left alone -- data-parallel ranks all-reduce fsum and finish with c3d_proto_ema (the
 * "per-class prototype sums" exchange).
 * cmap = NULL: sim / feat hold one row per pixel.  cmap = [B*n] int32: sim / feat are COMPACT -- only the
 * labelled pixels were normalised and multiplied with the bank (their rows in any order) -- and cmap[pixel] is
 * the pixel's row in them (read at labelled pixels only); target stays indexed by pixel, noise too unless
 * noise_by_row != 0: then noise is [rows][M], indexed like sim / feat (the caller drew variates for the compact rows
 * only: 0.6 MB instead of 84 MB at 8x64x2048).                                                          */
int c3d_proto_learn(const float* sim, const float* feat, const int32_t* pred, const float* ln_w,
                    const float* ln_b, float ln_eps, const int32_t* counts, const int32_t* idx, int32_t* rows, const float* noise,
                    const float* protos, float* protos_out, float* target, int32_t* assign,
                    int B, int n, int M, int C, int D, int ignore_label, float momentum,
                    float* fsum, const int32_t* cmap, int noise_by_row, c3d_stream stream);
/* EXPERIMENT (f16x2 gradients): per-tensor power-of-two scale of a gradient tensor from the maxima c3d_bn_bwd_apply
 * leaves in row 1 of its partials [C][2][n]: scale_out[0 .. scale_len) = 2^s (a per-channel `scale` array for c3d_src),
 * *inv_out = 2^-s (c3d_conv_desc.acc_scale_dev), s = target_log2 - ceil(log2 max|dz|).                              */
int c3d_grad_exponent(const float* partial, int n, int C, int target_log2, float* scale_out, int scale_len,
                      float* inv_out, c3d_stream stream);
/* The same without a reduction launch: c3d_bn_bwd_apply_gmax is c3d_bn_bwd_apply that also folds max |dz| into *gmax (the
 * bits of a non-negative float, zeroed by the caller; atomicMax: exact, order-free), c3d_grad_exponent_max turns that word
 * into the scale array and its inverse.                                                                              */
int c3d_bn_bwd_apply_gmax(const float* dy, int dy_cs, const float* a, int a_cs, int npix, int C, int mode,
                          const float* pre_scale, const float* pre_shift, const float* k1, const float* k2,
                          const float* k3, float* dz, int dz_cs, float* partial, float lrelu_slope, int bf16_mask,
                          uint32_t* gmax, c3d_stream stream);
int c3d_grad_exponent_max(const uint32_t* gmax, int target_log2, float* scale_out, int scale_len, float* inv_out,
                          c3d_stream stream);
/* EMA with the l2-normalised sums + final l2 normalisation (salsanext_proto.py:376-395, :402)  */
int c3d_proto_ema(const float* fsum, const float* protos, float* protos_out, int M, int C, int D,
                  int ignore_label, float momentum, c3d_stream stream);

/* ------------------------------------------------------------------ contrast loss + PL selection
 * contrast_pixel_loss.py:27-195, trainer.py:447-518                                          */

/* prob [n][C] -> w_anchor = exp(-H^2), w_pl = exp(-H), amax (any output may be NULL)         */
int c3d_entropy_stats(const float* prob, int64_t n, int C, float* w_anchor, float* w_pl,
                      int32_t* amax, c3d_stream stream);
/* entropy_based_selection: per (image b, class c present in train_label): k = int(cnt*ratio)
 * pixels with the largest w_pl/noise among {amax==c, eval>0}; noise [B][C][n] Exp(1);
 * tl_counts [B][C] = weak-label counts; chosen [B][n] zeroed scratch; scratch = 2*B*C + 2*B*n
 * int32 (bucketed keys).                                                                     */
int c3d_pl_select(const float* w_pl, const int32_t* amax, const int64_t* eval_label,
                  const int64_t* train_label, const float* noise, const int32_t* tl_counts,
                  int B, int n, int C, int ignore_label, float ratio, int32_t* scratch,
                  uint8_t* chosen, int64_t* labels_out, uint8_t* mask_out, c3d_stream stream);
/* the same with select_ratio (trainer.py:655-661, a function of the epoch) read from a device scalar: a captured
 * training step (hipGraph) is then valid for every epoch                                                */
int c3d_pl_select_dev(const float* w_pl, const int32_t* amax, const int64_t* eval_label,
                      const int64_t* train_label, const float* noise, const int32_t* tl_counts,
                      int B, int n, int C, int ignore_label, const float* ratio_dev, int32_t* scratch,
                      uint8_t* chosen, int64_t* labels_out, uint8_t* mask_out, c3d_stream stream);
/* anchor_sampling: bit-exact torch.multinomial(replacement=True) per present (b,c) pair;
 * the t-th present pair consumes uniforms[t][0..A).  counts/idx from c3d_group_compact.
 * slot [B*C], cum [B][C][n] are scratch.  Outputs anchor_idx [B*C][A] (pixel in image),
 * anchor_img / anchor_cls [B*C], *T = number of present pairs.                               */
int c3d_anchor_sample(const float* weights, const int32_t* counts, const int32_t* idx,
                      const double* uniforms, int B, int n, int C, int A, int ignore_label,
                      int32_t* slot, float* cum, int32_t* anchor_idx, int32_t* anchor_img,
                      int32_t* anchor_cls, int32_t* T, c3d_stream stream);
/* out[t*A+s] = l2_normalize(feat[img[t]][idx[t][s]]) for t < *T, zero rows otherwise         */
int c3d_gather_rows_l2(const float* feat, const int32_t* img, const int32_t* idx,
                       const int32_t* T, int Tmax, int A, int n, int D, float eps, float* out,
                       float* norm, c3d_stream stream);
/* dfeat[img[t]][idx[t][s]] += (*gscale) * dx[t*A+s]   (gscale may be NULL).  Repeated pixels inside a pair t are
 * summed in ascending s by one wave and added with a plain store (bit-reproducible); pixels must not repeat ACROSS
 * pairs (they cannot: a pixel has one class -- contrast_pixel_loss.py anchor sampling is per (image, class)).
 * rowmask (may be NULL, else pre-zeroed, (B*n + 31)/32 words): bit img*n + pixel is set for every row written.    */
int c3d_scatter_add_rows(const float* dx, const int32_t* img, const int32_t* idx,
                         const int32_t* T, int Tmax, int A, int n, int D, const float* gscale,
                         float* dfeat, uint32_t* rowmask, c3d_stream stream);
/* c3d_scatter_add_rows into a compact gradient: drows[t*A+s] (the slot of the FIRST occurrence s of a pixel in its
 * pair) = (*gscale) * sum of the pixel's rows, cmap[img*n + pixel] = that slot, rowmask bit set (pre-zeroed by the
 * caller; cmap needs no initialisation: it is only read where the bit is set).  Feeds c3d_bilinear_bwd_rows.      */
int c3d_scatter_rows_compact(const float* dx, const int32_t* img, const int32_t* idx, const int32_t* T, int Tmax,
                             int A, int n, int D, const float* gscale, float* drows, int32_t* cmap,
                             uint32_t* rowmask, c3d_stream stream);
/* InfoNCE over cosine logits [Tmax*A][ld] (first ncols=(C-1)*M columns valid, column class
 * 1 + j/M): replaces logits by d(mean loss)/d(logits) in place, row_loss per row, *loss.     */
int c3d_infonce_rows(float* logits, int ld, const int32_t* row_cls, const int32_t* T, int Tmax,
                     int A, int M, int ncols, float temperature, float base_temperature,
                     float* row_loss, float* loss, c3d_stream stream);

/* ------------------------------------------------------------------ per-iteration metrics
 * tasks/weak_segmentation/trainer.py:713-730, pc_processor/metrics/iou_eval.py:35-58          */

/* For each of the n points of ONE scan: pred = argmax_c prob[pix][c] (first maximum) with
 * pix = uy[i]*W + ux[i] (SemanticKitti / nuScenes, trainer.py:718-719) or pix = uy[i] when
 * ux == NULL (SemanticPOSS, trainer.py:720-726: points i >= n_valid predict class 0), then
 * conf[pred][labels[i]] += 1 (int64 [C][C], accumulated).  prob is the NHWC image of that scan
 * with channel stride cstride; pred_out (int32 [n]) may be NULL.                               */
int c3d_unproject_confusion(const float* prob, int H, int W, int C, int cstride,
                            const int32_t* uy, const int32_t* ux, const int64_t* labels,
                            int64_t n, int64_t n_valid, int64_t* conf, int32_t* pred_out,
                            c3d_stream stream);
/* IOUEval.addBatch: conf[pred[i]][label[i]] += 1                                               */
int c3d_confusion_add(const int64_t* pred, const int64_t* label, int64_t n, int C, int64_t* conf,
                      c3d_stream stream);

/* ------------------------------------------------------------------ loss head   */

/* FocalSoftmaxLoss on probabilities (pc_processor/loss/focal_softmax.py:30-77, softmax=False):
 * loss_i = -(1-pt)^gamma * log(max(pt,1e-6)) * alpha[target_i] over the pixels with mask != 0
 * (mask NULL = all), out[0] = mean (0 when nothing is selected, as the reference's NaN guard),
 * out[1] = number of selected pixels.  partial = 2*nblk doubles of scratch.                     */
int c3d_focal_forward(const float* prob, int C, int cstride, const int64_t* target,
                      const uint8_t* mask, const float* alpha, float gamma, int64_t n,
                      double* partial, int nblk, float* out, c3d_stream stream);
/* dprob[i][target_i] += (*gscale) * d out[0] / d prob[i][target_i]   (stats = out of forward)  */
int c3d_focal_backward(const float* prob, int C, int cstride, const int64_t* target,
                       const uint8_t* mask, const float* alpha, float gamma, int64_t n,
                       const float* stats, const float* gscale, float* dprob, int dstride,
                       c3d_stream stream);
/* Lovasz_softmax(classes='present', per_image=False) on the P labelled pixels idx[0..P)
 * (pc_processor/loss/lovasz_softmax.py:56-68,101-176): one workgroup per class sorts the errors
 * |fg - p| in LDS (P <= c3d_lovasz_max_pixels()), forms the Jaccard gradient and the class loss;
 * loss_c/present [C], grad [C][P] = d loss_c / d prob[idx[p]][c], out[0] = mean over the present
 * classes, out[1] = their number.                                                              */
int c3d_lovasz_max_pixels(void);
int c3d_lovasz_forward(const float* prob, int C, int cstride, const int64_t* labels,
                       const int64_t* idx, int P, float* loss_c, float* present, float* grad,
                       float* out, c3d_stream stream);
/* dprob[idx[p]][c] += (*gscale) / out[1] * grad[c][p]                                          */
int c3d_lovasz_backward(const float* grad, const int64_t* idx, int P, int C, const float* stats,
                        const float* gscale, float* dprob, int dstride, c3d_stream stream);
/* Lovasz beyond c3d_lovasz_max_pixels() labelled pixels (fully supervised batches; the reference sorts any number,
 * lovasz_softmax.py:56-68,140-160): same outputs as c3d_lovasz_forward, the per-class sort is a device-wide segmented
 * radix sort.  workspace >= c3d_lovasz_workspace_bytes(C, P) bytes; C * P < 2^31.  The backward is
 * c3d_lovasz_backward (it has no capacity).                                                            */
int64_t c3d_lovasz_workspace_bytes(int C, int P);
int c3d_lovasz_forward_large(const float* prob, int C, int cstride, const int64_t* labels,
                             const int64_t* idx, int P, float* loss_c, float* present, float* grad,
                             float* out, void* workspace, int64_t workspace_bytes, c3d_stream stream);
/* The same with the number of labelled pixels read from DEVICE memory (*P_dev, clamped to P_cap <=
 * c3d_lovasz_max_pixels()): idx holds P_cap entries of which the first *P_dev are used, grad is [C][P_cap].  The
 * launches are shape-static: the training step can be captured in a hipGraph and replayed on batches with
 * different numbers of weak labels (the reference's host-side mask indexing, lovasz_softmax.py:140-160, becomes a
 * device-side compaction + these kernels).                                                       */
int c3d_lovasz_forward_dyn(const float* prob, int C, int cstride, const int64_t* labels,
                           const int64_t* idx, const int* P_dev, int P_cap, float* loss_c, float* present,
                           float* grad, float* out, c3d_stream stream);
int c3d_lovasz_backward_dyn(const float* grad, const int64_t* idx, const int* P_dev, int P_cap, int C,
                            const float* stats, const float* gscale, float* dprob, int dstride,
                            c3d_stream stream);

/* ------------------------------------------------------------------ kNN label clean-up
 * pc_processor/postproc/knn.py:36-142 (KNN.forward), un-batched as the reference: for each of
 * the n points (pixel px,py; range unproj_range) the `knn` candidates of the search x search
 * window of proj_range [H][W] with the smallest |range - own range| * inv_gauss[k] (window
 * padded with zeros, invalid pixels (< 0) at infinity, centre = the point itself) vote with their
 * proj_argmax label; candidates beyond `cutoff` (> 0) do not vote; out[i] = most voted class in
 * 1..nclasses-1 (first on ties).  inv_gauss = 1 - gaussian kernel, search*search floats.        */
int c3d_knn_vote(const float* proj_range, const int64_t* proj_argmax, int H, int W,
                 const float* unproj_range, const int64_t* px, const int64_t* py, int64_t n,
                 const float* inv_gauss, int search, int knn, float cutoff, int nclasses,
                 int64_t* out, c3d_stream stream);

/* ------------------------------------------------------------------ scan -> range image
 * pc_processor/dataset/preprocess/projection.py:43-115, augmentor.py:150-230,
 * pc_processor/dataset/semantic_kitti/wss_sem_kitti_loader.py:113-170                          */

/* In place on pc [n][stride] (x, y, z first): x = sx*x + tx, y = sy*y + ty, z = z + tz (float32),
 * then xyz = xyz (as float64) x rot^T rounded to float32; rot_dev = 9 doubles (row major) on the
 * device.  Identity rotation / zero translation / sx = sy = 1 reproduce the input bits.          */
int c3d_augment_points(float* pc, int n, int stride, float sx, float sy, float tx, float ty,
                       float tz, const double* rot_dev, c3d_stream stream);
/* RangeProjection.doProjection: per point depth (or `depth` [n] if not NULL), pixel coordinates
 * ux/uy [n] and udepth [n] (= cached_data), closest point per pixel through zbuf (H*W uint64 of
 * scratch).  Any of the image outputs may be NULL: proj_pc [H][W][cols] (-1 where empty),
 * proj_range [H][W], proj_idx [H][W] (-1), proj_mask [H][W] = (proj_idx > 0); and the loader
 * tensors of wss_sem_kitti_loader.py: feat5 [5][H][W] = (range, x, y, z, intensity with -1 -> 0),
 * eval_label / train_label [H][W] float32 = sem / weak label of the winning point (0 if empty).
 * fov_*: |fov_left|, fov_hori, |fov_down|, fov_vert in radians, as float32.                      */
int c3d_range_project(const float* pc, int n, int stride, int cols, const float* depth,
                      float fov_left_abs, float fov_hori, float fov_down_abs, float fov_vert,
                      int W, int H, int32_t* ux, int32_t* uy, float* udepth, uint64_t* zbuf,
                      float* proj_pc, float* proj_range, int32_t* proj_idx, int32_t* proj_mask,
                      const int64_t* sem, const int64_t* weak, float* feat5, float* eval_label,
                      float* train_label, c3d_stream stream);

/* ------------------------------------------------------------------ weak-label voxel sampler
 * tasks/prepare_data/gen_sem_weak_label_rand_grid.py:190-246 (SemanticData.__getitem__): voxel
 * grid of edge `voxel_size` over one scan (open3d 0.15.2 rule: origin = min_bound - voxel_size/2,
 * voxel = floor((p - origin)/voxel_size) in double; :190-202), voxels in np.unique(axis=0) order
 * with the label of their first point (:203-207), the `n_sample` voxels with label > 0 that
 * have the smallest priority[first point] (a uniform sample without replacement when the
 * priorities are i.i.d.; :218-225), and their label written to all points of the voxel
 * (propagate != 0) or to its first point only (:233-241).
 *   xyz [n][stride] float32 (x, y, z first), label int32 [n] (mapped classes, 0 = ignore),
 *   priority float32 [n], workspace >= c3d_voxel_sampler_workspace_bytes(n) bytes,
 *   point2voxel int32 [n][3] or NULL, weak int32 [n],
 *   stats int32 [5] = {points with non-finite / out-of-range coordinates, voxels, voxels with
 *   label > 0, voxels sampled (< n_sample: not enough valid voxels), labelled points}.          */
int64_t c3d_voxel_sampler_workspace_bytes(int n);
int c3d_voxel_weak_labels(const float* xyz, int n, int stride, const int32_t* label, double voxel_size,
                          const float* priority, int n_sample, int propagate, void* workspace,
                          int64_t workspace_bytes, int32_t* point2voxel, int32_t* weak, int32_t* stats,
                          c3d_stream stream);

/* ------------------------------------------------------------------ SqueezeSegV3 SAC block
 * pc_processor/models/squeezesegv3_Proto.py:468-503 (SACBlock).  NHWC fp32.
 * c3d_sac_im2col7:  xcol[B,H,W,160], column c*49 + ky*7 + kx = xyz[p + (ky-3, kx-3)][c] (c < 3; zero
 *   outside the image, columns 147..159 zero): conv7x7(xyz) (:474-477) becomes a 160 -> 9C GEMM.
 * c3d_sac_modulate: m[p][j] = feat[p + tap(j % 9)][j / 9] * sigmoid(att[p][j]*scale[j] + shift[j]),
 *   j = c*9 + tap in F.unfold order (:495-497); att = raw conv output, (scale, shift) its BatchNorm.
 * c3d_sac_modulate_bwd: datt = dm * feat_tap * s * (1-s) (gradient at the BatchNorm output), and
 *   dm <- dm * s in place; c3d_sac_fold: dfeat[q][c] (+)= sum_k dm[q - tap(k)][c*9 + k].               */
int c3d_sac_im2col7(const float* xyz, int B, int H, int W, int xcs, float* xcol, c3d_stream stream);
int c3d_sac_modulate(const float* feat, const float* att, const float* scale, const float* shift,
                     int B, int H, int W, int C, float* m, c3d_stream stream);
int c3d_sac_modulate_bwd(float* dm, const float* feat, const float* att, const float* scale,
                         const float* shift, int B, int H, int W, int C, float* datt, c3d_stream stream);
int c3d_sac_fold(const float* t, int B, int H, int W, int C, int accumulate, float* dfeat,
                 c3d_stream stream);

/* ------------------------------------------------------------------ SyncBatchNorm exchange through peer memory
 * tasks/weak_segmentation/trainer.py:54 (torch.nn.SyncBatchNorm.convert_sync_batchnorm): the all-reduce of the per-channel
 * statistics inside SyncBatchNorm's forward (sum, sum of squares) and backward (sum dy, sum dy * x_hat) -- 43 + 43 vectors of
 * <= 704 x 2 fp64 per training step.  One process per GPU on ONE node: every rank owns a mailbox in its device memory, maps
 * the mailboxes of its peers (hipIpc handles, passed over the existing torch.distributed group) and each exchange is one
 * small kernel: write the vector into every peer's mailbox (xGMI stores), publish a sequence number, wait for the peers'
 * (bounded), sum the slots in rank order -- bit-identical on every rank (csrc/peer_ops.hip).  Capturable in a hipGraph (the
 * sequence counter lives in device memory).  Gradient buckets and the prototype bank stay on RCCL. */
#define C3D_PEER_MAX_RANKS 8
typedef struct {
  void* mailbox[C3D_PEER_MAX_RANKS];   /* mailbox[r]: rank r's mailbox as mapped in THIS process (mailbox[rank] = own allocation) */
  int32_t rank, world;
  int32_t cap_doubles;                 /* slot capacity the mailboxes were sized for (c3d_peer_mailbox_bytes)                  */
  float timeout_s;                     /* an exchange gives up waiting for a peer after this long (<= 0: 600 s, the order of a
                                        * process group's collective timeout -- what the all-reduce it replaces would wait)     */
  int32_t one_device;                  /* 1: every rank of the group runs on THIS device (or world == 1): the payload's
                                        * write-through stores + drained flag need no system-scope fences around them           */
  int32_t reserved;
} c3d_peer_desc;
int c3d_peer_desc_bytes(void);          /* sizeof(c3d_peer_desc), for a binding's layout check (host call) */
int64_t c3d_peer_mailbox_bytes(int cap_doubles);
/* host calls (synchronous): allocate + zero a mailbox and return its 64-byte IPC handle / map a peer's / unmap / free */
int c3d_peer_alloc(int64_t bytes, void** ptr, void* handle64);
int c3d_peer_open(const void* handle64, void** ptr);
int c3d_peer_close(void* ptr);
int c3d_peer_free(void* ptr);
/* buf[0 .. n) (fp64, device) <- sum over ranks, in place; every rank of the group must make the same sequence of calls.
 * A failed exchange (a peer never arrived within timeout_s, or an earlier exchange of this rank failed: the status word is
 * sticky) leaves NaN in buf -- the dist.all_reduce it replaces (torch.nn.SyncBatchNorm, trainer.py:54) would block or raise;
 * a kernel cannot raise, so it poisons: whatever is computed from the result cannot pass for a statistic. */
int c3d_peer_allreduce_f64(const c3d_peer_desc* d, double* buf, int n, c3d_stream stream);
/* host read of the rank's status word (0 = fine, 1 = an exchange timed out: its result was not a sum) and call counter */
int c3d_peer_status(const c3d_peer_desc* d, int32_t* status_out, int64_t* calls_out);
/* the same status word as a device float (0.f / 1.f) written to *dst on `stream`, no host synchronisation: the data-parallel
 * wrapper appends it to its last gradient bucket, so that every rank learns of ANY rank's failure from the all-reduce it
 * makes anyway (coarse3d_amd/dist.py; the reference's DistributedDataParallel + NCCL watchdog, trainer.py:55-60) */
int c3d_peer_status_to(const c3d_peer_desc* d, float* dst, c3d_stream stream);
/* SyncBatchNorm in ONE launch per layer and direction (torch.nn.SyncBatchNorm's forward / backward all-reduce,
 * tasks/weak_segmentation/trainer.py:54): fold the per-tile partials [C][2][n] (conv epilogue / c3d_bn_bwd_reduce), exchange the
 * fp64 sums through the mailboxes, finish -- c3d_stat_reduce(2) + c3d_peer_allreduce_f64 + c3d_bn_finalize / c3d_bn_bwd_coeffs of
 * the three-launch path, the same arithmetic in the same order (the same bits).  count = elements per channel over ALL ranks.
 * scratch: device doubles, 2 C (forward) / 4 C (backward), 2 C <= cap_doubles; ticket: one device word, zero before the first
 * call (the kernel leaves it zero).  Counts as one exchange in every rank's call sequence.  A failed exchange writes NaN into
 * every output vector (scale .. save_invstd / k1 .. dbeta; the running statistics are left alone). */
int c3d_peer_bn_finalize_partials(const c3d_peer_desc* d, const float* partial, int n, double count, const float* gamma,
                                  const float* beta, float* running_mean, float* running_var, float momentum, float eps, int C,
                                  float* scale, float* shift, float* save_mean, float* save_invstd, double* scratch,
                                  uint32_t* ticket, c3d_stream stream);
int c3d_peer_bn_bwd_coeffs_partials(const c3d_peer_desc* d, const float* partial, int n, double count, const float* mean,
                                    const float* invstd, const float* gamma, int C, float* k1, float* k2, float* k3,
                                    float* dgamma, float* dbeta, double* scratch, uint32_t* ticket, c3d_stream stream);

#ifdef __cplusplus
}
#endif
#endif
