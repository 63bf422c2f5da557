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
/* number of workgroups conv kernels will use for an [B,H,W] image: size of stat partial bufs */
int c3d_conv_num_mtiles(int B, int H, int W);

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
} c3d_src;

typedef struct {
  c3d_src src[3];
  int32_t nsrc;
  int32_t B, H, W;
  int32_t Cout;
  int32_t ntaps;            /* 1, 4 or 9                                                 */
  int32_t tap_dy[9];        /* input offset of tap t relative to the output pixel        */
  int32_t tap_dx[9];
  const float* wpack;       /* packed by c3d_pack_weights: [tap][K/4][Cout][4]           */
  const float* bias;        /* [Cout] or NULL                                            */
  int32_t epi_lrelu;        /* LeakyReLU(0.01) after bias                                */
  float* out;               /* NHWC, channel stride out_cstride, first channel out_coff  */
  int32_t out_cstride;
  int32_t out_coff;
  int32_t accumulate;       /* out += result (gradient accumulation)                     */
  float* stat_partial;      /* NULL or [c3d_conv_num_mtiles][Cout][2]: per-tile sum,sumsq */
} c3d_conv_desc;

/* y = epilogue(conv(transform(cat(src)))) as an implicit GEMM on fp32 MFMA.
 * Replaces nn.Conv2d (+LeakyReLU, + the batch statistics of the following BatchNorm2d) in
 * ResContextBlock / ResBlock / UpBlock / cls_head / ProjectionV1
 * (pc_processor/models/salsanext_proto.py:41-62,82-132,164-208,318; projector.py:18-23),
 * and, with transposed weights, their input gradients.                                      */
int c3d_conv_forward(const c3d_conv_desc* d, c3d_stream stream);

/* Weight repack from the reference's OIHW layout [Cout][Cin][T] (T = kh*kw):
 *   mode 0 (forward):  dst[t][k/4][n][k%4] = W[n][k0 + k][t],  k < K=Cin_cnt,  n < Cout
 *   mode 1 (dgrad):    dst[t][k/4][n][k%4] = W[k][n0 + n][t],  k < Cout,       n < N=Cin_cnt
 * K is zero-padded to Kpad (multiple of 16).                                                */
int c3d_pack_weights(const float* w_oihw, float* dst, int Cout, int Cin, int T, int mode,
                     int c_off, int c_cnt, int Kpad, c3d_stream stream);

/* dW[cout][cin_off + cin][t] (OIHW, full Cin_total) = sum_pixels dz[p][cout] * x[p + tap t][cin]
 * and db[cout] = sum dz.  x is given as ONE transformed source (call once per source).
 * Replaces autograd's conv weight gradient for the layers listed above.
 * `partial` is scratch of c3d_wgrad_partial_floats() floats.                                 */
typedef struct {
  c3d_src x;                /* input of the forward conv (same transform as forward)      */
  const float* dz;          /* NHWC [B,H,W,Cout] gradient w.r.t. the conv output (pre-act) */
  int32_t dz_cstride;
  int32_t B, H, W, Cout;
  int32_t ntaps;
  int32_t tap_dy[9];
  int32_t tap_dx[9];
  int32_t Cin_total;        /* Cin of the OIHW weight tensor                              */
  int32_t cin_off;          /* where this source's channels start inside Cin_total        */
  float* dw;                /* OIHW gradient [Cout][Cin_total][T]                         */
  int32_t accumulate;
  float* partial;
} c3d_wgrad_desc;
int64_t c3d_wgrad_partial_floats(const c3d_wgrad_desc* d);
int c3d_conv_wgrad(const c3d_wgrad_desc* d, c3d_stream stream);

#ifdef __cplusplus
}
#endif
#endif
