"""Tensor-level wrappers over the C ABI (include/coarse3d_hip.h).

PyTorch is used for device memory and streams only; every function here launches the
hand-written HIP kernels through ctypes and raises if the library refuses the call."""
import ctypes as C

import torch

from . import _lib as L


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def to_nhwc(x):
    """[B,C,H,W] -> contiguous [B,H,W,C]."""
    return x.permute(0, 2, 3, 1).contiguous()


def from_nhwc(x):
    return x.permute(0, 3, 1, 2)


class Source:
    """One channel-concatenated conv input, transformed on load (c3d_src)."""
    __slots__ = ("t", "scale", "shift", "C", "coff", "lrelu")

    def __init__(self, t, scale=None, shift=None, C=None, coff=0, lrelu=False):
        assert t.dim() == 4 and t.is_contiguous() and t.dtype == torch.float32
        self.t, self.scale, self.shift = t, scale, shift
        self.C = t.shape[3] - coff if C is None else C
        self.coff, self.lrelu = coff, lrelu

    def fill(self, s):
        s.ptr = self.t.data_ptr()
        s.scale = self.scale.data_ptr() if self.scale is not None else None
        s.shift = self.shift.data_ptr() if self.shift is not None else None
        s.C, s.cstride, s.coff, s.lrelu = self.C, self.t.shape[3], self.coff, int(self.lrelu)


def conv_taps(kh, kw, dil, pad):
    """Input offsets (dy, dx) per tap, OIHW tap order (t = i*kw + j)."""
    return [(i * dil - pad, j * dil - pad) for i in range(kh) for j in range(kw)]


def negate_taps(taps):
    return [(-dy, -dx) for dy, dx in taps]


def pack_weights(w, mode=0, c_off=0, c_cnt=None, kpad=None):
    """OIHW weight -> MFMA operand layout [tap][K/4][N][4] (c3d_pack_weights)."""
    assert w.is_contiguous() and w.dtype == torch.float32
    cout, cin = w.shape[0], w.shape[1]
    t = w.shape[2] * w.shape[3]
    if c_cnt is None:
        c_cnt = cin - c_off
    k = c_cnt if mode == 0 else cout
    n = cout if mode == 0 else c_cnt
    if kpad is None:
        kpad = (k + 15) // 16 * 16
    dst = torch.empty(t * (kpad // 4) * n * 4, device=w.device, dtype=torch.float32)
    L.check(L.lib().c3d_pack_weights(_p(w), _p(dst), cout, cin, t, mode, c_off, c_cnt, kpad, _stream()),
            "c3d_pack_weights")
    return dst


def num_mtiles(b, h, w):
    return L.lib().c3d_conv_num_mtiles(b, h, w)


def conv_forward(srcs, wpack, bias, cout, taps, lrelu=False, stats=False, out=None, out_coff=0,
                 accumulate=False, stat_partial=None):
    """y = [LeakyReLU](conv(cat(transformed srcs)) + bias); optional per-tile channel stats."""
    d = L.ConvDesc()
    d.nsrc = len(srcs)
    for i, s in enumerate(srcs):
        s.fill(d.src[i])
    b, h, w = srcs[0].t.shape[:3]
    d.B, d.H, d.W, d.Cout = b, h, w, cout
    d.ntaps = len(taps)
    for i, (dy, dx) in enumerate(taps):
        d.tap_dy[i], d.tap_dx[i] = dy, dx
    d.wpack = wpack.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.epi_lrelu = int(lrelu)
    if out is None:
        out = torch.empty(b, h, w, cout, device=wpack.device, dtype=torch.float32)
    d.out, d.out_cstride, d.out_coff, d.accumulate = out.data_ptr(), out.shape[3], out_coff, int(accumulate)
    if stats and stat_partial is None:
        stat_partial = torch.empty(num_mtiles(b, h, w), cout, 2, device=wpack.device, dtype=torch.float32)
    d.stat_partial = stat_partial.data_ptr() if stat_partial is not None else None
    L.check(L.lib().c3d_conv_forward(C.byref(d), _stream()), "c3d_conv_forward")
    return out, stat_partial


def conv_wgrad(src, dz, dw, taps, cin_off=0, accumulate=False):
    """dw[:, cin_off:cin_off+src.C] (+)= sum_p dz[p] (x) transformed src[p + tap]."""
    d = L.WgradDesc()
    src.fill(d.x)
    b, h, w = src.t.shape[:3]
    d.dz, d.dz_cstride = dz.data_ptr(), dz.shape[3]
    d.B, d.H, d.W, d.Cout = b, h, w, dw.shape[0]
    d.ntaps = len(taps)
    for i, (dy, dx) in enumerate(taps):
        d.tap_dy[i], d.tap_dx[i] = dy, dx
    d.Cin_total, d.cin_off = dw.shape[1], cin_off
    d.dw, d.accumulate = dw.data_ptr(), int(accumulate)
    n = L.lib().c3d_wgrad_partial_floats(C.byref(d))
    part = torch.empty(n, device=dz.device, dtype=torch.float32)
    d.partial = part.data_ptr()
    L.check(L.lib().c3d_conv_wgrad(C.byref(d), _stream()), "c3d_conv_wgrad")
    return dw
