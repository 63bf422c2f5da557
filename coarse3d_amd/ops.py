"""Tensor-level wrappers over the C ABI (include/coarse3d_hip.h).

PyTorch is used for device memory and streams only; every function here launches the
hand-written HIP kernels through ctypes and raises if the library refuses the call."""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib as L


# bench.py sets this to a list to bracket every MFMA-engine launch with HIP events on the launch
# stream: entries are (kernel instance name, algorithmic FLOPs, start event, end event).
KERNEL_EVENTS = None
KERNEL_EVENT_FILTER = None       # None = every MFMA launch; else only launches of this kernel instance
# Matrix-pipe mode of the MFMA engines (c3d_conv_desc.mfma_bf16):
#   2 "bf16x3" (the library default since round 3 -- what bench.py measures is what an importer gets)
#              fp32 operands split exactly into three bf16 planes, six or eight of the nine plane products per
#              step on the bf16 matrix pipe, fp32 accumulate: fp32-class results (csrc/conv_x3.hip, conv_pw3.hip,
#              conv_bfp.hip, wgrad_tr.hip); every `pytest -m gpu` test runs on it
#   0 "f32"    v_mfma_f32_32x32x2_f32 everywhere (== an fmaf chain): the strict-IEEE engine; the whole GPU suite is
#              re-run on it by tests/test_gpu_configs.py::test_parity_suite_passes_on_the_strict_fp32_engine
#   1 "bf16"   operands rounded to bf16 inside the kernels, fp32 accumulate; activations stored as bf16
#              (STORAGE_BF16) or fp32 -- opt-in mixed precision, BASELINE configs[2]
_MODES = {"f32": 0, "bf16": 1, "bf16x3": 2}
DEFAULT_MATRIX = "bf16x3"
# C3D_MATRIX=f32 python -m pytest tests -m gpu   runs the WHOLE parity suite on the strict fp32-MFMA engine
MFMA_MODE = _MODES[__import__("os").environ.get("C3D_MATRIX", DEFAULT_MATRIX)]


# Activation storage of the "bf16" mode (BASELINE configs[2]): True = the backbone's activations and their
# gradients live in HBM as bf16 (fp32 arithmetic, statistics, master weights, probabilities and embeddings);
# False = fp32 tensors, bf16 MFMA operands only (round 1).  Tensors carry their own dtype: every wrapper below
# derives the library's bf16 flags from them, so the two layouts can meet (e.g. fp32 loss gradients entering a
# bf16 backbone).
STORAGE_BF16 = MFMA_MODE == 1


# bf16x3 engine: forward 3x3 / 2x2 convolutions run six instead of eight plane products when their output has at
# least this many pixels (B*H*W).  With six, each product is off by up to 2^-23 |a||b|; what the following
# BatchNorm makes of that depends on its population: measured (tools/noise_probe.py, gradient error against the
# float64 oracle relative to the fp32 oracle's own) 4.6x at 1 344 pixels (the 1x24x56 POSS case of
# tests/test_gpu_backbone.py: eight products 0.3x), but 0.79 vs 0.84 at 65 536 (2x64x512), 1.15 vs 1.19 at
# 144 640 (4x40x904), 2.2 vs 2.6 at 36 480 (2x40x456; the fp32 engine: 2.4) -- indistinguishable from eight
# and from the fp32-MFMA engine.  tests/test_gpu_backbone.py::test_gradient_noise_at_a_realistic_population holds it.
SIX_FWD_MIN_PIXELS = 32768


def set_matrix_precision(kind, storage=None):
    """kind: "f32" | "bf16" | "bf16x3".  storage ("bf16" | "f32", "bf16" mode only; default "bf16")."""
    global MFMA_MODE, STORAGE_BF16
    if kind not in _MODES:
        raise ValueError(f"matrix precision must be one of {sorted(_MODES)}, got {kind!r}")
    if storage not in (None, "f32", "bf16") or (storage == "bf16" and kind != "bf16"):
        raise ValueError("bf16 activation storage exists in the 'bf16' matrix mode only")
    MFMA_MODE = _MODES[kind]
    STORAGE_BF16 = kind == "bf16" and storage != "f32"


def matrix_precision_state():
    """(kind, storage) to hand back to ``set_matrix_precision`` -- for code that switches engines temporarily."""
    kind = {v: k for k, v in _MODES.items()}[MFMA_MODE]
    return kind, (("bf16" if STORAGE_BF16 else "f32") if kind == "bf16" else None)


def act_dtype():
    """dtype of the activations a fresh backbone pass should produce."""
    return torch.bfloat16 if STORAGE_BF16 else torch.float32


def _bf(*tensors):
    """bf16_mask of the C ABI: bit i set <=> the i-th activation tensor is bf16 (None counts as fp32)."""
    m = 0
    for i, t in enumerate(tensors):
        if t is not None and t.dtype == torch.bfloat16:
            m |= 1 << i
    return m


# Schedule selector of the bit-identity tests (c3d_conv_desc.variant): 0 = the library's choice; bits 0-1 for 1x1 convs
# with Cout > 64 on the bf16x3 engine (1 = fused kernel with eight waves, 2 = with four, 3 = round 2's phased kernel);
# bit 2 = round 2's phased kernel for nine-tap convs.  Same bits out of every variant (tests/test_gpu_conv.py).
# Round 6: 32 = conv_bfp's staged tile instead of the streaming 32-channel pointwise kernel (csrc/conv_pws.hip), 64 = no LDS-DMA
# prefetch of the bf16 engine's multiplier tile, 128 = the general instance of conv_x3f for input-gradient launches (instead of the
# transform-free one); a pack made by pack_weights_wino adds 16 by itself (Winograd variant, csrc/conv_wino.hip).
CONV_VARIANT = 0
# c3d_wgrad_desc.variant (fused weight-gradient launches): 0 = the library's choice, 1 = whole-window register sets, 2 = lean
# ones (same bits), +4 = a fused 1x1 launch keeps the unfused tile configuration, +128 = four producer waves in every
# instance (the three-plane 1x1 instances with small accumulators run eight)
WGRAD_VARIANT = int(os.environ.get("C3D_WGRAD_VARIANT", "0"))
F16X2_FWD = os.environ.get("C3D_F16X2_FWD", "0") == "1"
F16X2_BWD = os.environ.get("C3D_F16X2_BWD", "0") == "1"     # EXPERIMENT: multi-tap input gradients too (per-tensor exponent)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_dev = torch.cuda.current_device


def _stream():
    """The current HIP stream of the current device as the C ABI wants it.  (torch.cuda.current_stream() builds a
    Stream object through four Python layers: 9 us per call, ~450 calls per training step = 4 ms of the step's 9.5 ms
    of host work; the raw query is 0.3 us.)"""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(_cur_dev()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Timed:
    def __init__(self, name, flops, detail=None):
        self.name, self.flops, self.detail = name, flops, detail

    def __enter__(self):
        self.on = KERNEL_EVENTS is not None and (KERNEL_EVENT_FILTER is None or KERNEL_EVENT_FILTER == self.name)
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *a):
        if self.on:
            self.e1.record()
            KERNEL_EVENTS.append((self.name, self.flops, self.e0, self.e1, self.detail))


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
        assert t.dim() == 4 and t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16)
        self.t, self.scale, self.shift = t, scale, shift
        self.C = t.shape[3] - coff if C is None else C
        self.coff, self.lrelu = coff, lrelu

    def fill(self, s):
        s.ptr = self.t.data_ptr()
        s.scale = self.scale.data_ptr() if self.scale is not None else None
        s.shift = self.shift.data_ptr() if self.shift is not None else None
        s.C, s.cstride, s.coff, s.lrelu = self.C, self.t.shape[3], self.coff, int(self.lrelu)
        s.bf16 = int(self.t.dtype == torch.bfloat16)


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
    numel = t * (kpad // 4) * n * 4
    # bf16-pipe engines: the pre-split bf16 planes follow the fp32 image for the kernels that read them
    # (bf16x3 multi-tap convs: csrc/conv_x3.hip; 1x1 convs with more than 64 outputs in both bf16 modes:
    # csrc/conv_pw3.hip -- the "bf16" mode reads the first plane only, which is the RNE-rounded weight)
    planes = (MFMA_MODE == 2 and t > 1) or (MFMA_MODE == 1 and t == 9) or (MFMA_MODE != 0 and t == 1 and n > 64)
    dst = torch.empty(numel * 5 // 2 if planes else numel, device=w.device, dtype=torch.float32)
    L.check(L.lib().c3d_pack_weights(_p(w), _p(dst), cout, cin, t, mode | (2 if planes else 0), c_off, c_cnt, kpad,
                                     _stream()), "c3d_pack_weights")
    dst.c3d_planes = planes
    return dst


def pack_weights_wino(wpack, kpad, n, taps):
    """Winograd F(2x2, 3x3) weights U = G g G^T (three bf16 planes) from the fp32 image of a ``pack_weights`` pack of a
    nine-tap conv (either mode) and the launch's tap offsets -- csrc/conv_wino.hip, c3d_pack_weights_wino."""
    assert len(taps) == 9
    dil = max(max(abs(dy), abs(dx)) for dy, dx in taps)
    nbytes = L.lib().c3d_wino_pack_bytes(kpad, n)
    dst = torch.empty(nbytes // 2, device=wpack.device, dtype=torch.int16)
    dy = (C.c_int32 * 9)(*[t[0] for t in taps])
    dx = (C.c_int32 * 9)(*[t[1] for t in taps])
    L.check(L.lib().c3d_pack_weights_wino(_p(wpack), kpad, n, dy, dx, dil, _p(dst), _stream()), "c3d_pack_weights_wino")
    dst.c3d_wino = dil
    return dst


def wino_num_tiles(b, h, w, dil):
    return L.lib().c3d_conv_wino_num_tiles(b, h, w, dil)


class PackCache:
    """All weight repacks of a model in ONE launch per step.

    The first step packs layer by layer (and records which (weight, mode, slice) repacks the
    forward/backward plan asks for); from then on ``refresh()`` repacks every recorded entry with
    a single batched launch at the start of the forward and ``get()`` just returns the buffer.
    The table is rebuilt whenever a parameter's storage moves."""

    def __init__(self):
        self.entries = {}        # key -> (weight tensor, mode, c_off, c_cnt, kpad, dst)
        self.table = None
        self.ptrs = None
        self.fresh = False
        self.generation = 0      # bumped whenever the device table is (re)built: a captured hipGraph that
                                 # baked the old table's address must not be replayed any more

    def get(self, w, mode=0, c_off=0, c_cnt=None, kpad=None):
        key = (w.data_ptr(), tuple(w.shape), mode, c_off, c_cnt, kpad, MFMA_MODE)   # parameter storage is stable across steps
        ent = self.entries.get(key)
        if ent is not None and self.fresh:
            return ent[5]
        dst = pack_weights(w, mode, c_off, c_cnt, kpad)
        cc = (w.shape[1] - c_off) if c_cnt is None else c_cnt
        k = cc if mode == 0 else w.shape[0]
        self.entries[key] = (w, mode, c_off, cc, (k + 15) // 16 * 16 if kpad is None else kpad, dst)
        self.table = None
        return dst

    def refresh(self):
        """Repack every known entry from the current weights (one launch)."""
        if not self.entries:
            self.fresh = False
            return
        ptrs = tuple(e[0].data_ptr() for e in self.entries.values())
        if self.table is None or ptrs != self.ptrs:
            arr = (L.PackEntry * len(self.entries))()
            for i, (w, mode, c_off, cc, kpad, dst) in enumerate(self.entries.values()):
                arr[i].src, arr[i].dst = w.data_ptr(), dst.data_ptr()
                arr[i].Cout, arr[i].Cin, arr[i].T = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
                arr[i].mode, arr[i].c_off, arr[i].c_cnt, arr[i].Kpad = mode | (2 if getattr(dst, "c3d_planes", False) else 0), c_off, cc, kpad
            raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            dev = next(iter(self.entries.values()))[5].device
            self.table = raw.to(dev)
            self.ptrs = ptrs
            self.generation += 1
        _call("c3d_pack_weights_batch", _dp(self.table), len(self.entries), _stream())
        self.fresh = True


def _tile_rows(h):
    """c3d_tile_rows() in csrc/conv_mfma.hip (for the kernel-name mirror of the event timer only)."""
    if ((h + 7) // 8 * 8) * 3 <= h * 4:
        return 8
    return 2 if (h + 1) // 2 * 2 < (h + 3) // 4 * 4 else 4


def num_mtiles(b, h, w):
    return L.lib().c3d_conv_num_mtiles(b, h, w)


def _wide_cout_tiles(b, h, w, cout, tr):
    """64-cout tiles, or 32-cout ones when the 64-wide grid would cover too few CUs: c3d_wide_cout_tiles() in
    csrc/conv_common.h."""
    if cout <= 32:
        return False
    min_wg = 192
    return b * ((w + 31) // 32) * ((h + tr - 1) // tr) * ((cout + 63) // 64) >= min_wg


def _pw3_tile(b, h, w, cout):
    """Cout sub-tiles (8 = 256 couts, 4 = 128) of the wide pointwise kernel for this launch, 0 = the grid would leave
    most CUs idle and conv_bfp's 64-wide workgroups run instead.  Mirrors c3d_conv_forward() in csrc/conv_mfma.hip."""
    fill = int(os.environ.get("C3D_PW3_FILL", "128"))
    px_tiles = b * ((w + 31) // 32) * ((h + 7) // 8)
    wide = cout > 128
    if wide and px_tiles * ((cout + 255) // 256) < fill:
        wide = False
    if wide or px_tiles * ((cout + 127) // 128) >= fill:
        return 8 if wide else 4
    return 0


def _pw3_kernel_name(nt, k, cout, bf16_srcs=False):
    """Kernel instance c3d_conv_forward_pw3() launches (csrc/conv_pw3.hip): bf16x3 runs the fused kernel
    conv_pw3f_kernel<NT, WN> -- WN = 2: eight waves, WN = 1: four waves x 128 couts, two workgroups per CU
    (short K, couts a multiple of 128) -- the "bf16" mode conv_pw1_kernel<NT> over bf16 tensors (four chunks in flight),
    else round 2's conv_pw3_kernel<NT, 1>."""
    if MFMA_MODE != 2:
        return f"conv_pw1_kernel<{nt}, false>" if (bf16_srcs and (CONV_VARIANT & 3) != 3) else f"conv_pw3_kernel<{nt}, 1>"
    mode = CONV_VARIANT & 3
    if mode == 3:
        return f"conv_pw3_kernel<{nt}, 3>"
    if mode == 2 or (mode != 1 and k <= 256 and cout % 128 == 0):
        return "conv_pw3f_kernel<4, 1>"
    return f"conv_pw3f_kernel<{nt}, 2>"


def conv_forward(srcs, wpack, bias, cout, taps, lrelu=False, stats=False, out=None, out_coff=0,
                 accumulate=False, stat_partial=None, slope=0.0, grad=False, stat_mul=None, f16x2_inv=None,
                 stat_mul_optional=False):
    """y = [LeakyReLU](conv(cat(transformed srcs)) + bias); optional per-tile channel stats.
    ``grad``: this launch is an input-gradient convolution (transposed weights, negated taps): the
    bf16x3 engine then accumulates six plane products instead of eight (c3d_conv_desc.mfma_bf16 = 3)."""
    d = L.ConvDesc()
    d.nsrc = len(srcs)
    for i, s in enumerate(srcs):
        s.fill(d.src[i])
    b, h, w = srcs[0].t.shape[:3]
    d.B, d.H, d.W, d.Cout = b, h, w, cout
    d.ntaps = len(taps)
    for i, (dy, dx) in enumerate(taps):
        d.tap_dy[i], d.tap_dx[i] = dy, dx
    if MFMA_MODE == 2 and len(taps) > 1 and not getattr(wpack, "c3d_planes", False) and not getattr(wpack, "c3d_wino", 0):
        raise RuntimeError("bf16x3 mode needs weight packs made after ops.set_matrix_precision('bf16x3')")
    wino = getattr(wpack, "c3d_wino", 0)          # a pack_weights_wino pack: Winograd F(2x2, 3x3), csrc/conv_wino.hip
    d.wpack = wpack.data_ptr()
    d.wpack_planes = int(bool(getattr(wpack, "c3d_planes", False)) or bool(wino))
    d.bias = bias.data_ptr() if bias is not None else None
    d.epi_lrelu = int(lrelu)
    d.lrelu_slope = slope            # 0 = the SalsaNext default 0.01
    if out is None:      # the output lives in the layout of its (first) input: a bf16 chain stays bf16
        out = torch.empty(b, h, w, cout, device=wpack.device, dtype=srcs[0].t.dtype)
    d.out, d.out_cstride, d.out_coff, d.accumulate = out.data_ptr(), out.shape[3], out_coff, int(accumulate)
    d.out_bf16 = int(out.dtype == torch.bfloat16)
    d.variant = CONV_VARIANT | (16 if wino else 0)
    if stats and stat_partial is None:
        stat_partial = torch.empty(cout, 2, wino_num_tiles(b, h, w, wino) if wino else num_mtiles(b, h, w), device=wpack.device,
                                   dtype=torch.float32)
    d.stat_partial = stat_partial.data_ptr() if stat_partial is not None else None
    use_mul = stat_mul is not None
    if use_mul:      # (sum v, sum v * stat_mul) instead of (sum v, sum v^2): BatchNorm-backward sums in the epilogue
        if stat_partial is None or tuple(stat_mul.shape[:3]) != (b, h, w) or stat_mul.dtype != out.dtype:
            raise ValueError("conv_forward: stat_mul needs statistics and a tensor of the output's shape and dtype")
        d.stat_mul, d.stat_mul_cstride = stat_mul.data_ptr(), stat_mul.shape[3]
        d.stat_mul_bf16 = int(stat_mul.dtype == torch.bfloat16)
    tr = _tile_rows(h)
    nt_ = len(taps)
    timed = KERNEL_EVENTS is not None        # the kernel-name mirror below only serves the per-kernel event timers

    def kernel_name():
        return _conv_kernel_name(b, h, w, [s.C for s in srcs], [s.t.dtype == torch.bfloat16 for s in srcs], cout, taps, grad,
                                 bool(d.wpack_planes), bool(d.stat_mul), has_stats=bool(d.stat_partial), accumulate=bool(accumulate),
                                 out_room=(out.shape[3] - out_coff) if (out.shape[3] % 4 == 0 and out_coff % 4 == 0) else 0,
                                 plain=all(s_.scale is None and not s_.lrelu for s_ in srcs))
    # six plane products: every input gradient, and forward multi-tap convs whose BatchNorm population is large
    # (SIX_FWD_MIN_PIXELS).  conv_pw3 runs six in every launch (the flag is ignored there).
    six = MFMA_MODE == 2 and (grad or (nt_ > 1 and tr == 8 and b * h * w >= SIX_FWD_MIN_PIXELS))
    d.mfma_bf16 = 3 if six else MFMA_MODE
    if f16x2_inv is not None:
        # EXPERIMENT (C3D_F16X2_BWD=1): an input gradient on two fp16 planes; the source carries 2^s in its scale array
        # (grad_exponent), this device scalar 2^-s goes to the epilogue
        d.mfma_bf16 = 4
        d.acc_scale_dev = f16x2_inv.data_ptr()
    elif F16X2_FWD and MFMA_MODE == 2 and not grad and b * h * w >= SIX_FWD_MIN_PIXELS:
        # EXPERIMENT (C3D_F16X2_FWD=1; DESIGN.md round-4 list): forward convs over large BatchNorm populations -- where the
        # exact split already runs six products -- on two fp16 planes / three products, generic kernel
        d.mfma_bf16 = 4
    _sup = getattr(L.lib(), "c3d_conv_stat_mul_supported", None)      # (absent from an older build loaded through C3D_LIB)
    if use_mul and not (_sup(C.byref(d)) if _sup is not None else (MFMA_MODE == 2 and not d.out_bf16)):
        # the kernel this launch selects has no such epilogue (bf16x3 engine: fp32 tensors; bf16 engine: 8-row tiles over bf16
        # tensors): the caller runs the separate c3d_bn_bwd_reduce pass
        if not stat_mul_optional:
            raise ValueError("conv_forward: this launch's kernel has no BatchNorm-backward epilogue (c3d_conv_stat_mul_supported)")
        d.stat_mul, d.stat_mul_bf16, d.stat_partial, stat_partial = None, 0, None, None
    if not timed:
        L.check(L.lib().c3d_conv_forward(C.byref(d), _stream()), "c3d_conv_forward")
        return out, stat_partial
    name, halo = kernel_name()
    with _Timed(name, 2.0 * b * h * w * cout * len(taps) * sum(s.C for s in srcs),
                (h, w, sum(s.C for s in srcs), cout, nt_, halo, int(accumulate))):
        L.check(L.lib().c3d_conv_forward(C.byref(d), _stream()), "c3d_conv_forward")
    return out, stat_partial


def _conv_kernel_name(b, h, w, src_c, src_bf16, cout, taps, grad, wpack_planes, stat_mul, has_stats=False, accumulate=False, out_room=None, plain=False):
    """The kernel instance c3d_conv_forward launches for this problem, as rocprofv3 prints it -- a mirror of the dispatch in
    csrc/conv_mfma.hip, conv_x3.hip, conv_pw3.hip and conv_bfp.hip for the per-kernel event timers of bench.py
    (tests/test_cpu_kernel_names.py holds every name it can produce against the symbols of the built library)."""
    tr = _tile_rows(h)
    nt_ = len(taps)
    halo = max(max(abs(dy), abs(dx)) for dy, dx in taps)
    sm_ = "true" if stat_mul else "false"      # (bf16 engine: the instance with the BatchNorm-backward epilogue)
    hh = 0 if nt_ == 1 else (1 if (nt_ == 4 or halo <= 1) else 2)
    k32 = tr == 8 and nt_ == 1 and all(c % 32 == 0 for c in src_c)
    if MFMA_MODE == 2 and tr == 8 and nt_ > 1 and not (nt_ in (3, 6) and (CONV_VARIANT & 4)):      # mirrors c3d_conv_forward_x3() in csrc/conv_x3.hip
        # nine taps: the fused kernel (round 3); four taps: fused too since round 5; CONV_VARIANT & 4 forces the phased one
        fused_ = not (CONV_VARIANT & 4)      # (four taps: fused since round 5; three / six: fused only)
        # (names as rocprofv3 prints them: the fused kernel carries its plane count and its bf16-source flag as fifth and
        #  sixth template arguments)
        six_ = grad or b * h * w >= SIX_FWD_MIN_PIXELS          # (the fourth template argument: six plane products)
        # (eighth argument, round 6: the transform-free instance -- six products, no source with an affine or a LeakyReLU on load:
        #  the input-gradient launches; CONV_VARIANT & 128 keeps the general one)
        plain_ = bool(fused_ and six_ and plain and not (CONV_VARIANT & 128))
        name = (f"conv_x3{'f' if fused_ else ''}_kernel<{2 if _wide_cout_tiles(b, h, w, cout, tr) else 1}, {hh}, {nt_}, "
                f"{'true' if six_ else 'false'}{(', 3, false, false, ' + ('true' if plain_ else 'false')) if fused_ else ''}>")
    elif (MFMA_MODE == 1 and tr == 8 and nt_ == 9 and wpack_planes and not (CONV_VARIANT & 4)
          and all(src_bf16)):      # the fused nine-tap kernel with one plane (csrc/conv_x3.hip)
        name = f"conv_x3f_kernel<{2 if _wide_cout_tiles(b, h, w, cout, tr) else 1}, {hh}, 9, true, 1, true, {sm_}, false>"
    elif MFMA_MODE and tr == 8 and nt_ == 1 and cout > 64 and wpack_planes and _pw3_tile(b, h, w, cout):     # csrc/conv_pw3.hip
        name = _pw3_kernel_name(_pw3_tile(b, h, w, cout), sum(src_c), cout,
                                bf16_srcs=all(src_bf16))
    elif (MFMA_MODE == 2 and tr == 8 and nt_ == 1 and cout <= 32 and not has_stats and not (CONV_VARIANT & 32)
          and (grad or b * h * w >= SIX_FWD_MIN_PIXELS) and not any(src_bf16) and sum(src_c) == 32
          and not (cout % 4 and accumulate) and (out_room is None or (cout + 3) // 4 * 4 <= out_room)):
        # 32-channel 1x1 convs without statistics (and the class head) with six plane products: the streaming kernel
        # (csrc/conv_pws.hip, round 6; c3d_conv_pws_takes)
        name = "conv_pws_kernel<1, 2, false>"
    elif MFMA_MODE:     # mirrors dispatch_bfp() in csrc/conv_bfp.hip (the trailing template argument: raw bf16 staging)
        np_ = 3 if MFMA_MODE == 2 else 1
        wide_ = _wide_cout_tiles(b, h, w, cout, tr)
        bfs_ = (np_ == 1 and tr == 8 and nt_ in (1, 4) and not (CONV_VARIANT & 8) and all(src_bf16))
        if bfs_:        # launch_bfp_bf16_sources(): deeper K chunks where every source's width allows
            c32_, c64_ = all(c % 32 == 0 for c in src_c), all(c % 64 == 0 for c in src_c)
            ck_ = (64 if c64_ else 32 if c32_ else 16) if nt_ == 1 else (32 if c32_ else 16)
            name = f"conv_bfp_kernel<8, {2 if wide_ else 1}, {ck_}, {hh}, {nt_}, 1, true, {sm_}>"
        else:
            name = (f"conv_bfp_kernel<8, {2 if wide_ else 1}, 32, 0, 1, {np_}, false, false>" if k32 else
                    f"conv_bfp_kernel<{tr}, {2 if (wide_ or tr == 2) else 1}, 16, {hh}, {nt_}, {np_}, false, false>")
    elif k32:         # mirrors c3d_conv_forward / launch_taps() in csrc/conv_mfma.hip
        wide = cout > 64 and (cout + 127) // 128 * 128 <= (cout + 63) // 64 * 64
        name = f"conv_mfma_kernel<8, {4 if wide else (2 if cout > 32 else 1)}, 32, 0, 1>"
    else:
        name = f"conv_mfma_kernel<{tr}, {2 if (cout > 32 or tr == 2) else 1}, 16, {hh}, {nt_}>"
    return name, halo


def _wgrad_kernel_name(ci, co, nt, halo, fused=False, raw=False, h=0, pre=False):
    """Mirrors c3d_wgrad_cfg() (csrc/wgrad_common.h), plan() (wgrad_mfma.hip) and the launch tables of wgrad_mfma.hip /
    wgrad_tr.hip (names as rocprofv3 prints them; the eleventh template argument: lean register sets, the twelfth and thirteenth: producer
    and consumer waves, round 5)."""
    tr = MFMA_MODE != 0
    hl = 1 if halo <= 1 else 2
    x3 = tr and MFMA_MODE == 2          # three planes: the smaller pixel tiles of c3d_wgrad_cfg
    lean_ok = False
    if nt == 1:
        wide = ci >= 96 and co >= 192 and not (fused and x3)      # (a fused launch takes the 128 x 128 slice)
        cfg = ("1, 2, 4, 2, 2, 1, 0" if wide else "1, 2, 2, 2, 2, 1, 0" if (ci >= 96 and co >= 96)
               else f"1, 2, 2, 1, 1, {2 if tr else 4}, 0" if co > 32 else "1, 1, 1, 1, 1, 4, 0")
        trw = 1
    elif nt in (3, 4):        # (three taps run in the four-tap instances, six in the nine-tap ones)
        if x3 and co > 32 and ci % 64 == 0 and halo <= 1:
            cfg, trw, lean_ok = f"4, 1, 2, 2, 1, 2, {hl}", 2, True
        elif co > 32:
            cfg, trw = f"4, 1, 2, 1, 1, {2 if x3 else 4}, {hl}", (2 if x3 else 4)
        else:
            cfg, trw, lean_ok = f"4, 1, 1, 1, 1, 4, {hl}", 4, hl == 2
    elif co > 32:
        trw = 4 if (tr and not x3) else 2
        cfg = f"9, 1, 1, 1, 2, {trw}, {hl}"
    else:
        cfg, trw, lean_ok = f"9, 1, 1, 1, 1, 4, {hl}", 4, hl == 2
    if tr:       # (ninth template argument: BatchNorm backward applied on load, conv_wgrad(fuse=...); tenth: raw bf16 stages, four
        # tiles in flight -- the bf16 engine with bf16 tensors on both sides; eleventh: lean register sets)
        lean = bool(fused and x3 and lean_ok and (h + trw - 1) // trw >= 8)
        # c3d_wgrad_producer_waves(): eight producer waves in the small 1x1 instances; nine-tap launches: eight + eight
        # consumer waves with the taps split (launch_tr_id in csrc/wgrad_tr.hip) -- round 6: the fused ones of the three-plane
        # engine too, their producer waves split by tensor (not with a pre-activation affine, not with variant & 256); the
        # 32-cout instance with a two-pixel halo always in its lean form there
        roles = fused and x3 and nt == 9 and not pre and not (WGRAD_VARIANT & 256)
        ncw = 8 if (nt == 9 and (not fused or roles) and not (WGRAD_VARIANT & 128) and (x3 or co <= 32)) else 4
        if roles and ncw == 8:
            lean = lean_ok
        # ... and the fused four-tap launch over 64 x 64 slices (taps split 2 + 2; exactly four taps, one-pixel halo)
        if x3 and nt == 1 and cfg.startswith("1, 2, 4") and not fused and not (WGRAD_VARIANT & (256 | 128)):
            ncw = 8                     # (the 128 x 256 slice: cout tiles split across the consumer halves)
        if (x3 and nt == 4 and cfg.startswith("4, 1, 2, 2, 1, 2, 1") and not pre and not (WGRAD_VARIANT & (256 | 128))):
            ncw, lean = 8, fused        # (unfused launches too)
        npw = 8 if (ncw == 8 or (nt == 1 and not cfg.startswith("1, 2, 4") and not (WGRAD_VARIANT & 128))) else 4
        return (f"wgrad_tr_kernel<{3 if MFMA_MODE == 2 else 1}, {cfg}, {'true' if fused else 'false'}, "
                f"{'true' if (raw and MFMA_MODE == 1 and not fused) else 'false'}, {'true' if lean else 'false'}, {npw}, {ncw}>")
    return f"wgrad_mfma_kernel<{cfg}, {'true' if MFMA_MODE == 1 else 'false'}>"


def conv_wgrad(src, dz, dw, taps, cin_off=0, accumulate=False, slope=0.0, bias_partial=None, dbias=None, f16x2=None, fuse=None):
    """dw[:, cin_off:cin_off+src.C] (+)= sum_p dz[p] (x) transformed src[p + tap].
    bias_partial [>=Cout, 2, n] + dbias [Cout]: the launch that folds the weight-gradient strips folds the layer's
    bias-gradient partials too (instead of a separate bias_from_partials launch).
    fuse = (dy, act, k[, (pre_scale, pre_shift)]): BatchNorm / LeakyReLU backward on load (c3d_wgrad_desc.fuse_*) -- ``dz`` is then an OUTPUT:
    the launch forms dz = LeakyReLU'(act) * (k[0] * dy + k[1] * act + k[2]) (k None: LeakyReLU'(act) * dy) while it
    stages its tiles, writes it to ``dz`` and folds sum(dz) into ``dbias`` (if given).  bf16x3 engine, fp32 tensors of
    one shape; ``wgrad_fusable`` says whether a layer qualifies."""
    d = L.WgradDesc()
    src.fill(d.x)
    b, h, w = src.t.shape[:3]
    d.dz, d.dz_cstride = dz.data_ptr(), dz.shape[3]
    d.dz_bf16 = int(dz.dtype == torch.bfloat16)
    d.B, d.H, d.W, d.Cout = b, h, w, dw.shape[0]
    d.ntaps = len(taps)
    for i, (dy, dx) in enumerate(taps):
        d.tap_dy[i], d.tap_dx[i] = dy, dx
    d.Cin_total, d.cin_off = dw.shape[1], cin_off
    d.dw, d.accumulate = dw.data_ptr(), int(accumulate)
    d.lrelu_slope = slope
    d.mfma_bf16 = MFMA_MODE           # the tiling (and with it the partial-sum size) depends on the engine
    d.variant = WGRAD_VARIANT
    if f16x2 is not None:      # EXPERIMENT (C3D_F16X2_BWD=1): (scale [Cout] = 2^s, inv [1] = 2^-s) of grad_exponent_max
        d.mfma_bf16 = 4
        d.dz_scale, d.out_scale_dev = f16x2[0].data_ptr(), f16x2[1].data_ptr()
    if fuse is not None:
        dy, act, k = fuse[:3]
        pre = fuse[3] if len(fuse) > 3 else None          # (pre_scale, pre_shift): conv -> BatchNorm -> LeakyReLU layer
        if not (dy.shape == act.shape == dz.shape and dy.dtype == act.dtype == dz.dtype == torch.float32
                and dy.is_contiguous() and act.is_contiguous() and dz.data_ptr() != dy.data_ptr()):
            raise ValueError("conv_wgrad(fuse=...): dy, act and dz must be distinct contiguous fp32 tensors of one shape")
        d.fuse_dy, d.fuse_act = dy.data_ptr(), act.data_ptr()
        if k is not None:
            d.fuse_k1, d.fuse_k2, d.fuse_k3 = k[0].data_ptr(), k[1].data_ptr(), k[2].data_ptr()
        if pre is not None:
            d.fuse_pre_scale, d.fuse_pre_shift = pre[0].data_ptr(), pre[1].data_ptr()
        nsum = L.lib().c3d_wgrad_fused_sum_n(C.byref(d))
        if nsum <= 0:
            raise RuntimeError("conv_wgrad(fuse=...): this layer shape / engine has no fused form (ops.wgrad_fusable)")
        zsum = torch.empty(dw.shape[0], 2, nsum, device=dz.device, dtype=torch.float32)
        d.fuse_sum = zsum.data_ptr()
        if bias_partial is not None:
            raise ValueError("conv_wgrad(fuse=...): the bias partials are the launch's own sums of dz")
        if dbias is not None:
            bias_partial = zsum
    # (after the fuse_* fields: a fused launch may take another tile configuration, and with it another partial size)
    n = L.lib().c3d_wgrad_partial_floats(C.byref(d))
    part = torch.empty(n, device=dz.device, dtype=torch.float32)
    d.partial = part.data_ptr()
    if bias_partial is not None:
        if dbias is None or dbias.shape[0] != dw.shape[0] or bias_partial.shape[0] < dw.shape[0]:
            raise ValueError("conv_wgrad: bias_partial needs a dbias of Cout entries")
        d.bias_partial, d.dbias, d.bias_n = bias_partial.data_ptr(), dbias.data_ptr(), bias_partial.shape[2]
    folds = WGRAD_FOLDS
    if folds is not None and not accumulate:
        rec = L.WgradFold()
        d.fold_out = C.addressof(rec)
    if KERNEL_EVENTS is None:        # (the kernel-name mirror only serves the per-kernel event timers)
        L.check(L.lib().c3d_conv_wgrad(C.byref(d), _stream()), "c3d_conv_wgrad")
        if d.fold_out:
            folds.add(rec, part, bias_partial)
        return dw
    halo = max(max(abs(dy), abs(dx)) for dy, dx in taps)
    co, ci, nt = dw.shape[0], src.C, len(taps)
    name = _wgrad_kernel_name(ci, co, nt, halo, fused=fuse is not None,
                              raw=src.t.dtype == torch.bfloat16 and dz.dtype == torch.bfloat16, h=h,
                              pre=fuse is not None and len(fuse) > 3 and fuse[3] is not None)
    with _Timed(name, 2.0 * b * h * w * dw.shape[0] * len(taps) * src.C, (h, w, ci, co, nt, halo, int(accumulate))):
        L.check(L.lib().c3d_conv_wgrad(C.byref(d), _stream()), "c3d_conv_wgrad")
    if d.fold_out:
        folds.add(rec, part, bias_partial)
    return dw


class WgradFolds:
    """Pending folds of weight-gradient partials (c3d_wgrad_desc.fold_out).  While one is installed (``ops.WGRAD_FOLDS``)
    ``conv_wgrad`` launches its matrix kernel only and queues the strips -> dw fold here; ``flush()`` runs them all in
    ceil(n / 32) launches (round 3: 74 launches of ~9 us per step, one behind every weight-gradient launch).  Nothing in
    the backward pass reads a weight gradient, so the backbone flushes once at its end (data parallel: whenever a
    gradient bucket is about to be sent).  The partial buffers are kept alive until the flush."""

    def __init__(self):
        self.records, self.keep = [], []

    def add(self, rec, *tensors):
        self.records.append(rec)
        self.keep.append(tensors)

    def flush(self):
        n = len(self.records)
        if n:
            arr = (L.WgradFold * n)(*self.records)
            L.check(L.lib().c3d_wgrad_fold_batch(arr, n, _stream()), "c3d_wgrad_fold_batch")
        self.records, self.keep = [], []


WGRAD_FOLDS = None       # installed by Backbone.backward for the duration of the pass


def wgrad_fusable(src, act, cout):
    """Whether conv_wgrad(fuse=...) exists for a layer: exact-split engine, fp32 tensors, unpadded channel count."""
    return (MFMA_MODE == 2 and src.t.dtype == torch.float32 and act.dtype == torch.float32 and act.shape[3] == cout
            and cout % 4 == 0 and tuple(src.t.shape[:3]) == tuple(act.shape[:3]))


# ---------------------------------------------------------------------------- BatchNorm
def _call(name, *args):
    L.check(getattr(L.lib(), name)(*args), name)


def _dp(t):
    return t.data_ptr() if t is not None else None


def stat_reduce(partial, c, sums=None, copy=False):
    """partial [C,2,n] fp32 -> sums [C,2] fp64 (copy=True: returns (sums, a second copy written by the same launch))."""
    if sums is None:
        sums = torch.empty(c, 2, device=partial.device, dtype=torch.float64)
    if copy:
        other = torch.empty_like(sums)
        _call("c3d_stat_reduce2", _dp(partial), partial.shape[2], c, _dp(sums), _dp(other), _stream())
        return sums, other
    _call("c3d_stat_reduce", _dp(partial), partial.shape[2], c, _dp(sums), _stream())
    return sums


def bn_finalize(sums, count, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5):
    c = gamma.shape[0]
    buf = torch.empty(4, c, device=gamma.device, dtype=torch.float32)
    _call("c3d_bn_finalize", _dp(sums), float(count), _dp(gamma), _dp(beta), _dp(running_mean),
          _dp(running_var), momentum, eps, c, _dp(buf[0]), _dp(buf[1]), _dp(buf[2]), _dp(buf[3]), _stream())
    return buf[0], buf[1], buf[2], buf[3]      # scale, shift, mean, invstd


def bn_finalize_partials(partial, count, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5):
    """stat_reduce + bn_finalize in one launch (single-rank path)."""
    c = gamma.shape[0]
    buf = torch.empty(4, c, device=gamma.device, dtype=torch.float32)
    _call("c3d_bn_finalize_partials", _dp(partial), partial.shape[2], float(count), _dp(gamma), _dp(beta),
          _dp(running_mean), _dp(running_var), momentum, eps, c, _dp(buf[0]), _dp(buf[1]), _dp(buf[2]), _dp(buf[3]),
          _stream())
    return buf[0], buf[1], buf[2], buf[3]


def bn_bwd_coeffs_partials(partial, count, mean, invstd, gamma, dgamma, dbeta):
    c = gamma.shape[0]
    k = torch.empty(3, c, device=gamma.device, dtype=torch.float32)
    _call("c3d_bn_bwd_coeffs_partials", _dp(partial), partial.shape[2], float(count), _dp(mean), _dp(invstd),
          _dp(gamma), c, _dp(k[0]), _dp(k[1]), _dp(k[2]), _dp(dgamma), _dp(dbeta), _stream())
    return k


def bias_from_partials(partial, out, accumulate=False):
    _call("c3d_bias_from_partials", _dp(partial), partial.shape[2], out.shape[0], _dp(out), int(accumulate), _stream())
    return out


def bn_eval_affine(gamma, beta, running_mean, running_var, eps=1e-5):
    c = gamma.shape[0]
    buf = torch.empty(2, c, device=gamma.device, dtype=torch.float32)
    _call("c3d_bn_eval_affine", _dp(gamma), _dp(beta), _dp(running_mean), _dp(running_var), eps, c,
          _dp(buf[0]), _dp(buf[1]), _stream())
    return buf[0], buf[1]


def bn_bwd_blocks(npix):
    return L.lib().c3d_bn_bwd_num_blocks(npix)


_BN_BWD_MAX_C = 1024      # channels one c3d_bn_bwd_* launch handles; wider layers (SqueezeSegV3's 9C attention maps) go in slices


def _off(t, c0):
    return None if t is None else t.data_ptr() + t.element_size() * c0


def bn_bwd_reduce(dy, a, c, mode=0, pre_scale=None, pre_shift=None, slope=0.0):
    npix = dy.numel() // dy.shape[-1]
    part = torch.empty(c, 2, bn_bwd_blocks(npix), device=dy.device, dtype=torch.float32)
    for c0 in range(0, c, _BN_BWD_MAX_C):
        cc = min(_BN_BWD_MAX_C, c - c0)
        _call("c3d_bn_bwd_reduce", _off(dy, c0), dy.shape[-1], _off(a, c0), a.shape[-1], npix, cc, mode, _off(pre_scale, c0),
              _off(pre_shift, c0), _dp(part[c0:c0 + cc]), float(slope), _bf(dy, a), _stream())
    return part


def grad_exponent(apply_partial, c, target_log2=8):
    """(scale [c] = 2^s, inv [1] = 2^-s) from the max |dz| row of bn_bwd_apply's partials: the per-tensor exponent that
    brings a gradient tensor into fp16's range (times the kernel's own 2^6: its maximum lands at 2^14)."""
    n = apply_partial.shape[2]
    scale = torch.empty(c, device=apply_partial.device, dtype=torch.float32)
    inv = torch.empty(1, device=apply_partial.device, dtype=torch.float32)
    _call("c3d_grad_exponent", _dp(apply_partial), n, min(c, apply_partial.shape[0]), target_log2, _dp(scale), c, _dp(inv), _stream())
    return scale, inv


def grad_exponent_max(gmax, c, target_log2=8):
    """grad_exponent from the word bn_bwd_apply(gmax=...) filled."""
    scale = torch.empty(c, device=gmax.device, dtype=torch.float32)
    inv = torch.empty(1, device=gmax.device, dtype=torch.float32)
    _call("c3d_grad_exponent_max", _dp(gmax), target_log2, _dp(scale), c, _dp(inv), _stream())
    return scale, inv


def bn_bwd_coeffs(sums, count, mean, invstd, gamma, dgamma, dbeta, sums_param=None):
    c = gamma.shape[0]
    k = torch.empty(3, c, device=gamma.device, dtype=torch.float32)
    _call("c3d_bn_bwd_coeffs", _dp(sums), _dp(sums_param), float(count), _dp(mean), _dp(invstd), _dp(gamma), c, _dp(k[0]),
          _dp(k[1]), _dp(k[2]), _dp(dgamma), _dp(dbeta), _stream())
    return k


def bn_bwd_apply(dy, a, c, mode, k=None, pre_scale=None, pre_shift=None, dz=None, slope=0.0, gmax=None):
    """dz = act'(.) * (k1*dy + k2*a + k3); returns (dz, partial [C,2,nblk] with sum(dz) in row 0, max |dz| in row 1).
    gmax: optional zeroed int32 [1]; receives the bits of max |dz| over the tensor (the f16x2 gradient experiment)."""
    npix = dy.numel() // dy.shape[-1]
    if dz is None:
        dz = torch.empty(dy.shape[:-1] + (c,), device=dy.device, dtype=a.dtype)
    part = torch.empty(c, 2, bn_bwd_blocks(npix), device=dy.device, dtype=torch.float32)
    k1, k2, k3 = (k[0], k[1], k[2]) if k is not None else (None, None, None)
    for c0 in range(0, c, _BN_BWD_MAX_C):
        cc = min(_BN_BWD_MAX_C, c - c0)
        if gmax is not None:
            _call("c3d_bn_bwd_apply_gmax", _off(dy, c0), dy.shape[-1], _off(a, c0), a.shape[-1], npix, cc, mode, _off(pre_scale, c0),
                  _off(pre_shift, c0), _off(k1, c0), _off(k2, c0), _off(k3, c0), _off(dz, c0), dz.shape[-1],
                  _dp(part[c0:c0 + cc]), float(slope), _bf(dy, a, dz), _dp(gmax), _stream())
            continue
        _call("c3d_bn_bwd_apply", _off(dy, c0), dy.shape[-1], _off(a, c0), a.shape[-1], npix, cc, mode, _off(pre_scale, c0),
              _off(pre_shift, c0), _off(k1, c0), _off(k2, c0), _off(k3, c0), _off(dz, c0), dz.shape[-1],
              _dp(part[c0:c0 + cc]), float(slope), _bf(dy, a, dz), _stream())
    return dz, part


def sums_to_f32(sums, col, out, accumulate=False):
    _call("c3d_sums_to_f32", _dp(sums), out.shape[0], col, _dp(out), int(accumulate), _stream())
    return out


# ---------------------------------------------------------------------------- glue
def input_norm(x, eval_label, mean, std):
    b, cn, h, w = x.shape
    out = torch.empty_like(x)
    _call("c3d_input_norm", _dp(x), _dp(eval_label), _dp(mean), _dp(std), b, cn, h * w, _dp(out), _stream())
    return out


def conv_in5(x_nchw, w, bias):
    b, cn, h, wd = x_nchw.shape
    out = torch.empty(b, h, wd, 32, device=x_nchw.device, dtype=act_dtype())
    _call("c3d_conv_in5", _dp(x_nchw), _dp(w), _dp(bias), b, cn, h * wd, _dp(out), _bf(out), _stream())
    return out


def conv_in5_wgrad(x_nchw, dz, dw):
    b, cn, h, wd = x_nchw.shape
    part = torch.empty(1024 * 32 * 8, device=dz.device, dtype=torch.float32)
    _call("c3d_conv_in5_wgrad", _dp(x_nchw), _dp(dz), b, cn, h * wd, _dp(part), _dp(dw), _bf(dz), _stream())
    return dw


def affine_add(x, a, scale=None, shift=None, out=None, slope=0.0):
    """out = x + act(a*scale + shift); act = LeakyReLU(slope) or the identity (slope 0)."""
    c = a.shape[-1]
    if out is None:
        out = torch.empty_like(a)
    _call("c3d_affine_add", _dp(x), _dp(a), _dp(scale), _dp(shift), a.numel() // c, c, float(slope), _dp(out),
          _bf(x, a, out), _stream())
    return out


def cols_resample(x, up):
    """NHWC [B,H,W,C]: up=False keeps the even columns (-> W/2), up=True inserts zero columns (-> 2W)."""
    b, h, w, c = x.shape
    out = torch.empty(b, h, w * 2 if up else w // 2, c, device=x.device, dtype=torch.float32)
    _call("c3d_cols_resample", _dp(x), b * h, w, c, int(up), _dp(out), _stream())
    return out


def nchw_to_nhwc_pad(x, cp):
    b, cn, h, w = x.shape
    out = torch.empty(b, h, w, cp, device=x.device, dtype=torch.float32)
    _call("c3d_nchw_to_nhwc_pad", _dp(x.contiguous()), b, cn, h * w, cp, _dp(out), _stream())
    return out


def axpy(x, y, alpha=1.0, accumulate=True):
    _call("c3d_axpy", _dp(x), alpha, x.numel(), _dp(y), int(accumulate), _bf(x, y), _stream())
    return y


def maskpool(x, mask, pool):
    b, h, w, c = x.shape
    ho, wo = ((h + 1) // 2, (w + 1) // 2) if pool else (h, w)
    out = torch.empty(b, ho, wo, c, device=x.device, dtype=x.dtype)
    _call("c3d_maskpool", _dp(x), _dp(mask), b, h, w, c, int(pool), _dp(out), _bf(x, out), _stream())
    return out


def maskpool_bwd(dout, mask, extra, shape, pool):
    b, h, w, c = shape
    din = torch.empty(b, h, w, c, device=dout.device, dtype=dout.dtype)
    _call("c3d_maskpool_bwd", _dp(dout), _dp(mask), _dp(extra), b, h, w, c, int(pool), _dp(din), _bf(dout, extra, din),
          _stream())
    return din


def pixshuf_cat(xa, sc, sh, m3, m1, m2, skip):
    """PixelShuffle(2) of BN(xa) (+ dropout masks) concatenated with ``skip``; ``skip=None``: the shuffled part only
    (the consumer then takes the skip tensor as a second conv source instead of a copy of it)."""
    b, hs, ws, cx = xa.shape
    cs = skip.shape[-1] if skip is not None else 0
    out = torch.empty(b, 2 * hs, 2 * ws, cx // 4 + cs, device=xa.device, dtype=xa.dtype)
    _call("c3d_pixshuf_cat", _dp(xa), _dp(sc), _dp(sh), _dp(m3), _dp(m1), _dp(m2), _dp(skip), b, hs, ws, cx, cs,
          _dp(out), _bf(xa, skip, out), _stream())
    return out


def pixshuf_cat_bwd(dout, m3, m1, m2, xa_shape, cs, dskip, skip_accumulate):
    b, hs, ws, cx = xa_shape
    dxa = torch.empty(b, hs, ws, cx, device=dout.device, dtype=dout.dtype)
    _call("c3d_pixshuf_cat_bwd", _dp(dout), _dp(m3), _dp(m1), _dp(m2), b, hs, ws, cx, cs, _dp(dxa), _dp(dskip),
          int(skip_accumulate), _bf(dout, dxa, dskip), _stream())
    return dxa


def softmax(logits, c, ho=None, wo=None):
    b, h, w, cs = logits.shape
    ho, wo = ho or h, wo or w
    prob = torch.empty(b, ho, wo, c, device=logits.device, dtype=torch.float32)
    _call("c3d_softmax", _dp(logits), b, h, w, cs, c, ho, wo, _dp(prob), _stream())
    return prob


def softmax_bwd(prob, dprob, shape):
    b, h, w, cs = shape
    _, ho, wo, c = prob.shape
    dl = torch.empty(b, h, w, cs, device=prob.device, dtype=torch.float32)
    _call("c3d_softmax_bwd", _dp(prob), _dp(dprob), b, h, w, cs, c, ho, wo, _dp(dl), _stream())
    return dl


def bilinear(src, hd, wd, dst=None, dcoff=0, c=None, scoff=0, out_dtype=None):
    b, hs, ws, scs = src.shape
    c = c or scs
    if dst is None:
        dst = torch.empty(b, hd, wd, c, device=src.device, dtype=out_dtype or src.dtype)
    _call("c3d_bilinear", _dp(src), hs, ws, scs, scoff, _dp(dst), hd, wd, dst.shape[-1], dcoff, b, c, _bf(src, dst),
          _stream())
    return dst


def bilinear_sum2(src1, src2, hd, wd, out_dtype=torch.float32):
    """bilinear(src1 -> hd x wd) + bilinear(src2 -> hd x wd), NHWC fp32 tensors with the same channel count (the result may be a
    bf16 activation tensor)."""
    b, h1, w1, c = src1.shape
    _, h2, w2, c2 = src2.shape
    assert c == c2 and src1.dtype == src2.dtype == torch.float32 and src1.is_contiguous() and src2.is_contiguous()
    assert out_dtype in (torch.float32, torch.bfloat16)
    dst = torch.empty(b, hd, wd, c, device=src1.device, dtype=out_dtype)
    _call("c3d_bilinear_sum2", _dp(src1), h1, w1, _dp(src2), h2, w2, _dp(dst), hd, wd, b, c, int(out_dtype == torch.bfloat16),
          _stream())
    return dst


def bilinear_bwd(dsrc, ddst, dcoff=0, c=None, scoff=0, accumulate=False, rowmask=None):
    """rowmask: int32 bitmap over the pixels of ddst (scatter_add_rows writes it); clear bit = known zeros, not read."""
    b, hs, ws, scs = dsrc.shape
    _, hd, wd, dcs = ddst.shape
    c = c or scs
    if rowmask is not None and (rowmask.dtype != torch.int32 or rowmask.numel() * 32 < b * hd * wd):
        raise ValueError("bilinear_bwd: rowmask must be int32 with one bit per destination pixel")
    _call("c3d_bilinear_bwd", _dp(dsrc), hs, ws, scs, scoff, _dp(ddst), hd, wd, dcs, dcoff, b, c, int(accumulate),
          _bf(dsrc, ddst), _dp(rowmask), _stream())
    return dsrc


def bilinear_bwd_rows(dsrc, drows, cmap, rowmask, hd, wd, accumulate=False):
    """bilinear_bwd for a destination gradient in compact form (scatter_rows_compact): bit-identical to the dense call."""
    b, hs, ws, scs = dsrc.shape
    if drows.dtype != torch.float32 or cmap.dtype != torch.int32 or rowmask.dtype != torch.int32:
        raise ValueError("bilinear_bwd_rows: drows fp32, cmap / rowmask int32")
    if cmap.numel() < b * hd * wd or rowmask.numel() * 32 < b * hd * wd or drows.shape[-1] != scs:
        raise ValueError("bilinear_bwd_rows: cmap / rowmask must cover every destination pixel, rows must have C channels")
    _call("c3d_bilinear_bwd_rows", _dp(dsrc), hs, ws, scs, 0, _dp(drows), _dp(cmap), _dp(rowmask), hd, wd, b, scs,
          int(accumulate), int(dsrc.dtype == torch.bfloat16), _stream())
    return dsrc


def bilinear_rows(src, hd, wd, idx, img=None, a=1, count=None, l2=False, eps=1e-12):
    """Rows of bilinear(src -> hd x wd) (align_corners=True) at listed destination pixels, without the map.
    idx: int64 flat pixels b*hd*wd + p (img None) or int32 pixels of image img[r // a]; rows with r // a >= count[0]
    (int32 device scalar) are zero.  l2: rows l2-normalised, returns (rows, norms).  Rows are padded to a multiple of 32."""
    b, hs, ws, c = src.shape
    r = idx.numel()
    rpad = (r + 31) // 32 * 32            # the GEMM engine wants a multiple of 32 rows
    out = (torch.empty if rpad == r else torch.zeros)(rpad, c, device=src.device, dtype=torch.float32)
    norm = torch.ones(rpad, device=src.device, dtype=torch.float32) if l2 else None
    if idx.dtype not in (torch.int32, torch.int64) or (img is not None and idx.dtype != torch.int32):
        raise ValueError("bilinear_rows: idx int64 (flat) or int32 with img")
    _call("c3d_bilinear_rows", _dp(src), hs, ws, c, 0, int(src.dtype == torch.bfloat16), hd, wd, b, c, _dp(img), _dp(idx),
          int(idx.dtype == torch.int64), a, _dp(count), r, int(l2), eps, _dp(out), _dp(norm), _stream())
    return (out, norm) if l2 else out


def l2norm(x, eps=1e-12, want_norm=True):
    c = x.shape[-1]
    n = x.numel() // c
    y = torch.empty_like(x)
    norm = torch.empty(n, device=x.device, dtype=torch.float32) if want_norm else None
    _call("c3d_l2norm", _dp(x), n, c, eps, _dp(y), _dp(norm), _bf(x, y), _stream())
    return y, norm


def l2norm_bwd(y, norm, dy, eps=1e-12):
    c = y.shape[-1]
    dx = torch.empty_like(y)
    _call("c3d_l2norm_bwd", _dp(y), _dp(norm), _dp(dy), y.numel() // c, c, eps, _dp(dx), _bf(y, dy, dx), _stream())
    return dx


# ---------------------------------------------------------------------------- prototypes
def rownorm_ln_l2(x, w, b, ln_eps=1e-5, l2_eps=1e-12):
    c = x.shape[-1]
    out = torch.empty_like(x)
    _call("c3d_rownorm_ln_l2", _dp(x), x.numel() // c, c, _dp(w), _dp(b), ln_eps, l2_eps, _dp(out), _stream())
    return out


def proto_nearest(sim, m, c, w, b, eps=1e-5, want_nearest=False):
    n = sim.shape[0]
    nearest = torch.empty(n, c, device=sim.device, dtype=torch.float32) if want_nearest else None
    pred = torch.empty(n, device=sim.device, dtype=torch.int32)
    _call("c3d_proto_nearest", _dp(sim), n, m, c, _dp(w), _dp(b), eps, _dp(nearest), _dp(pred), _stream())
    return nearest, pred


def group_compact(labels, ncls, keep=None):
    """labels int64 [G, n] -> (counts int32 [G, ncls], idx int32 [G, ncls, n])."""
    g, n = labels.shape
    counts = torch.empty(g, ncls, device=labels.device, dtype=torch.int32)
    idx = torch.empty(g, ncls, n, device=labels.device, dtype=torch.int32)
    seg = torch.empty(g * 8 * ncls, device=labels.device, dtype=torch.int32)
    _call("c3d_group_compact", _dp(labels), _dp(keep), g, n, ncls, _dp(counts), _dp(idx), _dp(seg), _stream())
    return counts, idx


def label_hist(labels, ncls):
    """labels int64 [G, n] -> counts int32 [G, ncls] (classes >= 1)."""
    g, n = labels.shape
    counts = torch.empty(g, ncls, device=labels.device, dtype=torch.int32)
    _call("c3d_label_hist", _dp(labels), g, n, ncls, _dp(counts), _stream())
    return counts


def proto_learn(sim, feat, pred, counts, idx, noise, protos, m, c, ignore_label, momentum, ln_w=None, ln_b=None,
                ln_eps=1e-5, sums_reduce=None, cmap=None, noise_by_row=False):
    """counts [B, C], idx [B, C, n]: per-image ordered pixel lists of each class.
    ``sums_reduce`` (data parallel, optional): in-place all-reduce applied to the per-class
    feature sums + counts [C, M, D+1] before the EMA ("per-class prototype sums" exchange).
    ``cmap`` int32 [>= B*n]: sim / feat are compact (rows of the labelled pixels only) and cmap[pixel] is a pixel's row."""
    d = feat.shape[1]
    b, _, n = idx.shape
    ntot = b * n
    protos_out = torch.empty_like(protos)
    target = torch.zeros(ntot, device=feat.device, dtype=torch.float32)
    assign = torch.empty(ntot, device=feat.device, dtype=torch.int32)
    rows = torch.empty(c, ntot, device=feat.device, dtype=torch.int32)
    assert pred is not None or ln_w is not None
    fsum = torch.empty(c, m, d + 1, device=feat.device, dtype=torch.float32) if sums_reduce is not None else None
    _call("c3d_proto_learn", _dp(sim), _dp(feat), _dp(pred), _dp(ln_w), _dp(ln_b), ln_eps, _dp(counts), _dp(idx),
          _dp(rows), _dp(noise),
          _dp(protos), _dp(protos_out), _dp(target), _dp(assign), b, n, m, c, d, ignore_label, momentum, _dp(fsum),
          _dp(cmap), int(noise_by_row), _stream())
    if fsum is not None:
        sums_reduce(fsum)
        _call("c3d_proto_ema", _dp(fsum), _dp(protos), _dp(protos_out), m, c, d, ignore_label, momentum, _stream())
    return protos_out, target


# ---------------------------------------------------------------------------- loss path
def entropy_stats(prob, want_anchor=True, want_pl=True, want_amax=True):
    c = prob.shape[-1]
    n = prob.numel() // c
    dev = prob.device
    wa = torch.empty(n, device=dev, dtype=torch.float32) if want_anchor else None
    wp = torch.empty(n, device=dev, dtype=torch.float32) if want_pl else None
    am = torch.empty(n, device=dev, dtype=torch.int32) if want_amax else None
    _call("c3d_entropy_stats", _dp(prob), n, c, _dp(wa), _dp(wp), _dp(am), _stream())
    return wa, wp, am


def pl_select(w_pl, amax, eval_label, train_label, noise, tl_counts, b, n, c, ignore_label, ratio):
    dev = w_pl.device
    chosen = torch.zeros(b, n, device=dev, dtype=torch.uint8)
    labels = torch.empty(b, n, device=dev, dtype=torch.int64)
    mask = torch.empty(b, n, device=dev, dtype=torch.uint8)
    scratch = torch.empty(2 * b * c + 2 * b * n, device=dev, dtype=torch.int32)
    if isinstance(ratio, torch.Tensor):      # device scalar (fp32 [1]): the captured step reads the epoch's ratio from it
        if ratio.dtype != torch.float32 or ratio.numel() != 1 or ratio.device != dev:
            raise ValueError("pl_select: a tensor ratio must be one fp32 value on the labels' device")
        _call("c3d_pl_select_dev", _dp(w_pl), _dp(amax), _dp(eval_label), _dp(train_label), _dp(noise), _dp(tl_counts), b, n,
              c, ignore_label, _dp(ratio), _dp(scratch), _dp(chosen), _dp(labels), _dp(mask), _stream())
        return labels, mask.bool()
    _call("c3d_pl_select", _dp(w_pl), _dp(amax), _dp(eval_label), _dp(train_label), _dp(noise), _dp(tl_counts), b, n,
          c, ignore_label, float(ratio), _dp(scratch), _dp(chosen), _dp(labels), _dp(mask), _stream())
    return labels, mask.bool()


def anchor_sample(weights, counts, idx, uniforms, b, n, c, a, ignore_label):
    dev = weights.device
    slot = torch.empty(b * c, device=dev, dtype=torch.int32)
    cum = torch.empty(b, c, n, device=dev, dtype=torch.float32)
    a_idx = torch.zeros(b * c, a, device=dev, dtype=torch.int32)
    a_img = torch.zeros(b * c, device=dev, dtype=torch.int32)
    a_cls = torch.zeros(b * c, device=dev, dtype=torch.int32)
    t = torch.zeros(1, device=dev, dtype=torch.int32)
    _call("c3d_anchor_sample", _dp(weights), _dp(counts), _dp(idx), _dp(uniforms), b, n, c, a, ignore_label,
          _dp(slot), _dp(cum), _dp(a_idx), _dp(a_img), _dp(a_cls), _dp(t), _stream())
    return a_idx, a_img, a_cls, t


def gather_rows_l2(feat, img, idx, t, tmax, a, n, eps=1e-12):
    d = feat.shape[-1]
    rows = tmax * a
    rpad = (rows + 31) // 32 * 32            # the GEMM engine wants a multiple of 32 rows
    out = (torch.empty if rpad == rows else torch.zeros)(rpad, d, device=feat.device, dtype=torch.float32)
    norm = torch.ones(rpad, device=feat.device, dtype=torch.float32)
    _call("c3d_gather_rows_l2", _dp(feat), _dp(img), _dp(idx), _dp(t), tmax, a, n, d, eps, _dp(out), _dp(norm),
          _stream())
    return out, norm


def scatter_add_rows(dx, img, idx, t, tmax, a, n, dfeat, gscale=None, rowmask=None):
    """rowmask: optional pre-zeroed int32 bitmap ((B*n + 31)//32 words); the bit of every written row gets set."""
    d = dx.shape[-1]
    _call("c3d_scatter_add_rows", _dp(dx), _dp(img), _dp(idx), _dp(t), tmax, a, n, d, _dp(gscale), _dp(dfeat),
          _dp(rowmask), _stream())
    return dfeat


def scatter_rows_compact(dx, img, idx, t, tmax, a, n, b, gscale=None):
    """scatter_add_rows without the dense target: (drows [tmax*a, D], cmap int32 [b*n], rowmask int32 bitmap)."""
    d = dx.shape[-1]
    drows = torch.empty(tmax * a, d, device=dx.device, dtype=torch.float32)
    cmap = torch.empty(b * n, device=dx.device, dtype=torch.int32)
    rowmask = torch.zeros((b * n + 31) // 32, device=dx.device, dtype=torch.int32)
    _call("c3d_scatter_rows_compact", _dp(dx), _dp(img), _dp(idx), _dp(t), tmax, a, n, d, _dp(gscale), _dp(drows),
          _dp(cmap), _dp(rowmask), _stream())
    return drows, cmap, rowmask


def infonce_rows(logits, row_cls, t, tmax, a, m, ncols, temperature, base_temperature=0.07):
    dev = logits.device
    row_loss = torch.empty(tmax * a, device=dev, dtype=torch.float32)
    loss = torch.empty(1, device=dev, dtype=torch.float32)
    _call("c3d_infonce_rows", _dp(logits), logits.shape[-1], _dp(row_cls), _dp(t), tmax, a, m, ncols,
          temperature, base_temperature, _dp(row_loss), _dp(loss), _stream())
    return loss, row_loss


def gemm_rows(x, w_oihw_packed, cout, out=None):
    """[R, K] x packed weight -> [R, cout] on the MFMA conv engine (R multiple of 32)."""
    r, k = x.shape
    assert r % 32 == 0
    src = Source(x.view(1, r // 32, 32, k))
    y, _ = conv_forward([src], w_oihw_packed, None, cout, [(0, 0)],
                        out=None if out is None else out.view(1, r // 32, 32, out.shape[-1]))
    return y.view(r, -1)


# ---------------------------------------------------------------------------- metrics (N1)
def confusion_add(pred, label, conf):
    """conf[pred[i]][label[i]] += 1 (int64 [C,C], in place)."""
    _call("c3d_confusion_add", _dp(pred), _dp(label), pred.numel(), conf.shape[0], _dp(conf), _stream())
    return conf


def unproject_confusion(pred_2d, uy, ux, labels, conf, n_points=None):
    """pred_2d [C,H,W]-shaped probabilities of one scan (read in place when its memory is
    channels-last, i.e. a view of an NHWC buffer); uy/ux pixel coordinates per point (ux None:
    uy is the flat pixel index, SemanticPOSS); labels int64 [n].  Returns argmax per point."""
    c, h, w = pred_2d.shape
    p = pred_2d.permute(1, 2, 0)
    if not (p.stride(2) == 1 and p.stride(1) >= c and p.stride(0) == w * p.stride(1)):
        p = p.contiguous()
    dev = p.device
    uy = uy.to(dev, torch.int32).contiguous()
    uxp = ux.to(dev, torch.int32).contiguous() if ux is not None else None
    labels = labels.to(dev, torch.long).reshape(-1).contiguous()
    n = labels.numel() if n_points is None else n_points
    if labels.numel() != n:
        raise ValueError(f"labels has {labels.numel()} entries, expected {n}")
    nv = uy.numel()
    if ux is not None and ux.numel() != nv:
        raise ValueError("uproj_x_idx and uproj_y_idx differ in length")
    if nv > n:
        raise ValueError("more un-projection indices than points")
    out = torch.empty(n, device=dev, dtype=torch.int32)
    _call("c3d_unproject_confusion", _dp(p), h, w, c, p.stride(1), _dp(uy), _dp(uxp), _dp(labels), n, nv, _dp(conf),
          _dp(out), _stream())
    return out


# ---------------------------------------------------------------------------- loss head (N1)
def _rows(prob_nchw_like):
    """[B,C,H,W]-shaped probabilities -> ([N, cstride-strided] NHWC tensor, C, cstride): read in
    place when the memory is channels-last (a view of the backbone's NHWC buffer)."""
    c = prob_nchw_like.shape[1]
    p = prob_nchw_like.permute(0, 2, 3, 1)
    if not p.is_contiguous():
        p = p.contiguous()
    return p, c, c


def focal_forward(prob, target, mask, alpha, gamma):
    """prob [B,C,H,W]-shaped, target int64 [B,H,W], mask uint8/bool [B,H,W] or None ->
    stats float32 [2] = (mean focal loss, selected pixels)."""
    p, c, cs = _rows(prob)
    n = p.numel() // c
    nblk = max(1, min(1024, (n + 2047) // 2048))
    part = torch.empty(2 * nblk, device=p.device, dtype=torch.float64)
    out = torch.empty(2, device=p.device, dtype=torch.float32)
    _call("c3d_focal_forward", _dp(p), c, cs, _dp(target), _dp(mask), _dp(alpha), float(gamma), n, _dp(part), nblk,
          _dp(out), _stream())
    return out


def focal_backward(prob, target, mask, alpha, gamma, stats, gscale, dprob):
    """dprob (NHWC [B,H,W,C], accumulated in place) += gscale * d(mean focal)/d prob."""
    p, c, cs = _rows(prob)
    _call("c3d_focal_backward", _dp(p), c, cs, _dp(target), _dp(mask), _dp(alpha), float(gamma), p.numel() // c,
          _dp(stats), _dp(gscale), _dp(dprob), dprob.shape[-1], _stream())
    return dprob


def lovasz_max_pixels():
    return L.lib().c3d_lovasz_max_pixels()


def lovasz_forward(prob, labels, idx):
    """idx int64 [P]: flat positions of the labelled pixels.  Returns (stats [2] = (loss, present
    classes), grad [C,P])."""
    p, c, cs = _rows(prob)
    n_idx = idx.numel()
    loss_c = torch.empty(c, device=p.device, dtype=torch.float32)
    present = torch.empty(c, device=p.device, dtype=torch.float32)
    grad = torch.empty(c, max(n_idx, 1), device=p.device, dtype=torch.float32)
    out = torch.empty(2, device=p.device, dtype=torch.float32)
    if n_idx > lovasz_max_pixels():
        # beyond the LDS sort: device-wide segmented sort per class (fully supervised batches)
        nbytes = L.lib().c3d_lovasz_workspace_bytes(c, n_idx)
        ws = torch.empty(nbytes, device=p.device, dtype=torch.uint8)
        _call("c3d_lovasz_forward_large", _dp(p), c, cs, _dp(labels), _dp(idx), n_idx, _dp(loss_c), _dp(present), _dp(grad),
              _dp(out), _dp(ws), nbytes, _stream())
        return out, grad
    _call("c3d_lovasz_forward", _dp(p), c, cs, _dp(labels), _dp(idx), n_idx, _dp(loss_c), _dp(present), _dp(grad),
          _dp(out), _stream())
    return out, grad


def lovasz_forward_dyn(prob, labels, idx, count):
    """idx int64 [cap] (cap <= lovasz_max_pixels()), count int32 [1] on the device: the first ``count`` entries are
    the labelled pixels.  Shape-static (hipGraph-capturable) form of ``lovasz_forward``; grad is [C, cap]."""
    p, c, cs = _rows(prob)
    cap = idx.numel()
    loss_c = torch.empty(c, device=p.device, dtype=torch.float32)
    present = torch.empty(c, device=p.device, dtype=torch.float32)
    grad = torch.empty(c, cap, device=p.device, dtype=torch.float32)
    out = torch.empty(2, device=p.device, dtype=torch.float32)
    _call("c3d_lovasz_forward_dyn", _dp(p), c, cs, _dp(labels), _dp(idx), _dp(count), cap, _dp(loss_c), _dp(present),
          _dp(grad), _dp(out), _stream())
    return out, grad


def lovasz_backward_dyn(grad, idx, count, stats, gscale, dprob):
    _call("c3d_lovasz_backward_dyn", _dp(grad), _dp(idx), _dp(count), idx.numel(), grad.shape[0], _dp(stats), _dp(gscale),
          _dp(dprob), dprob.shape[-1], _stream())
    return dprob


def lovasz_backward(grad, idx, stats, gscale, dprob):
    _call("c3d_lovasz_backward", _dp(grad), _dp(idx), idx.numel(), grad.shape[0], _dp(stats), _dp(gscale), _dp(dprob),
          dprob.shape[-1], _stream())
    return dprob


# ---------------------------------------------------------------------------- scan -> range image (N2)
def augment_points(pc, sx, sy, trans, rot):
    """In place: flip (sx, sy = +-1), translation (float32), rotation (3x3 float64, points x rot^T)."""
    assert pc.is_cuda and pc.dtype == torch.float32 and pc.dim() == 2 and pc.is_contiguous()
    r = torch.as_tensor(np.asarray(rot, dtype=np.float64).reshape(9)).to(pc.device)
    _call("c3d_augment_points", _dp(pc), pc.shape[0], pc.shape[1], float(sx), float(sy), float(np.float32(trans[0])),
          float(np.float32(trans[1])), float(np.float32(trans[2])), _dp(r), _stream())
    return pc


def range_project(pc, depth, fov, w, h, want_image=True, sem=None, weak=None):
    """pc CUDA float32 [n, c]; fov = (|fov_left|, fov_hori, |fov_down|, fov_vert) as float32 values.
    Returns a dict of CUDA tensors (see c3d_range_project)."""
    assert pc.is_cuda and pc.dtype == torch.float32 and pc.dim() == 2 and pc.is_contiguous()
    n, c = pc.shape
    dev = pc.device
    out = {"ux": torch.empty(n, device=dev, dtype=torch.int32), "uy": torch.empty(n, device=dev, dtype=torch.int32),
           "udepth": torch.empty(n, device=dev, dtype=torch.float32),
           "proj_idx": torch.empty(h, w, device=dev, dtype=torch.int32),
           "proj_mask": torch.empty(h, w, device=dev, dtype=torch.int32)}
    zbuf = torch.empty(h * w, device=dev, dtype=torch.int64)
    if want_image:
        out["proj_pc"] = torch.empty(h, w, c, device=dev, dtype=torch.float32)
        out["proj_range"] = torch.empty(h, w, device=dev, dtype=torch.float32)
    if sem is not None:
        sem = sem.to(dev, torch.long).contiguous()
        weak = weak.to(dev, torch.long).contiguous()
        assert sem.numel() == n and weak.numel() == n
        out["feat5"] = torch.empty(5, h, w, device=dev, dtype=torch.float32)
        out["eval_label"] = torch.empty(h, w, device=dev, dtype=torch.float32)
        out["train_label"] = torch.empty(h, w, device=dev, dtype=torch.float32)
    if depth is not None:
        depth = depth.to(dev, torch.float32).contiguous()
        assert depth.numel() == n
    _call("c3d_range_project", _dp(pc), n, c, c, _dp(depth), fov[0], fov[1], fov[2], fov[3], w, h, _dp(out["ux"]),
          _dp(out["uy"]), _dp(out["udepth"]), _dp(zbuf), _dp(out.get("proj_pc")), _dp(out.get("proj_range")),
          _dp(out["proj_idx"]), _dp(out["proj_mask"]), _dp(sem), _dp(weak), _dp(out.get("feat5")),
          _dp(out.get("eval_label")), _dp(out.get("train_label")), _stream())
    return out


# ---------------------------------------------------------------------------- SqueezeSegV3 SAC block (N3)
def sac_im2col7(xyz4):
    """xyz [B,H,W,>=3] -> the 7x7 neighbourhood columns [B,H,W,160] (147 used)."""
    b, h, w, cs = xyz4.shape
    out = torch.empty(b, h, w, 160, device=xyz4.device, dtype=torch.float32)
    _call("c3d_sac_im2col7", _dp(xyz4), b, h, w, cs, _dp(out), _stream())
    return out


def sac_modulate(feat, att, scale, shift):
    b, h, w, c = feat.shape
    m = torch.empty(b, h, w, 9 * c, device=feat.device, dtype=torch.float32)
    _call("c3d_sac_modulate", _dp(feat), _dp(att), _dp(scale), _dp(shift), b, h, w, c, _dp(m), _stream())
    return m


def sac_modulate_bwd(dm, feat, att, scale, shift):
    """Returns d(att BatchNorm output); ``dm`` is overwritten with dm * sigmoid (input of sac_fold)."""
    b, h, w, c = feat.shape
    datt = torch.empty_like(dm)
    _call("c3d_sac_modulate_bwd", _dp(dm), _dp(feat), _dp(att), _dp(scale), _dp(shift), b, h, w, c, _dp(datt), _stream())
    return datt


def sac_fold(t, dfeat, accumulate):
    b, h, w, c = dfeat.shape
    _call("c3d_sac_fold", _dp(t), b, h, w, c, int(accumulate), _dp(dfeat), _stream())
    return dfeat


# ---------------------------------------------------------------------------- kNN clean-up (N4)
def knn_vote(proj_range, proj_argmax, unproj_range, px, py, inv_gauss, search, knn, cutoff, nclasses):
    h, w = proj_range.shape
    dev = proj_range.device
    pr = proj_range.to(torch.float32).contiguous()
    pa = proj_argmax.to(dev, torch.long).contiguous()
    ur = unproj_range.to(dev, torch.float32).contiguous()
    pxl, pyl = px.to(dev, torch.long).contiguous(), py.to(dev, torch.long).contiguous()
    n = ur.numel()
    out = torch.empty(n, device=dev, dtype=torch.long)
    _call("c3d_knn_vote", _dp(pr), _dp(pa), h, w, _dp(ur), _dp(pxl), _dp(pyl), n, _dp(inv_gauss.to(torch.float32).contiguous()),
          int(search), int(knn), float(cutoff), int(nclasses), _dp(out), _stream())
    return out
