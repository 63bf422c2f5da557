"""bf16 ACTIVATION STORAGE (BASELINE configs[2]): every kernel that reads or writes backbone
activations with the bf16 flag of the C ABI set, against the same kernel on fp32 tensors holding
the same (bf16-representable) values.  Arithmetic is fp32 in both, so the only difference allowed
is the final rounding of the stored result: |diff| <= 2^-8 |ref| (one bf16 ulp) + a tiny absolute
term.  Statistics, norms and parameter gradients are fp32 outputs and must agree to fp32 accuracy."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def r16(t):
    """fp32 tensor with bf16-representable values."""
    return t.to(torch.bfloat16).float()


def close16(got, ref, what):
    got, ref = got.float(), ref.float()
    err = (got - ref).abs()
    bound = ref.abs() * 2.0 ** -8 + 1e-6 * float(ref.abs().max())
    assert bool((err <= bound).all()), (what, float((err - bound).max()))


def close32(got, ref, what, tol=1e-5):
    e = float((got.double() - ref.double()).abs().max() / (ref.double().abs().max() + 1e-30))
    assert e < tol, (what, e)


@pytest.fixture(autouse=True)
def bf16_mode():
    from coarse3d_amd import ops
    prev = ops.matrix_precision_state()
    ops.set_matrix_precision("bf16")
    yield
    ops.set_matrix_precision(*prev)


def test_glue_kernels_bf16_vs_fp32_storage():
    from coarse3d_amd import ops
    g = torch.Generator(device=DEV).manual_seed(1)
    b, h, w, c = 2, 8, 64, 32
    x = r16(torch.randn(b, h, w, c, device=DEV, generator=g))
    a = r16(torch.randn(b, h, w, c, device=DEV, generator=g))
    sc, sh = torch.rand(c, device=DEV, generator=g) + 0.5, torch.randn(c, device=DEV, generator=g) * 0.2
    xb, ab = x.bfloat16(), a.bfloat16()
    # affine_add (incl. mixed layouts)
    ref = ops.affine_add(x, a, sc, sh)
    got = ops.affine_add(xb, ab, sc, sh)
    assert got.dtype == torch.bfloat16
    close16(got, ref, "affine_add")
    close16(ops.affine_add(x, ab, sc, sh, out=torch.empty_like(ab)), ref, "affine_add mixed")
    # axpy
    y32, y16 = a.clone(), ab.clone()
    ops.axpy(x, y32)
    ops.axpy(xb, y16)
    close16(y16, y32, "axpy")
    # maskpool / backward
    mask = (torch.rand(b, c, device=DEV, generator=g) > 0.2).float() * 1.25
    for pool in (True, False):
        close16(ops.maskpool(xb, mask, pool), ops.maskpool(x, mask, pool), f"maskpool {pool}")
        ho, wo = ((h + 1) // 2, (w + 1) // 2) if pool else (h, w)
        d = r16(torch.randn(b, ho, wo, c, device=DEV, generator=g))
        close16(ops.maskpool_bwd(d.bfloat16(), mask, ab, (b, h, w, c), pool), ops.maskpool_bwd(d, mask, a, (b, h, w, c), pool),
                f"maskpool_bwd {pool}")
    # pixshuf_cat / backward
    xa = r16(torch.randn(b, h // 2, w // 2, 64, device=DEV, generator=g))
    skip = r16(torch.randn(b, h, w, 16, device=DEV, generator=g))
    sc2, sh2 = torch.rand(64, device=DEV, generator=g) + 0.5, torch.randn(64, device=DEV, generator=g) * 0.2
    m3 = (torch.rand(b, 64, device=DEV, generator=g) > 0.2).float() * 1.25
    m1 = (torch.rand(b, 16, device=DEV, generator=g) > 0.2).float() * 1.25
    m2 = (torch.rand(b, 32, device=DEV, generator=g) > 0.2).float() * 1.25
    ref = ops.pixshuf_cat(xa, sc2, sh2, m3, m1, m2, skip)
    close16(ops.pixshuf_cat(xa.bfloat16(), sc2, sh2, m3, m1, m2, skip.bfloat16()), ref, "pixshuf_cat")
    dout = r16(torch.randn(b, h, w, 32, device=DEV, generator=g))
    ds32, ds16 = torch.zeros_like(skip), torch.zeros(skip.shape, device=DEV, dtype=torch.bfloat16)
    dx32 = ops.pixshuf_cat_bwd(dout, m3, m1, m2, tuple(xa.shape), 16, ds32, False)
    dx16 = ops.pixshuf_cat_bwd(dout.bfloat16(), m3, m1, m2, tuple(xa.shape), 16, ds16, False)
    close16(dx16, dx32, "pixshuf_cat_bwd dxa")
    close16(ds16, ds32, "pixshuf_cat_bwd dskip")
    # bilinear: bf16 -> bf16 into a wider tensor, bf16 -> fp32 (the embedding's exit), backward from fp32
    dst32 = torch.zeros(b, 2 * h, 2 * w, 48, device=DEV)
    dst16 = torch.zeros(b, 2 * h, 2 * w, 48, device=DEV, dtype=torch.bfloat16)
    ops.bilinear(x, 2 * h, 2 * w, dst=dst32, dcoff=16, c=c)
    ops.bilinear(xb, 2 * h, 2 * w, dst=dst16, dcoff=16, c=c)
    close16(dst16, dst32, "bilinear")
    close32(ops.bilinear(xb, 2 * h, 2 * w, out_dtype=torch.float32), ops.bilinear(x, 2 * h, 2 * w), "bilinear bf16->fp32")
    dd = torch.randn(b, 2 * h, 2 * w, c, device=DEV, generator=g)
    close16(ops.bilinear_bwd(torch.empty_like(xb), dd), ops.bilinear_bwd(torch.empty_like(x), dd), "bilinear_bwd fp32->bf16")
    # l2norm / backward
    y32, n32 = ops.l2norm(x)
    y16, n16 = ops.l2norm(xb)
    close16(y16, y32, "l2norm")
    close32(n16, n32, "l2norm norm")
    dy = r16(torch.randn(b, h, w, c, device=DEV, generator=g))
    close16(ops.l2norm_bwd(r16(y32).bfloat16(), n32, dy.bfloat16()), ops.l2norm_bwd(r16(y32), n32, dy), "l2norm_bwd")
    # first conv 5 -> 32 and its weight gradient
    xin = torch.randn(b, 5, h, w, device=DEV, generator=g)
    w5, b5 = torch.randn(32, 5, device=DEV, generator=g) * 0.3, torch.randn(32, device=DEV, generator=g) * 0.1
    o16 = ops.conv_in5(xin, w5, b5)
    assert o16.dtype == torch.bfloat16
    ops.set_matrix_precision("bf16", storage="f32")
    o32 = ops.conv_in5(xin, w5, b5)
    assert o32.dtype == torch.float32
    close16(o16, o32, "conv_in5")
    dz = r16(torch.randn(b, h, w, 32, device=DEV, generator=g))
    close32(ops.conv_in5_wgrad(xin, dz.bfloat16(), torch.empty(32, 5, device=DEV)),
            ops.conv_in5_wgrad(xin, dz, torch.empty(32, 5, device=DEV)), "conv_in5_wgrad")


def test_batchnorm_backward_bf16_vs_fp32_storage():
    from coarse3d_amd import ops
    g = torch.Generator(device=DEV).manual_seed(2)
    b, h, w, c = 2, 16, 64, 64
    dy = r16(torch.randn(b, h, w, c, device=DEV, generator=g))
    a = r16(torch.randn(b, h, w, c, device=DEV, generator=g))
    sc, sh = torch.rand(c, device=DEV, generator=g) + 0.5, torch.randn(c, device=DEV, generator=g) * 0.2
    k = torch.randn(3, c, device=DEV, generator=g) * 0.3
    for mode in (0, 1, 2):
        pre = (sc, sh) if mode == 1 else (None, None)
        close32(ops.bn_bwd_reduce(dy.bfloat16(), a.bfloat16(), c, mode, *pre), ops.bn_bwd_reduce(dy, a, c, mode, *pre),
                f"bn_bwd_reduce mode {mode}", 1e-5)
        dz32, p32 = ops.bn_bwd_apply(dy, a, c, mode, k if mode < 2 else None, *pre)
        dz16, p16 = ops.bn_bwd_apply(dy.bfloat16(), a.bfloat16(), c, mode, k if mode < 2 else None, *pre)
        assert dz16.dtype == torch.bfloat16
        close16(dz16, dz32, f"bn_bwd_apply mode {mode}")
        close32(p16[:, 0], p32[:, 0], f"bn_bwd_apply sum(dz) mode {mode}", 1e-5)     # statistics come from the fp32 values


@pytest.mark.parametrize("k,dil,pad,cin,cout", [(1, 1, 0, 64, 48), (3, 1, 1, 32, 64), (3, 2, 2, 32, 32), (2, 2, 1, 64, 64)])
def test_conv_engine_bf16_vs_fp32_storage(k, dil, pad, cin, cout):
    """conv_bfp NP = 1 with bf16 sources / bf16 output / accumulate, and the weight gradient with bf16 x
    and dz, against the same kernels on fp32 tensors holding the same values."""
    from coarse3d_amd import ops
    g = torch.Generator(device=DEV).manual_seed(3)
    b, h, w = 2, 16, 95
    x = r16(torch.randn(b, h, w, cin, device=DEV, generator=g))
    sc, sh = torch.rand(cin, device=DEV, generator=g) + 0.5, torch.randn(cin, device=DEV, generator=g) * 0.2
    wt = torch.randn(cout, cin, k, k, device=DEV, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, device=DEV, generator=g) * 0.1
    taps = ops.conv_taps(k, k, dil, pad)
    wp = ops.pack_weights(wt, 0)
    y32, p32 = ops.conv_forward([ops.Source(x, sc, sh)], wp, bias, cout, taps, lrelu=True, stats=True)
    y16, p16 = ops.conv_forward([ops.Source(x.bfloat16(), sc, sh)], wp, bias, cout, taps, lrelu=True, stats=True)
    assert y16.dtype == torch.bfloat16 and y32.dtype == torch.float32
    close16(y16, y32, "conv out")
    close32(p16, p32, "conv statistics partials", 1e-6)                              # taken before the rounding
    # input gradient (transposed weights), accumulating into an existing bf16 gradient
    cp = (cout + 15) // 16 * 16
    dz = torch.zeros(b, h, w, cp, device=DEV)
    dz[..., :cout] = r16(torch.randn(b, h, w, cout, device=DEV, generator=g))
    wd = ops.pack_weights(wt, 1, c_off=0, c_cnt=cin, kpad=cp)
    g0 = r16(torch.randn(b, h, w, cin, device=DEV, generator=g))
    g32, g16 = g0.clone(), g0.bfloat16()
    ops.conv_forward([ops.Source(dz)], wd, None, cin, ops.negate_taps(taps), out=g32, accumulate=True)
    ops.conv_forward([ops.Source(dz.bfloat16())], wd, None, cin, ops.negate_taps(taps), out=g16, accumulate=True)
    close16(g16, g32, "dgrad accumulate")
    # weight gradient: an fp32 result
    dw32, dw16 = torch.zeros_like(wt), torch.zeros_like(wt)
    ops.conv_wgrad(ops.Source(x, sc, sh), dz, dw32, taps)
    ops.conv_wgrad(ops.Source(x.bfloat16(), sc, sh), dz.bfloat16(), dw16, taps)
    close32(dw16, dw32, "wgrad", 1e-5)


@pytest.mark.parametrize("b,h,w,srcs,cout,dil,acc", [
    (2, 16, 95, (32,), 64, 1, False),       # ragged last pixel tile
    (2, 16, 64, (32,), 32, 2, False),       # 32-cout tiles, dilation 2
    (1, 8, 160, (64, 32), 64, 2, False),    # two sources (the UpBlock skip read in place)
    (2, 24, 64, (48,), 80, 1, False),       # ragged last cout tile, three chunks
    (4, 64, 256, (32,), 64, 1, True),       # >= 192 workgroups: 64-cout tiles; accumulating launch
    (1, 16, 128, (160,), 64, 1, False),     # ten chunks
])
def test_fused_nine_tap_kernel_with_one_plane_is_bit_identical(b, h, w, srcs, cout, dil, acc):
    """The bf16 engine's nine-tap convs over bf16 tensors run conv_x3f_kernel<..., 1, true> (round 4: staging dealt into the MFMA
    stream, one plane); c3d_conv_desc.variant & 4 keeps the phased conv_bfp kernel.  Same rounding points, same accumulation
    order: outputs and statistics partials must be the same bits."""
    from coarse3d_amd import ops
    g = torch.Generator(device=DEV).manual_seed(11)
    xs = [torch.randn(b, h, w, c, device=DEV, generator=g).bfloat16() for c in srcs]
    aff = [(torch.rand(c, device=DEV, generator=g) + 0.5, torch.randn(c, device=DEV, generator=g) * 0.2) for c in srcs]
    cin = sum(srcs)
    wt = torch.randn(cout, cin, 3, 3, device=DEV, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, device=DEV, generator=g) * 0.1
    taps = ops.conv_taps(3, 3, dil, dil)
    wp = ops.pack_weights(wt, 0)
    assert wp.c3d_planes
    sources = [ops.Source(x, sc, sh, lrelu=(i == 0)) for i, (x, (sc, sh)) in enumerate(zip(xs, aff))]
    out0 = torch.randn(b, h, w, cout, device=DEV, generator=g).bfloat16()
    res = {}
    for variant in (0, 4):
        ops.CONV_VARIANT = variant
        try:
            out = out0.clone()
            y, p = ops.conv_forward(sources, wp, bias, cout, taps, lrelu=True, stats=True, out=out, accumulate=acc)
        finally:
            ops.CONV_VARIANT = 0
        res[variant] = (y.clone(), p.clone())
    assert torch.equal(res[0][0], res[4][0]), float((res[0][0].float() - res[4][0].float()).abs().max())
    assert torch.equal(res[0][1], res[4][1])
    assert bool(torch.isfinite(res[0][0].float()).all())


@pytest.mark.parametrize("b,h,w,srcs,cout,acc", [
    (2, 16, 95, (64,), 96, False),          # ragged pixel tile, 128-cout tiles with a ragged last sub-tile
    (1, 8, 160, (64, 128), 704, False),     # two sources, 256-cout tiles (256 + 256 + 192), twelve chunks
    (2, 8, 64, (48,), 160, False),          # three chunks: the trip count is rounded up to four (a chunk of zeros)
    (1, 16, 128, (704,), 256, True),        # 44 chunks, accumulating launch (an input gradient into an existing one)
    (4, 32, 128, (16,), 272, False),        # one chunk
])
def test_wide_pointwise_kernel_with_four_chunks_in_flight_is_bit_identical(b, h, w, srcs, cout, acc):
    """The bf16 engine's 1x1 convs with more than 64 outputs over bf16 tensors run conv_pw1_kernel (round 4: four K chunks in
    flight in raw registers, buffer loads, affine table in LDS); c3d_conv_desc.variant & 3 == 3 keeps round 2's phased
    conv_pw3_kernel<NT, 1>.  Same rounding points and accumulation order: the same bits."""
    from coarse3d_amd import ops
    g = torch.Generator(device=DEV).manual_seed(17)
    xs = [torch.randn(b, h, w, c, device=DEV, generator=g).bfloat16() for c in srcs]
    cin = sum(srcs)
    wt = torch.randn(cout, cin, 1, 1, device=DEV, generator=g) / cin ** 0.5
    bias = torch.randn(cout, device=DEV, generator=g) * 0.1
    taps = [(0, 0)]
    wp = ops.pack_weights(wt, 0)
    assert wp.c3d_planes
    sources = []
    for i, x in enumerate(xs):
        if i == 0:       # BatchNorm affine + LeakyReLU on load
            sources.append(ops.Source(x, torch.rand(x.shape[3], device=DEV, generator=g) + 0.5,
                                      torch.randn(x.shape[3], device=DEV, generator=g) * 0.2, lrelu=True))
        else:            # a plain tensor
            sources.append(ops.Source(x))
    out0 = torch.randn(b, h, w, cout, device=DEV, generator=g).bfloat16()
    res = {}
    for variant in (0, 3):
        ops.CONV_VARIANT = variant
        try:
            out = out0.clone()
            y, p = ops.conv_forward(sources, wp, bias, cout, taps, lrelu=True, stats=True, out=out, accumulate=acc)
        finally:
            ops.CONV_VARIANT = 0
        res[variant] = (y.clone(), p.clone())
    assert torch.equal(res[0][0], res[3][0]), float((res[0][0].float() - res[3][0].float()).abs().max())
    assert torch.equal(res[0][1], res[3][1])
    assert bool(torch.isfinite(res[0][0].float()).all())


@pytest.mark.parametrize("k,dil,srcs,cout", [(1, 1, (64, 128), 704), (1, 1, (32,), 48), (3, 2, (32,), 64), (3, 1, (48,), 80), (2, 2, (64,), 64)])
def test_bf16_output_stores_match_the_fp32_output_of_the_same_kernel(k, dil, srcs, cout):
    """bf16 outputs leave the conv epilogue as 4-byte stores (the lanes of a cout pair swap one value per register pair); the
    same launch with an fp32 output tensor takes the plain stores.  Equal up to the final rounding, every element in place --
    fresh and accumulating launches, ragged pixel and cout tiles."""
    from coarse3d_amd import ops
    g = torch.Generator(device=DEV).manual_seed(23)
    b, h, w = 2, 16, 95
    xs = [torch.randn(b, h, w, c, device=DEV, generator=g).bfloat16() for c in srcs]
    cin = sum(srcs)
    wt = torch.randn(cout, cin, k, k, device=DEV, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, device=DEV, generator=g) * 0.1
    taps = ops.conv_taps(k, k, dil, dil if k == 3 else (1 if k == 2 else 0))
    wp = ops.pack_weights(wt, 0)
    sources = [ops.Source(x, torch.rand(x.shape[3], device=DEV, generator=g) + 0.5, torch.randn(x.shape[3], device=DEV, generator=g) * 0.2)
               for x in xs]
    for acc in (False, True):
        o0 = r16(torch.randn(b, h, w, cout + 16, device=DEV, generator=g))        # a wider tensor: channel offset 16
        o32, o16 = o0.clone(), o0.bfloat16()
        ops.conv_forward(sources, wp, bias, cout, taps, lrelu=True, out=o32, out_coff=16, accumulate=acc)
        ops.conv_forward(sources, wp, bias, cout, taps, lrelu=True, out=o16, out_coff=16, accumulate=acc)
        close16(o16[..., 16:], o32[..., 16:], f"bf16 stores, accumulate={acc}")
        assert torch.equal(o16[..., :16].float(), o0[..., :16])                     # nothing written beside the slice


@pytest.mark.parametrize("k,dil,pad,cin,cout", [(1, 1, 0, 64, 64), (3, 2, 2, 32, 64), (2, 2, 1, 64, 64), (1, 1, 0, 192, 256), (3, 1, 1, 32, 32),
                                                (3, 2, 2, 64, 32)])
def test_weight_gradient_with_four_raw_tiles_in_flight_equals_the_two_deep_kernel(k, dil, pad, cin, cout):
    """bf16 tensors on both sides: wgrad_tr_kernel<1, ..., RAW = true> (round 4: the staged units stay the 8 bytes they are loaded
    as, four tiles in flight).  fp32 tensors holding the same bf16 values take the two-deep kernel, which rounds them to the same
    bf16 operands.  Same tiles in the same order: the same bits -- at a size whose strips are long enough for the steady-state
    loop of the producers (16 tiles per strip) as well as at a small one (prologue / tail only)."""
    from coarse3d_amd import ops
    g = torch.Generator(device=DEV).manual_seed(29)
    for (b, h, w) in ((8, 64, 1024), (2, 16, 95)):
        x = r16(torch.randn(b, h, w, cin, device=DEV, generator=g))
        cp = (cout + 15) // 16 * 16
        dz = r16(torch.randn(b, h, w, cp, device=DEV, generator=g))
        sc, sh = torch.rand(cin, device=DEV, generator=g) + 0.5, torch.randn(cin, device=DEV, generator=g) * 0.2
        taps = ops.conv_taps(k, k, dil, pad)
        dw32 = torch.zeros(cout, cin, k, k, device=DEV)
        dw16 = torch.zeros_like(dw32)
        ops.conv_wgrad(ops.Source(x, sc, sh, lrelu=True), dz, dw32, taps)
        ops.conv_wgrad(ops.Source(x.bfloat16(), sc, sh, lrelu=True), dz.bfloat16(), dw16, taps)
        assert torch.equal(dw16, dw32), ((b, h, w), float((dw16 - dw32).abs().max() / dw32.abs().max()))
        assert float(dw32.abs().max()) > 0
        if k in (1, 3):  # round 5: eight producer waves in the small 1x1 instances (and, nine taps onto <= 32 couts, eight + eight waves with the
            # taps split across the consumers); c3d_wgrad_desc.variant & 128 keeps four + four: same bits
            dw4 = torch.zeros_like(dw32)
            ops.WGRAD_VARIANT = 128
            try:
                ops.conv_wgrad(ops.Source(x.bfloat16(), sc, sh, lrelu=True), dz.bfloat16(), dw4, taps)
            finally:
                ops.WGRAD_VARIANT = 0
            assert torch.equal(dw4, dw16)


@pytest.mark.parametrize("k,dil,pad,srcs,cout,acc", [
    (1, 1, 0, (64,), 64, False), (1, 1, 0, (192,), 64, False), (1, 1, 0, (64, 128), 48, False), (1, 1, 0, (96,), 32, False),
    (1, 1, 0, (48,), 32, False), (1, 1, 0, (704,), 64, True), (2, 2, 1, (64,), 64, False), (2, 2, 1, (128,), 48, True), (2, 2, 1, (48,), 32, False),
])
def test_deeper_chunks_of_raw_bf16_match_the_plain_kernel(k, dil, pad, srcs, cout, acc):
    """The bf16 engine's 1x1 / four-tap convs with <= 64 outputs over bf16 tensors stage their input as raw bf16 in K chunks of
    64 / 32 channels (conv_bfp_kernel<..., BFS = true>; c3d_conv_desc.variant & 8: the 32 / 16-channel chunks widened at load).
    1x1: the same k order, the same bits.  Four taps: the k steps of a chunk run tap by tap, so the fp32 accumulation order
    differs -- equal to fp32 rounding, far inside the bf16 rounding of the stored result."""
    from coarse3d_amd import ops
    g = torch.Generator(device=DEV).manual_seed(53)
    b, h, w = 2, 16, 95
    xs = [torch.randn(b, h, w, c, device=DEV, generator=g).bfloat16() for c in srcs]
    cin = sum(srcs)
    wt = torch.randn(cout, cin, k, k, device=DEV, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, device=DEV, generator=g) * 0.1
    taps = ops.conv_taps(k, k, dil, pad)
    wp = ops.pack_weights(wt, 0)
    sources = [ops.Source(x, torch.rand(x.shape[3], device=DEV, generator=g) + 0.5, torch.randn(x.shape[3], device=DEV, generator=g) * 0.2,
                          lrelu=(i == 0)) for i, x in enumerate(xs)]
    out0 = r16(torch.randn(b, h, w, cout, device=DEV, generator=g))        # (accumulating launches: the same old values in both layouts)
    res = {}
    for variant in (0, 8):
        ops.CONV_VARIANT = variant
        try:
            o32, o16 = out0.clone(), out0.bfloat16()
            _, p32 = ops.conv_forward(sources, wp, bias, cout, taps, lrelu=True, stats=True, out=o32, accumulate=acc)
            ops.conv_forward(sources, wp, bias, cout, taps, lrelu=True, out=o16, accumulate=acc)
        finally:
            ops.CONV_VARIANT = 0
        res[variant] = (o32.clone(), o16.clone(), p32.clone())
    if k == 1:
        for a_, b_ in zip(res[0], res[8]):
            assert torch.equal(a_, b_)
    else:
        close32(res[0][0], res[8][0], "four taps, fp32 output", 2e-6)
        close32(res[0][2], res[8][2], "four taps, statistics", 2e-6)
        close16(res[0][1], res[8][0], "four taps, bf16 output")


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,dil,pad,acc", [
    (2, 16, 128, 32, 32, 3, 1, 1, False), (2, 16, 128, 64, 64, 3, 2, 2, True), (2, 16, 128, 64, 64, 2, 2, 1, False),
    (2, 16, 128, 128, 128, 2, 2, 1, True), (2, 16, 128, 32, 64, 1, 1, 0, False), (2, 16, 128, 192, 64, 1, 1, 0, True),
    (4, 32, 512, 384, 128, 1, 1, 0, False), (2, 16, 100, 64, 32, 3, 1, 1, True), (1, 24, 72, 64, 64, 2, 2, 1, False),
])
def test_batchnorm_backward_sums_in_the_input_gradient_epilogue_over_bf16_tensors(B, H, W, Cin, Cout, k, dil, pad, acc):
    """Round 5 (VERDICT round 4, next #2 ii): ConvArgs::stat_mul for the bf16 engine.  The LAST input-gradient launch into a
    conv -> LeakyReLU -> BatchNorm layer's output leaves (sum dy, sum dy * a) per tile in its epilogue -- from the values AS
    STORED (rounded to bf16), i.e. exactly what c3d_bn_bwd_reduce would read in a pass of its own (BASELINE configs[2]:
    43 such passes, 1.1 ms of a 16.8 ms step).  Every kernel family of that engine over bf16 tensors (nine taps, four taps,
    narrow and 128-wide 1x1), with and without accumulation into an existing gradient, ragged tiles: the stored gradient is
    bit-identical to the launch without the epilogue, the folded sums agree with the separate pass to fp32 summation order."""
    from coarse3d_amd import ops
    g = torch.Generator(device=DEV).manual_seed(Cin + 3 * Cout + k)
    # an input-gradient launch: source = dz [.., Cin], output = the gradient [.., Cout] at a BatchNorm's output whose input is a
    dz = torch.randn(B, H, W, Cin, device=DEV, generator=g).bfloat16()
    a = torch.randn(B, H, W, Cout, device=DEV, generator=g).bfloat16()
    w = torch.randn(Cin, Cout, k, k, device=DEV, generator=g) * 0.1          # the forward layer's weight: Cout_fwd = Cin here
    taps = ops.negate_taps(ops.conv_taps(k, k, dil, pad))
    kp = (Cin + 15) // 16 * 16
    wd = ops.pack_weights(w, 1, c_off=0, c_cnt=Cout, kpad=kp)
    old = torch.randn(B, H, W, Cout, device=DEV, generator=g).bfloat16()
    outs = []
    for fused in (False, True):
        out = old.clone() if acc else torch.empty_like(old)
        part = torch.empty(Cout, 2, ops.num_mtiles(B, H, W), device=DEV, dtype=torch.float32) if fused else None
        _, p = ops.conv_forward([ops.Source(dz)], wd, None, Cout, taps, out=out, accumulate=acc, grad=True,
                                stat_partial=part, stat_mul=a if fused else None, stat_mul_optional=True)
        outs.append((out, p))
    torch.cuda.synchronize()
    assert outs[1][1] is not None, "this launch should have the epilogue"
    assert torch.equal(outs[0][0], outs[1][0])
    got = ops.stat_reduce(outs[1][1], Cout)
    ref = ops.stat_reduce(ops.bn_bwd_reduce(outs[0][0], a, Cout, 0), Cout)
    scale = ref.abs().max(dim=0).values.clamp_min(1e-30)
    assert float(((got - ref).abs() / scale).max()) < 2e-6, float(((got - ref).abs() / scale).max())
    # float64 restatement of the two sums from the stored tensors
    dy64, a64 = outs[0][0].double(), a.double()
    want = torch.stack([dy64.sum(dim=(0, 1, 2)), (dy64 * a64).sum(dim=(0, 1, 2))], 1)
    assert float(((got - want).abs() / scale).max()) < 2e-5


def test_stat_mul_is_refused_where_the_kernel_has_no_epilogue():
    from coarse3d_amd import ops
    g = torch.Generator(device=DEV).manual_seed(2)
    B, H, W, Cin, Cout = 2, 4, 64, 64, 64                     # 4-row image: 4-row tiles, the widening kernel
    dz = torch.randn(B, H, W, Cin, device=DEV, generator=g).bfloat16()
    a = torch.randn(B, H, W, Cout, device=DEV, generator=g).bfloat16()
    w = torch.randn(Cin, Cout, 3, 3, device=DEV, generator=g) * 0.1
    wd = ops.pack_weights(w, 1, c_off=0, c_cnt=Cout, kpad=64)
    taps = ops.negate_taps(ops.conv_taps(3, 3, 1, 1))
    part = torch.empty(Cout, 2, ops.num_mtiles(B, H, W), device=DEV)
    out = torch.empty_like(a)
    _, p = ops.conv_forward([ops.Source(dz)], wd, None, Cout, taps, out=out, grad=True, stat_partial=part, stat_mul=a,
                            stat_mul_optional=True)
    assert p is None                                          # the caller runs c3d_bn_bwd_reduce
    with pytest.raises(ValueError):
        ops.conv_forward([ops.Source(dz)], wd, None, Cout, taps, out=out, grad=True, stat_partial=part, stat_mul=a)
