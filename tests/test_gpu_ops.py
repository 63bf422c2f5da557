"""Kernel-level parity of the BatchNorm / glue HIP ops against plain PyTorch fp32 on CPU
(the same ops the oracle is built from).  Tolerance: 1e-4 of max|ref| unless stated."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2)


@pytest.mark.parametrize("B,H,W,C", [(2, 64, 128, 32), (2, 8, 16, 256), (1, 3, 113, 64), (2, 16, 32, 704)])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_bn_backward(B, H, W, C, mode):
    """dz of  z -> LeakyReLU -> BN  (mode 0),  z -> BN -> LeakyReLU (mode 1),  z -> LeakyReLU (2)."""
    from coarse3d_amd import ops
    g = torch.Generator().manual_seed(C + mode)
    z = torch.randn(B, C, H, W, generator=g).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g) * 0.2).requires_grad_(True)
    dy = torch.randn(B, C, H, W, generator=g)
    if mode == 0:
        a = F.leaky_relu(z, 0.01)
        y = F.batch_norm(a, None, None, gamma, beta, True, 0.1, 1e-5)
    elif mode == 1:
        a = z
        y = F.leaky_relu(F.batch_norm(z, None, None, gamma, beta, True, 0.1, 1e-5), 0.01)
    else:
        a = F.leaky_relu(z, 0.01)
        y = a
    y.backward(dy)
    a_d = nhwc(a.detach()).to(DEV)
    dy_d = nhwc(dy).to(DEV)
    n = B * H * W
    if mode == 2:
        dz, pz = ops.bn_bwd_apply(dy_d, a_d, C, 2)
    else:
        mean = a.detach().mean(dim=(0, 2, 3))
        var = a.detach().var(dim=(0, 2, 3), unbiased=False)
        invstd = 1.0 / torch.sqrt(var + 1e-5)
        scale = gamma.detach() * invstd
        shift = beta.detach() - mean * scale
        pre = (scale.to(DEV), shift.to(DEV)) if mode == 1 else (None, None)
        part = ops.bn_bwd_reduce(dy_d, a_d, C, mode, *pre)
        sums = ops.stat_reduce(part, C)
        dgamma = torch.empty(C, device=DEV)
        dbeta = torch.empty(C, device=DEV)
        k = ops.bn_bwd_coeffs(sums, n, mean.to(DEV), invstd.to(DEV), gamma.detach().to(DEV), dgamma, dbeta)
        dz, pz = ops.bn_bwd_apply(dy_d, a_d, C, mode, k, *pre)
        assert rel(dgamma, gamma.grad) < 1e-4
        assert rel(dbeta, beta.grad) < 1e-4
    assert rel(nchw(dz), z.grad) < 1e-4
    db = torch.empty(C, device=DEV)
    ops.sums_to_f32(ops.stat_reduce(pz, C), 0, db)
    if mode == 1:   # a bias in front of a BatchNorm has exactly zero gradient: both are rounding noise
        assert float(db.abs().max()) < 1e-3 * float(z.grad.abs().sum(dim=(0, 2, 3)).max())
    else:
        assert rel(db, z.grad.sum(dim=(0, 2, 3))) < 1e-4


def test_bn_forward_stats_and_running():
    from coarse3d_amd import ops
    g = torch.Generator().manual_seed(3)
    B, C, H, W = 2, 64, 16, 64
    x = torch.randn(B, C, H, W, generator=g) * 2 + 0.5
    bn = torch.nn.BatchNorm2d(C)
    bn.weight.data = torch.rand(C, generator=g) + 0.5
    bn.bias.data = torch.randn(C, generator=g)
    bn.train()
    y = bn(x)
    # per-tile partials as the conv epilogue would leave them: emulate with one "tile" per image row
    xd = nhwc(x).to(DEV)
    part = torch.stack([xd.reshape(-1, W, C).sum(1).t(), (xd.reshape(-1, W, C) ** 2).sum(1).t()], 1).contiguous()
    sums = ops.stat_reduce(part, C)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    sc, sh, mean, invstd = ops.bn_finalize(sums, B * H * W, bn.weight.data.to(DEV), bn.bias.data.to(DEV), rm, rv)
    yd = xd * sc + sh
    assert rel(nchw(yd), y) < 1e-4
    assert rel(rm, bn.running_mean) < 1e-4 and rel(rv, bn.running_var) < 1e-4
    bn.eval()
    sc2, sh2 = ops.bn_eval_affine(bn.weight.data.to(DEV), bn.bias.data.to(DEV), rm, rv)
    assert rel(nchw(xd * sc2 + sh2), bn(x)) < 1e-4


def test_input_norm_and_first_conv():
    from coarse3d_amd import ops
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 16, 80
    x = torch.randn(B, 5, H, W, generator=g)
    ev = torch.randint(0, 3, (B, H, W), generator=g)
    mean, std = torch.randn(5, generator=g), torch.rand(5, generator=g) + 0.5
    ref = (x - mean[None, :, None, None]) / std[None, :, None, None] * (ev > 0).unsqueeze(1)
    xn = ops.input_norm(x.to(DEV), ev.to(DEV), mean.to(DEV), std.to(DEV))
    assert rel(xn, ref) < 1e-6
    w = (torch.randn(32, 5, 1, 1, generator=g) * 0.4).requires_grad_(True)
    bias = torch.randn(32, generator=g)
    s = F.leaky_relu(F.conv2d(ref, w, bias), 0.01)
    sd = ops.conv_in5(xn, w.detach().reshape(32, 5).contiguous().to(DEV), bias.to(DEV))
    assert rel(nchw(sd), s) < 1e-5
    dz = torch.randn(B, 32, H, W, generator=g)
    F.conv2d(ref, w, bias).backward(dz)
    dw = torch.empty(32, 5, 1, 1, device=DEV)
    ops.conv_in5_wgrad(xn, nhwc(dz).to(DEV), dw)
    assert rel(dw, w.grad) < 1e-4


@pytest.mark.parametrize("pool", [True, False])
@pytest.mark.parametrize("H,W", [(16, 32), (3, 113), (2, 8)])
def test_maskpool(pool, H, W):
    from coarse3d_amd import ops
    g = torch.Generator().manual_seed(H * W)
    B, C = 2, 64
    x = torch.randn(B, C, H, W, generator=g).requires_grad_(True)
    mask = (torch.rand(B, C, generator=g) > 0.2).float() * 1.25
    extra = torch.randn(B, C, H, W, generator=g)
    y = x * mask[:, :, None, None]
    if pool:
        y = F.avg_pool2d(y, 3, 2, 1)
    dy = torch.randn(y.shape, generator=g)
    (y * dy).sum().backward()
    yd = ops.maskpool(nhwc(x.detach()).to(DEV), mask.to(DEV), pool)
    assert rel(nchw(yd), y) < 1e-5
    dx = ops.maskpool_bwd(nhwc(dy).to(DEV), mask.to(DEV), nhwc(extra).to(DEV), (B, H, W, C), pool)
    assert rel(nchw(dx), x.grad + extra) < 1e-5


def test_pixshuf_cat():
    from coarse3d_amd import ops
    g = torch.Generator().manual_seed(9)
    B, Hs, Ws, Cx, Cs = 2, 4, 8, 64, 32
    xa = torch.randn(B, Cx, Hs, Ws, generator=g).requires_grad_(True)
    skip = torch.randn(B, Cs, 2 * Hs, 2 * Ws, generator=g).requires_grad_(True)
    sc, sh = torch.rand(Cx, generator=g) + 0.5, torch.randn(Cx, generator=g)
    m3 = (torch.rand(B, Cx, generator=g) > 0.2).float() * 1.25
    m1 = (torch.rand(B, Cx // 4, generator=g) > 0.2).float() * 1.25
    m2 = (torch.rand(B, Cx // 4 + Cs, generator=g) > 0.2).float() * 1.25
    yb = (xa * sc[None, :, None, None] + sh[None, :, None, None])
    yb.retain_grad()
    u = F.pixel_shuffle(yb * m3[:, :, None, None], 2) * m1[:, :, None, None]
    out = torch.cat((u, skip), 1) * m2[:, :, None, None]
    dout = torch.randn(out.shape, generator=g)
    (out * dout).sum().backward()
    od = ops.pixshuf_cat(nhwc(xa.detach()).to(DEV), sc.to(DEV), sh.to(DEV), m3.to(DEV), m1.to(DEV), m2.to(DEV),
                         nhwc(skip.detach()).to(DEV))
    assert rel(nchw(od), out) < 1e-5
    dskip = torch.ones(B, 2 * Hs, 2 * Ws, Cs, device=DEV)
    dxa = ops.pixshuf_cat_bwd(nhwc(dout).to(DEV), m3.to(DEV), m1.to(DEV), m2.to(DEV), (B, Hs, Ws, Cx), Cs, dskip, True)
    assert rel(nchw(dxa), yb.grad) < 1e-5
    assert rel(nchw(dskip), skip.grad + 1) < 1e-5


def test_softmax_crop():
    from coarse3d_amd import ops
    g = torch.Generator().manual_seed(11)
    B, H, W, C = 2, 16, 24, 14
    lg = (torch.randn(B, C, H, W, generator=g) * 3).requires_grad_(True)
    p = F.softmax(lg[:, :, :-8, :-8], 1)
    dp = torch.randn(p.shape, generator=g)
    (p * dp).sum().backward()
    l32 = torch.zeros(B, H, W, 32)
    l32[..., :C] = nhwc(lg.detach())
    pd = ops.softmax(l32.to(DEV), C, H - 8, W - 8)
    assert rel(nchw(pd), p) < 1e-5
    dl = ops.softmax_bwd(pd, nhwc(dp).to(DEV), (B, H, W, 32))
    assert rel(nchw(dl[..., :C]), lg.grad) < 1e-4
    assert float(dl[..., C:].abs().max()) == 0.0


@pytest.mark.parametrize("Hs,Ws,Hd,Wd", [(64, 128, 32, 64), (16, 32, 32, 64), (4, 8, 32, 64), (32, 64, 32, 64),
                                          (32, 64, 12, 24), (16, 28, 32, 56)])
def test_bilinear(Hs, Ws, Hd, Wd):
    from coarse3d_amd import ops
    g = torch.Generator().manual_seed(Hs + Wd)
    B, C = 2, 32
    x = torch.randn(B, C, Hs, Ws, generator=g).requires_grad_(True)
    y = F.interpolate(x, size=(Hd, Wd), mode="bilinear", align_corners=True)
    dy = torch.randn(y.shape, generator=g)
    (y * dy).sum().backward()
    dst = torch.zeros(B, Hd, Wd, C + 16, device=DEV)
    ops.bilinear(nhwc(x.detach()).to(DEV), Hd, Wd, dst=dst, dcoff=16, c=C)
    assert rel(nchw(dst[..., 16:]), y) < 1e-5
    dsrc = torch.full((B, Hs, Ws, C), 7.0, device=DEV)
    ddst = torch.zeros(B, Hd, Wd, C + 16, device=DEV)
    ddst[..., 16:] = nhwc(dy).to(DEV)
    ops.bilinear_bwd(dsrc, ddst, dcoff=16, c=C)
    assert rel(nchw(dsrc), x.grad) < 1e-4
    ops.bilinear_bwd(dsrc, ddst, dcoff=16, c=C, accumulate=True)
    assert rel(nchw(dsrc), 2 * x.grad) < 1e-4


def test_l2norm_and_residual():
    from coarse3d_amd import ops
    g = torch.Generator().manual_seed(13)
    x = torch.randn(500, 256, generator=g).requires_grad_(True)
    x.data[7] = 0
    y = F.normalize(x, p=2, dim=1)
    dy = torch.randn(500, 256, generator=g)
    (y * dy).sum().backward()
    yd, norm = ops.l2norm(x.detach().to(DEV))
    assert rel(yd, y) < 1e-6
    dx = ops.l2norm_bwd(yd, norm, dy.to(DEV))
    mask = torch.ones(500, dtype=torch.bool)
    mask[7] = False
    assert rel(dx[mask.to(DEV)], x.grad[mask]) < 1e-4
    a, s = torch.randn(2, 8, 16, 32, generator=g), torch.randn(2, 8, 16, 32, generator=g)
    sc, sh = torch.rand(32, generator=g), torch.randn(32, generator=g)
    out = ops.affine_add(s.to(DEV), a.to(DEV), sc.to(DEV), sh.to(DEV))
    assert rel(out, s + a * sc + sh) < 1e-6
    yy = torch.ones(2, 8, 16, 32, device=DEV)
    ops.axpy(a.to(DEV), yy, 0.5, True)
    assert rel(yy, 1 + 0.5 * a) < 1e-6


@pytest.mark.parametrize("A,D,pool_n", [(50, 256, 12), (512, 256, 7), (130, 96, 40), (512, 256, 300), (700, 64, 90)])
def test_scatter_add_rows_repeated_anchors_bit_reproducible(A, D, pool_n):
    """Gradient hand-over of the contrast loss (autograd of `feats[img, :, idx]` in contrast_pixel_loss.py): anchors
    are drawn with replacement, so pixels repeat inside an (image, class) pair.  The kernel sums a pixel's rows in
    ascending order in one wave (no atomics): exact against an ordered fp32 reference, identical across runs."""
    from coarse3d_amd import ops
    g = torch.Generator().manual_seed(17)
    B, n, tmax, tn = 2, 4096, 12, 9
    img = torch.zeros(tmax, dtype=torch.int32)
    idx = torch.zeros(tmax, A, dtype=torch.int32)
    for t in range(tn):                                  # pair t owns the pixels == t (mod tmax): disjoint across pairs
        img[t] = t % B
        pool = torch.arange(t, n, tmax)[torch.randperm(n // tmax, generator=g)[:pool_n]]   # weak labels: a handful of pixels per pair
        idx[t] = pool[torch.randint(0, pool.numel(), (A,), generator=g)].to(torch.int32)
    dx = torch.randn(tmax * A, D, generator=g)
    gs = torch.tensor([0.37])
    ref = torch.zeros(B, n, D)
    acc = {}
    for t in range(tn):
        for s in range(A):                               # ascending s, fp32 adds: the kernel's order
            key = (int(img[t]), int(idx[t, s]))
            acc[key] = dx[t * A + s].clone() if key not in acc else acc[key] + dx[t * A + s]
    for (b, p), v in acc.items():
        ref[b, p] = gs * v
    outs = []
    for _ in range(3):
        dfeat = torch.zeros(B, n, D, device=DEV)
        ops.scatter_add_rows(dx.to(DEV), img.to(DEV), idx.to(DEV), torch.tensor([tn], dtype=torch.int32, device=DEV),
                             tmax, A, n, dfeat, gs.to(DEV))
        outs.append(dfeat.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.equal(outs[0], ref)
    # accumulate semantics: a second call adds to what is there
    ops.scatter_add_rows(dx.to(DEV), img.to(DEV), idx.to(DEV), torch.tensor([tn], dtype=torch.int32, device=DEV),
                         tmax, A, n, dfeat, gs.to(DEV))
    assert rel(dfeat, 2 * ref) < 1e-6


def test_row_mask_hint_scatter_and_bilinear_adjoint():
    """The contrast loss' dense gradient carries a bitmap of its non-zero pixel rows (c3d_scatter_add_rows) which the
    bilinear adjoint may use to skip known-zero rows: identical result with and without it; the hint is handed over
    only for the very tensor it was made for."""
    from coarse3d_amd import contrast, ops
    g = torch.Generator().manual_seed(23)
    B, H, W, D, A, tmax, tn = 2, 16, 64, 64, 20, 6, 5
    n = H * W
    img = torch.zeros(tmax, dtype=torch.int32)
    idx = torch.zeros(tmax, A, dtype=torch.int32)
    for t in range(tn):
        img[t] = t % B
        pool = torch.arange(t, n, tmax)[torch.randperm(n // tmax, generator=g)[:9]]
        idx[t] = pool[torch.randint(0, 9, (A,), generator=g)].to(torch.int32)
    dx = torch.randn(tmax * A, D, generator=g).to(DEV)
    dfeat = torch.zeros(B, H, W, D, device=DEV)
    mask = torch.zeros((B * n + 31) // 32, dtype=torch.int32, device=DEV)
    ops.scatter_add_rows(dx, img.to(DEV), idx.to(DEV), torch.tensor([tn], dtype=torch.int32, device=DEV), tmax, A, n, dfeat,
                         rowmask=mask)
    bits = ((mask.cpu().view(-1, 1) >> torch.arange(32, dtype=torch.int32)) & 1).flatten()[:B * n].bool()
    assert torch.equal(bits, (dfeat.cpu().view(B * n, D) != 0).any(1))
    a = ops.bilinear_bwd(torch.empty(B, H // 2, W // 2, D, device=DEV), dfeat)
    b = ops.bilinear_bwd(torch.empty(B, H // 2, W // 2, D, device=DEV), dfeat, rowmask=mask)
    assert torch.equal(a, b) and float(a.abs().max()) > 0
    up = ops.bilinear_bwd(torch.empty(B, 4 * H, 4 * W, D, device=DEV), dfeat, rowmask=mask)      # a down-sampling adjoint too
    assert torch.equal(up, ops.bilinear_bwd(torch.empty(B, 4 * H, 4 * W, D, device=DEV), dfeat))
    # hand-over: only (views of) the published tensor, once, and only while unmodified
    contrast._publish_row_hint(dfeat, mask)
    assert contrast.take_row_hint(dfeat.clone()) is None and contrast.take_row_hint(dfeat) is None     # slot consumed
    contrast._publish_row_hint(dfeat, mask)
    assert contrast.take_row_hint(dfeat.permute(0, 3, 1, 2).permute(0, 2, 3, 1)) is mask
    contrast._publish_row_hint(dfeat, mask)
    dfeat.mul_(2.0)
    assert contrast.take_row_hint(dfeat) is None


@pytest.mark.parametrize("src_dtype", [torch.float32, torch.bfloat16])
def test_rows_of_the_upsampled_embedding_without_the_map(src_dtype):
    """contrast.LowResFeat's three kernels against the dense path they replace (salsanext_proto.py:488-490 followed by
    the row selections of contrast_pixel_loss.py / prototype_learning), bit for bit: rows interpolated on demand
    (c3d_bilinear_rows, both index forms, with and without the l2 normalisation, rows past the count zero) equal the
    rows of the materialised map; the compact gradient (c3d_scatter_rows_compact + c3d_bilinear_bwd_rows) equals the
    adjoint over the zero-filled dense gradient, repeated anchors included."""
    from coarse3d_amd import ops
    g = torch.Generator().manual_seed(5)
    b, hs, ws, d, H, W = 2, 16, 64, 256, 32, 128
    n = H * W
    low = torch.randn(b, hs, ws, d, generator=g).to(DEV).to(src_dtype)
    dense = ops.bilinear(low, H, W, out_dtype=torch.float32)
    # flat int64 pixels + a device count (the labelled pixels of prototype_learning)
    idx = torch.randint(0, b * n, (1000,), generator=g).to(DEV)
    rows = ops.bilinear_rows(low, H, W, idx, count=torch.tensor([900], device=DEV, dtype=torch.int32))
    assert rows.shape == (1024, d) and torch.equal(rows[:900], dense.view(b * n, d)[idx[:900]])
    assert float(rows[900:].abs().max()) == 0.0
    # (image, pixel) pairs, l2-normalised (the anchors of the contrast loss); pixel sets of different pairs are disjoint
    A, tmax, tn = 50, 12, 9
    img = (torch.arange(tmax) % b).to(torch.int32).to(DEV)
    aidx = torch.stack([torch.randint(p * 100, p * 100 + 40, (A,), generator=g) for p in range(tmax)]).to(torch.int32).to(DEV)
    t = torch.tensor([tn], device=DEV, dtype=torch.int32)
    o1, n1 = ops.bilinear_rows(low, H, W, aidx, img=img, a=A, count=t, l2=True)
    o2, n2 = ops.gather_rows_l2(dense, img, aidx, t, tmax, A, n)
    assert torch.equal(o1, o2) and torch.equal(n1, n2) and float(o1[: tn * A].abs().max()) > 0
    # gradient: dense scatter + adjoint (with and without the row bitmap) vs compact scatter + adjoint
    dx = torch.randn(tmax * A, d, generator=g).to(DEV)
    gs = torch.tensor([0.37], device=DEV)
    dfeat = torch.zeros(b, H, W, d, device=DEV)
    rm = torch.zeros((b * n + 31) // 32, device=DEV, dtype=torch.int32)
    ops.scatter_add_rows(dx, img, aidx, t, tmax, A, n, dfeat, gs, rowmask=rm)
    want = ops.bilinear_bwd(torch.empty_like(low), dfeat)
    drows, cmap, rm2 = ops.scatter_rows_compact(dx, img, aidx, t, tmax, A, n, b, gs)
    got = ops.bilinear_bwd_rows(torch.empty_like(low), drows, cmap, rm2, H, W)
    assert torch.equal(rm, rm2) and torch.equal(want, got) and float(got.float().abs().max()) > 0
    assert torch.equal(want, ops.bilinear_bwd(torch.empty_like(low), dfeat, rowmask=rm))
