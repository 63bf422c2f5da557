"""Engine features the RangeNet backbone needs on top of SalsaNext's, each against
plain PyTorch on the CPU at 1e-4 of max|ref|: LeakyReLU slope 0.1 on load / in the epilogue /
in the BatchNorm backward, 4-tap kernels with offsets up to 2, stride-(1,2) convolution as six taps over the
column-pair view of its input, ConvTranspose2d([1,4],[1,2],[0,1]) as three taps onto the column-pair view of its
output (3- / 6-tap launches of every engine), the LeakyReLU residual add and the 5 -> 16 channel input repack."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def test_bn_then_lrelu_conv_with_slope():
    """z -> BN -> LeakyReLU(0.1) -> conv3x3 (no bias): the consumer applies the affine and the
    activation while staging; forward, input gradient (BN backward mode 1) and weight gradient."""
    from coarse3d_amd import ops
    g = torch.Generator().manual_seed(0)
    B, H, W, Ci, Co = 2, 8, 64, 32, 48
    z = torch.randn(B, Ci, H, W, generator=g).requires_grad_(True)
    gamma, beta = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.2
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / 17).requires_grad_(True)
    mean, var = z.mean((0, 2, 3)), z.var((0, 2, 3), unbiased=False)
    y = F.leaky_relu((z - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + 1e-5) * gamma[None, :, None, None]
                     + beta[None, :, None, None], 0.1)
    out = F.conv2d(y, w, padding=1)
    dout = torch.randn(out.shape, generator=g)
    out.backward(dout)
    invstd = 1 / torch.sqrt(var.detach() + 1e-5)
    scale, shift = gamma * invstd, beta - mean.detach() * gamma * invstd
    zd = ops.to_nhwc(z.detach()).to(DEV)
    src = ops.Source(zd, scale.to(DEV), shift.to(DEV), lrelu=True)
    taps = ops.conv_taps(3, 3, 1, 1)
    got, _ = ops.conv_forward([src], ops.pack_weights(w.detach().to(DEV), 0), None, Co, taps, slope=0.1)
    assert rel(ops.from_nhwc(got.cpu()), out.detach()) < 1e-4
    dzo = ops.to_nhwc(dout).to(DEV)
    dw = torch.zeros_like(w.detach(), device=DEV)
    ops.conv_wgrad(src, dzo, dw, taps, slope=0.1)
    assert rel(dw.cpu(), w.grad) < 1e-4
    dy, _ = ops.conv_forward([ops.Source(dzo)], ops.pack_weights(w.detach().to(DEV), 1), None, Ci, ops.negate_taps(taps))
    # BatchNorm backward, "BN then LeakyReLU" order (mode 1), slope 0.1
    part = ops.bn_bwd_reduce(dy, zd, Ci, 1, scale.to(DEV), shift.to(DEV), slope=0.1)
    dgam, dbet = torch.empty(Ci, device=DEV), torch.empty(Ci, device=DEV)
    k = ops.bn_bwd_coeffs_partials(part, B * H * W, mean.detach().to(DEV), invstd.to(DEV), gamma.to(DEV), dgam, dbet)
    dz, _ = ops.bn_bwd_apply(dy, zd, Ci, 1, k, scale.to(DEV), shift.to(DEV), slope=0.1)
    assert rel(ops.from_nhwc(dz.cpu()), z.grad) < 1e-4


class _Packs:
    pass


def _pair_helper():
    from coarse3d_amd.rangenet import RangeNetBackbone
    bb = RangeNetBackbone.__new__(RangeNetBackbone)
    bb.packs = _Packs()
    return bb


ENGINES = [("bf16x3", None, 1e-4), ("f32", None, 1e-4), ("bf16", "f32", 2e-2)]


@pytest.mark.parametrize("mode,storage,tol", ENGINES)
@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 8, 64, 32, 64), (1, 64, 256, 32, 64), (2, 12, 96, 64, 128), (1, 4, 64, 128, 32)])
def test_strided_conv_as_six_taps_over_the_column_pair_view(mode, storage, tol, B, H, W, Ci, Co):
    """Conv2d(3x3, stride (1, 2), padding 1) of RangeNet's encoder (rangenet_proto.py:192-200) as a stride-1 conv with six taps
    over the [B, H, W/2, 2C] view of its NHWC input (round 5; coarse3d_amd/rangenet.py::_down): forward, input gradient and weight
    gradient against torch on the CPU, every engine, tile heights 8 / 4 / 2, with a pending BatchNorm affine + LeakyReLU(0.1)."""
    from coarse3d_amd import ops
    from coarse3d_amd.rangenet import DOWN_TAPS
    g = torch.Generator().manual_seed(1 + Ci + Co + H)
    z = torch.randn(B, Ci, H, W, generator=g).requires_grad_(True)
    sc, sh = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.2
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)).requires_grad_(True)
    x = F.leaky_relu(z * sc[None, :, None, None] + sh[None, :, None, None], 0.1)
    ref = F.conv2d(x, w, stride=(1, 2), padding=1)
    dout = torch.randn(ref.shape, generator=g)
    ref.backward(dout)
    bb = _pair_helper()
    prev = ops.matrix_precision_state()
    ops.set_matrix_precision(mode, storage=storage) if storage else ops.set_matrix_precision(mode)
    try:
        w2 = bb._pair_weight("l", "down", w.detach().to(DEV))
        v = ops.to_nhwc(z.detach()).to(DEV).view(B, H, W // 2, 2 * Ci)
        src = ops.Source(v, sc.repeat(2).to(DEV), sh.repeat(2).to(DEV), lrelu=True)
        got, _ = ops.conv_forward([src], ops.pack_weights(w2, 0), None, Co, DOWN_TAPS, slope=0.1)
        assert got.shape == (B, H, W // 2, Co) and rel(ops.from_nhwc(got.cpu()), ref.detach()) < tol
        dzo = ops.to_nhwc(dout).to(DEV)
        dw2 = torch.zeros_like(w2)
        ops.conv_wgrad(src, dzo, dw2, DOWN_TAPS, slope=0.1)
        dw = torch.empty_like(w.detach(), device=DEV)
        bb._pair_weight_grad("l", "down", w.detach().to(DEV), dw2, dw)
        assert rel(dw.cpu(), w.grad) < tol
        dv, _ = ops.conv_forward([ops.Source(dzo)], ops.pack_weights(w2, 1, c_off=0, c_cnt=2 * Ci, kpad=(Co + 15) // 16 * 16), None, 2 * Ci,
                                 ops.negate_taps(DOWN_TAPS), grad=True)
        dx = ops.from_nhwc(dv.view(B, H, W, Ci).cpu())            # gradient at the activated input
    finally:
        ops.set_matrix_precision(*prev)
    xr = x.detach().requires_grad_(True)
    F.conv2d(xr, w.detach(), stride=(1, 2), padding=1).backward(dout)
    assert rel(dx, xr.grad) < tol


@pytest.mark.parametrize("mode,storage,tol", ENGINES)
@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 8, 32, 64, 32), (1, 64, 128, 64, 32), (2, 12, 64, 128, 64), (1, 4, 32, 256, 128)])
def test_transposed_conv_as_three_taps_onto_the_column_pair_view(mode, storage, tol, B, H, W, Ci, Co):
    """ConvTranspose2d(k=[1,4], stride=[1,2], padding=[0,1]) with bias (rangenet_proto.py:328-334) as three column taps from the
    input onto the [B, H, W, 2 Cout] view of the output (coarse3d_amd/rangenet.py::_up): forward, input gradient, weight gradient."""
    from coarse3d_amd import ops
    from coarse3d_amd.rangenet import UP_TAPS
    g = torch.Generator().manual_seed(2 + Ci + Co + H)
    x = torch.randn(B, Ci, H, W, generator=g).requires_grad_(True)
    wt = (torch.randn(Ci, Co, 1, 4, generator=g) / (2 * Ci ** 0.5)).requires_grad_(True)        # ConvTranspose2d layout [Cin, Cout, 1, 4]
    bias = torch.randn(Co, generator=g) * 0.1
    ref = F.conv_transpose2d(x, wt, bias, stride=(1, 2), padding=(0, 1))
    dout = torch.randn(ref.shape, generator=g)
    ref.backward(dout)
    bb = _pair_helper()
    prev = ops.matrix_precision_state()
    ops.set_matrix_precision(mode, storage=storage) if storage else ops.set_matrix_precision(mode)
    try:
        w3 = bb._pair_weight("l", "up", wt.detach().to(DEV))
        src = ops.Source(ops.to_nhwc(x.detach()).to(DEV))
        got, part = ops.conv_forward([src], ops.pack_weights(w3, 0), bias.repeat(2).to(DEV), 2 * Co, UP_TAPS, stats=True)
        out = got.view(B, H, 2 * W, Co)
        assert rel(ops.from_nhwc(out.cpu()), ref.detach()) < tol
        # the statistics epilogue, folded from the 2 Cout view channels to the Cout real ones
        n = part.shape[2]
        sums = part.view(2, Co, 2, n).permute(1, 2, 0, 3).reshape(Co, 2, 2 * n).double().sum(dim=2).cpu()
        o64 = out.double().cpu()
        assert rel(sums[:, 0], o64.sum(dim=(0, 1, 2))) < 1e-4 and rel(sums[:, 1], (o64 * o64).sum(dim=(0, 1, 2))) < 1e-4
        dz2 = ops.to_nhwc(dout).to(DEV).view(B, H, W, 2 * Co)
        dx, _ = ops.conv_forward([ops.Source(dz2)], ops.pack_weights(w3, 1, c_off=0, c_cnt=Ci, kpad=(2 * Co + 15) // 16 * 16), None, Ci,
                                 ops.negate_taps(UP_TAPS), grad=True)
        assert rel(ops.from_nhwc(dx.cpu()), x.grad) < tol
        dw3 = torch.zeros_like(w3)
        ops.conv_wgrad(src, dz2, dw3, UP_TAPS)
        dwt = torch.empty_like(wt.detach(), device=DEV)
        bb._pair_weight_grad("l", "up", wt.detach().to(DEV), dw3, dwt)
        assert rel(dwt.cpu(), wt.grad) < tol
    finally:
        ops.set_matrix_precision(*prev)


def test_column_resampling_kernel_is_its_own_adjoint():
    """c3d_cols_resample (rounds 2-4 built the strided / transposed convs on it; still exported): <down(a), b> == <a, up(b)>."""
    from coarse3d_amd import ops
    g = torch.Generator().manual_seed(1)
    B, H, W, Co = 2, 8, 64, 64
    a = torch.randn(B, H, W, Co, generator=g).to(DEV)
    b = torch.randn(B, H, W // 2, Co, generator=g).to(DEV)
    lhs = float((ops.cols_resample(a, False) * b).sum())
    rhs = float((a * ops.cols_resample(b, True)).sum())
    assert abs(lhs - rhs) < 1e-3 * abs(lhs)
    assert torch.equal(ops.cols_resample(a, False), a[:, :, ::2].contiguous())


def test_lrelu_residual_add_and_input_repack():
    from coarse3d_amd import ops
    g = torch.Generator().manual_seed(3)
    a, r = torch.randn(2, 8, 32, 64, generator=g), torch.randn(2, 8, 32, 64, generator=g)
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    got = ops.affine_add(r.to(DEV), a.to(DEV), sc.to(DEV), sh.to(DEV), slope=0.1).cpu()
    assert rel(got, r + F.leaky_relu(a * sc + sh, 0.1)) < 1e-6
    got0 = ops.affine_add(r.to(DEV), a.to(DEV), sc.to(DEV), sh.to(DEV)).cpu()
    assert rel(got0, r + a * sc + sh) < 1e-6
    x = torch.randn(2, 5, 8, 32, generator=g)
    p = ops.nchw_to_nhwc_pad(x.to(DEV), 16).cpu()
    assert torch.equal(p[..., :5], x.permute(0, 2, 3, 1)) and float(p[..., 5:].abs().max()) == 0.0
