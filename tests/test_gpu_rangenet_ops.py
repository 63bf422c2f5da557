"""Engine features the RangeNet backbone needs on top of SalsaNext's, each against
plain PyTorch on the CPU at 1e-4 of max|ref|: LeakyReLU slope 0.1 on load / in the epilogue /
in the BatchNorm backward, 4-tap kernels with offsets up to 2, stride-(1,2) convolution as
stride-1 + column subsampling, ConvTranspose2d([1,4],[1,2],[0,1]) as zero insertion + 4-tap conv,
the LeakyReLU residual add and the 5 -> 16 channel input repack."""
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


def test_strided_conv_as_stride1_plus_subsample():
    from coarse3d_amd import ops
    g = torch.Generator().manual_seed(1)
    B, H, W, Ci, Co = 2, 8, 64, 32, 64
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / 17
    ref = F.conv2d(x, w, stride=(1, 2), padding=1)
    full, _ = ops.conv_forward([ops.Source(ops.to_nhwc(x).to(DEV))], ops.pack_weights(w.to(DEV), 0), None, Co, ops.conv_taps(3, 3, 1, 1))
    got = ops.cols_resample(full, up=False)
    assert got.shape == (B, H, W // 2, Co) and rel(ops.from_nhwc(got.cpu()), ref) < 1e-4
    # adjoint pair: <down(a), b> == <a, up(b)>
    a = torch.randn(B, H, W, Co, generator=g).to(DEV)
    b = torch.randn(B, H, W // 2, Co, generator=g).to(DEV)
    lhs = float((ops.cols_resample(a, False) * b).sum())
    rhs = float((a * ops.cols_resample(b, True)).sum())
    assert abs(lhs - rhs) < 1e-3 * abs(lhs)


def test_transposed_conv_as_zero_insert_plus_4tap():
    """ConvTranspose2d(k=[1,4], stride=[1,2], padding=[0,1]) with bias: forward, input gradient,
    weight gradient."""
    from coarse3d_amd import ops
    g = torch.Generator().manual_seed(2)
    B, H, W, Ci, Co = 2, 8, 32, 64, 32
    x = torch.randn(B, Ci, H, W, generator=g).requires_grad_(True)
    wt = (torch.randn(Ci, Co, 1, 4, generator=g) / 16).requires_grad_(True)        # ConvTranspose2d layout [Cin, Cout, 1, 4]
    bias = torch.randn(Co, generator=g) * 0.1
    ref = F.conv_transpose2d(x, wt, bias, stride=(1, 2), padding=(0, 1))
    dout = torch.randn(ref.shape, generator=g)
    ref.backward(dout)
    # out[x] = sum_k u[x + 1 - k] * wt[:, :, 0, k]  with u = zero-inserted input
    taps = [(0, 1 - k) for k in range(4)]
    w_conv = wt.detach().permute(1, 0, 2, 3).contiguous()                           # OIHW [Cout, Cin, 1, 4]
    u = ops.cols_resample(ops.to_nhwc(x.detach()).to(DEV), up=True)
    got, _ = ops.conv_forward([ops.Source(u)], ops.pack_weights(w_conv.to(DEV), 0), bias.to(DEV), Co, taps)
    assert got.shape == (B, H, 2 * W, Co) and rel(ops.from_nhwc(got.cpu()), ref.detach()) < 1e-4
    dzo = ops.to_nhwc(dout).to(DEV)
    du, _ = ops.conv_forward([ops.Source(dzo)], ops.pack_weights(w_conv.to(DEV), 1), None, Ci, ops.negate_taps(taps))
    dx = ops.cols_resample(du, up=False)
    assert rel(ops.from_nhwc(dx.cpu()), x.grad) < 1e-4
    dw = torch.zeros_like(w_conv, device=DEV)
    ops.conv_wgrad(ops.Source(u), dzo, dw, taps)
    assert rel(dw.cpu().permute(1, 0, 2, 3), wt.grad) < 1e-4


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
