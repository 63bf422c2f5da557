"""Host side of RangeNet's stride-(1,2) / transposed convs (coarse3d_amd/rangenet.py, round 5): the weight re-indexing that turns
them into stride-1 convs over column-pair views, against torch.nn.functional on the CPU in float64 (no GPU, no library)."""
import torch
import torch.nn.functional as F

from coarse3d_amd.rangenet import DOWN_TAPS, UP_TAPS, RangeNetBackbone


class _Packs:
    pass


def _backbone():
    bb = RangeNetBackbone.__new__(RangeNetBackbone)
    bb.packs = _Packs()
    return bb


def _conv_taps(x, w, taps, bias=None):
    """The engine's convolution: out[p] = sum_t W[:, :, t] x[p + tap_t] (+ bias), zero outside the image.  x NHWC, w OIHW."""
    b, h, wd, c = x.shape
    co = w.shape[0]
    wt = w.reshape(co, c, -1)
    xp = F.pad(x, (0, 0, 1, 1, 1, 1))
    out = torch.zeros(b, h, wd, co, dtype=x.dtype)
    for t, (dy, dx) in enumerate(taps):
        out += torch.einsum("bhwc,oc->bhwo", xp[:, 1 + dy:1 + dy + h, 1 + dx:1 + dx + wd, :], wt[:, :, t])
    return out if bias is None else out + bias


def test_strided_conv_is_a_six_tap_conv_over_the_column_pair_view():
    """Conv2d(3x3, stride (1, 2), padding 1) (rangenet_proto.py:192-200) == six taps over [B, H, W/2, 2C]; the parameter
    gradient is the adjoint of the re-indexing."""
    g = torch.Generator().manual_seed(0)
    bb = _backbone()
    b, h, wd, c, co = 2, 5, 16, 8, 12
    x = torch.randn(b, c, h, wd, generator=g, dtype=torch.float64)
    w = torch.randn(co, c, 3, 3, generator=g, dtype=torch.float64)
    ref = F.conv2d(x, w, stride=(1, 2), padding=1).permute(0, 2, 3, 1)
    w2 = bb._pair_weight("l", "down", w)
    assert tuple(w2.shape) == (co, 2 * c, 3, 2) and len(DOWN_TAPS) == 6
    assert int((w2 == 0).sum()) == co * c * 3           # a quarter of the view weight: the even half under the tap at -1
    got = _conv_taps(x.permute(0, 2, 3, 1).contiguous().view(b, h, wd // 2, 2 * c), w2, DOWN_TAPS)
    assert float((got - ref).abs().max()) < 1e-12
    dw2 = torch.randn(w2.shape, generator=g, dtype=torch.float64)
    dw = torch.empty_like(w)
    bb._pair_weight_grad("l", "down", w, dw2, dw)
    assert abs(float((w2 * dw2).sum() - (w * dw).sum())) < 1e-10


def test_transposed_conv_is_a_three_tap_conv_onto_the_column_pair_view():
    """ConvTranspose2d([1, 4], stride [1, 2], padding [0, 1]) + bias (rangenet_proto.py:328-334) == three column taps from the
    input onto the pair view [B, H, W, 2 Cout] of the output."""
    g = torch.Generator().manual_seed(1)
    bb = _backbone()
    b, h, wd, ci, co = 2, 4, 10, 8, 6
    x = torch.randn(b, ci, h, wd, generator=g, dtype=torch.float64)
    wt = torch.randn(ci, co, 1, 4, generator=g, dtype=torch.float64)
    bias = torch.randn(co, generator=g, dtype=torch.float64)
    ref = F.conv_transpose2d(x, wt, bias, stride=(1, 2), padding=(0, 1)).permute(0, 2, 3, 1)
    w3 = bb._pair_weight("l", "up", wt)
    assert tuple(w3.shape) == (2 * co, ci, 1, 3) and len(UP_TAPS) == 3
    got = _conv_taps(x.permute(0, 2, 3, 1).contiguous(), w3, UP_TAPS, bias.repeat(2)).view(b, h, 2 * wd, co)
    assert float((got - ref).abs().max()) < 1e-12
    dw3 = torch.randn(w3.shape, generator=g, dtype=torch.float64)
    dwt = torch.empty_like(wt)
    bb._pair_weight_grad("l", "up", wt, dw3, dwt)
    assert abs(float((w3 * dw3).sum() - (wt * dwt).sum())) < 1e-10
