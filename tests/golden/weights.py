"""Closed-form (hash-seeded) weights shared by the golden-vector generator and the tests.

A 30 MB state_dict is too large to commit, so every tensor is regenerated bit-identically
from its *name* with NumPy's PCG64 stream (stable across NumPy versions)."""
import math
import sys
import zlib
from collections import OrderedDict

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
from oracle import coarse3d_oracle as oc  # noqa: E402


def _gen(name, salt=0):
    return np.random.Generator(np.random.PCG64(zlib.crc32(name.encode()) + 7919 * salt))


def closed_form_state(in_channel=5, nclasses=20, sub_proto=20, proj_dim=256, salt=0, base=32):
    st = OrderedDict()
    for name, (co, ci, kh, kw) in oc.conv_specs(in_channel, nclasses, base, proj_dim).items():
        bound = 1.0 / math.sqrt(ci * kh * kw)
        g = _gen(name, salt)
        st[f"{name}.weight"] = torch.from_numpy(
            g.uniform(-bound, bound, (co, ci, kh, kw)).astype(np.float32) * 1.7)
        st[f"{name}.bias"] = torch.from_numpy(g.uniform(-bound, bound, co).astype(np.float32))
    for name, c in oc.bn_specs(base).items():
        g = _gen(name, salt)
        st[f"{name}.weight"] = torch.from_numpy(g.uniform(0.5, 1.5, c).astype(np.float32))
        st[f"{name}.bias"] = torch.from_numpy(g.uniform(-0.2, 0.2, c).astype(np.float32))
        st[f"{name}.running_mean"] = torch.from_numpy(g.uniform(-0.1, 0.1, c).astype(np.float32))
        st[f"{name}.running_var"] = torch.from_numpy(g.uniform(0.5, 1.5, c).astype(np.float32))
        st[f"{name}.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    g = _gen("prototypes", salt)
    st["prototypes"] = torch.from_numpy(
        g.normal(0, 0.02, (nclasses, sub_proto, proj_dim)).astype(np.float32))
    for name, c in (("feat_norm", proj_dim), ("mask_norm", nclasses)):
        g = _gen(name, salt)
        st[f"{name}.weight"] = torch.from_numpy(g.uniform(0.5, 1.5, c).astype(np.float32))
        st[f"{name}.bias"] = torch.from_numpy(g.uniform(-0.2, 0.2, c).astype(np.float32))
    return st


def fc_state(features=256, classes=1000, salt=0):
    """Closed-form parameters of the ImageNet pre-training head FC (salsanext_proto.py:216-231): fc.linear.{weight, bias}."""
    g = _gen("fc.linear", salt)
    bound = 1.0 / math.sqrt(features)
    return OrderedDict([("fc.linear.weight", torch.from_numpy(g.uniform(-bound, bound, (classes, features)).astype(np.float32))),
                        ("fc.linear.bias", torch.from_numpy(g.uniform(-bound, bound, classes).astype(np.float32)))])


def block_state(kind, cin, cout, name="blk", salt=0):
    """Closed-form parameters of ONE block (kind in ctx/res/up) under prefix ``name``."""
    convs, bns = OrderedDict(), OrderedDict()
    if kind == "ctx":
        convs = {"conv1": (cout, cin, 1, 1), "conv2": (cout, cout, 3, 3), "conv3": (cout, cout, 3, 3)}
        bns = {"bn1": cout, "bn2": cout}
    elif kind == "res":
        convs = {"conv1": (cout, cin, 1, 1), "conv2": (cout, cin, 3, 3), "conv3": (cout, cout, 3, 3),
                 "conv4": (cout, cout, 2, 2), "conv5": (cout, 3 * cout, 1, 1)}
        bns = {f"bn{i}": cout for i in range(1, 5)}
    elif kind == "up":
        convs = {"conv1": (cout, cin // 4 + 2 * cout, 3, 3), "conv2": (cout, cout, 3, 3),
                 "conv3": (cout, cout, 2, 2), "conv4": (cout, 3 * cout, 1, 1)}
        bns = {f"bn{i}": cout for i in range(1, 5)}
    st = OrderedDict()
    for k, (co, ci, kh, kw) in convs.items():
        g = _gen(f"{kind}.{k}", salt)
        bound = 1.0 / math.sqrt(ci * kh * kw)
        st[f"{name}.{k}.weight"] = torch.from_numpy(
            g.uniform(-bound, bound, (co, ci, kh, kw)).astype(np.float32) * 1.7)
        st[f"{name}.{k}.bias"] = torch.from_numpy(g.uniform(-bound, bound, co).astype(np.float32))
    for k, c in bns.items():
        g = _gen(f"{kind}.{k}", salt)
        st[f"{name}.{k}.weight"] = torch.from_numpy(g.uniform(0.5, 1.5, c).astype(np.float32))
        st[f"{name}.{k}.bias"] = torch.from_numpy(g.uniform(-0.2, 0.2, c).astype(np.float32))
        st[f"{name}.{k}.running_mean"] = torch.zeros(c)
        st[f"{name}.{k}.running_var"] = torch.ones(c)
        st[f"{name}.{k}.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    return st


def synthetic_batch(b, h, w, ncls, seed, label_rate=1e-3, gh=8, gw=64):
    """BASELINE.md section 3 synthetic inputs: x ~ N(0,1); blocky eval labels; sparse weak labels."""
    g = np.random.Generator(np.random.PCG64(seed))
    x = g.standard_normal((b, 5, h, w)).astype(np.float32)
    gy, gx = max(h // gh, 1), max(w // gw, 1)
    grid = g.integers(0, ncls, (b, gy, gx))
    ev = grid[:, (np.arange(h) * gy // h)[:, None], (np.arange(w) * gx // w)[None, :]]
    keep = g.random((b, h, w)) < label_rate
    tr = ev * keep
    return (torch.from_numpy(x), torch.from_numpy(tr.astype(np.int64)),
            torch.from_numpy(ev.astype(np.int64)))


def dropout_masks_for(state_or_channels, b, seed):
    """[B,C] multipliers (0 or 1.25) for the 13 Dropout2d sites."""
    chans = {"resBlock2.dropout": 128, "resBlock3.dropout": 256, "resBlock4.dropout": 256,
             "resBlock5.dropout": 256,
             "upBlock1.dropout1": 64, "upBlock1.dropout2": 320, "upBlock1.dropout3": 128,
             "upBlock2.dropout1": 32, "upBlock2.dropout2": 288, "upBlock2.dropout3": 128,
             "upBlock3.dropout1": 32, "upBlock3.dropout2": 160, "upBlock3.dropout3": 64}
    g = np.random.Generator(np.random.PCG64(seed))
    return {k: torch.from_numpy((g.random((b, c)) >= 0.2).astype(np.float32) * 1.25)
            for k, c in chans.items()}


def rangenet_state(layers=21, nclasses=20, sub_proto=20, proj_dim=256, salt=0):
    """Closed-form RangeNetProto parameters (names / shapes from oracle/rangenet_oracle.py): 25 M
    values regenerated bit-identically wherever the tests run, so the 100 MB state never ships."""
    from oracle import rangenet_oracle as ro
    st = OrderedDict()
    st["prototypes"] = torch.from_numpy(_gen("rn.prototypes", salt).normal(0, 0.02, (nclasses, sub_proto, proj_dim)).astype(np.float32))
    for name, (shape, has_bias) in ro.conv_specs(layers, nclasses, proj_dim).items():
        fan_in = shape[1] * shape[2] * shape[3] if "upconv" not in name else shape[0] * shape[3] / 2.0
        bound = 1.0 / math.sqrt(fan_in)
        g = _gen("rn." + name, salt)
        st[f"{name}.weight"] = torch.from_numpy(g.uniform(-bound, bound, shape).astype(np.float32) * 1.7)
        if has_bias:
            st[f"{name}.bias"] = torch.from_numpy(g.uniform(-bound, bound, shape[1] if "upconv" in name else shape[0]).astype(np.float32))
    for name, c in ro.bn_specs(layers).items():
        g = _gen("rn." + name, salt)
        st[f"{name}.weight"] = torch.from_numpy(g.uniform(0.5, 1.5, c).astype(np.float32))
        st[f"{name}.bias"] = torch.from_numpy(g.uniform(-0.2, 0.2, c).astype(np.float32))
        st[f"{name}.running_mean"] = torch.from_numpy(g.uniform(-0.1, 0.1, c).astype(np.float32))
        st[f"{name}.running_var"] = torch.from_numpy(g.uniform(0.5, 1.5, c).astype(np.float32))
        st[f"{name}.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    for name, c in (("feat_norm", proj_dim), ("mask_norm", nclasses)):
        g = _gen("rn." + name, salt)
        st[f"{name}.weight"] = torch.from_numpy(g.uniform(0.5, 1.5, c).astype(np.float32))
        st[f"{name}.bias"] = torch.from_numpy(g.uniform(-0.2, 0.2, c).astype(np.float32))
    return st


def rangenet_masks(b, seed, layers=21):
    """Injected Dropout2d multipliers for the seven call sites (keep-probabilities as in the
    reference: 0.99 backbone, 0.999 decoder, 0.99 head; drawn with a larger drop rate so that the
    fixtures actually contain dropped planes)."""
    g = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for site, c, p in (("enc1", 64, 0.2), ("enc2", 128, 0.2), ("enc3", 256, 0.2), ("enc4", 512, 0.2), ("enc5", 1024, 0.2),
                       ("decoder", 32, 0.2), ("head", 32, 0.2)):
        out[site] = torch.from_numpy(((g.random((b, c)) >= p) / (1 - p)).astype(np.float32))
    return out


def rangenet_inputs(b, h, w, ncls, w_feat=None):
    """x, d(pred), d(feat) of the RangeNet fixtures (seeded CPU generator, fixed draw order)."""
    g = torch.Generator().manual_seed(1)
    x = torch.randn(b, 5, h, w, generator=g)
    dp = torch.randn(b, ncls, h, w, generator=g)
    df = torch.randn(b, 256, h, w if w_feat is None else w_feat, generator=g) * 0.05
    return x, dp, df


def squeezeseg_state(layers=21, nclasses=20, sub_proto=20, proj_dim=256, salt=0):
    """Closed-form SqueezeSegV3Proto parameters (names / shapes from oracle/squeezeseg_oracle.py)."""
    from oracle import squeezeseg_oracle as so
    st = OrderedDict()
    st["prototypes"] = torch.from_numpy(_gen("ss.prototypes", salt).normal(0, 0.02, (nclasses, sub_proto, proj_dim)).astype(np.float32))
    for name, (shape, has_bias) in so.conv_specs(layers, nclasses, proj_dim).items():
        fan_in = shape[1] * shape[2] * shape[3] if "upconv" not in name else shape[0] * shape[3] / 2.0
        bound = 1.0 / math.sqrt(fan_in)
        g = _gen("ss." + name, salt)
        st[f"{name}.weight"] = torch.from_numpy(g.uniform(-bound, bound, shape).astype(np.float32) * 1.7)
        if has_bias:
            st[f"{name}.bias"] = torch.from_numpy(g.uniform(-bound, bound, shape[1] if "upconv" in name else shape[0]).astype(np.float32))
    for name, c in so.bn_specs(layers).items():
        g = _gen("ss." + name, salt)
        st[f"{name}.weight"] = torch.from_numpy(g.uniform(0.5, 1.5, c).astype(np.float32))
        st[f"{name}.bias"] = torch.from_numpy(g.uniform(-0.2, 0.2, c).astype(np.float32))
        st[f"{name}.running_mean"] = torch.from_numpy(g.uniform(-0.1, 0.1, c).astype(np.float32))
        st[f"{name}.running_var"] = torch.from_numpy(g.uniform(0.5, 1.5, c).astype(np.float32))
        st[f"{name}.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    for name, c in (("feat_norm", proj_dim), ("mask_norm", nclasses)):
        g = _gen("ss." + name, salt)
        st[f"{name}.weight"] = torch.from_numpy(g.uniform(0.5, 1.5, c).astype(np.float32))
        st[f"{name}.bias"] = torch.from_numpy(g.uniform(-0.2, 0.2, c).astype(np.float32))
    return st


def squeezeseg_masks(b, seed):
    """Injected Dropout2d multipliers for the seven call sites of SqueezeSegV3Proto (drawn with a
    larger drop rate than the reference's 0.01 so that the fixtures contain dropped planes)."""
    from oracle import squeezeseg_oracle as so
    g = np.random.Generator(np.random.PCG64(seed))
    return {site: torch.from_numpy(((g.random((b, c)) >= 0.2) / 0.8).astype(np.float32))
            for site, c in zip(so.DROP_SITES, so.DROP_CHANNELS)}
