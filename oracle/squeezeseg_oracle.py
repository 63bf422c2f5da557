"""CPU oracle of the SqueezeSegV3 prototype backbone -- TEST INFRASTRUCTURE ONLY.

Restates the arithmetic of the reference ``SqueezeSegV3Proto`` (pc_processor/models/
squeezesegv3_Proto.py: SACBlock :468-503, Backbone :515-682, BasicBlock :685-715, Decoder
:721-829, forward :353-465) as plain functions over a ``state`` dict with the reference's
state_dict names, with injected Dropout2d masks.  Only tests/ may import this module; the product
path never does.

Parity: PINNED -- tests/test_oracle_golden.py checks it against vectors captured from the real
reference module (tests/golden/make_golden_round2.py::gold_squeezeseg)."""
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.01          # Backbone.bn_d / Decoder.bn_d (:526, :730)
SAC_BN_MOMENTUM = 0.1       # SACBlock's BatchNorm2d layers (:476, :481, :485)
PROJ_BN_MOMENTUM = 0.1
SLOPE = 0.1                 # nn.LeakyReLU(0.1)
MODEL_BLOCKS = {21: [1, 1, 2, 2, 1], 53: [1, 2, 8, 8, 4]}
ENC_PLANES = [(32, 64), (64, 128), (128, 256), (256, 256), (256, 256)]
ENC_DS = [True, True, True, False, False]          # strides [2, 2, 2, 1, 1] at OS = 8 (:543-568)
DEC_PLANES = [(256, 256), (256, 256), (256, 128), (128, 64), (64, 32)]      # dec5 .. dec1
DEC_UP = [False, False, True, True, True]          # decoder strides [1, 1, 2, 2, 2] (:733-750)
DROP_SITES = ("enc1", "enc2", "enc3", "enc4", "enc5", "decoder", "head")     # Dropout2d call order
DROP_CHANNELS = (64, 128, 256, 256, 256, 32, 32)
HEAD_IN = (256, 256, 128, 64)                      # head1..head4 (1x1, parameters only: forward uses head5)


def conv_specs(layers=21, nclasses=20, proj_dim=256):
    """name -> (weight shape, has_bias); ConvTranspose2d weights are [Cin, Cout, 1, 4]."""
    s = OrderedDict()
    s["backbone.conv1"] = ((32, 5, 3, 3), False)
    for i, ((ci, co), ds) in enumerate(zip(ENC_PLANES, ENC_DS), 1):
        for b in range(MODEL_BLOCKS[layers][i - 1]):
            n = f"backbone.enc{i}.residual_{b}"
            s[f"{n}.attention_x.0"] = ((9 * ci, 3, 7, 7), True)
            s[f"{n}.position_mlp_2.0"] = ((ci, 9 * ci, 1, 1), True)
            s[f"{n}.position_mlp_2.3"] = ((ci, ci, 3, 3), True)
        if ds:
            s[f"backbone.enc{i}.conv"] = ((co, ci, 3, 3), False)
    for i, (ci, co), up in zip((5, 4, 3, 2, 1), DEC_PLANES, DEC_UP):
        if up:
            s[f"decoder.dec{i}.upconv"] = ((ci, co, 1, 4), True)
        else:
            s[f"decoder.dec{i}.conv"] = ((co, ci, 3, 3), True)
        s[f"decoder.dec{i}.residual.conv1"] = ((ci, co, 1, 1), False)
        s[f"decoder.dec{i}.residual.conv2"] = ((co, ci, 3, 3), False)
    for k, ci in enumerate(HEAD_IN, 1):
        s[f"head{k}.1"] = ((nclasses, ci, 1, 1), True)
    s["head5.1"] = ((nclasses, 32, 3, 3), True)
    s["projector.proj.0"] = ((480, 480, 1, 1), True)
    s["projector.proj.3"] = ((proj_dim, 480, 1, 1), True)
    return s


def bn_specs(layers=21):
    s = OrderedDict()
    s["backbone.bn1"] = 32
    for i, ((ci, co), ds) in enumerate(zip(ENC_PLANES, ENC_DS), 1):
        for b in range(MODEL_BLOCKS[layers][i - 1]):
            n = f"backbone.enc{i}.residual_{b}"
            s[f"{n}.attention_x.1"] = 9 * ci
            s[f"{n}.position_mlp_2.1"] = ci
            s[f"{n}.position_mlp_2.4"] = ci
        if ds:
            s[f"backbone.enc{i}.bn"] = co
    for i, (ci, co) in zip((5, 4, 3, 2, 1), DEC_PLANES):
        s[f"decoder.dec{i}.bn"] = co
        s[f"decoder.dec{i}.residual.bn1"] = ci
        s[f"decoder.dec{i}.residual.bn2"] = co
    s["projector.proj.1"] = 480
    return s


def trainable_names(state):
    return [k for k, v in state.items() if v.is_floating_point() and v.dim() > 0 and k != "prototypes"
            and not k.endswith(("running_mean", "running_var")) and not k.startswith(("feat_norm", "mask_norm"))]


class _Ctx:
    def __init__(self, state, train, masks, update_running):
        self.p, self.train, self.masks, self.update_running = state, train, masks, update_running

    def bn(self, name, x, momentum):
        w, b = self.p[f"{name}.weight"], self.p[f"{name}.bias"]
        rm, rv = self.p[f"{name}.running_mean"], self.p[f"{name}.running_var"]
        if self.train:
            n = x.numel() // x.shape[1]
            mean = x.mean(dim=(0, 2, 3))
            var = x.var(dim=(0, 2, 3), unbiased=False)
            if self.update_running:
                with torch.no_grad():
                    rm.mul_(1 - momentum).add_(momentum * mean.detach())
                    rv.mul_(1 - momentum).add_(momentum * var.detach() * n / max(n - 1, 1))
                    self.p[f"{name}.num_batches_tracked"] += 1
            y = (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS)
            return y * w[None, :, None, None] + b[None, :, None, None]
        sc = w / torch.sqrt(rv + BN_EPS)
        return x * sc[None, :, None, None] + (b - rm * sc)[None, :, None, None]

    def drop(self, site, x):
        if not self.train or self.masks is None or site not in self.masks:
            return x
        return x * self.masks[site][:, :, None, None]


def sac_block(c, name, xyz, feature):
    """squeezesegv3_Proto.py:490-503: spatially-adaptive convolution."""
    p = c.p
    n, ch, h, w = feature.shape
    new_feature = F.unfold(feature, kernel_size=3, padding=1).view(n, -1, h, w)          # channel = c * 9 + tap
    att = F.conv2d(xyz, p[f"{name}.attention_x.0.weight"], p[f"{name}.attention_x.0.bias"], padding=3)
    att = torch.sigmoid(c.bn(f"{name}.attention_x.1", att, SAC_BN_MOMENTUM))
    new_feature = new_feature * att
    y = F.conv2d(new_feature, p[f"{name}.position_mlp_2.0.weight"], p[f"{name}.position_mlp_2.0.bias"])
    y = F.relu(c.bn(f"{name}.position_mlp_2.1", y, SAC_BN_MOMENTUM))
    y = F.conv2d(y, p[f"{name}.position_mlp_2.3.weight"], p[f"{name}.position_mlp_2.3.bias"], padding=1)
    y = F.relu(c.bn(f"{name}.position_mlp_2.4", y, SAC_BN_MOMENTUM))
    return y + feature


def basic_block(c, name, x):
    """:702-715: 1x1 -> BN -> LReLU -> 3x3 -> BN -> LReLU, plus the input."""
    p = c.p
    out = F.leaky_relu(c.bn(f"{name}.bn1", F.conv2d(x, p[f"{name}.conv1.weight"]), BN_MOMENTUM), SLOPE)
    out = F.leaky_relu(c.bn(f"{name}.bn2", F.conv2d(out, p[f"{name}.conv2.weight"], padding=1), BN_MOMENTUM), SLOPE)
    return out + x


def squeezeseg_forward(state, x, train=True, dropout_masks=None, return_feat=True, layers=21, update_running=True):
    """x [B,5,H,W] -> dict(pred_2d, feat_2d, logits).  ``dropout_masks``: site -> [B, C] multiplier for
    the seven Dropout2d calls (DROP_SITES), in forward order."""
    c = _Ctx(state, train, dropout_masks, update_running)
    p = state
    skips = {}
    os_ = 1
    xyz = x[:, 1:4]
    feature = F.leaky_relu(c.bn("backbone.bn1", F.conv2d(x, p["backbone.conv1.weight"], padding=1), BN_MOMENTUM), SLOPE)
    for i in range(1, 6):
        name = f"backbone.enc{i}"
        y = feature
        for b in range(MODEL_BLOCKS[layers][i - 1]):
            y = sac_block(c, f"{name}.residual_{b}", xyz, y)
        if ENC_DS[i - 1]:                                    # run_layer flag=True (:645-649)
            y = F.conv2d(y, p[f"{name}.conv.weight"], stride=(1, 2), padding=1)
            y = F.leaky_relu(c.bn(f"{name}.bn", y, BN_MOMENTUM), SLOPE)
            xyz = F.interpolate(xyz, size=(xyz.shape[2], xyz.shape[3] // 2), mode="bilinear", align_corners=True)
            skips[os_] = feature.detach()                    # :652-654: the input of the layer that shrank it
            os_ *= 2
        feature = c.drop(f"enc{i}", y)
    t = feature
    for i, (ci, co), up in zip((5, 4, 3, 2, 1), DEC_PLANES, DEC_UP):
        name = f"decoder.dec{i}"
        if up:
            y = F.conv_transpose2d(t, p[f"{name}.upconv.weight"], p[f"{name}.upconv.bias"], stride=(1, 2), padding=(0, 1))
        else:
            y = F.conv2d(t, p[f"{name}.conv.weight"], p[f"{name}.conv.bias"], padding=1)
        y = F.leaky_relu(c.bn(f"{name}.bn", y, BN_MOMENTUM), SLOPE)
        y = basic_block(c, f"{name}.residual", y)
        if up:
            os_ //= 2
            y = y + skips[os_]
        t = y
    t = c.drop("decoder", t)
    t = c.drop("head", t)
    logits = F.conv2d(t, p["head5.1.weight"], p["head5.1.bias"], padding=1)
    prob = F.softmax(logits, dim=1)
    out = {"pred_2d": prob.contiguous(), "logits": logits}
    if return_feat:
        h, w = prob.shape[2] // 2, prob.shape[3] // 2
        srcs = [skips[1], skips[2], skips[4], feature]       # the last one is NOT detached (:405-417)
        feat = torch.cat([F.interpolate(s, size=(h, w), mode="bilinear", align_corners=True) for s in srcs], 1)
        z = F.conv2d(feat, p["projector.proj.0.weight"], p["projector.proj.0.bias"])
        z = F.leaky_relu(c.bn("projector.proj.1", z, PROJ_BN_MOMENTUM), 0.01)
        emb = F.conv2d(z, p["projector.proj.3.weight"], p["projector.proj.3.bias"])
        emb = F.normalize(emb, p=2, dim=1)
        out["feat_2d"] = F.interpolate(emb, size=(prob.shape[2], prob.shape[3]), mode="bilinear", align_corners=True)
    return out
