"""CPU oracle of the RangeNet (Darknet-21/53) prototype backbone -- TEST INFRASTRUCTURE ONLY.

Restates the arithmetic of the reference ``RangeNetProto`` (pc_processor/models/rangenet_proto.py:
BasicBlock :38-63, Backbone :76-259, Decoder :261-372, RangeNetProto.forward :573-676) as plain
functions over a ``state`` dict with the reference's state_dict names, with injected Dropout2d
masks.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path never does.

Parity: PINNED -- tests/test_oracle_golden.py checks it against vectors captured from the real
reference (tests/golden/make_golden.py::gold_rangenet)."""
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.01          # Backbone.bn_d / Decoder.bn_d (rangenet_proto.py:88, :271)
PROJ_BN_MOMENTUM = 0.1      # ProjectionV1's BatchNorm2d default
SLOPE = 0.1                 # nn.LeakyReLU(0.1)
MODEL_BLOCKS = {21: [1, 1, 2, 2, 1], 53: [1, 2, 8, 8, 4]}
ENC_PLANES = [(32, 64), (64, 128), (128, 256), (256, 512), (512, 1024)]
DEC_PLANES = [(1024, 512), (512, 256), (256, 128), (128, 64), (64, 32)]      # dec5 .. dec1
DROP_SITES = ("enc1", "enc2", "enc3", "enc4", "enc5", "decoder", "head")     # Dropout2d call order


def conv_specs(layers=21, nclasses=20, proj_dim=256):
    """name -> (shape, has_bias); ConvTranspose2d weights are [Cin, Cout, 1, 4]."""
    s = OrderedDict()
    s["backbone.conv1"] = ((32, 5, 3, 3), False)
    for i, (ci, co) in enumerate(ENC_PLANES, 1):
        s[f"backbone.enc{i}.conv"] = ((co, ci, 3, 3), False)
        for b in range(MODEL_BLOCKS[layers][i - 1]):
            s[f"backbone.enc{i}.residual_{b}.conv1"] = ((ci, co, 1, 1), False)
            s[f"backbone.enc{i}.residual_{b}.conv2"] = ((co, ci, 3, 3), False)
    for i, (ci, co) in zip((5, 4, 3, 2, 1), DEC_PLANES):
        s[f"decoder.dec{i}.upconv"] = ((ci, co, 1, 4), True)
        s[f"decoder.dec{i}.residual.conv1"] = ((ci, co, 1, 1), False)
        s[f"decoder.dec{i}.residual.conv2"] = ((co, ci, 3, 3), False)
    s["head.1"] = ((nclasses, 32, 3, 3), True)
    s["projector.proj.0"] = ((480, 480, 1, 1), True)
    s["projector.proj.3"] = ((proj_dim, 480, 1, 1), True)
    return s


def bn_specs(layers=21):
    s = OrderedDict()
    s["backbone.bn1"] = 32
    for i, (ci, co) in enumerate(ENC_PLANES, 1):
        s[f"backbone.enc{i}.bn"] = co
        for b in range(MODEL_BLOCKS[layers][i - 1]):
            s[f"backbone.enc{i}.residual_{b}.bn1"] = ci
            s[f"backbone.enc{i}.residual_{b}.bn2"] = co
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
        self.bn_stats = OrderedDict()

    def bn_act(self, name, x, momentum=BN_MOMENTUM, act=True):
        w, b = self.p[f"{name}.weight"], self.p[f"{name}.bias"]
        rm, rv = self.p[f"{name}.running_mean"], self.p[f"{name}.running_var"]
        if self.train:
            n = x.numel() // x.shape[1]
            mean = x.mean(dim=(0, 2, 3))
            var = x.var(dim=(0, 2, 3), unbiased=False)
            self.bn_stats[name] = (mean.detach().clone(), var.detach().clone())
            if self.update_running:
                with torch.no_grad():
                    rm.mul_(1 - momentum).add_(momentum * mean.detach())
                    rv.mul_(1 - momentum).add_(momentum * var.detach() * n / max(n - 1, 1))
                    self.p[f"{name}.num_batches_tracked"] += 1
            y = (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS)
            y = y * w[None, :, None, None] + b[None, :, None, None]
        else:
            sc = w / torch.sqrt(rv + BN_EPS)
            y = x * sc[None, :, None, None] + (b - rm * sc)[None, :, None, None]
        return F.leaky_relu(y, SLOPE) if act else y

    def drop(self, site, x):
        if not self.train or self.masks is None or site not in self.masks:
            return x
        return x * self.masks[site][:, :, None, None]


def basic_block(c, name, x):
    """rangenet_proto.py:52-63: 1x1 -> BN -> LReLU -> 3x3 -> BN -> LReLU, plus the input."""
    p = c.p
    out = c.bn_act(f"{name}.bn1", F.conv2d(x, p[f"{name}.conv1.weight"]))
    out = c.bn_act(f"{name}.bn2", F.conv2d(out, p[f"{name}.conv2.weight"], padding=1))
    return out + x


def rangenet_forward(state, x, train=True, dropout_masks=None, return_feat=True, layers=21,
                     dataset="SemanticKitti", update_running=True):
    """x [B,5,H,W] -> dict(pred_2d, feat_2d, logits, bn_stats).  ``dropout_masks``: site ->
    [B, C] multiplier for the seven Dropout2d calls (DROP_SITES), in forward order."""
    c = _Ctx(state, train, dropout_masks, update_running)
    p = state
    w_in = x.shape[3]
    if dataset == "SemanticPOSS":                       # rangenet_proto.py:586-590: 24 zero columns
        x = F.pad(x, (0, 24))
    skips = {}
    os_ = 1
    t = c.bn_act("backbone.bn1", F.conv2d(x, p["backbone.conv1.weight"], padding=1))
    for i in range(1, 6):
        name = f"backbone.enc{i}"
        y = c.bn_act(f"{name}.bn", F.conv2d(t, p[f"{name}.conv.weight"], stride=(1, 2), padding=1))
        for b in range(MODEL_BLOCKS[layers][i - 1]):
            y = basic_block(c, f"{name}.residual_{b}", y)
        skips[os_] = t.detach()                         # run_layer: the input of the layer that shrank it
        os_ *= 2
        t = c.drop(f"enc{i}", y)
    for i in (5, 4, 3, 2, 1):
        name = f"decoder.dec{i}"
        y = F.conv_transpose2d(t, p[f"{name}.upconv.weight"], p[f"{name}.upconv.bias"], stride=(1, 2), padding=(0, 1))
        y = c.bn_act(f"{name}.bn", y)
        y = basic_block(c, f"{name}.residual", y)
        os_ //= 2
        t = y + skips[os_]
    t = c.drop("decoder", t)
    t = c.drop("head", t)
    logits = F.conv2d(t, p["head.1.weight"], p["head.1.bias"], padding=1)
    prob = F.softmax(logits, dim=1)
    if dataset == "SemanticPOSS":
        prob = prob[:, :, :, :w_in]
    out = {"pred_2d": prob.contiguous(), "logits": logits}
    if return_feat:
        h, w = prob.shape[2] // 2, prob.shape[3] // 2
        feat = torch.cat([F.interpolate(skips[k], size=(h, w), mode="bilinear", align_corners=True) for k in (1, 2, 4, 8)], 1)
        z = F.conv2d(feat, p["projector.proj.0.weight"], p["projector.proj.0.bias"])
        z = F.leaky_relu(c.bn_act("projector.proj.1", z, PROJ_BN_MOMENTUM, act=False), 0.01)   # nn.LeakyReLU() default slope
        emb = F.conv2d(z, p["projector.proj.3.weight"], p["projector.proj.3.bias"])
        emb = F.normalize(emb, p=2, dim=1)
        out["feat_2d"] = F.interpolate(emb, size=(x.shape[2], x.shape[3]), mode="bilinear", align_corners=True)
    out["bn_stats"] = c.bn_stats
    return out
