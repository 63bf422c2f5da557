"""RangeNetProto with the reference module API (pc_processor/models/rangenet_proto.py:375-676)
on the HIP engine (coarse3d_amd/rangenet.py).

Kept from the reference: constructor keywords (:376-391), ``forward`` signature and returned dict
keys, every ``state_dict`` key and tensor shape (``backbone.conv1.weight`` ...
``decoder.dec5.upconv.weight`` [Cin, Cout, 1, 4] ... ``head.1.weight``), ``prototypes`` re-bound
on update.  The child modules only hold parameters; the prototype pipeline and the data-parallel
hooks are inherited from SalsaNextProto (identical code in the reference, :437-571 vs
salsanext_proto.py:337-402).  The default initialisation uses PyTorch's layer defaults but is
not pinned to the reference's random stream (checkpoints are the expected way in)."""
import torch
import torch.nn as nn

from ... import ops as ops_mod
from ...rangenet import MODEL_BLOCKS, RangeNetBackbone
from .projector import ProjectionV1
from .salsanext_proto import SalsaNextProto

_ENC = [(32, 64), (64, 128), (128, 256), (256, 512), (512, 1024)]
_DEC = [(5, 1024, 512), (4, 512, 256), (3, 256, 128), (2, 128, 64), (1, 64, 32)]
# Dropout2d call sites in forward order: (name, channels, p for layers == 21, p for layers == 53)
_DROP_SITES = (("enc1", 64, 0.01, 0.05), ("enc2", 128, 0.01, 0.05), ("enc3", 256, 0.01, 0.05), ("enc4", 512, 0.01, 0.05),
               ("enc5", 1024, 0.01, 0.05), ("decoder", 32, 0.001, 0.005), ("head", 32, 0.01, 0.05))


class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError("parameter container of the HIP backbone; call RangeNetProto instead")


def _basic_block(inplanes, planes, bn_d):
    blk = _Holder()
    blk.conv1 = nn.Conv2d(inplanes, planes[0], kernel_size=1, bias=False)
    blk.bn1 = nn.BatchNorm2d(planes[0], momentum=bn_d)
    blk.conv2 = nn.Conv2d(planes[0], planes[1], kernel_size=3, padding=1, bias=False)
    blk.bn2 = nn.BatchNorm2d(planes[1], momentum=bn_d)
    return blk


class RangeNetProto(SalsaNextProto):
    _late_dfeat_ok = False           # (this backbone's backward has no late embedding branch: coarse3d_amd/backbone.py)
    def __init__(self, layers=21, nclasses=20, dataset="", path=None, path_append="", proj_dim=256, projection="v1",
                 proj_feat="mix", l2_norm=False, proto_mom=0.999, ignore_label=0, sub_proto_size=20,
                 use_prototype=False):
        nn.Module.__init__(self)
        if layers not in MODEL_BLOCKS:
            raise AssertionError(f"layers must be one of {sorted(MODEL_BLOCKS)}")
        if projection != "v1" or proj_feat != "mix":
            raise NotImplementedError("only projection='v1', proj_feat='mix' exist in the reference")
        self.layers, self.nclasses, self.dataset = layers, nclasses, dataset
        self.path, self.path_append, self.strict = path, path_append, False
        self.l2_norm, self.use_prototype, self.sub_proto_size = l2_norm, use_prototype, sub_proto_size
        self.ignore_label, self.proto_mom, self.projection, self.proj_feat = ignore_label, proto_mom, projection, proj_feat
        self.proj_dim = proj_dim
        bn_d = 0.01
        bb = _Holder()
        bb.conv1 = nn.Conv2d(5, 32, kernel_size=3, padding=1, bias=False)
        bb.bn1 = nn.BatchNorm2d(32, momentum=bn_d)
        for i, (ci, co) in enumerate(_ENC, 1):
            enc = _Holder()
            enc.conv = nn.Conv2d(ci, co, kernel_size=3, stride=(1, 2), padding=1, bias=False)
            enc.bn = nn.BatchNorm2d(co, momentum=bn_d)
            for b in range(MODEL_BLOCKS[layers][i - 1]):
                enc.add_module(f"residual_{b}", _basic_block(co, [ci, co], bn_d))
            bb.add_module(f"enc{i}", enc)
        self.backbone = bb
        dec = _Holder()
        for i, ci, co in _DEC:
            d = _Holder()
            d.upconv = nn.ConvTranspose2d(ci, co, kernel_size=(1, 4), stride=(1, 2), padding=(0, 1))
            d.bn = nn.BatchNorm2d(co, momentum=bn_d)
            d.residual = _basic_block(co, [ci, co], bn_d)
            dec.add_module(f"dec{i}", d)
        self.decoder = dec
        head = _Holder()
        head.add_module("1", nn.Conv2d(32, nclasses, kernel_size=3, padding=1))
        self.head = head
        self.projector = ProjectionV1(480, proj_dim)
        self.prototypes = nn.Parameter(torch.randn(nclasses, sub_proto_size, proj_dim), requires_grad=False)
        nn.init.trunc_normal_(self.prototypes, std=0.02)
        self.feat_norm = nn.LayerNorm(proj_dim)
        self.mask_norm = nn.LayerNorm(nclasses)
        # hooks (as SalsaNextProto)
        self.dropout_masks = None
        self.gumbel_noise = None
        self._bn_reduce = None
        self._world = 1
        self._proto_mean = None
        self._proto_sums_reduce = None
        self._grad_ready = None
        self._block_done = None
        self._flat_grads = None
        self._side = None
        self._packs = ops_mod.PackCache()

    def _make_backbone(self, P):
        reduce_fn, world = self._bn_exchange()
        return RangeNetBackbone(P, self.nclasses, self.dataset, reduce_fn, world, self._packs, self.layers,
                                self._side_stream_for(P))

    def _check_input(self, h, w):
        wp = w + 24 if self.dataset == "SemanticPOSS" else w
        assert wp % 32 == 0, "input width must be a multiple of 32 (after the SemanticPOSS padding of 24)"

    def _draw_masks(self, b, device):
        if self.dropout_masks is not None:
            return self.dropout_masks
        masks = {}
        for name, c, p21, p53 in _DROP_SITES:
            p = p21 if self.layers == 21 else p53
            masks[name] = (torch.rand(b, c, device=device) >= p).to(torch.float32) * (1.0 / (1.0 - p))
        return masks

    def forward(self, x, label=None, eval_mask=None, return_feat=False, proto_loss=False, proto_pl=None, unproj_data=None):
        return super().forward(x, label=label, eval_mask=eval_mask, return_feat=return_feat, proto_loss=proto_loss,
                               proto_pl=proto_pl)
