"""SqueezeSegV3Proto with the reference module API (pc_processor/models/squeezesegv3_Proto.py:
39-466) on the HIP engine (coarse3d_amd/squeezeseg.py).

Kept from the reference: constructor keywords (:40-58), ``forward`` signature (:353-362) and
returned dict keys, every ``state_dict`` key and tensor shape (``backbone.enc1.residual_0.
attention_x.0.weight`` [288, 3, 7, 7] ... ``decoder.dec3.upconv.weight`` [Cin, Cout, 1, 4] ...
``head1.1`` .. ``head5.1``), ``prototypes`` re-bound on update.  ``head1`` .. ``head4`` exist in the
reference (:76-95) but its forward only uses ``head5`` (:377-395): they are kept as parameters (a
released checkpoint loads) and never receive a gradient.  The prototype pipeline and the
data-parallel hooks are inherited from SalsaNextProto (identical code in the reference, :253-351).
The loading of pretrained sub-module files (``path`` / ``path_append``, :123-238) is file handling
outside the accelerated path: pass a state_dict to ``load_state_dict`` instead."""
import torch
import torch.nn as nn

from ... import ops as ops_mod
from ...squeezeseg import DEC_UP, ENC_DS, SqueezeSegBackbone
from ...rangenet import MODEL_BLOCKS
from .projector import ProjectionV1
from .salsanext_proto import SalsaNextProto

_ENC = [(32, 64), (64, 128), (128, 256), (256, 256), (256, 256)]
_DEC = [(5, 256, 256), (4, 256, 256), (3, 256, 128), (2, 128, 64), (1, 64, 32)]
# Dropout2d call sites in forward order (backbone.dropout x5, decoder.dropout, head5[0]): p = 0.01
_DROP_SITES = (("enc1", 64), ("enc2", 128), ("enc3", 256), ("enc4", 256), ("enc5", 256), ("decoder", 32), ("head", 32))
_DROP_P = 0.01


class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError("parameter container of the HIP backbone; call SqueezeSegV3Proto instead")


def _sac_block(c):
    blk = _Holder()
    blk.attention_x = nn.Sequential(nn.Conv2d(3, 9 * c, kernel_size=7, padding=3), nn.BatchNorm2d(9 * c, momentum=0.1))
    blk.position_mlp_2 = nn.Sequential(nn.Conv2d(9 * c, c, kernel_size=1), nn.BatchNorm2d(c, momentum=0.1), nn.ReLU(inplace=True),
                                       nn.Conv2d(c, c, kernel_size=3, padding=1), nn.BatchNorm2d(c, momentum=0.1),
                                       nn.ReLU(inplace=True))
    return blk


def _basic_block(inplanes, planes, bn_d):
    blk = _Holder()
    blk.conv1 = nn.Conv2d(inplanes, planes[0], kernel_size=1, bias=False)
    blk.bn1 = nn.BatchNorm2d(planes[0], momentum=bn_d)
    blk.conv2 = nn.Conv2d(planes[0], planes[1], kernel_size=3, padding=1, bias=False)
    blk.bn2 = nn.BatchNorm2d(planes[1], momentum=bn_d)
    return blk


class SqueezeSegV3Proto(SalsaNextProto):
    _late_dfeat_ok = False           # (this backbone's backward has no late embedding branch: coarse3d_amd/backbone.py)
    def __init__(self, nclasses, dataset="SemanticKitti", path=None, path_append="", strict=False, layers=21,
                 proj_dim=256, projection="v1", proj_feat="mix", l2_norm=True, proto_mom=0.999, ignore_label=0,
                 sub_proto_size=20, use_prototype=False, pred_3d=False):
        nn.Module.__init__(self)
        if layers not in MODEL_BLOCKS:
            raise AssertionError(f"layers must be one of {sorted(MODEL_BLOCKS)}")
        if projection != "v1":
            raise NotImplementedError                      # as the reference (:241-244)
        if path is not None:
            raise ValueError("pretrained sub-module files (path=...) are not read here: load a state_dict instead")
        self.nclasses, self.dataset, self.path, self.path_append, self.strict = nclasses, dataset, path, path_append, False
        self.layers = layers
        self.l2_norm, self.use_prototype, self.sub_proto_size = l2_norm, use_prototype, sub_proto_size
        self.ignore_label, self.proto_mom, self.projection, self.proj_feat = ignore_label, proto_mom, projection, proj_feat
        self.proj_dim = proj_dim
        bn_d = 0.01
        bb = _Holder()
        bb.conv1 = nn.Conv2d(5, 32, kernel_size=3, stride=1, padding=1, bias=False)
        bb.bn1 = nn.BatchNorm2d(32, momentum=bn_d)
        for i, ((ci, co), ds) in enumerate(zip(_ENC, ENC_DS), 1):
            enc = _Holder()
            for b in range(MODEL_BLOCKS[layers][i - 1]):
                enc.add_module(f"residual_{b}", _sac_block(ci))
            if ds:
                enc.conv = nn.Conv2d(ci, co, kernel_size=3, stride=(1, 2), padding=1, bias=False)
                enc.bn = nn.BatchNorm2d(co, momentum=bn_d)
            bb.add_module(f"enc{i}", enc)
        self.backbone = bb
        dec = _Holder()
        for i, ci, co in _DEC:
            d = _Holder()
            if DEC_UP[i]:
                d.upconv = nn.ConvTranspose2d(ci, co, kernel_size=(1, 4), stride=(1, 2), padding=(0, 1))
            else:
                d.conv = nn.Conv2d(ci, co, kernel_size=3, padding=1)
            d.bn = nn.BatchNorm2d(co, momentum=bn_d)
            d.residual = _basic_block(co, [ci, co], bn_d)
            dec.add_module(f"dec{i}", d)
        self.decoder = dec
        for k, ci in enumerate((256, 256, 128, 64), 1):
            h = _Holder()
            h.add_module("1", nn.Conv2d(ci, nclasses, kernel_size=1))
            self.add_module(f"head{k}", h)
        h5 = _Holder()
        h5.add_module("1", nn.Conv2d(32, nclasses, kernel_size=3, padding=1))
        self.head5 = h5
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
        return SqueezeSegBackbone(P, self.nclasses, self.dataset, reduce_fn, world, self._packs, self.layers,
                                  self._side_stream_for(P))

    def _list_trainable(self):
        unused = ("head1.", "head2.", "head3.", "head4.")           # parameters the reference's forward never touches
        return [(k, p) for k, p in self.named_parameters() if k not in self._SKIP and not k.startswith(unused)]

    def _check_input(self, h, w):
        assert w % 8 == 0, "input width must be a multiple of 8 (three stride-2 stages)"

    def _draw_masks(self, b, device):
        if self.dropout_masks is not None:
            return self.dropout_masks
        return {name: (torch.rand(b, c, device=device) >= _DROP_P).to(torch.float32) * (1.0 / (1.0 - _DROP_P))
                for name, c in _DROP_SITES}

    def forward(self, x, label=None, eval_mask=None, return_feat=True, proto_loss=False, proto_pl=None, unproj_data=None):
        return super().forward(x, label=label, eval_mask=eval_mask, return_feat=return_feat, proto_loss=proto_loss,
                               proto_pl=proto_pl)
