"""SqueezeSegV3 prototype backbone: explicit forward / backward over the HIP ops.

Mirrors the arithmetic of the reference ``SqueezeSegV3Proto.forward`` (pc_processor/models/
squeezesegv3_Proto.py: SACBlock :468-503, Backbone :515-682, BasicBlock :685-715, Decoder
:721-829, forward :353-465) on the engine built for SalsaNext / RangeNet.  What this family adds:

* the spatially-adaptive convolution (SAC): ``unfold(feature) * sigmoid(BN(conv7x7(xyz)))`` ->
  1x1 -> BN -> ReLU -> 3x3 -> BN -> ReLU, plus the input.  The 7x7 conv over the three coordinate
  channels runs as an im2col (built once per resolution level, shared by the level's blocks) +
  a 160 -> 9C pointwise MFMA GEMM with the BatchNorm statistics in its epilogue; the modulated
  unfold is one elementwise kernel (csrc/sac_ops.hip) producing the operand of the 1x1 GEMM;
* ReLU after BatchNorm = the engine's "BatchNorm then LeakyReLU(slope)" mode with a slope below
  fp32 resolution (1e-30), the sigmoid's BatchNorm = the same mode with slope 1 (identity);
* output stride 8: three stride-(1,2) stages, the last two encoder stages keep the width; the
  decoder's first two stages are plain 3x3 convs.
As in RangeNet the skips are detached (:653, :803) -- but the fourth embedding input is the live
backbone output (:415-417), so the embedding branch back-propagates into the encoder."""
from collections import OrderedDict

import torch

from . import contrast, ops
from .backbone import Act
from .rangenet import BN_MOM, MODEL_BLOCKS, SLOPE, RangeNetBackbone

RELU = 1e-30                     # LeakyReLU slope standing in for ReLU (below fp32 resolution of any activation)
SAC_BN_MOM = 0.1                 # SACBlock's BatchNorm2d layers use the PyTorch default (:476, :481, :485)
ENC_DS = (True, True, True, False, False)
DEC_UP = {5: False, 4: False, 3: True, 2: True, 1: True}


class SqueezeSegBackbone(RangeNetBackbone):
    def __init__(self, params, nclasses=20, dataset="SemanticKitti", reduce_fn=None, world_size=1, packs=None, layers=21,
                 side_stream=None):
        super().__init__(params, nclasses, dataset, reduce_fn, world_size, packs, layers, side_stream)
        self.blocks = MODEL_BLOCKS[layers]

    # ------------------------------------------------------------------ SAC block
    def _sac(self, name, xcol, feat):
        """feat: plain Act [B,H,W,C]; xcol: im2col of the level's xyz [B,H,W,160]."""
        c = feat.t.shape[3]
        w = self.P[f"{name}.attention_x.0.weight"]                       # [9C, 3, 7, 7]
        w_pad = torch.zeros(9 * c, 160, 1, 1, device=w.device, dtype=torch.float32)
        w_pad[:, :147, 0, 0] = w.reshape(9 * c, 147)
        dw_pad = torch.empty_like(w_pad) if self.train else None
        att = self._conv(f"{name}.attention_x.0", [xcol], 1, 1, 0, lrelu=False, bn=f"{name}.attention_x.1", slope=1.0,
                         bn_momentum=SAC_BN_MOM, weight=w_pad, dweight=dw_pad)
        m = Act(ops.sac_modulate(feat.t, att.t, att.scale, att.shift))
        p1 = self._conv(f"{name}.position_mlp_2.0", [m], 1, 1, 0, lrelu=False, bn=f"{name}.position_mlp_2.1", slope=RELU,
                        bn_momentum=SAC_BN_MOM)
        p2 = self._conv(f"{name}.position_mlp_2.3", [p1], 3, 1, 1, lrelu=False, bn=f"{name}.position_mlp_2.4",
                        src_lrelu=True, slope=RELU, bn_momentum=SAC_BN_MOM)
        out = Act(ops.affine_add(feat.t, p2.t, p2.scale, p2.shift, slope=RELU))
        self.tape[f"{name}.sac"] = (feat, att, m, p1, p2, out, dw_pad)
        return out

    def _sac_backward(self, name):
        feat, att, m, p1, p2, out, dw_pad = self.tape[f"{name}.sac"]
        g = out.grad
        out.grad = None
        self._conv_backward(f"{name}.position_mlp_2.3", g)              # g = d/d ReLU(BN(.))
        if feat.grad is None:
            feat.grad = g                                                # residual path: reuse the buffer
        else:
            ops.axpy(g, feat.grad)
        self._conv_backward(f"{name}.position_mlp_2.0", p1.grad)
        p1.grad = None
        datt = ops.sac_modulate_bwd(m.grad, feat.t, att.t, att.scale, att.shift)   # m.grad <- dm * sigmoid
        ops.sac_fold(m.grad, feat.grad, True)
        m.grad = None
        self._conv_backward(f"{name}.attention_x.0", datt)
        c = feat.t.shape[3]
        with self._fork():                                               # after the wgrad that fills dw_pad
            self.grads[f"{name}.attention_x.0.weight"].copy_(dw_pad[:, :147, 0, 0].reshape(9 * c, 3, 7, 7))

    # ------------------------------------------------------------------ forward
    def forward(self, x, train=True, dropout_masks=None, return_feat=True, update_running=True):
        """x [B,5,H,W] fp32 NCHW.  Returns dict with NHWC tensors: prob [B,H,W,C], logits [B,H,W,32],
        feat [B,H,W,256] (if return_feat)."""
        self.train, self.masks, self.update_running = train, dropout_masks, update_running
        self.packs.refresh()
        self.tape = OrderedDict()
        self.bn_seen = []
        b, _, h, w = x.shape
        assert w % 8 == 0, "W must be a multiple of 8 (three stride-2 stages)"
        x = x.contiguous()
        xin = Act(ops.nchw_to_nhwc_pad(x, 16))                           # 5 channels as a 16-channel MFMA operand
        xin.no_grad = True
        xyz = ops.nchw_to_nhwc_pad(x[:, 1:4].contiguous(), 4)            # feature[:, 1:4] (:660)
        w1 = self.P["backbone.conv1.weight"]
        dw1 = torch.empty(w1.shape[0], 16, 3, 3, device=w1.device) if train else None
        z = self._conv("backbone.conv1", [xin], 3, 1, 1, lrelu=False, bn="backbone.bn1", slope=SLOPE, bn_momentum=BN_MOM,
                       dweight=dw1)
        feature = self._materialise(z)
        self.tape["conv1.out"] = (feature, dw1)
        skips = {}
        os_ = 1
        xcol = None
        for i in range(1, 6):
            name = f"backbone.enc{i}"
            if xcol is None:
                xcol = Act(ops.sac_im2col7(xyz))
                xcol.no_grad = True
            cur = feature
            for bidx in range(self.blocks[i - 1]):
                cur = self._sac(f"{name}.residual_{bidx}", xcol, cur)
            if ENC_DS[i - 1]:                                             # run_layer(flag=True) :645-649
                cur = self._down(name, cur, False)
                xyz = ops.bilinear(xyz, h, xyz.shape[2] // 2)             # F.upsample_bilinear = align_corners=True
                xcol = None
                skips[os_] = feature                                      # the layer's INPUT, detached (:652-654)
                os_ *= 2
            m = self._mask(f"enc{i}")
            nxt = Act(ops.maskpool(cur.t, m, False)) if m is not None else cur
            self.tape[f"{name}.drop"] = (cur, nxt, m)
            feature = nxt
        t = feature
        for i in (5, 4, 3, 2, 1):
            name = f"decoder.dec{i}"
            if DEC_UP[i]:
                y = self._up(name, t)
            else:
                zc = self._conv(f"{name}.conv", [t], 3, 1, 1, lrelu=False, bn=f"{name}.bn", slope=SLOPE, bn_momentum=BN_MOM)
                y = self._materialise(zc)
                self.tape[f"{name}.cbr"] = (t, y)
            y = self._basic_block(f"{name}.residual", y)
            if DEC_UP[i]:
                os_ //= 2
                nxt = Act(ops.affine_add(y.t, skips[os_].t))
                self.tape[f"{name}.skip"] = (y, nxt)
                t = nxt
            else:
                t = y
        md, mh = self._mask("decoder"), self._mask("head")
        m = md * mh if (md is not None and mh is not None) else (md if md is not None else mh)
        th = Act(ops.maskpool(t.t, m, False)) if m is not None else t
        self.tape["head.drop"] = (t, th, m)
        logits = self._conv("head5.1", [th], 3, 1, 1, lrelu=False, cout_pad=32)
        prob = ops.softmax(logits.t, self.ncls, h, w)
        self._prob = prob
        out = {"prob": prob, "logits": logits.t}
        self.return_feat = return_feat
        if return_feat:
            hh, wh = h // 2, w // 2
            srcs = [skips[1], skips[2], skips[4], feature]                # :405-417; only the last one is live
            feat = torch.empty(b, hh, wh, sum(s.t.shape[3] for s in srcs), device=x.device, dtype=torch.float32)
            off = 0
            for s in srcs:
                ops.bilinear(s.t, hh, wh, dst=feat, dcoff=off, c=s.t.shape[3])
                off += s.t.shape[3]
            feat_a = Act(feat)
            z0 = self._conv("projector.proj.0", [feat_a], 1, 1, 0, lrelu=False, bn="projector.proj.1")
            emb = self._conv("projector.proj.3", [z0], 1, 1, 0, lrelu=False, src_lrelu=True)
            embn, norm = ops.l2norm(emb.t, 1e-12)
            out["feat"] = ops.bilinear(embn, h, w)
            self.tape["embed"] = (feat_a, z0, emb, embn, norm, feature, off - feature.t.shape[3])
        if train and update_running:
            torch._foreach_add_([self.P[f"{n}.num_batches_tracked"] for n in self.bn_seen], 1)
        return out

    # ------------------------------------------------------------------ backward
    def backward(self, d_prob=None, d_feat=None, grads=None):
        """d_prob [B,H,W,C], d_feat [B,H,W,256] (NHWC).  ``grads``: name -> preallocated gradient."""
        if grads is None:
            grads = {k: torch.empty_like(v) for k, v in self.P.items()
                     if v.is_floating_point() and v.dim() > 0 and not k.endswith(("running_mean", "running_var"))
                     and k != "prototypes" and not k.startswith(("feat_norm", "mask_norm", "head1", "head2", "head3", "head4"))}
            torch._foreach_zero_(list(grads.values()))      # a few multi-tensor launches instead of ~100 fills
        self.grads = grads

        def hook(tag):                                       # see Backbone.backward
            if self.on_block_done is not None:
                with self._fork():
                    self.on_block_done(tag)
        self.embed_ran = d_feat is not None and self.return_feat
        if self.embed_ran:
            feat_a, z0, emb, embn, norm, live, live_off = self.tape["embed"]
            d_embn = torch.empty_like(embn)
            d_feat = d_feat.contiguous()
            ops.bilinear_bwd(d_embn, d_feat, rowmask=contrast.take_row_hint(d_feat))
            d_emb = ops.l2norm_bwd(embn, norm, d_embn, 1e-12)
            self._conv_backward("projector.proj.3", d_emb)
            self._conv_backward("projector.proj.0", z0.grad)
            z0.grad = None
            live.grad = torch.empty_like(live.t)             # the backbone output is the one live embedding input
            ops.bilinear_bwd(live.grad, feat_a.grad, dcoff=live_off, c=live.t.shape[3])
            feat_a.grad = None
        else:
            for n in ("projector.proj.0", "projector.proj.1", "projector.proj.3"):
                for suffix in ("weight", "bias"):
                    grads[f"{n}.{suffix}"].zero_()
        hook("projector")
        if d_prob is None:
            raise ValueError("backward needs d_prob (the segmentation losses always produce it)")
        logits = self.tape["head5.1"].out
        dl = ops.softmax_bwd(self._prob, d_prob.contiguous(), tuple(logits.t.shape))
        self._conv_backward("head5.1", dl)
        t, th, m = self.tape["head.drop"]
        if th is not t:
            t.grad = ops.maskpool_bwd(th.grad, m, None, tuple(t.t.shape), False)
            th.grad = None
        hook("head5")
        for i in (1, 2, 3, 4, 5):
            name = f"decoder.dec{i}"
            if DEC_UP[i]:
                y, nxt = self.tape[f"{name}.skip"]
                y.grad = nxt.grad                            # the skip is detached: the sum passes its gradient on
                nxt.grad = None
                self._basic_block_backward(f"{name}.residual")
                self._up_backward(name)
            else:
                self._basic_block_backward(f"{name}.residual")
                tin, y = self.tape[f"{name}.cbr"]
                self._conv_backward(f"{name}.conv", y.grad)  # dgrad accumulates into tin.grad (embedding branch first)
                y.grad = None
            hook(name)
        for i in (5, 4, 3, 2, 1):
            name = f"backbone.enc{i}"
            cur, nxt, m = self.tape[f"{name}.drop"]
            if nxt is not cur:
                cur.grad = ops.maskpool_bwd(nxt.grad, m, None, tuple(cur.t.shape), False)
                nxt.grad = None
            if ENC_DS[i - 1]:
                self._down_backward(name)
            for bidx in reversed(range(self.blocks[i - 1])):
                self._sac_backward(f"{name}.residual_{bidx}")
            hook(name)
        feature, dw1 = self.tape["conv1.out"]
        self._conv_backward("backbone.conv1", feature.grad)
        feature.grad = None
        with self._fork():
            grads["backbone.conv1.weight"].copy_(dw1[:, :grads["backbone.conv1.weight"].shape[1]])
        hook("backbone.conv1")
        self._join()
        self.tape = None
        return grads
