"""RangeNet (Darknet-21/53) prototype backbone: explicit forward / backward over the HIP ops
(SURVEY 8f, N3).

Mirrors the arithmetic of the reference ``RangeNetProto.forward`` (pc_processor/models/
rangenet_proto.py: BasicBlock :38-63, Backbone :76-259, Decoder :261-372, forward :573-676) on
the engine built for SalsaNext (coarse3d_amd/backbone.py), with the three things this family
adds:

* conv -> BatchNorm -> LeakyReLU(0.1) order: the conv epilogue emits raw outputs + statistics,
  consumers apply affine + activation while staging (``src_lrelu`` with slope 0.1) and the
  BatchNorm backward runs in its "BN then activation" mode;
* stride-(1,2) 3x3 convs and ConvTranspose2d([1,4],[1,2],[0,1]) as stride-1 convs over COLUMN-PAIR VIEWS of their input /
  output ([B,H,W,C] read as [B,H,W/2,2C]: the same memory) with six / three taps -- no resampling pass, no zero-filled or
  full-width intermediate (``_down`` / ``_up`` below);
* skip connections are DETACHED in the reference (``skips[os] = x.detach()`` :217, ``skips[os]
  .detach()`` :353, and the 480-channel embedding input): no gradient flows through them.
"""
from collections import OrderedDict

import torch

from . import contrast, ops
from .backbone import Act, Backbone

SLOPE = 0.1                      # nn.LeakyReLU(0.1)
BN_MOM = 0.01                    # Backbone.bn_d / Decoder.bn_d
MODEL_BLOCKS = {21: [1, 1, 2, 2, 1], 53: [1, 2, 8, 8, 4]}
DOWN_TAPS = [(dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0)]      # stride-(1, 2) 3x3 conv over the column-pair view of its input
UP_TAPS = [(0, -1), (0, 0), (0, 1)]                                # ConvTranspose2d([1, 4], [1, 2], [0, 1]) onto the pair view of its output
DROP_SITES = ("enc1", "enc2", "enc3", "enc4", "enc5", "decoder", "head")


class RangeNetBackbone(Backbone):
    def __init__(self, params, nclasses=20, dataset="SemanticKitti", reduce_fn=None, world_size=1, packs=None, layers=21,
                 side_stream=None):
        super().__init__(params, nclasses, dataset, reduce_fn, world_size, packs, side_stream)
        self.blocks = MODEL_BLOCKS[layers]

    # ------------------------------------------------------------------ forward helpers
    def _materialise(self, a):
        """Act with a pending BatchNorm affine + LeakyReLU -> plain Act (one elementwise pass)."""
        return Act(ops.affine_add(None, a.t, a.scale, a.shift, slope=SLOPE))

    def _basic_block(self, name, r):
        """r: plain Act.  1x1 -> BN -> LReLU -> 3x3 -> BN -> LReLU, + r (rangenet_proto.py:52-63)."""
        a1 = self._conv(f"{name}.conv1", [r], 1, 1, 0, lrelu=False, bn=f"{name}.bn1", slope=SLOPE, bn_momentum=BN_MOM)
        a2 = self._conv(f"{name}.conv2", [a1], 3, 1, 1, lrelu=False, bn=f"{name}.bn2", src_lrelu=True, slope=SLOPE,
                        bn_momentum=BN_MOM)
        out = Act(ops.affine_add(r.t, a2.t, a2.scale, a2.shift, slope=SLOPE))
        self.tape[f"{name}.out"] = (r, a1, a2, out)
        return out

    def _basic_block_backward(self, name):
        r, a1, a2, out = self.tape[f"{name}.out"]
        g = out.grad
        out.grad = None
        self._conv_backward(f"{name}.conv2", g)             # g = d/d LReLU(BN2(.)); BN backward mode 1
        r.grad = g                                          # residual path: reuse the buffer, conv1's dgrad accumulates
        self._conv_backward(f"{name}.conv1", a1.grad)
        a1.grad = None

    # ---- stride-(1, 2) / transposed convs over COLUMN-PAIR VIEWS (round 5).  An NHWC tensor [B, H, W, C] read as
    # [B, H, W/2, 2C] is the same memory: pair x' holds column 2x' in channels 0 .. C-1 and column 2x'+1 in C .. 2C-1.
    #   * Conv2d(3x3, stride (1, 2), padding 1): out[y, x'] reads columns 2x'-1, 2x', 2x'+1 = (pair x'-1, odd half),
    #     (pair x', even half), (pair x', odd half): a stride-1 conv with the SIX taps (dy, dx') in {-1,0,1} x {-1,0} over 2C
    #     channels, whose weight [Cout, 2C, 3, 2] is the 3x3 weight re-indexed (a quarter of it zero).  12 C products per
    #     output instead of the 18 C of "stride-1 conv, then drop every other column", no full-width intermediate, no resampling.
    #   * ConvTranspose2d([1, 4], stride [1, 2], padding [0, 1]): out[2m] = in[m] w1 + in[m-1] w3, out[2m+1] = in[m+1] w0 +
    #     in[m] w2: the output's pair view [B, H, W, 2 Cout] is a stride-1 conv of the INPUT with the three column taps
    #     -1, 0, +1 and the weight [2 Cout, Cin, 1, 3] = ((w3 | 0), (w1 | w2), (0 | w0)); no zero-filled input.
    # Input and weight gradients are the engine's ordinary ones in view space (the gradient of a view is the view of the
    # gradient); the parameter's gradient is the same re-indexing read backwards.  The re-indexing is one gather per
    # layer and direction over an index built once (``_pair_index``).
    def _pair_index(self, name, kind, w):
        cache = self.packs.__dict__.setdefault("pair_idx", {})      # (the pack cache outlives the per-step backbone object)
        key = (name, kind, tuple(w.shape))
        if key not in cache:
            n = w.numel()
            src = torch.arange(n, dtype=torch.int64).view(w.shape)          # flat index of every parameter element
            if kind == "down":                                              # [Cout, C, 3, 3] -> [Cout, 2C, 3, 2]
                co, c = w.shape[:2]
                idx = torch.full((co, 2 * c, 3, 2), n, dtype=torch.int64)   # n = the zero appended behind the parameter
                idx[:, c:, :, 0] = src[:, :, :, 0]
                idx[:, :c, :, 1] = src[:, :, :, 1]
                idx[:, c:, :, 1] = src[:, :, :, 2]
            else:                                                           # [Cin, Cout, 1, 4] -> [2 Cout, Cin, 1, 3]
                ci, co = w.shape[:2]
                k = src[:, :, 0, :].permute(1, 0, 2)                        # [Cout, Cin, 4]
                idx = torch.full((2 * co, ci, 1, 3), n, dtype=torch.int64)
                idx[:co, :, 0, 0] = k[:, :, 3]
                idx[:co, :, 0, 1] = k[:, :, 1]
                idx[co:, :, 0, 1] = k[:, :, 2]
                idx[co:, :, 0, 2] = k[:, :, 0]
            flat = idx.reshape(-1)
            inv = torch.empty(n, dtype=torch.int64)                         # where each parameter element sits in the view weight
            sel = flat < n
            inv[flat[sel]] = torch.nonzero(sel).reshape(-1)
            cache[key] = (tuple(idx.shape), flat.to(w.device), inv.to(w.device))
        return cache[key]

    def _pair_weight(self, name, kind, w):
        shape, flat, _ = self._pair_index(name, kind, w)
        return torch.cat([w.reshape(-1), w.new_zeros(1)])[flat].view(shape)

    def _pair_weight_grad(self, name, kind, w, dw_view, out):
        _, _, inv = self._pair_index(name, kind, w)
        torch.index_select(dw_view.reshape(-1), 0, inv, out=out.view(-1))

    def _down(self, name, src, src_pending):
        """stride-(1,2) 3x3 conv (no bias) -> BN -> LReLU, materialised (rangenet_proto.py:194-206)."""
        w = self.P[f"{name}.conv.weight"]
        b, h, wd, c = src.t.shape
        v = Act(src.t.view(b, h, wd // 2, 2 * c))
        if src_pending:
            v.scale, v.shift = src.scale.repeat(2), src.shift.repeat(2)
        v.no_grad = src.no_grad
        w2 = self._pair_weight(name, "down", w)
        dw2 = torch.empty_like(w2) if self.train else None
        z = self._conv(f"{name}.conv", [v], 3, 1, 1, lrelu=False, bn=f"{name}.bn", src_lrelu=src_pending, taps=DOWN_TAPS,
                       slope=SLOPE, bn_momentum=BN_MOM, weight=w2, dweight=dw2)
        out = self._materialise(z)
        self.tape[f"{name}.down"] = (src, v, out, dw2)
        return out

    def _down_backward(self, name):
        src, v, out, dw2 = self.tape[f"{name}.down"]
        self._conv_backward(f"{name}.conv", out.grad)
        out.grad = None
        with self._fork():                                   # after the weight-gradient launch that fills dw2
            self._pair_weight_grad(name, "down", self.P[f"{name}.conv.weight"], dw2, self.grads[f"{name}.conv.weight"])
        if v.grad is not None:
            g = v.grad.view(src.t.shape)
            src.grad = g if src.grad is None else src.grad.add_(g)
            v.grad = None

    def _up(self, name, t):
        """ConvTranspose2d([1,4], stride [1,2], padding [0,1]) + bias -> BN -> LReLU, materialised."""
        w_t = self.P[f"{name}.upconv.weight"]               # [Cin, Cout, 1, 4]
        cin, cout = w_t.shape[:2]
        b, h, wd, _ = t.t.shape
        w3 = self._pair_weight(name, "up", w_t)             # [2 Cout, Cin, 1, 3]
        z2, part = ops.conv_forward([t.src()], ops.pack_weights(w3, 0), self.P[f"{name}.upconv.bias"].repeat(2), 2 * cout, UP_TAPS,
                                    lrelu=False, stats=self.train, slope=SLOPE)
        z = z2.view(b, h, 2 * wd, cout)
        if self.train:       # the statistics of channel c are those of view channels c and Cout + c: [2 Cout, 2, n] -> [Cout, 2, 2n]
            n = part.shape[2]
            part = part.view(2, cout, 2, n).permute(1, 2, 0, 3).reshape(cout, 2, 2 * n).contiguous()    # (n = 1: reshape returns a strided view)
        bn = self._bn_forward(f"{name}.bn", part if self.train else None, cout, b * h * 2 * wd, BN_MOM)
        out = Act(ops.affine_add(None, z, bn.scale, bn.shift, slope=SLOPE))
        if t.first_consumer is None:
            t.first_consumer = f"{name}.upconv"
        self.tape[f"{name}.up"] = (t, z, bn, out, w3)
        return out

    def _up_backward(self, name):
        t, z, bn, out, w3 = self.tape[f"{name}.up"]
        cout = z.shape[3]
        cin = t.t.shape[3]
        b, h, wd, _ = t.t.shape
        dz, pz = self._bn_backward(bn, out.grad, z, cout, 1, SLOPE)
        out.grad = None
        dz2 = dz.view(b, h, wd, 2 * cout)
        dw3 = torch.empty_like(w3)
        with self._fork(dz, pz):                             # weight-gradient chain: side stream (Backbone._conv_backward)
            ops.bias_from_partials(pz, self.grads[f"{name}.upconv.bias"])
            ops.conv_wgrad(t.src(), dz2, dw3, UP_TAPS, slope=SLOPE)
            self._pair_weight_grad(name, "up", self.P[f"{name}.upconv.weight"], dw3, self.grads[f"{name}.upconv.weight"])
        if not t.no_grad:
            wd_ = ops.pack_weights(w3, 1, c_off=0, c_cnt=cin, kpad=(2 * cout + 15) // 16 * 16)
            acc = t.grad is not None
            if not acc:
                t.grad = torch.empty_like(t.t)
            ops.conv_forward([ops.Source(dz2)], wd_, None, cin, ops.negate_taps(UP_TAPS), out=t.grad, accumulate=acc, grad=True)

    # ------------------------------------------------------------------ forward
    def forward(self, x, train=True, dropout_masks=None, return_feat=True, update_running=True):
        """x [B,5,H,W] fp32 NCHW.  Returns dict with NHWC tensors: prob [B,H,Wo,C], logits
        [B,H,W,32], feat [B,H,W,256] (if return_feat)."""
        self.train, self.masks, self.update_running = train, dropout_masks, update_running
        self.packs.refresh()
        self.tape = OrderedDict()
        self.bn_seen = []
        ho, wo = x.shape[2], x.shape[3]
        if self.dataset == "SemanticPOSS":                   # rangenet_proto.py:586-590
            x = torch.nn.functional.pad(x, (0, 24))
        assert x.shape[3] % 32 == 0, "W must be a multiple of 32 (five stride-2 stages)"
        xin = Act(ops.nchw_to_nhwc_pad(x.contiguous(), 16))  # 5 channels as a 16-channel MFMA operand
        xin.no_grad = True
        w1 = self.P["backbone.conv1.weight"]
        dw1 = torch.empty(w1.shape[0], 16, 3, 3, device=w1.device) if train else None
        t = self._conv("backbone.conv1", [xin], 3, 1, 1, lrelu=False, bn="backbone.bn1", slope=SLOPE, bn_momentum=BN_MOM,
                       dweight=dw1)
        self.tape["conv1.dw"] = dw1
        t_pending = True
        skips = {}
        os_ = 1
        for i in range(1, 6):
            name = f"backbone.enc{i}"
            cur = self._down(name, t, t_pending)
            for bidx in range(self.blocks[i - 1]):
                cur = self._basic_block(f"{name}.residual_{bidx}", cur)
            skips[os_] = (t, t_pending)
            os_ *= 2
            m = self._mask(f"enc{i}")
            nxt = Act(ops.maskpool(cur.t, m, False)) if m is not None else cur
            self.tape[f"{name}.drop"] = (cur, nxt, m)
            t, t_pending = nxt, False
        for i in (5, 4, 3, 2, 1):
            name = f"decoder.dec{i}"
            y = self._up(name, t)
            y = self._basic_block(f"{name}.residual", y)
            os_ //= 2
            sk, sk_pending = skips[os_]
            nxt = Act(ops.affine_add(y.t, sk.t, sk.scale if sk_pending else None, sk.shift if sk_pending else None,
                                     slope=SLOPE if sk_pending else 0.0))
            self.tape[f"{name}.skip"] = (y, nxt)
            t = nxt
        md, mh = self._mask("decoder"), self._mask("head")
        m = md * mh if (md is not None and mh is not None) else (md if md is not None else mh)
        th = Act(ops.maskpool(t.t, m, False)) if m is not None else t
        self.tape["head.drop"] = (t, th, m)
        logits = self._conv("head.1", [th], 3, 1, 1, lrelu=False, cout_pad=32)
        prob = ops.softmax(logits.t, self.ncls, ho, wo)
        self._prob = prob
        out = {"prob": prob, "logits": logits.t}
        self.return_feat = return_feat
        if return_feat:
            b, hp, wp = x.shape[0], x.shape[2], x.shape[3]
            hh, wh = ho // 2, wo // 2
            srcs = []
            for k in (1, 2, 4, 8):
                sk, pend = skips[k]
                srcs.append(self._materialise(sk).t if pend else sk.t)
            feat = torch.empty(b, hh, wh, sum(s.shape[3] for s in srcs), device=x.device, dtype=torch.float32)
            off = 0
            for s in srcs:
                ops.bilinear(s, hh, wh, dst=feat, dcoff=off, c=s.shape[3])
                off += s.shape[3]
            feat_a = Act(feat)
            feat_a.no_grad = True                            # the skips are detached: nothing upstream
            z0 = self._conv("projector.proj.0", [feat_a], 1, 1, 0, lrelu=False, bn="projector.proj.1")
            emb = self._conv("projector.proj.3", [z0], 1, 1, 0, lrelu=False, src_lrelu=True)
            embn, norm = ops.l2norm(emb.t, 1e-12)
            out["feat"] = ops.bilinear(embn, hp, wp)
            self.tape["embed"] = (feat_a, z0, emb, embn, norm)
        if train and update_running:
            torch._foreach_add_([self.P[f"{n}.num_batches_tracked"] for n in self.bn_seen], 1)
        return out

    # ------------------------------------------------------------------ backward
    def backward(self, d_prob=None, d_feat=None, grads=None):
        """d_prob [B,H,Wo,C], d_feat [B,H,W,256] (NHWC).  ``grads``: name -> preallocated gradient."""
        if grads is None:
            grads = {k: torch.empty_like(v) for k, v in self.P.items()
                     if v.is_floating_point() and v.dim() > 0 and not k.endswith(("running_mean", "running_var"))
                     and k != "prototypes" and not k.startswith(("feat_norm", "mask_norm"))}
            torch._foreach_zero_(list(grads.values()))      # a few multi-tensor launches instead of ~100 fills
        self.grads = grads
        def hook(tag):                                       # see Backbone.backward
            if self.on_block_done is not None:
                with self._fork():
                    self.on_block_done(tag)
        self.embed_ran = d_feat is not None and self.return_feat
        if self.embed_ran:
            feat_a, z0, emb, embn, norm = self.tape["embed"]
            d_embn = torch.empty_like(embn)
            d_feat = d_feat.contiguous()
            ops.bilinear_bwd(d_embn, d_feat, rowmask=contrast.take_row_hint(d_feat))
            d_emb = ops.l2norm_bwd(embn, norm, d_embn, 1e-12)
            self._conv_backward("projector.proj.3", d_emb)
            self._conv_backward("projector.proj.0", z0.grad)
            z0.grad = None
        else:
            for n in ("projector.proj.0", "projector.proj.1", "projector.proj.3"):
                for suffix in ("weight", "bias"):
                    grads[f"{n}.{suffix}"].zero_()
        hook("projector")
        if d_prob is None:
            raise ValueError("backward needs d_prob (the segmentation losses always produce it)")
        logits = self.tape["head.1"].out
        dl = ops.softmax_bwd(self._prob, d_prob.contiguous(), tuple(logits.t.shape))
        self._conv_backward("head.1", dl)
        t, th, m = self.tape["head.drop"]
        if th is not t:
            t.grad = ops.maskpool_bwd(th.grad, m, None, tuple(t.t.shape), False)
            th.grad = None
        hook("head")
        for i in (1, 2, 3, 4, 5):
            name = f"decoder.dec{i}"
            y, nxt = self.tape[f"{name}.skip"]
            y.grad = nxt.grad                                # the skip is detached: the sum passes its gradient on
            nxt.grad = None
            self._basic_block_backward(f"{name}.residual")
            self._up_backward(name)
            hook(f"decoder.dec{i}")
        for i in (5, 4, 3, 2, 1):
            name = f"backbone.enc{i}"
            cur, nxt, m = self.tape[f"{name}.drop"]
            if nxt is not cur:
                cur.grad = ops.maskpool_bwd(nxt.grad, m, None, tuple(cur.t.shape), False)
                nxt.grad = None
            for bidx in reversed(range(self.blocks[i - 1])):
                self._basic_block_backward(f"{name}.residual_{bidx}")
            self._down_backward(name)
            hook(f"backbone.enc{i}")
        rec = self.tape["backbone.conv1"]
        self._conv_backward("backbone.conv1", rec.out.grad)
        rec.out.grad = None
        dw1 = self.tape["conv1.dw"]
        with self._fork():
            grads["backbone.conv1.weight"].copy_(dw1[:, :grads["backbone.conv1.weight"].shape[1]])
        hook("backbone.conv1")
        self._join()
        self.tape = None
        return grads
