"""SalsaNext-style range-image backbone: hand-orchestrated forward / backward over the HIP ops.

Mirrors the arithmetic of the reference ``SalsaNextProto.forward`` encoder / decoder / head /
embedding branch (pc_processor/models/salsanext_proto.py:423-492, blocks :38-212,
projector.py:11-27) with an MI355X-first dataflow:

* activations are NHWC fp32 and every conv stores its *pre-BatchNorm* output ``a``; the
  BatchNorm affine (scale, shift) is applied by the consumers while they stage their input
  tiles in LDS, so the 43 normalised tensors never exist in HBM;
* batch statistics come out of the conv epilogue as per-tile partials (deterministic two-stage
  reduction, fp64 fold) -- the hook ``reduce_fn`` lets data-parallel ranks all-reduce the
  fp64 sums (SyncBatchNorm semantics, tasks/weak_segmentation/trainer.py:54);
* backward is explicit (no autograd tape): per layer BN-backward reduce -> coefficients ->
  dz, then MFMA wgrad and MFMA dgrad (the forward conv kernel with transposed weights).

Parameter tensors are taken by *reference-compatible names* (``downCntx.conv1.weight`` ...)
in OIHW layout, gradients are produced in the same layout.
"""
import contextlib
import weakref
import os
from collections import OrderedDict

import torch

from . import contrast, ops

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
# BatchNorm-backward sums in the epilogue of the last input-gradient conv (see _conv_backward); False: always the
# separate reduce pass (module flag for tests / A-B runs)
FUSE_BN_REDUCE = os.environ.get("C3D_FUSE_BN_REDUCE", "1") != "0"      # (the environment switch: same-box A/B runs)
# The same on the bf16 engine (round 5: ConvArgs::stat_mul over bf16 tensors, tests/test_gpu_bf16_storage.py).  OFF by default:
# measured on BASELINE configs[2] it removes 29 of the 43 reduce passes (1.12 -> 0.44 ms) but the launches that carry the
# epilogue cost more than the passes they replace -- first form (sixteen 2-byte multiplier loads per lane and sub-tile)
# conv_x3f<2,2,9,..> 79 -> 180 us, with the multiplier tile staged through LDS by 16-byte loads 79 -> 130 us, against ~24 us for
# the separate pass it replaces (which already runs at 4.9 TB/s): 18.20 vs 17.83 ms of kernels per step, 469 vs 478 img/s.
# Round 6: the multiplier tile now arrives by LDS-DMA under the last K chunk (conv_common.h: conv_mul_dma_issue) and the
# 64-cout launches keep their line-contiguous store order: conv_x3f<2,2,9,..> 130 -> 115 us (78 without the epilogue), 17.06 vs
# 16.84 ms of kernels per step, 473.8 vs 479.4 img/s (same box, alternating) -- still a loss, still off.  What the epilogue costs
# is not latency any more but the read of the multiplier tensor itself (134 MB per 64-channel full-resolution layer) plus a
# workgroup per CU where the tile's 16-32 KB of LDS cost one: the separate pass only re-reads dy on top, much of it from L2 / MALL.
FUSE_BN_REDUCE_BF16 = os.environ.get("C3D_FUSE_BN_REDUCE_BF16", "0") == "1"
# BatchNorm / LeakyReLU backward applied ON LOAD by the layer's first weight-gradient launch (round 4; ops.conv_wgrad(fuse=...)):
# the apply pass (dy, a -> dz: three tensor passes at HBM speed, 53 launches and the largest kernel of the round-3 step)
# disappears; dz is bit-identical.  False: the separate c3d_bn_bwd_apply pass (module flag for tests / A-B runs)
FUSE_BN_APPLY = True
# strips -> dw folds of the weight gradients queued and run in batches of 32 per launch (ops.WgradFolds); False: one fold
# launch behind every weight-gradient launch
DEFER_WGRAD_FOLDS = True


class Act:
    """An activation tensor as consumers see it: raw NHWC tensor + optional affine of its BN."""
    __slots__ = ("t", "scale", "shift", "grad", "mask", "no_grad", "producer", "first_consumer", "bwd_partial", "bn_alias")

    def __init__(self, t, scale=None, shift=None, mask=None):
        self.t, self.scale, self.shift = t, scale, shift
        self.grad = None      # gradient w.r.t. the affine-transformed value (NHWC, same shape)
        self.mask = mask      # Dropout2d multiplier [B,C] applied by the consumer (UpBlock out)
        self.no_grad = False  # True: nothing upstream needs d(loss)/d(this) (network input, detached skips)
        self.producer = None        # weakref to the conv record whose output this is (the tape holds the record: a strong
                                    # reference here made record <-> activation cycles, and EVERY activation of a launch-by-launch
                                    # step waited for Python's cyclic collector -- 285 GB after a few dozen full-size steps)
        self.first_consumer = None  # name of the first conv (forward order) that reads it = the LAST one to add to its gradient
        self.bwd_partial = None     # (sum dy, sum dy*a) partials taken in the epilogue of that last input-gradient launch
        self.bn_alias = None        # a residual sum x + BN(a): the Act of a -- its gradient IS the gradient at that BatchNorm's output

    def src(self, lrelu=False):
        return ops.Source(self.t, self.scale, self.shift, lrelu=lrelu)


class _ConvRec:
    __slots__ = ("name", "srcs", "src_lrelu", "taps", "cout", "mode", "bn", "out", "stats", "slope", "weight",
                 "dweight", "__weakref__")


class _BNRec:
    __slots__ = ("name", "mean", "invstd", "scale", "shift", "count")


class Backbone:
    def __init__(self, params, nclasses=20, dataset="SemanticKitti", reduce_fn=None, world_size=1, packs=None,
                 side_stream=None):
        """params: mapping name -> CUDA tensor (the module's parameters and buffers).
        side_stream: second HIP stream for the weight-gradient chain of the backward pass."""
        self.P = params
        # The weight-gradient chain can run on a side stream, under the exchange latency of a data-parallel step (SyncBN
        # statistics: 43 blocking exchanges in backward).  Opt-in (C3D_WGRAD_STREAM=1) since round 5, data parallel too:
        # kernels of two streams share the CUs and the BatchNorm backward cannot ride on the weight gradient's loads then --
        # the 1-rank data-parallel step measured 32.6 ms with it and 31.5 without (same box; the plain step 34.9 vs 31.4 when
        # forced on), i.e. it pays only where an exchange costs more than ~25 us, and the peer-memory exchange
        # (coarse3d_amd/peer.py) is a 4 us kernel plus the link latency.
        mode = os.environ.get("C3D_WGRAD_STREAM", "auto")
        use_side = mode == "1"
        self.side = side_stream if use_side else None
        self.ncls = nclasses
        self.dataset = dataset
        self.reduce_fn = reduce_fn
        self.world = world_size
        self.packs = packs if packs is not None else ops.PackCache()
        self.tape = None
        self.capture = None           # test hook: dict -> per conv layer (record, dy, dz) of the backward pass
        self.on_block_done = None     # data parallel: called with a block tag as soon as that
                                      # block's parameter gradients are final (backward order)

    # ------------------------------------------------------------------ forward helpers
    def _bn_forward(self, name, partial, c, count, momentum=BN_MOMENTUM):
        P = self.P
        rec = _BNRec()
        rec.name = name
        if self.train:
            rm = P[f"{name}.running_mean"] if self.update_running else None
            rv = P[f"{name}.running_var"] if self.update_running else None
            if self.reduce_fn is None:      # single rank: fold + finalize in one launch
                rec.scale, rec.shift, rec.mean, rec.invstd = ops.bn_finalize_partials(
                    partial, count, P[f"{name}.weight"], P[f"{name}.bias"], rm, rv, momentum, BN_EPS)
            elif getattr(self.reduce_fn, "bn_forward", None) is not None and self.reduce_fn.fits_channels(c) and partial.dtype == torch.float32:
                # SyncBN through peer memory: fold + exchange + finalize in one launch (coarse3d_amd/peer.py)
                count = count * self.world
                rec.scale, rec.shift, rec.mean, rec.invstd = self.reduce_fn.bn_forward(
                    partial, count, P[f"{name}.weight"], P[f"{name}.bias"], rm, rv, momentum, BN_EPS)
            else:                           # SyncBN: all-reduce the fp64 sums in between
                sums = ops.stat_reduce(partial, c)
                self.reduce_fn(sums)
                count = count * self.world
                rec.scale, rec.shift, rec.mean, rec.invstd = ops.bn_finalize(
                    sums, count, P[f"{name}.weight"], P[f"{name}.bias"], rm, rv, momentum, BN_EPS)
            rec.count = count
            self.bn_seen.append(name)
        else:
            rec.scale, rec.shift = ops.bn_eval_affine(P[f"{name}.weight"], P[f"{name}.bias"],
                                                      P[f"{name}.running_mean"], P[f"{name}.running_var"], BN_EPS)
            rec.mean = rec.invstd = None
            rec.count = count
        return rec

    def _bn_forward_group(self, pending):
        """SyncBN for several layers with no data dependence between them: their fp64 sums travel
        in ONE all-reduce.  pending: list of (conv record, bn name, partial, count, momentum)."""
        self._bn_group_end(self._bn_group_begin(pending))

    def _bn_group_begin(self, pending):
        """First half of ``_bn_forward_group``: fold the partials and START the exchange (asynchronously when the
        exchange hook offers ``begin`` / ``end``: the caller queues independent kernels before ``_bn_group_end``)."""
        if not (self.train and self.reduce_fn is not None):
            return (pending, None, None)
        dev = pending[0][2].device
        buf = torch.empty(sum(p[0].cout for p in pending), 2, device=dev, dtype=torch.float64)
        off = 0
        for rec, bn, partial, count, mom in pending:
            ops.stat_reduce(partial, rec.cout, sums=buf[off:off + rec.cout])
            off += rec.cout
        begin = getattr(self.reduce_fn, "begin", None)
        if begin is not None:
            return (pending, buf, ("async", begin(buf)))
        self.reduce_fn(buf)
        return (pending, buf, None)

    def _bn_group_end(self, state):
        pending, buf, work = state
        P = self.P
        if buf is None:
            for rec, bn, partial, count, mom in pending:
                rec.bn = self._bn_forward(bn, partial, rec.cout, count, mom)
                rec.out.scale, rec.out.shift = rec.bn.scale, rec.bn.shift
            return
        if work is not None:
            self.reduce_fn.end(work[1])
        off = 0
        for rec, bn, partial, count, mom in pending:
            r = _BNRec()
            r.name, r.count = bn, count * self.world
            rm = P[f"{bn}.running_mean"] if self.update_running else None
            rv = P[f"{bn}.running_var"] if self.update_running else None
            r.scale, r.shift, r.mean, r.invstd = ops.bn_finalize(
                buf[off:off + rec.cout], r.count, P[f"{bn}.weight"], P[f"{bn}.bias"], rm, rv, mom, BN_EPS)
            off += rec.cout
            self.bn_seen.append(bn)
            rec.bn = r
            rec.out.scale, rec.out.shift = r.scale, r.shift

    def _conv(self, name, srcs, k, dil, pad, lrelu=True, bn=None, src_lrelu=False, cout_pad=None, taps=None,
              slope=0.0, bn_momentum=BN_MOMENTUM, weight=None, dweight=None, defer_bn=None):
        """srcs: list[Act].  Returns Act of the conv output (pre-BN tensor + BN affine).
        ``defer_bn``: a list -- the BatchNorm finalisation is postponed and queued there for
        ``_bn_forward_group`` (the returned Act gets its affine then).
        ``taps`` overrides the k x k pattern; ``slope`` is the LeakyReLU slope of the on-load and
        epilogue activations (0 = 0.01); ``weight`` (OIHW) overrides ``P[name.weight]`` for layers
        whose parameter is stored in another layout, ``dweight`` then receives its gradient."""
        w = self.P[f"{name}.weight"] if weight is None else weight
        cout = w.shape[0]
        taps = ops.conv_taps(k, k, dil, pad) if taps is None else taps
        wp = self.packs.get(w, 0) if weight is None else ops.pack_weights(w, 0)
        b, h, wd = srcs[0].t.shape[:3]
        out = None
        if cout_pad is not None and cout_pad != cout:
            out = torch.zeros(b, h, wd, cout_pad, device=w.device, dtype=torch.float32)
        need_stats = bn is not None and self.train
        y, partial = ops.conv_forward([s.src(src_lrelu) for s in srcs], wp, self.P.get(f"{name}.bias"), cout, taps,
                                      lrelu=lrelu, stats=need_stats, out=out, slope=slope)
        rec = _ConvRec()
        rec.name, rec.srcs, rec.src_lrelu, rec.taps, rec.cout = name, srcs, src_lrelu, taps, cout
        rec.slope, rec.weight, rec.dweight = slope, weight, dweight
        rec.mode = 0 if (lrelu and bn) else (2 if lrelu else (1 if bn else 3))
        for s_ in srcs:
            if s_.first_consumer is None:
                s_.first_consumer = name
        if bn is not None and defer_bn is not None:
            rec.bn = None
            rec.out = Act(y)
            defer_bn.append((rec, bn, partial, b * h * wd, bn_momentum))
        else:
            rec.bn = self._bn_forward(bn, partial, cout, b * h * wd, bn_momentum) if bn is not None else None
            rec.out = Act(y, rec.bn.scale if rec.bn else None, rec.bn.shift if rec.bn else None)
        rec.out.producer = weakref.ref(rec)
        self.tape[name] = rec
        return rec.out

    def _mask(self, name):
        if not self.train or self.masks is None:
            return None
        return self.masks.get(name)

    # ------------------------------------------------------------------ blocks (forward)
    def _ctx_block(self, name, xin, first=False):
        if first:
            w = self.P[f"{name}.conv1.weight"]
            s_t = ops.conv_in5(xin, w.reshape(w.shape[0], -1).contiguous(), self.P[f"{name}.conv1.bias"])
            s = Act(s_t)
            self.tape[f"{name}.conv1"] = ("in5", xin, s)
        else:
            s = self._conv(f"{name}.conv1", [xin], 1, 1, 0, lrelu=True)
        a1 = self._conv(f"{name}.conv2", [s], 3, 1, 1, bn=f"{name}.bn1")
        a2 = self._conv(f"{name}.conv3", [a1], 3, 2, 2, bn=f"{name}.bn2")
        out = Act(ops.affine_add(s.t, a2.t, a2.scale, a2.shift))
        out.bn_alias = a2           # d(out) is d(BN(a2)): the last input-gradient launch into out can take bn2's backward sums
        self.tape[f"{name}.out"] = (s, a2, out)
        return out

    def _res_block(self, name, xin, pooling=True, drop_out=True):
        if self.train and self.reduce_fn is not None and hasattr(self.reduce_fn, "begin"):
            # SyncBN: bn1's statistics exchange has independent work to hide under -- the shortcut 1x1 conv
            grp = []
            r1 = self._conv(f"{name}.conv2", [xin], 3, 1, 1, bn=f"{name}.bn1", defer_bn=grp)
            pend = self._bn_group_begin(grp)
            short = self._conv(f"{name}.conv1", [xin], 1, 1, 0, lrelu=True)
            self._bn_group_end(pend)
            # backward runs conv2 before conv1 whatever the forward order: conv1's input gradient is the last one into xin
            if xin.first_consumer == f"{name}.conv2":
                xin.first_consumer = f"{name}.conv1"
        else:
            short = self._conv(f"{name}.conv1", [xin], 1, 1, 0, lrelu=True)
            r1 = self._conv(f"{name}.conv2", [xin], 3, 1, 1, bn=f"{name}.bn1")
        r2 = self._conv(f"{name}.conv3", [r1], 3, 2, 2, bn=f"{name}.bn2")
        r3 = self._conv(f"{name}.conv4", [r2], 2, 2, 1, bn=f"{name}.bn3")
        a5 = self._conv(f"{name}.conv5", [r1, r2, r3], 1, 1, 0, bn=f"{name}.bn4")
        res_a = Act(ops.affine_add(short.t, a5.t, a5.scale, a5.shift))
        mask = self._mask(f"{name}.dropout") if drop_out else None
        if pooling or mask is not None:
            res_b = Act(ops.maskpool(res_a.t, mask, pooling))
        else:
            res_b = res_a
        self.tape[f"{name}.tail"] = (short, a5, res_a, res_b, mask, pooling)
        return res_b, res_a

    def _up_block(self, name, xin, skip, drop_out=True, bn_group=None):
        """bn_group: already-queued BatchNorm finalisations (``_conv(defer_bn=...)``) of layers that
        do not depend on this block; they share the statistics exchange of this block's bn1."""
        m1 = self._mask(f"{name}.dropout1") if drop_out else None
        m2 = self._mask(f"{name}.dropout2") if drop_out else None
        # cat(PixelShuffle(x), skip): without a Dropout2d mask on the concatenation (upBlock4 in training, every block
        # in eval mode) the skip tensor is NOT copied -- conv1 (and its weight gradient) read it as a second source
        # and the input-gradient conv writes its gradient straight into the skip's (the concat as an index
        # remap on load)
        direct = m2 is None and skip.scale is None and skip.t.dtype == xin.t.dtype
        if xin.first_consumer is None:
            xin.first_consumer = "glue"      # PixelShuffle reads BN(x): its gradient arrives through a glue kernel
        up_b = Act(ops.pixshuf_cat(xin.t, xin.scale, xin.shift, xin.mask, m1, m2, None if direct else skip.t))
        c1_srcs = [up_b, skip] if direct else [up_b]
        if bn_group is not None:
            e1 = self._conv(f"{name}.conv1", c1_srcs, 3, 1, 1, bn=f"{name}.bn1", defer_bn=bn_group)
            self._bn_forward_group(bn_group)
        else:
            e1 = self._conv(f"{name}.conv1", c1_srcs, 3, 1, 1, bn=f"{name}.bn1")
        e2 = self._conv(f"{name}.conv2", [e1], 3, 2, 2, bn=f"{name}.bn2")
        e3 = self._conv(f"{name}.conv3", [e2], 2, 2, 1, bn=f"{name}.bn3")
        a4 = self._conv(f"{name}.conv4", [e1, e2, e3], 1, 1, 0, bn=f"{name}.bn4")
        a4.mask = self._mask(f"{name}.dropout3") if drop_out else None
        self.tape[f"{name}.head"] = (xin, skip, up_b, m1, m2, direct)
        return a4

    # ------------------------------------------------------------------ projector.proj.0 without the 704-channel concat
    def _split_projector(self):
        """ProjectionV1's first layer is a 1x1 conv over cat(resample(skip_i)) (salsanext_proto.py:466-483,
        projector.py:18).  Bilinear resampling is linear, so conv1x1(resample(x)) == resample(conv1x1(x)) per skip: the
        skips that get UPSAMPLED (256 channels at 1/4 and 1/16 of the embedding's pixels) are multiplied with their
        slice of the weight at their own resolution -- 60 % fewer MFMAs in forward, input gradient and weight gradient
        of the step's largest layer -- an identity-resampled skip is read in place, and the 704-channel concatenation
        never exists.  The arithmetic is reassociated (sum over channels before instead of after the interpolation:
        rounding-level differences, inside every golden's 1e-4).  bf16 activation storage: the low-resolution shares stay
        fp32 until their interpolated sum is stored."""
        return (len(self.skips) == 4
                and all(s.t.dtype == self.skips[0].t.dtype and s.scale is None for s in self.skips))

    def _proj0_forward(self, hh, wh, defer_bn):
        name, bn = "projector.proj.0", "projector.proj.1"
        w = self.P[f"{name}.weight"]
        cout = w.shape[0]
        b = self.skips[0].t.shape[0]
        hi, lo, off = [], [], 0          # (Act at the embedding resolution | skip kept at its own, channel offset, channels)
        for sk in self.skips:
            c = sk.t.shape[3]
            hs, ws = sk.t.shape[1], sk.t.shape[2]
            if hs * ws < hh * wh:
                lo.append((sk, off, c))
            elif (hs, ws) == (hh, wh):
                hi.append((sk, off, c, None))                       # identity resample: the skip itself is the source
            else:
                r = Act(ops.bilinear(sk.t, hh, wh))
                hi.append((r, off, c, sk))                          # downsampled copy (gradient goes back through it)
            off += c
        assert len(lo) in (1, 2) and hi
        taps = [(0, 0)]
        # low-resolution shares, then their interpolated sum as the initial value of z
        ts = []
        store = self.skips[0].t.dtype
        for sk, o, c in lo:
            t = torch.empty(b, sk.t.shape[1], sk.t.shape[2], cout, device=sk.t.device, dtype=torch.float32)
            ops.conv_forward([sk.src()], self.packs.get(w, 0, c_off=o, c_cnt=c), None, cout, taps, out=t)
            ts.append(t)
        if len(ts) == 2:
            z = ops.bilinear_sum2(ts[0], ts[1], hh, wh, out_dtype=store)
        else:
            z = ops.bilinear(ts[0], hh, wh, out_dtype=store)
        # the high-resolution sources are contiguous in the weight's input channels: ONE multi-source launch, which
        # accumulates into z and takes the BatchNorm statistics of the final values in its epilogue
        h_off, h_cnt = hi[0][1], sum(h[2] for h in hi)
        assert all(h[1] == h_off + sum(g[2] for g in hi[:i]) for i, h in enumerate(hi))
        need_stats = self.train
        _, partial = ops.conv_forward([h[0].src() for h in hi], self.packs.get(w, 0, c_off=h_off, c_cnt=h_cnt),
                                      self.P.get(f"{name}.bias"), cout, taps, stats=need_stats, out=z, accumulate=True)
        rec = _ConvRec()
        rec.name, rec.srcs, rec.src_lrelu, rec.taps, rec.cout = name, [], False, taps, cout
        rec.slope, rec.weight, rec.dweight, rec.mode = 0.0, None, None, 1
        rec.bn = None
        rec.out = Act(z)
        defer_bn.append((rec, bn, partial, b * hh * wh, BN_MOMENTUM))
        self.tape[name] = rec
        self.tape["proj0.split"] = (hi, lo)
        return rec.out

    def _proj0_backward(self, dy, k):
        """Backward of ``_proj0_forward``: BatchNorm backward at the embedding resolution, then per source the weight
        gradient and the input gradient at the SOURCE's resolution (the output gradient goes down through the adjoint
        of the interpolation once per low-resolution skip).  Initialises the gradients of all four skips."""
        name = "projector.proj.0"
        rec = self.tape[name]
        hi, lo = self.tape["proj0.split"]
        G = self.grads
        w = self.P[f"{name}.weight"]
        cout = rec.cout
        # the BatchNorm (-> LeakyReLU) backward of this layer on load of its first weight-gradient launch, like the other
        # layers (FUSE_BN_APPLY; 704 channels at half resolution: the one large apply pass that was left)
        first = hi[0][0]
        fuse = (FUSE_BN_APPLY and self.side is None and dy.dtype == torch.float32 and tuple(dy.shape) == tuple(rec.out.t.shape)
                and dy.is_contiguous() and ops.wgrad_fusable(first, rec.out.t, cout))
        if fuse:
            kk = self._bn_backward(rec.bn, dy, rec.out.t, cout, 1, 0.0, k, coeffs_only=True)
            dz, pz = torch.empty_like(rec.out.t), None
            db = G.get(f"{name}.bias")
            ops.conv_wgrad(first.src(), dz, G[f"{name}.weight"], rec.taps, cin_off=hi[0][1], dbias=db,
                           fuse=(dy, rec.out.t, kk, (rec.bn.scale, rec.bn.shift)))
        else:
            dz, pz = self._bn_backward(rec.bn, dy, rec.out.t, cout, 1, 0.0, k)
        if self.capture is not None:
            # test hook: the layer as the reference sees it -- ONE 1x1 conv over the resampled, concatenated skips
            hh, wh = dz.shape[1], dz.shape[2]
            ordered = sorted([(o, src) for src, o, c, _ in hi] + [(o, Act(ops.bilinear(sk.t, hh, wh))) for sk, o, c in lo],
                             key=lambda t: t[0])
            rec.srcs = [a for _, a in ordered]
            self.capture[name] = (rec, dy.clone(), dz.clone())
        dzs = []
        for sk, o, c in lo:
            d = torch.empty(sk.t.shape[0], sk.t.shape[1], sk.t.shape[2], cout, device=dz.device, dtype=dz.dtype)
            dzs.append(ops.bilinear_bwd(d, dz))
        dw = G[f"{name}.weight"]
        with self._fork(dz, pz, *dzs):
            db = None if fuse else G.get(f"{name}.bias")
            for src, o, c, _ in (hi[1:] if fuse else hi):
                ops.conv_wgrad(src.src(), dz, dw, rec.taps, cin_off=o, bias_partial=pz if db is not None else None, dbias=db)
                db = None
            for (sk, o, c), d in zip(lo, dzs):
                ops.conv_wgrad(sk.src(), d, dw, rec.taps, cin_off=o)
        kp = (cout + 15) // 16 * 16
        # (each skip's gradient is initialised here, or -- embedding branch behind the up blocks, backward(d_feat_ready=...) --
        #  added to what the decoder has written: one fp32 addition of the same two numbers either way)
        for src, o, c, orig in hi:
            wd = self.packs.get(w, 1, c_off=o, c_cnt=c, kpad=kp)
            if orig is None and src.grad is not None:               # the skip itself, already holding the decoder's share
                ops.conv_forward([ops.Source(dz)], wd, None, c, rec.taps, out=src.grad, accumulate=True, grad=True)
                continue
            g = torch.empty_like(src.t)
            ops.conv_forward([ops.Source(dz)], wd, None, c, rec.taps, out=g, grad=True)
            if orig is None:
                src.grad = g                                        # the skip itself
            else:
                acc = orig.grad is not None
                if not acc:
                    orig.grad = torch.empty_like(orig.t)
                ops.bilinear_bwd(orig.grad, g, accumulate=acc)
        for (sk, o, c), d in zip(lo, dzs):
            wd = self.packs.get(w, 1, c_off=o, c_cnt=c, kpad=kp)
            acc = sk.grad is not None
            if not acc:
                sk.grad = torch.empty_like(sk.t)
            ops.conv_forward([ops.Source(d)], wd, None, c, rec.taps, out=sk.grad, accumulate=acc, grad=True)
        rec.out.grad = None

    # ------------------------------------------------------------------ forward
    def forward(self, x, train=True, dropout_masks=None, return_feat=True, update_running=True, lazy_feat=False,
                encoder_only=False):
        """x [B,Cin,H,W] fp32 (NCHW, as the reference feeds it).  Returns dict with NHWC tensors:
        prob [B,Ho,Wo,C], logits [B,H,W,32], feat [B,Ho,Wo,256] (if return_feat).
        encoder_only: stop behind resBlock5 and return {"enc": [B,H/16,W/16,256]} -- the reference's ``classification=True``
        mode (salsanext_proto.py:445-447); ``backward_encoder`` is its backward pass."""
        self.train, self.masks, self.update_running = train, dropout_masks, update_running
        self.packs.refresh()             # every weight repack of this step in one launch
        self.tape = OrderedDict()
        self.bn_seen = []
        ho, wo = x.shape[2], x.shape[3]
        if self.dataset == "SemanticPOSS":
            x = torch.nn.functional.pad(x, (0, 8, 0, 8))
        x = x.contiguous()
        assert x.shape[2] % 16 == 0 and x.shape[3] % 16 == 0, "H and W must be multiples of 16"
        self.x = x
        d = self._ctx_block("downCntx", x, first=True)
        d = self._ctx_block("downCntx2", d)
        d = self._ctx_block("downCntx3", d)
        d0c, d0b = self._res_block("resBlock1", d, True, False)
        d1c, d1b = self._res_block("resBlock2", d0c)
        d2c, d2b = self._res_block("resBlock3", d1c)
        d3c, d3b = self._res_block("resBlock4", d2c)
        d5c, _ = self._res_block("resBlock5", d3c, pooling=False)
        if encoder_only:
            self.enc_out = d5c
            if train and update_running:
                torch._foreach_add_([self.P[f"{n}.num_batches_tracked"] for n in self.bn_seen], 1)
            enc = d5c.t if d5c.scale is None else None
            assert enc is not None
            return {"enc": enc}
        self.skips = (d0b, d1b, d2b, d3b)
        self.out_hw = (ho, wo)
        self.return_feat = return_feat
        group = None
        if return_feat:
            # embedding branch, first half (salsanext_proto.py:466-483 + projector.proj.0): it only needs
            # the encoder skips, so its 704-wide GEMM is issued here and its BatchNorm statistics
            # share ONE exchange with upBlock1.bn1 (SyncBN: one all-reduce less per step)
            hh, wh = ho // 2, wo // 2
            group = []
            if self._split_projector():
                z0 = self._proj0_forward(hh, wh, group)
                feat_a = None
            else:
                b = x.shape[0]
                feat = torch.empty(b, hh, wh, sum(s.t.shape[3] for s in self.skips), device=x.device,
                                   dtype=self.skips[0].t.dtype)
                off = 0
                for s in self.skips:
                    ops.bilinear(s.t, hh, wh, dst=feat, dcoff=off, c=s.t.shape[3])
                    off += s.t.shape[3]
                feat_a = Act(feat)
                z0 = self._conv("projector.proj.0", [feat_a], 1, 1, 0, lrelu=False, bn="projector.proj.1", defer_bn=group)
        u4 = self._up_block("upBlock1", d5c, d3b, bn_group=group)
        u3 = self._up_block("upBlock2", u4, d2b)
        u2 = self._up_block("upBlock3", u3, d1b)
        u1 = self._up_block("upBlock4", u2, d0b, drop_out=False)
        logits = self._conv("cls_head", [u1], 1, 1, 0, lrelu=False, cout_pad=32)
        prob = ops.softmax(logits.t, self.ncls, ho, wo)
        self._prob = prob
        out = {"prob": prob, "logits": logits.t}
        if return_feat:
            emb = self._conv("projector.proj.3", [z0], 1, 1, 0, lrelu=False, src_lrelu=True)
            embn, norm = ops.l2norm(emb.t, 1e-12)
            self.tape["embed"] = (feat_a, z0, emb, embn, norm)
            self.lazy_feat = lazy_feat
            if lazy_feat:
                # the caller interpolates the rows it reads (contrast.LowResFeat); backward() then receives the gradient
                # of THIS tensor
                out["feat_low"] = embn
            else:
                out["feat"] = ops.bilinear(embn, ho, wo, out_dtype=torch.float32)   # the embedding leaves the backbone in fp32
        if train and update_running:
            torch._foreach_add_([self.P[f"{n}.num_batches_tracked"] for n in self.bn_seen], 1)
        return out

    # ------------------------------------------------------------------ backward helpers
    def _accum(self, act, g):
        """act.grad += g  (g is a full tensor we own)."""
        if act.grad is None:
            act.grad = g
        else:
            ops.axpy(g, act.grad)

    def _bn_backward(self, bn, dy, a, c, mode, slope=0.0, k=None, part=None, gmax=None, coeffs_only=False):
        """dy: gradient w.r.t. BN(a) [mode 0] or LeakyReLU(BN(a)) [mode 1] -> (dz = d/da, partial with
        sum(dz)); writes the BatchNorm parameter gradients.  SyncBN: the fp64 sums are all-reduced
        (``k``: coefficients already computed by ``_bn_backward_group``)."""
        G = self.grads
        pre_s, pre_h = (bn.scale, bn.shift) if mode == 1 else (None, None)
        if k is None:
            if part is None:      # (else: taken in the epilogue of the last input-gradient conv, _conv_backward)
                part = ops.bn_bwd_reduce(dy, a, c, mode, pre_s, pre_h, slope=slope)
            if self.reduce_fn is None:
                k = ops.bn_bwd_coeffs_partials(part, bn.count, bn.mean, bn.invstd, self.P[f"{bn.name}.weight"],
                                               G[f"{bn.name}.weight"], G[f"{bn.name}.bias"])
            elif getattr(self.reduce_fn, "bn_backward", None) is not None and self.reduce_fn.fits_channels(c) and part.dtype == torch.float32:
                k = self.reduce_fn.bn_backward(part, bn.count, bn.mean, bn.invstd, self.P[f"{bn.name}.weight"],
                                               G[f"{bn.name}.weight"], G[f"{bn.name}.bias"])
            else:
                sums, local = ops.stat_reduce(part, c, copy=True)      # dgamma/dbeta stay rank-local (averaged with the other grads)
                self.reduce_fn(sums)
                k = ops.bn_bwd_coeffs(sums, bn.count, bn.mean, bn.invstd, self.P[f"{bn.name}.weight"],
                                      G[f"{bn.name}.weight"], G[f"{bn.name}.bias"], local)
        if coeffs_only:
            return k
        return ops.bn_bwd_apply(dy, a, c, mode, k, pre_s, pre_h, slope=slope, gmax=gmax)

    def _bn_backward_group(self, items):
        """SyncBN backward of layers whose output gradients are all available: the (sum dy, sum dy*a)
        vectors of every layer travel in ONE all-reduce.  items: list of (conv name, dy).
        Returns {conv name: coefficients} for ``_conv_backward(..., k=...)``; empty when there is
        nothing to exchange (single rank: each layer folds its own partials in one launch)."""
        if self.reduce_fn is None:
            return {}
        G = self.grads
        recs = [self.tape[n] for n, _ in items]
        assert all(r.mode in (0, 1) for r in recs)
        buf = torch.empty(sum(r.cout for r in recs), 2, device=items[0][1].device, dtype=torch.float64)
        off = 0
        for r, (_, dy) in zip(recs, items):
            pre_s, pre_h = (r.bn.scale, r.bn.shift) if r.mode == 1 else (None, None)
            part, r.out.bwd_partial = r.out.bwd_partial, None
            if part is None:
                part = ops.bn_bwd_reduce(dy, r.out.t, r.cout, r.mode, pre_s, pre_h, slope=r.slope)
            ops.stat_reduce(part, r.cout, sums=buf[off:off + r.cout])
            off += r.cout
        local = buf.clone()
        self.reduce_fn(buf)
        ks, off = {}, 0
        for r in recs:
            bn = r.bn
            ks[r.name] = ops.bn_bwd_coeffs(buf[off:off + r.cout], bn.count, bn.mean, bn.invstd,
                                           self.P[f"{bn.name}.weight"], G[f"{bn.name}.weight"],
                                           G[f"{bn.name}.bias"], local[off:off + r.cout])
            off += r.cout
        return ks

    def _fork(self, *tensors):
        """Context under which launches go to the side stream, ordered after everything issued so
        far on the current stream.  ``tensors`` were allocated on the current stream and are read
        by the side stream: the caching allocator must not hand their memory out again before
        the side stream is done with it."""
        if self.side is None:
            return contextlib.nullcontext()
        self.side.wait_stream(torch.cuda.current_stream())
        for t in tensors:
            if t is not None:
                t.record_stream(self.side)
        return torch.cuda.stream(self.side)

    def _join(self):
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)

    def _zeros(self, c, device):
        z = getattr(self, "_zero_cache", None)
        if z is None or z.numel() < c or z.device != device:
            z = self._zero_cache = torch.zeros(max(c, 1024), device=device, dtype=torch.float32)
        return z[:c]

    def _conv_backward(self, name, dy, k=None):
        """dy: gradient w.r.t. the layer's consumer-visible output (BN output if it has BN).
        k: BatchNorm-backward coefficients from ``_bn_backward_group`` (else computed here).

        Two chains leave dz: the input gradients (dgrad -> the next layer's BatchNorm backward,
        the critical path, stays on the current stream) and the weight gradients, which nothing in
        the rest of backward reads -- they go to the side stream, so that the matrix-bound wgrad
        kernels run under the HBM-bound BatchNorm-backward / glue kernels of the following layers
        (and, data parallel, under the SyncBN exchanges the main stream waits for)."""
        rec = self.tape[name]
        a = rec.out.t
        c = rec.cout
        cpad = a.shape[3]
        G = self.grads
        # EXPERIMENT (C3D_F16X2_BWD=1): multi-tap input gradients over large populations on two fp16 planes -- dz is then read
        # through a per-tensor exponent; the apply pass folds max |dz| into one word on the way (no reduction launch)
        gmax = None
        if (ops.F16X2_BWD and ops.MFMA_MODE == 2 and len(rec.taps) > 1 and rec.mode in (0, 1, 2) and a.dtype == torch.float32
                and a.shape[0] * a.shape[1] * a.shape[2] >= ops.SIX_FWD_MIN_PIXELS):
            pool = getattr(self, "_gmax_pool", None)
            if pool is None or self._gmax_next >= pool.numel():
                pool = self._gmax_pool = torch.zeros(256, device=a.device, dtype=torch.int32)
                self._gmax_next = 0
            gmax = pool[self._gmax_next:self._gmax_next + 1]
            self._gmax_next += 1
        # conv -> LeakyReLU [-> BatchNorm] layers on one stream: the first weight-gradient launch applies the BatchNorm /
        # LeakyReLU backward while it stages dy and writes dz for the input-gradient convs below (no apply pass).  With a
        # side stream (data parallel) the weight gradients must not sit on the critical path: separate pass as before.
        fuse = (FUSE_BN_APPLY and rec.mode in (0, 2) and self.side is None and gmax is None and dy.dtype == torch.float32
                and tuple(dy.shape) == tuple(a.shape) and dy.is_contiguous() and ops.wgrad_fusable(rec.srcs[0], a, c))
        fuse_args = None
        if fuse:
            kk = None
            if rec.mode == 0:
                kk = self._bn_backward(rec.bn, dy, a, c, 0, rec.slope, k, rec.out.bwd_partial, coeffs_only=True)
                rec.out.bwd_partial = None
            dz, pz = torch.empty_like(a), None
            fuse_args = (dy, a, kk)
        elif rec.mode == 0 or rec.mode == 1:
            dz, pz = self._bn_backward(rec.bn, dy, a, c, rec.mode, rec.slope, k, rec.out.bwd_partial, gmax=gmax)
            rec.out.bwd_partial = None
        elif rec.mode == 2:
            dz, pz = ops.bn_bwd_apply(dy, a, c, 2, slope=rec.slope, gmax=gmax)
        else:
            dz, pz = ops.bn_bwd_apply(dy, dy, cpad, 3, dz=dy)
        if self.capture is not None and not fuse:
            self.capture[name] = (rec, None if dy is dz else dy.clone(), dz.clone())
        w = self.P[f"{name}.weight"] if rec.weight is None else rec.weight
        dw = G[f"{name}.weight"] if rec.dweight is None else rec.dweight
        gscale = ginv = None
        if gmax is not None:
            gscale, ginv = ops.grad_exponent_max(gmax, dz.shape[3])
        wg16 = (gscale, ginv) if (gmax is not None and all(s.t.dtype == torch.float32 for s in rec.srcs)) else None
        with self._fork(dz, pz, *([gscale, ginv] if gmax is not None else [])):
            db = G.get(f"{name}.bias")       # folded by the first weight-gradient launch of the layer
            off = 0
            for s in rec.srcs:
                ops.conv_wgrad(s.src(rec.src_lrelu), dz, dw, rec.taps, cin_off=off, slope=rec.slope,
                               bias_partial=pz if (db is not None and fuse_args is None) else None, dbias=db, f16x2=wg16,
                               fuse=fuse_args)
                db = None
                fuse_args = None          # the first launch wrote dz: the other sources of a concatenated input read it
                off += s.t.shape[3]
        if self.capture is not None and fuse:
            self.capture[name] = (rec, dy.clone(), dz.clone())
        ntaps = ops.negate_taps(rec.taps)
        off = 0
        gsrc = ops.Source(dz)
        if gmax is not None:
            gsrc = ops.Source(dz, gscale, self._zeros(dz.shape[3], dz.device))
        for s in rec.srcs:
            cs = s.t.shape[3]
            if not getattr(s, "no_grad", False):
                kp = (dz.shape[3] + 15) // 16 * 16
                wd = (self.packs.get(w, 1, c_off=off, c_cnt=cs, kpad=kp) if rec.weight is None
                      else ops.pack_weights(w, 1, c_off=off, c_cnt=cs, kpad=kp))
                if s.grad is None:
                    s.grad = torch.empty_like(s.t)
                    acc = False
                else:
                    acc = True
                # This layer is the FIRST reader of s in forward order, i.e. the LAST one to add to its gradient: the
                # launch's epilogue sees the final dy(s).  If s is the output of a conv -> LeakyReLU -> BatchNorm layer, take
                # that BatchNorm's backward sums (sum dy, sum dy * a) there and spare its c3d_bn_bwd_reduce pass (two
                # tensor reads per layer; 31 of the 43 BatchNorm layers end this way).  bf16x3 engine, fp32 tensors.
                part = None
                # (a residual sum x + BN(a) passes its gradient on unchanged: the same sums, multiplied with a, for that layer)
                tgt = s if s.bn_alias is None else s.bn_alias
                p_ = tgt.producer() if tgt.producer is not None else None
                # (round 5: the bf16 engine too, over bf16 tensors -- its BatchNorm-backward reduce pass was the second
                #  largest kernel of BASELINE configs[2]; kernels without the epilogue answer part = None)
                if (FUSE_BN_REDUCE and (ops.MFMA_MODE == 2 or (ops.MFMA_MODE == 1 and FUSE_BN_REDUCE_BF16)) and self.train
                        and p_ is not None and p_.mode == 0 and p_.bn is not None
                        and s.first_consumer == name and tgt.t.dtype == s.grad.dtype and tgt.t.shape[3] == cs
                        and tgt.t.dtype == (torch.float32 if ops.MFMA_MODE == 2 else torch.bfloat16)
                        and tuple(tgt.t.shape) == tuple(s.t.shape)):
                    part = torch.empty(cs, 2, ops.num_mtiles(*s.t.shape[:3]), device=s.t.device, dtype=torch.float32)
                _, part = ops.conv_forward([gsrc], wd, None, cs, ntaps, out=s.grad, accumulate=acc, grad=True,
                                           stat_partial=part, stat_mul=tgt.t if part is not None else None, f16x2_inv=ginv,
                                           stat_mul_optional=True)
                tgt.bwd_partial = part
            off += cs
        rec.out.grad = None

    def _ctx_backward(self, name, first=False):
        s, a2, out = self.tape[f"{name}.out"]
        g = out.grad
        out.grad = None
        self._conv_backward(f"{name}.conv3", g)          # dy(a2) = g
        a1 = self.tape[f"{name}.conv3"].srcs[0]
        # s receives g (residual) + dgrad of conv2; reuse g as its gradient buffer
        s.grad = g
        self._conv_backward(f"{name}.conv2", a1.grad)
        a1.grad = None
        if first:
            _, x, s_act = self.tape[f"{name}.conv1"]
            dz, pz = ops.bn_bwd_apply(s.grad, s.t, 32, 2)
            with self._fork(dz, pz):
                ops.bias_from_partials(pz, self.grads[f"{name}.conv1.bias"])
                ops.conv_in5_wgrad(x, dz, self.grads[f"{name}.conv1.weight"])
        else:
            self._conv_backward(f"{name}.conv1", s.grad)
        s.grad = None

    def _res_backward(self, name):
        short, a5, res_a, res_b, mask, pooling = self.tape[f"{name}.tail"]
        if res_b is res_a:
            d = res_a.grad
        else:
            d = ops.maskpool_bwd(res_b.grad, mask, res_a.grad, tuple(res_a.t.shape), pooling)
            res_b.grad = None
        res_a.grad = None
        self._conv_backward(f"{name}.conv5", d)
        r1, r2, r3 = self.tape[f"{name}.conv5"].srcs
        self._conv_backward(f"{name}.conv4", r3.grad)
        r3.grad = None
        self._conv_backward(f"{name}.conv3", r2.grad)
        r2.grad = None
        self._conv_backward(f"{name}.conv2", r1.grad)
        r1.grad = None
        self._conv_backward(f"{name}.conv1", d)

    def _up_backward(self, name, k4=None):
        xin, skip, up_b, m1, m2, direct = self.tape[f"{name}.head"]
        a4 = self.tape[f"{name}.conv4"].out
        self._conv_backward(f"{name}.conv4", a4.grad, k4)
        e1, e2, e3 = self.tape[f"{name}.conv4"].srcs
        self._conv_backward(f"{name}.conv3", e3.grad)
        e3.grad = None
        self._conv_backward(f"{name}.conv2", e2.grad)
        e2.grad = None
        self._conv_backward(f"{name}.conv1", e1.grad)
        e1.grad = None
        if direct:       # conv1's input-gradient launch already wrote / accumulated the skip's gradient
            dxa = ops.pixshuf_cat_bwd(up_b.grad, xin.mask, m1, None, tuple(xin.t.shape), 0, None, False)
        else:
            if skip.grad is None:
                skip.grad = torch.empty_like(skip.t)
                acc = False
            else:
                acc = True
            dxa = ops.pixshuf_cat_bwd(up_b.grad, xin.mask, m1, m2, tuple(xin.t.shape), skip.t.shape[3], skip.grad, acc)
        up_b.grad = None
        self._accum(xin, dxa)

    # ------------------------------------------------------------------ backward
    def backward(self, d_prob=None, d_feat=None, grads=None, d_feat_ready=None):
        """d_prob [B,Ho,Wo,C], d_feat [B,Ho,Wo,256] (NHWC, either may be None).  ``grads``: optional
        dict name -> preallocated gradient tensor (reference layout); returned filled.
        ``d_feat_ready`` (round 5; single process only): a callable that returns ``d_feat`` -- and makes the current stream wait
        for it -- as late as the pass can take it: the embedding branch then runs BEHIND the four up blocks instead of in front
        of them, so that whoever computes that gradient (the contrast loss on a second stream, coarse3d_amd.trainer.TrainStep)
        runs under the decoder's backward.  The skip gradients see their contributions in another order (u + p instead of
        p + u per element: the same bits)."""
        # the strips -> dw folds of the weight gradients are queued and run in batches (ops.WgradFolds): at the end of the
        # pass, or -- data parallel -- whenever a block's gradients are about to be sent
        prev, ops.WGRAD_FOLDS = ops.WGRAD_FOLDS, (ops.WgradFolds() if DEFER_WGRAD_FOLDS else None)
        try:
            return self._backward(d_prob, d_feat, grads, d_feat_ready)
        finally:
            ops.WGRAD_FOLDS = prev

    def backward_encoder(self, d_enc, grads):
        """Backward of ``forward(..., encoder_only=True)``: d_enc [B,H/16,W/16,256] NHWC, the gradient at resBlock5's output;
        ``grads``: name -> preallocated gradient tensor of the ENCODER parameters (downCntx*, resBlock*)."""
        prev, ops.WGRAD_FOLDS = ops.WGRAD_FOLDS, (ops.WgradFolds() if DEFER_WGRAD_FOLDS else None)
        try:
            self.grads = grads
            self.enc_out.grad = d_enc.contiguous()
            for name in ("resBlock5", "resBlock4", "resBlock3", "resBlock2", "resBlock1"):
                self._res_backward(name)
            self._ctx_backward("downCntx3")
            self._ctx_backward("downCntx2")
            self._ctx_backward("downCntx", first=True)
            if ops.WGRAD_FOLDS is not None:
                with self._fork():
                    ops.WGRAD_FOLDS.flush()
            self._join()
            self.tape = None
            return grads
        finally:
            ops.WGRAD_FOLDS = prev

    def _backward(self, d_prob, d_feat, grads, d_feat_ready=None):
        P = self.P
        late = d_feat_ready is not None
        if late and (d_feat is not None or self.reduce_fn is not None or self.on_block_done is not None or not self.return_feat):
            raise ValueError("backward(d_feat_ready=...): single process, an embedding branch in the forward, no d_feat next to it")
        if grads is None:
            grads = {k: torch.empty_like(v) for k, v in P.items()
                     if v.is_floating_point() and v.dim() > 0 and not k.endswith(("running_mean", "running_var"))
                     and k not in ("prototypes", "feat_norm.weight", "feat_norm.bias", "mask_norm.weight",
                                   "mask_norm.bias")}
            torch._foreach_zero_(list(grads.values()))      # a few multi-tensor launches instead of ~100 fills
        self.grads = grads
        d0b, d1b, d2b, d3b = self.skips
        def done(tag):
            # data parallel: the block's gradients are final once the side stream (weight
            # gradients) has caught up with the main stream (BatchNorm / bias gradients); the
            # bucket all-reduce is issued from the side stream so the main stream never waits
            if self.on_block_done is not None:
                with self._fork():
                    if ops.WGRAD_FOLDS is not None:
                        ops.WGRAD_FOLDS.flush()
                    self.on_block_done(tag)

        if d_prob is None:
            raise ValueError("backward needs d_prob (the segmentation losses always produce it)")
        embed = (d_feat is not None or late) and self.return_feat
        self.embed_ran = embed

        def embed_head(d_feat):
            feat_a, z0, emb, embn, norm = self.tape["embed"]
            if self.lazy_feat:
                d_embn = d_feat.contiguous()
                if d_embn.shape != embn.shape or d_embn.dtype != embn.dtype:
                    raise ValueError("backward: lazy_feat forward expects the gradient of out['feat_low']")
            else:
                d_embn = torch.empty_like(embn)
                d_feat = d_feat.contiguous()
                # the contrast loss marks the ~10^3 pixel rows of its dense gradient that are not zero (contrast.take_row_hint)
                ops.bilinear_bwd(d_embn, d_feat, rowmask=contrast.take_row_hint(d_feat))
            d_emb = ops.l2norm_bwd(embn, norm, d_embn, 1e-12)
            self._conv_backward("projector.proj.3", d_emb)

        def embed_rest(ks):
            """the rest of the embedding branch: gather-form transposes into the skip gradients (initialised here, or -- late
            -- added to what the up blocks have written)"""
            feat_a, z0 = self.tape["embed"][0], self.tape["embed"][1]
            if "proj0.split" in self.tape:
                self._proj0_backward(z0.grad, ks.get("projector.proj.0"))
                z0.grad = None
                return
            self._conv_backward("projector.proj.0", z0.grad, ks.get("projector.proj.0"))
            z0.grad = None
            off = 0
            for s in self.skips:
                acc = s.grad is not None
                if not acc:
                    s.grad = torch.empty_like(s.t)
                ops.bilinear_bwd(s.grad, feat_a.grad, dcoff=off, c=s.t.shape[3], accumulate=acc)
                off += s.t.shape[3]
            feat_a.grad = None

        # ---- both heads down to their first BatchNorm: projector.proj.3 and cls_head
        if embed and not late:
            embed_head(d_feat)
        logits = self.tape["cls_head"].out
        dl = ops.softmax_bwd(self._prob, d_prob.contiguous(), tuple(logits.t.shape))
        self._conv_backward("cls_head", dl)
        a4 = self.tape["upBlock4.conv4"].out
        # ---- the two BatchNorm backwards that are ready now share one statistics exchange
        ks = self._bn_backward_group(([("projector.proj.0", self.tape["embed"][1].grad)] if (embed and not late) else [])
                                     + [("upBlock4.conv4", a4.grad)])
        # ---- rest of the embedding branch: it initialises the skip gradients (gather-form transposes)
        if embed and not late:
            embed_rest(ks)
        elif not embed:
            for n in ("projector.proj.0", "projector.proj.1", "projector.proj.3"):
                for suffix in ("weight", "bias"):
                    grads[f"{n}.{suffix}"].zero_()
        done("projector")
        done("cls_head")
        for name in ("upBlock4", "upBlock3", "upBlock2", "upBlock1"):
            self._up_backward(name, ks.get(f"{name}.conv4"))
            done(name)
        if late:         # the embedding branch behind the decoder: its gradient has had the decoder's backward to arrive
            embed_head(d_feat_ready())
            embed_rest({})
        for name in ("resBlock5", "resBlock4", "resBlock3", "resBlock2", "resBlock1"):
            self._res_backward(name)
            done(name)
        self._ctx_backward("downCntx3")
        done("downCntx3")
        self._ctx_backward("downCntx2")
        done("downCntx2")
        self._ctx_backward("downCntx", first=True)
        done("downCntx")
        if ops.WGRAD_FOLDS is not None:
            with self._fork():
                ops.WGRAD_FOLDS.flush()
        self._join()
        self.tape = None
        return grads
