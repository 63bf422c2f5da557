"""SalsaNextProto with the reference module API (pc_processor/models/salsanext_proto.py:253-532)
on the MI355X-native HIP backbone.

Drop-in surface kept from the reference:
  * constructor signature and defaults (:254-267), ``forward`` signature and returned dict keys
    ``pred_2d`` / ``feat_2d`` / ``contrast_logits`` / ``contrast_target`` (:404-532);
  * ``state_dict`` key names and OIHW weight layout (released checkpoints load);
  * ``prototypes`` is re-bound to a fresh ``nn.Parameter`` on every update (:394, :516);
  * ``nn.BatchNorm2d`` children so ``SyncBatchNorm.convert_sync_batchnorm`` still walks them.
The child modules below only HOLD parameters (same construction order as the reference, hence
the same default initialisation under the same seed); the arithmetic runs in
``coarse3d_amd.backbone.Backbone`` and ``coarse3d_amd.proto``.

Deliberate fixes: the debug lines that overwrite the inputs (:414-421) are
not replicated (Q1); both H and W are validated (Q4).
"""
import torch
import torch.distributed as dist
import torch.nn as nn

from ... import contrast
from ... import graphed as _graphed
from ... import ops as ops_mod
from ... import proto as proto_ops
from ...backbone import Backbone
from .projector import ProjectionV1

DROP_P = 0.2


class _ParamBlock(nn.Module):
    """Parameter holder: children are registered from a (name, kernel, dilation, pad, cin, cout)
    table in the order that fixes both the state_dict layout and the RNG consumption of the
    default initialisation.  It has no forward: the arithmetic lives in coarse3d_amd.backbone."""

    def __init__(self, table):
        super().__init__()
        for name, k, dil, pad, cin, cout in table:
            if k == 0:
                self.add_module(name, nn.BatchNorm2d(cout))
            else:
                self.add_module(name, nn.Conv2d(cin, cout, (k, k), dilation=dil, padding=pad))

    def forward(self, *a, **k):
        raise RuntimeError("parameter container of the HIP backbone; call SalsaNextProto instead")


def ResContextBlock(cin, cout):
    """conv1x1 -> [3x3 -> BN] -> [3x3 d2 -> BN], residual (reference :38-65)."""
    return _ParamBlock([("conv1", 1, 1, 0, cin, cout), ("conv2", 3, 1, 1, cout, cout), ("bn1", 0, 0, 0, 0, cout),
                        ("conv3", 3, 2, 2, cout, cout), ("bn2", 0, 0, 0, 0, cout)])


def ResBlock(cin, cout, dropout_rate=DROP_P, pooling=True, drop_out=True):
    """shortcut 1x1; 3x3 -> 3x3 d2 -> 2x2 d2 chain, concat 1x1; optional Dropout2d + AvgPool (:68-148)."""
    blk = _ParamBlock([("conv1", 1, 1, 0, cin, cout), ("conv2", 3, 1, 1, cin, cout), ("bn1", 0, 0, 0, 0, cout),
                       ("conv3", 3, 2, 2, cout, cout), ("bn2", 0, 0, 0, 0, cout),
                       ("conv4", 2, 2, 1, cout, cout), ("bn3", 0, 0, 0, 0, cout),
                       ("conv5", 1, 1, 0, 3 * cout, cout), ("bn4", 0, 0, 0, 0, cout)])
    blk.pooling, blk.drop_out, blk.dropout_rate = pooling, drop_out, dropout_rate
    return blk


def UpBlock(cin, cout, dropout_rate=DROP_P, drop_out=True):
    """PixelShuffle(2) + skip concat, 3x3 -> 3x3 d2 -> 2x2 d2 chain, concat 1x1 (:151-212)."""
    blk = _ParamBlock([("conv1", 3, 1, 1, cin // 4 + 2 * cout, cout), ("bn1", 0, 0, 0, 0, cout),
                       ("conv2", 3, 2, 2, cout, cout), ("bn2", 0, 0, 0, 0, cout),
                       ("conv3", 2, 2, 1, cout, cout), ("bn3", 0, 0, 0, 0, cout),
                       ("conv4", 1, 1, 0, 3 * cout, cout), ("bn4", 0, 0, 0, 0, cout)])
    blk.drop_out, blk.dropout_rate = drop_out, dropout_rate
    return blk


# (site, channels as a multiple of base_channels) of the 13 active Dropout2d layers
_DROP_SITES = (("resBlock2.dropout", 4), ("resBlock3.dropout", 8), ("resBlock4.dropout", 8),
               ("resBlock5.dropout", 8),
               ("upBlock1.dropout1", 2), ("upBlock1.dropout2", 10), ("upBlock1.dropout3", 4),
               ("upBlock2.dropout1", 1), ("upBlock2.dropout2", 9), ("upBlock2.dropout3", 4),
               ("upBlock3.dropout1", 1), ("upBlock3.dropout2", 5), ("upBlock3.dropout3", 2))


class FC(nn.Module):
    """The classifier of the reference's ImageNet pre-training mode (salsanext_proto.py:216-231): global average pool +
    Linear(base_channels, 1000).  Same names (``fc.linear.*``) and construction as the reference's FC."""

    def __init__(self, base_channels):
        super().__init__()
        self.base_channels = base_channels
        self.pool = nn.AdaptiveAvgPool2d((1, 1))
        self.linear = nn.Linear(in_features=base_channels, out_features=1000, bias=True)

    def forward(self, x):
        return self.linear(self.pool(x).view(-1, self.base_channels))


class _EncoderFn(torch.autograd.Function):
    """x, encoder parameters -> resBlock5's output [B,256,H/16,W/16] (``classification=True``, salsanext_proto.py:445-447);
    backward = the explicit HIP backward plan of the encoder."""

    @staticmethod
    def forward(ctx, model, x, masks, names, *tensors):
        bb = model._make_backbone(model._tensor_dict())
        out = bb.forward(x.detach().float(), model.training, masks, False, encoder_only=True)
        ctx.bb, ctx.names, ctx.model = bb, names, model
        return out["enc"].permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, d_enc):
        model, bb = ctx.model, ctx.bb
        P = dict(model._cached()[0])
        grads = {n: torch.zeros_like(P[n]) for n in ctx.names}
        enc_dtype = bb.enc_out.t.dtype
        bb.backward_encoder(d_enc.permute(0, 2, 3, 1).contiguous().to(enc_dtype), grads)
        ctx.bb = None
        return (None, None, None, None) + tuple(grads[n] for n in ctx.names)


class _BackboneFn(torch.autograd.Function):
    """x, parameters -> (pred_2d, feat_2d); backward = the explicit HIP backward plan."""

    @staticmethod
    def forward(ctx, model, x, masks, return_feat, names, *tensors):
        # (training mode only: a captured eval forward -- serving.GraphedInference -- hands out the tensors of its graph)
        lazy = bool(return_feat) and model._lazy_feat()
        ctx.graphed = None
        if model.graph_backbone and model.training:
            # forward / backward of the backbone as two hipGraphs behind the module API (coarse3d_amd/graphed.py)
            gb = model._graphed_backbone()
            keep = model._last_keep if masks is not None else None
            injected = masks is not None and masks is model.dropout_masks
            if gb.why_not(model, x, masks, injected) is None:
                res = gb.forward(model, x, masks, keep, bool(return_feat), lazy, names, any(ctx.needs_input_grad))
                if res is not None:
                    import weakref
                    pred, feat, ent = res
                    ent.ctx_ref = weakref.ref(ctx)
                    ctx.graphed = (gb, ent, lazy)
                    ctx.names, ctx.model, ctx.return_feat, ctx.bb = names, model, return_feat, None
                    if feat is None:
                        feat = x.new_zeros(())
                        ctx.mark_non_differentiable(feat)
                    return pred, feat
        bb = model._make_backbone(model._tensor_dict())
        bb.on_block_done = model._block_done
        out = (bb.forward(x.detach().float(), model.training, masks, return_feat, lazy_feat=True) if lazy
               else bb.forward(x.detach().float(), model.training, masks, return_feat))
        ctx.bb, ctx.names, ctx.model = bb, names, model
        ctx.return_feat = return_feat
        pred = out["prob"].permute(0, 3, 1, 2)
        # lazy: the l2-normalised embedding BEFORE its last interpolation (half resolution); SalsaNextProto.forward
        # wraps it in a contrast.LowResFeat and ``feat_2d`` becomes an on-demand entry of the output
        feat = out["feat_low" if lazy else "feat"].permute(0, 3, 1, 2) if return_feat else x.new_zeros(())
        ctx.mark_non_differentiable(*([] if return_feat else [feat]))
        return pred, feat

    @staticmethod
    def backward(ctx, d_pred, d_feat):
        model = ctx.model
        if ctx.graphed is not None:
            gb, ent, lazy = ctx.graphed
            d_prob = d_pred.permute(0, 2, 3, 1) if d_pred is not None else None
            d_f = d_feat.permute(0, 2, 3, 1) if (ctx.return_feat and d_feat is not None) else None
            grads, live = gb.backward(ent, d_prob, d_f)
            model._grads_live = live
            if model._bind_grads and all(p_.grad is None for _, p_ in model._cached()[0]):
                # coarse3d_amd.trainer.TrainStep owns the optimiser loop: the persistent buffers become param.grad
                for n, p_ in model._cached()[0]:
                    if live or not n.startswith("projector."):
                        p_.grad = grads[n]
                return (None,) * (5 + len(ctx.names))
            # handed to autograd: copies (ONE flat copy, then views), so that whatever autograd / the caller does with
            # param.grad never aliases the buffers the next replay writes
            flat = model._own_flat[1].clone()
            out, off = [], 0
            for n, p_ in model._cached()[0]:
                k = p_.numel()
                out.append(flat[off:off + k].view_as(p_) if (live or not n.startswith("projector.")) else None)
                off += k
            byname = dict(zip((n for n, _ in model._cached()[0]), out))
            return (None, None, None, None, None) + tuple(byname[n] for n in ctx.names)
        bb = ctx.bb
        prob = bb._prob
        d_prob = (d_pred.permute(0, 2, 3, 1).contiguous() if d_pred is not None
                  else torch.zeros_like(prob))
        d_f = d_feat.permute(0, 2, 3, 1).contiguous() if (ctx.return_feat and d_feat is not None) else None
        bound = model._bound_grad_views(ctx.names)
        # coarse3d_amd.trainer.TrainStep computes the embedding's gradient itself, on a second stream (the contrast loss is
        # not in autograd's graph then): (event, NHWC gradient of the half-resolution embedding).  The backbone takes it
        # behind its decoder blocks (Backbone.backward(d_feat_ready=...)).
        late, model._late_dfeat = model._late_dfeat, None
        ready = None
        if late is not None and ctx.return_feat:
            # (autograd hands this Function a materialised ZERO gradient for the embedding nobody in its graph read: TrainStep
            #  kept the contrast loss out of the graph, the gradient is the one it computed)
            ev, d_low = late
            d_f = None

            def ready():
                torch.cuda.current_stream().wait_event(ev)
                return d_low
        gbuf = bound if bound is not None else model._grad_buffers(ctx.names)
        # (the other backbones' backward passes have no such argument: TrainStep only offers it to this class' own)
        grads = bb.backward(d_prob, d_f, grads=gbuf, d_feat_ready=ready) if ready is not None else bb.backward(d_prob, d_f, grads=gbuf)
        ctx.bb = None
        # Without the embedding branch in the graph (contrast warm-up epochs: return_feat=False, trainer.py:625-630; or a
        # loss that never read feat_2d) the reference's projector parameters get NO gradient (autograd leaves .grad at
        # None and AdamW skips them: no weight decay, no step count).  Same here: their buffers hold zeros, .grad stays None.
        live = getattr(bb, "embed_ran", True)
        model._grads_live = live
        if model._grad_ready is not None:
            model._grad_ready()
        if bound is not None:
            # single-process training step that owns its optimiser loop (coarse3d_amd.trainer.TrainStep sets
            # ``_bind_grads``): the gradients were written into one persistent flat buffer and become param.grad
            # directly -- no 192 allocations at the start of every backward (the stream idled ~0.4 ms behind
            # them) and nothing for AccumulateGrad to do
            for n, p_ in model._cached()[0]:
                if live or not n.startswith("projector."):
                    p_.grad = bound[n]
            return (None,) * (5 + len(ctx.names))
        if model._flat_grads is not None:
            # coarse3d_amd.dist.DataParallel: the gradients live in its flat buffer (being all-reduced
            # in place right now); finish_gradients() binds param.grad to those views.  Handing them
            # to autograd would make AccumulateGrad clone ~200 tensors out of a buffer in flight.
            return (None,) * (5 + len(ctx.names))
        return (None, None, None, None, None) + tuple(grads[n] if (live or not n.startswith("projector.")) else None
                                                      for n in ctx.names)


class SalsaNextProto(nn.Module):
    def __init__(self, in_channel=5, nclasses=20, sub_proto_size=20, ignore_label=0, use_prototype=False,
                 softmax=True, proj_dim=256, projection="v1", classification=False, proto_mom=0.999,
                 dataset="SemanticKitti"):
        super().__init__()
        # ``softmax`` is accepted and stored like the reference does (:261, :279) -- and, like the
        # reference, never read again: forward applies F.softmax unconditionally (:460)
        self.nclasses = nclasses
        self.base_channels = 32
        self.proj_dim = proj_dim
        self.softmax = softmax
        self.projection = projection
        self.classification = classification
        self.use_prototype = use_prototype
        self.sub_proto_size = sub_proto_size
        self.ignore_label = ignore_label
        self.proto_mom = proto_mom
        self.dataset = dataset
        bc = self.base_channels

        self.downCntx = ResContextBlock(in_channel, bc)
        self.downCntx2 = ResContextBlock(bc, bc)
        self.downCntx3 = ResContextBlock(bc, bc)
        self.resBlock1 = ResBlock(bc, 2 * bc, DROP_P, pooling=True, drop_out=False)
        self.resBlock2 = ResBlock(2 * bc, 4 * bc, DROP_P, pooling=True)
        self.resBlock3 = ResBlock(4 * bc, 8 * bc, DROP_P, pooling=True)
        self.resBlock4 = ResBlock(8 * bc, 8 * bc, DROP_P, pooling=True)
        self.resBlock5 = ResBlock(8 * bc, 8 * bc, DROP_P, pooling=False)
        if self.classification:            # ImageNet pre-training head (salsanext_proto.py:308-309), registered where the reference does
            self.fc = FC(8 * bc)
        self.upBlock1 = UpBlock(8 * bc, 4 * bc, DROP_P)
        self.upBlock2 = UpBlock(4 * bc, 4 * bc, DROP_P)
        self.upBlock3 = UpBlock(4 * bc, 2 * bc, DROP_P)
        self.upBlock4 = UpBlock(2 * bc, bc, DROP_P, drop_out=False)
        self.cls_head = nn.Conv2d(bc, nclasses, kernel_size=(1, 1))
        self.projector = ProjectionV1(bc * 22, proj_dim)
        self.prototypes = nn.Parameter(torch.randn(nclasses, sub_proto_size, proj_dim), requires_grad=False)
        nn.init.trunc_normal_(self.prototypes, std=0.02)
        self.feat_norm = nn.LayerNorm(proj_dim)
        self.mask_norm = nn.LayerNorm(nclasses)

        # hooks (not part of the reference surface)
        self.dropout_masks = None     # test hook: dict site -> [B, C] multipliers (0 or 1/(1-p))
        self.gumbel_noise = None      # test hook: Exp(1) variates [N, M] indexed by pixel
        self._bn_reduce = None        # data parallel: in-place all-reduce of fp64 BN sums
        self._world = 1
        self._proto_mean = None       # data parallel: mean of the bank over ranks
        self._proto_sums_reduce = None   # data parallel alternative: all-reduce of the per-class feature sums
        self._grad_ready = None       # data parallel: called when all gradients are written
        self._block_done = None       # data parallel: called per block in backward order
        self._flat_grads = None
        self._grads_live = True       # False after a backward pass without the embedding branch: projector.* have no gradient
        self._bind_grads = False      # TrainStep: write the gradients into one persistent buffer and bind param.grad
        self._static_bank = False     # TrainStep(graph=True): update the prototype bank in place instead of re-binding it
        self._labelled_hint = None    # TrainStep: (idx, count) of the labelled pixels for THIS forward (consumed by it)
        self._own_flat = None
        self._cache = None
        self._side = None             # second HIP stream (weight-gradient chain of the backward pass)
        self.graph_backbone = _graphed.ENABLED_BY_DEFAULT   # True: forward / backward of the backbone as two hipGraphs (graphed.py)
        self._gb = None
        self._last_keep = None
        self._packs = ops_mod.PackCache()   # batched weight repacking (one launch per step)

    # ------------------------------------------------------------------ plumbing
    def _bn_exchange(self):
        """(in-place fp64 all-reduce or None, number of ranks) for the BatchNorm statistics.

        Wired explicitly by coarse3d_amd.dist.DataParallel, or -- the reference trainer's own wrap,
        tasks/weak_segmentation/trainer.py:54-60 -- switched on by the module tree itself: after
        ``torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)`` the BatchNorm children ARE
        SyncBatchNorm modules, and like those (torch/nn/modules/batchnorm.py: need_sync = training
        and world_size > 1) the statistics are exchanged over their process group whenever a
        group is initialised.  Plain BatchNorm2d children keep rank-local statistics, exactly as
        under stock DistributedDataParallel without the conversion."""
        if self._bn_reduce is not None:
            return self._bn_reduce, self._world
        if not (self.training and dist.is_available() and dist.is_initialized()):
            return None, 1
        sync = [m for m in self.modules() if isinstance(m, nn.SyncBatchNorm)]
        if not sync:
            return None, 1
        n_bn = sum(isinstance(m, nn.modules.batchnorm._BatchNorm) for m in self.modules())
        if len(sync) != n_bn:
            raise RuntimeError(f"{len(sync)} of {n_bn} BatchNorm layers are SyncBatchNorm: convert the whole model "
                               "(torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)) or none of it")
        group = sync[0].process_group
        if any(m.process_group is not group for m in sync):
            raise RuntimeError("all SyncBatchNorm layers of the model must share one process group")
        world = dist.get_world_size(group)
        if world < 2:
            return None, 1

        def reduce_(t):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return t
        return reduce_, world

    def _bank_exchange(self):
        """Mean of the updated bank over ranks.  The reference does it whenever a process group is
        initialised (salsanext_proto.py:397-400), i.e. also under its own stock-DDP wrap."""
        if self._proto_mean is not None or self._proto_sums_reduce is not None:
            return self._proto_mean
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from ... import dist as c3d_dist
            return c3d_dist.world_mean
        return None

    def _side_stream_for(self, P):
        dev = next(iter(P.values())).device
        if dev.type != "cuda":
            return None
        if self._side is None or self._side.device != dev:
            self._side = torch.cuda.Stream(device=dev)
        return self._side

    def _make_backbone(self, P):
        reduce_fn, world = self._bn_exchange()
        return Backbone(P, self.nclasses, self.dataset, reduce_fn, world, self._packs, self._side_stream_for(P))

    def _check_input(self, h, w):
        hp, wp = (h + 8, w + 8) if self.dataset == "SemanticPOSS" else (h, w)
        assert hp % 16 == 0 and wp % 16 == 0, "input height and width must be multiples of 16"

    _SKIP = ("prototypes", "feat_norm.weight", "feat_norm.bias", "mask_norm.weight", "mask_norm.bias",
             "fc.linear.weight", "fc.linear.bias")
    # class-level defaults (subclasses with their own __init__ inherit them)
    _bind_grads = False
    _grads_live = True
    _static_bank = False
    _labelled_hint = None
    _own_flat = None
    _cache = None
    graph_backbone = False
    _gb = None
    _last_keep = None
    classification = False           # (class-level defaults: the other backbones' constructors do not run this class' __init__)
    _late_dfeat = None               # (event, gradient): see _BackboneFn.backward
    _late_dfeat_ok = True            # this class' backbone takes the embedding's gradient behind its decoder blocks

    def _graphed_backbone(self):
        if self._gb is None:
            self._gb = _graphed.GraphedBackbone(self)
        return self._gb

    def _graph_grad_views(self, names):
        """name -> view of the persistent flat gradient buffer (the one ``_bound_grad_views`` binds), unconditionally: the
        backward GRAPH writes there on every replay."""
        named = self._cached()[0]
        if self._own_flat is None or self._own_flat[0] != names:
            total = sum(p.numel() for _, p in named)
            flat = torch.zeros(total, device=named[0][1].device, dtype=torch.float32)
            views, off = {}, 0
            for n, p in named:
                views[n] = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
            self._own_flat = (names, flat, views)
        return self._own_flat[2]

    def _list_trainable(self):
        """(name, parameter) of everything the backward pass writes a gradient for (uncached; subclasses override)."""
        return [(k, p) for k, p in self.named_parameters() if k not in self._SKIP]

    def _cached(self):
        """(trainable (name, parameter) list, their names, backbone tensor dict), built once: walking the module tree
        (named_parameters / named_buffers, ~400 detach calls) costs ~1 ms of host time per step, during which the
        stream is idle at the step boundary.  The parameters keep their identity and storage across steps (the
        optimiser updates in place); whatever moves them goes through ``_apply`` (.to / .cuda / .float), which drops
        the cache; ``invalidate_caches()`` is there for anything else (a parameter replaced by hand)."""
        c = self._cache
        if c is None:
            named = self._list_trainable()
            d = {k: p.detach() for k, p in self.named_parameters() if k not in self._SKIP}
            d.update({k: v for k, v in self.named_buffers()})
            c = self._cache = (named, tuple(k for k, _ in named), d)
        return c

    def invalidate_caches(self):
        self._cache = None
        self._own_flat = None

    def _apply(self, fn, *a, **k):
        self.invalidate_caches()
        return super()._apply(fn, *a, **k)

    def _tensor_dict(self):
        return self._cached()[2]

    def _trainable(self):
        return list(self._cached()[0])

    def _bound_grad_views(self, names):
        """name -> gradient view of one persistent flat buffer, or None when the direct-binding path does not apply
        (not enabled, data parallel, or some param.grad already holds a value that has to be accumulated into)."""
        if not self._bind_grads or self._flat_grads is not None:
            return None
        named = self._cached()[0]
        if any(p.grad is not None for _, p in named):
            return None
        if self._own_flat is None or self._own_flat[0] != names:
            total = sum(p.numel() for _, p in named)
            flat = torch.zeros(total, device=named[0][1].device, dtype=torch.float32)
            views, off = {}, 0
            for n, p in named:
                views[n] = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
            self._own_flat = (names, flat, views)
        return self._own_flat[2]

    def _grad_buffers(self, names):
        if self._flat_grads is not None:
            return self._flat_grads
        P = dict(self._cached()[0])
        return {n: torch.empty_like(P[n]) for n in names}

    def _draw_masks(self, b, device):
        if self.dropout_masks is not None:
            return self.dropout_masks
        # one draw for all sites, laid out site after site ([b, c] blocks): the per-site masks are contiguous VIEWS
        # (thirteen slice copies per step cost ~0.6 ms of host time with the stream idle)
        total = sum(m for _, m in _DROP_SITES) * self.base_channels
        keep = (torch.rand(b * total, device=device) >= DROP_P).to(torch.float32) * (1.0 / (1.0 - DROP_P))
        self._last_keep = keep           # (the graphed backbone copies the draw into its static buffer)
        masks, off = {}, 0
        for name, mult in _DROP_SITES:
            c = mult * self.base_channels
            masks[name] = keep[off:off + b * c].view(b, c)
            off += b * c
        return masks

    def _lazy_feat(self):
        """Whether ``feat_2d`` is an on-demand entry of the output (contrast.LowResFeat): this class' backbone, in
        training mode, unless C3D_LAZY_FEAT=0."""
        return contrast.LAZY_FEAT_ON and self.training and type(self) is SalsaNextProto

    # ------------------------------------------------------------------ forward
    def forward(self, x, label=None, eval_mask=None, return_feat=True, proto_loss=False, proto_pl=None):
        b, c, h, w = x.shape
        self._check_input(h, w)
        masks = self._draw_masks(b, x.device) if self.training else None
        named, names, _ = self._cached()
        if self.classification:
            # salsanext_proto.py:445-447: the encoder, then the classifier -- the decoder, the heads and the bank are not run
            enc_named = [(n, p) for n, p in named if n.startswith(("downCntx", "resBlock"))]
            enc = _EncoderFn.apply(self, x, masks, tuple(n for n, _ in enc_named), *[p for _, p in enc_named])
            return self.fc(enc.float())
        pred, feat = _BackboneFn.apply(self, x, masks, bool(return_feat), names, *[p for _, p in named])
        out = proto_ops.LazyOutputs({"pred_2d": pred})
        labelled, self._labelled_hint = self._labelled_hint, None
        if not return_feat:
            return out
        if self._lazy_feat():
            # ``feat`` is the embedding before its last interpolation: ``feat_2d`` (1.07 GB at 8x64x2048) is computed
            # when somebody reads it; the consumers of the training step read rows through ``out.feat_rows``
            out.feat_rows = contrast.LowResFeat(feat, (h, w))
            out.lazy["feat_2d"] = out.feat_rows.dense
        else:
            out.feat_rows = None
            out["feat_2d"] = feat
        if self.use_prototype and label is not None and eval_mask is not None:
            with torch.no_grad():
                P = {"prototypes": self.prototypes.data, "feat_norm.weight": self.feat_norm.weight.data,
                     "feat_norm.bias": self.feat_norm.bias.data, "mask_norm.weight": self.mask_norm.weight.data,
                     "mask_norm.bias": self.mask_norm.bias.data}
                feat_nhwc = out.feat_rows if out.feat_rows is not None else feat.detach().permute(0, 2, 3, 1).contiguous()
                res = proto_ops.prototype_step(
                    feat_nhwc, P, label.reshape(-1).long() if proto_loss else None, proto_loss,
                    noise=self.gumbel_noise, momentum=self.proto_mom, ignore_label=self.ignore_label,
                    world_mean=self._bank_exchange(), ema_base=proto_pl, sums_reduce=self._proto_sums_reduce,
                    labelled=labelled)
                self.prototypes.data.copy_(res["bank_l2"])          # in-place renormalisation (:502)
                if proto_pl is not None:
                    self.prototypes = nn.Parameter(proto_pl.clone(), requires_grad=False)
                if proto_loss:
                    if self._static_bank:
                        # a captured training step (coarse3d_amd.trainer.TrainStep(graph=True)) replays kernels with
                        # baked-in addresses: the bank is updated IN PLACE (callers re-read the attribute either way)
                        self.prototypes.data.copy_(res["new_bank"])
                    else:
                        self.prototypes = nn.Parameter(res["new_bank"], requires_grad=False)
                    if callable(res["contrast_logits"]):
                        out.lazy["contrast_logits"] = res["contrast_logits"]
                    else:
                        out["contrast_logits"] = res["contrast_logits"]
                    out["contrast_target"] = res["contrast_target"]
        return out
