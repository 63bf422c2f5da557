"""The backbone's forward and backward as two hipGraphs behind the plain module API.

``SalsaNextProto`` is a drop-in for the reference module (pc_processor/models/salsanext_proto.py:253-532): the reference's
own trainer loop -- ``out = model(x, ...)``, its own loss modules, ``loss.backward()``, ``optimizer.step()``
(tasks/weak_segmentation/trainer.py:621-704) -- works unchanged.  Issued that way the backbone is ~270 + ~330 kernel launches
through ctypes per step, and on a box with a slow host the HOST bounds the step (measured: 82.8 img/s launch by launch
against 269.7 with the whole step captured by ``coarse3d_amd.trainer.TrainStep(graph=True)``, same GPU).  ``TrainStep`` is
this package's own loop; somebody who keeps the reference's loop does not use it.

``model.graph_backbone = True`` (or ``C3D_GRAPH_BACKBONE=1``) closes that gap at the module boundary, in the manner of
``torch.cuda.make_graphed_callables``: per (input shape, mode) the autograd Function behind ``forward`` captures
``Backbone.forward`` into one hipGraph and, right behind it, ``Backbone.backward`` into a second one (same memory pool: the
activations the backward reads are the forward graph's own tensors).  A call then costs three small copies + one graph
launch in forward and two copies + one launch in backward; the values are those of the launch-by-launch path, bit for bit
(tests/test_gpu_step.py).  Everything outside the backbone -- the prototype update, the losses, the optimiser -- stays what
the caller runs.

What it refuses (and runs launch by launch instead, silently correct): data-parallel wrappers (their exchanges fire from
Python hooks), injected dropout masks (test hook), a second forward while the previous one still waits for its backward
(gradient accumulation over several forwards: the second would overwrite the first one's activations), and everything
before ``warmup`` eager calls of a configuration.  The outputs handed to autograd are copies of the graph's static tensors
(one 84 MB + one 67 MB copy at 8 x 64 x 2048): a caller may keep ``pred_2d`` across steps as it could before."""
import os

import torch

from . import ops

ENABLED_BY_DEFAULT = os.environ.get("C3D_GRAPH_BACKBONE", "0") == "1"


class _Entry:
    __slots__ = ("eager", "g_fwd", "g_bwd", "bb", "sx", "skeep", "smasks", "out", "sdprob", "sdfeat", "grads", "packs_generation",
                 "sig", "pending", "embed", "ctx_ref")

    def __init__(self):
        self.eager, self.g_fwd, self.g_bwd, self.pending, self.ctx_ref = 0, None, None, False, None


class GraphedBackbone:
    def __init__(self, model, warmup=2, max_entries=4):
        if warmup < 2:
            raise ValueError("warmup >= 2: the first eager pass records the weight repacks, the second builds their table")
        self.model, self.warmup, self.max_entries = model, warmup, max_entries
        self.entries = {}
        self.pool = None
        self.replays = 0
        self.captures = 0
        self.fallbacks = 0

    # ------------------------------------------------------------------ eligibility
    def _signature(self):
        # what the captured kernels read by raw pointer: the backbone's parameters and buffers (NOT the prototype bank, which
        # the reference re-binds to a fresh Parameter on every update, nor the LayerNorms of the prototype path)
        return tuple(t.data_ptr() for t in self.model._tensor_dict().values())

    def why_not(self, model, x, masks, injected_masks):
        import coarse3d_amd
        if not x.is_cuda:
            return "CPU tensor"
        if type(model).__name__ != "SalsaNextProto" or getattr(model, "classification", False):
            return "only SalsaNextProto's segmentation forward is captured (the other backbones run launch by launch)"
        if not coarse3d_amd.GRAPH_REPLAY_SAFE and os.environ.get("C3D_GRAPH_UNSAFE") != "1":
            return "the HIP runtime was initialised without DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (coarse3d_amd/__init__.py)"
        if model._bn_exchange()[0] is not None or model._flat_grads is not None or model._block_done is not None or model._grad_ready is not None:
            return "data-parallel hooks"
        if injected_masks:
            return "injected dropout masks"
        if torch.cuda.is_current_stream_capturing():
            return "already inside a capture"
        return None

    # ------------------------------------------------------------------ forward
    def forward(self, model, x, masks, keep, return_feat, lazy, names, needs_grad):
        """Returns (prob_nchw_view, feat_nchw_view or None, entry) or None (= run launch by launch)."""
        key = (tuple(x.shape), x.dtype, bool(model.training), bool(return_feat), bool(lazy), ops.matrix_precision_state(),
               bool(needs_grad))
        ent = self.entries.get(key)
        if ent is None:
            if len(self.entries) >= self.max_entries:
                self.entries.pop(next(iter(self.entries)))
                if not any(e.g_fwd is not None for e in self.entries.values()):
                    self.pool = None
            ent = self.entries[key] = _Entry()
        packs = model._packs
        if ent.g_fwd is not None and (ent.packs_generation != packs.generation or ent.sig != self._signature()):
            ent.g_fwd = ent.g_bwd = None                     # weight-pack table or parameter storage moved: stale addresses
            ent.eager = min(ent.eager, self.warmup - 1)
            if not any(e.g_fwd is not None for e in self.entries.values()):
                self.pool = None
        if ent.g_fwd is None and ent.eager < self.warmup:
            ent.eager += 1
            return None
        if ent.pending and (ent.ctx_ref is None or ent.ctx_ref() is None):
            ent.pending = False                              # that forward's autograd node is gone: no backward will come
        if ent.pending:                                      # the previous forward of this entry still waits for its backward
            self.fallbacks += 1
            return None
        if ent.g_fwd is None:
            self._capture(ent, model, x, masks, keep, return_feat, lazy, names, needs_grad)
        ent.sx.copy_(x)
        if ent.skeep is not None:
            ent.skeep.copy_(keep)
        ent.g_fwd.replay()
        self.replays += 1
        ent.pending = bool(needs_grad)
        prob = ent.out["prob"].clone().permute(0, 3, 1, 2)
        feat = None
        if return_feat:
            feat = ent.out["feat_low" if lazy else "feat"].clone().permute(0, 3, 1, 2)
        return prob, feat, ent

    def _capture(self, ent, model, x, masks, keep, return_feat, lazy, names, needs_grad):
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        ent.sx = x.detach().float().clone()
        ent.skeep, ent.smasks = None, None
        if masks is not None:
            # the masks are views of ONE draw (SalsaNextProto._draw_masks): a static copy of the draw, the same views
            ent.skeep = keep.clone()
            ent.smasks, off = {}, 0
            for name, m in masks.items():
                n = m.numel()
                ent.smasks[name] = ent.skeep[off:off + n].view_as(m)
                off += n
            assert off == keep.numel()
        packs = model._packs
        gen = packs.generation
        torch.cuda.synchronize()
        bb = model._make_backbone(model._tensor_dict())
        g_fwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_fwd, pool=self.pool, capture_error_mode="thread_local"), torch.no_grad():
            out = bb.forward(ent.sx, model.training, ent.smasks, return_feat, lazy_feat=bool(lazy))
        if packs.generation != gen:
            raise RuntimeError("the weight-pack table was rebuilt during graph capture")
        ent.g_fwd, ent.bb, ent.out = g_fwd, bb, out
        ent.packs_generation, ent.sig = gen, self._signature()
        ent.g_bwd = None
        self.captures += 1
        if needs_grad:
            # the backward graph right behind it: it reads the forward graph's activations (same pool)
            ent.embed = bool(return_feat)
            ent.sdprob = torch.zeros_like(out["prob"])
            ent.sdfeat = torch.zeros_like(out["feat_low" if lazy else "feat"]) if return_feat else None
            ent.grads = model._graph_grad_views(names)
            g_bwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_bwd, pool=self.pool, capture_error_mode="thread_local"), torch.no_grad():
                bb.backward(ent.sdprob, ent.sdfeat, grads=ent.grads)
            ent.g_bwd = g_bwd

    # ------------------------------------------------------------------ backward
    def backward(self, ent, d_prob_nhwc, d_feat_nhwc):
        """Replays the backward graph; returns (gradient views, embedding branch live).  d_*: NHWC contiguous or None."""
        if ent.g_bwd is None:
            raise RuntimeError("graphed backbone: backward of a forward that ran without gradients")
        if d_prob_nhwc is None:
            ent.sdprob.zero_()
        else:
            ent.sdprob.copy_(d_prob_nhwc)
        # the rule of the launch-by-launch path (Backbone.embed_ran): the projector's parameters have a gradient only if the
        # embedding branch is in this graph (return_feat) AND somebody read feat_2d.  With return_feat=False -- the contrast
        # warm-up epochs, trainer.py:625-630 -- the graph has no embedding branch and projector.* must stay at grad None
        # (AdamW then skips them: no weight decay, no step count), as under the reference
        live = bool(ent.embed) and ent.sdfeat is not None and d_feat_nhwc is not None
        if ent.sdfeat is not None:
            if d_feat_nhwc is None:
                # nobody read feat_2d: the reference's projector then has NO gradient (None, AdamW skips it).  The graph has the
                # embedding branch in it: fed zeros it adds exact zeros to the skips' gradients; the projector's are not bound
                ent.sdfeat.zero_()
            else:
                ent.sdfeat.copy_(d_feat_nhwc)
        ent.g_bwd.replay()
        ent.pending = False
        return ent.grads, live
