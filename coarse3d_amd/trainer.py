"""One optimisation step of the weak-segmentation trainer on the HIP path.

Mirrors tasks/weak_segmentation/trainer.py:599-704 (input normalisation, model forward with the
prototype bank update, focal + Lovasz, entropy-based pseudo-label selection, prototype
contrastive loss, backward, AdamW, scheduler) with the per-iteration host synchronisations
(``.item()`` logging :750-802, barrier + scalar all-reduces :740-743) removed from the step.
Config keys follow tasks/weak_segmentation/option.py:43-49 / config_semantic_kitti.yaml:20-42.

``proto_loss`` defaults to False like the reference trainer, which never passes it to the model
(trainer.py:625-630): the bank is then only l2-renormalised each step.  ``proto_loss=True`` runs
the prototype update (Sinkhorn + EMA + bank exchange, salsanext_proto.py:337-402) inside the
step -- what BASELINE.json's north star measures; bench.py and the golden step test opt in.
"""
import os

import numpy as np
import torch

from . import contrast, loss_head, ops
from .pc_processor.loss import ContrastMEMLoss, FocalSoftmaxLoss, Lovasz_softmax
from .pc_processor.loss.lovasz_softmax import valid_indices


def select_ratio_for(epoch, n_epochs):
    """trainer.py:655-661."""
    return float(np.log(1 + (1 + epoch) / n_epochs) / np.log(2) * 0.5)


class TrainStep:
    def __init__(self, model, n_classes, *, lr=1e-3, n_epochs=100, temperature=0.07, num_anchor=512,
                 loss_w_ce_2d=1.0, loss_w_lov_2d=1.0, loss_w_contrast=0.1, contrast_warmup=0,
                 entropy_selection=True, ignore_cls=0, cls_weight=None, feature_mean=None, feature_std=None,
                 proto_loss=False, optimizer=None, scheduler=None, inputs_resident=False, graph=None, graph_warmup=2):
        self.model = model
        self.net = model.module if hasattr(model, "module") else model
        # This step owns zero_grad / backward / optimizer.step: on a plain (unwrapped) model the backward pass may
        # write the gradients into one persistent buffer and bind param.grad itself (models that know the switch;
        # under a wrapper -- DistributedDataParallel hooks, coarse3d_amd.dist.DataParallel -- gradients keep going
        # through autograd / the wrapper's flat buffer).
        if self.net is model and hasattr(model, "_bind_grads"):
            model._bind_grads = True
        self.n_classes = n_classes
        self.n_epochs = n_epochs
        self.w_ce, self.w_lov, self.w_con = loss_w_ce_2d, loss_w_lov_2d, loss_w_contrast
        self.contrast_warmup = contrast_warmup
        self.entropy_selection = entropy_selection
        self.ignore_cls = ignore_cls
        self.proto_loss = proto_loss
        dev = next(self.net.parameters()).device
        if cls_weight is None:
            alpha = np.ones(n_classes, dtype=np.float32)
        else:                                           # trainer.py:351-354
            alpha = np.log(1 + np.asarray(cls_weight, dtype=np.float64))
            alpha = (alpha / alpha.max()).astype(np.float32)
        alpha[0] = 0
        self.focal = FocalSoftmaxLoss(n_classes, gamma=2, alpha=alpha, softmax=False)
        self.lovasz = Lovasz_softmax(ignore=ignore_cls, per_image=False, softmax=False)
        self.contrast = ContrastMEMLoss(ignore_label=ignore_cls, temperature=temperature, num_anchor=num_anchor)
        self.mean = torch.as_tensor(feature_mean, dtype=torch.float32, device=dev) if feature_mean is not None else None
        self.std = torch.as_tensor(feature_std, dtype=torch.float32, device=dev) if feature_std is not None else None
        # trainer.py:146-151: AdamW(params, lr) -- cfg.weight_decay is NOT passed (default 0.01).  On the device the
        # same optimiser runs in its fused form (one multi-tensor launch per ~30 tensors instead of ~20 elementwise
        # passes over the 192 parameter tensors: 0.38 -> 0.1 ms per step).
        if optimizer is None:
            fused = dev.type == "cuda"
            named = views = flat_buf = None
            if fused:
                if self.net is model and getattr(model, "_bind_grads", False):
                    named, names, _ = model._cached()
                    views = model._bound_grad_views(names)    # None while some param.grad is pending (e.g. a second TrainStep)
                    flat_buf = model._own_flat[1] if views is not None else None
                elif hasattr(model, "flat") and getattr(self.net, "_flat_grads", None) is not None:
                    # coarse3d_amd.dist.DataParallel: the gradients already live in ONE flat buffer (laid out in backward
                    # completion order, all-reduced in place); the parameters and moments take the same layout
                    P = dict(self.net._trainable())
                    named = [(n, P[n]) for n in model.flat.names]
                    views, flat_buf = model.flat.views, model.flat.flat
            if views is not None:
                # one parameter group, one formula: step all tensors as ONE flat buffer (coarse3d_amd/optim.py); parameters
                # without a gradient (the projector during a contrast warm-up) are skipped as torch's AdamW skips them
                from .optim import FlatAdamW
                optimizer = FlatAdamW(named, views, flat_buf, lr=lr, all_params=list(self.net.parameters()))
                own = self.net._own_flat
                self.net.invalidate_caches()             # parameter storage moved into the flat buffer
                self.net._own_flat = own
                if hasattr(self.net, "_packs"):
                    self.net._packs = ops.PackCache()
            else:
                optimizer = torch.optim.AdamW(self.net.parameters(), lr=lr, **({"fused": True} if fused else {}))
        self.optimizer = optimizer
        self.scheduler = scheduler
        self._side = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        # Round 5, OFF by default (C3D_OVERLAP_CONTRAST=1 / overlap_contrast = True): pseudo-label selection + contrast loss + its
        # gradient on a SECOND stream, under the segmentation losses and the decoder's backward; the backbone takes the
        # embedding's gradient behind its up blocks.  Same kernels, same values: bit-identical to the sequential step
        # (tests/test_gpu_step.py), and the replayed graph does run the two branches on two queues (tools/trace_overlap.py) --
        # but the step is not faster (30.71 / 30.80 / 30.85 vs 30.75 / 30.82 / 30.75 ms, same box, alternating): the branch's
        # ~0.8 ms are HBM-bound streams over the 1M-pixel maps (entropy statistics 22 us alone, 390 us next to the backward's
        # kernels), not idle latency.  Plain (unwrapped) SalsaNextProto with the rows-on-demand embedding only.
        self.overlap_contrast = os.environ.get("C3D_OVERLAP_CONTRAST", "0") == "1"
        self._side2 = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self.late_steps = 0          # steps (eager bodies / captures) that took the second-stream form
        # True: the caller guarantees that the label tensors handed to step() are complete in HBM
        # (e.g. delivered by a prefetcher on its own copy stream and synchronised) -- the side
        # stream may then read them without waiting for the main stream.
        self.inputs_resident = inputs_resident
        # focal + Lovasz on the fused HIP loss head when the labelled pixels fit its LDS sort
        # (weak labels: always); otherwise the sync-free PyTorch-op restatements
        self.fused_loss_head = True
        self.pl_noise = None     # test hook: Exp(1) noise [B, C, HW] for the pseudo-label selection
        # prototype similarity at the labelled pixels only (False: the full [N, C*M] map every step)
        self.sparse_proto = True
        # captured step (hipGraph): opt-in, C3D_GRAPH=1 makes it the default of this process
        self.graph = (os.environ.get("C3D_GRAPH", "0") == "1") if graph is None else bool(graph)
        if graph_warmup < 2:
            raise ValueError("graph_warmup >= 2: the first eager step records the weight repacks the plan needs, the "
                             "second builds the batched-repack table; neither may happen inside a capture")
        self.graph_warmup = graph_warmup
        if self.graph and hasattr(self.optimizer, "tensor_lr"):
            self.optimizer.tensor_lr = True          # a captured update must not bake a Python float in
        # One graph per (input shape, dtype, embedding branch on / off) -- NOT per epoch: the one epoch-dependent number
        # of the step, the pseudo-label ratio (trainer.py:655-661), travels as a device scalar like the learning rate.
        # All captures share one memory pool (only one step's tensors are live at a time) and at most ``max_graphs``
        # are kept (least recently used goes first), so a 100-epoch run holds one step's activations, not one per epoch.
        self._graphs = {}                # key -> dict(graph, static inputs, results, eager steps seen, ...)
        self.max_graphs = 4
        self._pool = None
        self._ratio_t = torch.zeros(1, device=dev, dtype=torch.float32) if dev.type == "cuda" else None
        self._ratio_seen = None
        self._replays = 0
        self._captures = 0
        # labelled-pixel count of every replay, copied to pinned host memory without a synchronisation and checked
        # when it has arrived (a step or two later): more labelled pixels than the shape-static list holds raises
        self._cnt_ring = []
        self._cnt_host = None
        self._cnt_free = None

    def step(self, x, train_label, eval_label, epoch=0):
        """x [B,5,H,W] fp32, labels [B,H,W] int64 (0 = ignore).  Returns dict of 0-dim loss tensors
        (still on the device: nothing here synchronises with the host except Lovasz' nonzero).

        ``graph=True``: after ``graph_warmup`` eager steps the whole step -- input normalisation, forward,
        prototype update, losses, pseudo-label selection, backward, the data-parallel exchanges, AdamW: ~700 kernel
        launches -- is captured in ONE hipGraph per (input shape, embedding branch on / off) and replayed with a single
        launch (the host needs ~20 ms to enqueue the launches of a step one by one; a 32x1024 step takes the GPU less
        than that).  The returned tensors are then the graph's static outputs: they are overwritten by the next step."""
        if self.graph:
            return self._graph_step(x, train_label, eval_label, epoch)
        lov_valid = None
        if self.w_lov > 0 and self._side is not None:
            # The Lovasz loss needs the list of labelled pixels, whose length is data dependent (the
            # only host synchronisation of the step).  It depends on the labels alone: compute it now
            # on a side stream; with ``inputs_resident`` it does not wait for the previous step still
            # running on the main stream, so the host blocks for microseconds and the main stream
            # never drains.
            if not self.inputs_resident:
                self._side.wait_stream(torch.cuda.current_stream())   # labels may still be in flight on the main stream
            with torch.cuda.stream(self._side):
                lov_valid = valid_indices(train_label, self.ignore_cls)
            torch.cuda.current_stream().wait_stream(self._side)
            lov_valid.record_stream(torch.cuda.current_stream())
        res = self._body(x, train_label, eval_label, epoch, lov_valid, None)
        if self.scheduler is not None:
            self.scheduler.step()
        return res

    def _body(self, x, train_label, eval_label, epoch, lov_valid, lov_count, ratio=None):
        """Everything of the step that runs on the device.  ``lov_count`` (device int32 [1]): ``lov_valid`` is the
        fixed-capacity list of ``loss_head.valid_indices_static`` (the shape-static, capturable form)."""
        net = self.net
        wss_mask = train_label > 0
        if self.mean is not None:
            x = ops.input_norm(x.contiguous(), eval_label.contiguous(), self.mean, self.std)
        return_feat = epoch >= self.contrast_warmup
        if (self.sparse_proto and return_feat and self.proto_loss and lov_valid is not None and self.ignore_cls == getattr(net, "ignore_label", None)
                and hasattr(net, "_labelled_hint")):
            # the prototype update only needs the embedding rows of the labelled pixels -- the list the Lovasz head uses
            # (same labels, same ignore class): hand it to the forward so that LayerNorm + l2 + the similarity GEMM run
            # on those rows instead of all B*H*W.  Capacity: the fused loss head's; more labelled pixels
            # than that take the dense path.
            if lov_count is not None:
                net._labelled_hint = (lov_valid, lov_count)
            elif lov_valid.numel() <= ops.lovasz_max_pixels():
                net._labelled_hint = loss_head.valid_indices_static(train_label, self.ignore_cls)
        out = self.model(x, label=train_label if return_feat else None, eval_mask=wss_mask if return_feat else None,
                         return_feat=return_feat, proto_loss=self.proto_loss)
        pred = out["pred_2d"]
        total = pred.new_zeros(())
        res = {}
        late_c = None
        feats_rows = getattr(out, "feat_rows", None) if (self.w_con > 0 and return_feat) else None
        if (self.overlap_contrast and self._side2 is not None and isinstance(feats_rows, contrast.LowResFeat) and net is self.model
                and getattr(net, "_late_dfeat_ok", False) and not getattr(net, "graph_backbone", False) and feats_rows.low.requires_grad
                and not self.contrast.keep_debug and not self.contrast.is_debug):
            cur = torch.cuda.current_stream()
            s2 = self._side2
            s2.wait_stream(cur)
            with torch.cuda.stream(s2), torch.no_grad():
                lab_c, mask_c = self._contra_labels(pred, train_label, eval_label, wss_mask, epoch, ratio)
                loss_c, d_low = self.contrast(feats=feats_rows, output=pred, labels=lab_c, keep_mask=mask_c,
                                              proto_queue=net.prototypes.detach().unsqueeze(0), explicit_grad_scale=self.w_con)
                ev = torch.cuda.Event()
                ev.record(s2)
            for t_ in (lab_c, mask_c, loss_c, d_low):       # allocated on the second stream, read on this one from here on
                t_.record_stream(cur)
            net._late_dfeat = (ev, d_low)
            late_c = (ev, loss_c, lab_c, mask_c)
            self.late_steps += 1
        fused = (self.fused_loss_head and lov_valid is not None and self.ignore_cls == 0
                 and (lov_count is not None or loss_head.fused_available(lov_valid.numel())))
        if fused:      # focal + Lovasz forward/backward on the HIP loss-head kernels
            if self.focal.alpha.device != pred.device:       # once: a per-step H2D copy would drain the stream
                self.focal.alpha = self.focal.alpha.to(pred.device)
            ce, lov = loss_head.loss_head(pred, train_label, wss_mask, self.focal.alpha,
                                          self.focal.gamma, lov_valid, self.w_ce > 0, self.w_lov > 0, count=lov_count)
            if self.w_ce > 0:
                res["ce"] = ce
                total = total + self.w_ce * ce
            res["lov"] = lov
            total = total + self.w_lov * lov
        else:
            if self.w_ce > 0:
                res["ce"] = self.focal(pred, train_label, mask=wss_mask)
                total = total + self.w_ce * res["ce"]
            if self.w_lov > 0:
                res["lov"] = self.lovasz(pred, train_label, valid=lov_valid)
                total = total + self.w_lov * res["lov"]
        if late_c is not None:
            res["labels_contra"], res["mask_contra"] = late_c[2], late_c[3]
        elif self.w_con > 0 and return_feat:
            lab_c, mask_c = self._contra_labels(pred, train_label, eval_label, wss_mask, epoch, ratio)
            res["labels_contra"], res["mask_contra"] = lab_c, mask_c
            queue = net.prototypes.detach().unsqueeze(0)
            # (the embedding as rows-on-demand where the model offers it: the anchors are interpolated from the
            #  half-resolution tensor and ``feat_2d`` is never materialised; contrast.LowResFeat)
            feats = getattr(out, "feat_rows", None)
            res["contrast"] = self.contrast(feats=feats if feats is not None else out["feat_2d"], output=pred,
                                            labels=lab_c, keep_mask=mask_c, proto_queue=queue)
            total = total + self.w_con * res["contrast"]
        self.optimizer.zero_grad(set_to_none=True)
        total.backward()
        if late_c is not None:
            net._late_dfeat = None
            torch.cuda.current_stream().wait_event(late_c[0])       # (the backbone has waited for it already)
            res["contrast"] = late_c[1]
            total = total.detach() + self.w_con * late_c[1]
        if hasattr(self.model, "finish_gradients"):
            self.model.finish_gradients()
        self.optimizer.step()
        res["loss"] = total.detach()
        res["pred_2d"] = pred.detach()
        return res

    def _contra_labels(self, pred, train_label, eval_label, wss_mask, epoch, ratio):
        """Labels / keep mask of the contrast loss: entropy-selected pseudo labels (trainer.py:656-662) or the weak labels."""
        if not self.entropy_selection:
            return train_label, wss_mask
        with torch.no_grad():
            if ratio is None:       # (captured / shape-static step: the device scalar TrainStep._sync_ratio keeps)
                ratio = select_ratio_for(epoch, self.n_epochs)
            return contrast.entropy_selection(pred.detach().permute(0, 2, 3, 1).contiguous(), train_label, eval_label, ratio,
                                              noise=self.pl_noise, ignore_cls=self.ignore_cls)

    # ------------------------------------------------------------------ captured step
    def _graph_supported(self):
        """The captured step covers a plain model and ``coarse3d_amd.dist.DataParallel`` (the exchanges are then part
        of the graph: RCCL collectives are capturable); anything it cannot express raises here instead of replaying
        something else."""
        if self.net is not self.model and not hasattr(self.model, "flat"):
            return ("the model is wrapped by something other than coarse3d_amd.dist.DataParallel (stock "
                    "DistributedDataParallel reduces through autograd hooks on the host)")
        if self.net is not self.model:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_backend() != "nccl":
                return f"the process group's backend is {dist.get_backend()!r}: only RCCL collectives can be captured"
        if not (self.fused_loss_head and self.ignore_cls == 0 and self.w_lov > 0):
            return "the captured step needs the fused loss head (ignore_cls == 0, Lovasz on)"
        if self.pl_noise is not None or self.contrast.uniforms is not None or self.contrast.perms is not None:
            return "injected randomness (test hooks) lives on the host"
        if getattr(self.net, "dropout_masks", None) is not None or getattr(self.net, "gumbel_noise", None) is not None:
            return "injected randomness (test hooks) lives on the host"
        if not hasattr(self.net, "_static_bank"):
            return "the model does not know the in-place bank update"
        import coarse3d_amd
        if not coarse3d_amd.GRAPH_REPLAY_SAFE and os.environ.get("C3D_GRAPH_UNSAFE") != "1":
            return ("the HIP runtime was initialised before `import coarse3d_amd` without DEBUG_CLR_GRAPH_PACKET_CAPTURE=0: a "
                    "captured step would fault when replayed after ~2 000 unrelated launches (coarse3d_amd/__init__.py).  Import "
                    "coarse3d_amd before the first torch.cuda call, export the variable, or set C3D_GRAPH_UNSAFE=1 to capture anyway")
        if type(self.optimizer).__name__ != "FlatAdamW":
            return ("the captured step needs the flat optimiser (coarse3d_amd.optim.FlatAdamW: learning rate as a device "
                    "scalar, gradient-less parameters skipped); a user-supplied optimiser runs with graph=False")
        return None

    def _sync_ratio(self, epoch):
        r = float(np.float32(select_ratio_for(epoch, self.n_epochs)))
        if r != self._ratio_seen:
            self._ratio_t.fill_(r)
            self._ratio_seen = r

    def _graph_step(self, x, train_label, eval_label, epoch):
        why = self._graph_supported()
        if why is not None:
            raise RuntimeError(f"TrainStep(graph=True): {why}")
        return_feat = epoch >= self.contrast_warmup
        key = (tuple(x.shape), str(x.dtype), bool(return_feat))
        ent = self._graphs.get(key)
        if ent is None:
            ent = self._graphs[key] = {"graph": None, "eager": 0, "used": 0}
        ent["used"] = self._replays + self._captures + sum(e["eager"] for e in self._graphs.values())
        opt = self.optimizer
        opt.sync_lr()                                    # schedulers write param_groups["lr"]; the graph reads a device scalar
        self._sync_ratio(epoch)
        self._poll_capacity()
        packs = getattr(self.net, "_packs", None)
        if ent["graph"] is not None and packs is not None and ent["packs_generation"] != packs.generation:
            ent["graph"] = None                          # the weight-repack table moved: its address is baked into the graph
            ent["eager"] = min(ent["eager"], self.graph_warmup - 1)
        if ent["graph"] is not None and ent.get("opt_generation") != getattr(opt, "generation", None):
            # the optimiser's segments were re-cut (another configuration's eager step) or replaced (load_state_dict):
            # the graph increments step counters and updates slices that are no longer the optimiser's
            ent["graph"] = None
            ent["eager"] = min(ent["eager"], self.graph_warmup - 1)
        self.net._static_bank = True
        if ent["graph"] is None and ent["eager"] < self.graph_warmup:
            # lazy initialisation (weight-pack table, kernel attributes, philox state, the optimiser's segments) must
            # happen outside a capture: the first steps of every configuration run eagerly -- on the same shape-static
            # loss head and device scalars the graph uses, so that eager and captured steps are the same arithmetic
            ent["eager"] += 1
            idx, cnt = loss_head.valid_indices_static(train_label, self.ignore_cls)
            res = self._body(x, train_label, eval_label, epoch, idx, cnt, ratio=self._ratio_t)
            self._check_capacity(int(cnt))
            if self.scheduler is not None:
                self.scheduler.step()
            return res
        if ent["graph"] is None:
            self._evict(keep=key)
            if not any(e["graph"] is not None for e in self._graphs.values()):
                # the last graph that used the shared pool is gone (dropped above, or evicted): the allocator has released
                # the pool and refuses its handle ("use_count > 0" assert in capture_begin) -- start a new one
                self._pool = None
            sx, st, se = x.clone(), train_label.clone(), eval_label.clone()
            self._check_capacity(int(loss_head.valid_indices_static(st, self.ignore_cls)[1]))
            if self._pool is None:
                self._pool = torch.cuda.graph_pool_handle()
            self._drain_process_group()
            g = torch.cuda.CUDAGraph()
            from . import dist as c3d_dist
            counts0 = dict(c3d_dist.COUNTS)
            exposed, c3d_dist.EXPOSED = c3d_dist.EXPOSED, None      # HIP events cannot time launches inside a capture
            try:
                # thread_local: torch.distributed's watchdog thread may query events while this thread captures
                with torch.cuda.graph(g, pool=self._pool, capture_error_mode="thread_local"):
                    idx, cnt = loss_head.valid_indices_static(st, self.ignore_cls)
                    res = self._body(sx, st, se, epoch, idx, cnt, ratio=self._ratio_t)
            finally:
                c3d_dist.EXPOSED = exposed
            res["lov_count"] = cnt
            ent.update(graph=g, sx=sx, st=st, se=se, res=res,
                       packs_generation=packs.generation if packs is not None else 0,
                       opt_generation=getattr(opt, "generation", None),
                       collectives={k: c3d_dist.COUNTS[k] - counts0[k] for k in counts0})
            for k in counts0:                            # the capture itself executed nothing
                c3d_dist.COUNTS[k] = counts0[k]
            self._captures += 1
        ent["sx"].copy_(x)
        ent["st"].copy_(train_label)
        ent["se"].copy_(eval_label)
        ent["graph"].replay()
        self._replays += 1
        if ent["collectives"]:
            from . import dist as c3d_dist
            for k, v in ent["collectives"].items():
                c3d_dist.COUNTS[k] += v
        self._watch_capacity(ent["res"]["lov_count"])
        if hasattr(self.model, "watch_status"):
            # data parallel: the health word of this replay's exchanges (coarse3d_amd/dist.py) -- host code does not replay
            self.model.watch_status()
        if self.scheduler is not None:
            self.scheduler.step()
        return ent["res"]

    @staticmethod
    def _drain_process_group():
        """Before a capture that contains collectives: torch.distributed's watchdog thread polls the completion events of
        the collectives issued so far, and HIP refuses an event query once the event's stream (the process group's
        communication stream) has joined a capture -- which it does with the first captured collective.  Let everything
        issued so far finish and the watchdog retire it; collectives issued DURING a capture are not handed to the
        watchdog."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return
        torch.cuda.synchronize()
        pg = dist.distributed_c10d._get_default_group()
        wait = getattr(pg, "_wait_for_pending_works", None)
        if wait is not None:
            wait()
        else:                                            # older torch: the watchdog sweeps every 100 ms
            import time
            time.sleep(0.5)

    def _evict(self, keep):
        """Drop least-recently-used captured graphs beyond ``max_graphs`` (their tensors live in the shared pool, which
        later captures reuse)."""
        live = [k for k, e in self._graphs.items() if e["graph"] is not None and k != keep]
        live.sort(key=lambda k: self._graphs[k]["used"])
        while len(live) + 1 > self.max_graphs:
            k = live.pop(0)
            del self._graphs[k]

    def _check_capacity(self, n):
        """More labelled pixels than the fused loss head sorts in LDS cannot go through the shape-static step (its
        list would be truncated)."""
        if n > ops.lovasz_max_pixels():
            raise RuntimeError(f"TrainStep(graph=True): {n} labelled pixels exceed the fused loss head's capacity "
                               f"({ops.lovasz_max_pixels()}); train fully supervised batches with graph=False")

    def _watch_capacity(self, count):
        """Queue an asynchronous read-back of this replay's labelled-pixel count: 4 bytes into one of eight slots of a
        pinned buffer allocated once (a fresh pinned allocation per step costs ~10 ms of host time while the GPU is busy)."""
        if self._cnt_host is None:
            self._cnt_host = torch.zeros(8, dtype=torch.int32).pin_memory()
            self._cnt_free = list(range(8))
        if not self._cnt_free:                           # never more than eight steps behind
            ev0, slot0 = self._cnt_ring.pop(0)
            ev0.synchronize()
            self._cnt_free.append(slot0)
            self._check_capacity(int(self._cnt_host[slot0]))
        slot = self._cnt_free.pop(0)
        self._cnt_host[slot:slot + 1].copy_(count, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._cnt_ring.append((ev, slot))

    def flush(self):
        """Wait for every queued labelled-pixel count and check it.  The capacity check of a captured step is asynchronous:
        an overflow raises one to eight steps late, AFTER the offending replay's optimizer update (made with a truncated
        Lovasz / prototype list) has been applied -- the weights, moments and bank are contaminated from that step on, so
        treat the error as fatal for the run (resume from the last checkpoint with graph=False).  Call this before writing
        a checkpoint and at the end of an epoch so that an overflow in the last steps cannot go unnoticed."""
        while self._cnt_ring:
            ev, slot = self._cnt_ring.pop(0)
            ev.synchronize()
            self._cnt_free.append(slot)
            self._check_capacity(int(self._cnt_host[slot]))
        if hasattr(self.model, "check_status"):
            self.model.check_status(final=True)

    def _poll_capacity(self):
        while self._cnt_ring and self._cnt_ring[0][0].query():
            _, slot = self._cnt_ring.pop(0)
            self._cnt_free.append(slot)
            self._check_capacity(int(self._cnt_host[slot]))
