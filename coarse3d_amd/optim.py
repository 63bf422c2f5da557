"""AdamW over ONE flat parameter / gradient / state buffer.

The reference builds ``torch.optim.AdamW(model.parameters(), lr)`` (tasks/weak_segmentation/trainer.py:146-151: one
parameter group, default betas / eps / weight decay on every tensor).  With one group the update is the same elementwise
formula on every parameter, so 192 tensors can be stepped as one: torch's own fused AdamW kernel (``torch._fused_adamw_``)
over single flat buffers is ONE launch instead of six multi-tensor launches of ~30 small tensors each (6 x 40 us -> ~20 us at
7.5 M parameters), and the host side of the step shrinks from ~5 ms to ~0.1 ms.  Same arithmetic, bit-identical update
(tests/test_gpu_step.py).

The parameters keep their identity (``p.data`` becomes a view of the flat buffer), ``param_groups`` lists them as
usual (learning-rate schedulers work unchanged), and the gradients are expected in the model's persistent flat gradient
buffer (``SalsaNextProto._bound_grad_views``); a gradient that arrived any other way is copied in first.

**Parameters without a gradient are skipped, as torch's AdamW skips them** (no weight decay, no step count, no moment
decay): the flat buffers are cut into *segments* -- maximal runs of parameters that have always been stepped together --
each with its own step counter, and one ``_fused_adamw_`` call steps the segments whose parameters carry a gradient.
With the reference's shipped ``contrast_warmup: 5`` (config_semantic_kitti.yaml:20) the projector has no gradient during
the first epochs: the buffer splits once into [backbone | projector], the backbone segment is stepped alone until the
embedding branch switches on.  A captured step (hipGraph) bakes the list of active segments in, which is why
``TrainStep`` keeps one graph per (shape, embedding branch on / off).

``state_dict()`` / ``load_state_dict()`` speak the checkpoint layout of ``torch.optim.AdamW(model.parameters())`` -- the
reference's optimiser (trainer.py:129, main.py:141,154): ``param_groups[0]["params"]`` indexes ALL parameters of the model
in ``model.parameters()`` order (197 for SalsaNextProto: the frozen ``prototypes`` is index 0, ``feat_norm.*`` /
``mask_norm.*`` are 193-196), and ``state`` holds entries for the parameters that have been stepped."""
import torch


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, named_params, grad_views, flat_grad, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                 all_params=None):
        """named_params: (name, parameter) of the parameters that live in the flat buffers, in buffer order.
        all_params: every parameter of the model in ``model.parameters()`` order (the index space of checkpoints and of
        ``param_groups[0]["params"]``); default: the flat ones only."""
        named = list(named_params)
        params = [p for _, p in named]
        every = list(all_params) if all_params is not None else params
        pos = {id(p): i for i, p in enumerate(every)}
        if any(id(p) not in pos for p in params):
            raise ValueError("FlatAdamW: all_params must contain every flat parameter")
        super().__init__(every, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        total = sum(p.numel() for p in params)
        if flat_grad.numel() != total:
            raise ValueError("FlatAdamW: the flat gradient buffer does not match the parameters")
        dev = params[0].device
        self.flat_param = torch.empty(total, device=dev, dtype=torch.float32)
        self._offsets = [0]
        with torch.no_grad():
            for p in params:
                off = self._offsets[-1]
                view = self.flat_param[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view                     # the Parameter object (and everything holding it) stays
                self._offsets.append(off + p.numel())
        self.flat_grad = flat_grad
        self.grad_views = [grad_views[n] for n, _ in named]
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        # segments [first flat parameter, one past the last) with their own step counters; one segment until some
        # parameters are stepped without the others
        self._segments = [[0, len(params), torch.zeros((), device=dev, dtype=torch.float32)]]
        # the learning rate as a device scalar: a captured step (TrainStep(graph=True)) must not bake a Python float in;
        # refreshed from param_groups[0]["lr"] (what schedulers write) by ``sync_lr`` outside the graph
        self.lr_t = torch.full((), float(lr), device=dev, dtype=torch.float32)
        self._lr_seen = float(lr)
        # False: the update takes the learning rate as a Python float (double precision inside the kernel, bit-identical
        # to torch.optim.AdamW(fused=True)); True (TrainStep(graph=True)): as the float32 device scalar above
        self.tensor_lr = False
        self._params = params
        self._index = [pos[id(p)] for p in params]       # flat parameter -> index in model.parameters()
        # bumped whenever the segment list (boundaries or step-counter tensors) is replaced: a captured step has the
        # old slices and counters baked in and must be dropped (TrainStep._graph_step compares it, like packs.generation)
        self.generation = 0

    @property
    def step_t(self):
        """Step counter of the first segment (the whole buffer while every parameter has always had a gradient)."""
        return self._segments[0][2]

    def zero_grad(self, set_to_none=True):
        if set_to_none:
            for p in self._params:
                p.grad = None
        else:
            self.flat_grad.zero_()

    def _split(self, active):
        """Cut the segments so that each one is entirely active or entirely inactive (a cut clones the step counter:
        both halves have the same history up to now).  Not during a capture: the segment list is part of what a
        graph bakes in, so the eager steps ahead of a capture have to run in the same configuration."""
        out = []
        for lo, hi, st in self._segments:
            start = lo
            for i in range(lo + 1, hi + 1):
                if i == hi or active[i] != active[start]:
                    out.append([start, i, st if start == lo else st.clone()])
                    start = i
        if len(out) != len(self._segments):
            if self.flat_param.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FlatAdamW: the set of parameters with a gradient changed inside a graph capture; run "
                                   "an eager step in this configuration first")
            self._segments = out
            self.generation += 1

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._check_aliasing()
        active = []
        for p, v in zip(self._params, self.grad_views):
            g = p.grad
            active.append(g is not None)
            if g is not None and g.data_ptr() != v.data_ptr():   # came through autograd's AccumulateGrad, not the bound buffer
                v.copy_(g)
        self._split(active)
        segs = [s for s in self._segments if active[s[0]]]
        if not segs:
            return loss
        grp = self.param_groups[0]
        if not (self.flat_param.is_cuda and torch.cuda.is_current_stream_capturing()):
            self.sync_lr()
        o = self._offsets
        cut = lambda t: [t[o[lo]:o[hi]] for lo, hi, _ in segs]    # noqa: E731
        steps = [s[2] for s in segs]
        torch._foreach_add_(steps, 1)
        torch._fused_adamw_(cut(self.flat_param), cut(self.flat_grad), cut(self.exp_avg), cut(self.exp_avg_sq), [], steps,
                            amsgrad=False, lr=self.lr_t if self.tensor_lr else float(grp["lr"]), beta1=grp["betas"][0],
                            beta2=grp["betas"][1],
                            weight_decay=grp["weight_decay"], eps=grp["eps"], maximize=False, grad_scale=None, found_inf=None)
        return loss

    def sync_lr(self):
        """Copy param_groups[0]["lr"] into the device scalar the update reads (no-op while it is unchanged)."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_seen:
            self.lr_t.fill_(lr)
            self._lr_seen = lr

    def _check_aliasing(self):
        """``p.data`` must still be the views of ``flat_param`` made in __init__: model.double() / .to(device) /
        load_state_dict(assign=True) rebind it, and the flat update would then move an orphaned buffer while the
        model silently stops training."""
        lo = self.flat_param.data_ptr()
        hi = lo + self.flat_param.numel() * 4
        for p in (self._params[0], self._params[-1]):
            if not (lo <= p.data_ptr() < hi) or p.dtype != torch.float32:
                raise RuntimeError("FlatAdamW: a parameter no longer lives in the flat buffer (dtype / device change or "
                                   "load_state_dict(assign=True) after the optimiser was built): rebuild the optimiser")

    def state_dict(self):
        """The layout of ``torch.optim.AdamW(model.parameters()).state_dict()`` -- what the reference writes into its
        checkpoints (tasks/weak_segmentation/main.py:141,154) and reads back with ``optimizer.load_state_dict``
        (trainer.py:129): ``param_groups[0]["params"] = range(number of model parameters)`` and per-parameter
        ``state[i] = {step, exp_avg, exp_avg_sq}`` for the parameters that have been stepped (parameters that never
        had a gradient -- the frozen bank and LayerNorms, the projector during the contrast warm-up -- have no entry,
        as in torch)."""
        groups = [dict({k: v for k, v in g.items() if k != "params"}, params=list(range(len(g["params"]))))
                  for g in self.param_groups]
        state = {}
        o = self._offsets
        for lo, hi, st in self._segments:
            if float(st) <= 0:
                continue
            for j in range(lo, hi):
                p = self._params[j]
                state[self._index[j]] = {"step": st.clone(), "exp_avg": self.exp_avg[o[j]:o[j + 1]].view_as(p).clone(),
                                         "exp_avg_sq": self.exp_avg_sq[o[j]:o[j + 1]].view_as(p).clone()}
        return {"state": state, "param_groups": groups}

    @torch.no_grad()
    def load_state_dict(self, sd):
        """Accepts ``torch.optim.AdamW(model.parameters())`` checkpoints (this class, torch, the reference), the
        trainable-parameters-only layout round 3 wrote (indices 0..191) and round 2's flat layout
        (``{"flat": True, step, exp_avg, exp_avg_sq}``).  Parameters without an entry start from zero moments and step
        0; runs of parameters with equal step counts become the segments of the flat update."""
        dev = self.flat_param.device
        if self.flat_param.is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("FlatAdamW.load_state_dict inside a graph capture")
        self.generation += 1                 # new step-counter tensors / segment boundaries: captured steps are stale
        if sd.get("flat"):
            self._segments = [[0, len(self._params), torch.zeros((), device=dev, dtype=torch.float32)]]
            self.step_t.copy_(sd["step"])
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        else:
            if len(sd["param_groups"]) != 1:
                raise ValueError("FlatAdamW.load_state_dict: expected one parameter group")
            ids = list(sd["param_groups"][0]["params"])
            n_all = len(self.param_groups[0]["params"])
            if len(ids) == n_all:
                where = [ids[i] for i in self._index]                 # model.parameters() order
            elif len(ids) == len(self._params):
                where = ids                                          # the trainable parameters only (round 3)
            else:
                raise ValueError(f"FlatAdamW.load_state_dict: the checkpoint covers {len(ids)} parameters; this model has "
                                 f"{n_all} ({len(self._params)} trainable)")
            state = sd["state"]
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            steps = []
            o = self._offsets
            for j, key in enumerate(where):
                st = state.get(key)
                if st is None:
                    steps.append(0.0)
                    continue
                n = o[j + 1] - o[j]
                if st["exp_avg"].numel() != n:
                    raise ValueError(f"FlatAdamW.load_state_dict: state {key} does not fit its parameter")
                self.exp_avg[o[j]:o[j + 1]].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[o[j]:o[j + 1]].copy_(st["exp_avg_sq"].reshape(-1))
                steps.append(float(st["step"]))
            segs, start = [], 0
            for j in range(1, len(steps) + 1):
                if j == len(steps) or steps[j] != steps[start]:
                    segs.append([start, j, torch.full((), steps[start], device=dev, dtype=torch.float32)])
                    start = j
            self._segments = segs
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in s.items() if k != "params"})
