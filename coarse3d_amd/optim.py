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
``state_dict()`` / ``load_state_dict()`` speak torch.optim.AdamW's checkpoint layout.

One difference from per-parameter AdamW is structural: the flat update steps EVERY parameter, so it is only used when
every parameter receives a gradient in every step.  ``TrainStep`` therefore falls back to ``torch.optim.AdamW`` when
``contrast_warmup > 0`` (the projector has no gradient during the warm-up epochs; the reference's AdamW skips such
parameters -- no weight decay, no step count -- and so must we)."""
import torch


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, named_params, grad_views, flat_grad, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        named = list(named_params)
        params = [p for _, p in named]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        total = sum(p.numel() for p in params)
        if flat_grad.numel() != total:
            raise ValueError("FlatAdamW: the flat gradient buffer does not match the parameters")
        dev = params[0].device
        self.flat_param = torch.empty(total, device=dev, dtype=torch.float32)
        off = 0
        with torch.no_grad():
            for p in params:
                view = self.flat_param[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view                     # the Parameter object (and everything holding it) stays
                off += p.numel()
        self.flat_grad = flat_grad
        self.grad_views = [grad_views[n] for n, _ in named]
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        self.step_t = torch.zeros((), device=dev, dtype=torch.float32)
        # the learning rate as a device scalar: a captured step (TrainStep(graph=True)) must not bake a Python float in;
        # refreshed from param_groups[0]["lr"] (what schedulers write) by ``sync_lr`` outside the graph
        self.lr_t = torch.full((), float(lr), device=dev, dtype=torch.float32)
        self._lr_seen = float(lr)
        # False: the update takes the learning rate as a Python float (double precision inside the kernel, bit-identical
        # to torch.optim.AdamW(fused=True)); True (TrainStep(graph=True)): as the float32 device scalar above
        self.tensor_lr = False
        self._params = params

    def zero_grad(self, set_to_none=True):
        if set_to_none:
            for p in self._params:
                p.grad = None
        else:
            self.flat_grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._check_aliasing()
        for p, v in zip(self._params, self.grad_views):
            g = p.grad
            if g is None:
                raise RuntimeError("FlatAdamW.step: a parameter has no gradient (one flat update covers all of them)")
            if g.data_ptr() != v.data_ptr():      # came through autograd's AccumulateGrad instead of the bound buffer
                v.copy_(g)
        grp = self.param_groups[0]
        if not (self.flat_param.is_cuda and torch.cuda.is_current_stream_capturing()):
            self.sync_lr()
        torch._foreach_add_([self.step_t], 1)
        torch._fused_adamw_([self.flat_param], [self.flat_grad], [self.exp_avg], [self.exp_avg_sq], [], [self.step_t],
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
        """The layout of ``torch.optim.AdamW.state_dict()`` -- what the reference writes into its checkpoints
        (tasks/weak_segmentation/main.py:141,154) and reads back with ``optimizer.load_state_dict``
        (trainer.py:129): per-parameter ``state[i] = {step, exp_avg, exp_avg_sq}`` sliced from the flat buffers and
        ``param_groups`` with parameter indices, so that checkpoints move freely between this optimiser,
        torch.optim.AdamW and the reference.  Before the first step the state is empty, as torch's is."""
        groups = [dict({k: v for k, v in g.items() if k != "params"}, params=list(range(len(self._params))))
                  for g in self.param_groups]
        state = {}
        if float(self.step_t) > 0:
            off = 0
            for i, p in enumerate(self._params):
                n = p.numel()
                state[i] = {"step": self.step_t.clone(), "exp_avg": self.exp_avg[off:off + n].view_as(p).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + n].view_as(p).clone()}
                off += n
        return {"state": state, "param_groups": groups}

    @torch.no_grad()
    def load_state_dict(self, sd):
        """Accepts the standard AdamW layout (from this class, torch.optim.AdamW or a reference checkpoint) and the
        flat layout round 2 wrote (``{"flat": True, step, exp_avg, exp_avg_sq}``).  One update covers all
        parameters, so the per-parameter step counts must agree (they do whenever all parameters were trained
        together, which is the only way the reference trains them)."""
        if sd.get("flat"):
            self.step_t.copy_(sd["step"])
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        else:
            if len(sd["param_groups"]) != 1 or len(sd["param_groups"][0]["params"]) != len(self._params):
                raise ValueError("FlatAdamW.load_state_dict: expected one parameter group over the same parameters")
            state = sd["state"]
            ids = sd["param_groups"][0]["params"]
            if not state:
                self.step_t.zero_()
                self.exp_avg.zero_()
                self.exp_avg_sq.zero_()
            else:
                if any(i not in state for i in ids):
                    raise ValueError("FlatAdamW.load_state_dict: AdamW state is missing for some parameters (the flat "
                                     "update steps all of them together)")
                steps = {float(state[i]["step"]) for i in ids}
                if len(steps) != 1:
                    raise ValueError(f"FlatAdamW.load_state_dict: per-parameter step counts differ ({sorted(steps)[:4]}...): "
                                     "load this checkpoint into torch.optim.AdamW (TrainStep(..., optimizer=...))")
                self.step_t.fill_(steps.pop())
                off = 0
                for i, p in zip(ids, self._params):
                    n = p.numel()
                    if state[i]["exp_avg"].numel() != n:
                        raise ValueError(f"FlatAdamW.load_state_dict: state {i} does not fit its parameter")
                    self.exp_avg[off:off + n].copy_(state[i]["exp_avg"].reshape(-1))
                    self.exp_avg_sq[off:off + n].copy_(state[i]["exp_avg_sq"].reshape(-1))
                    off += n
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in s.items() if k != "params"})
