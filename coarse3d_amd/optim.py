"""AdamW over ONE flat parameter / gradient / state buffer.

The reference builds ``torch.optim.AdamW(model.parameters(), lr)`` (tasks/weak_segmentation/trainer.py:146-151: one
parameter group, default betas / eps / weight decay on every tensor).  With one group the update is the same elementwise
formula on every parameter, so 192 tensors can be stepped as one: torch's own fused AdamW kernel (``torch._fused_adamw_``)
over single flat buffers is ONE launch instead of six multi-tensor launches of ~30 small tensors each (6 x 40 us -> ~20 us at
7.5 M parameters), and the host side of the step shrinks from ~5 ms to ~0.1 ms.  Same arithmetic, bit-identical update
(tests/test_gpu_step.py).

The parameters keep their identity (``p.data`` becomes a view of the flat buffer), ``param_groups`` lists them as
usual (learning-rate schedulers work unchanged), and the gradients are expected in the model's persistent flat gradient
buffer (``SalsaNextProto._bound_grad_views``); a gradient that arrived any other way is copied in first."""
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
        for p, v in zip(self._params, self.grad_views):
            g = p.grad
            if g is None:
                raise RuntimeError("FlatAdamW.step: a parameter has no gradient (one flat update covers all of them)")
            if g.data_ptr() != v.data_ptr():      # came through autograd's AccumulateGrad instead of the bound buffer
                v.copy_(g)
        grp = self.param_groups[0]
        torch._foreach_add_([self.step_t], 1)
        torch._fused_adamw_([self.flat_param], [self.flat_grad], [self.exp_avg], [self.exp_avg_sq], [], [self.step_t],
                            amsgrad=False, lr=float(grp["lr"]), beta1=grp["betas"][0], beta2=grp["betas"][1],
                            weight_decay=grp["weight_decay"], eps=grp["eps"], maximize=False, grad_scale=None, found_inf=None)
        return loss

    def state_dict(self):
        return {"flat": True, "step": self.step_t.clone(), "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        if not sd.get("flat"):
            raise ValueError("FlatAdamW.load_state_dict: not a FlatAdamW state (per-parameter AdamW states have another layout)")
        self.step_t.copy_(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
