"""Adam state that lives in the engine's flat arenas, presented with torch.optim.Adam's interface where the reference
touches it: `state_dict()` / `load_state_dict()` produce and accept exactly the dict torch.optim.Adam(lr, betas=[0.5, 0.999])
writes into a checkpoint (train/dcgan_trainer.py:61-62,86-91), so `.pt` files are interchangeable with the reference's."""
import torch


class EngineAdam:
    def __init__(self, engine, tag, params, lr, betas=(0.5, 0.999), eps=1e-8):
        self.engine, self.tag = engine, tag
        self.params = list(params)                      # [(name, Parameter)] in module.parameters() order
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                             capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False)
        self.param_groups = [dict(self.defaults, params=[p for _, p in self.params])]
        if not hasattr(engine, "_t_engine"):
            engine._t_engine = engine.t             # the step count every network of this engine shares so far (see step())

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    def zero_grad(self, set_to_none=False):
        self.engine.arenas[f"{self.tag}_grads"].zero_()

    def step(self):
        """`optimizer.step()` of a caller that keeps the reference's loop (train/dcgan_trainer.py:180,189) on the HIP modules: one
        flat jck_adam launch over this network's arena (parameters, gradients - what the modules' `.grad` alias - and moments),
        torch.optim.Adam's arithmetic.  The trainers of this package never call it: their step runs Adam inside
        jck_engine_phase.  Each optimiser counts its own steps from the engine's; the engine's count follows the larger one."""
        from ._lib import cur_stream, lib
        eng, a = self.engine, self.engine.arenas
        eng.join()
        # this network's step count: the engine's count at its last own step (or loaded checkpoint), plus the steps taken here since
        ep = getattr(eng, "_module_epoch", 0)
        if getattr(self, "_epoch", None) != ep:
            self._epoch, self._own_t = ep, getattr(eng, "_t_engine", eng.t)
        t = self._own_t + 1
        # gradients: autograd accumulates in place into `.grad` tensors that alias the arena (adopt_modules), but Module.zero_grad()
        # drops them by default (set_to_none=True, what train/dcgan_trainer.py:155,182 do) and the next backward then allocates
        # fresh ones - those are gathered into the arena here
        views = eng.named_views(self.tag, "grads")
        with torch.no_grad():
            for name, prm in self.params:
                v = views[name]
                if prm.grad is None:
                    v.zero_()
                elif prm.grad.data_ptr() != v.data_ptr():
                    v.copy_(prm.grad.to(v.dtype).view_as(v))
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        p = a[f"{self.tag}_params"]
        lib.jck_adam(p, a[f"{self.tag}_grads"], a[f"{self.tag}_m"], a[f"{self.tag}_v"], p.numel(), float(g["lr"]), float(b1), float(b2),
                     float(g["eps"]), t, 1.0, cur_stream())
        self._own_t = t
        eng.t = max(eng.t, t)                   # what state_dict() writes as "step"
        eng.mark_weights_changed()              # the packed GEMM operands are rebuilt before their next use

    def state_dict(self):
        m, v = self.engine.named_views(self.tag, "m"), self.engine.named_views(self.tag, "v")
        state = {}
        if self.engine.t > 0:
            for i, (name, _) in enumerate(self.params):
                state[i] = {"step": torch.tensor(float(self.engine.t)), "exp_avg": m[name].detach().clone(),
                            "exp_avg_sq": v[name].detach().clone()}
        group = {k: (list(val) if k == "betas" else val) for k, val in self.defaults.items()}
        group["betas"] = tuple(self.defaults["betas"])
        group["params"] = list(range(len(self.params)))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        m, v = self.engine.named_views(self.tag, "m"), self.engine.named_views(self.tag, "v")
        steps = set()
        for i, (name, _) in enumerate(self.params):
            st = sd["state"].get(i)
            if st is None:
                m[name].zero_()
                v[name].zero_()
                continue
            m[name].copy_(st["exp_avg"].to(m[name].device).view_as(m[name]))
            v[name].copy_(st["exp_avg_sq"].to(v[name].device).view_as(v[name]))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ: {sorted(steps)}")
        if steps:
            self.engine.t = steps.pop()
            self.engine._t_engine = self.engine.t
            self.engine._module_epoch = getattr(self.engine, "_module_epoch", 0) + 1
        g = sd["param_groups"][0]
        self.param_groups[0]["lr"] = g["lr"]
