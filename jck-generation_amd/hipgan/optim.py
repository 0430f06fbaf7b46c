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

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    def zero_grad(self, set_to_none=False):
        self.engine.arenas[f"{self.tag}_grads"].zero_()

    def step(self):
        raise RuntimeError("the optimiser step is fused into the native engine step (jck_engine_phase); it is not called "
                           "separately")

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
        g = sd["param_groups"][0]
        self.param_groups[0]["lr"] = g["lr"]
