"""CPU oracle for the DCGAN / CGAN training hot path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this
module; the product (`jck-generation_amd/`) never does and fails loudly without its HIP library.

It restates, in plain fp32 `torch` CPU ops, what the reference computes on the path named by
BASELINE.json's north_star.  The arithmetic itself lives in PyTorch/ATen (third party, not vendored
under /root/reference, no version pin in the reference repo); this file restates the reference's
*algorithm* - network topology, step order, loss, penalty, optimiser - on top of the same library:

    nets       model/DCGAN.py:6-67, model/CGAN.py:79-162
    init       model/DCGAN.py:70-76  (weights_init, applied G then D in train/dcgan_trainer.py:54-55)
    DCGAN step train/dcgan_trainer.py:155-189     GP: train/dcgan_trainer.py:110-127
    CGAN step  train/cgan_trainer.py:173-213      GP: train/cgan_trainer.py:114-131
    Adam       torch.optim.Adam(lr, betas=[0.5, 0.999]) as constructed at train/dcgan_trainer.py:61-62
    BCE        nn.BCELoss on a sigmoid output (train/dcgan_trainer.py:64) incl. the -100 log clamp

Parity pin: `tests/test_oracle_golden.py` checks this file against fixtures captured by running the
reference's own trainers in the build container (`tests/golden/make_golden.py`).  Status: PINNED
for nets, step scalars, gradients, post-step weights, BN running stats; Inception features are
"parity unpinned" (weights not available offline).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LRELU = 0.2
LAMBDA_GP = 10.0          # train/dcgan_trainer.py:49
LABEL_REAL = 0.9          # train/dcgan_trainer.py:136
LABEL_FAKE = 0.1          # train/dcgan_trainer.py:137
NOISE_KEEP, NOISE_MIX = 0.9, 0.1   # train/dcgan_trainer.py:160,171

# (c_small, c_big) per stride-2 stage.  D: conv(big -> small); G: convT(small -> big).
D_CH = [(64, 3), (128, 64), (256, 128), (512, 256)]          # model/DCGAN.py:10-22
G_CH = [(512, 256), (256, 128), (128, 64), (64, 3)]          # model/DCGAN.py:46-58


def channel_plan(image_size=64):
    """-> (D_CH, G_CH, channels at the 4x4 end).  64 = the reference (above).  128 = BASELINE.json configs[4]: the reference
    has NO 128x128 behaviour (its nets are hard-wired to 3x64x64) - this restatement grows the nets by one
    stride-2 stage at the deep end, the usual DCGAN recipe, and is PARITY UNPINNED against the reference (nothing to pin to);
    it is pinned per operator against torch itself (tests/test_ops_gpu.py) and serves as the oracle of the HIP 128x128 path."""
    if image_size == 64:
        return D_CH, G_CH, 512
    assert image_size == 128, "image_size must be 64 or 128"
    return D_CH + [(1024, 512)], [(1024, 512)] + G_CH, 1024


def _n_norm(p):
    return sum(1 for k in p if k.startswith("norm") and k.endswith(".weight"))


# ----------------------------------------------------------------------------------------------
# parameter construction: consumes the global RNG exactly like `Generator()`, `Discriminator()`
# followed by `.apply(weights_init)` on G then D.
# ----------------------------------------------------------------------------------------------
def _bn_state(p: Dict[str, torch.Tensor], name: str, c: int):
    p[f"{name}.weight"] = torch.ones(c)
    p[f"{name}.bias"] = torch.zeros(c)
    p[f"{name}.running_mean"] = torch.zeros(c)
    p[f"{name}.running_var"] = torch.ones(c)
    p[f"{name}.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)


def _construct_g(family: str, image_size: int = 64) -> Dict[str, torch.Tensor]:
    zin = 100 if family == "dcgan" else 200                  # model/CGAN.py:132
    p: Dict[str, torch.Tensor] = {}
    _, g_ch, c_top = channel_plan(image_size)
    chans = [(zin, c_top)] + [(a, b) for a, b in g_ch]
    for i, (ci, co) in enumerate(chans, 1):
        st, pad = (1, 0) if i == 1 else (2, 1)
        p[f"conv{i}.weight"] = torch.nn.ConvTranspose2d(ci, co, 4, st, pad, bias=False).weight.detach().clone()
        if i < len(chans):
            _bn_state(p, f"norm{i}", co)
    return p


def _construct_d(family: str, image_size: int = 64) -> Dict[str, torch.Tensor]:
    p: Dict[str, torch.Tensor] = {}
    d_ch, _, c_top = channel_plan(image_size)
    if family == "cgan":                                     # model/CGAN.py:83
        lin = torch.nn.Linear(100, 200)
        p["label_embedding.weight"], p["label_embedding.bias"] = lin.weight.detach().clone(), lin.bias.detach().clone()
    for i, (cs, cb) in enumerate(d_ch, 1):
        p[f"conv{i}.weight"] = torch.nn.Conv2d(cb, cs, 4, 2, 1, bias=False).weight.detach().clone()
        _bn_state(p, f"norm{i}", cs)
    if family == "dcgan":
        p[f"conv{len(d_ch) + 1}.weight"] = torch.nn.Conv2d(c_top, 1, 4, 1, 0, bias=False).weight.detach().clone()
    else:                                                    # model/CGAN.py:103-106
        l1 = torch.nn.Linear(8192 + 200, 256)
        p["linear1.weight"], p["linear1.bias"] = l1.weight.detach().clone(), l1.bias.detach().clone()
        l2 = torch.nn.Linear(256, 1)
        p["linear2.weight"], p["linear2.bias"] = l2.weight.detach().clone(), l2.bias.detach().clone()
    return p


def _weights_init(p: Dict[str, torch.Tensor]):
    """model/DCGAN.py:70-76 in `.apply` order (registration order of the sub-modules)."""
    for k in list(p.keys()):
        mod, _, leaf = k.rpartition(".")
        if mod.startswith("conv") and leaf == "weight":
            p[k].normal_(0.0, 0.02)
        elif mod.startswith("norm") and leaf == "weight":
            p[k].normal_(1.0, 0.02)
        elif mod.startswith("norm") and leaf == "bias":
            p[k].zero_()


def is_param(k: str) -> bool:
    return not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))


def build_params(family: str = "dcgan", image_size: int = 64):
    """-> (g_state, d_state) with the reference's state-dict keys; draws from the global RNG."""
    g = _construct_g(family, image_size)
    d = _construct_d(family, image_size)
    _weights_init(g)
    _weights_init(d)
    return g, d


# ----------------------------------------------------------------------------------------------
# functional nets (train-mode BatchNorm always: the reference never calls .eval() on G/D)
# ----------------------------------------------------------------------------------------------
def _bn(p, name, x, update_stats=True):
    rm, rv = p[f"{name}.running_mean"], p[f"{name}.running_var"]
    if update_stats:
        p[f"{name}.num_batches_tracked"] += 1
        return F.batch_norm(x, rm, rv, p[f"{name}.weight"], p[f"{name}.bias"], True, BN_MOMENTUM, BN_EPS)
    return F.batch_norm(x, None, None, p[f"{name}.weight"], p[f"{name}.bias"], True, BN_MOMENTUM, BN_EPS)


def generator(p, z, labels=None):
    """model/DCGAN.py:61-67, model/CGAN.py:151-162."""
    if labels is not None:
        z = torch.cat([z, labels.reshape(-1, 100, 1, 1)], 1)      # int64 one-hot promoted to float
    n = _n_norm(p)                                             # 4 (the reference), 5 for the 128x128 plan
    h = F.conv_transpose2d(z, p["conv1.weight"], None, 1, 0)
    h = F.relu(_bn(p, "norm1", h))
    for i in range(2, n + 1):
        h = F.conv_transpose2d(h, p[f"conv{i}.weight"], None, 2, 1)
        h = F.relu(_bn(p, f"norm{i}", h))
    return torch.tanh(F.conv_transpose2d(h, p[f"conv{n + 1}.weight"], None, 2, 1))


def discriminator(p, x, labels=None, drop_mask=None, drop_p=0.25):
    """model/DCGAN.py:29-35, model/CGAN.py:109-123.  `drop_mask` (0/1, [B,256]) replaces the RNG draw
    of nn.Dropout(0.25) when given; None draws from the global generator like the reference."""
    h = x
    n = _n_norm(p)
    for i in range(1, n + 1):
        h = F.conv2d(h, p[f"conv{i}.weight"], None, 2, 1)
        h = F.leaky_relu(_bn(p, f"norm{i}", h), LRELU)
    if f"conv{n + 1}.weight" in p:
        return torch.sigmoid(F.conv2d(h, p[f"conv{n + 1}.weight"], None, 1, 0))
    e = F.leaky_relu(F.linear(labels.float(), p["label_embedding.weight"], p["label_embedding.bias"]), LRELU)
    con = torch.cat([h.flatten(1), e], 1)
    h = F.linear(con, p["linear1.weight"], p["linear1.bias"])
    if drop_mask is None:
        h = F.dropout(h, drop_p, True)
    else:
        h = h * drop_mask / (1.0 - drop_p)
    return torch.sigmoid(F.linear(h, p["linear2.weight"], p["linear2.bias"]))


def bce(prob, target: float):
    """nn.BCELoss(mean): -[t*max(log p,-100) + (1-t)*max(log(1-p),-100)]."""
    return F.binary_cross_entropy(prob, torch.full_like(prob, target))


def gradient_penalty(d_state, real, fake, alpha, labels=None, drop_mask=None, create_graph=True, disc=None, store=None, grad_store=None):
    """train/dcgan_trainer.py:110-127.  disc / store / grad_store: the bf16-storage emulation's net, its rounding of the
    interpolated image and of the image gradient the norm is taken of (oracle/bf16_emu.py)."""
    inter = alpha * real + ((1 - alpha) * fake)
    inter = (store(inter) if store is not None else inter).requires_grad_(True)
    di = (disc or discriminator)(d_state, inter, labels, drop_mask)
    grads = torch.autograd.grad(di, inter, torch.ones_like(di), create_graph=create_graph, retain_graph=True)[0]
    if grad_store is not None:
        grads = grad_store(grads)
    grads = grads.view(grads.size(0), -1)
    return ((grads.norm(2, dim=1) - 1) ** 2).mean()


# ----------------------------------------------------------------------------------------------
# Adam, restated (torch.optim.Adam single-tensor path, amsgrad=False, weight_decay=0, eps=1e-8)
# ----------------------------------------------------------------------------------------------
class Adam:
    def __init__(self, params: Dict[str, torch.Tensor], lr, b1=0.5, b2=0.999, eps=1e-8):
        self.p, self.lr, self.b1, self.b2, self.eps, self.t = params, lr, b1, b2, eps, 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}

    @torch.no_grad()
    def step(self, grads: Dict[str, torch.Tensor]):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        for k, w in self.p.items():
            g = grads[k]
            self.m[k].lerp_(g, 1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            w.addcdiv_(self.m[k], denom, value=-(self.lr / bc1))


# ----------------------------------------------------------------------------------------------
# the training step
# ----------------------------------------------------------------------------------------------
class GanOracle:
    """State + one-step function for either family.

    `step(real, labels=None, noise=None)`: `noise` = dict(n1, z, n2, alpha[, m1..m4]) of CPU tensors
    to teacher-force every random draw; with None the draws come from the global torch generator in
    the reference's order (n1, z, n2, [dropout masks inside each D call], alpha).
    """

    def __init__(self, family="dcgan", lr=2e-4, seed: Optional[int] = 12345, skip_dead_wgrad=False, emulate_bf16=False,
                 image_size=64):
        """emulate_bf16: round to bf16 wherever the HIP fast path stores a tensor in bf16 (oracle/bf16_emu.py) -
        the envelope for the fast path's distance from this fp32 restatement.  Default off = the pinned oracle."""
        assert family in ("dcgan", "cgan")
        self.emu = emulate_bf16
        self._grad_store = None
        if emulate_bf16 and family == "cgan":       # twice-differentiable restatement: the penalty is back-propagated
            from . import bf16_emu
            self._gen, self._disc, self._store = bf16_emu.generator_cgan_bf16, bf16_emu.discriminator_cgan_bf16, bf16_emu.store_dd
            self._grad_store = bf16_emu.st          # the penalty's image gradient is stored in bf16 before its norm is taken
        elif emulate_bf16:
            from . import bf16_emu
            self._gen, self._disc, self._store = bf16_emu.generator_bf16, bf16_emu.discriminator_bf16, bf16_emu.store
        else:
            self._gen, self._disc, self._store = generator, discriminator, None
        if seed is not None:
            torch.manual_seed(seed)
        self.family = family
        assert image_size == 64 or family == "dcgan", "the 128x128 plan exists for DCGAN only"
        self.g, self.d = build_params(family, image_size)
        self.gp_params = {k: v for k, v in self.g.items() if is_param(k)}
        self.dp_params = {k: v for k, v in self.d.items() if is_param(k)}
        self.opt_g = Adam(self.gp_params, lr)
        self.opt_d = Adam(self.dp_params, lr)
        self.d_grads: Dict[str, torch.Tensor] = {}
        self.g_grads: Dict[str, torch.Tensor] = {}
        self.skip_dead_wgrad = skip_dead_wgrad

    # -- helpers ------------------------------------------------------------------------------
    def _req(self, params, flag=True):
        for v in params.values():
            v.requires_grad_(flag)

    def sample(self, z, labels=None):
        """train/dcgan_trainer.py:199-200: no_grad, train-mode BN (running stats DO move)."""
        with torch.no_grad():
            return self._gen(self.g, z, labels)

    def step(self, real, labels=None, noise=None):
        ctx = self.phase_d(real, labels, noise)
        self.apply_d(ctx["d_grads"])
        self.phase_g(ctx)
        self.apply_g(ctx["g_grads"])
        return self.finish(ctx)

    # The step is split at the two optimiser calls so that an N-replica run can average gradients
    # between `phase_*` and `apply_*`.
    def phase_d(self, real, labels=None, noise=None):
        fam = self.family
        B = real.size(0)
        nz = noise or {}
        mask = lambda k: nz.get(k) if fam == "cgan" else None
        self._req(self.dp_params)
        self._req(self.gp_params)
        dnames = list(self.dp_params)
        dpl = [self.dp_params[k] for k in dnames]

        n1 = nz["n1"] if "n1" in nz else torch.randn(real.size())
        real = NOISE_KEEP * real + NOISE_MIX * n1
        if self.emu:
            real = self._store(real)
        out_real = self._disc(self.d, real, labels, mask("m1")).view(-1)
        e_real = bce(out_real, LABEL_REAL)
        z = nz["z"] if "z" in nz else torch.randn(B, 100, 1, 1)
        fake_raw = self._gen(self.g, z, labels)
        n2 = nz["n2"] if "n2" in nz else torch.randn(fake_raw.size())
        fake = NOISE_KEEP * fake_raw + NOISE_MIX * n2
        if self.emu:
            fake = self._store(fake)
        out_fake = self._disc(self.d, fake.detach(), labels, mask("m2")).view(-1)
        e_fake = bce(out_fake, LABEL_FAKE)
        alpha = nz["alpha"] if "alpha" in nz else torch.rand(B, 1, 1, 1)
        if fam == "dcgan":
            # train/dcgan_trainer.py:164,175: two separate backward calls; the penalty (178) is never
            # back-propagated - it only enters the logged error_d (179) and moves D's BN statistics.
            dg = torch.autograd.grad(e_real + e_fake, dpl, retain_graph=False)
            gp = gradient_penalty(self.d, real, fake, alpha, create_graph=not self.emu, disc=self._disc, store=self._store).detach()
        else:
            # train/cgan_trainer.py:200-203: one backward over real + fake + 10*GP (double backward).
            gp = gradient_penalty(self.d, real.detach(), fake.detach(), alpha, labels, mask("m3"), disc=self._disc, store=self._store,
                                  grad_store=self._grad_store)
            dg = torch.autograd.grad(e_real + e_fake + LAMBDA_GP * gp, dpl)
            gp = gp.detach()
        e_d = (e_real + e_fake).detach() + LAMBDA_GP * gp
        self.d_grads = dict(zip(dnames, dg))
        return {"d_grads": self.d_grads, "fake": fake, "labels": labels, "m4": mask("m4"),
                "loss_d": float(e_d), "gp": float(gp), "loss_real": float(e_real.detach()), "loss_fake": float(e_fake.detach()),
                "out_real": out_real.detach(), "out_fake": out_fake.detach(), "real_noisy": real.detach()}

    def apply_d(self, grads):
        self.opt_d.step(grads)

    def phase_g(self, ctx):
        gnames = list(self.gp_params)
        gpl = [self.gp_params[k] for k in gnames]
        out_g = self._disc(self.d, ctx["fake"], ctx["labels"], ctx["m4"]).view(-1)
        e_g = bce(out_g, LABEL_REAL)
        gg = torch.autograd.grad(e_g, gpl)
        self.g_grads = dict(zip(gnames, gg))
        ctx.update(g_grads=self.g_grads, loss_g=float(e_g.detach()), out_g=out_g.detach())
        return ctx

    def apply_g(self, grads):
        self.opt_g.step(grads)

    def finish(self, ctx):
        self._req(self.dp_params, False)
        self._req(self.gp_params, False)
        return {"loss_d": ctx["loss_d"], "loss_g": ctx["loss_g"], "gp": ctx["gp"],
                "loss_real": ctx["loss_real"], "loss_fake": ctx["loss_fake"],
                "out_real": ctx["out_real"], "out_fake": ctx["out_fake"], "out_g": ctx["out_g"],
                "d_x": float(ctx["out_real"].mean()), "d_gz1": float(ctx["out_fake"].mean()),
                "d_gz2": float(ctx["out_g"].mean()), "fake": ctx["fake"].detach(),
                "real_noisy": ctx["real_noisy"]}


# ----------------------------------------------------------------------------------------------
# N-replica data parallel oracle: N copies on N shards, gradients averaged
# ----------------------------------------------------------------------------------------------
def ddp_step(replicas, reals, labels=None, noises=None):
    """One data-parallel step over `len(replicas)` GanOracle copies holding identical weights: each
    sees its own shard (local BatchNorm statistics, its own noise), the D and G gradients are
    averaged across replicas before each optimiser step.  Returns the per-replica result dicts."""
    n = len(replicas)
    labels = labels or [None] * n
    noises = noises or [None] * n
    ctxs = [r.phase_d(x, l, nz) for r, x, l, nz in zip(replicas, reals, labels, noises)]
    avg = lambda key: {k: torch.stack([c[key][k] for c in ctxs]).mean(0) for k in ctxs[0][key]}
    dg = avg("d_grads")
    for r in replicas:
        r.apply_d(dg)
    for r, c in zip(replicas, ctxs):
        r.phase_g(c)
    gg = avg("g_grads")
    for r in replicas:
        r.apply_g(gg)
    return [r.finish(c) for r, c in zip(replicas, ctxs)]
