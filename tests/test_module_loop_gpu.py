"""The reference's training-loop STATEMENTS (train/dcgan_trainer.py:155-189, restated here - nothing is imported from the
reference) driven through the drop-in surface the way a maintainer who keeps the reference's trainer would: the HIP nn.Modules
(model/DCGAN.py of this package, hipgan.functional autograd Functions), nn.BCELoss, `loss.backward()`, and `optimizer.step()` =
hipgan.optim.EngineAdam.step() (one flat jck_adam launch over the arena the modules' parameters alias) - against the CPU oracle
(pinned to the reference) from identical weights, batch and noise.  Exact-fp32 path: losses 1e-3 per step, post-step weights
compared through Adam's lr-sized steps."""
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


def _noise(B, seed):
    g = torch.Generator().manual_seed(seed)
    return {"n1": torch.randn(B, 3, 64, 64, generator=g), "z": torch.randn(B, 100, 1, 1, generator=g),
            "n2": torch.randn(B, 3, 64, 64, generator=g), "alpha": torch.rand(B, 1, 1, 1, generator=g)}


def test_reference_loop_statements_on_the_hip_modules_with_engine_adam():
    from hipgan.engine import DcganEngine
    from hipgan.optim import EngineAdam
    from model import DCGAN
    from oracle.gan_oracle import GanOracle
    from util import synth_images
    B, lr, steps = 8, 2e-4, 2
    orc = GanOracle("dcgan", lr=lr, seed=12345)
    model_g, model_d = DCGAN.Generator().cuda(), DCGAN.Discriminator().cuda()
    model_g.prec = model_d.prec = "f32"
    eng = DcganEngine(batch=B, prec="f32")
    eng.adopt_modules(model_g, model_d)                       # parameters, .grad and Adam moments live in the engine's arenas
    eng.load_state(orc.g, orc.d)                              # ... so loading the oracle's state there is loading it into the modules
    opt_g = EngineAdam(eng, "g", model_g.named_parameters(), lr, betas=[0.5, 0.999])
    opt_d = EngineAdam(eng, "d", model_d.named_parameters(), lr, betas=[0.5, 0.999])
    criterion = nn.BCELoss()
    imgs = synth_images(B * steps)
    for s in range(steps):
        real_cpu, nz = imgs[s * B:(s + 1) * B], _noise(B, 300 + s)
        ref = orc.step(real_cpu, None, nz)
        n1, z, n2, alpha = (nz[k].cuda() for k in ("n1", "z", "n2", "alpha"))
        # ---- train/dcgan_trainer.py:155-180 -----------------------------------------------------------------------------
        model_d.zero_grad()
        real = 0.9 * real_cpu.cuda() + 0.1 * n1                                            # :160 instance noise
        label = torch.full((B,), 0.9, dtype=torch.float, device="cuda")                    # :161 smoothed labels
        output = model_d(real).view(-1)
        error_real = criterion(output, label)
        error_real.backward()                                                              # :164
        d_x = output.mean().item()
        fake = model_g(z)                                                                  # :169
        fake = 0.9 * fake + 0.1 * n2                                                       # :171
        label.fill_(0.1)
        output = model_d(fake.detach()).view(-1)
        error_fake = criterion(output, label)
        error_fake.backward()                                                              # :175
        d_gz1 = output.mean().item()
        inter = (alpha * real + (1 - alpha) * fake.detach()).requires_grad_(True)          # :110-127, value only in DCGAN
        d_inter = model_d(inter)
        grads = torch.autograd.grad(outputs=d_inter, inputs=inter, grad_outputs=torch.ones_like(d_inter))[0]
        gp = ((grads.view(B, -1).norm(2, dim=1) - 1) ** 2).mean()
        error_d = error_real + error_fake + 10 * gp                                        # :179
        opt_d.step()                                                                       # :180
        # ---- :182-189 ---------------------------------------------------------------------------------------------------
        model_g.zero_grad()
        label.fill_(0.9)
        output = model_d(fake).view(-1)
        error_g = criterion(output, label)
        error_g.backward()                                                                 # :187
        d_gz2 = output.mean().item()
        opt_g.step()                                                                       # :189
        got = {"loss_d": error_d.item(), "loss_g": error_g.item(), "gp": gp.item(), "loss_real": error_real.item(),
               "loss_fake": error_fake.item(), "d_x": d_x, "d_gz1": d_gz1, "d_gz2": d_gz2}
        for k, v in got.items():
            assert abs(v - ref[k]) <= 1e-3 * max(abs(ref[k]), 1e-12), (s, k, v, ref[k])
        # G's gradients of this step against the oracle's (relative L2 per tensor; D's were zeroed by model_d.zero_grad() only
        # at the top of the NEXT iteration, and hold real + fake + penalty-pass-free sums like the reference's)
        for k, p in model_g.named_parameters():
            r = orc.g_grads[k]
            l2 = ((p.grad.float().cpu() - r).norm() / (r.norm() + 1e-30)).item()
            # step 0 starts from identical state; step 1 from weights that have been through one Adam step of each path (ReLU /
            # LeakyReLU branch flips at batch 8: tests/test_step_gpu.py restarts from the oracle's state for that reason)
            assert l2 < (2e-2 if s == 0 else 1e-1), (s, k, l2)
    # weights after two optimiser steps each: Adam moves every element by ~lr per step, an element whose gradient is within
    # rounding of zero may have moved the other way (tests/test_step_gpu.py) - bound the distance by 2.5 steps
    for tag, mod, refp in (("g", model_g, orc.g), ("d", model_d, orc.d)):
        for k, p in mod.named_parameters():
            assert (p.detach().cpu() - refp[k]).abs().max().item() <= 2.5 * steps * lr, (tag, k)
            frac_far = ((p.detach().cpu() - refp[k]).abs() > 0.25 * lr).float().mean().item()
            assert frac_far < 0.15, (tag, k, frac_far)           # ... and most elements agree to a quarter of a step (the second step's
                                                                 # size is lr * g2-dependent: 6 % gradient noise at batch 8 moves it)
    sd = opt_d.state_dict()
    assert float(sd["state"][0]["step"]) == steps and eng.t == steps


def _cgan_noise(B, seed, labels):
    g = torch.Generator().manual_seed(seed)
    nz = {"n1": torch.randn(B, 3, 64, 64, generator=g), "z": torch.randn(B, 100, 1, 1, generator=g),
          "n2": torch.randn(B, 3, 64, 64, generator=g), "alpha": torch.rand(B, 1, 1, 1, generator=g), "labels": labels}
    for i in range(4):
        nz[f"m{i + 1}"] = (torch.rand(B, 256, generator=g) >= 0.25).float()
    return nz


def test_gradient_penalty_function_matches_double_backward_of_the_oracle():
    """hipgan.functional.gradient_penalty on the HIP CGAN discriminator: value and d(gp)/d(theta_D) against the oracle's
    autograd.grad(create_graph=True) + backward (train/cgan_trainer.py:114-131) on the same weights, images, alpha and mask."""
    from hipgan import functional as HF
    from model import CGAN
    from oracle import gan_oracle as go
    from util import synth_images, synth_onehot
    B = 8
    orc = go.GanOracle("cgan", lr=2e-4, seed=12345)
    d = CGAN.Discriminator().cuda()
    d.prec = "f32"
    d.load_state_dict({k: v.clone() for k, v in orc.d.items()})
    g = torch.Generator().manual_seed(5)
    real, fake = synth_images(B), torch.tanh(torch.randn(B, 3, 64, 64, generator=g))
    labels = synth_onehot(B)[0]
    alpha, mask = torch.rand(B, 1, 1, 1, generator=g), (torch.rand(B, 256, generator=g) >= 0.25).float()
    dp = {k: v.clone().requires_grad_(go.is_param(k)) for k, v in orc.d.items()}
    ref = go.gradient_penalty(dp, real, fake, alpha, labels, mask)
    names = [k for k in dp if go.is_param(k)]
    ref_grads = dict(zip(names, torch.autograd.grad(ref, [dp[k] for k in names], allow_unused=True)))
    gp = HF.gradient_penalty(d, real.cuda(), fake.cuda(), labels=labels.cuda(), alpha=alpha.cuda(), drop_mask=mask.cuda())
    assert abs(gp.item() - ref.item()) <= 1e-3 * abs(ref.item()), (gp.item(), ref.item())
    (10.0 * gp).backward()
    for k, p in d.named_parameters():
        r = ref_grads[k]
        if r is None:                       # parameters the penalty does not depend on (e.g. linear2.bias): exact zeros
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        l2 = ((p.grad.float().cpu() - 10.0 * r).norm() / (10.0 * r.norm() + 1e-30)).item()
        assert l2 < 5e-3, (k, l2)


def test_reference_cgan_loop_statements_on_the_hip_modules():
    """train/cgan_trainer.py:181-213 restated on the HIP modules with torch.optim.Adam, as a maintainer who keeps the reference's
    trainer and swaps only the model classes would run it: the penalty is BACK-PROPAGATED (error_d = real + fake + 10 gp)."""
    from hipgan import functional as HF
    from model import CGAN
    from oracle.gan_oracle import GanOracle
    from util import synth_images, synth_onehot
    B, lr = 8, 2e-4
    orc = GanOracle("cgan", lr=lr, seed=12345)
    model_g, model_d = CGAN.Generator().cuda(), CGAN.Discriminator().cuda()
    model_g.prec = model_d.prec = "f32"
    model_g.load_state_dict({k: v.clone() for k, v in orc.g.items()})
    model_d.load_state_dict({k: v.clone() for k, v in orc.d.items()})
    opt_g = torch.optim.Adam(model_g.parameters(), lr=lr, betas=[0.5, 0.999])
    opt_d = torch.optim.Adam(model_d.parameters(), lr=lr, betas=[0.5, 0.999])
    criterion = nn.BCELoss()
    real_cpu, labels_cpu = synth_images(B), synth_onehot(B)[0]
    nz = _cgan_noise(B, 700, labels_cpu)
    ref = orc.step(real_cpu, labels_cpu, nz)
    n1, z, n2, alpha = (nz[k].cuda() for k in ("n1", "z", "n2", "alpha"))
    m = [nz[f"m{i + 1}"].cuda() for i in range(4)]
    labels = labels_cpu.cuda()
    D = lambda x, mask: HF.cgan_discriminator(model_d, x, labels, "f32", mask=mask)       # model_d(x, labels) with the Dropout draw fixed
    # ---- D phase -------------------------------------------------------------------------------------------------------
    model_d.zero_grad()
    real = 0.9 * real_cpu.cuda() + 0.1 * n1
    label_real = torch.full((B,), 0.9, dtype=torch.float, device="cuda")       # (one tensor per target: error_d is back-propagated
    label_fake = torch.full((B,), 0.1, dtype=torch.float, device="cuda")       #  once, after all three terms exist)
    output = D(real, m[0]).view(-1)
    error_real = criterion(output, label_real)
    d_x = output.mean().item()
    fake = model_g(z, labels)
    fake = 0.9 * fake + 0.1 * n2
    output = D(fake.detach(), m[1]).view(-1)
    error_fake = criterion(output, label_fake)
    d_gz1 = output.mean().item()
    gp = HF.gradient_penalty(model_d, real.detach(), fake.detach(), labels=labels, alpha=alpha, drop_mask=m[2])
    error_d = error_real + error_fake + 10 * gp
    error_d.backward()                                                                     # :203 - through the penalty too
    for k, p in model_d.named_parameters():
        r = orc.d_grads[k]
        l2 = ((p.grad.float().cpu() - r).norm() / (r.norm() + 1e-30)).item()
        assert l2 < 5e-3, ("d_grads", k, l2)
    opt_d.step()
    # ---- G phase -------------------------------------------------------------------------------------------------------
    model_g.zero_grad()
    output = D(fake, m[3]).view(-1)
    error_g = criterion(output, label_real)
    error_g.backward()
    d_gz2 = output.mean().item()
    for k, p in model_g.named_parameters():
        r = orc.g_grads[k]
        l2 = ((p.grad.float().cpu() - r).norm() / (r.norm() + 1e-30)).item()
        assert l2 < 2e-2, ("g_grads", k, l2)
    opt_g.step()
    got = {"loss_d": error_d.item(), "loss_g": error_g.item(), "gp": gp.item(), "loss_real": error_real.item(),
           "loss_fake": error_fake.item(), "d_x": d_x, "d_gz1": d_gz1, "d_gz2": d_gz2}
    for k, v in got.items():
        assert abs(v - ref[k]) <= 1e-3 * max(abs(ref[k]), 1e-12), (k, v, ref[k])
    for tag, mod, refp in (("g", model_g, orc.g), ("d", model_d, orc.d)):
        for k, p in mod.named_parameters():
            assert (p.detach().cpu() - refp[k]).abs().max().item() <= 2.5 * lr, (tag, k)
