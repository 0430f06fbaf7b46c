"""Pins the CPU oracle (oracle/*.py) against fixtures captured from the reference itself
(tests/golden/make_golden.py ran the reference's unmodified trainers / Metrics in the build
container).  CPU only.  Tolerances: fp32 re-association noise only (1e-5 relative on scalars,
2e-4 on gradient digests whose elements pass through ~10 layers)."""
import numpy as np
import pytest
import torch

from oracle import gan_oracle as O
from oracle import metrics_oracle as MO
from util import check_digest, check_digest_dict, load_golden, rel, synth_images, synth_onehot

SCALAR_RTOL = 2e-5
TENSOR_RTOL = 3e-4


def _inputs(B):
    gen = torch.Generator().manual_seed(77)
    z = torch.randn(B, 100, 1, 1, generator=gen)
    x = synth_images(B, seed=99)
    rg = torch.randn(B, 3, 64, 64, generator=gen)
    rd = torch.randn(B, generator=gen)
    oh, _ = synth_onehot(B, seed=5)
    return z, x, rg, rd, oh


@pytest.mark.parametrize("fam", ["dcgan", "cgan"])
def test_modules(fam):
    gold = load_golden("modules")[fam]
    torch.manual_seed(12345)
    g, d = O.build_params(fam)
    check_digest_dict(g, gold["init_g"], 0, 0, "init_g")
    check_digest_dict(d, gold["init_d"], 0, 0, "init_d")
    z, x, rg, rd, oh = _inputs(4)
    x.requires_grad_(True)
    gp = [v.requires_grad_(True) for k, v in g.items() if O.is_param(k)]
    dp = [v.requires_grad_(True) for k, v in d.items() if O.is_param(k)]
    lab = oh if fam == "cgan" else None
    fake = O.generator(g, z, lab)
    dout = O.discriminator(d, x, lab, drop_mask=torch.ones(4, 256) if fam == "cgan" else None, drop_p=0.0).view(-1)
    check_digest(fake, gold["g_out"], 1e-5, 1e-7, "g_out")
    np.testing.assert_allclose(dout.detach().double().numpy(), gold["d_out"]["vals"], rtol=1e-5)
    gg = torch.autograd.grad((fake * rg).sum(), gp)
    dg = torch.autograd.grad((dout * rd).sum(), dp + [x])
    gnames = [k for k in g if O.is_param(k)]
    dnames = [k for k in d if O.is_param(k)]
    check_digest_dict(dict(zip(gnames, gg)), gold["g_grads"], TENSOR_RTOL, 1e-6, "g_grads")
    check_digest_dict(dict(zip(dnames, dg[:-1])), gold["d_grads"], TENSOR_RTOL, 1e-6, "d_grads")
    check_digest(dg[-1], gold["d_xgrad"], TENSOR_RTOL, 1e-7, "d_xgrad")
    check_digest_dict(g, gold["g_post"], 1e-5, 1e-7, "g_post(bn running stats)")
    check_digest_dict(d, gold["d_post"], 1e-5, 1e-7, "d_post(bn running stats)")


def _replay(fam, key):
    gold = load_golden(f"{fam}_steps")[key]
    B, steps, lr = gold["B"], gold["steps"], gold["lr"]
    orc = O.GanOracle(fam, lr=lr, seed=12345)
    check_digest_dict(orc.g, gold["init_g"], 0, 0, "init_g")
    check_digest_dict(orc.d, gold["init_d"], 0, 0, "init_d")
    imgs = synth_images(B * steps)
    if fam == "dcgan":
        fixed = torch.randn(64, 100, 1, 1)                      # train/dcgan_trainer.py:139
        labels, fixed_l = None, None
    else:                                                       # train/cgan_trainer.py:144-153
        fixed = torch.vstack([torch.randn(10, 100, 1, 1) for _ in range(100)])
        fixed_l = torch.vstack([torch.nn.functional.one_hot(torch.tensor(i), 100).repeat(10, 1) for i in range(100)])
        labels, _ = synth_onehot(B * steps)
    res = []
    for s in range(steps):
        lab = labels[s * B:(s + 1) * B] if labels is not None else None
        r = orc.step(imgs[s * B:(s + 1) * B], lab)
        r["d_grads"], r["g_grads"] = dict(orc.d_grads), dict(orc.g_grads)
        r["g_state"] = {k: v.clone() for k, v in orc.g.items()}
        r["d_state"] = {k: v.clone() for k, v in orc.d.items()}
        res.append(r)
        if s == 0 or s == steps - 1:                            # eval branch, train/dcgan_trainer.py:198-200
            orc.sample(fixed, fixed_l)
    return gold, orc, res


@pytest.mark.parametrize("key", ["B8", "B64"])
def test_dcgan_steps(key):
    gold, orc, res = _replay("dcgan", key)
    for s, (r, gs) in enumerate(zip(res, gold["step"])):
        assert rel(r["loss_real"], gs["crit"][0]["loss"]) < SCALAR_RTOL
        assert rel(r["loss_fake"], gs["crit"][1]["loss"]) < SCALAR_RTOL * 10
        assert rel(r["loss_g"], gs["crit"][2]["loss"]) < SCALAR_RTOL * 50, (s, r["loss_g"], gs["crit"][2]["loss"])
        assert rel(r["gp"], gs["gp"]) < 2e-4
        assert rel(r["loss_d"], gold["losses_d"][s]) < 1e-4
        assert rel(r["loss_g"], gold["losses_g"][s]) < 1e-3
        np.testing.assert_allclose(r["out_real"].double().numpy(), gs["crit"][0]["out"], rtol=2e-4 * (1 + 5 * s))
        if s == 0:   # identical state: tight.  Later steps inherit Adam's amplification of 1e-7 noise.
            check_digest_dict(r["d_grads"], gs["d_grads"], TENSOR_RTOL, 1e-7, f"s{s}.d_grads")
            check_digest_dict(r["g_grads"], gs["g_grads"], TENSOR_RTOL, 1e-7, f"s{s}.g_grads")
            check_digest_dict(r["d_state"], gs["d_post"], 1e-5, 2e-7, f"s{s}.d_post", skip=("running", "num_batches"))
            check_digest_dict(r["g_state"], gs["g_post"], 1e-5, 2e-7, f"s{s}.g_post")
            check_digest_dict(r["d_state"], gs["d_post_bn"], 1e-5, 1e-7, f"s{s}.d_bn")
    check_digest_dict(orc.g, gold["final_g"], 5e-3, 2e-5, "final_g")
    check_digest_dict(orc.d, gold["final_d"], 5e-3, 2e-5, "final_d")
    assert gold["ckpt_keys"] == ["model_d", "model_g", "optimizer_d", "optimizer_g"]


def test_dcgan_default_lr_clamp():
    """CLI default lr 0.1 (main.py:54): after one step D saturates and BCELoss's -100 log clamp is hit:
    loss_g = 0.1*100 = 10, loss_d = 10 + 90 + 10*GP(=1) = 110."""
    gold = load_golden("dcgan_steps")["B8_lr0.1"]
    orc = O.GanOracle("dcgan", lr=0.1, seed=12345)
    fixed = torch.randn(64, 100, 1, 1)
    imgs = synth_images(8 * 4)
    for s in range(4):
        r = orc.step(imgs[s * 8:(s + 1) * 8])
        if s == 0:
            orc.sample(fixed)
        assert rel(r["loss_d"], gold["losses_d"][s]) < 1e-5, (s, r["loss_d"])
        assert rel(r["loss_g"], gold["losses_g"][s]) < 1e-5, (s, r["loss_g"])
        assert rel(r["gp"], gold["gp"][s]) < 2e-4
    assert gold["losses_d"][1:] == [110.0] * 3


@pytest.mark.parametrize("key", ["B8", "B32"])
def test_cgan_steps(key):
    gold, orc, res = _replay("cgan", key)
    for s, (r, gs) in enumerate(zip(res, gold["step"])):
        tol = 1e-4 if s == 0 else 5e-3
        assert rel(r["gp"], gs["gp"]) < tol * 3
        assert rel(r["loss_d"], gold["losses_d"][s]) < tol
        assert rel(r["loss_g"], gold["losses_g"][s]) < tol * 10
        if s == 0:
            check_digest_dict(r["d_grads"], gs["d_grads"], 1e-3, 1e-7, f"s{s}.d_grads")
            check_digest_dict(r["g_grads"], gs["g_grads"], 1e-3, 1e-7, f"s{s}.g_grads")
            check_digest_dict(r["d_state"], gs["d_post"], 1e-5, 2e-7, f"s{s}.d_post", skip=("running", "num_batches"))
            check_digest_dict(r["g_state"], gs["g_post"], 1e-5, 2e-7, f"s{s}.g_post")


def test_metrics_arithmetic():
    gold = load_golden("metrics")
    rng = np.random.default_rng(7)
    real = (1.5 * rng.standard_normal((5000, 100)) + 0.2).astype(np.float32)
    fake = rng.standard_normal((1000, 100)).astype(np.float32)
    real_targets = rng.integers(0, 100, size=5000)
    fake_targets = np.repeat(np.arange(100), 10)
    assert rel(MO.inception_score(fake), gold["is"]) < 1e-5
    assert rel(MO.fid(real, fake), gold["fid"]) < 1e-8
    assert rel(MO.intra_fid(real, real_targets, fake, fake_targets), gold["intra_fid"]) < 1e-8
    assert rel(MO.inception_score(fake[:64]), gold["is64"]) < 1e-5
    assert rel(MO.fid(real, fake[:64]), gold["fid64"]) < 1e-7


def test_selfdivergence_fixture_is_sane():
    """The reference against itself (8 vs 1 thread) is exact for the first steps and chaotic later; the
    fixture is the long-horizon noise floor any implementation is judged against."""
    sd = load_golden("selfdiv")
    assert max(sd["rel_d"][:2]) < 1e-5
    assert max(sd["rel_d"]) > 1e-3
