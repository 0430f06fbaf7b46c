"""CGAN step on the GPU (engine family 1) against the CPU oracle with identical weights, batch, labels, noise and dropout
masks: label-embedding concat, Linear head, Dropout, and the BACK-PROPAGATED gradient penalty (double backward through conv,
train-mode BatchNorm, LeakyReLU, Linear, Dropout, Sigmoid - train/cgan_trainer.py:200-203).  Tolerances as in
tests/test_step_gpu.py (losses 1e-3 per step from identical state on the exact-fp32 path)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _noise(B, seed, labels):
    g = torch.Generator().manual_seed(seed)
    nz = {"n1": torch.randn(B, 3, 64, 64, generator=g), "z": torch.randn(B, 100, 1, 1, generator=g),
          "n2": torch.randn(B, 3, 64, 64, generator=g), "alpha": torch.rand(B, 1, 1, 1, generator=g), "labels": labels}
    for i in range(4):
        nz[f"m{i + 1}"] = (torch.rand(B, 256, generator=g) >= 0.25).float()
    return nz


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-12)


def _cmp(views, refs, tol_l2, what):
    bad = []
    for k, r in refs.items():
        g = views[k].detach().float().cpu().view(r.shape)
        l2 = ((g - r).norm() / (r.norm() + 1e-30)).item()
        mx = ((g - r).abs().max() / (r.abs().max() + 1e-30)).item()
        if l2 > tol_l2:
            bad.append(f"{what}:{k}: rel-l2 {l2:.3e} max/max {mx:.3e}")
    assert not bad, "\\n".join(bad)


@pytest.mark.parametrize("prec,B,tol,gtol", [("f32", 8, 1e-3, 2e-2), ("f32", 32, 1e-3, 5e-3)])
def test_cgan_step_parity(prec, B, tol, gtol):
    """(the bf16 path: tests/test_bf16_envelope.py, per tensor within 2x the measured error at B = 8 / 64 / 256)"""
    from hipgan.engine import CganEngine
    from oracle.gan_oracle import GanOracle
    from util import synth_images, synth_onehot
    orc = GanOracle("cgan", lr=2e-4, seed=12345)
    eng = CganEngine(batch=B, prec=prec)
    eng.load_state(orc.g, orc.d)
    imgs = synth_images(B * 2)
    onehot, _ = synth_onehot(B * 2)
    for s in range(2):
        real, lab = imgs[s * B:(s + 1) * B], onehot[s * B:(s + 1) * B]
        nz = _noise(B, 300 + s, lab)
        if s > 0:       # teacher forcing: restart from the oracle's state
            eng.load_state(orc.g, orc.d)
            for tag, opt in (("g", orc.opt_g), ("d", orc.opt_d)):
                for what, src in (("m", opt.m), ("v", opt.v)):
                    v = eng.named_views(tag, what)
                    for k, t in src.items():
                        v[k].copy_(t.view(v[k].shape))
            eng.t = orc.opt_d.t
        ref = orc.step(real, lab, nz)
        got = eng.step(real.cuda(), {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in nz.items()}, lr=2e-4)
        for k in ("loss_real", "loss_fake", "gp", "loss_d", "loss_g", "d_x", "d_gz1", "d_gz2"):
            assert _rel(got[k], ref[k]) < tol, (s, k, got[k], ref[k])
        _cmp(eng.named_views("d", "grads"), orc.d_grads, gtol, f"s{s}.d_grads")
        _cmp(eng.named_views("g", "grads"), orc.g_grads, gtol, f"s{s}.g_grads")


def test_cgan_state_layout_matches_reference_keys():
    from hipgan.engine import CganEngine
    from util import load_golden
    gold = load_golden("cgan_steps")["B8"]
    eng = CganEngine(batch=4, prec="f32")
    gs, ds = eng.state_dicts()
    assert list(gs.keys()) == gold["ckpt_g_keys"] and list(ds.keys()) == gold["ckpt_d_keys"]
    assert ds["linear1.weight"].shape == (256, 8392) and ds["label_embedding.weight"].shape == (200, 100)
    assert gs["conv1.weight"].shape == (200, 512, 4, 4)


@pytest.mark.parametrize("prec,tol,gtol", [("f32", 1e-3, 3e-2)])
def test_cgan_full_size_step_batch256(prec, tol, gtol):
    """BASELINE.json configs[3] at its full size (batch 256, 10-class one-hot labels in the 100-wide encoding): one step against
    the oracle - production tile shapes, split-K Linear(8392,256), wave-specialised kernels, the double backward at scale."""
    from hipgan.engine import CganEngine
    from oracle.gan_oracle import GanOracle
    from util import synth_images
    B = 256
    orc = GanOracle("cgan", lr=2e-4, seed=12345)
    eng = CganEngine(batch=B, prec=prec)
    eng.load_state(orc.g, orc.d)
    real = synth_images(B)
    g = torch.Generator().manual_seed(77)
    lab = torch.nn.functional.one_hot(torch.randint(0, 10, (B,), generator=g), 100).to(torch.int64)
    nz = _noise(B, 900, lab)
    ref = orc.step(real, lab, nz)
    got = eng.step(real.cuda(), {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in nz.items()}, lr=2e-4)
    for k in ("loss_real", "loss_fake", "gp", "loss_d", "loss_g", "d_x", "d_gz1", "d_gz2"):
        assert _rel(got[k], ref[k]) < tol, (k, got[k], ref[k])
    _cmp(eng.named_views("d", "grads"), orc.d_grads, gtol, "d_grads")
    # G's gradients come through the D that Adam has just stepped (tests/test_step_gpu.py::test_full_size_step_batch256): measured
    # relative L2 <= 1.5e-2 per tensor at this batch (tools/measure_tol.py, round 5); D's are <= 5e-4
    _cmp(eng.named_views("g", "grads"), orc.g_grads, gtol, "g_grads")


def test_cgan_per_pass_schedule_still_matches(monkeypatch):
    """JCK_BATCHED=0 keeps the separate real / fake passes (also what batches that are not a multiple of 8 run)."""
    monkeypatch.setenv("JCK_BATCHED", "0")
    test_cgan_step_parity("f32", 8, 1e-3, 2e-2)


@pytest.mark.parametrize("prec,B", [("bf16", 256), ("f32", 16)])
def test_lazy_join_with_the_weight_gradient_stream_is_bitwise_the_plain_order(prec, B):
    """JCK_PHASE_LAZY_JOIN (include/jckgan.h; the default of a single-GPU CGAN step): D's loss and penalty phases return
    without waiting for the weight-gradient stream and the optimiser phase takes the bottom conv weight last.  Same kernels
    on the same values: three steps with and without it leave bit-identical weights, Adam moments, gradients and scalars
    (a missing dependency shows here as a difference - the small exact-fp32 case has the shortest kernels)."""
    from hipgan.engine import CganEngine
    from oracle.gan_oracle import build_params
    from util import synth_images, synth_onehot
    torch.manual_seed(12345)
    g, d = build_params("cgan")
    imgs = synth_images(B * 3)
    onehot, _ = synth_onehot(B * 3)
    runs = []
    for lazy in (True, False):
        eng = CganEngine(batch=B, prec=prec)
        eng.lazy_join = lazy
        eng.load_state(g, d)
        sc = []
        for s in range(3):
            lab = onehot[s * B:(s + 1) * B]
            nz = _noise(B, 700 + s, lab)
            sc.append(eng.step(imgs[s * B:(s + 1) * B].cuda(), {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in nz.items()}, lr=2e-4))
        torch.cuda.synchronize()
        runs.append((sc, {k: v.clone() for k, v in eng.arenas.items()}))
    (s0, a0), (s1, a1) = runs
    assert s0 == s1, (s0, s1)
    for k in a0:
        assert torch.equal(a0[k], a1[k]), k


@pytest.mark.parametrize("env", [{"JCK_WGRAD_SIDE": "0"}, {"JCK_EXT_EVENTS": "0"}, {"JCK_OVERLAP": "0"}, {"JCK_HEAD_SIDE": "0"},
                                 {"JCK_REAL_SIDE": "0"}, {"JCK_CBUF_DIRECT": "0"}, {"JCK_LAZY_JOIN": "0"}, {"JCK_HEAD_FUSE": "0"}])
def test_cgan_alternative_stream_layouts_give_the_same_step(env, monkeypatch):
    """Every switch that moves CGAN work between the two streams (or keeps a copy / a join the default drops) is the same
    arithmetic: one exact-fp32 step of each against the oracle."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    test_cgan_step_parity("f32", 8, 1e-3, 2e-2)


@pytest.mark.parametrize("prec,B", [("bf16", 64), ("f32", 16)])
def test_head_middle_in_one_launch_is_bitwise_the_four(prec, B, monkeypatch):
    """csrc/ew.hpp: cg_head_mid_kernel = linear_finish + head_fwd + head_dgrad + dropout backward of CGAN's head (JCK_HEAD_FUSE=0
    keeps the four launches), used by the batched D pass (3 groups) and by G's loss pass.  Same arithmetic in the same order, every
    intermediate rounded where the separate launches stored it: two steps leave bit-identical weights, moments, gradients, scalars."""
    from hipgan.engine import CganEngine
    from oracle.gan_oracle import build_params
    from util import synth_images, synth_onehot
    torch.manual_seed(12345)
    g, d = build_params("cgan")
    imgs = synth_images(B * 2)
    onehot, _ = synth_onehot(B * 2)
    runs = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("JCK_HEAD_FUSE", fuse)
        eng = CganEngine(batch=B, prec=prec)
        eng.load_state(g, d)
        sc = []
        for s in range(2):
            lab = onehot[s * B:(s + 1) * B]
            nz = _noise(B, 900 + s, lab)
            sc.append(eng.step(imgs[s * B:(s + 1) * B].cuda(), {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in nz.items()}, lr=2e-4))
        torch.cuda.synchronize()
        runs.append((sc, {k: v.clone() for k, v in eng.arenas.items()}))
    (s0, a0), (s1, a1) = runs
    assert s0 == s1, (s0, s1)
    for k in a0:
        assert torch.equal(a0[k], a1[k]), k
