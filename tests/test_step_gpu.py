"""Whole-step parity on the GPU: the native DCGAN step engine (through the C ABI) against the CPU oracle
(oracle/gan_oracle.py, itself pinned to the reference) on the same weights, batch and noise.

Tolerances (relative):
  JCK_PREC_F32 (fp32 storage, exact-fp32 MFMA v_mfma_f32_16x16x4_f32): the north star's 1e-3 on losses from identical
      state, per step; measured ~1e-6.  Gradients are compared in relative L2 (5e-3) as well as max-norm (3e-2): the
      LeakyReLU/ReLU derivative is discontinuous at 0, so a pre-activation within rounding distance of 0 can take the other
      branch than on the CPU and change a handful of gradient elements by 0.8*g - an effect the reference shows against
      itself once its own summation order changes (tests/golden/selfdiv.json).
  JCK_PREC_BF16 (fast path): 3e-2 on losses per step here; gradients, per tensor, against stated limits that follow from a
      bf16-storage emulation of the oracle - see tests/test_bf16_envelope.py, which also holds the free-running trajectory
      and the run-to-run determinism; tests/test_trainer_gpu.py holds the bf16 trainer's first step to the fixture recorded
      from the reference's own trainer at 1e-2."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _noise(B, seed):
    g = torch.Generator().manual_seed(seed)
    return {"n1": torch.randn(B, 3, 64, 64, generator=g), "z": torch.randn(B, 100, 1, 1, generator=g),
            "n2": torch.randn(B, 3, 64, 64, generator=g), "alpha": torch.rand(B, 1, 1, 1, generator=g)}


def mnist_shaped(n):
    """BASELINE.json configs[0]: MNIST-shaped synthetic input, adapted as SURVEY section 8(d) prescribes - [n,1,28,28] gray
    in [0,1] (seeded), repeated to 3 channels, bilinear resize to 64x64, Normalize(0.5, 0.5); the nets only take 3x64x64."""
    g = torch.Generator().manual_seed(2024)
    x = torch.rand(n, 1, 28, 28, generator=g)
    x = x * (torch.rand(n, 1, 28, 28, generator=g) > 0.6)                # mostly-black strokes-on-background statistics
    x = torch.nn.functional.interpolate(x.repeat(1, 3, 1, 1), size=64, mode="bilinear", align_corners=False)
    return (x - 0.5) / 0.5


def _run(B, steps, prec, lr=2e-4, teacher_forced=True, images=None):
    from hipgan.engine import DcganEngine
    from oracle.gan_oracle import GanOracle
    from util import synth_images
    orc = GanOracle("dcgan", lr=lr, seed=12345)
    eng = DcganEngine(batch=B, prec=prec)
    eng.load_state(orc.g, orc.d)
    imgs = synth_images(B * steps) if images is None else images(B * steps)
    out = []
    for s in range(steps):
        real, nz = imgs[s * B:(s + 1) * B], _noise(B, 100 + s)
        if teacher_forced and s > 0:      # restart the engine from the oracle's state: isolates one step
            eng.load_state(orc.g, orc.d)
            for tag, opt in (("g", orc.opt_g), ("d", orc.opt_d)):
                for what, src in (("m", opt.m), ("v", opt.v)):
                    v = eng.named_views(tag, what)
                    for k, t in src.items():
                        v[k].copy_(t)
            eng.t = orc.opt_d.t
        ref = orc.step(real, None, nz)
        got = eng.step(real.cuda(), {k: v.cuda() for k, v in nz.items()}, lr=lr)
        out.append((ref, got, {k: v.clone() for k, v in orc.d_grads.items()}, {k: v.clone() for k, v in orc.g_grads.items()}))
    return orc, eng, out


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-12)


def _cmp_tensors(views, refs, tol, what, l2tol=None, atol=0.0):
    bad = []
    for k, r in refs.items():
        g = views[k].detach().float().cpu().view(r.shape)
        scale = r.abs().max().item() + 1e-30
        err = (g - r).abs().max().item()
        l2 = ((g - r).norm() / (r.norm() + 1e-30)).item()
        if err > tol * scale + atol or (l2tol is not None and l2 > l2tol):
            bad.append(f"{what}:{k}: max err {err:.3e} vs scale {scale:.3e} (tol {tol}), rel-l2 {l2:.3e} (tol {l2tol})")
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("B", [8, 64])
def test_step_parity_f32(B):
    orc, eng, out = _run(B, 3, "f32")
    for s, (ref, got, dgr, ggr) in enumerate(out):
        for k in ("loss_d", "loss_g", "gp", "loss_real", "loss_fake", "d_x", "d_gz1", "d_gz2"):
            assert _rel(got[k], ref[k]) < 1e-3, (s, k, got[k], ref[k])
    # last step: gradients, weights, Adam state and BN running statistics
    ref, got, dgr, ggr = out[-1]
    _cmp_tensors(eng.named_views("d", "grads"), dgr, 3e-2, "d_grads", 5e-3)
    _cmp_tensors(eng.named_views("g", "grads"), ggr, 3e-2, "g_grads", 5e-3)
    # post-Adam weights: Adam turns a gradient into ~lr*sign(g), so an element whose gradient is within rounding of 0 can
    # move the other way (difference up to 2*lr per step).  Max-norm gets that absolute slack; relative L2 stays tight.
    _cmp_tensors(eng.named_views("d"), {k: v for k, v in orc.d.items() if v.dtype == torch.float32}, 2e-4, "d_state", 2e-3, atol=4e-4)
    _cmp_tensors(eng.named_views("g"), {k: v for k, v in orc.g.items() if v.dtype == torch.float32}, 2e-4, "g_state", 2e-3, atol=4e-4)
    assert int(eng.named_views("d")["norm1.num_batches_tracked"]) == int(orc.d["norm1.num_batches_tracked"])
    assert int(eng.named_views("g")["norm4.num_batches_tracked"]) == int(orc.g["norm4.num_batches_tracked"])
    fake = eng.tensor("fake").view(B, 64, 64, 4)[..., :3].permute(0, 3, 1, 2).float().cpu()
    assert (fake - ref["fake"]).abs().max().item() < 5e-4


def test_step_first_step_tight_f32():
    """Step 0 from identical state: assert 1e-4 (10x under the north star's 1e-3)."""
    orc, eng, out = _run(16, 1, "f32")
    ref, got, dgr, ggr = out[0]
    for k in ("loss_d", "loss_g", "gp"):
        assert _rel(got[k], ref[k]) < 1e-4, (k, got[k], ref[k])


@pytest.mark.parametrize("B", [8, 64])
def test_step_parity_bf16(B):
    orc, eng, out = _run(B, 3, "bf16")
    for s, (ref, got, dgr, ggr) in enumerate(out):
        for k in ("loss_d", "loss_g", "loss_real", "loss_fake"):
            assert _rel(got[k], ref[k]) < 3e-2, (s, k, got[k], ref[k])
        assert _rel(got["gp"], ref["gp"]) < 6e-2, (s, got["gp"], ref["gp"])
    # gradients: tests/test_bf16_envelope.py (per tensor, <= 2x the measured error)


def test_free_running_golden_f32():
    """Free-running (no teacher forcing) against the fixture captured from the reference's own trainer run
    (tests/golden/dcgan_steps.json, B=8, 3 steps, noise from the global CPU generator in the reference's order)."""
    from hipgan.engine import DcganEngine
    from oracle.gan_oracle import build_params
    from util import load_golden, synth_images
    gold = load_golden("dcgan_steps")["B8"]
    B = 8
    torch.manual_seed(12345)
    g, d = build_params("dcgan")
    eng = DcganEngine(batch=B, prec="f32")
    eng.load_state(g, d)
    fixed = torch.randn(64, 100, 1, 1)
    imgs = synth_images(B * 3)
    for s in range(3):
        nz = {"n1": torch.randn(B, 3, 64, 64), "z": torch.randn(B, 100, 1, 1), "n2": torch.randn(B, 3, 64, 64),
              "alpha": torch.rand(B, 1, 1, 1)}
        got = eng.step(imgs[s * B:(s + 1) * B].cuda(), {k: v.cuda() for k, v in nz.items()}, lr=2e-4)
        tol = 1e-3 if s == 0 else 5e-3          # later steps inherit Adam's amplification of rounding noise
        assert _rel(got["loss_d"], gold["losses_d"][s]) < tol, (s, got["loss_d"], gold["losses_d"][s])
        assert _rel(got["loss_g"], gold["losses_g"][s]) < tol * 3, (s, got["loss_g"], gold["losses_g"][s])
        assert _rel(got["gp"], gold["step"][s]["gp"]) < tol * 3
        if s == 0 or s == 2:
            DcganEngine(batch=64, share=eng).sample(fixed.cuda())        # the eval branch moves G's BN running statistics


def test_default_lr_clamp_plateau():
    """lr 0.1 (main.py:54): D saturates after one step; BCELoss's -100 clamp gives loss_g = 10, loss_d = 110."""
    from hipgan.engine import DcganEngine
    from oracle.gan_oracle import build_params
    from util import load_golden, synth_images
    gold = load_golden("dcgan_steps")["B8_lr0.1"]
    B = 8
    torch.manual_seed(12345)
    g, d = build_params("dcgan")
    eng = DcganEngine(batch=B, prec="f32")
    eng.load_state(g, d)
    fixed = torch.randn(64, 100, 1, 1)
    imgs = synth_images(B * 4)
    for s in range(4):
        nz = {"n1": torch.randn(B, 3, 64, 64), "z": torch.randn(B, 100, 1, 1), "n2": torch.randn(B, 3, 64, 64),
              "alpha": torch.rand(B, 1, 1, 1)}
        got = eng.step(imgs[s * B:(s + 1) * B].cuda(), {k: v.cuda() for k, v in nz.items()}, lr=0.1)
        if s == 0:
            DcganEngine(batch=64, share=eng).sample(fixed.cuda())
        assert _rel(got["loss_d"], gold["losses_d"][s]) < 1e-3, (s, got)
        assert _rel(got["loss_g"], gold["losses_g"][s]) < 1e-3, (s, got)


def test_sample_matches_oracle():
    from hipgan.engine import DcganEngine
    from oracle.gan_oracle import GanOracle
    orc = GanOracle("dcgan", seed=12345)
    eng = DcganEngine(batch=16, prec="f32")
    eng.load_state(orc.g, orc.d)
    z = torch.randn(40, 100, 1, 1, generator=torch.Generator().manual_seed(3))
    with pytest.raises(Exception):
        eng.sample(z.cuda())                       # 40 > batch: one BN batch cannot be split silently
    big = DcganEngine(batch=40, share=eng)         # second workspace geometry on the same state
    got = big.sample(z.cuda()).cpu()
    ref = orc.sample(z)
    assert (got - ref).abs().max().item() < 2e-4
    got2 = eng.sample(z[:12].cuda()).cpu()
    assert (got2 - orc.sample(z[:12])).abs().max().item() < 2e-4
    assert int(eng.named_views("g")["norm1.num_batches_tracked"]) == 2


@pytest.mark.parametrize("prec,tol", [("f32", 1e-3), ("bf16", 3e-2)])
def test_config0_mnist_shaped_batch64(prec, tol):
    """BASELINE.json configs[0] (the reference's own CPU-runnable case): batch 64, MNIST-shaped input; per-step losses of the
    HIP path against the oracle (= the reference's trainer arithmetic) from identical state."""
    orc, eng, out = _run(64, 2, prec, images=mnist_shaped)
    for s, (ref, got, dgr, ggr) in enumerate(out):
        for k in ("loss_d", "loss_g", "gp", "loss_real", "loss_fake", "d_x", "d_gz1", "d_gz2"):
            assert _rel(got[k], ref[k]) < tol, (s, k, got[k], ref[k])


@pytest.mark.parametrize("prec,tol", [("f32", 1e-3), ("bf16", 3e-2)])
def test_full_size_step_batch256(prec, tol):
    """BASELINE.json configs[1] at its full size (batch 256: the batched three-pass D schedule, 128x128 LDS-DMA tiles,
    split-K weight gradients at their production shapes): one step against the oracle, plus what must hold at any size -
    D(x), D(G(z)) are means of probabilities, loss_d = loss_real + loss_fake + 10 gp, and BatchNorm saw 4 D and 1 G batches."""
    orc, eng, out = _run(256, 1, prec)
    ref, got, dgr, ggr = out[0]
    for k in ("loss_d", "loss_g", "gp", "loss_real", "loss_fake", "d_x", "d_gz1", "d_gz2"):
        assert _rel(got[k], ref[k]) < tol, (k, got[k], ref[k])
    assert 0.0 < got["d_x"] < 1.0 and 0.0 < got["d_gz1"] < 1.0 and 0.0 < got["d_gz2"] < 1.0
    assert abs(got["loss_d"] - (got["loss_real"] + got["loss_fake"] + 10.0 * got["gp"])) < 1e-4 * max(1.0, abs(got["loss_d"]))
    assert int(eng.named_views("d")["norm1.num_batches_tracked"]) == 4 and int(eng.named_views("g")["norm1.num_batches_tracked"]) == 1
    # D's gradients come from identical weights.  G's come through the D that Adam has just stepped: an element of D whose
    # gradient was within rounding of 0 moved the other way (2*lr), which at batch 256 shows as ~1e-2 in G's gradients.
    if prec == "f32":
        # max-norm bounds = 4x what round 5 measured at this batch (tools/measure_tol.py: D <= 4.9e-3, G <= 1.7e-2 of a tensor's largest
        # element; relative L2: D <= 1.4e-3, G <= 1.7e-2)
        _cmp_tensors(eng.named_views("d", "grads"), dgr, 2e-2, "d_grads", 5e-3)
        _cmp_tensors(eng.named_views("g", "grads"), ggr, 6e-2, "g_grads", 3e-2)
    else:       # per tensor against the fp32 oracle: what bf16 STORAGE costs at this batch (the emulation of the storage format,
                # oracle/bf16_emu.py, is 0.086 / 0.164 away from the fp32 oracle on D's / G's worst tensor; tests/test_bf16_envelope.py
                # holds hip to 1.25x the emulation tensor by tensor)
        lim = {"d": 0.11, "g": 0.21}
        for tag, refs in (("d", dgr), ("g", ggr)):
            views = eng.named_views(tag, "grads")
            for k, r in refs.items():
                l2 = ((views[k].detach().float().cpu().view(r.shape) - r).norm() / (r.norm() + 1e-30)).item()
                assert l2 <= lim[tag], (tag, k, l2)


@pytest.mark.parametrize("env", [{"JCK_BATCHED": "0"}, {"JCK_OVERLAP": "0"}, {"JCK_WGRAD_SIDE": "0"}, {"JCK_BN_RES": "0"}])
def test_alternative_schedules_give_the_same_step(env, monkeypatch):
    """The schedules kept behind environment switches (per-pass D passes with stream overlap - what a batch that is not a
    multiple of 8 runs -, no overlap at all, weight gradients on the main stream, BatchNorm backward as three launches) must
    all be the same arithmetic: one exact-fp32 step of each against the oracle.  The switches are read when an engine is created."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    orc, eng, out = _run(16, 1, "f32")
    ref, got, dgr, ggr = out[0]
    for k in ("loss_d", "loss_g", "gp", "loss_real", "loss_fake", "d_x", "d_gz1", "d_gz2"):
        assert _rel(got[k], ref[k]) < 1e-3, (env, k, got[k], ref[k])
    _cmp_tensors(eng.named_views("d", "grads"), dgr, 3e-2, "d_grads", 5e-3)
    _cmp_tensors(eng.named_views("g", "grads"), ggr, 3e-2, "g_grads", 2e-2)


def test_resident_batchnorm_backward_in_the_step_matches_the_three_launch_form(monkeypatch):
    """bf16, batch 64 (where the resident one-launch BatchNorm backward of csrc/bnres.hpp takes every single-group pass): one step
    with it and one with JCK_BN_RES=0 from the same state and noise.  Same arithmetic per element, another summation order of
    the two per-channel sums: scalars to 2e-3, every gradient tensor to 2e-2 relative L2 (one bf16 ulp of g_y amplified through
    the layers below)."""
    runs = []
    for res in ("1", "0"):
        monkeypatch.setenv("JCK_BN_RES", res)
        orc, eng, out = _run(64, 1, "bf16")
        torch.cuda.synchronize()
        runs.append((out[0][1], {t: {k: v.detach().float().cpu().clone() for k, v in eng.named_views(t, "grads").items()} for t in "dg"}))
    (s1, g1), (s0, g0) = runs
    for k in ("loss_d", "loss_g", "gp", "loss_real", "loss_fake", "d_x", "d_gz1", "d_gz2"):
        assert _rel(s1[k], s0[k]) < 2e-3, (k, s1[k], s0[k])
    for t in "dg":
        for k, r in g0[t].items():
            l2 = ((g1[t][k] - r).norm() / (r.norm() + 1e-30)).item()
            assert l2 < 2e-2, (t, k, l2)


def test_batched_schedule_with_a_non_power_of_two_batch():
    """Batch 24 (ragged last batches are multiples of 8 for CIFAR's 50000 = 195*256 + 80): the batched 3B D pass with groups
    of 24 images - tiles, statistic slots and BatchNorm groups must still line up."""
    orc, eng, out = _run(24, 2, "f32")
    for s, (ref, got, dgr, ggr) in enumerate(out):
        for k in ("loss_d", "loss_g", "gp", "loss_real", "loss_fake", "d_x", "d_gz1", "d_gz2"):
            assert _rel(got[k], ref[k]) < 1e-3, (s, k, got[k], ref[k])
    ref, got, dgr, ggr = out[-1]
    _cmp_tensors(eng.named_views("d", "grads"), dgr, 3e-2, "d_grads", 5e-3)


def test_per_pass_schedule_at_the_8gpu_tail_batch():
    """Batch 106 = what every rank of an 8-GPU run gets as the last batch of a CIFAR epoch (6250 % 256): not a multiple of 8,
    so the per-pass schedule with stream overlap runs, with tiles that end inside an image."""
    orc, eng, out = _run(106, 1, "f32")
    ref, got, dgr, ggr = out[0]
    for k in ("loss_d", "loss_g", "gp", "loss_real", "loss_fake", "d_x", "d_gz1", "d_gz2"):
        assert _rel(got[k], ref[k]) < 1e-3, (k, got[k], ref[k])
    _cmp_tensors(eng.named_views("d", "grads"), dgr, 3e-2, "d_grads", 5e-3)
    _cmp_tensors(eng.named_views("g", "grads"), ggr, 3e-2, "g_grads", 2e-2)


@pytest.mark.parametrize("family,B,inkernel,env", [("dcgan", 64, False, {}), ("dcgan", 64, True, {}), ("cgan", 32, True, {}), ("dcgan", 24, False, {}),
                                                   ("dcgan", 40, True, {"JCK_BATCHED": "0"})])
def test_launches_folded_into_their_neighbours_leave_the_step_bit_for_bit(family, B, inkernel, env, monkeypatch):
    """Round 4 folds small launches into their neighbours: D.zero_grad() into the step's set-step launch, G.zero_grad() into D's
    Adam launch, the tanh + noise-mix backward of G's output into the epilogue of D.conv1's input gradient (on the value as the
    separate launch read it: rounded to bf16 first), conv5's input gradient into the launch that forms the logits (CGAN: the middle
    of its head as one launch).  JCK_FOLD_ZERO=0 / JCK_FUSE_TANH=0 / JCK_HEAD_FUSE=0 keep the launches.  Three bf16 steps each
    way, with explicit noise tensors and with the instance noise drawn inside the kernels (as the trainers and bench.py run):
    bit-identical weights, moments, gradients and scalars (batched schedule, CGAN, per-pass schedule)."""
    import bf16_error as be
    from hipgan.engine import CganEngine, DcganEngine
    from oracle.gan_oracle import build_params
    from util import synth_images
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    torch.manual_seed(12345)
    g, d = build_params(family)
    imgs = synth_images(B * 3)
    runs = []
    for plain in (False, True):
        for k in ("JCK_FUSE_TANH", "JCK_FOLD_ZERO", "JCK_HEAD_FUSE"):
            monkeypatch.setenv(k, "0" if plain else "1")
        eng = (CganEngine if family == "cgan" else DcganEngine)(batch=B, prec="bf16")
        eng.load_state(g, d)
        eng.set_noise_seed(4242)
        sc = []
        for s in range(3):
            lab = be.labels_for(B, 5 + s) if family == "cgan" else None
            real = imgs[s * B:(s + 1) * B].cuda()
            if inkernel:
                eng.step_async(real, None, 2e-4, labels=lab.cuda() if lab is not None else None)
                sc.append(eng.scalars())
            else:
                nz = be.noise_for(family, B, 40 + s, lab)
                sc.append(eng.step(real, {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in nz.items()}, lr=2e-4))
        torch.cuda.synchronize()
        runs.append((sc, {k: v.clone() for k, v in eng.arenas.items()}, eng.tensor("fake").clone()))
    (s0, a0, f0), (s1, a1, f1) = runs
    assert s0 == s1, (s0, s1)
    assert torch.equal(f0, f1)
    for k in a0:
        assert torch.equal(a0[k], a1[k]), k


@pytest.mark.parametrize("B,prec", [(64, "bf16"), (16, "f32"), (256, "bf16")])
def test_real_forward_beside_the_generator_is_the_3b_forward(B, prec, monkeypatch):
    """The batched D pass runs D(real)'s input transform and conv stack on the weight-gradient stream beside G's forward and
    [fake | penalty] as one 2B forward behind it (csrc/engine.hip, JCK_REAL_SIDE; the split PHASE_D_REAL_FWD makes across steps).
    BatchNorm is per batch either way and every tile stays inside its batch; what may differ from the one 3B forward
    (JCK_REAL_SIDE=0) is how a batch's statistics are split into partial sums (the launch geometry).  One step of each against
    the fp32 oracle: exact-fp32 both inside the parity tolerances and 1e-4 from each other; bf16 tensor by tensor no further
    from the oracle than the 3B form, and closer to it than either is to the oracle (a rounding-boundary flip of a stored
    activation amplified through the layers below - G.conv1's gradient moves by 3.8 % at batch 256, 0.12 being the storage
    format's own distance there, tests/test_bf16_envelope.py)."""
    runs = []
    for side in ("1", "0"):
        monkeypatch.setenv("JCK_REAL_SIDE", side)
        orc, eng, out = _run(B, 1, prec)
        torch.cuda.synchronize()
        runs.append((out[-1][1], {t: {k: v.detach().float().cpu().clone() for k, v in eng.named_views(t, "grads").items()} for t in "dg"}))
    ref_s, ref = out[-1][0], {"d": out[-1][2], "g": out[-1][3]}
    (s1, g1), (s0, g0) = runs
    stol = 1e-5 if prec == "f32" else 2e-3
    for k in ("loss_d", "loss_g", "gp", "loss_real", "loss_fake", "d_x", "d_gz1", "d_gz2"):
        assert _rel(s1[k], s0[k]) < stol, (k, s1[k], s0[k])
        if prec == "f32":
            assert _rel(s1[k], ref_s[k]) < 1e-3, (k, s1[k], ref_s[k])
    l2 = lambda a, b: ((a - b).norm() / (b.norm() + 1e-30)).item()
    for t in "dg":
        for k, r in ref[t].items():
            d1, d0, d10 = l2(g1[t][k].view(r.shape), r), l2(g0[t][k].view(r.shape), r), l2(g1[t][k], g0[t][k])
            if prec == "f32":
                assert d10 < 1e-4 and d1 < 2e-2, (t, k, d10, d1)
            else:
                assert d1 <= 1.25 * d0 + 1e-3, (t, k, d1, d0)
                assert d10 <= max(2e-2, 0.75 * max(d0, d1)), (t, k, d10, d0, d1)


@pytest.mark.parametrize("B,prec", [(32, "bf16"), (8, "f32")])
def test_real_forward_beside_the_generator_cgan(B, prec, monkeypatch):
    """The same schedule in CGAN's batched pass (its head runs once over the 3B rows of the concat buffer, which the two forwards
    fill in two pieces), with the instance noise drawn inside the kernels: one step from the same state, scalars and gradients
    to summation-order accuracy (exact-fp32 1e-5 / 1e-4; bf16 2e-3 / 6e-2, the double backward amplifies more)."""
    import bf16_error as be
    from hipgan.engine import CganEngine
    from oracle.gan_oracle import build_params
    from util import synth_images
    torch.manual_seed(12345)
    g, d = build_params("cgan")
    imgs = synth_images(B)
    runs = []
    for side in ("1", "0"):
        monkeypatch.setenv("JCK_REAL_SIDE", side)
        eng = CganEngine(batch=B, prec=prec)
        eng.load_state(g, d)
        eng.set_noise_seed(999)
        eng.step_async(imgs.cuda(), None, 2e-4, labels=be.labels_for(B, 5).cuda())
        sc = eng.scalars()
        torch.cuda.synchronize()
        runs.append((sc, {t: {k: v.detach().float().cpu().clone() for k, v in eng.named_views(t, "grads").items()} for t in "dg"},
                     eng.tensor("fake").clone()))
    (s1, g1, f1), (s0, g0, f0) = runs
    assert torch.equal(f0, f1)                       # G's forward is untouched
    stol, gtol = (1e-5, 1e-4) if prec == "f32" else (2e-3, 6e-2)
    for k in s0:
        assert _rel(s1[k], s0[k]) < stol, (k, s1[k], s0[k])
    for t in "dg":
        for k, r in g0[t].items():
            l2 = ((g1[t][k] - r).norm() / (r.norm() + 1e-30)).item()
            assert l2 < gtol, (t, k, l2)


def test_per_pass_schedule_in_bf16_keeps_one_resident_launch_in_flight(monkeypatch):
    """bf16 at batch 106: the per-pass schedule runs the penalty pass on its own stream beside D(fake).  Two resident BatchNorm
    backward launches at once would share the engine's barrier words (and could starve each other of CUs), so that pass takes the
    three-launch form: the barrier never times out (engine.scalars() raises if one did), and tensor by tensor the step is no
    further from the fp32 oracle than the all three-launch step (JCK_BN_RES=0) and closer to that step than either is to the
    oracle (another summation order of the per-channel sums, amplified by bf16 rounding through the layers below)."""
    runs = []
    for res in ("1", "0"):
        monkeypatch.setenv("JCK_BN_RES", res)
        orc, eng, out = _run(106, 1, "bf16")
        torch.cuda.synchronize()
        runs.append((out[-1][1], {t: {k: v.detach().float().cpu().clone() for k, v in eng.named_views(t, "grads").items()} for t in "dg"}))
    ref = {"d": out[-1][2], "g": out[-1][3]}
    (s1, g1), (s0, g0) = runs
    for k in ("loss_d", "loss_g", "gp", "loss_real", "loss_fake", "d_x", "d_gz1", "d_gz2"):
        assert _rel(s1[k], s0[k]) < 2e-3, (k, s1[k], s0[k])
    l2 = lambda a, b: ((a - b).norm() / (b.norm() + 1e-30)).item()
    for t in "dg":
        for k, r in ref[t].items():
            d1, d0, d10 = l2(g1[t][k].view(r.shape), r), l2(g0[t][k].view(r.shape), r), l2(g1[t][k], g0[t][k])
            assert d1 <= 1.25 * d0 + 1e-3, (t, k, d1, d0)
            assert d10 <= max(2e-2, 0.75 * max(d0, d1)), (t, k, d10, d0, d1)


def test_separate_generator_learning_rate_keeps_the_step_scalars():
    """an optimiser phase called with another learning rate than the step's loss phases used to re-run
    jck_engine_set_step, which zeroed the step's accumulator rows - loss_d / loss_g / gp were then logged as 0.  Two engines from
    one state and one noise draw, lr 2e-4 for both nets vs lr_g 1e-4: the logged scalars and D's update must be bitwise equal,
    and G's first Adam step (lr * m_hat / (sqrt(v_hat) + eps): proportional to lr) exactly half as large."""
    from hipgan.engine import DcganEngine
    from oracle.gan_oracle import GanOracle
    from util import synth_images
    B = 16
    orc = GanOracle("dcgan", lr=2e-4, seed=12345)
    real, nz = synth_images(B).cuda(), {k: v.cuda() for k, v in _noise(B, 7).items()}
    res = []
    for lr_g in (None, 1e-4):
        eng = DcganEngine(batch=B, prec="f32")
        eng.load_state(orc.g, orc.d)
        g0 = {k: v.detach().clone() for k, v in eng.named_views("g").items() if k.endswith("weight")}
        eng.step_async(real, nz, lr=2e-4, lr_g=lr_g)
        sc = eng.scalars()
        res.append((sc, {k: eng.named_views("g")[k].detach() - v for k, v in g0.items()},
                    {k: v.detach().clone() for k, v in eng.named_views("d").items()}))
    (s0, dg0, d0), (s1, dg1, d1) = res
    for k in ("loss_d", "loss_g", "gp", "loss_real", "loss_fake", "d_x", "d_gz1", "d_gz2"):
        assert s1[k] == s0[k] and (s1[k] != 0.0 or k == "gp"), (k, s0[k], s1[k])
    assert s1["loss_d"] > 0.1 and s1["loss_g"] > 0.1
    for k in d0:
        assert torch.equal(d0[k], d1[k]), k
    for k in dg0:
        a, b = dg0[k].float(), dg1[k].float()
        assert a.abs().max() > 0
        assert torch.allclose(b, 0.5 * a, rtol=1e-3, atol=2e-9), (k, (b - 0.5 * a).abs().max().item(), a.abs().max().item())
