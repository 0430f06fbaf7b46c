"""The bf16 fast path (the one bench.py times) held to MEASURED error, not to a guessed tolerance.

* per step: every gradient tensor and logged scalar of one teacher-forced step is within 2x the distance recorded in
  tests/golden/bf16_error_table.json (made on a MI355X by tests/golden/make_bf16_error_table.py; table in
  profiles/r02_bf16_error_table.md) - against the fp32 oracle (pinned to the reference) AND against the oracle with bf16
  storage emulation (oracle/bf16_emu.py), which removes the storage format from the comparison;
* envelope: the distance hip <-> fp32 oracle is explained by bf16 storage: it stays within 2x the distance emulation <->
  fp32 oracle measured in the same test, per tensor and along a 30-step free-running trajectory;
* determinism: no float atomics on the path - two runs of the same steps give bitwise identical weights and scalars."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [("dcgan", 8, 2), ("dcgan", 64, 2), ("dcgan", 256, 1), ("cgan", 8, 2), ("cgan", 64, 2), ("cgan", 256, 1)]
FLOOR = {"scalars": 2e-4, "d_grads": 1e-3, "g_grads": 1e-3}      # absolute slack under which a distance is rounding noise


def _table(family, B):
    from util import load_golden
    return load_golden("bf16_error_table")["cases"][f"{family}_B{B}"]


def limit(family, B, group, key, metric="hip_vs_ref"):
    """2x the recorded distance (+ a floor for distances that are rounding noise)."""
    return 2.0 * _table(family, B)[group][key][metric] + FLOOR[group]


@pytest.mark.parametrize("family,B,steps", CASES)
def test_bf16_step_within_twice_the_measured_error(family, B, steps):
    import bf16_error as be
    rows = be.measure(family, B, steps)
    tab = _table(family, B)
    bad = []
    for group in ("scalars", "d_grads", "g_grads"):
        for k in rows[0][group]:
            for m in ("hip_vs_ref", "hip_vs_emu"):
                if m not in rows[0][group][k]:
                    continue
                got = max(r[group][k][m] for r in rows)
                lim = 2.0 * tab[group][k][m] + FLOOR[group]
                if not got <= lim:
                    bad.append(f"{group}:{k}:{m}: {got:.3e} > 2 x {tab[group][k][m]:.3e}")
    assert not bad, "\n".join(bad)
    if family == "dcgan":
        # the storage format explains the distance from the fp32 oracle: tensor by tensor hip is no further from it than the
        # emulation is (measured: equal within a few %), and with the format matched hip and the emulation are closer to each
        # other than either is to the fp32 oracle.  What is left between them is summation order: a 1e-6 relative perturbation
        # of D's conv outputs moves the EMULATION's own gradients by ~2e-2 (bf16 rounding-boundary flips).
        for group in ("d_grads", "g_grads"):
            for k in rows[0][group]:
                h = max(r[group][k]["hip_vs_ref"] for r in rows)
                e = max(r[group][k]["emu_vs_ref"] for r in rows)
                assert h <= 1.25 * e + FLOOR[group], (group, k, h, e)
        for group in ("d_grads", "g_grads"):
            assert be.worst(rows, group, "hip_vs_emu") <= 0.75 * be.worst(rows, group, "hip_vs_ref") + FLOOR[group], group


@pytest.mark.parametrize("family,prec,B", [("dcgan", "bf16", 16), ("dcgan", "f32", 16), ("dcgan", "bf16", 256), ("cgan", "bf16", 16)])
def test_two_runs_are_bitwise_identical(family, prec, B):
    """Same weights, batch and noise twice, two steps each: gradients, weights, Adam moments, BN statistics and the logged
    scalars must be bit-for-bit equal (the reductions that used float atomics in round 1 are two-stage sums now)."""
    import bf16_error as be
    from hipgan.engine import CganEngine, DcganEngine
    from oracle.gan_oracle import build_params
    from util import synth_images
    torch.manual_seed(12345)
    g, d = build_params(family)
    imgs = synth_images(B * 2)
    runs = []
    for _ in range(2):
        eng = (CganEngine if family == "cgan" else DcganEngine)(batch=B, prec=prec)
        eng.load_state(g, d)
        sc = []
        for s in range(2):
            lab = be.labels_for(B, 5 + s) if family == "cgan" else None
            nz = be.noise_for(family, B, 40 + s, lab)
            sc.append(eng.step(imgs[s * B:(s + 1) * B].cuda(), {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in nz.items()}, lr=2e-4))
        torch.cuda.synchronize()
        runs.append((sc, {k: v.clone() for k, v in eng.arenas.items()}))
    (s0, a0), (s1, a1) = runs
    assert s0 == s1, (s0, s1)
    for k in a0:
        assert torch.equal(a0[k], a1[k]), k


def test_bf16_trajectory_inside_the_storage_envelope():
    """30 free-running steps at B=64.  A GAN trajectory is chaotic (the reference diverges from ITSELF by 2.9 % in loss_d over
    steps 10..29 when only its thread count changes, tests/golden/selfdiv.json), so a per-step tolerance is meaningless after
    a few steps.  What can be asserted: the bf16 engine's divergence from the fp32 oracle is the divergence that bf16 STORAGE
    causes - it stays within 2x the divergence of the bf16-storage emulation (same oracle code, CPU, fp32 accumulate) from the
    fp32 oracle over the same steps, it tracks step by step while the trajectories are close, and the loss level stays."""
    from hipgan.engine import DcganEngine
    from oracle.gan_oracle import GanOracle
    import bf16_error as be
    from util import synth_images
    B, steps = 64, 30
    ref = GanOracle("dcgan", lr=2e-4, seed=12345)
    emu = GanOracle("dcgan", lr=2e-4, seed=12345, emulate_bf16=True)
    eng = DcganEngine(batch=B, prec="bf16")
    eng.load_state(ref.g, ref.d)
    imgs = synth_images(B * 4)
    rd_h, rd_e, ld_ref, ld_got = [], [], [], []
    for s in range(steps):
        real, nz = imgs[(s % 4) * B:(s % 4 + 1) * B], be.noise_for("dcgan", B, 500 + s)
        r = ref.step(real, None, nz)
        e = emu.step(real, None, nz)
        h = eng.step(real.cuda(), {k: v.cuda() for k, v in nz.items()}, lr=2e-4)
        rd_h.append(be.rel(h["loss_d"], r["loss_d"]))
        rd_e.append(be.rel(e["loss_d"], r["loss_d"]))
        ld_ref.append(r["loss_d"])
        ld_got.append(h["loss_d"])
        assert h["loss_d"] == h["loss_d"] and h["loss_g"] == h["loss_g"]     # no NaN
    assert max(rd_h[:5]) < 3e-2, rd_h[:5]
    assert max(rd_h[:5]) <= 2.0 * max(rd_e[:5]) + 2e-3, (rd_h[:5], rd_e[:5])
    mh, me = sum(rd_h[10:30]) / 20, sum(rd_e[10:30]) / 20
    assert mh <= 2.0 * me + 1e-2, (mh, me, rd_h, rd_e)
    m_ref, m_got = sum(ld_ref[10:30]) / 20, sum(ld_got[10:30]) / 20
    assert abs(m_got - m_ref) < 0.15 * abs(m_ref), (m_got, m_ref)
