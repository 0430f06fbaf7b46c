"""The bf16 fast path (the one bench.py times) held to the ORACLE, not to numbers recorded from itself.

* per step (teacher-forced, identical state): three runs - hip (engine, bf16), ref (fp32 CPU oracle, pinned to the reference)
  and emu (the same oracle with a rounding to bf16 wherever the HIP path stores bf16, oracle/bf16_emu.py).  Gated:
    - hip vs emu, per gradient tensor (relative L2) and per logged scalar: stated limits below.  With the storage format
      matched, what is left is summation order amplified by bf16 rounding-boundary flips (a 1e-6 relative perturbation of the
      EMULATION's own conv outputs moves its D gradients by ~2e-2, measured with the CPU oracle alone);
    - hip vs ref <= 1.25 x emu vs ref per tensor: the fast path's distance from the reference IS the storage format's;
    - CGAN goes through the same gates: its emulation differentiates twice through the rounding (round 4).
  (Round 2 gated every tensor at 2x a table recorded from the HIP path - a regression gate, not parity; the table is still
  produced by tests/golden/make_bf16_error_table.py as a report, profiles/r03_bf16_error_table.md, and gates nothing.)
* envelope along a 30-step free-running trajectory: within 2x the emulation's own divergence from the fp32 oracle;
* determinism: no float atomics on the path - two runs of the same steps give bitwise identical weights and scalars."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [("dcgan", 8, 2), ("dcgan", 64, 2), ("dcgan", 256, 1), ("cgan", 8, 2), ("cgan", 64, 2), ("cgan", 256, 1)]
FLOOR = {"scalars": 2e-4, "d_grads": 1e-3, "g_grads": 1e-3}      # absolute slack under which a distance is rounding noise
# hip vs emu (both families): relative L2 per gradient tensor / relative per scalar.  The flips thin out as the batch grows.
LIM_EMU = {8: {"d_grads": 0.11, "g_grads": 0.16, "scalars": 3.0e-2},
           64: {"d_grads": 0.08, "g_grads": 0.14, "scalars": 2.0e-2},
           256: {"d_grads": 0.05, "g_grads": 0.12, "scalars": 5.0e-3}}


@pytest.mark.parametrize("family,B,steps", CASES)
def test_bf16_step_against_the_storage_emulation(family, B, steps):
    """Both families (round 4: the CGAN emulation is written with twice-differentiable rounding primitives, so its back-propagated
    penalty goes through the same storage points - oracle/bf16_emu.py; round 3 held CGAN's gradients to limits recorded from the
    HIP path itself)."""
    import bf16_error as be
    rows = be.measure(family, B, steps)
    bad = []
    lim = LIM_EMU[B]
    for group in ("scalars", "d_grads", "g_grads"):
        for k in rows[0][group]:
            got = max(r[group][k]["hip_vs_emu"] for r in rows)
            if not got <= lim[group] + FLOOR[group]:
                bad.append(f"{group}:{k}:hip_vs_emu: {got:.3e} > {lim[group]:.3e}")
    assert not bad, "\n".join(bad)
    # the storage format explains the distance from the fp32 oracle: tensor by tensor hip is no further from it than the
    # emulation is (measured: equal within a few %), and with the format matched hip and the emulation are closer to each
    # other than either is to the fp32 oracle
    for group in ("d_grads", "g_grads"):
        for k in rows[0][group]:
            h = max(r[group][k]["hip_vs_ref"] for r in rows)
            e = max(r[group][k]["emu_vs_ref"] for r in rows)
            assert h <= 1.25 * e + FLOOR[group], (group, k, h, e)
    for group in ("d_grads", "g_grads"):
        assert be.worst(rows, group, "hip_vs_emu") <= 0.75 * be.worst(rows, group, "hip_vs_ref") + FLOOR[group], group


@pytest.mark.parametrize("family,prec,B", [("dcgan", "bf16", 16), ("dcgan", "f32", 16), ("dcgan", "bf16", 256), ("cgan", "bf16", 16),
                                            ("cgan", "f32", 8)])
def test_two_runs_are_bitwise_identical(family, prec, B):
    """Same weights, batch and noise twice, two steps each: gradients, weights, Adam moments, BN statistics and the logged
    scalars must be bit-for-bit equal (the reductions that used float atomics in round 1 are two-stage sums now).  The small
    exact-fp32 cases have the shortest kernels: they are the ones that show a missing cross-stream dependency (round 3: an
    image-gradient launch moved to the second stream raced with the optimiser's repack here, one run in three)."""
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
