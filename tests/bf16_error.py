"""Measured distance of the bf16 fast path from the fp32 oracle, tensor by tensor.

Used by `tests/golden/make_bf16_error_table.py` (records the table on a MI355X) and by `tests/test_bf16_envelope.py` (holds
the path to <= 2x the recorded values).  Since round 2 the step has no float atomics, so the same inputs give the same bits
on every run and every MI355X: the recorded numbers reproduce exactly and the 2x is pure slack for toolchain changes.

Three runs of one step from IDENTICAL state (weights, Adam moments, BN statistics, batch, noise):
    ref  fp32 CPU oracle (oracle/gan_oracle.py, pinned to the reference)
    emu  the same oracle with bf16 rounding wherever the HIP path stores bf16 (oracle/bf16_emu.py)
    hip  the engine, JCK_PREC_BF16
and three distances per gradient tensor (relative L2) and per logged scalar (relative):
    hip_vs_ref   what the fast path costs in accuracy            (asserted <= 2x table)
    emu_vs_ref   what bf16 storage costs by construction         (the envelope: hip_vs_ref must be explained by it)
    hip_vs_emu   what is left once the storage format is matched (summation order, activation-sign flips)
"""
import torch

SCALARS = ("loss_d", "loss_g", "gp", "loss_real", "loss_fake", "d_x", "d_gz1", "d_gz2")


def noise_for(family, B, seed, labels=None, size=64):
    g = torch.Generator().manual_seed(seed)
    nz = {"n1": torch.randn(B, 3, size, size, generator=g), "z": torch.randn(B, 100, 1, 1, generator=g),
          "n2": torch.randn(B, 3, size, size, generator=g), "alpha": torch.rand(B, 1, 1, 1, generator=g)}
    if family == "cgan":
        nz["labels"] = labels
        for i in range(4):
            nz[f"m{i + 1}"] = (torch.rand(B, 256, generator=g) >= 0.25).float()
    return nz


def labels_for(B, seed=77, classes=10):
    g = torch.Generator().manual_seed(seed)
    return torch.nn.functional.one_hot(torch.randint(0, classes, (B,), generator=g), 100).to(torch.int64)


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-12)


def rel_l2(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _force_oracle(dst, src):
    """dst oracle := src oracle's state (weights, BN buffers, Adam moments and step count), in place."""
    with torch.no_grad():
        for a, b in ((dst.g, src.g), (dst.d, src.d)):
            for k in a:
                a[k].copy_(b[k])
        for od, os_ in ((dst.opt_g, src.opt_g), (dst.opt_d, src.opt_d)):
            for k in od.m:
                od.m[k].copy_(os_.m[k])
                od.v[k].copy_(os_.v[k])
            od.t = os_.t


def _force_engine(eng, orc):
    eng.load_state(orc.g, orc.d)
    for tag, opt in (("g", orc.opt_g), ("d", orc.opt_d)):
        for what, src in (("m", opt.m), ("v", opt.v)):
            v = eng.named_views(tag, what)
            for k, t in src.items():
                v[k].copy_(t.view(v[k].shape))
    eng.t = orc.opt_d.t


def measure(family, B, steps=2, lr=2e-4, with_emu=True, size=64):
    """-> list (one entry per teacher-forced step) of {"scalars": {k: {hip_vs_ref, ...}}, "d_grads": {...}, "g_grads": {...}}."""
    from hipgan.engine import CganEngine, DcganEngine
    from oracle.gan_oracle import GanOracle
    from util import synth_images
    ref = GanOracle(family, lr=lr, seed=12345, image_size=size)
    emu = GanOracle(family, lr=lr, seed=12345, emulate_bf16=True, image_size=size) if with_emu else None
    eng = (CganEngine if family == "cgan" else DcganEngine)(batch=B, prec="bf16", **({"image_size": size} if size != 64 else {}))
    imgs = synth_images(B * steps) if size == 64 else torch.nn.functional.interpolate(synth_images(B * steps), size=size, mode="nearest")
    out = []
    for s in range(steps):
        real = imgs[s * B:(s + 1) * B]
        lab = labels_for(B, 77 + s) if family == "cgan" else None
        nz = noise_for(family, B, 100 + s, lab, size)
        _force_engine(eng, ref)
        if emu is not None:
            _force_oracle(emu, ref)
        r = ref.step(real, lab, nz)
        e = emu.step(real, lab, nz) if emu is not None else None
        h = eng.step(real.cuda(), {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in nz.items()}, lr=lr)
        row = {"scalars": {}, "d_grads": {}, "g_grads": {}}
        for k in SCALARS:
            row["scalars"][k] = {"hip_vs_ref": rel(h[k], r[k])}
            if e is not None:
                row["scalars"][k].update(emu_vs_ref=rel(e[k], r[k]), hip_vs_emu=rel(h[k], e[k]))
        for tag, rg, eg in (("d", ref.d_grads, emu.d_grads if emu else None), ("g", ref.g_grads, emu.g_grads if emu else None)):
            views = eng.named_views(tag, "grads")
            for k, t in rg.items():
                d = {"hip_vs_ref": rel_l2(views[k].view(t.shape), t)}
                if eg is not None:
                    d.update(emu_vs_ref=rel_l2(eg[k], t), hip_vs_emu=rel_l2(views[k].view(t.shape), eg[k]))
                row[f"{tag}_grads"][k] = d
        out.append(row)
    return out


def worst(rows, group, key):
    """max over steps and tensors of one distance."""
    return max(v[key] for row in rows for v in row[group].values() if key in v)
