#!/usr/bin/env python3
"""Golden-vector generator.  RUNS ONLY IN THE BUILD CONTAINER.

Imports the *unmodified* reference (`/root/reference`) under in-memory stubs for the two
third-party packages this image lacks (torchinfo, torchvision) and records what the reference
computes on seeded synthetic inputs.  Only DATA (scalars, checksums, sampled elements) is written
to `tests/golden/*.json`; no reference source travels.  Method: SURVEY.md section 8c.

    python tests/golden/make_golden.py            # regenerates every fixture (about 2 min of CPU)

Fixtures
  modules.json      forward/backward of the four nets on B=4 seeded inputs
  dcgan_steps.json  real DCGANTrainer.train() for 3 steps, B=8 and B=64, lr 2e-4 (+ lr 0.1 clamp run)
  cgan_steps.json   real CGANTrainer.train() for 2 steps, B=8 and B=32, lr 2e-4
  metrics.json      Metrics.inception_score / fid / intra_fid on seeded 100-d features
  selfdiv.json      reference vs itself (8 threads vs 1 thread): long-horizon noise floor
"""
import argparse
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
MODEL_SEED = 12345          # change_randomseed.py:1
DATA_SEED = 2024


# --------------------------------------------------------------------------------------------
# stubs for the packages the image lacks (ordinary ModuleNotFoundError otherwise)
# --------------------------------------------------------------------------------------------
def install_stubs():
    import matplotlib
    matplotlib.use("Agg")
    ti = types.ModuleType("torchinfo")
    ti.summary = lambda *a, **k: "<summary stub>"
    sys.modules["torchinfo"] = ti

    tv = types.ModuleType("torchvision")
    tvu = types.ModuleType("torchvision.utils")

    def make_grid(t, padding=2, normalize=False, **k):
        t = t.detach().cpu()
        return t[0] if t.dim() == 4 else t
    tvu.make_grid = make_grid
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")

    def resize(img, size, **k):
        return torch.nn.functional.interpolate(img, size=size, mode="bilinear", align_corners=False)
    tvf.resize = resize
    tvt.functional = tvf
    tvm = types.ModuleType("torchvision.models")
    tvd = types.ModuleType("torchvision.datasets")
    tv.utils, tv.transforms, tv.models, tv.datasets = tvu, tvt, tvm, tvd
    for name, mod in [("torchvision", tv), ("torchvision.utils", tvu), ("torchvision.transforms", tvt),
                      ("torchvision.transforms.functional", tvf), ("torchvision.models", tvm),
                      ("torchvision.datasets", tvd)]:
        sys.modules[name] = mod


# --------------------------------------------------------------------------------------------
# tensor digests
# --------------------------------------------------------------------------------------------
def sample_idx(name, numel, k=8):
    h = int(hashlib.sha256(name.encode()).hexdigest()[:8], 16)
    rng = np.random.default_rng(h)
    return rng.integers(0, numel, size=min(k, numel)).tolist()


def digest(name, t):
    a = t.detach().cpu().double().reshape(-1).numpy()
    idx = sample_idx(name, a.size)
    return {"shape": list(t.shape), "sum": float(a.sum()), "abssum": float(np.abs(a).sum()),
            "l2": float(np.sqrt((a * a).sum())), "idx": idx, "vals": [float(a[i]) for i in idx]}


def digest_dict(prefix, named):
    return {k: digest(prefix + k, v) for k, v in named}


def synth_images(n, seed=DATA_SEED):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, 3, 64, 64, generator=g) * 2 - 1


def synth_onehot(n, seed=DATA_SEED + 1, classes=100):
    g = torch.Generator().manual_seed(seed)
    lab = torch.randint(0, classes, (n,), generator=g)
    return torch.nn.functional.one_hot(lab, 100).to(torch.int64), lab


# --------------------------------------------------------------------------------------------
# module-level goldens
# --------------------------------------------------------------------------------------------
def gen_modules():
    from model import DCGAN, CGAN
    out = {}
    B = 4
    for fam, mod in (("dcgan", DCGAN), ("cgan", CGAN)):
        torch.manual_seed(MODEL_SEED)
        g, d = mod.Generator(), mod.Discriminator()
        g.apply(mod.weights_init)
        d.apply(mod.weights_init)
        gen = torch.Generator().manual_seed(77)
        z = torch.randn(B, 100, 1, 1, generator=gen)
        x = synth_images(B, seed=99).requires_grad_(True)
        rg = torch.randn(B, 3, 64, 64, generator=gen)
        rd = torch.randn(B, generator=gen)
        oh, _ = synth_onehot(B, seed=5)
        rec = {"init_g": digest_dict(fam + ".g.", g.state_dict().items()),
               "init_d": digest_dict(fam + ".d.", d.state_dict().items())}
        if fam == "cgan":
            d.drop1.p = 0.0    # dropout mask is RNG-dependent; module golden pins the p=0 function
            fake = g(z, oh)
            dout = d(x, oh).view(-1)
        else:
            fake = g(z)
            dout = d(x).view(-1)
        (fake * rg).sum().backward()
        (dout * rd).sum().backward()
        rec["g_out"] = digest(fam + ".g_out", fake)
        rec["d_out"] = {"vals": dout.detach().double().tolist()}
        rec["g_grads"] = digest_dict(fam + ".gg.", [(k, p.grad) for k, p in g.named_parameters()])
        rec["d_grads"] = digest_dict(fam + ".dg.", [(k, p.grad) for k, p in d.named_parameters()])
        rec["d_xgrad"] = digest(fam + ".d_xgrad", x.grad)
        rec["g_post"] = digest_dict(fam + ".gpost.", g.state_dict().items())
        rec["d_post"] = digest_dict(fam + ".dpost.", d.state_dict().items())
        out[fam] = rec
    return out


# --------------------------------------------------------------------------------------------
# real trainers under a synthetic data_pre
# --------------------------------------------------------------------------------------------
class Args:
    def __init__(self, **k):
        self.__dict__.update(k)


class SynthPre:
    """duck type of preprocess.*DataPreprocessor: get_data_loader() -> (batches, metric_source)."""

    def __init__(self, batches):
        self.batches = batches
        self.idx_to_labels = {i: str(i) for i in range(100)}

    def get_data_loader(self):
        return self.batches, None


class FakeMetrics:
    def __init__(self, *_):
        pass

    def inception_score(self, *_a, **_k):
        return 1.0

    def fid(self, *_a, **_k):
        return 1.0

    def intra_fid(self, *_a, **_k):
        return 1.0


def run_trainer(fam, B, steps, lr, tmp):
    """Runs the reference's own <X>Trainer.train() and records per-step quantities through wrappers
    around objects the trainer exposes (criterion, optimizers, compute_gradient_penalty)."""
    import matplotlib.pyplot as plt
    from logger.main_logger import MainLogger
    MainLogger._instance = None
    MainLogger._initialized = False
    import logging
    logging.getLogger("main").handlers.clear()
    if fam == "dcgan":
        from model import DCGAN as M
        from train import dcgan_trainer as T
        TR = T.DCGANTrainer
    else:
        from model import CGAN as M
        from train import cgan_trainer as T
        TR = T.CGANTrainer
    T.Metrics = FakeMetrics
    imgs = synth_images(B * steps)
    if fam == "dcgan":
        batches = [(imgs[i * B:(i + 1) * B],) for i in range(steps)]
    else:
        oh, _ = synth_onehot(B * steps)
        batches = [(imgs[i * B:(i + 1) * B], oh[i * B:(i + 1) * B]) for i in range(steps)]
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        args = Args(epoch=1, max_learning_rate=lr, model_path="golden", log_file=0,
                    save_path=os.path.join(tmp, "save", fam, "golden"), batch_size=B, num_worker=0)
        torch.manual_seed(MODEL_SEED)
        g, d = M.Generator(), M.Discriminator()
        tr = TR(args, g, d, SynthPre(batches))
        rec = {"B": B, "steps": steps, "lr": lr,
               "init_g": digest_dict(f"{fam}.g.", g.state_dict().items()),
               "init_d": digest_dict(f"{fam}.d.", d.state_dict().items()), "step": []}
        cur = {"crit": [], "gp": None}

        crit = tr.criterion

        def crit_wrap(o, t):
            l = crit(o, t)
            cur["crit"].append({"out": o.detach().double().tolist(), "loss": float(l.detach())})
            return l
        tr.criterion = crit_wrap
        gp_fn = tr.compute_gradient_penalty

        def gp_wrap(*a):
            v = gp_fn(*a)
            cur["gp"] = float(v.detach())
            return v
        tr.compute_gradient_penalty = gp_wrap
        od_step, og_step = tr.optimizer_d.step, tr.optimizer_g.step

        def od_wrap(*a, **k):
            cur["d_grads"] = digest_dict(f"{fam}.dg.", [(n, p.grad) for n, p in d.named_parameters()])
            r = od_step(*a, **k)
            cur["d_post"] = digest_dict(f"{fam}.dpost.", d.state_dict().items())
            return r

        def og_wrap(*a, **k):
            cur["g_grads"] = digest_dict(f"{fam}.gg.", [(n, p.grad) for n, p in g.named_parameters()])
            r = og_step(*a, **k)
            cur["g_post"] = digest_dict(f"{fam}.gpost.", g.state_dict().items())
            # D's BN running stats move once more in the G phase (4th D pass)
            cur["d_post_bn"] = digest_dict(f"{fam}.dpostbn.", [(n, b) for n, b in d.named_buffers()])
            rec["step"].append(dict(cur))
            cur["crit"] = []
            return r
        tr.optimizer_d.step, tr.optimizer_g.step = od_wrap, og_wrap
        captured = {}
        plot = plt.plot

        def plot_wrap(x, y, *a, **k):
            captured[k.get("label", "?")] = [float(v) for v in y]
            return plot(x, y, *a, **k)
        plt.plot = plot_wrap
        try:
            tr.train()
        finally:
            plt.plot = plot
            plt.close("all")
        rec["losses_d"] = captured.get("Discriminator Loss")
        rec["losses_g"] = captured.get("Generator Loss")
        rec["final_g"] = digest_dict(f"{fam}.gfin.", g.state_dict().items())
        rec["final_d"] = digest_dict(f"{fam}.dfin.", d.state_dict().items())
        rec["saved"] = sorted(os.path.relpath(os.path.join(r, f), tmp) for r, _, fs in os.walk(tmp) for f in fs)
        pts = [p for p in rec["saved"] if p.endswith(".pt")]
        if pts:
            ck = torch.load(os.path.join(tmp, pts[0]), weights_only=False)
            rec["ckpt_keys"] = sorted(ck.keys())
            rec["ckpt_g_keys"] = list(ck["model_g"].keys())
            rec["ckpt_d_keys"] = list(ck["model_d"].keys())
        return rec
    finally:
        os.chdir(cwd)


def gen_steps(fam):
    import tempfile
    out = {}
    cfgs = [(8, 3, 2e-4), (64, 3, 2e-4)] if fam == "dcgan" else [(8, 2, 2e-4), (32, 2, 2e-4)]
    for B, steps, lr in cfgs:
        with tempfile.TemporaryDirectory() as tmp:
            out[f"B{B}"] = run_trainer(fam, B, steps, lr, tmp)
    if fam == "dcgan":
        with tempfile.TemporaryDirectory() as tmp:
            r = run_trainer(fam, 8, 4, 0.1, tmp)       # CLI default lr: pins the -100 clamp plateau
            out["B8_lr0.1"] = {k: r[k] for k in ("B", "steps", "lr", "losses_d", "losses_g")}
            out["B8_lr0.1"]["crit"] = [[c["loss"] for c in s["crit"]] for s in r["step"]]
            out["B8_lr0.1"]["gp"] = [s["gp"] for s in r["step"]]
    return out


# --------------------------------------------------------------------------------------------
# metrics arithmetic (reference's Metrics methods, unmodified, on supplied features)
# --------------------------------------------------------------------------------------------
def gen_metrics():
    import metrics as RM
    rng = np.random.default_rng(7)
    real = (1.5 * rng.standard_normal((5000, 100)) + 0.2).astype(np.float32)
    fake = rng.standard_normal((1000, 100)).astype(np.float32)
    real_targets = rng.integers(0, 100, size=5000).tolist()
    m = RM.Metrics.__new__(RM.Metrics)
    # the superclass table is data inside Metrics.__init__; rebuild the two index maps the same way
    src = {}
    tbl = [[4, 30, 55, 72, 95], [1, 32, 67, 73, 91], [54, 62, 70, 82, 92], [9, 10, 16, 28, 61], [0, 51, 53, 57, 83],
           [22, 39, 40, 86, 87], [5, 20, 25, 84, 94], [6, 7, 14, 18, 24], [3, 42, 43, 88, 97], [12, 17, 37, 68, 76],
           [23, 33, 49, 60, 71], [15, 19, 21, 31, 38], [34, 63, 64, 66, 75], [26, 45, 77, 79, 99], [2, 11, 35, 46, 98],
           [27, 29, 44, 78, 93], [36, 50, 65, 74, 80], [47, 52, 56, 59, 96], [8, 13, 48, 58, 90], [41, 69, 81, 85, 89]]
    for s, cl in enumerate(tbl):
        for c in cl:
            src[c] = s
    fake_targets = [i for i in range(100) for _ in range(10)]
    m.real_superclass_idx = {s: [i for i, t in enumerate(real_targets) if src[t] == s] for s in range(20)}
    m.fake_superclass_idx = {s: [i for i, t in enumerate(fake_targets) if src[t] == s] for s in range(20)}
    m.real_features = real
    m.device = torch.device("cpu")

    def extract(images, real=False, softmax=False):
        feats = np.vstack([b.numpy() for b in images])
        if softmax:
            feats = torch.softmax(torch.from_numpy(feats), dim=1).numpy()
        return feats
    m._Metrics__extract_features = extract
    ft = torch.from_numpy(fake)
    dl = torch.utils.data.DataLoader(ft, batch_size=128, shuffle=False)
    rec = {"recipe": {"rng": 7, "real": "1.5*N(0,1)+0.2 [5000,100] f32", "fake": "N(0,1) [1000,100] f32",
                      "real_targets": "integers(0,100,5000)"},
           "is": float(m.inception_score(dl)), "fid": float(m.fid(dl)), "intra_fid": float(m.intra_fid(ft))}
    dl64 = torch.utils.data.DataLoader(ft[:64], batch_size=64)
    rec["is64"] = float(m.inception_score(dl64))
    rec["fid64"] = float(m.fid(dl64))
    return rec


# --------------------------------------------------------------------------------------------
# reference vs itself: chaotic-trajectory noise floor
# --------------------------------------------------------------------------------------------
def gen_selfdiv(steps=30, B=64):
    from model import DCGAN as M
    import copy
    torch.manual_seed(MODEL_SEED)
    g0, d0 = M.Generator(), M.Discriminator()
    g0.apply(M.weights_init)
    d0.apply(M.weights_init)
    imgs = synth_images(B * 4)
    gen = torch.Generator().manual_seed(31)
    noise = [(torch.randn(B, 3, 64, 64, generator=gen), torch.randn(B, 100, 1, 1, generator=gen),
              torch.randn(B, 3, 64, 64, generator=gen), torch.rand(B, 1, 1, 1, generator=gen)) for _ in range(steps)]
    crit = torch.nn.BCELoss()

    def run(nthreads):
        torch.set_num_threads(nthreads)
        g, d = copy.deepcopy(g0), copy.deepcopy(d0)
        og = torch.optim.Adam(g.parameters(), lr=2e-4, betas=[0.5, 0.999])
        od = torch.optim.Adam(d.parameters(), lr=2e-4, betas=[0.5, 0.999])
        ld, lg = [], []
        for s in range(steps):
            n1, z, n2, al = noise[s]
            real = imgs[(s % 4) * B:(s % 4 + 1) * B]
            # statements of train/dcgan_trainer.py:155-189 with the RNG draws replaced by the shared tensors
            d.zero_grad()
            label = torch.full((B,), 0.9)
            real = 0.9 * real + 0.1 * n1
            out = d(real).view(-1)
            e_real = crit(out, label)
            e_real.backward()
            fake = g(z)
            label.fill_(0.1)
            fake = 0.9 * fake + 0.1 * n2
            out = d(fake.detach()).view(-1)
            e_fake = crit(out, label)
            e_fake.backward()
            inter = (al * real + (1 - al) * fake).requires_grad_(True)
            di = d(inter)
            gr = torch.autograd.grad(di, inter, torch.ones_like(di), create_graph=True, retain_graph=True)[0]
            gp = ((gr.view(B, -1).norm(2, dim=1) - 1) ** 2).mean()
            e_d = e_real + e_fake + 10.0 * gp
            od.step()
            g.zero_grad()
            label.fill_(0.9)
            out = d(fake).view(-1)
            e_g = crit(out, label)
            e_g.backward()
            og.step()
            ld.append(float(e_d))
            lg.append(float(e_g))
        return ld, lg
    d8, g8 = run(8)
    d1, g1 = run(1)
    torch.set_num_threads(8)
    return {"B": B, "steps": steps, "loss_d_8t": d8, "loss_g_8t": g8, "loss_d_1t": d1, "loss_g_1t": g1,
            "rel_d": [abs(a - b) / max(abs(a), 1e-12) for a, b in zip(d8, d1)],
            "rel_g": [abs(a - b) / max(abs(a), 1e-12) for a, b in zip(g8, g1)]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    assert os.path.isdir(REF), "the reference is only mounted in the build container"
    install_stubs()
    sys.path.insert(0, REF)
    torch.set_num_threads(8)
    jobs = {"modules": gen_modules, "dcgan_steps": lambda: gen_steps("dcgan"),
            "cgan_steps": lambda: gen_steps("cgan"), "metrics": gen_metrics, "selfdiv": gen_selfdiv}
    for name, fn in jobs.items():
        if a.only and name not in a.only.split(","):
            continue
        rec = fn()
        rec["_meta"] = {"torch": torch.__version__, "threads": torch.get_num_threads(),
                        "generator": "tests/golden/make_golden.py", "model_seed": MODEL_SEED, "data_seed": DATA_SEED}
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(rec, f)
        print(name, "->", os.path.getsize(os.path.join(HERE, name + ".json")), "bytes")


if __name__ == "__main__":
    main()
