"""CPU tests of the host-side mirror of the reference interface (no GPU, no compute calls into the library)."""
import argparse
import os
import sys
import types

import numpy as np
import pytest
import torch

from util import load_golden, rel


def test_cli_flags_match_reference():
    import main
    a = main.get_arg_parse([])
    # reference main.py:43-57: names and defaults
    assert vars(a) == {"test": 0, "model_path": "", "log_file": 1, "model": main.ModelEnum.DCGAN, "num_worker": 0,
                       "batch_size": 128, "epoch": 100, "max_learning_rate": 0.1, "min_learning_rate": 1e-4,
                       "weight_decay": 5e-4, "nesterov": 1}
    b = main.get_arg_parse(["-m", "CGAN", "-b", "256", "-e", "3", "-mlr", "0.0002", "-pm", "run1", "-lf", "0", "-w", "2"])
    assert str(b.model) == "CGAN" and b.batch_size == 256 and b.epoch == 3 and b.max_learning_rate == 2e-4
    assert b.model_path == "run1" and b.log_file == 0 and b.num_worker == 2
    from change_randomseed import RANDOMSEED
    assert RANDOMSEED == 12345


def test_modules_have_reference_state_dict():
    from model import DCGAN
    gold = load_golden("dcgan_steps")["B8"]
    g, d = DCGAN.Generator(), DCGAN.Discriminator()
    assert list(g.state_dict().keys()) == gold["ckpt_g_keys"]
    assert list(d.state_dict().keys()) == gold["ckpt_d_keys"]
    assert sum(p.numel() for p in g.parameters()) == 3576704 and sum(p.numel() for p in d.parameters()) == 2765696
    from train.trainer import Trainer
    from train.dcgan_trainer import DCGANTrainer
    assert issubclass(DCGANTrainer, Trainer)
    with pytest.raises(TypeError):
        Trainer()


def test_trainer_refuses_cpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hipgan import JckError
    from model import DCGAN
    from train.dcgan_trainer import DCGANTrainer
    args = argparse.Namespace(epoch=1, max_learning_rate=2e-4, model_path="x", log_file=0, save_path="/tmp/x", batch_size=4,
                              num_worker=0)
    with pytest.raises(JckError):
        DCGANTrainer(args, DCGAN.Generator(), DCGAN.Discriminator(), None)


def test_metrics_arithmetic_matches_reference_golden():
    import metrics as M
    gold = load_golden("metrics")
    rng = np.random.default_rng(7)
    real = (1.5 * rng.standard_normal((5000, 100)) + 0.2).astype(np.float32)
    fake = rng.standard_normal((1000, 100)).astype(np.float32)
    real_targets = rng.integers(0, 100, size=5000).tolist()
    src = types.SimpleNamespace(targets=real_targets)
    m = M.Metrics(src, real_features=real)
    ft = torch.from_numpy(fake)
    dl = torch.utils.data.DataLoader(ft, batch_size=128, shuffle=False)
    assert rel(m.inception_score(dl), gold["is"]) < 1e-5
    assert rel(m.fid(dl), gold["fid"]) < 1e-8
    assert rel(m.intra_fid(ft), gold["intra_fid"]) < 1e-8
    dl64 = torch.utils.data.DataLoader(ft[:64], batch_size=64)
    assert rel(m.inception_score(dl64), gold["is64"]) < 1e-5
    assert rel(m.fid(dl64), gold["fid64"]) < 1e-7
    # the DCGAN path of the reference hands Metrics a source without .targets (its defect): accepted here
    M.Metrics(object(), real_features=real).fid(dl)


def test_metrics_unavailable_without_weights():
    import metrics as M
    with pytest.raises(M.MetricsUnavailable):
        M.Metrics(types.SimpleNamespace(targets=[0]))


def test_engine_adam_state_dict_is_torch_compatible():
    """A checkpoint's optimizer entry must load into torch.optim.Adam built as the reference builds it - and back."""
    from hipgan.optim import EngineAdam
    from model import DCGAN
    g = DCGAN.Generator()
    views_m = {k: torch.full_like(p, 0.25) for k, p in g.named_parameters()}
    views_v = {k: torch.full_like(p, 0.5) for k, p in g.named_parameters()}
    eng = types.SimpleNamespace(t=7, named_views=lambda tag, what: views_m if what == "m" else views_v,
                                arenas={"g_grads": torch.zeros(1)})
    opt = EngineAdam(eng, "g", g.named_parameters(), 2e-4, betas=[0.5, 0.999])
    sd = opt.state_dict()
    ref = torch.optim.Adam(g.parameters(), lr=0.1, betas=[0.5, 0.999])
    ref.load_state_dict(sd)                                  # raises if the structure is not torch's
    assert ref.param_groups[0]["lr"] == 2e-4 and tuple(ref.param_groups[0]["betas"]) == (0.5, 0.999)
    st = ref.state[next(iter(g.parameters()))]
    assert float(st["step"]) == 7 and float(st["exp_avg"].flatten()[0]) == 0.25 and float(st["exp_avg_sq"].flatten()[0]) == 0.5
    # and the reverse: a torch-written state dict loads into the engine-backed optimiser
    for p in g.parameters():
        p.grad = torch.ones_like(p)
    ref.step()
    sd2 = ref.state_dict()
    opt.load_state_dict(sd2)
    assert eng.t == 8
    assert torch.equal(views_m["conv1.weight"], sd2["state"][0]["exp_avg"])
    assert set(sd["param_groups"][0].keys()) >= {"lr", "betas", "eps", "weight_decay", "amsgrad", "params"}


def test_synthetic_preprocessor_duck_type(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    from logger.main_logger import MainLogger
    MainLogger._instance, MainLogger._initialized = None, False
    from preprocess.dcgan_data_preprocessor import DCGANDataPreprocessor
    args = argparse.Namespace(batch_size=16, num_worker=0, log_file=0, save_path=str(tmp_path))
    pre = DCGANDataPreprocessor(args, synthetic_size=40)
    pre.transform_data()
    loader, metric_src = pre.get_data_loader()
    assert len(loader) == 3 and hasattr(metric_src, "targets") and len(metric_src.targets) == 40
    batch = next(iter(loader))
    if torch.cuda.is_available():        # (run on a GPU box: the dataset lives in HBM as uint8 and a batch is an index vector)
        assert batch[0].size(0) == 16 and batch[0].data.dtype == torch.uint8
    else:
        assert batch[0].shape == (16, 3, 64, 64) and batch[0].dtype == torch.float32
        assert float(batch[0].min()) >= -1.0 and float(batch[0].max()) <= 1.0
    sizes = [b[0].size(0) for b in loader]
    assert sorted(sizes) == [8, 16, 16]                      # ragged last batch, like CIFAR's 50000 % 128


def test_logger_and_time_format(tmp_path):
    from logger.main_logger import MainLogger
    from logger.utils import time_to_str
    MainLogger._instance, MainLogger._initialized = None, False
    a = MainLogger(argparse.Namespace(log_file=1, save_path=str(tmp_path)))
    assert MainLogger(None) is a
    a.debug("hello")
    assert any(f.endswith(".log") for f in os.listdir(tmp_path))
    assert time_to_str(3725.5) == "1.0h 2.0m 5.5"
    import logging
    logging.getLogger("main").handlers.clear()
    MainLogger._instance, MainLogger._initialized = None, False


def test_synthetic_data_must_be_asked_for(tmp_path, monkeypatch):
    """No local CIFAR-100 and no explicit opt-in: the preprocessor refuses instead of silently training on random pixels."""
    import argparse
    import preprocess.dcgan_data_preprocessor as P
    monkeypatch.setattr(P, "CIFAR_DIR", str(tmp_path / "nowhere"))
    monkeypatch.delenv("JCKGAN_SYNTHETIC", raising=False)
    args = argparse.Namespace(batch_size=4, num_worker=0, log_file=0, save_path=str(tmp_path))
    with pytest.raises(FileNotFoundError):
        P.DCGANDataPreprocessor(args)
    monkeypatch.setenv("JCKGAN_SYNTHETIC", "1")
    assert P.DCGANDataPreprocessor(args).images.shape[1:] == (3, 32, 32)


def test_cgan_bf16_emulation_differentiates_twice_and_stays_near_the_fp32_oracle():
    """oracle/bf16_emu.py for CGAN: the back-propagated penalty (train/cgan_trainer.py:200-203) goes through straight-through
    rounding, so one emulated step from the oracle's state must (i) run - create_graph=True through every rounding point -,
    (ii) differ from the fp32 oracle (the rounding is really applied) and (iii) stay within the storage format's envelope
    (relative L2 of every gradient tensor below 0.3, scalars below 5e-2; measured at batch 8: 0.11 / 0.19 worst D / G tensor)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bf16_error as be
    from oracle.gan_oracle import GanOracle
    from util import synth_images
    B = 4
    ref = GanOracle("cgan", seed=12345)
    emu = GanOracle("cgan", seed=12345, emulate_bf16=True)
    lab = be.labels_for(B, 77)
    nz = be.noise_for("cgan", B, 100, lab)
    r, e = ref.step(synth_images(B), lab, nz), emu.step(synth_images(B), lab, nz)
    for k in ("loss_d", "loss_g", "gp", "loss_real", "loss_fake"):
        assert be.rel(e[k], r[k]) < 5e-2, (k, e[k], r[k])
    dist = [be.rel_l2(eg[k], t) for rg, eg in ((ref.d_grads, emu.d_grads), (ref.g_grads, emu.g_grads)) for k, t in rg.items()]
    assert max(dist) < 0.3 and max(dist) > 1e-3, dist


def test_bench_parent_starts_ranks_without_touching_torch(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent spawns the ranks itself, never imports torch (a process that has
    initialised the GPU must not start others on this pool) and returns the children's status - here, without a GPU, every rank
    refuses to run (no CPU fallback on the product path), so the launch as a whole must fail and say which ranks did."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: tests/test_bench_multirank_gpu.py covers the launch")
    # sitecustomize in a scratch directory makes `import torch` fatal in the PARENT only (children get a clean PYTHONPATH back)
    (tmp_path / "sitecustomize.py").write_text(
        "import os, sys\n"
        "if 'WORLD_SIZE' not in os.environ:\n"
        "    class _NoTorch:\n"
        "        def find_spec(self, name, path=None, target=None):\n"
        "            if name == 'torch':\n"
        "                raise ImportError('the self-launching parent must not import torch')\n"
        "            return None\n"
        "    sys.meta_path.insert(0, _NoTorch())\n")
    env = dict(os.environ, PYTHONPATH=str(tmp_path) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "8"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "ranks failed" in r.stderr and "must not import torch" not in r.stderr, r.stderr[-1500:]
