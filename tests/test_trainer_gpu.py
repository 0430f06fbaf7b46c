"""The drop-in entry surface on the GPU: DCGANTrainer(args, Generator(), Discriminator(), data_pre).train() - wired exactly
as the reference's main.py:83-96 wires it - against the fixture recorded from the reference's own trainer
(tests/golden/dcgan_steps.json: same seed, same batches, noise from the CPU generator in the same order)."""
import argparse
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


class SynthPre:
    def __init__(self, batches):
        self.batches = batches

    def get_data_loader(self):
        return self.batches, None


def _fresh_logger():
    import logging
    from logger.main_logger import MainLogger
    logging.getLogger("main").handlers.clear()
    MainLogger._instance, MainLogger._initialized = None, False


@pytest.mark.parametrize("key", ["B8", "B64"])
def test_trainer_reproduces_reference_run(key, tmp_path, monkeypatch):
    from model import DCGAN
    from train.dcgan_trainer import DCGANTrainer
    from util import check_digest_dict, load_golden, rel, synth_images
    gold = load_golden("dcgan_steps")[key]
    B, steps = gold["B"], gold["steps"]
    monkeypatch.chdir(tmp_path)
    _fresh_logger()
    imgs = synth_images(B * steps)
    batches = [(imgs[i * B:(i + 1) * B],) for i in range(steps)]
    args = argparse.Namespace(epoch=1, max_learning_rate=gold["lr"], model_path="golden", log_file=0,
                              save_path=str(tmp_path / "save" / "dcgan" / "golden"), batch_size=B, num_worker=0)
    torch.manual_seed(12345)
    g, d = DCGAN.Generator(), DCGAN.Discriminator()
    tr = DCGANTrainer(args, g, d, SynthPre(batches), prec="f32", host_rng=True)
    losses_d, losses_g = tr.train()
    assert len(losses_d) == steps
    for s in range(steps):
        tol = 1e-3 if s == 0 else 5e-3            # free-running: later steps inherit Adam's amplification of rounding
        assert rel(losses_d[s], gold["losses_d"][s]) < tol, (s, losses_d, gold["losses_d"])
        assert rel(losses_g[s], gold["losses_g"][s]) < tol * 3, (s, losses_g, gold["losses_g"])
    # the modules ARE the live training state (zero copy) and match the reference's final weights
    check_digest_dict({k: v for k, v in g.state_dict().items()}, gold["final_g"], 2e-2, 5e-4, "final_g")
    check_digest_dict({k: v for k, v in d.state_dict().items()}, gold["final_d"], 2e-2, 5e-4, "final_d")
    # checkpoint: same files and dict as the reference wrote (no Inception weights here -> 'latest' instead of fid/is)
    saved = sorted(os.path.relpath(os.path.join(r, f), tmp_path) for r, _, fs in os.walk(tmp_path) for f in fs)
    pts = [p for p in saved if p.endswith(".pt")]
    assert pts and "save/dcgan/golden/real_image.png" in saved and "save/dcgan/golden/loss.png" in saved
    ck = torch.load(tmp_path / pts[0], weights_only=False)
    assert sorted(ck.keys()) == gold["ckpt_keys"]
    assert list(ck["model_g"].keys()) == gold["ckpt_g_keys"] and list(ck["model_d"].keys()) == gold["ckpt_d_keys"]
    # ... and it loads into plain torch objects built the reference's way
    g2, d2 = DCGAN.Generator(), DCGAN.Discriminator()
    g2.load_state_dict(ck["model_g"])
    d2.load_state_dict(ck["model_d"])
    torch.optim.Adam(g2.parameters(), lr=0.1, betas=[0.5, 0.999]).load_state_dict(ck["optimizer_g"])
    torch.optim.Adam(d2.parameters(), lr=0.1, betas=[0.5, 0.999]).load_state_dict(ck["optimizer_d"])
    # resume path (commented out in the reference): state round-trips
    tr.load_model(str(tmp_path / pts[0]))
    _fresh_logger()


def test_bf16_trainer_step0_against_the_reference_run(tmp_path, monkeypatch):
    """The bf16 fast path (what bench.py times) against the fixture recorded from the REFERENCE's own DCGANTrainer.train()
    (tests/golden/dcgan_steps.json, batch 64, CPU-generator noise in the reference's order): the losses of the first step -
    identical state on both sides - within 1e-2 relative (the bf16 per-step gate), later steps inside the
    trajectory's own amplification."""
    from model import DCGAN
    from train.dcgan_trainer import DCGANTrainer
    from util import load_golden, rel, synth_images
    gold = load_golden("dcgan_steps")["B64"]
    B, steps = gold["B"], gold["steps"]
    monkeypatch.chdir(tmp_path)
    _fresh_logger()
    imgs = synth_images(B * steps)
    batches = [(imgs[i * B:(i + 1) * B],) for i in range(steps)]
    args = argparse.Namespace(epoch=1, max_learning_rate=gold["lr"], model_path="golden", log_file=0,
                              save_path=str(tmp_path / "save" / "dcgan" / "golden"), batch_size=B, num_worker=0)
    torch.manual_seed(12345)
    g, d = DCGAN.Generator(), DCGAN.Discriminator()
    tr = DCGANTrainer(args, g, d, SynthPre(batches), prec="bf16", host_rng=True)
    losses_d, losses_g = tr.train()
    assert rel(losses_d[0], gold["losses_d"][0]) < 1e-2, (losses_d, gold["losses_d"])
    assert rel(losses_g[0], gold["losses_g"][0]) < 1e-2, (losses_g, gold["losses_g"])
    for s in range(1, steps):
        assert rel(losses_d[s], gold["losses_d"][s]) < 3e-2 and rel(losses_g[s], gold["losses_g"][s]) < 6e-2, (s, losses_d, losses_g)
    _fresh_logger()


def test_bf16_cgan_trainer_step0_against_the_reference_run(tmp_path, monkeypatch):
    """Same for CGAN (tests/golden/cgan_steps.json, batch 32, back-propagated penalty, CPU-generator dropout masks)."""
    from model import CGAN
    from train.cgan_trainer import CGANTrainer
    from util import load_golden, rel, synth_images, synth_onehot
    gold = load_golden("cgan_steps")["B32"]
    B, steps = gold["B"], gold["steps"]
    monkeypatch.chdir(tmp_path)
    _fresh_logger()
    imgs = synth_images(B * steps)
    oh, _ = synth_onehot(B * steps)
    batches = [(imgs[i * B:(i + 1) * B], oh[i * B:(i + 1) * B]) for i in range(steps)]
    args = argparse.Namespace(epoch=1, max_learning_rate=gold["lr"], model_path="golden", log_file=0,
                              save_path=str(tmp_path / "save" / "cgan" / "golden"), batch_size=B, num_worker=0)
    torch.manual_seed(12345)
    g, d = CGAN.Generator(), CGAN.Discriminator()
    tr = CGANTrainer(args, g, d, SynthPreC(batches), prec="bf16", host_rng=True)
    losses_d, losses_g = tr.train()
    assert rel(losses_d[0], gold["losses_d"][0]) < 1e-2, (losses_d, gold["losses_d"])
    assert rel(losses_g[0], gold["losses_g"][0]) < 1e-2, (losses_g, gold["losses_g"])
    _fresh_logger()


def test_main_wiring_with_ragged_last_batch(tmp_path, monkeypatch):
    """python main.py -m DCGAN on synthetic data: 40 images, batch 16 -> batches of 16, 16, 8 (second engine bound to the
    same arenas), device RNG, bf16 fast path; losses finite, modules updated, G usable as an nn.Module afterwards."""
    monkeypatch.chdir(tmp_path)
    _fresh_logger()
    import main
    from model import DCGAN
    from preprocess.dcgan_data_preprocessor import DCGANDataPreprocessor
    from train.dcgan_trainer import DCGANTrainer
    args = main.get_arg_parse(["-m", "DCGAN", "-b", "16", "-e", "2", "-mlr", "0.0002", "-pm", "t", "-lf", "1"])
    args.save_path = str(tmp_path / "save" / "dcgan" / "t")
    pre = DCGANDataPreprocessor(args, synthetic_size=40)
    pre.transform_data()
    g, d = DCGAN.Generator(), DCGAN.Discriminator()
    w0 = g.conv3.weight.detach().clone()
    tr = DCGANTrainer(args, g, d, pre)
    ld, lg = tr.train()
    assert len(ld) == 6 and all(map(lambda v: v == v and abs(v) < 1e3, ld + lg))
    assert not torch.equal(g.conv3.weight.detach().cpu(), w0)
    assert int(d.norm1.num_batches_tracked) == 6 * 4 and int(g.norm1.num_batches_tracked) >= 6
    out = g(torch.randn(5, 100, 1, 1, device="cuda"))
    assert out.shape == (5, 3, 64, 64) and float(out.abs().max()) <= 1.0
    assert any(f.endswith(".log") for f in os.listdir(args.save_path))
    _fresh_logger()


class SynthPreC(SynthPre):
    idx_to_labels = {i: str(i) for i in range(100)}


@pytest.mark.parametrize("key", ["B8", "B32"])
def test_cgan_trainer_reproduces_reference_run(key, tmp_path, monkeypatch):
    """CGANTrainer on the exact-fp32 path with CPU-generator noise and dropout masks against the fixture recorded from the
    reference's own CGANTrainer.train() (tests/golden/cgan_steps.json): includes the back-propagated gradient penalty."""
    from model import CGAN
    from train.cgan_trainer import CGANTrainer
    from util import load_golden, rel, synth_images, synth_onehot
    gold = load_golden("cgan_steps")[key]
    B, steps = gold["B"], gold["steps"]
    monkeypatch.chdir(tmp_path)
    _fresh_logger()
    imgs = synth_images(B * steps)
    oh, _ = synth_onehot(B * steps)
    batches = [(imgs[i * B:(i + 1) * B], oh[i * B:(i + 1) * B]) for i in range(steps)]
    args = argparse.Namespace(epoch=1, max_learning_rate=gold["lr"], model_path="golden", log_file=0,
                              save_path=str(tmp_path / "save" / "cgan" / "golden"), batch_size=B, num_worker=0)
    torch.manual_seed(12345)
    g, d = CGAN.Generator(), CGAN.Discriminator()
    tr = CGANTrainer(args, g, d, SynthPreC(batches), prec="f32", host_rng=True)
    losses_d, losses_g = tr.train()
    for s in range(steps):
        tol = 1e-3 if s == 0 else 1e-2
        assert rel(losses_d[s], gold["losses_d"][s]) < tol, (s, losses_d, gold["losses_d"])
        assert rel(losses_g[s], gold["losses_g"][s]) < tol * 5, (s, losses_g, gold["losses_g"])
    saved = sorted(os.path.relpath(os.path.join(r, f), tmp_path) for r, _, fs in os.walk(tmp_path) for f in fs)
    pts = [p for p in saved if p.endswith(".pt")]
    ck = torch.load(tmp_path / pts[0], weights_only=False)
    assert sorted(ck.keys()) == gold["ckpt_keys"]
    assert list(ck["model_g"].keys()) == gold["ckpt_g_keys"] and list(ck["model_d"].keys()) == gold["ckpt_d_keys"]
    g2, d2 = CGAN.Generator(), CGAN.Discriminator()
    g2.load_state_dict(ck["model_g"])
    d2.load_state_dict(ck["model_d"])
    torch.optim.Adam(d2.parameters(), lr=0.1, betas=[0.5, 0.999]).load_state_dict(ck["optimizer_d"])
    _fresh_logger()


def test_evaluation_branch_runs_on_a_side_stream_with_the_hip_metric_network(tmp_path, monkeypatch):
    """The every-500-iterations branch (reference train/dcgan_trainer.py:198-221) with a real metric network: G(fixed_noise) ->
    299x299 -> Inception-v3 (HIP chain, seeded random weights: the fine-tuned ones are not available offline) -> IS / FID with
    device-side fp64 mean / covariance -> best-score checkpoints.  The device part sits on a side stream and the host part is
    deferred (train/async_eval.py); the checkpoint must still hold the state of the EVALUATION iteration, not of the moment it
    is written."""
    from inception import InceptionV3Hip
    from metrics import Metrics
    from model import DCGAN
    from oracle.inception_oracle import random_state_dict
    from train.dcgan_trainer import DCGANTrainer
    from util import synth_images
    monkeypatch.chdir(tmp_path)
    _fresh_logger()
    B, steps = 8, 3
    imgs = synth_images(B * steps)
    batches = [(imgs[i * B:(i + 1) * B],) for i in range(steps)]
    args = argparse.Namespace(epoch=1, max_learning_rate=2e-4, model_path="ev", log_file=0,
                              save_path=str(tmp_path / "save" / "dcgan" / "ev"), batch_size=B, num_worker=0)
    net = InceptionV3Hip(random_state_dict(0))
    real = torch.randn(300, 100, generator=torch.Generator().manual_seed(1)).numpy()
    monkeypatch.setattr(DCGANTrainer, "_make_metrics", lambda self, loader: Metrics(None, extractor=net, real_features=real))
    torch.manual_seed(12345)
    tr = DCGANTrainer(args, DCGAN.Generator(), DCGAN.Discriminator(), SynthPre(batches), prec="f32")
    tr.train()
    root = tmp_path / "save" / "dcgan" / "ev"
    for typ in ("fid", "is"):
        pts = [f for f in os.listdir(root / typ) if f.endswith(".pt")]
        assert len(pts) == 1, (typ, pts)                       # older checkpoints of the folder are deleted, as in the reference
        it = int(pts[0].split("_")[0])
        assert it in (0, steps - 1)                            # evaluated at iteration 0 and at the very last one
        ck = torch.load(root / typ / pts[0], weights_only=False)
        assert float(ck["optimizer_g"]["state"][0]["step"]) == it + 1        # the snapshot of THAT iteration
        assert int(ck["model_g"]["norm1.num_batches_tracked"]) == (it + 1) + (1 if it == 0 else 2)   # steps so far + samplings so far
        assert any(f.endswith("_fake_image.png") for f in os.listdir(root / typ))
    _fresh_logger()
