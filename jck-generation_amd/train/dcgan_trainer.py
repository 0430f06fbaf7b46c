"""DCGAN trainer for MI355X - drop-in for the reference's train/dcgan_trainer.py.

Same constructor `DCGANTrainer(args, model_g, model_d, data_pre)`, same `train()`, `save_model(typ, iters, value, images)`,
`compute_gradient_penalty(real, fake)`, same log line, same checkpoint dict {'model_g','model_d','optimizer_g','optimizer_d'}
and file names.  What differs is the execution: one iteration (reference train/dcgan_trainer.py:155-189) is a single native
schedule of gfx950 kernels (hipgan.engine.DcganEngine -> jck_engine_phase): the modules' parameters live in the engine's flat
arenas (zero copy), the six logged scalars stay on the device and are read back only when a line is logged, the optimiser
step is a fused flat Adam, and with torch.distributed initialised the D/G gradient arenas are all-reduced over RCCL (D's
reduction overlapping the gradient-penalty pass).  No CPU fallback.
"""
import argparse
import os
import time
from datetime import datetime

import numpy as np
import torch
import torch.nn as nn

from hipgan import JckError
from hipgan.dist import GradReducer, ReplicaGuard
from hipgan.engine import SCALAR_NAMES, DcganEngine, DeviceBatch
from hipgan.optim import EngineAdam
from logger.main_logger import MainLogger
from logger.utils import time_to_str
from model.DCGAN import weights_init
from train.async_eval import AsyncEval, snapshot_to_cpu
from train.trainer import Trainer
from utils import require_gpu

EVAL_EVERY, LOG_EVERY = 500, 100          # train/dcgan_trainer.py:198,191


def inception_input(fake):
    """[-1,1] images [N,3,64,64] on the device -> what the metric network is fed (reference :202-206): 0.5*x + 0.5,
    F.resize to 299x299 (bilinear), ImageNet normalisation - fused in jck_resize_norm."""
    from hipgan import lib
    from hipgan._lib import cur_stream
    fake = fake.to(torch.float32).contiguous()
    n = fake.size(0)
    out = torch.empty(n, 3, 299, 299, dtype=torch.float32, device=fake.device)
    mean = torch.tensor([0.485, 0.456, 0.406], device=fake.device)
    std = torch.tensor([0.229, 0.224, 0.225], device=fake.device)
    lib.jck_resize_norm(fake, out, n, 3, fake.size(2), fake.size(3), 299, 299, 0.5, 0.5, mean, std, cur_stream())
    return out


def _as_tensor(batch):
    """fp32 NCHW view of a loader batch (a DeviceBatch is transformed on the device first)."""
    return batch.materialize() if isinstance(batch, DeviceBatch) else batch


def _make_grid(images, nrow=8, padding=2, normalize=True):
    """Minimal stand-in for torchvision.utils.make_grid (absent in this image): [N,3,H,W] -> [3,H',W']."""
    x = images.detach().float().cpu()
    if normalize:
        lo, hi = float(x.min()), float(x.max())
        x = (x - lo) / max(hi - lo, 1e-5)
    n, c, h, w = x.shape
    ncol = min(nrow, n)
    nr = (n + ncol - 1) // ncol
    grid = torch.zeros(c, nr * (h + padding) + padding, ncol * (w + padding) + padding)
    for i in range(n):
        r, q = divmod(i, ncol)
        grid[:, padding + r * (h + padding):padding + r * (h + padding) + h,
             padding + q * (w + padding):padding + q * (w + padding) + w] = x[i]
    return grid


def _save_png(path, chw, title=None):
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        plt.clf()
        plt.axis("off")
        if title:
            plt.title(title)
        plt.imshow(np.transpose(chw.numpy(), (1, 2, 0)))
        plt.savefig(path)
        plt.close("all")
    except Exception as e:          # plotting is cosmetic; never take the run down
        MainLogger().warning(f"could not write {path}: {e}")


class DCGANTrainer(Trainer):
    def __init__(self, args: argparse.Namespace, model_g: nn.Module, model_d: nn.Module, data_pre, prec=None, host_rng=None):
        """prec: "bf16" (fast, default) or "f32" (exact-fp32 parity path); env JCKGAN_PREC.
        host_rng: draw every random tensor from the CPU generator in the reference's order and upload it (bit-identical
        noise to a CPU run of the reference; env JCKGAN_HOST_RNG=1).  Default: Philox on the device."""
        self.logger = MainLogger(args)
        self.device = require_gpu("DCGANTrainer")
        self.epoch = args.epoch
        self.max_lr = args.max_learning_rate
        self.lambda_gp = 10.0
        self.prec = prec or os.environ.get("JCKGAN_PREC", "bf16")

        self.host_rng = bool(int(os.environ.get("JCKGAN_HOST_RNG", "0"))) if host_rng is None else host_rng
        if self.host_rng:       # weights_init must consume the CPU generator, as it does in a CPU run of the reference
            model_g.apply(weights_init)
            model_d.apply(weights_init)
        self.model_g = model_g.to(self.device)
        self.model_d = model_d.to(self.device)
        n_g = sum(p.numel() for p in self.model_g.parameters())
        n_d = sum(p.numel() for p in self.model_d.parameters())
        self.logger.debug(f"Generator: {n_g} parameters\n{self.model_g}")
        self.logger.debug(f"Discriminator: {n_d} parameters\n{self.model_d}")
        if not self.host_rng:   # reference order (train/dcgan_trainer.py:46-55): move, then initialise on the device
            self.model_g.apply(weights_init)
            self.model_d.apply(weights_init)
        self.model_g.prec = self.model_d.prec = self.prec

        self.data_pre = data_pre
        self.train_loader, metric_loader = self.data_pre.get_data_loader()
        self.metric = self._make_metrics(metric_loader)

        # data parallel: identical initial weights on every rank, gradients all-reduced per step
        self.world, self.rank = 1, 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world, self.rank = torch.distributed.get_world_size(), torch.distributed.get_rank()
        self.batch_size = int(getattr(args, "batch_size", 128))
        self.engine = DcganEngine(batch=self.batch_size, prec=self.prec, device=self.device,
                                  image_size=getattr(self.model_g, "image_size", 64))     # 128: the configs[4] topology
        self.engine.adopt_modules(self.model_g, self.model_d)
        if self.world > 1:
            for key in ("g_params", "d_params", "g_bn", "d_bn"):
                torch.distributed.broadcast(self.engine.arenas[key], src=0)
            self.engine.mark_weights_changed()
        self._tail_engines = {}
        self.reducer = GradReducer(self.world) if self.world > 1 else None
        # every rank must hold the same parameters after a step: checked after the first steps and at every evaluation point; on a
        # mismatch the state is re-broadcast and the engines fall back to the plain all-reduce schedule (hipgan/dist.py)
        self.guard = (ReplicaGuard(lambda: [self.engine] + list(self._tail_engines.values()), self.world, log=self.logger.debug)
                      if self.world > 1 else None)
        # data parallel: main.py seeds every rank alike (identical initial weights), so the NOISE needs its own per-rank stream -
        # otherwise every replica draws the same z / instance noise / alpha, generates the same fake batch, and the all-reduce
        # averages N identical gradients (effective batch B instead of N*B for G and for the fake half of D)
        self.noise_gen = self.host_gen = None
        if self.world > 1:
            from change_randomseed import RANDOMSEED
            self.noise_gen = torch.Generator(device=self.device).manual_seed(RANDOMSEED + 1 + self.rank)
            self.host_gen = torch.Generator().manual_seed(RANDOMSEED + 1 + self.rank)
        # The step's own draws (z, alpha, instance noise, CGAN's dropout masks: Philox keyed by this seed and the step) follow the
        # run's seed on ONE GPU too: in the reference change_randomseed.RANDOMSEED / torch.manual_seed govern all of it
        # (main.py:31-37), so a different seed must give a different noise sequence here as well.
        from change_randomseed import RANDOMSEED as _SEED
        self._noise_seed = (int(torch.initial_seed()) ^ (_SEED << 20)) + 1 + self.rank
        self.engine.set_noise_seed(self._noise_seed)

        self.optimizer_g = EngineAdam(self.engine, "g", self.model_g.named_parameters(), self.max_lr, betas=[0.5, 0.999])
        self.optimizer_d = EngineAdam(self.engine, "d", self.model_d.named_parameters(), self.max_lr, betas=[0.5, 0.999])
        self.criterion = nn.BCELoss()       # kept for API parity; the step uses the fused head kernel (same -100 clamp)

        datetime_now = args.model_path if getattr(args, "model_path", "") != "" else datetime.now().strftime("%Y%m%d_%H%M%S")
        self.model_save_path = os.path.join(".", "save", "dcgan", datetime_now)
        os.makedirs(self.model_save_path, exist_ok=True)
        self.logger.debug(f"save path: {self.model_save_path}")

    # ------------------------------------------------------------------------------------------------------
    def _make_metrics(self, metric_loader):
        try:
            from metrics import Metrics
            return Metrics(metric_loader)
        except Exception as e:
            self.logger.warning(f"Inception-score / FID evaluation disabled: {e}")
            return None

    def _engine_for(self, b):
        if b == self.batch_size:
            return self.engine
        if b not in self._tail_engines:
            self._tail_engines[b] = DcganEngine(batch=b, share=self.engine)
            self._tail_engines[b].set_noise_seed(self._noise_seed)
        return self._tail_engines[b]

    # ------------------------------------------------------------------------------------------------------
    def save_model(self, typ, iters, value, images, snapshot=None):
        """snapshot: the state captured at the evaluation iteration (train/async_eval.py); None = the live state."""
        if self.rank != 0:
            return
        self.engine.join()                      # the G phase of the last step may still be in flight on its stream
        save_path = os.path.join(self.model_save_path, typ)
        os.makedirs(save_path, exist_ok=True)
        for filename in os.listdir(save_path):
            fp = os.path.join(save_path, filename)
            if os.path.isfile(fp) and filename.endswith(".pt"):
                os.remove(fp)
        state = snapshot_to_cpu(snapshot) if snapshot is not None else {
            "model_g": {k: v.detach().cpu().clone() for k, v in self.model_g.state_dict().items()},
            "model_d": {k: v.detach().cpu().clone() for k, v in self.model_d.state_dict().items()},
            "optimizer_g": self.optimizer_g.state_dict(), "optimizer_d": self.optimizer_d.state_dict()}
        self.engine.check()                     # (the copies above synchronised) never checkpoint a step whose grid barrier timed out
        torch.save(state, os.path.join(save_path, f"{iters}_{value:.04f}.pt"))
        _save_png(os.path.join(save_path, f"{iters}_fake_image.png"), _make_grid(images, padding=2, normalize=True), "fake images")
        self.logger.debug(f"{iters} model save")

    def load_model(self, path):
        """Resume from a checkpoint written by save_model() or by the reference (its load_model is commented out)."""
        saved = torch.load(path, map_location="cpu", weights_only=False)
        self.model_g.load_state_dict(saved["model_g"])
        self.model_d.load_state_dict(saved["model_d"])
        self.optimizer_g.load_state_dict(saved["optimizer_g"])
        self.optimizer_d.load_state_dict(saved["optimizer_d"])
        self.engine.mark_weights_changed()

    def compute_gradient_penalty(self, real_data, fake_data):
        """Stand-alone value of the penalty (reference :110-127) through the autograd Functions of the HIP path."""
        alpha = torch.rand(real_data.size(0), 1, 1, 1, device=self.device)
        inter = (alpha * real_data + ((1 - alpha) * fake_data)).detach().requires_grad_(True)
        d_inter = self.model_d(inter)
        grads = torch.autograd.grad(outputs=d_inter, inputs=inter, grad_outputs=torch.ones_like(d_inter))[0]
        grads = grads.view(grads.size(0), -1)
        return ((grads.norm(2, dim=1) - 1) ** 2).mean()

    # ------------------------------------------------------------------------------------------------------
    def _evaluate(self, fixed_noise, iters, best):
        """Device part of the evaluation (reference :198-212) on a side stream; training resumes behind the sampling kernels
        only (train/async_eval.py).  The host part runs in _finish_eval."""
        self._finish_eval(best, wait=True)          # the evaluation of 500 iterations ago, if its host part is still owed
        eng = self._engine_for(fixed_noise.size(0))

        def device_part(fake):
            if self.metric is None:
                return {"images": fake}
            x = inception_input(fake)                # :202-206 in one device pass
            logits = self.metric.logits(x)           # ONE pass of the metric network: IS and FID are scored on the same features
            out = {"images": x, "logits": logits}
            if logits.is_cuda:
                out.update(self.metric.fake_stats_device(logits))      # fp64 mean / covariance on the device
            return out
        self._eval.launch(iters, lambda: eng.sample(fixed_noise), device_part)      # sample: one train-mode BN batch (:199-200)

    def _finish_eval(self, best, wait):
        p = self._eval.take(wait)
        if p is None:
            return
        iters, host, snap = p["iters"], p["host"], p["snapshot"]
        if "logits" not in host:
            self.save_model("latest", iters, 0.0, host["images"], snap)
            return
        if "mu" in host:
            inception_score, fid = self.metric.scores_from_stats(host["logits"], host)
        else:
            inception_score, fid = self.metric.scores_from_logits(host["logits"])
        self.logger.debug(f"inception score: {inception_score}\tfid: {fid}")
        if best["fid"] > fid:
            best["fid"] = fid
            self.logger.debug(f"{iters} lowest fid")
            self.save_model("fid", iters, fid, host["images"], snap)
        if best["is"] < inception_score:
            best["is"] = inception_score
            self.logger.debug(f"{iters} highest is")
            self.save_model("is", iters, inception_score, host["images"], snap)

    def train(self):
        loader = self.train_loader
        n_iter = self.epoch * len(loader)
        fixed_noise = torch.randn(64, 100, 1, 1).to(self.device) if self.host_rng else torch.randn(64, 100, 1, 1, device=self.device)
        best = {"fid": 1e10, "is": 0}
        self._eval = AsyncEval(self)
        if self.rank == 0:
            real_batch = next(iter(loader))
            _save_png(os.path.join(self.model_save_path, "real_image.png"),
                      _make_grid(_as_tensor(real_batch[0])[:64], padding=5, normalize=True), "real images")
        history = torch.zeros(max(n_iter, 1), len(SCALAR_NAMES), device=self.device)     # every step's scalars, on the device
        reduce = self.reducer.start if self.reducer else None
        start = time.time()
        self.logger.debug("train start")
        iters = 0
        for epoch in range(self.epoch):
            if hasattr(getattr(loader, "sampler", None), "set_epoch"):
                loader.sampler.set_epoch(epoch)                   # host-data path: a new shuffle / shard every epoch
            to_dev = lambda d: d[0] if isinstance(d[0], DeviceBatch) else d[0].to(self.device, torch.float32, non_blocking=True).contiguous()
            it, nxt_real = iter(loader), None
            nxt = next(it, None)
            for i in range(len(loader)):
                if nxt is None:
                    break
                real = nxt_real if nxt_real is not None else to_dev(nxt)
                nxt = next(it, None)
                # the next batch is announced to the step, which runs the forward half of its D(real) pass under G's gradient
                # all-reduce (data parallel) / beside Adam(G) and the repack (one GPU): hipgan/engine.py step_async next_real;
                # same-sized batches only (one engine per size)
                nxt_real = to_dev(nxt) if (nxt is not None and not self.host_rng) else None
                if nxt_real is not None and nxt_real.size(0) != real.size(0):
                    announce = None
                else:
                    announce = nxt_real
                eng = self._engine_for(real.size(0))
                noise = None
                if self.host_rng:           # reference order: train/dcgan_trainer.py:160,168,171,111
                    b, hg = real.size(0), self.host_gen
                    noise = {"n1": torch.randn(b, 3, 64, 64, generator=hg), "z": torch.randn(b, 100, 1, 1, generator=hg),
                             "n2": torch.randn(b, 3, 64, 64, generator=hg), "alpha": torch.rand(b, 1, 1, 1, generator=hg)}
                elif self.noise_gen is not None:
                    noise = eng.draw_noise(self.noise_gen, fast=eng.fast_noise)
                eng.step_async(real, noise, self.optimizer_d.lr, reduce_d=reduce, reduce_g=reduce, grad_scale=1.0 / self.world,
                               next_real=announce)
                eng.record_scalars(history[iters])
                if i % LOG_EVERY == 0:
                    self._finish_eval(best, wait=False)              # host part of a finished evaluation (scores, checkpoint)
                    s = eng.scalars()                                # the only host sync of the iteration
                    self.logger.debug(f"[{epoch}/{self.epoch}][{i}/{len(loader)}]\tloss_d: {s['loss_d']:.4f}\tloss_g: {s['loss_g']:.4f}"
                                      + f"\tD(x): {s['d_x']:.4f}\tD(G(z)): {s['d_gz1']:.4f} / {s['d_gz2']:.4f}")
                at_eval = (iters % EVAL_EVERY == 0) or ((epoch == self.epoch - 1) and (i == len(loader) - 1))
                if self.guard is not None and (iters == 2 or (at_eval and iters > 2)):
                    self.guard.check(f"after iteration {iters} ")
                if at_eval:
                    self._evaluate(fixed_noise, iters, best)
                iters += 1
        self._finish_eval(best, wait=True)
        self.engine.join()
        torch.cuda.synchronize()
        self.engine.check()
        end = time.time()
        self.logger.debug(f"train finish\ttiem: {time_to_str(end - start)}")
        hist = history[:iters].cpu()
        self.losses_d, self.losses_g = hist[:, 0].tolist(), hist[:, 1].tolist()
        if self.rank == 0:
            self._plot_losses()
        return self.losses_d, self.losses_g

    def _plot_losses(self):
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            plt.clf()
            plt.figure(figsize=(8, 6))
            x = range(1, len(self.losses_g) + 1)
            plt.plot(x, self.losses_d, label="Discriminator Loss")
            plt.plot(x, self.losses_g, label="Generator Loss")
            plt.title("Discriminator and Generator Loss")
            plt.xlabel("Iterations")
            plt.ylabel("Loss")
            plt.legend()
            plt.savefig(os.path.join(self.model_save_path, "loss.png"))
            plt.close("all")
        except Exception as e:
            self.logger.warning(f"could not write loss.png: {e}")
