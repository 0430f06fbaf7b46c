"""CGAN trainer for MI355X - drop-in for the reference's train/cgan_trainer.py: `CGANTrainer(args, model_g, model_d, data_pre)`,
`train()`, `save_model(typ, iters, inception_score, fid, intra_fid, images)`, `save_image(path, iters, images)`,
`compute_gradient_penalty(real, fake, labels)`; same log line and checkpoint dict.  One iteration (reference :173-213) is the
native engine schedule of family 1: D(real), G(z,l), D(fake), the gradient penalty INCLUDING its back-propagation (double
backward, closed form), Adam(D), D(fake) again, backward into G, Adam(G)."""
import os
import time

import numpy as np
import torch
import torch.nn as nn

from hipgan.dist import GradReducer, ReplicaGuard
from hipgan.engine import SCALAR_NAMES, CganEngine, DeviceBatch
from hipgan.optim import EngineAdam
from logger.main_logger import MainLogger
from logger.utils import time_to_str
from model.CGAN import weights_init
from train.dcgan_trainer import EVAL_EVERY, LOG_EVERY, DCGANTrainer, _as_tensor, _make_grid, _save_png, inception_input
from train.async_eval import AsyncEval, snapshot_to_cpu
from train.trainer import Trainer
from utils import require_gpu


class CGANTrainer(DCGANTrainer):
    def __init__(self, args, model_g, model_d, data_pre, prec=None, host_rng=None):
        Trainer.__init__(self)
        self.logger = MainLogger(args)
        self.device = require_gpu("CGANTrainer")
        self.epoch, self.max_lr, self.lambda_gp = args.epoch, args.max_learning_rate, 10.0
        self.prec = prec or os.environ.get("JCKGAN_PREC", "bf16")
        self.host_rng = bool(int(os.environ.get("JCKGAN_HOST_RNG", "0"))) if host_rng is None else host_rng
        if self.host_rng:
            model_g.apply(weights_init)
            model_d.apply(weights_init)
        self.model_g, self.model_d = model_g.to(self.device), model_d.to(self.device)
        self.logger.debug(f"Generator: {sum(p.numel() for p in model_g.parameters())} parameters\n{self.model_g}")
        self.logger.debug(f"Discriminator: {sum(p.numel() for p in model_d.parameters())} parameters\n{self.model_d}")
        if not self.host_rng:
            self.model_g.apply(weights_init)
            self.model_d.apply(weights_init)
        self.model_g.prec = self.model_d.prec = self.prec
        self.data_pre = data_pre
        self.train_loader, metric_loader = self.data_pre.get_data_loader()
        self.metric = self._make_metrics(metric_loader)
        self.world, self.rank = 1, 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world, self.rank = torch.distributed.get_world_size(), torch.distributed.get_rank()
        self.batch_size = int(getattr(args, "batch_size", 128))
        self.engine = CganEngine(batch=self.batch_size, prec=self.prec, device=self.device)
        self.engine.adopt_modules(self.model_g, self.model_d)
        if self.world > 1:
            for key in ("g_params", "d_params", "g_bn", "d_bn"):
                torch.distributed.broadcast(self.engine.arenas[key], src=0)
            self.engine.mark_weights_changed()
        self._tail_engines = {}
        self.reducer = GradReducer(self.world) if self.world > 1 else None
        self.guard = (ReplicaGuard(lambda: [self.engine] + list(self._tail_engines.values()), self.world, log=self.logger.debug)
                      if self.world > 1 else None)         # replicas must stay identical: hipgan/dist.py
        # per-rank noise stream in data-parallel runs (every rank is seeded alike by main.py; see DCGANTrainer)
        self.noise_gen = self.host_gen = None
        if self.world > 1:
            from change_randomseed import RANDOMSEED
            self.noise_gen = torch.Generator(device=self.device).manual_seed(RANDOMSEED + 1 + self.rank)
            self.host_gen = torch.Generator().manual_seed(RANDOMSEED + 1 + self.rank)
        from change_randomseed import RANDOMSEED as _SEED     # one GPU too: the seed governs the step's own draws (see DCGANTrainer)
        self._noise_seed = (int(torch.initial_seed()) ^ (_SEED << 20)) + 1 + self.rank
        self.engine.set_noise_seed(self._noise_seed)
        self.optimizer_g = EngineAdam(self.engine, "g", self.model_g.named_parameters(), self.max_lr, betas=[0.5, 0.999])
        self.optimizer_d = EngineAdam(self.engine, "d", self.model_d.named_parameters(), self.max_lr, betas=[0.5, 0.999])
        self.criterion = nn.BCELoss()
        self.model_save_path = args.save_path                       # reference :66
        os.makedirs(self.model_save_path, exist_ok=True)
        self.logger.debug(f"save path: {self.model_save_path}")

    def _engine_for(self, b):
        if b == self.batch_size:
            return self.engine
        if b not in self._tail_engines:
            self._tail_engines[b] = CganEngine(batch=b, share=self.engine)
            self._tail_engines[b].set_noise_seed(self._noise_seed)
        return self._tail_engines[b]

    # ------------------------------------------------------------------------------------------------------
    def save_model(self, typ, iters, inception_score, fid, intra_fid, images, snapshot=None):
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
        torch.save(state, os.path.join(save_path, f"{iters}_{inception_score:.04f}_{fid:.04f}_{intra_fid:.04f}.pt"))
        self.save_image(save_path, iters, images)

    def save_image(self, path, iters, images):
        """10x10 grid, one image per class, titled with the class name (reference :93-103)."""
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            plt.clf()
            fig = plt.figure(figsize=(10, 10))
            for i in range(min(100, images.shape[0])):
                fig.add_subplot(10, 10, i + 1)
                plt.title(self.data_pre.idx_to_labels[i])
                plt.axis("off")
                plt.imshow(np.clip(np.transpose(images[i].numpy(), (1, 2, 0)), 0, 1))
            plt.savefig(os.path.join(path, f"{iters}_fake_image.png"))
            plt.close("all")
        except Exception as e:
            self.logger.warning(f"could not write the class grid: {e}")

    def compute_gradient_penalty(self, real_data, fake_data, labels_data):
        """The reference's penalty (:114-131) as a tensor that can be back-propagated into D's parameters, as :200-203 do
        (`error_d = error_real + error_fake + 10 * gp; error_d.backward()`): hipgan.functional.gradient_penalty, whose backward
        is the engine's closed-form double backward.  train() itself runs the penalty inside the native step."""
        from hipgan import functional as HF
        return HF.gradient_penalty(self.model_d, real_data.detach(), fake_data.detach(), labels=labels_data)

    # ------------------------------------------------------------------------------------------------------
    def _evaluate(self, fixed_noise, fixed_labels, iters, best, image_save_path):
        """Device part of the evaluation (reference :222-252) on a side stream (train/async_eval.py): 1000 images = ONE
        BatchNorm batch as in the reference, the fused 299x299 resize + normalise, ONE pass of the metric network, fp64 mean /
        covariance of the logits (all 1000 and per superclass) - training resumes behind the sampling kernels only."""
        self._finish_eval(best, wait=True)
        eng = self._engine_for(fixed_noise.size(0))
        self._image_save_path = image_save_path

        def device_part(fake):
            out = {"denorm": (0.5 * fake + 0.5)[::10].contiguous()}
            if self.metric is not None:
                logits = self.metric.logits(inception_input(fake))       # :227-231 in one device pass, then the metric network
                out["logits"] = logits
                if logits.is_cuda:
                    out.update(self.metric.fake_stats_device(logits, intra=True))
            return out
        self._eval.launch(iters, lambda: eng.sample(fixed_noise, fixed_labels), device_part)

    def _finish_eval(self, best, wait):
        p = self._eval.take(wait)
        if p is None:
            return
        iters, host, snap = p["iters"], p["host"], p["snapshot"]
        denorm = host["denorm"]
        if "logits" in host:
            if "mu" in host:
                inception_score, fid, intra = self.metric.scores_from_stats(host["logits"], host, intra=True)
            else:
                inception_score, fid, intra = self.metric.scores_from_logits(host["logits"], intra=True)
            self.logger.debug(f"inception score: {inception_score}\tfid: {fid}\tintra fid: {intra}")
            if best["fid"] > fid:
                best["fid"] = fid
                self.logger.debug(f"{iters} lowest fid")
                self.save_model("fid", iters, inception_score, fid, intra, denorm, snap)
            if best["intra"] > intra:
                best["intra"] = intra
                self.logger.debug(f"{iters} lowest intra fid")
                self.save_model("intra_fid", iters, inception_score, fid, intra, denorm, snap)
            if best["is"] < inception_score:
                best["is"] = inception_score
                self.logger.debug(f"{iters} highest is")
                self.save_model("is", iters, inception_score, fid, intra, denorm, snap)
        else:
            self.save_model("latest", iters, 0.0, 0.0, 0.0, denorm, snap)
        if self.rank == 0:
            self.save_image(self._image_save_path, iters, denorm)

    def train(self):
        loader = self.train_loader
        n_iter = self.epoch * len(loader)
        dev = self.device
        # 100 classes x 10 samples (reference :144-153); the draw order matters in host-RNG mode
        noises = [torch.randn(10, 100, 1, 1) if self.host_rng else torch.randn(10, 100, 1, 1, device=dev) for _ in range(100)]
        fixed_noise = torch.vstack(noises).to(dev)
        fixed_labels = torch.nn.functional.one_hot(torch.arange(100).repeat_interleave(10), 100).to(torch.int64).to(dev)
        best = {"fid": 1e10, "intra": 1e10, "is": 0}
        self._eval = AsyncEval(self)
        if self.rank == 0:
            real_batch = next(iter(loader))
            _save_png(os.path.join(self.model_save_path, "real_image.png"), _make_grid(_as_tensor(real_batch[0])[:64], padding=5, normalize=True),
                      "real images")
        image_save_path = os.path.join(self.model_save_path, "img")
        os.makedirs(image_save_path, exist_ok=True)
        history = torch.zeros(max(n_iter, 1), len(SCALAR_NAMES), device=dev)
        reduce = self.reducer.start if self.reducer else None
        start = time.time()
        self.logger.debug("train start")
        iters = 0
        for epoch in range(self.epoch):
            if hasattr(getattr(loader, "sampler", None), "set_epoch"):
                loader.sampler.set_epoch(epoch)                   # host-data path: a new shuffle / shard every epoch
            for i, data in enumerate(loader):
                real = data[0] if isinstance(data[0], DeviceBatch) else data[0].to(dev, torch.float32, non_blocking=True).contiguous()
                labels = data[1].to(dev, torch.int64, non_blocking=True).contiguous()
                b = real.size(0)
                eng = self._engine_for(b)
                if self.host_rng:       # CPU generator in the reference's order (:181,183[dropout],189,192,194,115,118,209)
                    hg = self.host_gen
                    keep = lambda: torch.empty(b, 256).bernoulli_(0.75, generator=hg)
                    noise = {"n1": torch.randn(b, 3, 64, 64, generator=hg)}
                    noise["m1"] = keep()
                    noise["z"] = torch.randn(b, 100, 1, 1, generator=hg)
                    noise["n2"] = torch.randn(b, 3, 64, 64, generator=hg)
                    noise["m2"] = keep()
                    noise["alpha"] = torch.rand(b, 1, 1, 1, generator=hg)
                    noise["m3"] = keep()
                    noise["m4"] = keep()
                    noise["labels"] = labels
                else:
                    noise = eng.draw_noise(self.noise_gen, labels=labels, fast=eng.fast_noise)
                eng.step_async(real, noise, self.optimizer_d.lr, reduce_d=reduce, reduce_g=reduce, grad_scale=1.0 / self.world)
                eng.record_scalars(history[iters])
                if i % LOG_EVERY == 0:
                    self._finish_eval(best, wait=False)              # host part of a finished evaluation (scores, checkpoint)
                    s = eng.scalars()
                    self.logger.debug(f"[{epoch}/{self.epoch}][{i}/{len(loader)}]\tloss_d: {s['loss_d']:.4f}\tloss_g: {s['loss_g']:.4f}"
                                      + f"\tD(x): {s['d_x']:.4f}\tD(G(z)): {s['d_gz1']:.4f} / {s['d_gz2']:.4f}")
                at_eval = (iters % EVAL_EVERY == 0) or ((epoch == self.epoch - 1) and (i == len(loader) - 1))
                if self.guard is not None and (iters == 2 or (at_eval and iters > 2)):
                    self.guard.check(f"after iteration {iters} ")
                if at_eval:
                    self._evaluate(fixed_noise, fixed_labels, iters, best, image_save_path)
                iters += 1
        self._finish_eval(best, wait=True)
        self.engine.join()
        torch.cuda.synchronize()
        self.engine.check()
        self.logger.debug(f"train finish\ttiem: {time_to_str(time.time() - start)}")
        hist = history[:iters].cpu()
        self.losses_d, self.losses_g = hist[:, 0].tolist(), hist[:, 1].tolist()
        if self.rank == 0:
            self._plot_losses()
        return self.losses_d, self.losses_g
