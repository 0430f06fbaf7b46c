"""Evaluation branch of the training loop on a side HIP stream (reference train/dcgan_trainer.py:198-221,
train/cgan_trainer.py:222-252).

The reference stops training every 500 iterations: G(fixed_noise), resize to 299x299, two or three Inception-v3 passes, numpy /
scipy on the host, torch.save.  Here the device part - sampling (ONE train-mode BatchNorm batch, as in the reference), the
fused resize + normalise, the metric network, the copies to pinned host memory - is enqueued on a second stream, and the
training stream waits only for the SAMPLING kernels and the checkpoint snapshot's device copies (they read G's weights and move
its BatchNorm running statistics, so the next step must come after them, exactly where the reference has them).  The host part (softmax / KL, fp64 mean-cov read-back,
scipy sqrtm, log line, checkpoint) runs when the copies have landed - checked at the next log points, forced before the next
evaluation and at the end of training.  What is written is a snapshot taken at the evaluation iteration, so the checkpoint
holds the same state the reference would have saved."""
import torch


def to_host_async(t):
    """device tensor -> pinned host tensor, copy enqueued on the current stream."""
    if not t.is_cuda:
        return t
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t, non_blocking=True)
    return h


def checkpoint_snapshot(trainer):
    """What save_model writes (train/dcgan_trainer.py:86-91), copied on the device NOW: the live state moves on while the
    scores of this evaluation are still being computed."""
    return {"model_g": {k: v.detach().clone() for k, v in trainer.model_g.state_dict().items()},
            "model_d": {k: v.detach().clone() for k, v in trainer.model_d.state_dict().items()},
            "optimizer_g": trainer.optimizer_g.state_dict(), "optimizer_d": trainer.optimizer_d.state_dict()}


def snapshot_to_cpu(snap):
    return {"model_g": {k: v.cpu() for k, v in snap["model_g"].items()}, "model_d": {k: v.cpu() for k, v in snap["model_d"].items()},
            "optimizer_g": snap["optimizer_g"], "optimizer_d": snap["optimizer_d"]}


class AsyncEval:
    def __init__(self, trainer):
        self.tr = trainer
        self.stream = None
        self.pending = None

    def launch(self, iters, sample, device_part):
        """sample() -> fake images on the device (reads and moves G's state); device_part(fake) -> dict of device tensors to
        bring to the host.  Both run on the side stream; the caller's stream waits for `sample` only."""
        tr = self.tr
        main = torch.cuda.current_stream()
        tr.engine.join()                                   # the last training step may still be in flight on the engine's streams
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=tr.device)
        self.stream.wait_stream(main)
        with torch.cuda.stream(self.stream):
            fake = sample()
            snap = checkpoint_snapshot(tr) if tr.rank == 0 else None      # device copies, BEFORE training may touch the state again
            sampled = torch.cuda.Event()
            sampled.record(self.stream)
            host = {k: to_host_async(v) for k, v in device_part(fake).items()}
            done = torch.cuda.Event()
            done.record(self.stream)
        main.wait_event(sampled)                           # training resumes behind sampling + snapshot, not behind the metric network
        self.pending = {"iters": iters, "snapshot": snap, "host": host, "done": done}

    def take(self, wait):
        """-> the pending evaluation once its device part has finished (None if nothing is pending / not finished and not wait)."""
        p = self.pending
        if p is None or (not wait and not p["done"].query()):
            return None
        p["done"].synchronize()
        self.pending = None
        return p
