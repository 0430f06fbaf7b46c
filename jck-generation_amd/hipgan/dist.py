"""Data-parallel gradient exchange: one flat fp32 all-reduce per network over RCCL (torch.distributed "nccl").

The all-reduce is issued asynchronously: ProcessGroupNCCL runs it on its own HIP stream after the work already queued on the
issuing stream; `start(flat)` returns a callable that makes the compute stream wait for the result.  Gradients are SUM-reduced;
the 1/world factor is folded into the Adam kernel (grad_scale).  Where the collectives sit in the step (D's arena in two pieces
under D's own backward, G's under the next batch's D(real) forward) is the engine's business: hipgan/engine.py step_async.

ReplicaGuard is the safety net under that overlap (the exchange points are train/dcgan_trainer.py:180,189 of the reference,
under DistributedDataParallel): every rank must hold the same parameters after a step - the all-reduced gradients went through
the same Adam - so a checksum whose MAX and MIN over the ranks differ means the schedule misbehaved on this machine.  The
overlapped schedule has been bit-tested against the plain order with one RCCL rank and with two gloo ranks, never on two
devices; a trainer must not train on silently diverged replicas if real xGMI disagrees."""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, world, group=None, force=False):
        """force: issue the collectives even for world == 1 (tests: a one-rank RCCL group runs the real ProcessGroupNCCL
        stream path, and SUM over one rank must leave every bit unchanged)."""
        self.world = world
        self.group = group
        self.force = force

    def start(self, flat):
        """All-reduce (SUM) of a flat gradient arena (or a slice of one); returns a callable that makes the current stream wait
        for it, or None when there is nothing to exchange."""
        if self.world == 1 and not self.force:
            return None
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return work.wait


class ReplicaGuard:
    """check(): True when every rank holds the same parameters.  On a mismatch: rank 0's state (parameters, Adam moments,
    BatchNorm buffers) is broadcast again, the engines fall back to the plain schedule (one all-reduce per network, waited for
    before its Adam: engine.ddp_overlap = False) and the event is logged once per occurrence - the run continues on consistent
    replicas instead of training on diverged ones."""
    KEYS = ("g_params", "d_params", "g_m", "g_v", "d_m", "d_v", "g_bn", "d_bn")

    def __init__(self, engines, world, group=None, log=None):
        self.engines = engines if callable(engines) else (lambda e=engines: [e])      # callable -> the engines in use (one per batch size)
        self.world, self.group, self.log = world, group, log
        self.mismatches = 0

    def in_sync(self):
        if self.world <= 1:
            return True
        eng = self.engines()[0]
        eng.join()
        a = eng.arenas
        checks = [e for e in self.engines() if hasattr(e, "check")]
        c = torch.stack([a["d_params"].double().sum(), a["g_params"].double().sum(),
                         a["d_params"].double().abs().sum(), a["g_params"].double().abs().sum()])
        hi, lo = c.clone(), c.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        same = bool(torch.equal(hi, lo))          # (host sync)
        for e in checks:
            e.check()                             # a grid-barrier timeout on this rank: raise rather than train on
        return same

    def check(self, where=""):
        if self.in_sync():
            return True
        self.mismatches += 1
        engs = self.engines()
        was_overlapped = any(getattr(e, "ddp_overlap", False) for e in engs)
        first = engs[0]
        for key in self.KEYS:
            if key in first.arenas:
                dist.broadcast(first.arenas[key], src=0, group=self.group)        # (engines of one trainer share their arenas)
        for e in engs:
            e.ddp_overlap = False
            e.mark_weights_changed()
        if self.log:
            self.log(f"data parallel: the replicas' parameters differ {where}- state re-broadcast from rank 0"
                     + ("; falling back to one all-reduce per network, waited for before its Adam (the overlapped schedule failed the check)"
                        if was_overlapped else " (plain schedule already in use: check the interconnect / collectives)"))
        return False
