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
    def __init__(self, world, group=None, force=False, native=None):
        """force: issue the collectives even for world == 1 (tests: a one-rank RCCL group runs the real ProcessGroupNCCL
        stream path, and SUM over one rank must leave every bit unchanged).
        native (default: the environment's JCK_RCCL_NATIVE == "1"): the collectives go through the library's own RCCL
        communicator (include/jckgan.h jck_comm_*: {init, enqueue, wait} on a stream of its own, ordered by events) instead of
        torch.distributed.all_reduce; torch.distributed only carries the 128-byte communicator id to the ranks.  Opt-in:
        like every N > 1 path of this build it has run against one device only."""
        import os
        self.world = world
        self.group = group
        self.force = force
        self.native = (os.environ.get("JCK_RCCL_NATIVE") == "1") if native is None else bool(native)
        self._comm = None
        if self.native and (world > 1 or force):
            self._comm = self._create_comm()

    def _create_comm(self):
        import ctypes as C
        from ._lib import lib
        dev = torch.device("cuda", torch.cuda.current_device())
        ident = torch.zeros(128, dtype=torch.uint8)
        rank = dist.get_rank(self.group) if (dist.is_available() and dist.is_initialized()) else 0
        if rank == 0:
            buf = (C.c_ubyte * 128)()
            lib.jck_comm_unique_id(buf)
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        if self.world > 1:
            # the id travels over the process group that exists anyway (device tensors under "nccl", host tensors under "gloo")
            on_dev = dist.get_backend(self.group) == "nccl"
            t = ident.to(dev) if on_dev else ident
            dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            ident = t.cpu()
        raw = (C.c_ubyte * 128)(*ident.tolist())
        out = C.c_void_p()
        lib.jck_comm_create(C.byref(out), raw, self.world, rank)
        return out

    def close(self):
        if self._comm is not None:
            from ._lib import lib
            lib.jck_comm_destroy(self._comm)
            self._comm = None

    def start(self, flat):
        """All-reduce (SUM) of a flat gradient arena (or a slice of one); returns a callable that makes the current stream wait
        for it, or None when there is nothing to exchange."""
        if self.world == 1 and not self.force:
            return None
        if self._comm is not None:
            import ctypes as C
            from ._lib import lib
            if flat.dtype != torch.float32 or not flat.is_contiguous() or not flat.is_cuda:
                raise ValueError("native RCCL all-reduce takes a contiguous float32 device tensor")
            ticket = C.c_int(-1)
            lib.jck_comm_allreduce_enqueue(self._comm, flat.data_ptr(), flat.numel(), torch.cuda.current_stream().cuda_stream,
                                           C.byref(ticket))
            comm, tk = self._comm, ticket.value
            return lambda: lib.jck_comm_wait(comm, tk, torch.cuda.current_stream().cuda_stream)
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
