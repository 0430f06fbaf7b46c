"""Data-parallel gradient exchange: one flat fp32 all-reduce per network over RCCL (torch.distributed "nccl").

The all-reduce is issued asynchronously: ProcessGroupNCCL runs it on its own HIP stream after the work already queued
on the issuing stream.  In the batched schedule the tail of D's arena (conv4.weight .. conv5.weight, 76 % of the bytes) is
final right after the first weight-gradient product of the backward pass: the engine records an event there
(jck_engine_grad_bucket) and that slice is reduced from a side stream under the rest of the backward; in the per-pass
schedule the gradient-penalty pass (no gradients in DCGAN) runs under the whole all-reduce.  `start(flat)` returns a callable that makes the compute stream wait for the result.
Gradients are SUM-reduced; the 1/world factor is folded into the Adam kernel (grad_scale)."""
import os

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, world, group=None, force=False):
        """force: issue the collectives even for world == 1 (tests: a one-rank RCCL group runs the real ProcessGroupNCCL
        stream path, and SUM over one rank must leave every bit unchanged)."""
        self.world = world
        self.group = group
        self.force = force
        self._comm = None

    def start(self, flat, early=None):
        """All-reduce (SUM) of a flat gradient arena; returns a callable that makes the current stream wait for it.
        early = (offset, wait_on): flat[offset:] is already final once `wait_on(stream_handle)` has made a stream wait for
        the engine's bucket event - that slice is reduced on a side stream at once, under the rest of the backward pass,
        and only flat[:offset] waits for the end of the phase.  EXPERIMENTAL, off by default (JCK_DDP_BUCKETS=1 enables):
        the ordering of the two collectives on ProcessGroupNCCL's stream is exercised by tests/test_ddp_gpu.py through a
        world-size-1 RCCL group only - no run on two or more devices has validated it yet (no multi-GPU node was available
        to the build)."""
        if self.world == 1 and not self.force:
            return None
        works = []
        if early is not None and os.environ.get("JCK_DDP_BUCKETS", "0") == "1":
            off, wait_on = early
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=flat.device)
            wait_on(self._comm.cuda_stream)
            with torch.cuda.stream(self._comm):
                works.append(dist.all_reduce(flat[off:], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            works.append(dist.all_reduce(flat[:off], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

        def wait():
            for w in works:
                w.wait()
        return wait
