"""Data-parallel gradient exchange: one flat fp32 all-reduce per network over RCCL (torch.distributed "nccl").

The all-reduce is issued asynchronously: ProcessGroupNCCL runs it on its own HIP stream after the work already queued
on the compute stream, so kernels enqueued afterwards (the gradient-penalty pass, which produces no gradients in
DCGAN) overlap it on xGMI.  `start(flat)` returns a callable that makes the compute stream wait for the result.
Gradients are SUM-reduced; the 1/world factor is folded into the Adam kernel (grad_scale)."""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, world, group=None):
        self.world = world
        self.group = group

    def start(self, flat):
        if self.world == 1:
            return None
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return work.wait
