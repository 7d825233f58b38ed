"""Data-parallel collectives for the native trainer: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

  * ``allreduce_flat_grads``  -- ONE all-reduce of the flat fp32 gradient buffer per optimizer
    step (the reference's DDP reduces 597 MB after every micro-batch, ref:train_v33_ddp.py:363-374
    has no ``no_sync``), issued on a side stream so it overlaps whatever the compute stream still
    has queued; averaged like DDP.
  * ``all_gather_with_grad``  -- all-gather of the positive document vectors for cross-GPU
    in-batch negatives (BASELINE config 4; not in the reference); backward = reduce-scatter(sum).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

_side_stream = {}


def world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def allreduce_flat_grads(flat: torch.Tensor, bucket_mb: int = 0) -> None:
    """Average `flat` across ranks in place.  On GPU the collective(s) run on a dedicated stream
    ordered after the gradient producers; the compute stream waits on them before the optimizer.
    bucket_mb > 0 splits the buffer (xGMI is point-to-point: a few large transfers keep all
    links busy; the default is one collective)."""
    w = world()
    if w == 1:
        return
    if flat.is_cuda:
        dev = flat.device
        side = _side_stream.get(dev)
        if side is None:
            side = _side_stream[dev] = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            if bucket_mb and bucket_mb > 0:
                step = bucket_mb * 1024 * 1024 // 4
                for o in range(0, flat.numel(), step):
                    dist.all_reduce(flat[o:o + step], op=dist.ReduceOp.AVG)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.AVG)
        cur.wait_stream(side)
        flat.record_stream(side)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(w)


class _AllGatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: torch.Tensor):
        w = world()
        out = torch.empty((w * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.contiguous())
        ctx.rows = x.shape[0]
        return out

    @staticmethod
    def backward(ctx, g: torch.Tensor):
        w = world()
        g = g.contiguous()
        if g.is_cuda:
            out = torch.empty((ctx.rows,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
            dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM)
            return out
        dist.all_reduce(g, op=dist.ReduceOp.SUM)          # gloo has no reduce_scatter
        r = rank()
        return g[r * ctx.rows:(r + 1) * ctx.rows].clone()


def all_gather_with_grad(x: torch.Tensor) -> torch.Tensor:
    """[B, V] per rank -> [world*B, V] (rank-major), differentiable."""
    if world() == 1:
        return x
    return _AllGatherRows.apply(x)
