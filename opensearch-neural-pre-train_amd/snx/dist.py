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

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

_side_stream = {}


def world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def active() -> bool:
    """Collectives are issued when more than one rank exists -- or, with SNX_DIST_FORCE=1, whenever a process
    group is initialised (lets a single-GPU box drive all_reduce / all_gather / reduce_scatter through RCCL)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("SNX_DIST_FORCE", "0") == "1"


def exchange_stream(dev) -> "torch.cuda.Stream":
    side = _side_stream.get(dev)
    if side is None:
        side = _side_stream[dev] = torch.cuda.Stream(device=dev)
    return side


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def allreduce_flat_grads(flat: torch.Tensor, bucket_mb: int = 0) -> None:
    """Average `flat` across ranks in place.  On GPU the collective(s) run on a dedicated stream
    ordered after the gradient producers; the compute stream waits on them before the optimizer.
    bucket_mb > 0 splits the buffer (xGMI is point-to-point: a few large transfers keep all
    links busy; the default is one collective)."""
    w = world()
    if not active():
        return
    if flat.is_cuda:
        dev = flat.device
        side = exchange_stream(dev)
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


class BucketedGradSync:
    """Gradient exchange overlapped with the backward (the role of torch DDP's bucketed reducer,
    ref:src/train/cli/train_v33_ddp.py:539-544, for the native runtime).

    The native backward runs as `n_buckets` consecutive unit ranges (include/snx.h
    snx_model_backward_units).  Buckets follow COMPLETION order, which is also address order in the flat
    gradient buffer (canonical parameter order = forward order): [tail + last layers] first, [first layers +
    embeddings] last; each bucket is one contiguous slice -> one all-reduce(AVG) on the exchange stream, which
    the native call makes wait for the producers (launch stream and weight-gradient stream).  The compute
    stream never waits until `wait()` (before clip + AdamW).  xGMI is point-to-point: few large collectives
    keep all 7 links busy, so the default is 4 buckets of ~120-230 MB rather than DDP's 25 MB."""

    def __init__(self, device, n_buckets: int = 4):
        self.device = torch.device(device)
        self.n_buckets = max(1, int(n_buckets))
        # CPU tensors (gloo, host-logic tests) have no streams: the collectives then simply run in program order
        self.stream = exchange_stream(self.device) if self.device.type == "cuda" else None
        self.armed = False
        self.pending = False
        self.slices: List[Tuple[int, int]] = []          # what the last armed backward reduced (tests)

    def unit_ranges(self, n_units: int) -> List[Tuple[int, int]]:
        """Split units [0, n_units) into n_buckets consecutive ranges; the tail unit rides with the first
        layers' bucket and the embedding unit with the last."""
        layers = n_units - 2
        nb = min(self.n_buckets, max(1, layers))
        cuts = [1 + (layers * i) // nb for i in range(nb + 1)]
        cuts[0], cuts[-1] = 0, n_units
        return [(cuts[i], cuts[i + 1]) for i in range(nb) if cuts[i] < cuts[i + 1]]

    def arm(self, on: bool = True) -> None:
        """The NEXT backward is the last of its accumulation window: exchange while it runs."""
        self.armed = bool(on) and active()
        if self.armed:
            self.slices = []

    def reduce_slice(self, flat: torch.Tensor, lo: int, hi: int) -> None:
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.AVG)
        else:                                            # gloo has no AVG
            dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM)
            flat[lo:hi].div_(world())
        self.slices.append((lo, hi))

    def finished_backward(self) -> None:
        self.armed = False
        self.pending = True

    def wait(self, flat: torch.Tensor) -> bool:
        """Compute stream waits for the exchange; False when no overlapped exchange was done."""
        if not self.pending:
            return False
        if self.stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
            flat.record_stream(self.stream)
        self.pending = False
        return True


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
    if not active():
        return x
    return _AllGatherRows.apply(x)
