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


_staging_logged = False


def rccl() -> bool:
    """True when the default group can move CUDA tensors through RCCL: backend "nccl", or a composite such as
    "cpu:gloo,cuda:nccl" (what a plain `init_process_group()` creates).  Device tensors under any other backend (gloo:
    `tests/test_gpu_dist.py` runs two ranks on ONE GPU that way, which RCCL refuses) are exchanged through host
    copies made on the exchange stream -- same ordering, no device collectives; that choice is logged once, because it
    is correct but an order of magnitude slower than the collective."""
    global _staging_logged
    ok = "nccl" in str(dist.get_backend()).lower()
    if not ok and not _staging_logged and torch.cuda.is_available():
        _staging_logged = True
        import logging
        logging.getLogger(__name__).warning(
            "snx.dist: process group backend %r has no RCCL for device tensors; gradient buckets and gathered "
            "positives are staged through host memory (rehearsal mode)", dist.get_backend())
    return ok


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
            if not rccl():
                host = flat.cpu()                        # stream-ordered copy: waits for the producers `side` waits for
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                flat.copy_(host.div_(w))
            elif bucket_mb and bucket_mb > 0:
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
    embeddings] last; each bucket is one contiguous slice -> one collective on the exchange stream, which
    the native call makes wait for the producers (launch stream and weight-gradient stream).  The compute
    stream never waits until `wait()` (before clip + AdamW).  xGMI is point-to-point: few large collectives
    keep all 7 links busy, so the default is 4 buckets of ~120-230 MB rather than DDP's 25 MB.

    Which backward exchanges.  A micro-step may be ONE native backward (fused passes) or THREE (query /
    positive / negative forwards, the reference's call pattern ref:train_v33_ddp.py:339-343 -> one reduction at
    :364).  All of them add into the same flat buffer, so a slice may be reduced only by the LAST backward that
    writes it: `arm()` opens a micro-step, every saving forward made while armed takes a token (`on_forward`),
    every backward hands it back (`claim_backward`) and only the one that returns the last outstanding token
    runs the bucketed exchange.  A backward whose forward predates `arm()` disarms (the caller's
    `sync_gradients()` then reduces the whole buffer), and a backward that arrives between a finished exchange
    and `wait()` raises: it would add local gradients to slices that are already averaged.

    `mode`: "allreduce" (one all_reduce(AVG) per bucket, RCCL picks the algorithm) or "rs_ag" (reduce-scatter +
    all-gather in place: each rank reduces 1/world of the bucket over its 7 direct xGMI links, then the shards
    are gathered -- the direct, non-ring form).  Env SNX_GRAD_EXCHANGE selects it."""

    def __init__(self, device, n_buckets: int = 4, mode: Optional[str] = None):
        self.device = torch.device(device)
        self.n_buckets = max(1, int(n_buckets))
        self.mode = mode or os.environ.get("SNX_GRAD_EXCHANGE", "allreduce")
        if self.mode not in ("allreduce", "rs_ag"):
            raise ValueError(f"SNX_GRAD_EXCHANGE must be 'allreduce' or 'rs_ag', got {self.mode!r}")
        # CPU tensors (gloo, host-logic tests) have no streams: the collectives then simply run in program order
        self.stream = exchange_stream(self.device) if self.device.type == "cuda" else None
        self.armed = False
        self.pending = False
        self.epoch = 0                                   # one per arm(): tokens of older forwards do not count
        self.outstanding = 0                             # saving forwards of this micro-step not yet back-propagated
        self.slices: List[Tuple[int, int]] = []          # what the last armed backward reduced (tests)
        self.log: List[tuple] = []                       # ("fwd"|"bwd"|"exchange"|"reduce", ...) call order (tests)
        self.keep_log = os.environ.get("SNX_GRAD_SYNC_LOG", "0") == "1"
        # CUs the persistent GEMM kernels (one workgroup per CU) leave to RCCL's channel workgroups while the exchange
        # overlaps the backward (include/snx.h snx_set_reserved_cus).  OPT-IN (default 0 = none): no multi-GPU node was
        # available in rounds 1-5, so whether a 224-workgroup launch (-12.5 % GEMM rate in the armed backward) beats a
        # 256-workgroup launch that finds a few CUs held by RCCL is UNMEASURED.  A/B on a node:
        # SNX_EXCHANGE_RESERVED_CUS=0 / 16 / 32.  A non-zero value changes the token partition of the weight-gradient
        # GEMM (another fp32 summation tree): gradients then differ from the default's in the last bits.
        self.reserved_cus = int(os.environ.get("SNX_EXCHANGE_RESERVED_CUS", "0"))

    def _note(self, *ev) -> None:
        if self.keep_log:
            self.log.append(ev)

    def unit_ranges(self, n_units: int) -> List[Tuple[int, int]]:
        """Split units [0, n_units) into n_buckets consecutive ranges; the tail unit rides with the first
        layers' bucket and the embedding unit with the last."""
        layers = n_units - 2
        nb = min(self.n_buckets, max(1, layers))
        cuts = [1 + (layers * i) // nb for i in range(nb + 1)]
        cuts[0], cuts[-1] = 0, n_units
        return [(cuts[i], cuts[i + 1]) for i in range(nb) if cuts[i] < cuts[i + 1]]

    def arm(self, on: bool = True) -> None:
        """Opens a micro-step.  on=True: it is the last of its accumulation window -- the last backward of the
        forwards made from here on exchanges while it runs."""
        if self.pending:
            raise RuntimeError("arm() between an overlapped gradient exchange and sync_gradients(): the averaged "
                               "gradients have not been consumed yet")
        self.armed = bool(on) and active()
        self.epoch += 1
        self.outstanding = 0
        if self.armed:
            self.slices = []

    def on_forward(self) -> Optional[int]:
        """A forward that saves for backward, made while armed: returns the token its backward hands back."""
        if not self.armed:
            return None
        self.outstanding += 1
        self._note("fwd", self.epoch, self.outstanding)
        return self.epoch

    def claim_backward(self, token: Optional[int]) -> bool:
        """Called by every backward that adds into the flat buffer.  True: this is the last backward of the
        armed micro-step -> run the bucketed exchange inside it."""
        if self.pending:
            raise RuntimeError("a backward ran after the overlapped gradient exchange of this window and before "
                               "sync_gradients(): its gradients would be added to already averaged slices")
        if not self.armed:
            self._note("bwd", token, False)
            return False
        if token != self.epoch:
            # graph built before arm() (a custom loop that armed too late): no overlap, reduce everything later
            self.armed = False
            self._note("bwd", token, False)
            return False
        self.outstanding -= 1
        last = self.outstanding == 0
        self._note("bwd", token, last)
        return last

    def _reduce(self, t: torch.Tensor) -> None:
        """Average the 1-D fp32 slice `t` over ranks in place (current stream / program order)."""
        w = world()
        if t.is_cuda and not rccl():                     # host-staged (see rccl())
            host = t.cpu()
            self._reduce(host)
            t.copy_(host)
            return
        gpu = t.is_cuda
        if self.mode == "rs_ag" and t.numel() >= w:
            chunk = t.numel() // w
            body, r = t[:chunk * w], rank()
            mine = body[r * chunk:(r + 1) * chunk]
            if gpu:                                      # RCCL: in-place forms (shard r of the buffer is rank r's)
                dist.reduce_scatter_tensor(mine, body, op=dist.ReduceOp.AVG)
                dist.all_gather_into_tensor(body, mine)
            else:                                        # gloo has neither: one reduce per owner, then all_gather
                for o in range(w):
                    dist.reduce(body[o * chunk:(o + 1) * chunk], dst=o, op=dist.ReduceOp.SUM)
                mine.div_(w)
                parts = [torch.empty_like(mine) for _ in range(w)]
                dist.all_gather(parts, mine.clone())
                for o in range(w):
                    body[o * chunk:(o + 1) * chunk].copy_(parts[o])
            t = t[chunk * w:]                            # < world elements left over
            if t.numel() == 0:
                return
        if gpu:
            dist.all_reduce(t, op=dist.ReduceOp.AVG)
        else:                                            # gloo has no AVG
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            t.div_(w)

    def reduce_slice(self, flat: torch.Tensor, lo: int, hi: int) -> None:
        self._note("reduce", lo, hi)
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                self._reduce(flat[lo:hi])
        else:
            self._reduce(flat[lo:hi])
        self.slices.append((lo, hi))

    def run_backward(self, token: Optional[int], flat: torch.Tensor, n_units: int, run_all, run_units,
                     param_range) -> None:
        """The sequencing every native backward goes through (EncoderRuntime.backward_impl; the 2-rank gloo
        test drives it with a stand-in for the kernels).  run_all(): the whole backward in one native call;
        run_units(ub, ue): units [ub, ue), after which the exchange stream has been made to wait for their
        producers; param_range(ub, ue) -> flat slices complete after those units."""
        if not self.claim_backward(token):
            run_all()
            return
        self._note("exchange", self.epoch)
        for ub, ue in self.unit_ranges(n_units):
            run_units(ub, ue)
            for lo, hi in param_range(ub, ue):
                self.reduce_slice(flat, lo, hi)
        self.finished_backward()

    def finished_backward(self) -> None:
        self.armed = False
        self.pending = True

    def wait(self, flat: torch.Tensor) -> bool:
        """Compute stream waits for the exchange; False when no overlapped exchange was done."""
        if not self.pending:
            self.armed = False                           # an armed micro-step whose backward never came
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
        ctx.rows = x.shape[0]
        if x.is_cuda and not rccl():                     # host-staged (see rccl())
            host = torch.empty((w * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype)
            dist.all_gather_into_tensor(host, x.contiguous().cpu())
            return host.to(x.device)
        out = torch.empty((w * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.contiguous())
        return out

    @staticmethod
    def backward(ctx, g: torch.Tensor):
        w = world()
        g = g.contiguous()
        if g.is_cuda and rccl():
            out = torch.empty((ctx.rows,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
            dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM)
            return out
        dev = g.device
        g = g.cpu() if g.is_cuda else g
        dist.all_reduce(g, op=dist.ReduceOp.SUM)          # gloo has no reduce_scatter
        r = rank()
        return g[r * ctx.rows:(r + 1) * ctx.rows].clone().to(dev)


def all_gather_with_grad(x: torch.Tensor) -> torch.Tensor:
    """[B, V] per rank -> [world*B, V] (rank-major), differentiable.  Inline on the current stream."""
    if not active():
        return x
    return _AllGatherRows.apply(x)


class _AllGatherRowsOnExchangeStream(torch.autograd.Function):
    """_AllGatherRows with the collective issued on the EXCHANGE stream (collective 2: "issue on a side
    stream after the positive pass so it hides under the negative pass").  forward() orders the exchange stream behind
    what the compute stream has queued so far (the positive pass that produced `x`), issues the all-gather there and
    returns at once: the compute stream goes on with the negative pass.  The result may be READ on the compute stream only
    after `PendingGather.wait()` -- the one place the compute stream waits, right in front of the loss kernels.  Backward:
    the reduce-scatter of the gathered gradient, inline (its result is needed at once by the positive pass's backward)."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, box: list):
        w = world()
        ctx.rows = x.shape[0]
        x = x.contiguous()
        dev = x.device
        out = torch.empty((w * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=dev)   # compute stream's allocation
        side = exchange_stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            if rccl():
                dist.all_gather_into_tensor(out, x)
            else:                                         # host-staged rehearsal (see rccl())
                host = torch.empty(tuple(out.shape), dtype=x.dtype)
                dist.all_gather_into_tensor(host, x.cpu())
                out.copy_(host)
            box.append(side.record_event())
        x.record_stream(side)
        out.record_stream(side)
        return out

    @staticmethod
    def backward(ctx, g: torch.Tensor):
        return _AllGatherRows.backward(ctx, g), None


class PendingGather:
    """An all-gather in flight on the exchange stream; `wait()` makes the current stream wait for it and hands out the
    gathered, differentiable tensor."""

    def __init__(self, out: torch.Tensor, event):
        self._out, self._event = out, event

    def wait(self) -> torch.Tensor:
        if self._event is not None:
            torch.cuda.current_stream(self._out.device).wait_event(self._event)
            self._event = None
        return self._out


def all_gather_with_grad_async(x: torch.Tensor) -> PendingGather:
    """`all_gather_with_grad` whose collective runs on the exchange stream while the caller's stream continues; CPU tensors
    (gloo host tests) and inactive groups degrade to the inline form.  Same values, same gradient as the inline form."""
    if not active() or not x.is_cuda:
        return PendingGather(all_gather_with_grad(x), None)
    box: list = []
    out = _AllGatherRowsOnExchangeStream.apply(x, box)
    return PendingGather(out, box[0])
