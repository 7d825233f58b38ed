"""Host-side runtime that binds a set of ``nn.Parameter``s to the native encoder entry points
(``snx_model_forward`` / ``snx_model_backward``) and exposes them to autograd.

PyTorch here is plumbing only: device memory (caching allocator), the current stream, and the
autograd graph boundary.  All arithmetic happens in libsnx.so."""
from __future__ import annotations

import ctypes as C
import logging
import os
import weakref
from collections import OrderedDict
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple  # noqa: F401

import torch

from ._lib import check, fn
from .ops import _p, _stream, rope_table

SNX_FWD_SAVE_FOR_BACKWARD = 1
_SEQ_CACHE_ENTRIES = 64
logger = logging.getLogger(__name__)


class ModelDesc(C.Structure):
    _fields_ = [("vocab", C.c_int32), ("hidden", C.c_int32), ("inter", C.c_int32), ("layers", C.c_int32),
                ("heads", C.c_int32), ("head_dim", C.c_int32), ("global_every", C.c_int32), ("window", C.c_int32),
                ("pad_id", C.c_int32), ("reserved0", C.c_int32), ("ln_eps", C.c_float), ("reserved1", C.c_float)]


@dataclass
class EncoderGeometry:
    """The architecture constants the kernels need (field names follow the HF config.json)."""
    vocab_size: int = 50000
    hidden_size: int = 768
    intermediate_size: int = 1152
    num_hidden_layers: int = 22
    num_attention_heads: int = 12
    global_attn_every_n_layers: int = 3
    local_attention: int = 128
    global_rope_theta: float = 160000.0
    local_rope_theta: float = 10000.0
    norm_eps: float = 1e-5
    pad_token_id: int = 49999
    max_position_embeddings: int = 16384

    def desc(self) -> ModelDesc:
        hd = self.hidden_size // self.num_attention_heads
        return ModelDesc(self.vocab_size, self.hidden_size, self.intermediate_size, self.num_hidden_layers,
                         self.num_attention_heads, hd, self.global_attn_every_n_layers, self.local_attention // 2,
                         self.pad_token_id, 0, self.norm_eps, 0.0)

    def bf16_unsupported_reason(self) -> Optional[str]:
        """None when the bf16 kernels' tilings take this geometry, else why not."""
        hd = self.hidden_size // self.num_attention_heads
        if hd != 64 or self.hidden_size % 256 or self.hidden_size > 1024 or self.intermediate_size % 64:
            return ("the bf16 kernels need head_dim == 64, hidden % 256 == 0 (<= 1024), intermediate % 64 == 0; "
                    f"got hidden={self.hidden_size}, heads={self.num_attention_heads}, intermediate={self.intermediate_size}")
        if (2 * self.intermediate_size) % 128 or self.intermediate_size % 128 or self.hidden_size % 128:
            return "the weight-gradient GEMM needs hidden, intermediate multiples of 128"
        return None

    def check_supported(self):
        r = self.bf16_unsupported_reason()
        if r:
            raise ValueError("snx: " + r)

    def check_f32_supported(self):
        hd = self.hidden_size // self.num_attention_heads
        if hd * self.num_attention_heads != self.hidden_size or hd > 64 or hd % 2:
            raise ValueError("snx fp32 path: head_dim must be even and <= 64")

    def n_params(self) -> int:
        return 2 + (5 + 6 * (self.num_hidden_layers - 1)) + 4


class EncoderRuntime:
    """Binds parameters (canonical order, see include/snx.h) to the native forward/backward."""

    def __init__(self, geom: EncoderGeometry, params: Sequence[torch.nn.Parameter]):
        geom.check_f32_supported()             # the bf16 kernels' tighter limits are checked when that path is taken
        self.geom = geom
        self.params = list(params)
        self._desc = geom.desc()
        n = geom.n_params()
        if n != len(self.params):
            raise ValueError(f"expected {n} parameter tensors in canonical order, got {len(self.params)}")
        self._ptr_key = None
        self._ptr_arr = None
        self._wcache: Optional[torch.Tensor] = None
        self._wcache_key = None
        self._rope: Dict[Tuple[int, str], Tuple[torch.Tensor, torch.Tensor]] = {}
        self._seq_cache: "OrderedDict[tuple, tuple]" = OrderedDict()   # LRU, bounded (dynamic padding)
        self.direct_grads = False
        self.flat_grad: Optional[torch.Tensor] = None
        self.flat_param: Optional[torch.Tensor] = None
        # parity tests read the routing of the latest forward from here; off by default because it would
        # keep the multi-GB activation arena alive until the next forward (SNX_KEEP_LAST_CTX=1 or set it)
        self.keep_last_ctx = os.environ.get("SNX_KEEP_LAST_CTX", "0") == "1"
        self.last_ctx = None
        # bucketed gradient exchange overlapped with the backward (snx.dist.BucketedGradSync); armed per call
        self.grad_sync = None
        # micro-step arena (StepArena below): rows / sequences / passes of the micro-step in progress, the capacity learnt from
        # the recent micro-steps (their maxima), the arena being filled
        self.step_arena_on = os.environ.get("SNX_STEP_ARENA", "1") != "0"
        self._step_phase = "bwd"                             # "fwd": forwards of a micro-step are arriving
        self._step_tot = [0, 0, 0]                           # token rows, sequences, passes of the micro-step in progress
        self._step_hist: list = []                           # totals of the last micro-steps
        self._arena: Optional["StepArena"] = None
        # what the arena did, readable by the caller (a pass that leaves the arena costs ~15 % of the micro-step: it must
        # not be silent).  placed: passes that went into an arena; fell_back: passes that ran on the ordinary path although
        # a capacity had been learnt (did not fit / too many passes); zero_filled: placed passes whose output took no part in
        # the loss (their gradient was back-propagated as zeros); stale_dropped: open arenas abandoned because a placed
        # pass died without a backward or the weights changed under them.  The first fallback / zero fill / stale drop of
        # a process also logs one warning.
        self.arena_stats: Dict[str, int] = {"placed": 0, "fell_back": 0, "zero_filled": 0, "stale_dropped": 0}
        self._arena_warned: set = set()

    # ------------------------------------------------------------------ parameter plumbing
    def _device(self):
        d = self.params[0].device
        if d.type != "cuda":
            raise RuntimeError("SPLADEModernBERT (snx backend) runs on an AMD GPU only: move the module to "
                               "cuda:<local_rank>; there is no CPU fallback for the product path")
        return d

    def param_offsets(self) -> List[int]:
        """Element offset of every parameter in the flat (canonical-order) gradient / parameter buffers."""
        offs, off = [], 0
        for p in self.params:
            offs.append(off)
            off += p.numel()
        return offs + [off]

    def unit_param_range(self, unit_begin: int, unit_end: int) -> List[Tuple[int, int]]:
        """Flat element ranges whose gradients are COMPLETE once backward units [unit_begin, unit_end) have run
        (include/snx.h snx_model_backward_units: unit 0 = tail, 1 + i = layer L-1-i, L + 1 = embeddings)."""
        L = self.geom.num_hidden_layers
        offs = self.param_offsets()
        first = lambda l: 2 if l == 0 else 7 + 6 * (l - 1)   # noqa: E731  index of the layer's first parameter
        tail = 7 + 6 * (L - 1)
        out = []
        if unit_begin == 0:
            out.append((offs[tail], offs[tail + 4]))
        lo_unit, hi_unit = max(unit_begin, 1), min(unit_end, L + 1)
        if lo_unit < hi_unit:
            l_hi, l_lo = L - lo_unit, L + 1 - hi_unit
            begin = offs[first(l_lo)]
            end = offs[tail] if l_hi == L - 1 else offs[first(l_hi + 1)]
            if unit_end == L + 2 and l_lo == 0:
                begin = 0                                   # embeddings + their norm sit right below layer 0
            out.append((begin, end))
        elif unit_end == L + 2:
            out.append((0, offs[2]))
        if unit_end == L + 2 and lo_unit < hi_unit and L + 1 - hi_unit != 0:
            out.append((0, offs[2]))
        out.sort()
        merged = [out[0]]
        for lo, hi in out[1:]:                              # adjacent slices -> one collective
            if lo == merged[-1][1]:
                merged[-1] = (merged[-1][0], hi)
            else:
                merged.append((lo, hi))
        return merged

    def _param_ptrs(self):
        key = tuple(p.data_ptr() for p in self.params)
        if key != self._ptr_key:
            for p in self.params:
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise ValueError("parameters must be contiguous fp32 (master weights)")
            self._ptr_arr = (C.c_void_p * len(key))(*key)
            self._ptr_key = key
            self._wcache_key = None
        return self._ptr_arr

    def _weights(self):
        """bf16 weight cache, refreshed when any parameter changed (optimizer step / load)."""
        ptrs = self._param_ptrs()
        dev = self._device()
        key = tuple(p._version for p in self.params)
        if self._wcache is None or self._wcache.device != dev:
            nbytes = fn("snx_weight_cache_bytes")(C.byref(self._desc))
            self._wcache = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self._wcache_key = None
        if key != self._wcache_key:
            check(fn("snx_weight_cache_refresh")(C.byref(self._desc), ptrs, _p(self._wcache), _stream()),
                  "snx_weight_cache_refresh")
            self._wcache_key = key
        return self._wcache

    def _rope_tables(self, max_pos: int, dev, head_dim: int = 64):
        k = (max_pos, str(dev), head_dim)
        if k not in self._rope:
            self._rope[k] = (rope_table(max_pos, head_dim, self.geom.global_rope_theta, dev),
                             rope_table(max_pos, head_dim, self.geom.local_rope_theta, dev))
        return self._rope[k]

    # ------------------------------------------------------------------ gradients
    def enable_direct_grads(self, on: bool = True):
        """Accumulate parameter gradients straight into one flat fp32 buffer whose slices ARE the
        ``.grad`` tensors (no per-call gradient tensors, one buffer for the RCCL all-reduce).
        Autograd then sees no parameter gradients from the encoder: not for torch DDP."""
        self.direct_grads = on
        if on:
            dev = self._device()
            total = sum(p.numel() for p in self.params)
            if self.flat_grad is None or self.flat_grad.device != dev:
                self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
            off = 0
            for p in self.params:
                p.grad = self.flat_grad[off:off + p.numel()].view_as(p)
                off += p.numel()

    def flatten_parameters(self):
        """Re-home every parameter into ONE contiguous fp32 buffer (canonical order); the
        nn.Parameters become views of it.  Lets the optimizer run as a single fused kernel."""
        dev = self._device()
        if self.flat_param is not None and self.flat_param.device == dev and \
                all(p.data_ptr() == self.flat_param.data_ptr() + 4 * o for p, o in zip(self.params, self._flat_offs)):
            return
        total = sum(p.numel() for p in self.params)
        flat = torch.empty(total, dtype=torch.float32, device=dev)
        offs, off = [], 0
        with torch.no_grad():
            for p in self.params:
                v = flat[off:off + p.numel()].view_as(p)
                v.copy_(p.data)
                p.data = v
                offs.append(off)
                off += p.numel()
        self.flat_param, self._flat_offs = flat, offs
        self._ptr_key = None

    def mark_weights_dirty(self):
        """Parameters were modified behind torch's back (fused optimizer): refresh the bf16 cache."""
        self._wcache_key = None

    def zero_grads(self):
        if self.flat_grad is not None:
            self.flat_grad.zero_()

    def _grad_ptrs(self, tensors):
        return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    # ------------------------------------------------------------------ forward / backward
    def forward_impl(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, save: bool, lengths=None):
        return self.forward_many_impl([(input_ids, attention_mask)], save, lengths)

    @staticmethod
    def precision() -> str:
        """What the reference computes in: bf16 inside ``torch.autocast("cuda", torch.bfloat16)`` -- its trainer,
        ref:src/train/cli/train_v33_ddp.py:337 --, fp32 otherwise (bare ``model(...)``, ref:src/model/splade_modern.py:50-88,
        and the inference encoder, ref:benchmark/encoders.py:309-345).  SNX_PRECISION=bf16|fp32 overrides."""
        forced = os.environ.get("SNX_PRECISION", "auto")
        if forced in ("bf16", "fp32"):
            return forced
        return "bf16" if (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16) else "fp32"

    def _layout(self, shapes, dev):
        """(cu_seqlens int32 [nseq + 1], pos int32 [T], groups C array) of dense [B_i, S_i] batches laid end to end (cached)."""
        key = (tuple(shapes), str(dev))
        lay = self._seq_cache.get(key)
        if lay is not None:
            self._seq_cache.move_to_end(key)
            return lay
        cu, pos, groups, t0, s0 = [0], [], [len(shapes)], 0, 0
        for B, S in shapes:
            cu += [t0 + (b + 1) * S for b in range(B)]
            pos.append(torch.arange(S, dtype=torch.int32).repeat(B))
            groups += [s0, B, S]
            t0 += B * S
            s0 += B
        lay = (torch.tensor(cu, dtype=torch.int32).to(dev), torch.cat(pos).to(dev), (C.c_int32 * len(groups))(*groups))
        self._seq_cache[key] = lay
        while len(self._seq_cache) > _SEQ_CACHE_ENTRIES:
            self._seq_cache.popitem(last=False)
        return lay

    def forward_many_impl(self, pairs, save: bool, lengths=None):
        """One native forward over several [B_i, S_i] batches laid end to end (sequence groups).
        -> sparse [sum B_i, V], token_weights [sum B_i*S_i] (flat, padded layout), arena, aux.
        ``lengths`` (optional, one CPU int tensor [B_i] per pair, right-padded inputs): run UNPADDED
        -- only the valid tokens are gathered and computed (the kernels take cu_seqlens); the
        lengths come from the host-side collator output, so no device sync is needed."""
        dev = self._device()
        fp32 = self.precision() == "fp32"
        if not fp32:
            self.geom.check_supported()
        shapes = []
        for ids, mask in pairs:
            if ids.dim() != 2 or mask.shape != ids.shape:
                raise ValueError("input_ids and attention_mask must both be [batch, seq_len]")
            if ids.device != dev or mask.device != dev:
                raise ValueError("inputs must live on the module's device")
            if ids.shape[1] > self.geom.max_position_embeddings or ids.shape[1] > 8192:
                raise ValueError("sequence too long")
            shapes.append((int(ids.shape[0]), int(ids.shape[1])))
        if len(pairs) == 1:
            ids = pairs[0][0].to(torch.int64).contiguous().view(-1)
            mask = pairs[0][1].to(torch.int64).contiguous().view(-1)
        else:
            ids = torch.cat([p[0].to(torch.int64).reshape(-1) for p in pairs])
            mask = torch.cat([p[1].to(torch.int64).reshape(-1) for p in pairs])
        T_pad = sum(B * S for B, S in shapes)
        nseq = sum(B for B, _ in shapes)
        scatter = None
        if lengths is not None:
            if len(lengths) != len(shapes):
                raise ValueError("one lengths tensor per input pair")
            cu, pos, idx, groups, t0, p0, s0 = [0], [], [], [len(shapes)], 0, 0, 0
            for (B, S), ln in zip(shapes, lengths):
                ln = torch.as_tensor(ln, dtype=torch.int64, device="cpu")
                if ln.numel() != B or int(ln.min()) < 1 or int(ln.max()) > S:
                    raise ValueError("lengths must be in [1, seq_len] for every sequence")
                valid = (torch.arange(S)[None, :] < ln[:, None])
                idx.append(valid.view(-1).nonzero().view(-1) + p0)
                pos.append(torch.arange(S, dtype=torch.int32)[None, :].expand(B, S)[valid])
                for n in ln.tolist():
                    t0 += n
                    cu.append(t0)
                groups += [s0, B, int(ln.max())]
                p0 += B * S
                s0 += B
            scatter = torch.cat(idx).to(dev, non_blocking=True)
            ids = ids[scatter]
            mask = torch.ones_like(ids)
            cu = torch.tensor(cu, dtype=torch.int32).to(dev, non_blocking=True)
            pos = torch.cat(pos).contiguous().to(dev, non_blocking=True)
            groups = (C.c_int32 * len(groups))(*groups)
            T = t0
            smax = max(groups[3 + 3 * i] for i in range(len(shapes)))
        else:
            lay = self._layout(shapes, dev)
            cu, pos, groups = lay
            T = T_pad
            smax = max(S for _, S in shapes)
        hd = self.geom.hidden_size // self.geom.num_attention_heads
        with torch.cuda.device(dev):            # native launches go to THIS device's current stream
            rg, rl = self._rope_tables(max(smax, 64), dev, hd)
            sparse = torch.empty((nseq, self.geom.vocab_size), dtype=torch.float32, device=dev)
            tw = torch.empty((T,), dtype=torch.float32, device=dev)
            if fp32:
                nbytes = fn("snx_model_workspace_bytes_f32")(C.byref(self._desc), T, nseq, int(save))
                saved = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                check(fn("snx_model_forward_f32")(C.byref(self._desc), self._param_ptrs(), _p(ids), _p(mask), _p(cu), _p(pos),
                                                  _p(rg), _p(rl), _p(saved), _p(sparse), _p(tw), T, nseq,
                                                  SNX_FWD_SAVE_FOR_BACKWARD if save else 0, _stream()), "snx_model_forward_f32")
            else:
                wc = self._weights()
                nbytes = fn("snx_model_workspace_bytes")(C.byref(self._desc), T, nseq, int(save))
                saved = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                check(fn("snx_model_forward")(C.byref(self._desc), self._param_ptrs(), _p(wc), _p(ids), _p(mask), _p(cu),
                                              _p(pos), _p(rg), _p(rl), _p(saved), _p(sparse), _p(tw),
                                              groups if (len(shapes) > 1 or lengths is not None) else None, T, nseq, smax,
                                              SNX_FWD_SAVE_FOR_BACKWARD if save else 0, _stream()), "snx_model_forward")
        if scatter is not None:                      # token_weights back to the padded layout (0 at padding)
            tw_full = torch.zeros((T_pad,), dtype=torch.float32, device=dev)
            tw_full[scatter] = tw
            tw = tw_full
        aux = (ids, mask, cu, pos, rg, rl, T, nseq, smax, groups, fp32)
        return sparse, tw, saved, aux

    def routing_rows(self, saved: torch.Tensor, aux) -> torch.Tensor:
        """Arg-max sequence position per (sequence, vocab) entry chosen by the fused max-pool
        ([B, V] int64; entries whose pooled value is 0 carry no gradient)."""
        T, B = aux[6], aux[7]
        if aux[10]:
            raise NotImplementedError("routing_rows: bf16 path only (the fp32 path keeps 64-bit keys)")
        off = fn("snx_model_keys_offset")(C.byref(self._desc), T, B)
        keys = saved[off:off + B * self.geom.vocab_size * 4].view(torch.int32).view(B, -1).to(torch.int64) & 0xFFFFFFFF
        return 0xFFFF - (keys & 0xFFFF)

    def backward_impl(self, saved: torch.Tensor, aux, g_sparse: torch.Tensor, sync_token=None, plan=None):
        """``plan`` = (T_plan, nseq_plan): `saved` was laid out for that many rows / sequences (a micro-step arena that
        holds fewer passes than planned); default: the arena of exactly these T rows."""
        ids, mask, cu, pos, rg, rl, T, B, S, groups, fp32 = aux  # B = total sequences, S = longest
        Tp, Bp = plan if plan is not None else (T, B)
        dev = self._device()
        if g_sparse.shape != (B, self.geom.vocab_size):
            raise ValueError("bad gradient shape")
        g = g_sparse.to(torch.float32).contiguous()
        if self.direct_grads:
            grads = [p.grad for p in self.params]
            ret = None
        else:
            total = sum(p.numel() for p in self.params)
            flat = torch.zeros(total, dtype=torch.float32, device=dev)
            grads, off = [], 0
            for p in self.params:
                grads.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            ret = grads
        sync = self.grad_sync if (self.direct_grads and self.grad_sync is not None) else None
        with torch.cuda.device(dev):
            if fp32:
                nbytes = fn("snx_model_bwd_workspace_bytes_f32")(C.byref(self._desc), T)
                scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                args32 = (C.byref(self._desc), self._param_ptrs(), self._grad_ptrs(grads), _p(ids), _p(mask), _p(cu), _p(pos),
                          _p(rg), _p(rl), _p(saved), _p(g), _p(scratch), T, B)
            else:
                nbytes = fn("snx_model_bwd_workspace_bytes")(C.byref(self._desc), Tp, Bp, S)
                scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                args = (C.byref(self._desc), self._param_ptrs(), self._grad_ptrs(grads), _p(self._weights()), _p(ids),
                        _p(mask), _p(cu), _p(pos), _p(rg), _p(rl), _p(saved), _p(g), _p(scratch), groups, Tp, Bp, T, B, S)
                n_units = self.geom.num_hidden_layers + 2

            def run_all():
                if fp32:
                    check(fn("snx_model_backward_f32")(*args32, _stream()), "snx_model_backward_f32")
                else:
                    check(fn("snx_model_backward_units_range")(*args, 0, n_units, None, _stream()),
                          "snx_model_backward_units_range")

            def run_units(ub, ue):
                if fp32:                        # one native call; the exchange stream then waits for the launch stream
                    if ub == 0:
                        run_all()
                    sync.stream.wait_stream(torch.cuda.current_stream(dev))
                    return
                # the native call makes the exchange stream wait for the launch stream and the weight-gradient stream
                check(fn("snx_model_backward_units_range")(*args, ub, ue, C.c_void_p(sync.stream.cuda_stream), _stream()),
                      "snx_model_backward_units_range")

            if sync is None:
                run_all()
            else:
                # gradient exchange overlapped with the LAST backward of an armed micro-step (BucketedGradSync): after
                # each unit range its finished slice of the flat gradient is reduced on the exchange stream while the
                # next range computes; while it runs the persistent kernels leave CUs to RCCL's channel workgroups
                reserve = sync.reserved_cus if (not fp32 and sync.armed and sync_token == sync.epoch
                                                and sync.outstanding == 1) else 0
                if reserve:
                    fn("snx_set_reserved_cus")(reserve)
                try:
                    sync.run_backward(sync_token, self.flat_grad, self.geom.num_hidden_layers + 2, run_all, run_units,
                                      self.unit_param_range)
                finally:
                    if reserve:
                        fn("snx_set_reserved_cus")(0)
        return ret

    # ------------------------------------------------------------------ micro-step arena
    def _arena_eligible(self) -> bool:
        return (self.step_arena_on and not self.keep_last_ctx and self.precision() == "bf16" and
                self.geom.bf16_unsupported_reason() is None)

    def step_arena_capacity(self):
        """(token rows, sequences) an arena is laid out for: the maxima of the last micro-steps (dynamic padding makes
        them fluctuate), or None while no micro-step with at least two passes has been seen."""
        h = [t for t in self._step_hist if t[2] >= 2]
        if not h:
            return None
        return max(t[0] for t in h), max(t[1] for t in h)

    def _step_forward(self, ids: torch.Tensor) -> None:
        if self._step_phase == "bwd":                        # first forward after a backward: a new micro-step begins
            self._step_phase, self._step_tot = "fwd", [0, 0, 0]
        self._step_tot[0] += int(ids.shape[0]) * int(ids.shape[1])
        self._step_tot[1] += int(ids.shape[0])
        self._step_tot[2] += 1

    def step_arena_backward_begins(self) -> None:
        """Any backward of a saving single-batch forward: the micro-step's forwards are complete."""
        if self._step_phase == "fwd":
            self._step_phase = "bwd"
            if 2 <= self._step_tot[2] <= StepArena.MAX_PASSES:   # (a run of forwards that is no micro-step teaches nothing)
                self._step_hist = (self._step_hist + [tuple(self._step_tot)])[-8:]

    def step_arena_observe(self, ids: torch.Tensor) -> bool:
        """A single-batch saving forward on the ordinary path.  True: its backward reports back."""
        if not self._arena_eligible():
            self._step_hist, self._step_phase = [], "bwd"
            return False
        self._step_forward(ids)
        return True

    def _arena_note(self, what: str, msg: str) -> None:
        self.arena_stats[what] += 1
        if what not in self._arena_warned:
            self._arena_warned.add(what)
            logger.warning("snx micro-step arena: %s (counted in EncoderRuntime.arena_stats[%r]; further ones are only "
                           "counted)", msg, what)

    def _weights_key(self):
        return tuple(p._version for p in self.params)

    def _drop_stale_arena(self, a: "StepArena", why: str) -> None:
        """An OPEN arena whose micro-step will never complete: a grad-enabled forward used for logging / evaluation, an
        exception or a `continue` between the forwards and loss.backward().  Its live nodes (if any) keep working on their
        own -- the arena back-propagates once all of them have reported -- but no further pass joins it, and the forwards
        counted so far do not become a micro-step of the capacity history."""
        a.closed = True
        if self._arena is a:
            self._arena = None
        self._step_phase, self._step_tot = "bwd", [0, 0, 0]
        self._arena_note("stale_dropped", "an open arena was abandoned (" + why + ")")

    def step_arena_place(self, ids: torch.Tensor, mask: torch.Tensor):
        """This pass into the micro-step's shared arena -> (sparse [B, V], token_weights [B*S], arena, index), or None when
        no capacity has been learnt yet or the pass does not fit what is left of it (then it runs on the ordinary path,
        which also owns the shape / length errors: an input the ordinary path would refuse is never placed)."""
        if not self._arena_eligible() or ids.dim() != 2 or mask.shape != ids.shape or ids.device != self._device() or \
                mask.device != ids.device:
            return None
        if ids.shape[1] > self.geom.max_position_embeddings or ids.shape[1] > 8192 or ids.shape[0] < 1 or ids.shape[1] < 1:
            return None                                      # forward_many_impl raises "sequence too long"
        a = self._arena
        if a is not None and a.placed > 0:
            why = a.stale_reason()
            if why:
                self._drop_stale_arena(a, why)
                a = None
        cap = self.step_arena_capacity()
        if cap is None:
            return None
        B, S = int(ids.shape[0]), int(ids.shape[1])
        if a is None:
            if B * S > cap[0] or B > cap[1]:
                self._arena_note("fell_back", f"a pass of {B} x {S} tokens exceeds the learnt capacity {cap}: it runs on "
                                 "the ordinary path (its own backward)")
                return None
            a = self._arena = StepArena(self, cap[0], cap[1])
        if not a.fits(B, S):
            self._arena_note("fell_back", f"a pass of {B} x {S} tokens does not fit what is left of the arena "
                             f"({a.row0[-1]} of {a.T} rows, {a.seq0[-1]} of {a.nseq} sequences, {a.placed} passes used): it "
                             "runs on the ordinary path (its own backward)")
            return None
        self._step_forward(ids)
        self.arena_stats["placed"] += 1
        return a.place(ids, mask)

    def __call__(self, input_ids, attention_mask):
        (out,) = self.forward_many([(input_ids, attention_mask)])
        return out

    def forward_many(self, pairs, lengths=None):
        """[(ids [B_i,S_i], mask)] -> [(sparse_repr [B_i,V], token_weights [B_i,S_i])], one native pass.
        ``lengths``: optional per-pair CPU length tensors -> unpadded (varlen) execution."""
        pairs = list(pairs)
        flat = [t for p in pairs for t in p]
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.params):
            # one autograd output per pass: a slice taken OUTSIDE the Function would cost its backward a zero fill of the
            # whole [rows, V] gradient, a copy and an add per pass (three of each per micro-step, 9.6 M floats each)
            res = _SpladeEncodeFn.apply(self, len(pairs), lengths, *flat, *self.params)
            sparses, tw = res[:-1], res[-1]
        else:
            sparse, tw, _, _ = self.forward_many_impl(pairs, save=False, lengths=lengths)
            sparses, r0 = [], 0
            for ids, _ in pairs:
                sparses.append(sparse[r0:r0 + ids.shape[0]])
                r0 += ids.shape[0]
        out, t0 = [], 0
        for (ids, _), sp in zip(pairs, sparses):
            B, S = ids.shape
            out.append((sp, tw[t0:t0 + B * S].view(B, S)))
            t0 += B * S
        return out


class StepArena:
    """The forwards of ONE micro-step placed in ONE activation arena, back-propagated by ONE native backward.

    The reference's loop calls model(...) three times per micro-step (query, positive, negative:
    ref:src/train/cli/train_v33_ddp.py:339-343) and back-propagates once (:364).  Run as three independent passes that costs
    10 ms per micro-step of DEVICE time over the fused pass (402 + 69 GEMM launches of a third of the rows each:
    profiles/r05_caller_breakdown.txt), two thirds of it in the backward.  Without touching the caller: the runtime learns
    how many token rows and sequences a micro-step holds (the maxima of the last eight: the reference's collator pads to the
    longest of each batch, ref:src/train/data/dataloader.py:95-118, so they fluctuate), lays an arena out for that capacity
    and lets every model(...) call fill the next row range (snx_model_forward_range: the same kernels on shifted pointers,
    outputs bit-identical to a stand-alone pass).  The backward calls of the autograd nodes only hand in their output
    gradients; the LAST one to report runs the fused native backward over the rows actually filled
    (snx_model_backward_units_range) -- the launches, and the bucketed gradient exchange, of the fused micro-step, and bit
    for bit its gradients.  A pass that does not fit what is left of the capacity runs on the ordinary path.

    Limits (documented, checked): bf16 path, dense [B, S] batches (no `lengths`), at most eight passes, every placed pass
    must take part in the loss -- if the autograd engine finishes a backward pass with a placed node unreported, the
    remaining gradient is back-propagated with zeros for it in the flat-gradient mode and raised as an error otherwise."""

    MAX_PASSES = 8

    def __init__(self, rt: "EncoderRuntime", t_cap: int, nseq_cap: int):
        self.rt = rt
        dev = rt._device()
        self.dev = dev
        self.T, self.nseq = int(t_cap), int(nseq_cap)
        self.shapes: list = []
        self.row0, self.seq0 = [0], [0]
        V = rt.geom.vocab_size
        with torch.cuda.device(dev):
            nbytes = fn("snx_model_workspace_bytes")(C.byref(rt._desc), self.T, self.nseq, 1)
            self.saved = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self.sparse = torch.empty((self.nseq, V), dtype=torch.float32, device=dev)
            self.tw = torch.empty((self.T,), dtype=torch.float32, device=dev)
            self.ids = torch.empty((self.T,), dtype=torch.int64, device=dev)
            self.mask = torch.empty((self.T,), dtype=torch.int64, device=dev)
        self.placed = self.reported = 0
        self.grads: list = []
        self.nodes: list = []                                # weak references to the autograd nodes of the placed passes
        self.got: list = []                                  # which of them have reported
        self.wkey = None                                     # parameter versions the first placed pass computed with
        self.closed = self.done = self.reporting = False

    def attach(self, k: int, node) -> None:
        """The autograd node (ctx) of placed pass k.  A node that dies without a backward (its outputs were only logged, an
        exception unwound the micro-step) can never report: the arena learns that from the dead reference."""
        try:
            self.nodes[k] = weakref.ref(node)
        except TypeError:                                    # a node type without weak references: never counted dead
            self.nodes[k] = None

    def dead(self) -> int:
        return sum(1 for k, r in enumerate(self.nodes) if r is not None and not self.got[k] and r() is None)

    def stale_reason(self) -> str:
        """Why an OPEN arena (no backward yet) cannot be the micro-step in progress any more, or ''."""
        if self.closed:
            return ""
        if self.dead():
            return "a placed model(...) output was dropped without a backward"
        rt = self.rt
        if self.wkey is not None and (rt._wcache_key is None or rt._weights_key() != self.wkey):
            return "the parameters changed (optimizer step / load) after its first pass"
        return ""

    def fits(self, B: int, S: int) -> bool:
        return (not self.closed and self.placed < self.MAX_PASSES and self.row0[-1] + B * S <= self.T and
                self.seq0[-1] + B <= self.nseq)

    def place(self, ids: torch.Tensor, mask: torch.Tensor):
        rt, k = self.rt, self.placed
        B, S = int(ids.shape[0]), int(ids.shape[1])
        r0, s0, T = self.row0[-1], self.seq0[-1], B * S
        cu, pos, _ = rt._layout([(B, S)], self.dev)
        ids_k, mask_k = self.ids[r0:r0 + T], self.mask[r0:r0 + T]
        ids_k.copy_(ids.reshape(-1))
        mask_k.copy_(mask.reshape(-1))
        hd = rt.geom.hidden_size // rt.geom.num_attention_heads
        with torch.cuda.device(self.dev):
            rg, rl = rt._rope_tables(max(S, 64), self.dev, hd)
            wc = rt._weights()
            if k == 0:
                self.wkey = rt._wcache_key
            check(fn("snx_model_forward_range")(C.byref(rt._desc), rt._param_ptrs(), _p(wc), _p(ids_k), _p(mask_k),
                                                _p(cu), _p(pos), _p(rg), _p(rl), _p(self.saved), _p(self.sparse), _p(self.tw),
                                                None, self.T, self.nseq, r0, s0, T, B, S, SNX_FWD_SAVE_FOR_BACKWARD,
                                                _stream()), "snx_model_forward_range")
        self.shapes.append((B, S))
        self.row0.append(r0 + T)
        self.seq0.append(s0 + B)
        self.grads.append(None)
        self.nodes.append(None)
        self.got.append(False)
        self.placed += 1
        return self.sparse[s0:s0 + B], self.tw[r0:r0 + T], self, k

    def routing_rows(self, k: int) -> torch.Tensor:
        """Arg-max sequence position per (sequence, vocab) entry of pass k (as EncoderRuntime.routing_rows)."""
        rt, V = self.rt, self.rt.geom.vocab_size
        B = self.shapes[k][0]
        off = fn("snx_model_keys_offset")(C.byref(rt._desc), self.T, self.nseq) + self.seq0[k] * V * 4
        keys = self.saved[off:off + B * V * 4].view(torch.int32).view(B, -1).to(torch.int64) & 0xFFFFFFFF
        return 0xFFFF - (keys & 0xFFFF)

    def report(self, k: int, g: Optional[torch.Tensor], token):
        """Backward of node k: hand in dL/d sparse_k.  Returns the parameter gradients from the LAST node to report
        (None in the flat-gradient mode, where the native backward accumulates in place)."""
        rt = self.rt
        if not self.reporting:
            self.reporting = True
            was_open = not self.closed
            self.closed = True                               # no further pass can join; the engine tells us when it is done
            if rt._arena is self:
                rt._arena = None
            if was_open:                                     # (an arena dropped as stale is no micro-step of the history)
                rt.step_arena_backward_begins()
            torch.autograd.Variable._execution_engine.queue_callback(self._engine_done)
        self.grads[k] = g
        self.got[k] = True
        self.reported += 1
        sync = rt.grad_sync if (rt.direct_grads and rt.grad_sync is not None) else None
        if self.reported + self.dead() < self.placed:        # (a dead node never reports: zeros stand in for it)
            if sync is not None:
                sync.claim_backward(token)                   # hands the token back; never the last one outstanding
            return None
        return self._run(token)

    def _run(self, token):
        rt, n = self.rt, self.placed
        self.done = True
        shapes = list(self.shapes)
        missing = sum(1 for g in self.grads if g is None)
        if missing:
            rt._arena_note("zero_filled", f"{missing} of {n} placed passes took no part in the loss: zeros were "
                           "back-propagated for them")
        T, nseq = self.row0[-1], self.seq0[-1]
        V = rt.geom.vocab_size
        parts = [g.to(torch.float32) if g is not None else torch.zeros((b, V), dtype=torch.float32, device=self.dev)
                 for g, (b, _) in zip(self.grads, shapes)]
        g_all = _gather_rows(parts, [b for b, _ in shapes], V, self.dev)
        cu, pos, groups = rt._layout(shapes, self.dev)
        smax = max(s for _, s in shapes)
        hd = rt.geom.hidden_size // rt.geom.num_attention_heads
        rg, rl = rt._rope_tables(max(smax, 64), self.dev, hd)
        aux = (self.ids[:T], self.mask[:T], cu, pos, rg, rl, T, nseq, smax, groups if n > 1 else None, False)
        grads = rt.backward_impl(self.saved, aux, g_all, token, plan=(self.T, self.nseq))
        self.saved = None
        self.grads = []
        return grads

    def _engine_done(self):
        """End of the autograd engine's backward pass: a placed pass that never reported took no part in the loss."""
        if self.done:
            return
        rt = self.rt
        if not rt.direct_grads:
            raise RuntimeError("snx: a model(...) output of this micro-step took no part in the loss, so its backward never "
                               "ran and the deferred backward of the micro-step arena cannot return its gradients through "
                               "autograd; set SNX_STEP_ARENA=0 for such loops")
        sync = rt.grad_sync
        token = None
        if sync is not None and sync.armed:
            # the unreported nodes still hold tokens: give them back so that this backward is the last outstanding one
            missing = self.placed - self.reported
            for _ in range(missing - 1):
                sync.claim_backward(sync.epoch)
            token = sync.epoch
        self._run(token)


def _gather_rows(gs, rows, vocab: int, device) -> torch.Tensor:
    """The per-pass output gradients as ONE [sum(rows), V] fp32 tensor.  In place (no kernel) when they already are
    consecutive row ranges of one buffer -- what snx.loss.SpladeLossFn.backward hands over --, else one concatenation;
    a pass whose output took no part in the loss contributes zeros."""
    gs = [g if g is None else (g if (g.dtype == torch.float32 and g.is_contiguous()) else g.to(torch.float32).contiguous())
          for g in gs]
    if all(g is not None for g in gs):
        first = gs[0]
        end = first.data_ptr() + first.numel() * 4
        adjacent = True
        for g in gs[1:]:
            if g.untyped_storage().data_ptr() != first.untyped_storage().data_ptr() or g.data_ptr() != end:
                adjacent = False
                break
            end += g.numel() * 4
        if adjacent and len(gs) > 1:
            total = sum(rows)
            need = first.storage_offset() + total * vocab
            if first.untyped_storage().nbytes() >= need * 4:
                return torch.empty(0, dtype=torch.float32, device=device).set_(
                    first.untyped_storage(), first.storage_offset(), (total, vocab), (vocab, 1))
        if len(gs) == 1:
            return gs[0]
    parts = [g if g is not None else torch.zeros((b, vocab), dtype=torch.float32, device=device)
             for g, b in zip(gs, rows)]
    return torch.cat(parts, dim=0)


class _SpladeEncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rt: EncoderRuntime, n_pairs: int, lengths, *args):
        pairs = [(args[2 * i], args[2 * i + 1]) for i in range(n_pairs)]
        ctx.n_in = 2 * n_pairs + 1
        ctx.rt, ctx.arena, ctx.counted = rt, None, False
        if n_pairs == 1 and lengths is None:
            placed = rt.step_arena_place(pairs[0][0], pairs[0][1])
            if placed is not None:                       # this pass went into the micro-step's shared arena
                sparse, tw, ctx.arena, ctx.k = placed
                ctx.arena.attach(ctx.k, ctx)
                ctx.sync_token = rt.grad_sync.on_forward() if (rt.direct_grads and rt.grad_sync is not None) else None
                ctx.mark_non_differentiable(tw)
                ctx.rows, ctx.vocab = [sparse.shape[0]], sparse.shape[1]
                return sparse, tw
            ctx.counted = rt.step_arena_observe(pairs[0][0])
        sparse, tw, saved, aux = rt.forward_many_impl(pairs, save=True, lengths=lengths)
        ctx.saved_arena, ctx.aux = saved, aux
        # an armed micro-step counts its saving forwards: only the backward of the last outstanding one exchanges
        ctx.sync_token = rt.grad_sync.on_forward() if (rt.direct_grads and rt.grad_sync is not None) else None
        if rt.keep_last_ctx:
            rt.last_ctx = (saved, aux)      # parity tests: routing of the latest forward
        ctx.mark_non_differentiable(tw)
        ctx.rows = [ids.shape[0] for ids, _ in pairs]
        ctx.vocab = sparse.shape[1]
        outs, r0 = [], 0
        for b in ctx.rows:
            outs.append(sparse[r0:r0 + b])
            r0 += b
        return tuple(outs) + (tw,)

    @staticmethod
    def backward(ctx, *gs):
        rt = ctx.rt
        head = (None, None) + tuple(None for _ in range(ctx.n_in))
        if ctx.arena is not None:
            grads = ctx.arena.report(ctx.k, gs[0], ctx.sync_token)
            ctx.arena = None
            return head + (tuple(None for _ in rt.params) if grads is None else tuple(grads))
        g_sparse = _gather_rows(gs[:-1], ctx.rows, ctx.vocab, ctx.saved_arena.device)
        grads = rt.backward_impl(ctx.saved_arena, ctx.aux, g_sparse, ctx.sync_token)
        if ctx.counted:
            rt.step_arena_backward_begins()
        ctx.saved_arena = None
        head = (None, None) + tuple(None for _ in range(ctx.n_in))
        if grads is None:
            return head + tuple(None for _ in rt.params)
        return head + tuple(grads)
