"""Fused global-norm clip + AdamW on flat buffers (``snx_adamw_clip_step``).

Drop-in for ``clip_grad_norm_`` + ``torch.optim.AdamW`` as used by
ref:src/train/cli/train_v33_ddp.py:367-374,560-581: same ``param_groups`` (so LambdaLR and the
checkpoint code work unchanged), same ``state_dict`` layout (``step`` / ``exp_avg`` /
``exp_avg_sq`` per parameter -- the moments are views into two flat fp32 buffers), hence
checkpoints are interchangeable with torch's AdamW."""
from __future__ import annotations

import ctypes as C
from typing import List

import torch

from ._lib import check, fn
from .ops import _p, _stream


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, runtime, param_groups: List[dict], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0.0)
        super().__init__(param_groups, defaults)
        self.rt = runtime
        runtime.flatten_parameters()
        runtime.enable_direct_grads(True)
        flat = runtime.flat_param
        self._m = torch.zeros_like(flat)
        self._v = torch.zeros_like(flat)
        self._norm = torch.zeros(1, dtype=torch.float32, device=flat.device)
        self._scratch = torch.empty(fn("snx_adamw_scratch_bytes")(), dtype=torch.uint8, device=flat.device)
        self._steps = 0
        # offsets of every parameter in the flat order; the weight-decay-free elements must be ONE range
        offs, off = {}, 0
        for p in runtime.params:
            offs[id(p)] = (off, p.numel())
            off += p.numel()
        wds = {}
        for g in self.param_groups:
            for p in g["params"]:
                if id(p) not in offs:
                    raise ValueError("FusedAdamW: every optimised parameter must belong to the bound runtime")
                wds[id(p)] = float(g["weight_decay"])
        if len(wds) != len(offs):
            raise ValueError("FusedAdamW: all runtime parameters must be optimised")
        decays = {w for w in wds.values() if w != 0.0}
        if len(decays) > 1:
            raise ValueError("FusedAdamW supports one non-zero weight_decay value")
        self._wd = decays.pop() if decays else 0.0
        nd = sorted(offs[i] for i, w in wds.items() if w == 0.0) if self._wd != 0.0 else []
        for (a, n), (b, _) in zip(nd, nd[1:]):
            if a + n != b:
                raise ValueError("FusedAdamW: the no-decay parameters must be contiguous in the flat order")
        self._nodecay = (nd[0][0], nd[-1][0] + nd[-1][1]) if nd else (0, 0)
        for p in runtime.params:
            o, n = offs[id(p)]
            self.state[p] = {"step": torch.tensor(0.0), "exp_avg": self._m[o:o + n].view_as(p),
                             "exp_avg_sq": self._v[o:o + n].view_as(p)}
        self._offs = offs

    @torch.no_grad()
    def step(self, closure=None, max_norm: float = 0.0):
        """One optimizer step; ``max_norm > 0`` first clips the global gradient norm (the pre-clip
        norm is left in ``self.grad_norm``, a device scalar -- no host sync)."""
        lrs = {float(g["lr"]) for g in self.param_groups}
        if len(lrs) != 1:
            raise ValueError("FusedAdamW needs one learning rate for all groups")
        g0 = self.param_groups[0]
        self._steps += 1
        hp = (C.c_float * 6)(lrs.pop(), g0["betas"][0], g0["betas"][1], g0["eps"], self._wd, float(max_norm))
        flat = self.rt.flat_param
        check(fn("snx_adamw_clip_step")(_p(flat), _p(self.rt.flat_grad), _p(self._m), _p(self._v), flat.numel(), hp,
                                        self._steps, self._nodecay[0], self._nodecay[1], _p(self._norm),
                                        _p(self._scratch), _stream()), "snx_adamw_clip_step")
        self.rt.mark_weights_dirty()
        for st in self.state.values():
            st["step"] = torch.tensor(float(self._steps))
        return None

    @property
    def grad_norm(self) -> torch.Tensor:
        return self._norm

    def zero_grad(self, set_to_none: bool = False):
        self.rt.zero_grads()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        steps = 0
        for p in self.rt.params:          # re-home the loaded moments into the flat buffers
            st = self.state[p]
            o, n = self._offs[id(p)]
            self._m[o:o + n].view_as(p).copy_(st["exp_avg"])
            self._v[o:o + n].view_as(p).copy_(st["exp_avg_sq"])
            st["exp_avg"], st["exp_avg_sq"] = self._m[o:o + n].view_as(p), self._v[o:o + n].view_as(p)
            steps = max(steps, int(float(st["step"])))
        self._steps = steps
