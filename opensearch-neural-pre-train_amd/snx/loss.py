"""Autograd binding of the device-side SPLADELossV33 (``snx_loss_fwd`` / ``snx_loss_bwd``)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from ._lib import check, fn
from .ops import _p, _stream


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.float32).contiguous()


class SpladeLossFn(torch.autograd.Function):
    """(q [B,V], p [Bp,V], n [B*k,V]) -> (loss 0-d, scalars [9]).  scalars = {loss, infonce, flops_q,
    flops_d, flops_neg, margin_mse, nonzero_q, nonzero_d, kd}; only `loss` is differentiable.
    hp = (temperature, lambda_q, lambda_d, lambda_neg, lambda_margin_mse, lambda_kd, kd_temperature)."""

    @staticmethod
    def forward(ctx, q, p, n, tpos, tneg, hp, k, label_off, bf16_mm, tscores=None):
        if not q.is_cuda:
            raise RuntimeError("SPLADELossV33 (snx backend) needs GPU tensors; there is no CPU fallback")
        q, p, n = _f32c(q), _f32c(p), _f32c(n)
        B, V = q.shape
        Bp = p.shape[0]
        if p.shape[1] != V or n.shape != (B * k, V):
            raise ValueError(f"loss: inconsistent shapes q{tuple(q.shape)} p{tuple(p.shape)} n{tuple(n.shape)} k={k}")
        tp = _f32c(tpos) if tpos is not None else None
        tn = _f32c(tneg).view(-1) if tneg is not None else None
        if tp is not None and (tp.numel() != B or tn is None or tn.numel() != B * k):
            raise ValueError("teacher score shapes must be [B] and [B,k]")
        hp = tuple(float(x) for x in hp)
        if len(hp) == 5:
            hp = hp + (0.0, 1.0)
        ts = _f32c(tscores) if (tscores is not None and hp[5] > 0) else None
        if ts is not None and tuple(ts.shape) != (B, B):
            raise ValueError(f"teacher_scores must be [B, B] = [{B}, {B}], got {tuple(ts.shape)}")
        dims = (C.c_int32 * 6)(B, Bp, k, V, label_off, int(bf16_mm))
        hpa = (C.c_float * 7)(*hp)
        nbytes = fn("snx_loss_workspace_bytes")(B, Bp, k, V)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
        out = torch.empty(9, dtype=torch.float32, device=q.device)
        check(fn("snx_loss_fwd")(_p(q), _p(p), _p(n), _p(tp), _p(tn), _p(ts), hpa, dims, _p(ws), _p(out), _stream()),
              "snx_loss_fwd")
        ctx.save_for_backward(q, p, n, ws)
        ctx.dims, ctx.hp, ctx.use_kd = dims, hpa, int(ts is not None)
        loss = out[0].clone()
        scalars = out
        ctx.mark_non_differentiable(scalars)
        return loss, scalars

    @staticmethod
    def backward(ctx, gloss, _gs):
        q, p, n, ws = ctx.saved_tensors
        g = gloss.to(torch.float32).contiguous().view(1)
        # one buffer, three row ranges: when q / p / n are the outputs of ONE native pass (forward_many), its backward
        # recognises the adjacent gradients and reads the buffer in place (snx/encoder.py _gather_rows)
        g_all = torch.empty((q.shape[0] + p.shape[0] + n.shape[0], q.shape[1]), dtype=torch.float32, device=q.device)
        dq, dp, dn = g_all[:q.shape[0]], g_all[q.shape[0]:q.shape[0] + p.shape[0]], g_all[q.shape[0] + p.shape[0]:]
        check(fn("snx_loss_bwd")(_p(q), _p(p), _p(n), _p(g), ctx.hp, ctx.dims, _p(ws), ctx.use_kd, _p(dq), _p(dp), _p(dn),
                                 _stream()), "snx_loss_bwd")
        return dq, dp, dn, None, None, None, None, None, None, None


def splade_loss(q, p, n, hp, k: int, tpos: Optional[torch.Tensor] = None, tneg: Optional[torch.Tensor] = None,
                label_off: int = 0, bf16_mm: bool = False, tscores: Optional[torch.Tensor] = None):
    return SpladeLossFn.apply(q, p, n, tpos, tneg, tuple(hp), int(k), int(label_off), bool(bf16_mm), tscores)
