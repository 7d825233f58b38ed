"""SPLADELossV33 -- InfoNCE (in-batch + hard negatives) + FLOPS regularisation with the
quadratic lambda warm-up (+ optional MarginMSE), computed on the GPU by libsnx.so.

Same constructor and ``forward`` signature, same ``loss_dict`` keys and ``get_avg_nonzero`` as
ref:src/model/losses.py:14-301.  Differences:
  * the values of ``loss_dict`` are 0-d device tensors (the lambdas stay python floats) instead of
    ``.item()`` results: ``f"{x:.4f}"``, ``x > 0`` and ``float(x)`` all work and synchronise only
    when actually used, which removes the reference's 10 host syncs per micro-step
    (ref:losses.py:276-294);
  * extension (not in the reference, config 4 of BASELINE.json): ``positive_repr`` may hold the
    all-gathered positives of every rank ([world*B, V]) together with ``label_offset=rank*B``;
  * the KL-distillation branch (ref:losses.py:239-253; never fed by the trainer, ref:train_v33_ddp.py:353-360) is
    part of the same device kernels: it reuses InfoNCE's V-chunked in-batch dots -- no vendor-library call on this path.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from snx.loss import splade_loss


class SPLADELossV33(nn.Module):
    def __init__(self, lambda_q: float = 1e-2, lambda_d: float = 3e-3, temperature: float = 1.0,
                 flops_warmup_steps: int = 20000, lambda_kd: float = 0.0, kd_temperature: float = 1.0,
                 lambda_initial_ratio: float = 0.1, lambda_margin_mse: float = 0.0, lambda_neg: float = 0.0):
        super().__init__()
        self.lambda_q = lambda_q
        self.lambda_d = lambda_d
        self.temperature = temperature
        self.flops_warmup_steps = flops_warmup_steps
        self.lambda_kd = lambda_kd
        self.kd_temperature = kd_temperature
        self.lambda_initial_ratio = lambda_initial_ratio
        self.lambda_margin_mse = lambda_margin_mse
        self.lambda_neg = lambda_neg if lambda_neg > 0 else lambda_d        # ref:losses.py:50
        self._ema: Optional[torch.Tensor] = None                            # (nonzero_q, nonzero_d) EMA, on device
        self._count = 0

    def _lambda_schedule(self, step: int, target_lambda: float) -> float:
        """lambda(t) = target * (r0 + (1 - r0) * (t/T)^2), target once t >= T (ref:losses.py:75-90)."""
        if step >= self.flops_warmup_steps:
            return target_lambda
        ratio = step / max(self.flops_warmup_steps, 1)
        r0 = self.lambda_initial_ratio
        return target_lambda * (r0 + (1.0 - r0) * ratio * ratio)

    def forward(self, anchor_repr: torch.Tensor, positive_repr: torch.Tensor, negative_repr: torch.Tensor,
                global_step: int = 0, teacher_scores: Optional[torch.Tensor] = None,
                teacher_pos_scores: Optional[torch.Tensor] = None,
                teacher_neg_scores: Optional[torch.Tensor] = None, **kwargs) -> Tuple[torch.Tensor, Dict]:
        step = int(global_step)
        lam_q = self._lambda_schedule(step, self.lambda_q)
        lam_d = self._lambda_schedule(step, self.lambda_d)
        lam_neg = self._lambda_schedule(step, self.lambda_neg)
        B, V = anchor_repr.shape
        if negative_repr.dim() == 3:
            k = negative_repr.shape[1]
            neg2d = negative_repr.reshape(B * k, V)
        else:
            k, neg2d = 1, negative_repr
        use_mm = self.lambda_margin_mse > 0 and teacher_pos_scores is not None and teacher_neg_scores is not None
        use_kd = self.lambda_kd > 0 and teacher_scores is not None          # inactive in the V33 trainer
        hp = (self.temperature, lam_q, lam_d, lam_neg, self.lambda_margin_mse if use_mm else 0.0,
              self.lambda_kd if use_kd else 0.0, self.kd_temperature)
        bf16_mm = torch.is_autocast_enabled()        # the reference's torch.mm runs in bf16 under autocast
        loss, sc = splade_loss(anchor_repr, positive_repr, neg2d, hp, k,
                               tpos=teacher_pos_scores if use_mm else None,
                               tneg=teacher_neg_scores if use_mm else None,
                               label_off=int(kwargs.get("label_offset", 0)), bf16_mm=bf16_mm,
                               tscores=teacher_scores if use_kd else None)
        zero = getattr(self, "_zero", None)               # one cached 0-d zero per device (no fill kernel per call)
        if zero is None or zero.device != sc.device:
            zero = self._zero = sc.new_zeros(())
        kd_loss = sc[8] if use_kd else zero
        with torch.no_grad():
            nz = sc[6:8]
            self._ema = 0.1 * nz if self._ema is None else torch.add(0.9 * self._ema, nz, alpha=0.1)
            self._count += 1
        loss_dict = {
            "infonce": sc[1], "flops_q": sc[2], "flops_d": sc[3], "flops_neg": sc[4],
            "lambda_q": lam_q, "lambda_d": lam_d, "lambda_neg": lam_neg,
            "kd": kd_loss.detach(), "margin_mse": sc[5] if use_mm else zero,
            "nonzero_q": sc[6], "nonzero_d": sc[7],
        }
        return loss, loss_dict

    def get_avg_nonzero(self) -> Tuple[float, float]:
        """EMA (0.9/0.1) of the non-zero counts (query, doc); synchronises with the device."""
        if self._ema is None:
            return 0.0, 0.0
        q, d = self._ema.tolist()
        return q, d
