"""CPU oracle for the SPLADE-ModernBERT training path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it.  The product path
(``opensearch-neural-pre-train_amd/``) never imports anything under ``oracle/`` and
fails loudly when the HIP library is missing.

It is a from-scratch restatement, in plain PyTorch CPU ops, of what the reference computes on
the hot path.  Citations (``ref:`` = /root/reference, ``hf:`` = transformers 5.15.0
``models/modernbert/modeling_modernbert.py``):

  * ModernBERT-MLM forward ............ hf:52-71 (embeddings), hf:89-91 (GeGLU MLP),
    hf:136-163 + hf:188-219 (RoPE, half-split rotate), hf:262-301 (attention),
    hf:318-333 (pre-LN layer, layer 0 has no attn_norm), hf:476 (final norm),
    hf:489-490 (head), hf:550 (tied decoder + bias); masks ``masking_utils.py:141-150``.
  * SPLADE tail ....................... ref:src/model/splade_modern.py:50-88
  * SPLADELossV33 ..................... ref:src/model/losses.py:57-297
  * optimizer / schedule / step ....... ref:src/train/cli/train_v33_ddp.py:289-374,560-592

Parity pin: this oracle is checked against golden vectors captured from the reference code
itself (``tools/make_golden.py`` imports the reference by file path in the build container;
fixtures under ``tests/golden/``; transformers 5.15.0).  See ``tests/test_oracle_golden.py``.

Two numeric modes:
  * ``mode="fp32"``  – the reference CPU path (autocast("cuda") is disabled on a CPU host, so
    the reference computes pure fp32).
  * ``mode="bf16"``  – hand-applied cast points of the reference GPU path under
    ``torch.autocast("cuda", bf16)`` (SURVEY.md §2.3 "Cast points"): every Linear rounds its
    input and weight to bf16, accumulates in fp32 and rounds the output to bf16; residual
    stream, LayerNorm, RoPE, softmax, log1p and the loss reductions are fp32; GELU/GeGLU
    products are evaluated on bf16 tensors; ``torch.mm`` in InfoNCE rounds both operands to
    bf16.  This is the mode the HIP kernels are compared against at tight tolerance.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

BF16 = torch.bfloat16


# --------------------------------------------------------------------------------------
# configuration (field names follow huggingface/v33/config.json)
# --------------------------------------------------------------------------------------
@dataclass
class EncoderConfig:
    vocab_size: int = 50000
    hidden_size: int = 768
    intermediate_size: int = 1152
    num_hidden_layers: int = 22
    num_attention_heads: int = 12
    global_attn_every_n_layers: int = 3
    local_attention: int = 128          # total window; half-window = local_attention // 2
    global_rope_theta: float = 160000.0
    local_rope_theta: float = 10000.0
    norm_eps: float = 1e-5
    pad_token_id: int = 49999
    initializer_range: float = 0.02
    initializer_cutoff_factor: float = 2.0

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def sliding_window(self) -> int:
        return self.local_attention // 2

    def is_global(self, layer: int) -> bool:
        return layer % self.global_attn_every_n_layers == 0

    @staticmethod
    def tiny() -> "EncoderConfig":
        """The tiny parity config of SURVEY.md §7 (H64, 4 heads, I96, L4, V512, window +-4)."""
        return EncoderConfig(vocab_size=512, hidden_size=64, intermediate_size=96,
                             num_hidden_layers=4, num_attention_heads=4, local_attention=8,
                             pad_token_id=511)


def param_names(cfg: EncoderConfig) -> List[str]:
    """State-dict key order as seen through SPLADEModernBERT, tied decoder
    weight listed once (as the embedding)."""
    names = ["model.model.embeddings.tok_embeddings.weight", "model.model.embeddings.norm.weight"]
    for i in range(cfg.num_hidden_layers):
        p = f"model.model.layers.{i}."
        if i > 0:
            names.append(p + "attn_norm.weight")
        names += [p + "attn.Wqkv.weight", p + "attn.Wo.weight", p + "mlp_norm.weight",
                  p + "mlp.Wi.weight", p + "mlp.Wo.weight"]
    names += ["model.model.final_norm.weight", "model.head.dense.weight", "model.head.norm.weight",
              "model.decoder.bias"]
    return names


def param_shapes(cfg: EncoderConfig) -> Dict[str, Tuple[int, ...]]:
    H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    out: Dict[str, Tuple[int, ...]] = {}
    for n in param_names(cfg):
        if n.endswith("tok_embeddings.weight"):
            out[n] = (V, H)
        elif n.endswith("Wqkv.weight"):
            out[n] = (3 * H, H)
        elif n.endswith("attn.Wo.weight") or n.endswith("head.dense.weight"):
            out[n] = (H, H)
        elif n.endswith("Wi.weight"):
            out[n] = (2 * I, H)
        elif n.endswith("mlp.Wo.weight"):
            out[n] = (H, I)
        elif n.endswith("decoder.bias"):
            out[n] = (V,)
        else:
            out[n] = (H,)
    return out


def init_params(cfg: EncoderConfig, seed: int = 42) -> Dict[str, torch.Tensor]:
    """Random init following hf:353-390: trunc-normal(+-cutoff*std); std=initializer_range for
    embeddings/Wqkv/Wi, initializer_range/sqrt(2L) for attn.Wo/mlp.Wo/head.dense; LN weights 1;
    decoder bias 0.  (The draw order is this file's own; golden fixtures load THESE tensors
    into the reference module, so both sides see identical weights.)"""
    g = torch.Generator().manual_seed(seed)
    std_in = cfg.initializer_range
    std_out = cfg.initializer_range / math.sqrt(2.0 * cfg.num_hidden_layers)
    cut = cfg.initializer_cutoff_factor
    out: Dict[str, torch.Tensor] = {}
    for n, shp in param_shapes(cfg).items():
        if len(shp) == 1:
            out[n] = torch.zeros(shp) if n.endswith("bias") else torch.ones(shp)
            continue
        std = std_in if (n.endswith("tok_embeddings.weight") or n.endswith("Wqkv.weight")
                         or n.endswith("Wi.weight")) else std_out
        t = torch.empty(shp)
        torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-cut * std, b=cut * std, generator=g)
        out[n] = t
    return out


def perturb_params(params: Dict[str, torch.Tensor], seed: int = 7, ln_jitter: float = 0.2,
                   bias_std: float = 0.05, scale: float = 1.0, bias_mean: float = 0.0) -> Dict[str, torch.Tensor]:
    """Make LN weights / decoder bias non-trivial (so tests exercise them) and optionally scale
    the matrices (larger logits -> denser, less degenerate sparse vectors)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for n, t in params.items():
        if t.dim() == 1 and n.endswith("bias"):
            out[n] = torch.randn(t.shape, generator=g) * bias_std + bias_mean
        elif t.dim() == 1:
            out[n] = 1.0 + (torch.rand(t.shape, generator=g) - 0.5) * 2 * ln_jitter
        else:
            out[n] = t * scale
    return out


# --------------------------------------------------------------------------------------
# numeric-mode helpers
# --------------------------------------------------------------------------------------
def _r(x: torch.Tensor, mode: str) -> torch.Tensor:
    """Round to bf16 precision (kept as a bf16 tensor so autograd rounds the gradient too,
    as it does for a bf16 activation under autocast)."""
    return x.to(BF16) if mode == "bf16" else x


# Test hook (tests/test_oracle_ulp_floor.py): a torch.Generator seed.  When set, every bf16-mode Linear sums its
# contraction index in a permuted order -- the same products, another fp32 summation order, i.e. a SECOND correct bf16
# implementation of the reference's arithmetic, which is what the ULP / gate-flip tolerances of the GPU parity tests
# have to allow for.
CONTRACTION_PERM_SEED: Optional[int] = None


def _linear(x: torch.Tensor, w: torch.Tensor, mode: str, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.Linear under autocast: bf16 operands, fp32 accumulate, bf16 result (bias added in
    fp32 before the single rounding)."""
    if mode == "fp32":
        return F.linear(x, w, bias)
    if CONTRACTION_PERM_SEED is not None:
        perm = torch.randperm(w.shape[1], generator=torch.Generator().manual_seed(CONTRACTION_PERM_SEED + w.shape[1]))
        y = x.to(BF16).float()[..., perm] @ w.to(BF16).float()[:, perm].t()
    else:
        y = x.to(BF16).float() @ w.to(BF16).float().t()
    if bias is not None:
        y = y + bias.to(BF16).float()
    return y.to(BF16)


def _layer_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    # autocast runs layer_norm in fp32; output fp32 (hf:61,312,314,420,487; bias=False)
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), None, eps)


def _gelu(x: torch.Tensor, mode: str) -> torch.Tensor:
    # exact-erf GELU evaluated on the tensor's own dtype (bf16 tensors: fp32 math, one rounding)
    return _r(F.gelu(x.float()), mode) if mode == "bf16" else F.gelu(x)


def rope_tables(S: int, head_dim: int, theta: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """hf:136-163: inv_freq = theta^(-2i/d), emb = cat(freqs, freqs), fp32 cos/sin [S, d]."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    pos = torch.arange(S, dtype=torch.float)
    freqs = pos[:, None] * inv_freq[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def _rotate_half(x: torch.Tensor) -> torch.Tensor:
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def _apply_rope(t: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    # hf:214-219: up-cast to fp32, rotate, cast back to the input dtype
    tf = t.float()
    return (tf * cos + _rotate_half(tf) * sin).to(t.dtype)


def attention_bias(attention_mask: torch.Tensor, window: Optional[int]) -> torch.Tensor:
    """Boolean visibility [B, 1, S, S]: key j visible to query i iff mask[b, j] == 1 and (global
    or |i - j| <= window).  masking_utils.py:141-150 (bidirectional, inclusive distance)."""
    B, S = attention_mask.shape
    vis = attention_mask.bool()[:, None, None, :].expand(B, 1, S, S)
    if window is not None:
        idx = torch.arange(S)
        band = (idx[:, None] - idx[None, :]).abs() <= window
        vis = vis & band[None, None]
    return vis


def _attention(q, k, v, vis, scale: float, mode: str) -> torch.Tensor:
    """softmax(q k^T * scale + mask) v, bidirectional (hf:166-185 / SDPA).  Rows with no visible
    key produce a finite (uniform) result like the reference's finfo.min masking; they are
    padded queries and are zeroed by the SPLADE mask later."""
    s = (q.float() @ k.float().transpose(-1, -2)) * scale
    s = s.masked_fill(~vis, torch.finfo(torch.float32).min)
    if mode == "bf16":
        # flash-style cast points of the HIP kernel: fp32 scores, p~ = exp(s - m) rounded to
        # bf16 for the PV product, fp32 normaliser from the unrounded p~.
        m = s.max(dim=-1, keepdim=True).values
        p = torch.exp(s - m)
        l = p.sum(dim=-1, keepdim=True)
        o = (p.to(BF16).float() @ v.float()) / l
        return o.to(BF16)
    p = torch.softmax(s, dim=-1)
    return p @ v


def encoder_logits(params: Dict[str, torch.Tensor], cfg: EncoderConfig, input_ids: torch.Tensor,
                   attention_mask: torch.Tensor, mode: str = "fp32",
                   return_hidden: bool = False):
    """ModernBertForMaskedLM.forward(...).logits  -> [B, S, V] (bf16 tensor in bf16 mode)."""
    P = lambda n: params["model." + n]  # noqa: E731
    B, S = input_ids.shape
    H, nh, hd = cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim
    E = P("model.embeddings.tok_embeddings.weight")
    h = _layer_norm(F.embedding(input_ids, E), P("model.embeddings.norm.weight"), cfg.norm_eps)
    tabs = {True: rope_tables(S, hd, cfg.global_rope_theta), False: rope_tables(S, hd, cfg.local_rope_theta)}
    vis = {True: attention_bias(attention_mask, None), False: attention_bias(attention_mask, cfg.sliding_window)}
    for i in range(cfg.num_hidden_layers):
        pre = f"model.layers.{i}."
        g = cfg.is_global(i)
        x = h if i == 0 else _layer_norm(h, P(pre + "attn_norm.weight"), cfg.norm_eps)
        qkv = _linear(x, P(pre + "attn.Wqkv.weight"), mode).view(B, S, 3, nh, hd)
        q, k, v = (t.transpose(1, 2) for t in qkv.unbind(dim=2))          # [B, nh, S, hd]
        cos, sin = tabs[g]
        q, k = _apply_rope(q, cos, sin), _apply_rope(k, cos, sin)
        a = _attention(q, k, v, vis[g], hd ** -0.5, mode)                  # [B, nh, S, hd]
        a = a.transpose(1, 2).reshape(B, S, H)
        h = h + _linear(a, P(pre + "attn.Wo.weight"), mode)                # fp32 residual
        x = _layer_norm(h, P(pre + "mlp_norm.weight"), cfg.norm_eps)
        u = _linear(x, P(pre + "mlp.Wi.weight"), mode)
        a_in, gate = u.chunk(2, dim=-1)
        y = _gelu(a_in, mode) * gate                                       # bf16*bf16 -> bf16
        h = h + _linear(y, P(pre + "mlp.Wo.weight"), mode)
    h = _layer_norm(h, P("model.final_norm.weight"), cfg.norm_eps)
    d = _gelu(_linear(h, P("head.dense.weight"), mode), mode)
    hd_out = _layer_norm(d, P("head.norm.weight"), cfg.norm_eps)
    logits = _linear(hd_out, E, mode, bias=P("decoder.bias"))
    if return_hidden:
        return logits, hd_out
    return logits


def splade_forward(params, cfg, input_ids, attention_mask, mode: str = "fp32", route_rows=None):
    """SPLADEModernBERT.forward (ref:src/model/splade_modern.py:50-88) ->
    (sparse_repr [B, V] fp32, token_weights [B, S] fp32).
    ``route_rows`` [B, V] (test-only): take the value at the given sequence position instead of
    the max -- pins the max-pool routing so that gradients of two bf16 implementations can be
    compared tightly (a 1-ulp logit difference otherwise re-routes tied entries)."""
    logits = encoder_logits(params, cfg, input_ids, attention_mask, mode)
    s = torch.log1p(torch.relu(logits).float())          # log1p autocasts to fp32
    s = s * attention_mask.unsqueeze(-1).float()
    if route_rows is not None:
        sparse_repr = torch.gather(s, 1, route_rows.clamp(0, s.shape[1] - 1).unsqueeze(1)).squeeze(1)
        return sparse_repr, s.max(dim=-1).values
    sparse_repr = s.max(dim=1).values
    token_weights = s.max(dim=-1).values
    return sparse_repr, token_weights


# --------------------------------------------------------------------------------------
# SPLADELossV33 (ref:src/model/losses.py)
# --------------------------------------------------------------------------------------
@dataclass
class LossConfig:
    lambda_q: float = 1e-2
    lambda_d: float = 3e-3
    temperature: float = 1.0
    flops_warmup_steps: int = 20000
    lambda_kd: float = 0.0
    kd_temperature: float = 1.0
    lambda_initial_ratio: float = 0.1
    lambda_margin_mse: float = 0.0
    lambda_neg: float = 0.0


def lambda_schedule(step: int, target: float, warmup: int, r0: float) -> float:
    # ref:losses.py:75-90
    if step >= warmup:
        return target
    ratio = step / max(warmup, 1)
    return target * (r0 + (1.0 - r0) * ratio * ratio)


def flops_loss(w: torch.Tensor) -> torch.Tensor:
    # ref:losses.py:57-73
    return (w.mean(dim=0) ** 2).sum()


def loss_v33(lc: LossConfig, anchor, positive, negative, global_step: int = 0,
             teacher_pos_scores=None, teacher_neg_scores=None, mode: str = "fp32", teacher_scores=None):
    """SPLADELossV33.forward (ref:losses.py:183-297) -> (loss, dict of python floats).
    ``mode="bf16"`` rounds the two operands of the in-batch ``torch.mm`` to bf16 (and its
    result to bf16), as autocast does on the reference GPU path (ref:losses.py:155)."""
    B = anchor.shape[0]
    if mode == "bf16":
        inb = (anchor.to(BF16).float() @ positive.to(BF16).float().t()).to(BF16).float()
    else:
        inb = anchor @ positive.t()
    inb = inb / lc.temperature
    if negative.dim() == 3:
        hard = (anchor.unsqueeze(1) * negative).sum(-1) / lc.temperature        # [B, k]
    else:
        hard = ((anchor * negative).sum(-1) / lc.temperature).unsqueeze(1)       # [B, 1]
    scores = torch.cat([inb, hard], dim=1)
    infonce = F.cross_entropy(scores, torch.arange(B))
    fq, fd = flops_loss(anchor), flops_loss(positive)
    fneg = flops_loss(negative.reshape(-1, negative.shape[-1]))
    lam_neg_target = lc.lambda_neg if lc.lambda_neg > 0 else lc.lambda_d         # ref:losses.py:50
    lq = lambda_schedule(global_step, lc.lambda_q, lc.flops_warmup_steps, lc.lambda_initial_ratio)
    ld = lambda_schedule(global_step, lc.lambda_d, lc.flops_warmup_steps, lc.lambda_initial_ratio)
    ln = lambda_schedule(global_step, lam_neg_target, lc.flops_warmup_steps, lc.lambda_initial_ratio)
    loss = infonce + lq * fq + ld * fd + ln * fneg
    kd = torch.tensor(0.0)
    if lc.lambda_kd > 0 and teacher_scores is not None:                         # ref:losses.py:239-253
        if mode == "bf16":
            student = (anchor.to(BF16).float() @ positive.to(BF16).float().t()).to(BF16).float()
        else:
            student = anchor @ positive.t()
        student = student / lc.kd_temperature
        kd = F.kl_div(F.log_softmax(student, dim=-1), F.softmax(teacher_scores / lc.kd_temperature, dim=-1),
                      reduction="batchmean")
        loss = loss + lc.lambda_kd * kd
    mmse = torch.tensor(0.0)
    if lc.lambda_margin_mse > 0 and teacher_pos_scores is not None and teacher_neg_scores is not None:
        sp = (anchor * positive).sum(-1)
        if negative.dim() == 3:
            sn = (anchor.unsqueeze(1) * negative).sum(-1)
            sm, tm = sp.unsqueeze(1) - sn, teacher_pos_scores.unsqueeze(1) - teacher_neg_scores
        else:
            sn = (anchor * negative).sum(-1)
            sm, tm = sp - sn, teacher_pos_scores - teacher_neg_scores
        mmse = F.mse_loss(sm, tm)
        loss = loss + lc.lambda_margin_mse * mmse
    with torch.no_grad():
        nzq = (anchor > 0).float().sum(-1).mean()
        nzd = (positive > 0).float().sum(-1).mean()
    d = {"infonce": infonce.item(), "flops_q": fq.item(), "flops_d": fd.item(), "flops_neg": fneg.item(),
         "lambda_q": lq, "lambda_d": ld, "lambda_neg": ln, "kd": float(kd.item()), "margin_mse": float(mmse.item()),
         "nonzero_q": nzq.item(), "nonzero_d": nzd.item()}
    return loss, d


def cross_rank_infonce(anchor_local, positive_all, negative_local, rank: int, temperature: float):
    """Config-4 identity: in-batch scores against the all-gathered positives,
    label = rank*B + arange(B).  Not in the reference; defined by this identity."""
    B = anchor_local.shape[0]
    inb = anchor_local @ positive_all.t() / temperature
    if negative_local.dim() == 3:
        hard = (anchor_local.unsqueeze(1) * negative_local).sum(-1) / temperature
    else:
        hard = ((anchor_local * negative_local).sum(-1) / temperature).unsqueeze(1)
    scores = torch.cat([inb, hard], dim=1)
    return F.cross_entropy(scores, rank * B + torch.arange(B))


# --------------------------------------------------------------------------------------
# optimizer / schedule / one training epoch (ref:train_v33_ddp.py:289-374,560-592)
# --------------------------------------------------------------------------------------
def cosine_lr_factor(step: int, warmup: int, total: int, num_cycles: float = 0.5) -> float:
    # transformers/optimization.py:134-140
    if step < warmup:
        return float(step) / float(max(1, warmup))
    progress = float(step - warmup) / float(max(1, total - warmup))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


def weight_decay_of(name: str, wd: float) -> float:
    """ref:train_v33_ddp.py:560-577 -- the no-decay substrings only ever match ``decoder.bias``
    under ModernBERT naming, so every LayerNorm weight IS decayed (quirk kept)."""
    no_decay = ("bias", "LayerNorm.weight", "layer_norm.weight")
    return 0.0 if any(nd in name for nd in no_decay) else wd


@dataclass
class TrainState:
    params: Dict[str, torch.Tensor]
    exp_avg: Dict[str, torch.Tensor] = field(default_factory=dict)
    exp_avg_sq: Dict[str, torch.Tensor] = field(default_factory=dict)
    opt_step: int = 0


def adamw_step(st: TrainState, grads: Dict[str, torch.Tensor], lr: float, wd: float,
               clip: float, beta1=0.9, beta2=0.999, eps=1e-8) -> float:
    """clip_grad_norm_(max_norm=clip) then torch.optim.AdamW (decoupled decay). Returns the
    pre-clip global L2 norm."""
    total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
    coef = min(1.0, clip / (total + 1e-6))
    st.opt_step += 1
    t = st.opt_step
    for n, p in st.params.items():
        g = grads[n] * coef
        if n not in st.exp_avg:
            st.exp_avg[n] = torch.zeros_like(p)
            st.exp_avg_sq[n] = torch.zeros_like(p)
        p.mul_(1.0 - lr * weight_decay_of(n, wd))
        st.exp_avg[n].mul_(beta1).add_(g, alpha=1 - beta1)
        st.exp_avg_sq[n].mul_(beta2).addcmul_(g, g, value=1 - beta2)
        bc1, bc2 = 1 - beta1 ** t, 1 - beta2 ** t
        denom = (st.exp_avg_sq[n].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(st.exp_avg[n], denom, value=-lr / bc1)
    return total


def train_micro_steps(cfg: EncoderConfig, lc: LossConfig, st: TrainState, batches: List[dict],
                      grad_accum: int, base_lr: float, wd: float, clip: float, warmup: int,
                      total_steps: int, global_step: int = 0, mode: str = "fp32",
                      world_grads_hook=None, route_rows=None, grads_out: Optional[list] = None,
                      free_losses_out: Optional[list] = None):
    """The micro-batch loop of ``train_epoch`` (ref:train_v33_ddp.py:316-374) on plain tensors.
    Returns (per-micro-step losses, per-micro-step loss dicts, global_step).
    ``route_rows`` (test-only): per batch a (query, positive, negative) triple of [B, V] max-pool routings
    to pin (see ``splade_forward``); ``grads_out``: receives the accumulated gradient dict of every
    optimizer step (before clipping); ``free_losses_out``: with pinned routing, also receives each
    micro-step's (loss, loss dict) under the oracle's OWN (free) routing at the same parameters."""
    losses, dicts = [], []
    acc: Dict[str, torch.Tensor] = {}
    for bi, b in enumerate(batches):
        leaves = {n: p.detach().clone().requires_grad_(True) for n, p in st.params.items()}
        rq, rp, rn = route_rows[bi] if route_rows is not None else (None, None, None)
        q, _ = splade_forward(leaves, cfg, b["query_input_ids"], b["query_attention_mask"], mode, rq)
        p_, _ = splade_forward(leaves, cfg, b["positive_input_ids"], b["positive_attention_mask"], mode, rp)
        n_, _ = splade_forward(leaves, cfg, b["negative_input_ids"], b["negative_attention_mask"], mode, rn)
        k = int(b.get("num_negatives", 1))
        if k > 1:
            n_ = n_.view(q.shape[0], k, -1)
        loss, d = loss_v33(lc, q, p_, n_, global_step, b.get("teacher_pos_scores"),
                           b.get("teacher_neg_scores"), mode)
        (loss / grad_accum).backward()
        if free_losses_out is not None:
            with torch.no_grad():
                fq = splade_forward(leaves, cfg, b["query_input_ids"], b["query_attention_mask"], mode)[0]
                fp = splade_forward(leaves, cfg, b["positive_input_ids"], b["positive_attention_mask"], mode)[0]
                fn_ = splade_forward(leaves, cfg, b["negative_input_ids"], b["negative_attention_mask"], mode)[0]
                if k > 1:
                    fn_ = fn_.view(fq.shape[0], k, -1)
                fl, fd = loss_v33(lc, fq, fp, fn_, global_step, b.get("teacher_pos_scores"),
                                  b.get("teacher_neg_scores"), mode)
                free_losses_out.append((float(fl), fd))
        for n, leaf in leaves.items():
            g = leaf.grad if leaf.grad is not None else torch.zeros_like(leaf)
            acc[n] = g if n not in acc else acc[n] + g
        losses.append(float(loss.item()))
        dicts.append(d)
        if (bi + 1) % grad_accum == 0:
            if world_grads_hook is not None:
                acc = world_grads_hook(acc)
            if grads_out is not None:
                grads_out.append({n: g.clone() for n, g in acc.items()})
            lr = base_lr * cosine_lr_factor(global_step, warmup, total_steps)
            adamw_step(st, acc, lr, wd, clip)
            acc = {}
            global_step += 1
    return losses, dicts, global_step


# --------------------------------------------------------------------------------------
# synthetic inputs (SURVEY.md §8(d) "Synthetic inputs")
# --------------------------------------------------------------------------------------
def synth_ids(B: int, S: int, cfg: EncoderConfig, gen: torch.Generator, ragged: bool,
              min_len: int = 8) -> Tuple[torch.Tensor, torch.Tensor]:
    """ids: position 0 = 0 (<s>), last valid = 1 (eos), interior randint(6, pad); right-padded
    with pad id, mask 0 on pad.  ragged=False -> full length."""
    ids = torch.randint(6, cfg.pad_token_id, (B, S), generator=gen)
    if ragged:
        lens = torch.randint(min(min_len, S), S + 1, (B,), generator=gen)
    else:
        lens = torch.full((B,), S)
    mask = (torch.arange(S)[None, :] < lens[:, None]).long()
    ids[:, 0] = 0
    ids[torch.arange(B), lens - 1] = 1
    ids = torch.where(mask.bool(), ids, torch.full_like(ids, cfg.pad_token_id))
    return ids, mask


def synth_batch(B: int, Sq: int, Sd: int, cfg: EncoderConfig, gen: torch.Generator, k: int = 1,
                ragged: bool = False, teacher: bool = False) -> dict:
    q, qm = synth_ids(B, Sq, cfg, gen, ragged)
    p, pm = synth_ids(B, Sd, cfg, gen, ragged)
    n, nm = synth_ids(B * k, Sd, cfg, gen, ragged)
    out = {"query_input_ids": q, "query_attention_mask": qm, "positive_input_ids": p,
           "positive_attention_mask": pm, "negative_input_ids": n, "negative_attention_mask": nm,
           "num_negatives": k}
    if teacher:
        out["teacher_pos_scores"] = 0.5 + 0.5 * torch.rand(B, generator=gen)
        tn = 0.6 * torch.rand(B, k, generator=gen)
        out["teacher_neg_scores"] = tn if k > 1 else tn[:, 0]
    return out


# ---------------------------------------------------------------------------------------------------
# Inference post-processing (ref:benchmark/encoders.py:309-345, NeuralSparseEncoderV33._encode_batch)
# ---------------------------------------------------------------------------------------------------
def encode_postprocess(rep_row, tokens, special_ids, top_k=None):
    """One row of sparse_repr -> ordered list of (token, weight), restating ref:encoders.py:320-343:
    non-zero (> 0) entries in vocab-id order (:322-323), special ids skipped (:327-328), tokens that are empty
    or start with "[" / "<" skipped (:331), dict semantics for duplicate token strings (:332), and when more
    than top_k entries remain the top_k by weight with Python's stable sort (:334-340) -- i.e. ties keep
    vocab-id order."""
    special = set(int(s) for s in special_ids)
    d = {}
    for idx, w in enumerate(rep_row):
        w = float(w)
        if not w > 0:
            continue
        if idx in special:
            continue
        tok = tokens[idx]
        if tok and not tok.startswith(("[", "<")):
            d[tok] = w
    if top_k is not None and len(d) > top_k:
        d = dict(sorted(d.items(), key=lambda kv: kv[1], reverse=True)[:top_k])
    return list(d.items())


def allowed_vocab_mask(tokens, special_ids):
    """uint8 [V]: the per-id part of the filter above (what the device kernel takes)."""
    special = set(int(s) for s in special_ids)
    return [int(i not in special and bool(t) and not t.startswith(("[", "<"))) for i, t in enumerate(tokens)]
