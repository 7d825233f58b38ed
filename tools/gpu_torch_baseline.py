#!/usr/bin/env python3
"""Reference-equivalent EAGER PyTorch baseline on the same MI355X (measurement only, not a product path).

What the reference runs on its GPUs (ref:src/train/cli/train_v33_ddp.py:316-374): three
`transformers` ModernBertForMaskedLM forwards under autocast(bf16) (ref:src/model/splade_modern.py:50-88),
the SPLADE tail, InfoNCE + FLOPS (ref:src/model/losses.py:57-73,136-181,183-297), backward, and every
4th micro-step clip + AdamW.  Here: the same third-party module built from the A.X-Encoder-base geometry
with random init (no hub access), the tail and the loss restated in a few lines of torch, synthetic
full-length q64/d256 batches of 64 triplets.  Prints triplets/s for DESIGN.md's comparison row."""
import argparse
import json
import time

import torch
import torch.nn.functional as F


def build_model(dev):
    from transformers import AutoModelForMaskedLM, ModernBertConfig
    cfg = ModernBertConfig(
        vocab_size=50000, hidden_size=768, intermediate_size=1152, num_hidden_layers=22, num_attention_heads=12,
        hidden_activation="gelu", max_position_embeddings=16384, norm_eps=1e-5, norm_bias=False, pad_token_id=49999,
        eos_token_id=1, bos_token_id=0, cls_token_id=0, sep_token_id=1, global_rope_theta=160000.0,
        local_rope_theta=10000.0, attention_bias=False, attention_dropout=0.0, global_attn_every_n_layers=3,
        local_attention=128, embedding_dropout=0.0, mlp_bias=False, mlp_dropout=0.0, decoder_bias=True,
        classifier_bias=False, classifier_activation="gelu", sparse_prediction=False, reference_compile=False,
        attn_implementation="sdpa")
    torch.manual_seed(42)
    return AutoModelForMaskedLM.from_config(cfg).to(dev)


def splade(model, ids, mask):                      # ref:src/model/splade_modern.py:69-86
    logits = model(input_ids=ids, attention_mask=mask).logits
    s = torch.log1p(torch.relu(logits)) * mask.unsqueeze(-1).float()
    return s.max(dim=1).values


def loss_fn(q, p, n, lam_q, lam_d):                # ref:src/model/losses.py:136-181 (in-batch + 1 hard negative), :57-73
    pos = q @ p.t()
    neg = (q * n).sum(-1, keepdim=True)
    ce = F.cross_entropy(torch.cat([pos, neg], dim=1), torch.arange(q.shape[0], device=q.device))
    flops = lambda w: (w.mean(0) ** 2).sum()
    return ce + lam_q * flops(q) + lam_d * (flops(p) + flops(n))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    model = build_model(dev)
    decay = [p for n, p in model.named_parameters() if "bias" not in n]
    nodecay = [p for n, p in model.named_parameters() if "bias" in n]
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.01}, {"params": nodecay, "weight_decay": 0.0}], lr=2e-5)
    B, g = args.batch, torch.Generator().manual_seed(42)

    def batch(S):
        ids = torch.randint(6, 49999, (B, S), generator=g)
        ids[:, 0], ids[:, -1] = 0, 1
        return ids.to(dev), torch.ones(B, S, dtype=torch.long, device=dev)
    data = [(batch(64), batch(256), batch(256)) for _ in range(4)]

    def step(i):
        (qi, qm), (pi, pm), (ni, nm) = data[i % 4]
        with torch.autocast("cuda", dtype=torch.bfloat16):
            q, p, n = splade(model, qi, qm), splade(model, pi, pm), splade(model, ni, nm)
            loss = loss_fn(q.float(), p.float(), n.float(), 0.001, 0.0003)
        (loss / 4).backward()
        if (i + 1) % 4 == 0:
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
            opt.zero_grad(set_to_none=True)
        return loss

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"what": "eager PyTorch + transformers ModernBERT (sdpa, autocast bf16), same shapes",
                      "triplets_per_s": args.steps * B / dt, "ms_per_step": 1e3 * dt / args.steps,
                      "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30, "final_loss": float(loss),
                      "torch": torch.__version__}))


if __name__ == "__main__":
    main()
