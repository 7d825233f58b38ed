"""Where the free-routing gradient thresholds of tests/test_gpu_model.py come from (CPU only).

`sparse_repr = max over the sequence` routes each (sequence, vocab) gradient to ONE token position.  In bf16
the logits of neighbouring positions tie or nearly tie, so two implementations that differ by one bf16 ulp in
a logit send that gradient to different rows.  This test measures the effect on the ORACLE ITSELF -- emulated
bf16 vs fp32, same weights, same batches as the GPU test -- and pins it: the worst per-tensor cosine is
0.989-0.994 and the worst relative L2 error 0.11-0.15.  A bf16 implementation therefore cannot be held to
the protocol's cos >= 0.999 / rel <= 2e-2 under FREE routing (the GPU test uses 0.98 / 0.2 there); the tight
bound is enforced where the routing is pinned."""
import pytest
import torch

from oracle import splade_oracle as O


def _grads(mode, k, Sq, Sd, margin):
    cfg = O.EncoderConfig(vocab_size=1000, hidden_size=256, intermediate_size=384, num_hidden_layers=4,
                          num_attention_heads=4, local_attention=16, pad_token_id=999)
    params = O.perturb_params(O.init_params(cfg, seed=3), seed=4, scale=2.0, bias_mean=-0.1)
    gen = torch.Generator().manual_seed(100 + k)
    b = O.synth_batch(6, Sq, Sd, cfg, gen, k=k, ragged=True, teacher=margin > 0)
    lc = O.LossConfig(lambda_q=0.01, lambda_d=0.003, temperature=20.0, flops_warmup_steps=50,
                      lambda_initial_ratio=0.1, lambda_margin_mse=margin)
    leaves = {n: p.clone().requires_grad_(True) for n, p in params.items()}
    outs = [O.splade_forward(leaves, cfg, b[t + "_input_ids"], b[t + "_attention_mask"], mode)[0]
            for t in ("query", "positive", "negative")]
    n3 = outs[2].view(6, k, -1) if k > 1 else outs[2]
    loss, _ = O.loss_v33(lc, outs[0], outs[1], n3, 20, b.get("teacher_pos_scores"), b.get("teacher_neg_scores"), mode)
    loss.backward()
    return {n: l.grad for n, l in leaves.items()}


@pytest.mark.parametrize("k,margin,Sq,Sd", [(1, 0.0, 8, 12), (1, 0.0, 40, 150), (2, 0.05, 40, 150)])
def test_bf16_vs_fp32_free_routing_gradient_floor(k, margin, Sq, Sd):
    a, b = _grads("bf16", k, Sq, Sd, margin), _grads("fp32", k, Sq, Sd, margin)
    worst_cos, worst_rel = 1.0, 0.0
    for n in a:
        g, r = a[n].double().flatten(), b[n].double().flatten()
        worst_cos = min(worst_cos, float(g @ r / (g.norm() * r.norm() + 1e-30)))
        worst_rel = max(worst_rel, float((g - r).norm() / (r.norm() + 1e-30)))
    # the floor is far from the tight protocol bound ...
    assert worst_cos < 0.997 and worst_rel > 0.05, (worst_cos, worst_rel)
    # ... and inside the free-routing thresholds the GPU test uses
    assert worst_cos >= 0.98 and worst_rel <= 0.2, (worst_cos, worst_rel)
