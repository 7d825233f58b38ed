"""Where the VALUE tolerance (tests/helpers.assert_ulp_statement) and the decoder.bias gradient bound of
tests/test_gpu_parity_full.py come from -- on the CPU, from the oracle alone.

Two evaluations of the oracle's bf16 mode on the 149 M model, identical but for the ORDER in which every Linear sums
its contraction index (oracle.CONTRACTION_PERM_SEED: same products, another fp32 summation order) are two correct
implementations of the reference's arithmetic.  Whatever separates them is a property of bf16 at this depth (22 layers,
768-wide rows), not of any kernel:

  * values: the distribution of bf16-ulp distances between the two sets of sparse outputs must lie INSIDE the bounds
    assert_ulp_statement enforces on the HIP path -- and be non-trivial (a large share of logits DOES move by an ulp),
    which is why SURVEY 8(d)(ii)'s "<= 1e-3 abs" cannot be the test;
  * decoder.bias: its gradient is the plain sum over tokens of the routed coefficients g / (1 + x) [x > 0]; an entry whose
    logit lies within the bf16 noise of zero has its relu gate open in one evaluation and shut in the other.  Under
    PINNED routing (both back-propagate through the same arg-max rows) the relative L2 distance of the two decoder.bias
    gradients is measured here, and the same distance with the entries whose logit is within 6 sigma of zero masked
    out: the first EXCEEDS the protocol's 2e-2 (so the unmasked tensor cannot be held to it by any correct bf16
    implementation), the second obeys it -- which is the form tests/test_gpu_parity_full.py asserts."""
import os

import pytest
import torch

from oracle import splade_oracle as O
from tests.helpers import assert_ulp_statement, sparse_ulp_stats

SIGMA = 2e-3                     # absolute noise of a decoder logit between two correct bf16 evaluations


@pytest.fixture(scope="module")
def pair():
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cfg = O.EncoderConfig()
    params = O.perturb_params(O.init_params(cfg, seed=42), seed=7, bias_mean=-0.2)     # the GPU parity tests' weights
    gen = torch.Generator().manual_seed(5151)
    b = O.synth_batch(2, 64, 192, cfg, gen, k=2, ragged=True, teacher=True)
    lc = O.LossConfig(lambda_q=0.01, lambda_d=0.003, temperature=1.0, flops_warmup_steps=5000, lambda_margin_mse=0.5,
                      lambda_initial_ratio=0.5)                                          # config 5's loss (golden g7)

    def run(perm_seed, rows):
        O.CONTRACTION_PERM_SEED = perm_seed
        try:
            bias = params["model.decoder.bias"].clone().requires_grad_(True)
            leaves = dict(params, **{"model.decoder.bias": bias})
            outs, logits, free_rows = {}, {}, {}
            for tag in ("query", "positive", "negative"):
                ids, mask = b[tag + "_input_ids"], b[tag + "_attention_mask"]
                with torch.no_grad():
                    lg = O.encoder_logits(params, cfg, ids, mask, "bf16").float()
                    s = torch.log1p(torch.relu(lg)) * mask.unsqueeze(-1).float()
                    free_rows[tag] = s.argmax(dim=1)
                    logits[tag] = lg
                outs[tag] = O.splade_forward(leaves, cfg, ids, mask, "bf16",
                                             route_rows=(rows or free_rows)[tag])[0]
            n3 = outs["negative"].view(2, 2, -1)
            loss, _ = O.loss_v33(lc, outs["query"], outs["positive"], n3, 1000, b["teacher_pos_scores"],
                                 b["teacher_neg_scores"], "bf16")
            loss.backward()
            return {k: v.detach() for k, v in outs.items()}, logits, free_rows, bias.grad.clone()
        finally:
            O.CONTRACTION_PERM_SEED = None
    a = run(None, None)
    c = run(1234, a[2])                                    # routing pinned to the first evaluation's arg-max rows
    return a, c


def test_two_correct_bf16_evaluations_reproduce_the_ulp_distribution(pair):
    (oa, _, _, _), (ob, _, _, _) = pair
    for tag in ("query", "positive", "negative"):
        st = sparse_ulp_stats(ob[tag], oa[tag])
        assert_ulp_statement(st, tag)                      # inside the bounds the HIP path is held to ...
        # ... and far from "every value within 1e-3 / one ulp": a sizeable share of the logits lands on the neighbouring
        # bf16 value, the largest difference of the sparse values is several 1e-3
        assert 0.05 <= st["ulp1"] <= 0.40 and st["ulp0"] <= 0.95, (tag, st)
        assert st["max_abs"] >= 2e-3, (tag, st)


def test_decoder_bias_gradient_floor_is_the_relu_gate(pair):
    (_, la, rows, ga), (_, lb, _, gb) = pair
    rel = float((ga - gb).double().norm() / ga.double().norm())
    # entries (vocabulary ids) that own a routed logit within 6 sigma of zero in either evaluation: their gate may flip
    near = torch.zeros_like(ga, dtype=torch.bool)
    for tag in rows:
        for lg in (la[tag], lb[tag]):
            x = torch.gather(lg, 1, rows[tag].unsqueeze(1)).squeeze(1)           # [B, V] routed logits
            near |= (x.abs() <= 6 * SIGMA).any(dim=0)
    keep = ~near
    rel_kept = float((ga[keep] - gb[keep]).double().norm() / ga[keep].double().norm())
    print(f"decoder.bias pinned-routing rel-L2 between two correct bf16 evaluations: {rel:.3e}; without the "
          f"{int(near.sum())} ids whose routed logit is within 6 sigma of zero: {rel_kept:.3e}")
    assert rel_kept <= 2e-2, rel_kept                      # away from the gate the protocol's bound holds (measured 1.1e-2)
    assert rel >= 2e-2, rel                                # the gate alone breaks it (measured 4.0e-2): hence the mask
    assert int(near.sum()) <= 0.02 * near.numel()          # and the mask is a small minority of the ids (106 of 50,000)
