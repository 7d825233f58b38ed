"""Full-size (149 M) parity of the HIP path at the SURVEY 8(d) protocol, incl. BASELINE config 5.

  * values: an ULP statement (tests/helpers.sparse_ulp_stats) vs the oracle in emulated-bf16 mode;
  * loss terms rel <= 2e-3 and per-tensor gradients cos >= 0.999 / rel-L2 <= 2e-2 vs the oracle back-propagating
    through the SAME max-pool routing the HIP forward chose (the only discontinuity of the path);
  * the reference's own fp32 outputs (goldens g7 = config 5, g8 = g3 batch with an unsaturated InfoNCE) beside
    it, at the looser bf16-vs-fp32 bounds;
  * a top-k fixture whose rank gaps exceed the value error, so that index equality is actually asserted.
Needs a real MI355X: pytest -m gpu."""
import json
import os

import numpy as np
import pytest
import torch

from tests.helpers import assert_ulp_statement, sparse_ulp_stats, topk_rank_check

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
OUT = os.path.join(os.path.dirname(os.path.dirname(__file__)), "gpurun_out")

@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _report(name, obj):
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "parity_report.jsonl"), "a") as f:
        f.write(json.dumps({"test": name, **obj}) + "\n")


def _grad_stats(got, ref):
    g, r = got.double().flatten(), ref.double().flatten()
    return float((g @ r) / (g.norm() * r.norm() + 1e-30)), float((g - r).norm() / (r.norm() + 1e-30))


@pytest.fixture(scope="module")
def full(dev):
    from oracle import splade_oracle as O
    from tests.test_gpu_model import _build_model
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cfg = O.EncoderConfig()
    params = O.perturb_params(O.init_params(cfg, seed=42), seed=7, bias_mean=-0.2)
    model = _build_model(cfg, params, dev)
    model.runtime.keep_last_ctx = True
    return cfg, params, model


def _run_hip(model, dev, b, lkw, step, k):
    from src.model.losses import SPLADELossV33
    D = lambda t: t.to(dev) if torch.is_tensor(t) else t   # noqa: E731
    model.zero_grad(set_to_none=True)
    outs, tws, rows = {}, {}, {}
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        for tag in ("query", "positive", "negative"):
            outs[tag], tws[tag] = model(D(b[tag + "_input_ids"]), D(b[tag + "_attention_mask"]))
            rows[tag] = model.runtime.routing_rows(*model.runtime.last_ctx).cpu()
        B = outs["query"].shape[0]
        n3 = outs["negative"].view(B, k, -1) if k > 1 else outs["negative"]
        lf = SPLADELossV33(**lkw).to(dev)
        loss, d = lf(anchor_repr=outs["query"], positive_repr=outs["positive"], negative_repr=n3, global_step=step,
                     teacher_pos_scores=D(b.get("teacher_pos_scores")), teacher_neg_scores=D(b.get("teacher_neg_scores")))
    loss.backward()
    return outs, tws, rows, loss, d


def _run_oracle(cfg, params, b, lkw, step, k, rows):
    from oracle import splade_oracle as O
    leaves = {n: p.clone().requires_grad_(True) for n, p in params.items()}
    free, pinned = {}, {}
    for tag in ("query", "positive", "negative"):
        with torch.no_grad():
            free[tag] = O.splade_forward(params, cfg, b[tag + "_input_ids"], b[tag + "_attention_mask"], "bf16")
        pinned[tag] = O.splade_forward(leaves, cfg, b[tag + "_input_ids"], b[tag + "_attention_mask"], "bf16",
                                       route_rows=rows[tag])[0]
    B = pinned["query"].shape[0]
    n3 = pinned["negative"].view(B, k, -1) if k > 1 else pinned["negative"]
    lc = O.LossConfig(**lkw)
    loss, d = O.loss_v33(lc, pinned["query"], pinned["positive"], n3, step, b.get("teacher_pos_scores"),
                         b.get("teacher_neg_scores"), "bf16")
    loss.backward()
    for tag in ("query", "positive", "negative"):          # the values whose relu gate the oracle's gradient went through
        free[tag + "_pinned"] = pinned[tag].detach()
    return free, loss, d, {n: l.grad for n, l in leaves.items()}


def _check_against_oracle(name, model, outs, tws, loss, d, free, oloss, od, ograds, b):
    rep = {}
    for tag in ("query", "positive", "negative"):
        st = sparse_ulp_stats(outs[tag], free[tag][0])
        stw = sparse_ulp_stats(tws[tag], free[tag][1])
        rep[tag] = {"sparse": st, "token_weights": stw}
    rep["loss"] = {"got": float(loss.detach()), "oracle_bf16_pinned": float(oloss.detach()),
                   "terms": {k_: [float(d[k_]), od[k_]] for k_ in ("infonce", "flops_q", "flops_d", "flops_neg", "margin_mse")}}
    stats = {n_: _grad_stats(p.grad.cpu(), ograds[n_]) for n_, p in model.named_parameters()}
    worst = min(stats.items(), key=lambda kv: kv[1][0])
    rep["worst_grad"] = [worst[0], *worst[1]]
    rep["max_grad_rel"] = max(v[1] for v in stats.values())
    rep["grad_norm_ratio_minmax"] = [min(float(p.grad.double().norm() / ograds[n_].double().norm()) for n_, p in model.named_parameters()),
                                     max(float(p.grad.double().norm() / ograds[n_].double().norm()) for n_, p in model.named_parameters())]
    _report(name, rep)
    for tag in ("query", "positive", "negative"):
        assert torch.isfinite(outs[tag]).all()
        assert_ulp_statement(rep[tag]["sparse"], tag)
        assert_ulp_statement(rep[tag]["token_weights"], tag + " token_weights")
        assert (tws[tag].detach().cpu()[b[tag + "_attention_mask"] == 0] == 0).all()
    assert float(loss) == pytest.approx(float(oloss), rel=2e-3)
    for key in ("infonce", "flops_q", "flops_d", "flops_neg", "margin_mse"):
        assert float(d[key]) == pytest.approx(od[key], rel=2e-3, abs=2e-3), (key, float(d[key]), od[key])
    # rel-L2 <= 2e-2 for every tensor.  decoder.bias is held to it on the vocabulary ids AWAY FROM THE RELU GATE: its gradient
    # is the plain sum of the routed coefficients g / (1 + x) [x > 0], and an entry whose logit lies within the bf16 noise
    # of zero (sigma ~2e-3 absolute) has its gate open in one implementation and shut in the other, i.e.
    # contributes its whole g or nothing.  tests/test_oracle_ulp_floor.py measures it on the CPU between two correct bf16
    # evaluations of the ORACLE (same products, another fp32 summation order, pinned routing): rel-L2 4.0e-2 over all ids,
    # 1.1e-2 without the ~0.2 % of ids that own a routed logit within 6 sigma of zero -- so no bound on the unmasked tensor
    # below that floor means anything (rounds 2-3 measured 1.45e-2 .. 2.06e-2 and round 3 raised the bound to 3e-2 after a
    # red run: replaced by this).  The matrices average the effect away over their 768 columns and stay unmasked.
    gate = float(np.log1p(6 * 2e-3))
    near = torch.zeros(outs["query"].shape[-1], dtype=torch.bool)
    for tag in ("query", "positive", "negative"):
        a, o = outs[tag].detach().float().cpu(), free[tag + "_pinned"].float()
        near |= ((torch.maximum(a, o) > 0) & (torch.minimum(a, o) <= gate)).any(dim=0)
    keep = ~near
    gb = dict(model.named_parameters())["model.decoder.bias"].grad.detach().cpu()[keep]
    ob = ograds["model.decoder.bias"][keep]
    stats["model.decoder.bias"] = _grad_stats(gb, ob)
    rep["decoder_bias"] = {"ids_near_the_gate": int(near.sum()), "kept_cos_rel": list(stats["model.decoder.bias"])}
    assert int(near.sum()) <= 0.02 * near.numel(), int(near.sum())          # the mask stays a small minority of the ids
    bad = {n_: v for n_, v in stats.items() if v[0] < 0.999 or v[1] > 2e-2}
    assert not bad, bad
    return rep


def _check_against_reference(name, model, outs, loss, d, z, meta):
    """bf16 path vs the reference's fp32 golden: reported; bounds at the bf16-vs-fp32 background level
    (torch's own CPU bf16 autocast vs fp32 gives max 6.9e-3 / mean 1.4e-3 on this model)."""
    rep = {}
    for tag, key in (("q", "query"), ("p", "positive"), ("n", "negative")):
        ref_full = torch.from_numpy(z[f"out::{tag}_full"].astype(np.float32))
        sr = outs[key].detach().cpu()
        diff = (sr - ref_full).abs()
        ti = torch.topk(sr, 50, dim=-1).indices
        ri = torch.from_numpy(z[f"out::{tag}_topi"])[:, :50]
        overlap = float(np.mean([len(set(a.tolist()) & set(b_.tolist())) / 50 for a, b_ in zip(ti, ri)]))
        rep[tag] = {"max": float(diff.max()), "mean": float(diff.mean()), "top50_overlap": overlap}
        assert diff.max().item() < 1.5e-2 and diff.mean().item() < 2e-3, rep[tag]
        assert overlap > 0.85, rep[tag]
    rep["loss"] = {"got": float(loss.detach()), "reference_fp32": meta["loss"]}
    for key in ("flops_q", "flops_d", "flops_neg"):
        assert float(d[key]) == pytest.approx(meta["loss_dict"][key], rel=1e-2), key
    norms = dict(zip(meta["grad_names"], meta["grad_norms"]))
    ratios = {n_: float(p.grad.double().norm()) / max(norms[n_], 1e-30) for n_, p in model.named_parameters()}
    rep["grad_norm_ratio_minmax"] = [min(ratios.values()), max(ratios.values())]
    pg = dict(model.named_parameters())
    probe = {}
    for key in z.files:
        if key.startswith("gprobe::model"):
            g = pg[key[8:]].grad
            probe[key[8:]] = _grad_stats((g[:8, :64] if g.dim() == 2 else g[:512]).cpu(), torch.from_numpy(z[key]))
    rep["grad_probe_cos_rel"] = probe
    _report(name, rep)
    return rep


def _golden(name):
    z = np.load(os.path.join(G, name + ".npz"))
    meta = json.load(open(os.path.join(G, name + ".json")))
    b = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in::")}
    return z, meta, b


def test_config5_d512_k4_margin_mse(dev, full):
    """BASELINE config 5 at full size: q64 / d512, k = 4 negatives ([B*k, S] -> view [B, k, V],
    ref:train_v33_ddp.py:346-350), MarginMSE 0.5 with teacher scores (ref:configs/train_v34_multi_neg.yaml:20-28).
    512-token sequences run the streaming attention kernels (global layers: S = 512 keys; local: 129-key band)
    and four decoder row chunks."""
    cfg, params, model = full
    z, meta, b = _golden("g7_cfg5_d512_k4")
    k = meta["num_negatives"]
    outs, tws, rows, loss, d = _run_hip(model, dev, b, meta["loss_kwargs"], meta["global_step"], k)
    ref_rep = _check_against_reference("cfg5_vs_reference_fp32_g7", model, outs, loss, d, z, meta)
    # MarginMSE squares margins of a few thousand: the reference's fp32 value is reproduced by the bf16 path
    assert float(d["margin_mse"]) == pytest.approx(meta["loss_dict"]["margin_mse"], rel=5e-3), float(d["margin_mse"])
    assert float(loss) == pytest.approx(meta["loss"], rel=5e-3)
    for n_, (cos, rel) in ref_rep["grad_probe_cos_rel"].items():
        assert cos > 0.99, (n_, cos, rel)            # MarginMSE-dominated gradient: smooth in the sparse values
    free, oloss, od, ograds = _run_oracle(cfg, params, b, meta["loss_kwargs"], meta["global_step"], k, rows)
    _check_against_oracle("cfg5_vs_oracle_bf16", model, outs, tws, loss, d, free, oloss, od, ograds, b)
    model.zero_grad(set_to_none=True)


def test_unsaturated_infonce_full_size(dev, full):
    """The g3 batch (B=4, q64/d256) at tau = 500: InfoNCE carries information (g3's tau = 1 is a one-hot on
    dot products ~2e4), so loss and gradients discriminate.  vs oracle-bf16 under pinned routing at the
    protocol's bounds, and vs the reference's fp32 run (g8) for the gradient-norm question of round 1."""
    cfg, params, model = full
    z, meta, b = _golden("g8_full_unsaturated")
    outs, tws, rows, loss, d = _run_hip(model, dev, b, meta["loss_kwargs"], meta["global_step"], 1)
    ref_rep = _check_against_reference("unsaturated_vs_reference_fp32_g8", model, outs, loss, d, z, meta)
    assert float(d["infonce"]) == pytest.approx(meta["loss_dict"]["infonce"], abs=0.1)     # bf16 mm at |score| ~40
    free, oloss, od, ograds = _run_oracle(cfg, params, b, meta["loss_kwargs"], meta["global_step"], 1, rows)
    rep = _check_against_oracle("unsaturated_vs_oracle_bf16", model, outs, tws, loss, d, free, oloss, od, ograds, b)
    # against the matching-precision oracle there is no one-sided gradient-norm bias
    lo, hi = rep["grad_norm_ratio_minmax"]
    assert 0.98 < lo and hi < 1.02, (lo, hi)
    model.zero_grad(set_to_none=True)


def test_topk_indices_exact_on_a_fixture_that_bites(dev, full):
    """'top-k token indices bit-exact' (north star) is vacuous on near-flat random-init outputs (round 1: <1 % of
    the ranks were comparable).  Here 160 vocabulary ids get a decoder-bias ladder spaced 2 % apart in
    log1p space, i.e. well above one bf16 ulp of the logit, so that most of the top-64 ranks are separated
    by more than twice the value error and their INDICES must agree with the oracle."""
    from oracle import splade_oracle as O
    cfg, params, model = full
    gen = torch.Generator().manual_seed(4321)
    ladder_ids = torch.randperm(cfg.vocab_size - 10, generator=gen)[:160] + 6
    j = torch.arange(160, dtype=torch.float32)
    ladder = 3.5 * torch.exp(0.02 * j) - 1.0                      # logits 2.5 .. 84
    p2 = dict(params)
    bias = params["model.decoder.bias"].clone()
    bias[ladder_ids] = ladder
    p2["model.decoder.bias"] = bias
    old = model.model.decoder.bias.detach().clone()
    try:
        with torch.no_grad():
            model.model.decoder.bias.copy_(bias.to(dev))
        ids, mask = O.synth_ids(4, 256, cfg, torch.Generator().manual_seed(778), ragged=True)
        with torch.no_grad(), torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            ref, _ = O.splade_forward(p2, cfg, ids, mask, "bf16")
            got, _ = model(ids.to(dev), mask.to(dev))
    finally:
        with torch.no_grad():
            model.model.decoder.bias.copy_(old)
    st = sparse_ulp_stats(got, ref)
    tk = topk_rank_check(got, ref, 64, st["max_abs"])
    _report("topk_ladder", {"ulp": st, "topk": tk})
    assert st["far"] == 0, st
    assert tk["checked_frac"] >= 0.5, tk
    assert tk["equal"], tk


def test_sparse_regime_support_and_topk(dev, full):
    """The regime training lives in: a trained V33 model keeps ~54 vocabulary dimensions per
    document active (ref:huggingface/v33/README.md:240-245), random init keeps all 50,000.  One scalar shift of the decoder
    bias (the pooled logit of rank 55, averaged over the batch) puts the random-init model there without a hand-built
    ladder.  In that regime the INDEX statement of the north star is the sparse vector's SUPPORT: every dimension the
    oracle has active must be active here and vice versa, except where the oracle's logit lies within twice the measured
    value error of the ReLU gate; top-k ranks are compared wherever both neighbour gaps exceed twice the error
    (the tail of 50,000 near-Gaussian logits is spaced ~sigma / (rank * 4.6): only the first ranks can qualify -- the
    fraction is reported, the ladder fixture above is the test that bites on ranks).  Forward + the routed backward run
    on the sparse pattern (finite gradients; the gate zeroes every inactive entry)."""
    from oracle import splade_oracle as O
    cfg, params, model = full
    ids, mask = O.synth_ids(4, 256, cfg, torch.Generator().manual_seed(779), ragged=True)
    with torch.no_grad():
        ref0, _ = O.splade_forward(params, cfg, ids, mask, "bf16")
    shift = float(torch.topk(torch.expm1(ref0.double()), 55, dim=-1).values[:, -1].mean())
    p2 = dict(params)
    p2["model.decoder.bias"] = params["model.decoder.bias"] - shift
    old = model.model.decoder.bias.detach().clone()
    try:
        with torch.no_grad():
            model.model.decoder.bias.copy_(p2["model.decoder.bias"].to(dev))
            ref, _ = O.splade_forward(p2, cfg, ids, mask, "bf16")
        model.zero_grad(set_to_none=True)
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            got, _ = model(ids.to(dev), mask.to(dev))
        (got * torch.rand_like(got)).sum().backward()
        finite = all(torch.isfinite(p_.grad).all() for p_ in model.parameters())
        g_bias = model.model.decoder.bias.grad.detach().cpu()
    finally:
        with torch.no_grad():
            model.model.decoder.bias.copy_(old)
        model.zero_grad(set_to_none=True)
    got = got.detach().cpu()
    st = sparse_ulp_stats(got, ref)
    active_ref, active_got = ref > 0, got > 0
    n_ref = active_ref.sum(1).tolist()
    err = max(st["max_abs"], 1e-3)
    near_gate = torch.expm1(ref.double()).abs() <= 2 * err                 # |logit| within twice the error of the gate
    # an entry the oracle gated off has no recoverable logit: it may appear here only with a value inside the error
    flips = active_ref != active_got
    bad = flips & ~(near_gate & active_ref) & ~(active_got & ~active_ref & (got.double() <= 2 * err))
    k = max(8, min(min(n_ref), 48))
    tk = topk_rank_check(got, ref, k, st["max_abs"])
    _report("sparse_regime", {"bias_shift": -shift, "active_dims_oracle": n_ref, "active_dims_hip": active_got.sum(1).tolist(),
                              "support_flips": int(flips.sum()), "support_flips_outside_the_gate_band": int(bad.sum()),
                              "ulp": st, "topk_k": k, "topk": tk})
    # the regime: tens of active dimensions per document, not 50,000 (the batch is ragged: a 12-token document keeps fewer)
    assert 25 <= sum(n_ref) / len(n_ref) <= 110 and all(4 <= n <= 200 for n in n_ref), n_ref
    assert int(bad.sum()) == 0, int(bad.sum())
    assert st["far"] == 0 and st["max_abs"] <= 8e-3, st
    assert tk["equal"], tk
    assert finite
    # decoder.bias gradient lives only on active entries (the ReLU gate zeroes the rest)
    assert int((g_bias != 0).sum()) <= int(active_got.any(0).sum())


def test_full_size_values_ulp_statement(dev, full):
    """Forward only, 2 x 256 ragged tokens, default (near-flat) outputs: the ULP statement at full size."""
    from oracle import splade_oracle as O
    cfg, params, model = full
    ids, mask = O.synth_ids(2, 256, cfg, torch.Generator().manual_seed(777), ragged=True)
    with torch.no_grad(), torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        ref, ref_tw = O.splade_forward(params, cfg, ids, mask, "bf16")
        got, got_tw = model(ids.to(dev), mask.to(dev))
    st, stw = sparse_ulp_stats(got, ref), sparse_ulp_stats(got_tw, ref_tw)
    tk = topk_rank_check(got, ref, 64, st["max_abs"])
    _report("full_vs_oracle_bf16_ulp", {"sparse": st, "token_weights": stw, "topk": tk})
    assert_ulp_statement(st, "sparse")
    assert_ulp_statement(stw, "token_weights")
    assert tk["equal"], tk
    assert (got_tw.cpu()[mask == 0] == 0).all()


def test_two_runs_are_reproducible(dev, full):
    """SURVEY 5 'deterministic two-run bit-compare': same inputs twice through forward + backward.
    Forward outputs, token weights and the max-pool routing are BIT-identical (no atomics in the forward);
    the routed decoder backward is deterministic (bucket slots are assigned in vocabulary order), and since round 5
    so are the weight-gradient GEMMs, the LayerNorm weight gradients and the embedding gradient (ordered reductions
    through the backward's workspace, include/snx.h "det_reduce"; rounds 1-4: fp32 global atomics in arrival order,
    1.9e-7 relative L2 between two runs): EVERY one of the 137 gradient tensors must be bit-equal."""
    cfg, params, model = full
    z, meta, b = _golden("g8_full_unsaturated")
    runs = []
    for _ in range(2):
        outs, tws, rows, loss, d = _run_hip(model, dev, b, meta["loss_kwargs"], meta["global_step"], 1)
        runs.append((outs, tws, rows, float(loss), {n_: p.grad.detach().clone() for n_, p in model.named_parameters()}))
    (o1, t1, r1, l1, g1), (o2, t2, r2, l2, g2) = runs
    for tag in ("query", "positive", "negative"):
        assert torch.equal(o1[tag], o2[tag]) and torch.equal(t1[tag], t2[tag]) and torch.equal(r1[tag], r2[tag]), tag
    assert l1 == l2
    spread = {n_: float((g1[n_] - g2[n_]).double().norm() / (g1[n_].double().norm() + 1e-30)) for n_ in g1}
    exact = [n_ for n_, v in spread.items() if v == 0.0]
    _report("two_run_determinism", {"max_rel_spread": max(spread.values()), "bitwise_equal_tensors": len(exact),
                                    "tensors": len(spread)})
    assert len(exact) == len(spread), sorted(spread.items(), key=lambda kv: -kv[1])[:5]
    model.zero_grad(set_to_none=True)


def test_full_size_parity_on_the_256_wide_kernels(dev):
    """The goldens' passes are shorter than the thresholds from which the two persistent kernels take over (8,192
    token rows for the weight-gradient GEMM, 2,048 for the decoder); their thresholds are read once per process, so a
    child process runs the unsaturated full-size parity test (loss terms, all 137 gradient tensors under pinned
    routing, ulp statement of the outputs) with both lowered: positive / negative passes of 1,024 rows then go
    through gemm_tn256.hip and decoder256.hip."""
    import subprocess
    import sys
    env = dict(os.environ, SNX_TN256_MIN_M="1024", SNX_DEC256_MIN_T="256")
    root = os.path.dirname(os.path.dirname(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity_full.py", "-q", "-x", "-m", "gpu", "-k",
                        "test_unsaturated_infonce_full_size"], cwd=root, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_bench_shape_micro_step_against_the_oracle(dev, full):
    """The shape the headline number is quoted on (BASELINE config 2: bs 64, q64 / d256, one fused pass of T = 36,864
    token rows -- the 256x256 NT kernel with its short tiles and column groups, the 256x256 dW kernel with its
    stream-K tail, the 256x192 decoder with atomicMax across 144 row tiles) had only `isfinite(loss)` as a
    model-level check (round-2 review, Weak #4).  A sequence's output does not depend on its batch mates, so four
    sampled sequences (a query, two positives, a negative) are checked against the oracle run on those four alone;
    and the whole step is repeated with the 256-wide NT kernel switched off: outputs AND all 137 gradient tensors
    must agree bit for bit (both kernels sum k in the same order), which pins the new kernel at the production shape."""
    from oracle import splade_oracle as O
    from snx._lib import fn
    from src.model.losses import SPLADELossV33
    cfg, params, model = full
    gen = torch.Generator().manual_seed(2026)
    B, Sq, Sd = 64, 64, 256

    def ids(S):
        x = torch.randint(6, cfg.pad_token_id, (B, S), generator=gen)
        x[:, 0] = 0
        x[:, -1] = 1
        return x
    q, p, n = ids(Sq), ids(Sd), ids(Sd)
    ones = lambda t: torch.ones_like(t)   # noqa: E731
    lf = SPLADELossV33(temperature=500.0).to(dev)

    def step():
        model.zero_grad(set_to_none=True)
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            (a, _), (pp, tw_p), (nn_, _) = model.forward_many([(q.to(dev), ones(q).to(dev)), (p.to(dev), ones(p).to(dev)),
                                                               (n.to(dev), ones(n).to(dev))])
            loss, _ = lf(anchor_repr=a, positive_repr=pp, negative_repr=nn_, global_step=50)
        loss.backward()
        torch.cuda.synchronize()
        return a, pp, nn_, tw_p, loss, {k: v.grad.detach().clone() for k, v in model.named_parameters()}

    fn("snx_nt256_configure")(1, 0)                       # default shape policy
    a, pp, nn_, tw_p, loss, grads = step()
    assert torch.isfinite(loss)
    rep = {}
    for tag, out, src, rows in (("query", a, q, [5]), ("positive", pp, p, [0, 63]), ("negative", nn_, n, [17])):
        with torch.no_grad():
            ref, ref_tw = O.splade_forward(params, cfg, src[rows], torch.ones_like(src[rows]), "bf16")
        st = sparse_ulp_stats(out[rows], ref)
        rep[tag] = st
        assert_ulp_statement(st, "bench shape " + tag)
        if tag == "positive":
            assert_ulp_statement(sparse_ulp_stats(tw_p[rows], ref_tw), "bench shape positive token_weights")
    _report("bench_shape_vs_oracle_bf16", rep)
    fn("snx_nt256_configure")(0, 0)                       # every NT GEMM on the 128x128 kernel
    try:
        a0, p0, n0, tw0, loss0, grads0 = step()
    finally:
        fn("snx_nt256_configure")(1, 0)
    assert torch.equal(a, a0) and torch.equal(pp, p0) and torch.equal(nn_, n0) and torch.equal(tw_p, tw0)
    assert float(loss) == float(loss0)
    # the dX GEMMs are bit-identical on either kernel and the weight gradients are reduced in a fixed order: same bits
    for k in grads:
        assert torch.equal(grads[k], grads0[k]), (k, float((grads[k] - grads0[k]).abs().max()))
    model.zero_grad(set_to_none=True)


def test_unchanged_three_call_loop_at_the_bench_shape_equals_the_fused_pass(dev, full):
    """The reference's literal micro-step (three model(...) calls, one loss, one backward: ref:train_v33_ddp.py:339-364) at
    the bench's shape, 149 M parameters: the first micro-step teaches the runtime the pattern, the second runs through the
    micro-step arena (snx.encoder.StepArena: every call fills its 4,096 / 16,384 rows of ONE 36,864-row arena, ONE deferred
    backward on the full-size launches).  Outputs, loss and all 137 gradient tensors of that second micro-step must equal the
    explicitly fused pass bit for bit."""
    from src.model.losses import SPLADELossV33
    cfg, params, model = full
    rt = model.runtime
    keep, rt.keep_last_ctx = rt.keep_last_ctx, False          # (the parity instrumentation pins per-pass arenas)
    gen = torch.Generator().manual_seed(2027)
    B, Sq, Sd = 64, 64, 256

    def ids(S):
        x = torch.randint(6, cfg.pad_token_id, (B, S), generator=gen)
        x[:, 0] = 0
        x[:, -1] = 1
        return x.to(dev)
    batches = [(ids(Sq), ids(Sd), ids(Sd)) for _ in range(2)]
    lf = SPLADELossV33(temperature=500.0).to(dev)
    try:
        def three_call(q, p, n):
            model.zero_grad(set_to_none=True)
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                a, _ = model(q, torch.ones_like(q))
                pp, _ = model(p, torch.ones_like(p))
                nn_, _ = model(n, torch.ones_like(n))
                loss, _ = lf(anchor_repr=a, positive_repr=pp, negative_repr=nn_, global_step=50)
            loss.backward()
            return a.detach(), pp.detach(), nn_.detach(), float(loss), {k: v.grad.detach().clone() for k, v in model.named_parameters()}

        rt._step_hist, rt._step_phase, rt._arena = [], "bwd", None
        three_call(*batches[0])                               # ordinary path: learns 36,864 rows, 192 sequences, 3 passes
        assert rt.step_arena_capacity() == (B * (Sq + 2 * Sd), 3 * B)
        a, pp, nn_, loss, grads = three_call(*batches[1])     # through the arena
        assert rt._step_hist[-1] == (B * (Sq + 2 * Sd), 3 * B, 3) and rt._arena is None
        model.zero_grad(set_to_none=True)
        q, p, n = batches[1]
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            (a2, _), (p2, _), (n2, _) = model.forward_many([(q, torch.ones_like(q)), (p, torch.ones_like(p)), (n, torch.ones_like(n))])
            loss2, _ = lf(anchor_repr=a2, positive_repr=p2, negative_repr=n2, global_step=50)
        loss2.backward()
        assert torch.equal(a, a2) and torch.equal(pp, p2) and torch.equal(nn_, n2) and loss == float(loss2)
        for k, v in model.named_parameters():
            assert torch.equal(v.grad, grads[k]), (k, float((v.grad - grads[k]).abs().max()))
    finally:
        rt.keep_last_ctx = keep
        rt._step_hist, rt._step_phase, rt._arena = [], "bwd", None
        model.zero_grad(set_to_none=True)
