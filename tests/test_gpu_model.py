"""End-to-end parity of the HIP path behind the reference API (SPLADEModernBERT + SPLADELossV33)
against the CPU oracle and the committed golden vectors.  Needs a real MI355X: pytest -m gpu.

Tolerance protocol:
  * loss kernels, fp32 operands ............ vs golden g4 (reference outputs): rel 2e-5
  * bf16 production path vs the oracle in emulated-bf16 mode (same cast points): sparse values as an
    ULP statement (tests/helpers.sparse_ulp_stats: no entry more than one bf16 ulp of the logit away,
    bounded fraction of one-ulp flips, mean <= 1e-3), top-k indices exact where the oracle's rank gap >
    2x the value error; loss terms rel 2e-3 under pinned routing; per-tensor gradient cosine >= 0.999
    and rel-L2 <= 2e-2 under pinned routing (free routing: at the oracle's own bf16-vs-fp32 floor,
    tests/test_oracle_floor.py)
  * bf16 production path vs the reference fp32 golden (g3, full size): reported, bound at the
    bf16-vs-fp32 background (torch's own CPU bf16 autocast vs fp32: max 6.9e-3 / mean 1.4e-3)."""
import json
import math
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


RESID_IN_LN_DEFAULT = 1          # (tests restore the process-wide switch they flip)


def _small_cfg():
    from oracle import splade_oracle as O
    return O.EncoderConfig(vocab_size=1000, hidden_size=256, intermediate_size=384, num_hidden_layers=4,
                           num_attention_heads=4, local_attention=16, pad_token_id=999)


def _build_model(cfg, params, dev):
    from src.model.splade_modern import SPLADEModernBERT
    geom = dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                global_attn_every_n_layers=cfg.global_attn_every_n_layers, local_attention=cfg.local_attention,
                global_rope_theta=cfg.global_rope_theta, local_rope_theta=cfg.local_rope_theta,
                norm_eps=cfg.norm_eps, pad_token_id=cfg.pad_token_id)
    m = SPLADEModernBERT(config=geom)
    sd = {k: v.clone() for k, v in params.items()}
    sd["model.decoder.weight"] = sd["model.model.embeddings.tok_embeddings.weight"]
    m.load_state_dict(sd, strict=True)
    return m.to(dev)


def test_state_dict_contract(dev):
    from oracle import splade_oracle as O
    cfg = _small_cfg()
    m = _build_model(cfg, O.init_params(cfg, 1), dev)
    keys = list(m.state_dict().keys())
    assert len(keys) == len(O.param_names(cfg)) + 1 and "model.decoder.weight" in keys
    assert [n for n, _ in m.named_parameters()] == O.param_names(cfg)
    assert m.state_dict()["model.decoder.weight"].data_ptr() == m.state_dict()["model.model.embeddings.tok_embeddings.weight"].data_ptr()
    assert m.vocab_size == 1000 and m.hidden_size == 256


def _grad_stats(got, ref):
    g, r = got.double().flatten(), ref.double().flatten()
    cos = float((g @ r) / (g.norm() * r.norm() + 1e-30))
    rel = float((g - r).norm() / (r.norm() + 1e-30))
    return cos, rel


@pytest.mark.parametrize("k,margin,Sq,Sd,cos_min,rel_max", [(1, 0.0, 8, 12, 0.99, 0.15), (1, 0.0, 40, 150, 0.98, 0.2),
                                                            (2, 0.05, 40, 150, 0.98, 0.2)])
def test_small_model_forward_backward_vs_oracle(dev, k, margin, Sq, Sd, cos_min, rel_max):
    """Gradient tolerance: the max-pool routes each (b, v) gradient to ONE sequence position; bf16
    logits tie or nearly tie often, so a 1-ulp accumulation-order flip re-routes that entry.  With
    free routing the oracle's OWN bf16-vs-fp32 gradients agree only to cos 0.989-0.994 / rel 0.11-0.15 on
    exactly these cases (tests/test_oracle_floor.py asserts that floor on the CPU), so they are held to
    cos >= 0.98 / rel <= 0.2 and the numbers are reported; the tight bound is enforced by the
    pinned-routing test below."""
    from oracle import splade_oracle as O
    from src.model.losses import SPLADELossV33
    cfg = _small_cfg()
    params = O.perturb_params(O.init_params(cfg, seed=3), seed=4, scale=2.0, bias_mean=-0.1)
    gen = torch.Generator().manual_seed(100 + k)
    b = O.synth_batch(6, Sq, Sd, cfg, gen, k=k, ragged=True, teacher=margin > 0)
    lc = O.LossConfig(lambda_q=0.01, lambda_d=0.003, temperature=20.0, flops_warmup_steps=50,
                      lambda_initial_ratio=0.1, lambda_margin_mse=margin)
    step = 20
    # oracle, emulated-bf16 mode
    leaves = {n: p.clone().requires_grad_(True) for n, p in params.items()}
    oq, oqt = O.splade_forward(leaves, cfg, b["query_input_ids"], b["query_attention_mask"], "bf16")
    op, opt = O.splade_forward(leaves, cfg, b["positive_input_ids"], b["positive_attention_mask"], "bf16")
    on, ont = O.splade_forward(leaves, cfg, b["negative_input_ids"], b["negative_attention_mask"], "bf16")
    on3 = on.view(6, k, -1) if k > 1 else on
    oloss, od = O.loss_v33(lc, oq, op, on3, step, b.get("teacher_pos_scores"), b.get("teacher_neg_scores"), "bf16")
    oloss.backward()
    # HIP path through the reference API
    model = _build_model(cfg, params, dev)
    loss_fn = SPLADELossV33(lambda_q=lc.lambda_q, lambda_d=lc.lambda_d, temperature=lc.temperature,
                            flops_warmup_steps=lc.flops_warmup_steps, lambda_initial_ratio=lc.lambda_initial_ratio,
                            lambda_margin_mse=margin).to(dev)
    D = lambda t: t.to(dev) if torch.is_tensor(t) else t   # noqa: E731
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        q, qt = model(D(b["query_input_ids"]), D(b["query_attention_mask"]))
        p, pt = model(D(b["positive_input_ids"]), D(b["positive_attention_mask"]))
        n, nt = model(D(b["negative_input_ids"]), D(b["negative_attention_mask"]))
        n3 = n.view(6, k, -1) if k > 1 else n
        loss, d = loss_fn(anchor_repr=q, positive_repr=p, negative_repr=n3, global_step=step,
                          teacher_pos_scores=D(b.get("teacher_pos_scores")),
                          teacher_neg_scores=D(b.get("teacher_neg_scores")))
    loss.backward()
    rep = {}
    for tag, got, ref in (("q", q, oq), ("p", p, op), ("n", n, on), ("qt", qt, oqt), ("pt", pt, opt)):
        rep[tag] = st = sparse_ulp_stats(got, ref)
        assert torch.isfinite(got).all()
        assert_ulp_statement(st, tag)
    # padded positions are exactly zero
    assert (pt.detach().cpu()[b["positive_attention_mask"] == 0] == 0).all()
    # top-k indices exact where the oracle gap allows
    tk = topk_rank_check(p, op, 32, max(rep["p"]["max_abs"], 1e-6))
    assert tk["equal"], tk
    rep["topk"] = tk
    rep["loss"] = {"got": float(loss), "ref": float(oloss)}
    assert float(loss) == pytest.approx(float(oloss), rel=5e-3)
    for key in ("infonce", "flops_q", "flops_d", "flops_neg", "margin_mse", "nonzero_q", "nonzero_d"):
        assert float(d[key]) == pytest.approx(od[key], rel=1e-2, abs=1e-4), key
    worst = (1.0, 0.0, "")
    for name, prm in model.named_parameters():
        cos, rel = _grad_stats(prm.grad.cpu(), leaves[name].grad)
        rep.setdefault("grads", {})[name] = [cos, rel]
        if cos < worst[0]:
            worst = (cos, rel, name)
    rep["worst_grad"] = worst
    _report(f"small_fwd_bwd_k{k}_q{Sq}_d{Sd}", rep)
    bad = {n_: v for n_, v in rep["grads"].items() if v[0] < cos_min or v[1] > rel_max}
    assert not bad, bad


# the last case is config 5's shape class: 4 negatives per query and documents beyond 256 tokens (streaming
# attention kernels, three decoder row chunks) with MarginMSE teacher scores
@pytest.mark.parametrize("k,margin,Sq,Sd", [(1, 0.0, 40, 150), (2, 0.05, 24, 70), (4, 0.05, 30, 300)])
def test_small_model_gradients_with_pinned_routing(dev, k, margin, Sq, Sd):
    """Backward parity at the survey's tight bound (cos >= 0.999, rel-L2 <= 2e-2 per tensor): the
    oracle (emulated bf16) back-propagates through the SAME max-pool routing the HIP forward chose
    (read back from the saved arg-max keys), which removes the only discontinuity of the path."""
    from oracle import splade_oracle as O
    from src.model.losses import SPLADELossV33
    cfg = _small_cfg()
    params = O.perturb_params(O.init_params(cfg, seed=3), seed=4, scale=2.0, bias_mean=-0.1)
    gen = torch.Generator().manual_seed(300 + k)
    b = O.synth_batch(6, Sq, Sd, cfg, gen, k=k, ragged=True, teacher=margin > 0)
    lc = O.LossConfig(lambda_q=0.01, lambda_d=0.003, temperature=20.0, flops_warmup_steps=50,
                      lambda_initial_ratio=0.1, lambda_margin_mse=margin)
    model = _build_model(cfg, params, dev)
    model.runtime.keep_last_ctx = True
    loss_fn = SPLADELossV33(lambda_q=lc.lambda_q, lambda_d=lc.lambda_d, temperature=lc.temperature,
                            flops_warmup_steps=lc.flops_warmup_steps, lambda_initial_ratio=lc.lambda_initial_ratio,
                            lambda_margin_mse=margin).to(dev)
    D = lambda t: t.to(dev) if torch.is_tensor(t) else t   # noqa: E731
    rows, outs = {}, {}
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        for tag in ("query", "positive", "negative"):
            outs[tag], _ = model(D(b[tag + "_input_ids"]), D(b[tag + "_attention_mask"]))
            rows[tag] = model.runtime.routing_rows(*model.runtime.last_ctx).cpu()
        n3 = outs["negative"].view(6, k, -1) if k > 1 else outs["negative"]
        loss, _ = loss_fn(anchor_repr=outs["query"], positive_repr=outs["positive"], negative_repr=n3, global_step=20,
                          teacher_pos_scores=D(b.get("teacher_pos_scores")), teacher_neg_scores=D(b.get("teacher_neg_scores")))
    loss.backward()
    leaves = {n: p.clone().requires_grad_(True) for n, p in params.items()}
    o = {t: O.splade_forward(leaves, cfg, b[t + "_input_ids"], b[t + "_attention_mask"], "bf16", route_rows=rows[t])[0]
         for t in ("query", "positive", "negative")}
    on3 = o["negative"].view(6, k, -1) if k > 1 else o["negative"]
    oloss, _ = O.loss_v33(lc, o["query"], o["positive"], on3, 20, b.get("teacher_pos_scores"), b.get("teacher_neg_scores"), "bf16")
    oloss.backward()
    assert float(loss) == pytest.approx(float(oloss), rel=2e-3)
    stats = {name: _grad_stats(prm.grad.cpu(), leaves[name].grad) for name, prm in model.named_parameters()}
    worst = min(stats.items(), key=lambda kv: kv[1][0])
    _report(f"small_pinned_routing_k{k}", {"worst": [worst[0], *worst[1]], "max_rel": max(v[1] for v in stats.values())})
    bad = {n_: v for n_, v in stats.items() if v[0] < 0.999 or v[1] > 2e-2}
    assert not bad, bad


def test_loss_kernels_vs_golden_g4(dev):
    """fp32-operand loss kernels vs the reference's own outputs (24 cases: B in {4,64}, k in
    {1,4,7}, warm-up steps, MarginMSE, lambda_neg fallback)."""
    from src.model.losses import SPLADELossV33
    z = np.load(os.path.join(G, "g4_loss_vectors.npz"))
    cases = json.load(open(os.path.join(G, "g4_loss_vectors.json")))
    for c in cases:
        pre = f"c{c['id']}::"
        a, p, n = (torch.from_numpy(z[pre + x]).to(dev).requires_grad_(True) for x in "apn")
        lf = SPLADELossV33(**c["loss_kwargs"]).to(dev)
        loss, d = lf(anchor_repr=a, positive_repr=p, negative_repr=n, global_step=c["step"],
                     teacher_pos_scores=torch.from_numpy(z[pre + "tp"]).to(dev),
                     teacher_neg_scores=torch.from_numpy(z[pre + "tn"]).to(dev))
        assert float(loss) == pytest.approx(c["loss"], rel=2e-5), c["id"]
        for key, v in c["loss_dict"].items():
            assert float(d[key]) == pytest.approx(v, rel=2e-5, abs=1e-6), (c["id"], key)
        loss.backward()
        for x, t in zip("apn", (a, p, n)):
            np.testing.assert_allclose(t.grad.cpu().numpy(), z[pre + "d" + x], atol=2e-6, rtol=5e-4)
        assert lf.get_avg_nonzero()[0] == pytest.approx(c["avg_nonzero"][0], rel=1e-5)


def test_loss_kd_branch_vs_golden_g9(dev):
    """KL-distillation branch of SPLADELossV33 (ref:src/model/losses.py:239-253; never fed by the V33 trainer):
    device loss kernels + the KD term vs the reference's own outputs, fp32 operands."""
    from src.model.losses import SPLADELossV33
    z = np.load(os.path.join(G, "g9_loss_kd.npz"))
    cases = json.load(open(os.path.join(G, "g9_loss_kd.json")))
    for c in cases:
        pre = f"c{c['id']}::"
        a, p, n = (torch.from_numpy(z[pre + x]).to(dev).requires_grad_(True) for x in "apn")
        lf = SPLADELossV33(**c["loss_kwargs"]).to(dev)
        loss, d = lf(anchor_repr=a, positive_repr=p, negative_repr=n, global_step=c["step"],
                     teacher_scores=torch.from_numpy(z[pre + "ts"]).to(dev),
                     teacher_pos_scores=torch.from_numpy(z[pre + "tp"]).to(dev),
                     teacher_neg_scores=torch.from_numpy(z[pre + "tn"]).to(dev))
        assert float(loss) == pytest.approx(c["loss"], rel=2e-5), c["id"]
        for key, v in c["loss_dict"].items():
            assert float(d[key]) == pytest.approx(v, rel=2e-5, abs=1e-6), (c["id"], key)
        loss.backward()
        for x, t in zip("apn", (a, p, n)):
            np.testing.assert_allclose(t.grad.cpu().numpy(), z[pre + "d" + x], atol=2e-6, rtol=5e-4)


def test_loss_cross_rank_identity(dev):
    """config-4 extension: anchors of `rank` against all-gathered positives == the oracle identity."""
    from oracle import splade_oracle as O
    from snx.loss import splade_loss
    g = torch.Generator().manual_seed(5)
    B, W, V, rank = 8, 4, 300, 2
    a = torch.relu(torch.rand(B, V, generator=g) - 0.5)
    pall = torch.relu(torch.rand(W * B, V, generator=g) - 0.5)
    n = torch.relu(torch.rand(B, V, generator=g) - 0.5)
    al, pl, nl = (t.clone().requires_grad_(True) for t in (a, pall, n))
    ref = O.cross_rank_infonce(al, pl, nl, rank, 2.0)
    ref.backward()
    ad, pd, nd = (t.to(dev).requires_grad_(True) for t in (a, pall, n))
    loss, sc = splade_loss(ad, pd, nd, (2.0, 0.0, 0.0, 0.0, 0.0), 1, label_off=rank * B)
    loss.backward()
    assert float(sc[1]) == pytest.approx(float(ref), rel=1e-5)
    for got, r in ((ad, al), (pd, pl), (nd, nl)):
        np.testing.assert_allclose(got.grad.cpu().numpy(), r.grad.numpy(), atol=1e-7, rtol=1e-4)


@pytest.fixture(scope="module")
def full_setup(dev):
    from oracle import splade_oracle as O
    cfg = O.EncoderConfig()
    params = O.perturb_params(O.init_params(cfg, seed=42), seed=7, bias_mean=-0.2)
    return cfg, params, _build_model(cfg, params, dev)


def test_full_size_vs_reference_golden_g3(dev, full_setup):
    """149M model, B=4, q64/d256 ragged: HIP bf16 path vs the REFERENCE's fp32 outputs (g3)."""
    from src.model.losses import SPLADELossV33
    cfg, params, model = full_setup
    z = np.load(os.path.join(G, "g3_full_fwd_bwd.npz"))
    meta = json.load(open(os.path.join(G, "g3_full_fwd_bwd.json")))
    b = {k[4:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith("in::")}
    rep = {}
    outs = {}
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        for tag, pre in (("q", "query"), ("p", "positive"), ("n", "negative")):
            sr, tw = model(b[pre + "_input_ids"], b[pre + "_attention_mask"])
            outs[tag] = sr
            ref_full = torch.from_numpy(z[f"out::{tag}_full"].astype(np.float32))
            diff = (sr.detach().cpu() - ref_full).abs()
            tv, ti = torch.topk(sr.detach().cpu(), 256, dim=-1)
            rv, ri = torch.from_numpy(z[f"out::{tag}_topv"]), torch.from_numpy(z[f"out::{tag}_topi"])
            overlap = np.mean([len(set(ti[r, :50].tolist()) & set(ri[r, :50].tolist())) / 50 for r in range(ti.shape[0])])
            tdiff = (tv - rv).abs()
            twd = (tw.detach().cpu().reshape(-1) - torch.from_numpy(z[f"out::{tag}_tw"]).reshape(-1)).abs()
            rep[tag] = {"full_max": float(diff.max()), "full_mean": float(diff.mean()), "top256_val_max": float(tdiff.max()),
                        "top50_overlap": float(overlap), "tw_max": float(twd.max()), "scale": float(rv.max())}
            assert torch.isfinite(sr).all()
            assert diff.max().item() < 1.5e-2 and diff.mean().item() < 2e-3, rep[tag]
            assert overlap > 0.85, rep[tag]
        lf = SPLADELossV33(**meta["loss_kwargs"]).to(dev)
        loss, d = lf(anchor_repr=outs["q"], positive_repr=outs["p"], negative_repr=outs["n"],
                     global_step=meta["global_step"])
    rep["loss"] = {"got": float(loss), "ref": meta["loss"], "infonce": float(d["infonce"]),
                   "ref_infonce": meta["loss_dict"]["infonce"]}
    for key in ("flops_q", "flops_d", "flops_neg"):
        assert float(d[key]) == pytest.approx(meta["loss_dict"][key], rel=1e-2), key
    loss.backward()
    norms = dict(zip(meta["grad_names"], meta["grad_norms"]))
    gn = {}
    for name, prm in model.named_parameters():
        gn[name] = float(prm.grad.double().norm())
    rep["grad_norm_ratio_minmax"] = [min(gn[n] / max(norms[n], 1e-30) for n in gn),
                                     max(gn[n] / max(norms[n], 1e-30) for n in gn)]
    pg = dict(model.named_parameters())
    probe = {}
    for key in z.files:
        if key.startswith("gprobe::model"):
            g = pg[key[8:]].grad
            got = (g[:8, :64] if g.dim() == 2 else g[:512]).cpu()
            probe[key[8:]] = _grad_stats(got, torch.from_numpy(z[key]))
    rep["grad_probe_cos_rel"] = probe
    _report("full_vs_reference_fp32_g3", rep)
    for n_, (cos, rel) in probe.items():   # bf16 path vs fp32 reference, saturated InfoNCE: noise floor ~0.98
        assert cos > 0.95, (n_, cos, rel)
    model.zero_grad(set_to_none=True)


def test_other_geometry_wider_model(dev):
    """Nothing in the kernels is specialised to the 149 M model's widths beyond head_dim 64, hidden % 256 and
    intermediate % 128 (`EncoderGeometry.bf16_unsupported_reason`; ModernBERT-large's 2,624 is not, its fp32 path runs):
    a wider model -- hidden 1,024, 16 heads, intermediate 2,560, local window 128; three layers and a small vocabulary
    here to keep the CPU oracle quick -- through the bf16 kernels against the oracle's emulated-bf16 forward,
    the fp32 path against the oracle's fp32 forward, and a backward through both that leaves finite gradients of the
    right scale."""
    from oracle import splade_oracle as O
    from tests.helpers import sparse_ulp_stats
    cfg = O.EncoderConfig(vocab_size=2000, hidden_size=1024, intermediate_size=2560, num_hidden_layers=3,
                          num_attention_heads=16, local_attention=128, pad_token_id=1999)
    params = O.perturb_params(O.init_params(cfg, seed=21), seed=22, scale=2.0, bias_mean=-0.1)
    gen = torch.Generator().manual_seed(5)
    B, S = 3, 200
    ids = torch.randint(5, 1990, (B, S), generator=gen)
    mask = torch.ones(B, S, dtype=torch.int64)
    mask[1, 150:] = 0
    mask[2, 77:] = 0
    ids[mask == 0] = cfg.pad_token_id
    model = _build_model(cfg, params, dev)
    want16, _ = O.splade_forward(params, cfg, ids, mask, "bf16")
    want32, _ = O.splade_forward(params, cfg, ids, mask, "fp32")
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        got16, _ = model(ids.to(dev), mask.to(dev))
    # (1,024 inputs per decoder row and weights perturbed to twice the spread: the logit noise of section 2's model is
    # ~1.5x that of the 149 M fixtures, hence the wider 6-sigma floor and looser shares than assert_ulp_statement's)
    st = sparse_ulp_stats(got16, want16, logit_abs_tol=2.5e-2)
    assert st["far"] == 0 and st["ulp0"] >= 0.5 and st["ulp2"] + st["floor"] <= 0.06, st
    assert st["mean_abs"] <= 3e-3 and st["max_abs"] <= 2e-2 and abs(st["bias"]) <= 4e-4, st
    got32, _ = model(ids.to(dev), mask.to(dev))                  # outside autocast: the fp32 path
    err32 = float((got32.detach().cpu() - want32).abs().max())
    assert err32 <= 2e-5 * max(1.0, float(want32.abs().max())), err32
    _report("wider_model_h1024_i2560", {"bf16": st, "fp32_max_abs": err32})
    for out in (got16, got32):
        model.zero_grad(set_to_none=True)
        out.square().sum().backward()
        gn = {n: float(p.grad.float().norm()) for n, p in model.named_parameters()}
        assert all(v == v and v < 1e12 for v in gn.values()), gn
        assert gn["model.model.layers.0.attn.Wqkv.weight"] > 0 and gn["model.model.embeddings.tok_embeddings.weight"] > 0


def test_weight_cache_batched_refresh_equals_per_tensor(dev, full_setup, monkeypatch):
    """The bf16 weight cache (natural + transposed copies, Wi in the GeGLU row order) is rebuilt behind every optimizer
    step: one launch per shape class with the layer in blockIdx.z (7 launches) must write the bytes of the 157
    per-tensor launches it replaces (SNX_WCACHE_PER_TENSOR=1)."""
    import ctypes as C
    from snx._lib import check, fn
    cfg, params, model = full_setup
    rt = model.runtime
    ptrs = rt._param_ptrs()
    nbytes = fn("snx_weight_cache_bytes")(C.byref(rt._desc))
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    a = torch.full((nbytes,), 0x5A, dtype=torch.uint8, device=dev)
    b = torch.full((nbytes,), 0x5A, dtype=torch.uint8, device=dev)
    check(fn("snx_weight_cache_refresh")(C.byref(rt._desc), ptrs, C.c_void_p(a.data_ptr()), st), "batched")
    import snx
    snx.configure(wcache_per_tensor=1)
    try:
        check(fn("snx_weight_cache_refresh")(C.byref(rt._desc), ptrs, C.c_void_p(b.data_ptr()), st), "per tensor")
    finally:
        snx.configure(wcache_per_tensor=0)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def test_inference_matches_training_forward(dev, full_setup):
    from oracle import splade_oracle as O
    cfg, params, model = full_setup
    gen = torch.Generator().manual_seed(9)
    ids, mask = O.synth_ids(3, 64, cfg, gen, ragged=True)
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):      # the bf16 kernels (training precision)
        with torch.no_grad():
            a, _ = model(ids.to(dev), mask.to(dev))
        b, _ = model(ids.to(dev), mask.to(dev))
        assert torch.equal(a, b.detach())
        assert torch.equal(model.encode(ids.to(dev), mask.to(dev)).detach(), a)


def test_forward_many_equals_separate_passes(dev):
    """q/p/n batches in ONE native pass (sequence groups) give the same outputs and gradients."""
    from oracle import splade_oracle as O
    cfg = _small_cfg()
    params = O.perturb_params(O.init_params(cfg, seed=3), seed=4, scale=2.0, bias_mean=-0.1)
    gen = torch.Generator().manual_seed(55)
    b = O.synth_batch(5, 40, 150, cfg, gen, k=2, ragged=True)
    pairs = [(b[t + "_input_ids"].to(dev), b[t + "_attention_mask"].to(dev)) for t in ("query", "positive", "negative")]
    m1, m2 = _build_model(cfg, params, dev), _build_model(cfg, params, dev)
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        sep = [m1(i, m) for i, m in pairs]
        many = m2.forward_many(pairs)
    w = [torch.randn_like(s[0]) for s in sep]
    sum((s[0] * wi).sum() for s, wi in zip(sep, w)).backward()
    sum((s[0] * wi).sum() for s, wi in zip(many, w)).backward()
    for (s1, t1), (s2, t2) in zip(sep, many):
        assert torch.equal(s1, s2) and torch.equal(t1, t2)
    for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        cos, rel = _grad_stats(p2.grad, p1.grad)
        assert cos > 0.99999 and rel < 2e-3, (n1, cos, rel)      # three launches + accumulate vs one: another fp32 summation tree


def _three_call_step(model, pairs, weights, used=(0, 1, 2)):
    """The reference's call pattern (ref:train_v33_ddp.py:339-343,364): three model(...) calls, one backward."""
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        outs = [model(i, m) for i, m in pairs]
    sum((outs[j][0] * weights[j]).sum() for j in used).backward()
    return [(o[0].detach().clone(), o[1].detach().clone()) for o in outs]


def test_micro_step_arena_equals_the_fused_pass_bit_for_bit(dev):
    """snx.encoder.StepArena: from the second micro-step on the unchanged three-call loop fills ONE arena (a pass = the
    same kernels on a row range: snx_model_forward_range) and its three autograd nodes share ONE deferred native backward.
    Outputs must equal the stand-alone passes bit for bit; the accumulated gradients must equal those of the explicitly
    fused pass (forward_many) bit for bit -- same arena contents, same launches, ordered reductions; against three
    independent backwards only the fp32 summation tree differs (gross-error screen).  Then: a first pass of another length
    (dynamic padding), a pass that no longer fits the capacity (prefix backward over the passes placed so far + ordinary
    path) and an output left out of the loss (flat-gradient
    mode: the engine's end-of-backward callback back-propagates the rest)."""
    from oracle import splade_oracle as O
    cfg = _small_cfg()
    params = O.perturb_params(O.init_params(cfg, seed=3), seed=4, scale=2.0, bias_mean=-0.1)
    gen = torch.Generator().manual_seed(57)
    steps = []
    for _ in range(4):
        b = O.synth_batch(6, 24, 150, cfg, gen, k=1, ragged=True)
        steps.append([(b[t + "_input_ids"].to(dev), b[t + "_attention_mask"].to(dev)) for t in ("query", "positive", "negative")])
    w = [torch.randn(6, cfg.vocab_size, generator=torch.Generator().manual_seed(i), device="cpu").to(dev) for i in range(3)]
    m_sep, m_arena, m_fused = (_build_model(cfg, params, dev) for _ in range(3))
    m_sep.runtime.step_arena_on = False
    assert m_arena.runtime.step_arena_on
    for it, pairs in enumerate(steps[:3]):
        o_sep = _three_call_step(m_sep, pairs, w)
        o_arena = _three_call_step(m_arena, pairs, w)
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            many = m_fused.forward_many(pairs)
        sum((many[j][0] * w[j]).sum() for j in range(3)).backward()
        assert m_arena.runtime.step_arena_capacity() == (6 * (24 + 150 + 150), 18)
        for (s1, t1), (s2, t2), (s3, t3) in zip(o_sep, o_arena, many):
            assert torch.equal(s1, s2) and torch.equal(t1, t2) and torch.equal(s2, s3) and torch.equal(t2, t3), it
    # micro-step 0 ran on the ordinary path in both; 1 and 2 went through the arena: accumulated gradients
    g_sep = {n: p.grad.clone() for n, p in m_sep.named_parameters()}
    g_arena = {n: p.grad.clone() for n, p in m_arena.named_parameters()}
    g_fused = {n: p.grad.clone() for n, p in m_fused.named_parameters()}
    m_chk = _build_model(cfg, params, dev)                   # micro-step 0 as three passes, 1 and 2 fused: the arena's sum
    m_chk.runtime.step_arena_on = False
    _three_call_step(m_chk, steps[0], w)
    for pairs in steps[1:3]:
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            many = m_chk.forward_many(pairs)
        sum((many[j][0] * w[j]).sum() for j in range(3)).backward()
    for n, p in m_chk.named_parameters():
        assert torch.equal(p.grad, g_arena[n]), (n, float((p.grad - g_arena[n]).abs().max()))
    for n in g_sep:
        cos, rel = _grad_stats(g_arena[n], g_sep[n])
        assert cos > 0.99999 and rel < 2e-3, (n, cos, rel)
        cos, rel = _grad_stats(g_arena[n], g_fused[n])
        assert cos > 0.99999 and rel < 2e-3, (n, cos, rel)
    # ---- the first pass of a micro-step may change its length (dynamic padding of the queries): still one arena
    for m in (m_arena, m_chk):
        m.zero_grad(set_to_none=True)
    short_q = (steps[3][0][0][:, :16].contiguous(), steps[3][0][1][:, :16].contiguous())
    varied = [short_q, steps[3][1], steps[3][2]]
    _three_call_step(m_arena, varied, w)
    assert m_arena.runtime._step_hist[-1] == (6 * (16 + 150 + 150), 18, 3)          # all three went through the arena
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        many = m_chk.forward_many(varied)
    sum((many[j][0] * w[j]).sum() for j in range(3)).backward()
    for (n, p1), (_, p2) in zip(m_chk.named_parameters(), m_arena.named_parameters()):
        assert torch.equal(p1.grad, p2.grad), n
    _three_call_step(m_arena, steps[2], w)                   # ... and back to the long queries: the capacity is the maximum
    assert m_arena.runtime.step_arena_capacity() == (6 * (24 + 150 + 150), 18)
    # ---- a pass that does not fit what is left of the capacity: (q, p) placed, then twice the rows -> the arena back-propagates
    # its two passes as a prefix, the third runs on the ordinary path
    for m in (m_sep, m_arena):
        m.zero_grad(set_to_none=True)
    big = (torch.cat([steps[3][2][0]] * 2, 1).contiguous(), torch.cat([steps[3][2][1]] * 2, 1).contiguous())   # [6, 300]
    broken = [steps[3][0], steps[3][1], big]
    o_sep = _three_call_step(m_sep, broken, w)
    o_arena = _three_call_step(m_arena, broken, w)
    assert m_arena.runtime._step_hist[-1] == (6 * (24 + 150 + 300), 18, 3)            # ... and the capacity has grown
    for (s1, t1), (s2, t2) in zip(o_sep, o_arena):
        assert torch.equal(s1, s2) and torch.equal(t1, t2)
    for (n, p1), (_, p2) in zip(m_sep.named_parameters(), m_arena.named_parameters()):
        cos, rel = _grad_stats(p2.grad, p1.grad)
        assert cos > 0.99999 and rel < 2e-3, (n, cos, rel)
    # ---- an output left out of the loss, in the flat-gradient mode (NativeDataParallel's)
    from src.train.core import ddp_trainer as T
    w_sep, w_arena = T.NativeDataParallel(_build_model(cfg, params, dev)), T.NativeDataParallel(_build_model(cfg, params, dev))
    w_sep.module.runtime.step_arena_on = False
    for pairs, used in ((steps[0], (0, 1, 2)), (steps[1], (0, 2)), (steps[2], (0, 1, 2))):
        _three_call_step(w_sep, pairs, w, used)
        _three_call_step(w_arena, pairs, w, used)
    fa, fb = w_sep.module.runtime.flat_grad, w_arena.module.runtime.flat_grad
    assert float((fa - fb).double().norm()) <= 2e-3 * float(fa.double().norm())


def test_micro_step_arena_above_4096_rows_with_fewer_rows_than_capacity(dev):
    """the LayerNorm backward's partial-row workspace is sized by the PLAN's row count while a
    dynamically padded micro-step holds fewer rows; the block count cdiv(T, rows_per_block(T)) is not monotone in T
    (8,192 rows: 1,024 blocks, 9,000 rows: 750), so a 9,000-row capacity followed by an 8,192-row micro-step returned
    SNX_E_ARG.  The workspace size is a monotone bound now: the deferred backward over T < T_plan rows runs and equals the
    fused pass of the same rows bit for bit."""
    from oracle import splade_oracle as O
    cfg = _small_cfg()
    params = O.perturb_params(O.init_params(cfg, seed=3), seed=4, scale=2.0, bias_mean=-0.1)
    gen = torch.Generator().manual_seed(91)

    def step(B, Sq, Sd):
        b = O.synth_batch(B, Sq, Sd, cfg, gen, k=1, ragged=True)
        out = []
        for t, S in (("query", Sq), ("positive", Sd), ("negative", Sd)):
            ids, mask = b[t + "_input_ids"], b[t + "_attention_mask"]
            pad = S - ids.shape[1]                            # the collator pads to the longest item: force the full width
            if pad:
                ids = torch.cat([ids, torch.full((B, pad), cfg.pad_token_id, dtype=ids.dtype)], 1)
                mask = torch.cat([mask, torch.zeros((B, pad), dtype=mask.dtype)], 1)
            out.append((ids.to(dev), mask.to(dev)))
        return out
    big, small = step(20, 50, 200), step(20, 50, 180)          # 9,000 rows, then 8,200 (1,024 -> 1,025 blocks of 8 rows)
    small2 = step(16, 64, 224)                                 # 8,192 rows exactly (16 sequences of the 20 planned)
    assert sum(i.numel() for i, _ in big) == 9000 and sum(i.numel() for i, _ in small2) == 8192
    w = [torch.randn(20, cfg.vocab_size, generator=torch.Generator().manual_seed(i), device="cpu").to(dev) for i in range(3)]
    m_arena, m_chk = _build_model(cfg, params, dev), _build_model(cfg, params, dev)
    m_chk.runtime.step_arena_on = False
    _three_call_step(m_arena, big, w)                          # ordinary path: teaches the capacity (9,000 rows, 60 sequences)
    _three_call_step(m_chk, big, w)
    assert m_arena.runtime.step_arena_capacity() == (9000, 60)
    for pairs in (small, small2, big):
        ww = [x[:pairs[0][0].shape[0]] for x in w]
        _three_call_step(m_arena, pairs, ww)
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            many = m_chk.forward_many(pairs)
        sum((many[j][0] * ww[j]).sum() for j in range(3)).backward()
    assert m_arena.runtime.arena_stats["placed"] == 9 and m_arena.runtime.arena_stats["fell_back"] == 0
    for (n, p1), (_, p2) in zip(m_chk.named_parameters(), m_arena.named_parameters()):
        assert torch.equal(p1.grad, p2.grad), (n, float((p1.grad - p2.grad).abs().max()))


def test_micro_step_arena_survives_a_forward_without_a_backward_and_counts_what_it_did(dev, caplog):
    """ADVICE round 5 (medium) + VERDICT round 5 item 9.  (a) A grad-enabled forward whose output is dropped (logging,
    evaluation, an exception before loss.backward()) used to leave its arena open: the next micro-step's passes were placed
    behind it and `_engine_done` raised inside a valid backward.  The arena keeps weak references to its autograd nodes
    and is abandoned at the next placement.  (b) The same after an optimizer step between the passes.  (c) Every way a
    pass leaves the arena is counted in EncoderRuntime.arena_stats and warned about once."""
    import logging
    from oracle import splade_oracle as O
    cfg = _small_cfg()
    params = O.perturb_params(O.init_params(cfg, seed=3), seed=4, scale=2.0, bias_mean=-0.1)
    gen = torch.Generator().manual_seed(58)
    steps = []
    for _ in range(3):
        b = O.synth_batch(6, 24, 150, cfg, gen, k=1, ragged=False)
        steps.append([(b[t + "_input_ids"].to(dev), b[t + "_attention_mask"].to(dev)) for t in ("query", "positive", "negative")])
    w = [torch.randn(6, cfg.vocab_size, generator=torch.Generator().manual_seed(i), device="cpu").to(dev) for i in range(3)]
    m, m_chk = _build_model(cfg, params, dev), _build_model(cfg, params, dev)
    m_chk.runtime.step_arena_on = False
    rt = m.runtime
    _three_call_step(m, steps[0], w)                           # teaches the capacity
    _three_call_step(m_chk, steps[0], w)
    assert rt.arena_stats == {"placed": 0, "fell_back": 0, "zero_filled": 0, "stale_dropped": 0}
    with caplog.at_level(logging.WARNING, logger="snx.encoder"):
        # (a) a logging forward: placed, then dropped without a backward
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            probe = m(*steps[1][0])
        assert rt._arena is not None and rt._arena.placed == 1
        probe_val = float(probe[0].sum())                      # used, never back-propagated
        del probe
        _three_call_step(m, steps[1], w)                       # a normal micro-step right behind it: must not raise
        _three_call_step(m_chk, steps[1], w)
        assert rt.arena_stats["stale_dropped"] == 1 and rt._arena is None
        assert rt._step_hist[-1] == (6 * (24 + 150 + 150), 18, 3)   # the stray pass did not inflate the micro-step's totals
        # (b) parameters change between two passes (a custom loop stepping the optimizer early): no pass joins the old arena
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            held = m(*steps[2][0])
        with torch.no_grad():
            for p_ in m.parameters():
                p_.add_(0.0)                                    # bumps the version counters, same values
        _three_call_step(m, steps[2], w)
        _three_call_step(m_chk, steps[2], w)
        assert rt.arena_stats["stale_dropped"] == 2
        del held
        # (c) a pass that does not fit
        big = (torch.cat([steps[2][2][0]] * 3, 1).contiguous(), torch.cat([steps[2][2][1]] * 3, 1).contiguous())
        _three_call_step(m, [steps[2][0], steps[2][1], big], w)
        _three_call_step(m_chk, [steps[2][0], steps[2][1], big], w)
        assert rt.arena_stats["fell_back"] == 1
    assert math.isfinite(probe_val)
    msgs = [r.getMessage() for r in caplog.records if "micro-step arena" in r.getMessage()]
    assert sum("abandoned" in x for x in msgs) == 1 and sum("ordinary path" in x for x in msgs) == 1, msgs
    for (n, p1), (_, p2) in zip(m_chk.named_parameters(), m.named_parameters()):
        cos, rel = _grad_stats(p2.grad, p1.grad)
        assert cos > 0.99999 and rel < 2e-3, (n, cos, rel)
    # zero fill (flat-gradient mode): counted
    from src.train.core import ddp_trainer as T
    wdp = T.NativeDataParallel(_build_model(cfg, params, dev))
    _three_call_step(wdp, steps[0], w)
    _three_call_step(wdp, steps[1], w, (0, 2))
    assert wdp.module.runtime.arena_stats["zero_filled"] == 1
    # an over-long input is refused by the ordinary path's check even when an arena is active
    too_long = (torch.zeros((1, 8193), dtype=torch.int64, device=dev), torch.ones((1, 8193), dtype=torch.int64, device=dev))
    with pytest.raises(ValueError, match="sequence too long"):
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            m(*too_long)


def test_residual_add_in_the_layernorm_equals_the_gemm_epilogue(dev):
    """snx_configure "resid_in_ln": the Wo GEMMs store bf16 and the following LayerNorm adds it to the fp32 stream
    (snx_ln_fwd_add) instead of the GEMM's residual epilogue -- h + float(bf16(A W^T)) either way: identical outputs, and
    BIT-identical gradients (the backward reads the same saved h; the weight gradients are reduced in a fixed order)."""
    import snx
    from oracle import splade_oracle as O
    cfg = _small_cfg()
    params = O.perturb_params(O.init_params(cfg, seed=3), seed=4, scale=2.0, bias_mean=-0.1)
    gen = torch.Generator().manual_seed(56)
    ids, mask = O.synth_ids(6, 130, cfg, gen, ragged=True)
    outs = []
    try:
        for mode in (0, 1):
            snx.configure(resid_in_ln=mode)
            m = _build_model(cfg, params, dev)
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                s_, t_ = m(ids.to(dev), mask.to(dev))
            (s_ * torch.linspace(0.5, 1.5, s_.shape[-1], device=dev)).sum().backward()
            outs.append((s_.detach(), t_.detach(), {n: p.grad.clone() for n, p in m.named_parameters()}))
    finally:
        snx.configure(resid_in_ln=RESID_IN_LN_DEFAULT)
    (s0, t0, g0), (s1, t1, g1) = outs
    assert torch.equal(s0, s1) and torch.equal(t0, t1)
    for n in g0:                                             # same saved h, same kernels, ordered reductions: same bits
        assert torch.equal(g0[n], g1[n]), (n, _grad_stats(g1[n], g0[n]))


def test_fused_adamw_matches_torch_adamw(dev):
    """snx_adamw_clip_step == clip_grad_norm_ + torch.optim.AdamW (wd grouping quirk, bias
    correction, clipping active and inactive), and the state dicts are interchangeable."""
    from oracle import splade_oracle as O
    from snx.optim import FusedAdamW
    from src.train.config.v33 import V33Config
    from src.train.core import ddp_trainer as T
    cfg = _small_cfg()
    params = O.init_params(cfg, seed=11)
    m1, m2 = _build_model(cfg, params, dev), _build_model(cfg, params, dev)
    conf = V33Config()
    conf.training.learning_rate = 1e-2
    w2 = T.NativeDataParallel(m2)
    o1, o2 = T.build_optimizer(m1, conf), T.build_optimizer(w2, conf)
    assert isinstance(o2, FusedAdamW) and not isinstance(o1, FusedAdamW)
    s1, s2 = T.build_scheduler(o1, 2, 10), T.build_scheduler(o2, 2, 10)
    gen = torch.Generator().manual_seed(0)
    for it in range(4):
        scale = 10.0 if it % 2 == 0 else 1e-3          # clipping on / off
        for p1, p2 in zip(m1.parameters(), m2.parameters()):
            g = (torch.randn(p1.shape, generator=gen) * scale).to(dev)
            p1.grad = g.clone()
            p2.grad.copy_(g)
        n1 = torch.nn.utils.clip_grad_norm_(m1.parameters(), 1.0)
        o1.step(); s1.step()
        T.optimizer_step(w2, o2, s2, conf)
        assert float(o2.grad_norm) == pytest.approx(float(n1), rel=1e-5)
        for (n_, p1), p2 in zip(m1.named_parameters(), m2.parameters()):
            assert torch.allclose(p1, p2, rtol=2e-5, atol=2e-7), (it, n_)
        assert float(m2.runtime.flat_grad.abs().max()) == 0.0
    sd1, sd2 = o1.state_dict(), o2.state_dict()
    assert sd1["param_groups"][1]["weight_decay"] == sd2["param_groups"][1]["weight_decay"] == 0.0
    for k in sd1["state"]:
        assert float(sd1["state"][k]["step"]) == float(sd2["state"][k]["step"]) == 4.0
        assert torch.allclose(sd1["state"][k]["exp_avg_sq"], sd2["state"][k]["exp_avg_sq"], rtol=1e-4, atol=1e-12)
    o2.load_state_dict(sd1)                              # torch AdamW state loads into the fused optimizer
    assert o2._steps == 4 and torch.allclose(o2.state[m2.model.head.dense.weight]["exp_avg"],
                                             o1.state[m1.model.head.dense.weight]["exp_avg"])
    # the forward sees the updated weights (bf16 cache refreshed after the fused step)
    ids, mask = O.synth_ids(2, 16, cfg, torch.Generator().manual_seed(1), ragged=True)
    with torch.no_grad():
        a, _ = m1(ids.to(dev), mask.to(dev))
        b, _ = m2(ids.to(dev), mask.to(dev))
    assert (a - b).abs().max().item() < 2e-2


def test_unpadded_execution_equals_padded(dev):
    """varlen path: only valid tokens are computed; outputs identical, gradients equal up to fp32 summation order
    (other token rows per workgroup: another summation tree); token_weights come back in the padded layout with zeros at padding."""
    from oracle import splade_oracle as O
    cfg = _small_cfg()
    params = O.perturb_params(O.init_params(cfg, seed=3), seed=4, scale=2.0, bias_mean=-0.1)
    gen = torch.Generator().manual_seed(66)
    b = O.synth_batch(5, 40, 150, cfg, gen, k=1, ragged=True)
    tags = ("query", "positive", "negative")
    pairs = [(b[t + "_input_ids"].to(dev), b[t + "_attention_mask"].to(dev)) for t in tags]
    lengths = [b[t + "_attention_mask"].sum(1) for t in tags]
    m1, m2 = _build_model(cfg, params, dev), _build_model(cfg, params, dev)
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        pad = m1.forward_many(pairs)
        pk = m2.forward_many(pairs, lengths)
    w = [torch.randn_like(s[0]) for s in pad]
    sum((s[0] * wi).sum() for s, wi in zip(pad, w)).backward()
    sum((s[0] * wi).sum() for s, wi in zip(pk, w)).backward()
    for (s1, t1), (s2, t2) in zip(pad, pk):
        assert torch.equal(s1, s2) and torch.equal(t1, t2)
    for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        cos, rel = _grad_stats(p2.grad, p1.grad)
        assert cos > 0.99999 and rel < 2e-3, (n1, cos, rel)
    with pytest.raises(ValueError), torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        m2.forward_many(pairs, [lengths[0] * 0, lengths[1], lengths[2]])


def train_epoch_case(dev, mode):
    """rows a14/a15: the trainer's micro-batch loop (accumulate 4, clip, AdamW with the wd-grouping quirk, cosine
    warm-up LR, lambda schedule on global_step; ref:train_v33_ddp.py:289-448) on the HIP path vs the oracle's
    restated loop in emulated-bf16 mode: 8 micro-steps = 2 optimizer steps.  The oracle back-propagates through
    the max-pool routing each HIP forward chose, so the comparison is tight: per-micro-step losses, the
    ACCUMULATED gradients handed to each optimizer step, and the PARAMETER VALUES after the two steps.
    mode: "native" (NativeDataParallel + fused AdamW), "plain" (bare module + torch AdamW),
    "ddp" (torch DistributedDataParallel around the module: the literal drop-in of ref:train_v33_ddp.py:539-544;
    needs an initialised process group)."""
    from torch.utils.data import DataLoader, Dataset
    from oracle import splade_oracle as O
    from src.model.losses import SPLADELossV33
    from src.train.config.v33 import V33Config
    from src.train.core import ddp_trainer as T

    cfg = _small_cfg()
    params = O.perturb_params(O.init_params(cfg, seed=3), seed=4, scale=2.0, bias_mean=-0.1)
    gen = torch.Generator().manual_seed(2024)
    batches = [O.synth_batch(4, 24, 70, cfg, gen, k=1, ragged=True) for _ in range(8)]
    conf = V33Config()
    conf.training.gradient_accumulation_steps = 4
    lr0 = 5e-4
    conf.training.learning_rate = lr0
    conf.training.log_every_n_steps = 1
    conf.loss.flops_warmup_steps = 4
    conf.loss.temperature = 20.0
    lc = O.LossConfig(lambda_q=conf.loss.lambda_q, lambda_d=conf.loss.lambda_d, temperature=20.0,
                      flops_warmup_steps=4, lambda_initial_ratio=conf.loss.lambda_initial_ratio)

    class DS(Dataset):
        def __len__(self):
            return len(batches)

        def __getitem__(self, i):
            return batches[i]
    dl = DataLoader(DS(), batch_size=None, shuffle=False)
    inner = _build_model(cfg, params, dev)
    rt = inner.runtime
    captured = []
    orig = rt.forward_many_impl

    def capturing(pairs, save, lengths=None):           # max-pool routing of every native forward
        out = orig(pairs, save, lengths)
        if save:
            captured.append(rt.routing_rows(out[2], out[3]).cpu())
        return out
    rt.forward_many_impl = capturing
    place = rt.step_arena_place

    def capturing_place(ids, mask):                     # ... and of every pass placed into a micro-step arena
        out = place(ids, mask)
        if out is not None:
            captured.append(out[2].routing_rows(out[3]).cpu())
        return out
    rt.step_arena_place = capturing_place
    if mode == "native":
        model = T.NativeDataParallel(inner)
    elif mode == "ddp":
        model = T.DDP(inner, device_ids=[dev.index or 0], broadcast_buffers=False, find_unused_parameters=False)
    else:
        model = inner
    loss_fn = SPLADELossV33(lambda_q=conf.loss.lambda_q, lambda_d=conf.loss.lambda_d, temperature=20.0,
                            flops_warmup_steps=4, lambda_initial_ratio=conf.loss.lambda_initial_ratio).to(dev)
    rec, rec_terms = [], []
    def _record(m, i, o):                                 # (a forward hook must return None to keep the output)
        rec.append(o[0].detach())
        rec_terms.append({k_: o[1][k_] for k_ in ("infonce", "flops_q", "flops_d", "flops_neg")})
    loss_fn.register_forward_hook(_record)
    opt = T.build_optimizer(model, conf)
    sch = T.build_scheduler(opt, 0, 4)       # no warm-up: lr(step 0) = lr0, lr(step 1) = 0.854 lr0 -> the second
    step_grads = []                          # window's losses are computed on UPDATED weights
    opt.register_step_pre_hook(lambda o, a, k: step_grads.append(
        {n_: p.grad.detach().clone().cpu() for n_, p in inner.named_parameters()}))
    avg, gs = T.train_epoch(model, dl, loss_fn, opt, sch, conf, epoch=1, global_step=0, device=dev)
    got = [float(x) for x in rec]
    assert gs == 2 and avg == pytest.approx(sum(got) / 8, rel=1e-5)

    # routing per micro-step as (query, positive, negative) triples
    rows_all = torch.cat(captured, 0)
    assert rows_all.shape[0] == 8 * 12
    routes = [tuple(rows_all[12 * i + 4 * j: 12 * i + 4 * j + 4] for j in range(3)) for i in range(8)]
    st = O.TrainState({n: p.clone() for n, p in params.items()})
    ref_grads, ref_losses = [], []
    _, _, gs_ref = O.train_micro_steps(cfg, lc, st, batches, grad_accum=4, base_lr=lr0, wd=0.01, clip=1.0,
                                       warmup=0, total_steps=4, global_step=0, mode="bf16",
                                       route_rows=routes, grads_out=ref_grads, free_losses_out=ref_losses)
    assert gs_ref == 2
    rep = {"losses": got, "ref": [r[0] for r in ref_losses]}
    _report(f"train_epoch_{mode}_losses", rep)
    # loss terms vs the oracle's own (free-routing) loss at its own weights.  FLOPS terms: protocol (iii), rel
    # 2e-3 (5e-3 in the second window, where the two weight sets differ by one Adam step's sign noise).
    # InfoNCE: the reference's in-batch torch.mm runs in bf16 under autocast (ref:losses.py:155), so every score
    # (~250 here) sits on a grid of 1-2 and a one-ulp flip of ONE score moves the batch-mean CE by up to
    # ulp / (tau * B) = 2 / (20 * 4) = 0.025 -- the bound used; the total follows.
    for i, (a, (b, bd)) in enumerate(zip(got, ref_losses)):
        for key in ("flops_q", "flops_d", "flops_neg"):
            assert float(rec_terms[i][key]) == pytest.approx(bd[key], rel=(2e-3 if i < 4 else 5e-3)), (i, key)
        assert float(rec_terms[i]["infonce"]) == pytest.approx(bd["infonce"], abs=0.025), (i, float(rec_terms[i]["infonce"]), bd["infonce"])
        assert a == pytest.approx(b, abs=0.03), (i, got, ref_losses)
    # accumulated gradients of both optimizer steps
    assert len(step_grads) == 2 == len(ref_grads)
    for si in range(2):
        # torch's clip_grad_norm_ has already scaled .grad when the step hook fires; the fused optimizer clips inside
        total = sum(float((g.double() ** 2).sum()) for g in ref_grads[si].values()) ** 0.5
        coef = 1.0 if mode == "native" else min(1.0, 1.0 / (total + 1e-6))
        stats = {n_: _grad_stats(step_grads[si][n_], ref_grads[si][n_] * coef) for n_ in ref_grads[si]}
        worst = min(stats.items(), key=lambda kv: kv[1][0])
        rep[f"step{si}_worst_grad"] = [worst[0], *worst[1]]
        bad = {n_: v for n_, v in stats.items() if v[0] < 0.999 or v[1] > 2e-2}
        assert not bad, (si, bad)
    # parameter values after the two optimizer steps.  An Adam update is lr * m/(sqrt(v)+eps) with |m/sqrt(v)| ~ 1:
    # elements whose gradient is small against its 1-2 % bf16-level error may take a different update (up to
    # 2 lr per step), everything else must agree to a few % of lr.
    lr = lr0
    num = den1 = den2 = 0.0
    worst_mean, worst_frac = 0.0, 0.0
    for n_, p in inner.named_parameters():
        d_got = (p.detach().cpu() - params[n_]).double().flatten()
        d_ref = (st.params[n_] - params[n_]).double().flatten()
        num += float(d_got @ d_ref); den1 += float(d_got @ d_got); den2 += float(d_ref @ d_ref)
        err = (d_got - d_ref).abs()
        worst_mean = max(worst_mean, float(err.mean()) / lr)
        worst_frac = max(worst_frac, float((err > 0.1 * lr).double().mean()))
        assert float(err.max()) <= 2.0 * 1.854 * lr * 1.01, n_
    cos = num / ((den1 * den2) ** 0.5)
    rep.update({"update_cos": cos, "worst_mean_abs_err_over_lr": worst_mean, "worst_frac_err_gt_0.1lr": worst_frac})
    _report(f"train_epoch_{mode}", rep)
    assert cos > 0.995, cos                               # measured 0.9972
    assert worst_mean < 0.03 and worst_frac < 0.03, rep   # measured 0.018 / 0.012


@pytest.mark.parametrize("mode", ["native", "plain"])
def test_train_epoch_vs_oracle_loop(dev, mode):
    train_epoch_case(dev, mode)


def test_train_epoch_on_the_256_wide_kernels(dev):
    """The trainer loop test above, in a child process with the thresholds of the two persistent kernels lowered (they
    are read once per process) so that the small model's 500-odd packed token rows go through gemm_tn256.hip (with a
    ragged rest of M % 64 rows on the 128x128 kernel and tiles half outside the 384- and 768-wide matrices) and
    decoder256.hip: same loss, accumulated-gradient and parameter-value bounds."""
    import subprocess
    import sys
    env = dict(os.environ, SNX_TN256_MIN_M="128", SNX_DEC256_MIN_T="64")
    root = os.path.dirname(os.path.dirname(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_model.py", "-q", "-x", "-m", "gpu", "-k",
                        "test_train_epoch_vs_oracle_loop and native"], cwd=root, env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_inference_encoder_matches_reference_postprocessing(dev):
    """`benchmark.encoders.NeuralSparseEncoderV33` (ref:benchmark/encoders.py:249-402): same methods / return types;
    its device-side top-k equals the oracle's restatement of `_encode_batch` applied to the model's own
    sparse_repr (ids, weights and order), for top_k None / small / large, via encode / encode_single /
    encode_for_query."""
    from benchmark.encoders import NeuralSparseEncoderV33
    from oracle import splade_oracle as O
    from src.model.splade_modern import SPLADEModernBERT
    from src.train.data.collator import create_tokenizer
    tok = create_tokenizer("hash:50000")
    model = SPLADEModernBERT()
    enc = NeuralSparseEncoderV33(checkpoint_path=None, device=dev, model=model, tokenizer=tok)
    texts = ["mi355x sparse retrieval with neural encoders", "one", "a b c d e f g h i j k l m n o p q r s t u v w x y z " * 20]
    tokens = tok.convert_ids_to_tokens(list(range(tok.vocab_size)))
    inputs = tok(texts, max_length=enc.doc_max_length)
    with torch.no_grad():
        rep, _ = enc.model(input_ids=inputs["input_ids"].to(dev), attention_mask=inputs["attention_mask"].to(dev))
    rep = rep.float().cpu()
    for top_k in (None, 5, 100, 3000):
        got = enc.encode(texts, batch_size=2, top_k=top_k)
        assert isinstance(got, list) and len(got) == 3 and all(isinstance(d, dict) for d in got)
        for j, d in enumerate(got):
            # batches of 2 pad differently from the batch of 3 above: compare on a per-text forward when they differ
            want = O.encode_postprocess(rep[j].tolist(), tokens, enc.special_token_ids, top_k)
            if j < 2:
                one = tok(texts[:2], max_length=enc.doc_max_length)
            else:
                one = tok(texts[2:], max_length=enc.doc_max_length)
            with torch.no_grad():
                r2, _ = enc.model(input_ids=one["input_ids"].to(dev), attention_mask=one["attention_mask"].to(dev))
            want = O.encode_postprocess(r2[j % 2].float().cpu().tolist(), tokens, enc.special_token_ids, top_k)
            assert list(d.items()) == want, (top_k, j)
    single = enc.encode_single(texts[1], top_k=7)
    assert isinstance(single, dict) and len(single) <= 7
    q = enc.encode_for_query(texts[0])
    assert isinstance(q, dict) and 0 < len(q) <= 100 and all(isinstance(v, float) for v in q.values())
    assert list(q.values()) == sorted(q.values(), reverse=True) or len(q) < 100
