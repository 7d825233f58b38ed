"""BASELINE.json config 1 AS WRITTEN, on the GPU: ref:configs/train_v33.yaml loss / optimizer values, B = 4, q <= 64 /
d <= 256 ragged, 256 synthetic triplets = 64 micro-steps = 16 optimizer steps (accumulate 4), the 149 M model
random-initialised with seed 42 (SURVEY 2.2 recipe), through `src.train.core.ddp_trainer.train_epoch`
(ref:src/train/cli/train_v33_ddp.py:289-448).

The pin is golden g10 (tools/make_golden.py::g10): the REFERENCE's own unmodified train_epoch on the same weights and
batches, fp32 on the CPU (where its hard-coded cuda autocast is off) -- per-micro-step loss and loss_dict, the update
norm of all 137 tensors, update slices of seven probe tensors.  Checked here:
  * fp32 kernels (the reference's arithmetic outside autocast): every one of the 64 per-micro-step losses and loss terms
    against the reference's, the update norms and probe slices after the 16 optimizer steps;
  * bf16 kernels (what the trainer runs under autocast): finite throughout, the FLOPS terms at the g3 protocol bound
    (rel 1e-2) in the first window -- at random init tau = 1 saturates InfoNCE (scores ~ 2e4), its bf16 value is reported,
    not bounded --, lambda schedule exact; plain module (fused passes), the reference's three-call loop (micro-step arena) and NativeDataParallel;
  * determinism: two NativeDataParallel runs end with bit-equal parameters.
Needs a real MI355X: pytest -m gpu."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
OUT = os.path.join(os.path.dirname(os.path.dirname(__file__)), "gpurun_out")
TERMS = ("infonce", "flops_q", "flops_d", "flops_neg")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def golden():
    meta = json.load(open(os.path.join(G, "g10_config1_train_epoch.json")))
    z = np.load(os.path.join(G, "g10_config1_train_epoch.npz"))
    return meta, z


@pytest.fixture(scope="module")
def setup(golden):
    from oracle import splade_oracle as O
    meta, _ = golden
    c = meta["conf"]
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cfg = O.EncoderConfig()
    params = O.init_params(cfg, seed=c["init_seed"])
    gen = torch.Generator().manual_seed(c["batch_seed"])
    batches = [O.synth_batch(c["batch"], c["q_len"], c["d_len"], cfg, gen, k=1, ragged=True) for _ in range(c["n_micro"])]
    return cfg, params, batches


def _report(name, obj):
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "parity_report.jsonl"), "a") as f:
        f.write(json.dumps({"test": name, **obj}) + "\n")


def _run(dev, setup, golden, mode, precision):
    """-> (per-micro-step losses, per-micro-step term dicts, {name: final parameter on the CPU}, global_step, avg_loss)"""
    from torch.utils.data import DataLoader, Dataset
    from src.model.losses import SPLADELossV33
    from src.train.config.v33 import V33Config
    from src.train.core import ddp_trainer as T
    from tests.test_gpu_model import _build_model
    cfg, params, batches = setup
    c = golden[0]["conf"]
    conf = V33Config()                                        # the dataclass defaults ARE train_v33.yaml's values ...
    assert (conf.loss.lambda_q, conf.loss.lambda_d, conf.loss.temperature, conf.loss.flops_warmup_steps) == \
        (c["lambda_q"], c["lambda_d"], c["temperature"], c["flops_warmup_steps"])
    assert (conf.training.learning_rate, conf.training.weight_decay, conf.training.gradient_clip,
            conf.training.gradient_accumulation_steps) == (c["lr"], c["wd"], c["clip"], c["accum"])
    conf.data.batch_size = c["batch"]                         # ... except config 1's per-process batch of 4

    class DS(Dataset):
        def __len__(self):
            return len(batches)

        def __getitem__(self, i):
            return batches[i]
    old = os.environ.get("SNX_PRECISION")
    old_fused = os.environ.get("SNX_FUSED_PASSES")
    os.environ["SNX_PRECISION"] = precision
    if mode == "three_call":                                  # the reference's literal loop: model(q); model(p); model(n)
        os.environ["SNX_FUSED_PASSES"] = "0"
    try:
        inner = _build_model(cfg, params, dev)
        model = T.NativeDataParallel(inner) if mode == "native" else inner
        loss_fn = SPLADELossV33(lambda_q=conf.loss.lambda_q, lambda_d=conf.loss.lambda_d, temperature=conf.loss.temperature,
                                flops_warmup_steps=conf.loss.flops_warmup_steps, lambda_kd=conf.loss.lambda_kd,
                                kd_temperature=conf.loss.kd_temperature, lambda_initial_ratio=conf.loss.lambda_initial_ratio,
                                lambda_margin_mse=conf.loss.lambda_margin_mse, lambda_neg=conf.loss.lambda_neg).to(dev)
        losses, terms = [], []

        def record(m, i, o):
            losses.append(o[0].detach())
            terms.append({k: o[1][k] for k in TERMS + ("lambda_q", "lambda_d", "lambda_neg")})
        loss_fn.register_forward_hook(record)
        opt = T.build_optimizer(model, conf)
        sch = T.build_scheduler(opt, c["warmup"], c["total_steps"])
        avg, gs = T.train_epoch(model, DataLoader(DS(), batch_size=None, shuffle=False), loss_fn, opt, sch, conf, 1, 0, dev)
        torch.cuda.synchronize()
        final = {n: p.detach().cpu().clone() for n, p in inner.named_parameters()}
        stats = dict(inner.runtime.arena_stats)
    finally:
        for key, val in (("SNX_PRECISION", old), ("SNX_FUSED_PASSES", old_fused)):
            if val is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = val
    got = [float(x) for x in losses]
    got_terms = [{k: float(v) for k, v in t.items()} for t in terms]
    del model, inner, opt
    torch.cuda.empty_cache()
    return got, got_terms, final, gs, avg, stats


def _update_stats(final, params, meta, z):
    names, ref_norms = meta["param_names"], meta["update_norms"]
    ratios = {}
    for n, rn in zip(names, ref_norms):
        key = n.replace("model.decoder.weight", "model.model.embeddings.tok_embeddings.weight")
        if key not in final:
            continue
        ratios[n] = float((final[key].double() - params[key].double()).norm()) / max(rn, 1e-30)
    probes = {}
    for key in z.files:
        if key.startswith("uprobe::"):
            n = key[8:]
            u = final[n].double() - params[n].double()
            got = (u[:8, :64] if u.dim() == 2 else u[:512]).flatten()
            ref = torch.from_numpy(np.asarray(z[key])).double().flatten()
            probes[n] = float((got @ ref) / (got.norm() * ref.norm() + 1e-30))
    return ratios, probes


def test_config1_fp32_kernels_reproduce_the_references_run(dev, setup, golden):
    meta, z = golden
    got, terms, final, gs, avg, _ = _run(dev, setup, golden, "plain", "fp32")
    want = np.asarray(z["losses"], dtype=np.float64)
    assert gs == meta["global_step"] == 16 and len(got) == 64 == len(want)
    assert all(np.isfinite(got))
    rel = np.abs(np.asarray(got) - want) / np.abs(want)
    rep = {"loss_rel_err_first_window": float(rel[:4].max()), "loss_rel_err_max": float(rel.max()), "avg_loss": avg,
           "ref_avg_loss": meta["avg_loss"]}
    term_rel = {}
    for key in TERMS:
        r = [abs(terms[i][key] - meta["dicts"][i][key]) / max(abs(meta["dicts"][i][key]), 1e-12) for i in range(64)]
        term_rel[key] = [max(r[:4]), max(r)]
    rep["term_rel_err_first_window_and_max"] = term_rel
    ratios, probes = _update_stats(final, setup[1], meta, z)
    rep["update_norm_ratio_minmax"] = [min(ratios.values()), max(ratios.values())]
    rep["probe_update_cos"] = probes
    _report("config1_fp32_vs_reference_train_epoch_g10", rep)
    # first window: identical weights -> only fp32 summation order separates the two runs (InfoNCE is a difference of
    # 50,000-term dot products of size ~2e4: 1e-4 of the loss is ~0.2 absolute, a few fp32 ulps of a score)
    assert rel[:4].max() <= 1e-4, rep
    for key in TERMS:
        assert term_rel[key][0] <= (1e-4 if key == "infonce" else 5e-5), (key, term_rel[key])
    # after optimizer steps Adam turns gradients whose sign is rounding noise into +-lr steps; the loss is insensitive to
    # exactly those elements (their gradient is ~0), so the run stays on the reference's trajectory
    # (measured round 6: 1.2e-5 on the loss, 2.6e-6 on the FLOPS terms over all 64 micro-steps)
    assert rel.max() <= 2e-4, rep
    for key in ("flops_q", "flops_d", "flops_neg"):
        assert term_rel[key][1] <= 5e-5, (key, term_rel[key])
    for i in range(64):
        for key in ("lambda_q", "lambda_d", "lambda_neg"):
            assert terms[i][key] == pytest.approx(meta["dicts"][i][key], rel=1e-12), (i, key)
    assert avg == pytest.approx(meta["avg_loss"], rel=2e-4)
    # every tensor's update after the 16 optimizer steps: norm within 1 % of the reference's (measured 0.9992 .. 1.0006), the
    # probe slices' direction cos >= 0.999 (measured >= 0.99994)
    assert 0.99 <= min(ratios.values()) and max(ratios.values()) <= 1.01, rep["update_norm_ratio_minmax"]
    assert min(probes.values()) >= 0.999, probes


@pytest.mark.parametrize("mode", ["plain", "three_call", "native"])
def test_config1_bf16_kernels_through_train_epoch(dev, setup, golden, mode):
    meta, z = golden
    got, terms, final, gs, avg, stats = _run(dev, setup, golden, mode, "bf16")
    assert gs == 16 and len(got) == 64 and all(np.isfinite(got)) and np.isfinite(avg)
    assert all(torch.isfinite(p).all() for p in final.values())
    rel = {}
    for key in ("flops_q", "flops_d", "flops_neg"):
        r = [abs(terms[i][key] - meta["dicts"][i][key]) / abs(meta["dicts"][i][key]) for i in range(64)]
        rel[key] = [max(r[:4]), max(r)]
        assert rel[key][0] <= 1e-3, (key, rel[key])           # bf16 kernels vs the reference's fp32 value (g3 protocol: 1e-2;
        assert rel[key][1] <= 5e-3, (key, rel[key])           # measured 3e-5 in the first window, 2.4e-4 over the 16 optimizer steps)
    for i in range(64):
        for key in ("lambda_q", "lambda_d", "lambda_neg"):
            assert terms[i][key] == pytest.approx(meta["dicts"][i][key], rel=1e-12), (i, key)
    ratios, probes = _update_stats(final, setup[1], meta, z)
    assert 0.9 <= min(ratios.values()) and max(ratios.values()) <= 1.1, (min(ratios.values()), max(ratios.values()))
    assert min(probes.values()) >= 0.9, probes            # (measured 0.97: bf16 gradients through Adam's sign-like first steps)
    inf = [abs(terms[i]["infonce"] - meta["dicts"][i]["infonce"]) for i in range(64)]
    _report(f"config1_bf16_{mode}_vs_reference_train_epoch_g10",
            {"flops_rel_err_first_window_and_max": rel, "infonce_abs_diff_max (saturated; reported)": max(inf),
             "ref_infonce_range": [min(d["infonce"] for d in meta["dicts"]), max(d["infonce"] for d in meta["dicts"])],
             "update_norm_ratio_minmax": [min(ratios.values()), max(ratios.values())], "probe_update_cos": probes,
             "arena_stats": stats, "avg_loss": avg, "ref_avg_loss": meta["avg_loss"]})
    if mode == "three_call":                                  # the unchanged loop went through the micro-step arena: every pass
        # placed or, where dynamic padding outgrew the learnt capacity, counted as a fallback -- never silently
        assert stats["placed"] + stats["fell_back"] >= 3 * 62 and stats["placed"] >= 3 * 40, stats
        assert stats["zero_filled"] == 0 and stats["stale_dropped"] == 0, stats


def test_config1_two_runs_are_bit_equal(dev, setup, golden):
    a = _run(dev, setup, golden, "native", "bf16")
    b = _run(dev, setup, golden, "native", "bf16")
    assert a[0] == b[0]
    for n in a[2]:
        assert torch.equal(a[2][n], b[2][n]), n
