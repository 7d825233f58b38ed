#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE CODE.

Build-container only (needs /root/reference and transformers; neither is needed at test time).
It loads the reference's hot-path files by file path
    /root/reference/src/model/splade_modern.py   (SPLADEModernBERT)
    /root/reference/src/model/losses.py          (SPLADELossV33)
    /root/reference/src/train/data/dataloader.py (TripletCollator)
    /root/reference/src/train/cli/train_v33_ddp.py (train_epoch; stubs only for ABSENT modules)
builds the inner ``ModernBertForMaskedLM`` offline from a config (never from the hub name),
loads weights produced by ``oracle.splade_oracle.init_params`` (so both sides see identical
tensors) and records inputs + outputs as small .npz/.json files.  Only data is written; no
reference source is copied.

    python tools/make_golden.py            # all fixtures
    python tools/make_golden.py g1 g4      # a subset

Fixtures: g1 tiny fwd/bwd (+g6 edge rows), g2 tiny train_epoch (1-proc and
2-proc gloo DDP), g3 full-size fwd/bwd summaries, g4 loss-only vectors, g5 collator layout,
g6 inference post-processing, g7 BASELINE config 5 (d512, k=4, MarginMSE) at full size, g8 the g3 batch
with an unsaturated InfoNCE, g9 the loss's KL-distillation branch, g10 BASELINE config 1 as written (64 micro-steps of the
149 M model through the reference's train_epoch).
"""
from __future__ import annotations

import hashlib
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import splade_oracle as O  # noqa: E402


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def hf_config(cfg: O.EncoderConfig):
    from transformers import ModernBertConfig
    with open(os.path.join(REF, "huggingface/v33/config.json")) as f:
        raw = json.load(f)
    for k in ("architectures", "model_type", "transformers_version", "dtype"):
        raw.pop(k, None)
    raw.update(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
               intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_hidden_layers,
               num_attention_heads=cfg.num_attention_heads, local_attention=cfg.local_attention,
               pad_token_id=cfg.pad_token_id)
    return ModernBertConfig(**raw)


def build_reference_model(cfg: O.EncoderConfig, params, attn_impl=None):
    from transformers import AutoModelForMaskedLM
    sm = _load(os.path.join(REF, "src/model/splade_modern.py"), "ref_splade_modern")
    hc = hf_config(cfg)
    if attn_impl:
        hc._attn_implementation = attn_impl
    inner = AutoModelForMaskedLM.from_config(hc)
    m = sm.SPLADEModernBERT.__new__(sm.SPLADEModernBERT)
    torch.nn.Module.__init__(m)
    m.model_name = "offline"
    m.model = inner
    m.config = inner.config
    m.relu = torch.nn.ReLU()
    sd = {k: v.clone() for k, v in params.items()}
    sd["model.decoder.weight"] = sd["model.model.embeddings.tok_embeddings.weight"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("inv_freq" in k for k in missing), missing
    m.train()
    return m


def ref_loss_module(**kw):
    lm = _load(os.path.join(REF, "src/model/losses.py"), "ref_losses")
    return lm.SPLADELossV33(**kw)


def np_(t):
    return t.detach().cpu().numpy()


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()


def edge_batch(cfg, gen, B, Sq, Sd, k, lens_q, lens_d, lens_n):
    b = O.synth_batch(B, Sq, Sd, cfg, gen, k=k, ragged=True, teacher=True)

    def fix(ids, mask, lens):
        S = ids.shape[1]
        lens = torch.tensor(lens)
        mask = (torch.arange(S)[None] < lens[:, None]).long()
        ids = torch.randint(6, cfg.pad_token_id, ids.shape, generator=gen)
        ids[:, 0] = 0
        ids[torch.arange(len(lens)), lens - 1] = 1
        ids = torch.where(mask.bool(), ids, torch.full_like(ids, cfg.pad_token_id))
        return ids, mask
    b["query_input_ids"], b["query_attention_mask"] = fix(b["query_input_ids"], None, lens_q)
    b["positive_input_ids"], b["positive_attention_mask"] = fix(b["positive_input_ids"], None, lens_d)
    b["negative_input_ids"], b["negative_attention_mask"] = fix(b["negative_input_ids"], None, lens_n)
    return b


def run_triplet(model, loss_fn, b, global_step):
    q, qt = model(b["query_input_ids"], b["query_attention_mask"])
    p, pt = model(b["positive_input_ids"], b["positive_attention_mask"])
    n, nt = model(b["negative_input_ids"], b["negative_attention_mask"])
    k = int(b["num_negatives"])
    n3 = n.view(q.shape[0], k, -1) if k > 1 else n
    for t in (q, p, n):
        t.retain_grad()
    loss, d = loss_fn(anchor_repr=q, positive_repr=p, negative_repr=n3, global_step=global_step,
                      teacher_pos_scores=b.get("teacher_pos_scores"),
                      teacher_neg_scores=b.get("teacher_neg_scores"))
    return loss, d, (q, p, n), (qt, pt, nt)


# ---------------------------------------------------------------------------------------
def g1():
    """tiny config: ragged + edge rows (single-token row, all-pad local windows), k=2 negatives,
    MarginMSE on; outputs, loss, all parameter grads; sdpa vs eager agreement recorded."""
    cfg = O.EncoderConfig.tiny()
    params = O.perturb_params(O.init_params(cfg, seed=42), seed=7, scale=3.0)
    gen = torch.Generator().manual_seed(1234)
    b = edge_batch(cfg, gen, B=4, Sq=16, Sd=32, k=2, lens_q=[16, 1, 5, 11],
                   lens_d=[32, 1, 9, 20], lens_n=[32, 3, 17, 1, 8, 25, 32, 12])
    lkw = dict(lambda_q=0.01, lambda_d=0.003, temperature=25.0, flops_warmup_steps=100,
               lambda_initial_ratio=0.1, lambda_margin_mse=0.01, lambda_neg=0.0)
    model = build_reference_model(cfg, params)
    loss_fn = ref_loss_module(**lkw)
    loss, d, reps, tws = run_triplet(model, loss_fn, b, global_step=37)
    loss.backward()
    grads = {n: p.grad for n, p in model.named_parameters()}
    # eager attention agreement
    model_e = build_reference_model(cfg, params, attn_impl="eager")
    with torch.no_grad():
        qe, _ = model_e(b["positive_input_ids"], b["positive_attention_mask"])
    eager_diff = float((qe - reps[1]).abs().max())
    arrs = {"w::" + k: np_(v) for k, v in params.items()}
    arrs.update({"in::" + k: np_(v) for k, v in b.items() if torch.is_tensor(v)})
    arrs.update({"out::q": np_(reps[0]), "out::p": np_(reps[1]), "out::n": np_(reps[2]),
                 "out::qt": np_(tws[0]), "out::pt": np_(tws[1]), "out::nt": np_(tws[2]),
                 "out::dq": np_(reps[0].grad), "out::dp": np_(reps[1].grad), "out::dn": np_(reps[2].grad),
                 "out::loss": np.float64(loss.item())})
    for n, g in grads.items():
        key = n.replace("model.decoder.weight", "model.model.embeddings.tok_embeddings.weight")
        arrs["g::" + key] = np_(g)
    np.savez_compressed(os.path.join(OUT, "g1_tiny_fwd_bwd.npz"), **arrs)
    meta = {"loss_kwargs": lkw, "global_step": 37, "num_negatives": 2, "loss_dict": d,
            "sdpa_vs_eager_max_abs": eager_diff, "transformers": __import__("transformers").__version__,
            "torch": torch.__version__}
    json.dump(meta, open(os.path.join(OUT, "g1_tiny_fwd_bwd.json"), "w"), indent=1)
    print("g1 loss", loss.item(), "eager diff", eager_diff)


# ---------------------------------------------------------------------------------------
def _import_ref_trainer():
    """ref:src/train/cli/train_v33_ddp.py with stubs ONLY for modules absent from the reference
    tree / this image (sentence_transformers, src.train.data package init, data.collator)."""
    if "src.train.cli.train_v33_ddp" in sys.modules:
        return sys.modules["src.train.cli.train_v33_ddp"]
    sys.path.insert(0, REF)
    st = types.ModuleType("sentence_transformers")
    st.SentenceTransformer = type("SentenceTransformer", (), {})
    sys.modules.setdefault("sentence_transformers", st)
    pkg = types.ModuleType("src.train.data")
    pkg.__path__ = [os.path.join(REF, "src/train/data")]
    pkg.load_training_data = lambda *a, **k: None
    sys.modules["src.train.data"] = pkg
    col = types.ModuleType("src.train.data.collator")
    col.create_tokenizer = lambda *a, **k: None
    sys.modules["src.train.data.collator"] = col
    import importlib
    return importlib.import_module("src.train.cli.train_v33_ddp")


class _Batches(torch.utils.data.Dataset):
    def __init__(self, batches):
        self.b = batches

    def __len__(self):
        return len(self.b)

    def __getitem__(self, i):
        return self.b[i]


def _g2_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T = _import_ref_trainer()
    cfg = O.EncoderConfig.tiny()
    params = O.perturb_params(O.init_params(cfg, seed=42), seed=7, scale=3.0)
    torch.manual_seed(42 + rank)
    model = build_reference_model(cfg, params)
    ddp = T.DDP(model, broadcast_buffers=False, find_unused_parameters=False)
    conf = T.V33Config()
    conf.training.gradient_accumulation_steps = 4
    conf.training.learning_rate = 5e-3      # large enough that 2 optimizer steps move weights visibly
    conf.training.log_every_n_steps = 1
    conf.loss.flops_warmup_steps = 4
    loss_fn = T.SPLADELossV33(lambda_q=conf.loss.lambda_q, lambda_d=conf.loss.lambda_d,
                              temperature=conf.loss.temperature,
                              flops_warmup_steps=conf.loss.flops_warmup_steps,
                              lambda_initial_ratio=conf.loss.lambda_initial_ratio)
    rec = []
    loss_fn.register_forward_hook(lambda m, i, o: rec.append((float(o[0].item()), dict(o[1]))))
    no_decay = ["bias", "LayerNorm.weight", "layer_norm.weight"]
    groups = [{"params": [p for n, p in ddp.named_parameters() if not any(nd in n for nd in no_decay)],
               "weight_decay": conf.training.weight_decay},
              {"params": [p for n, p in ddp.named_parameters() if any(nd in n for nd in no_decay)],
               "weight_decay": 0.0}]
    opt = T.AdamW(groups, lr=conf.training.learning_rate)
    total_steps, warm = 4, 1
    sched = T.get_cosine_schedule_with_warmup(opt, num_warmup_steps=warm, num_training_steps=total_steps)
    # every rank builds the same 8*world batches; DistributedSampler(shuffle=False) deals them out
    gen = torch.Generator().manual_seed(99)
    batches = [O.synth_batch(4, 16, 32, cfg, gen, k=1, ragged=True) for _ in range(8 * world)]
    ds = _Batches(batches)
    sampler = T.DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=False)
    dl = T.DataLoader(ds, batch_size=None, sampler=sampler)
    avg, gs = T.train_epoch(ddp, dl, loss_fn, opt, sched, conf, epoch=1, global_step=0,
                            device=torch.device("cpu"), tb_logger=None)
    if rank == 0:
        ret["losses"] = [r[0] for r in rec]
        ret["dicts"] = [r[1] for r in rec]
        ret["avg_loss"], ret["global_step"] = avg, gs
        ret["params"] = {n: p.detach().clone() for n, p in model.named_parameters()}
        ret["batches"] = batches
        ret["conf"] = {"lr": conf.training.learning_rate, "wd": conf.training.weight_decay,
                       "clip": conf.training.gradient_clip, "accum": 4, "warmup": warm,
                       "total_steps": total_steps, "lambda_q": conf.loss.lambda_q,
                       "lambda_d": conf.loss.lambda_d, "flops_warmup_steps": conf.loss.flops_warmup_steps,
                       "lambda_initial_ratio": conf.loss.lambda_initial_ratio, "world": world}
    dist.barrier()
    dist.destroy_process_group()


def _g2_run(world, port, tag):
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    if world == 1:
        _g2_worker(0, 1, port, ret)
    else:
        mp.spawn(_g2_worker, args=(world, port, ret), nprocs=world, join=True)
    ret = dict(ret)
    arrs = {"p::" + n.replace("model.decoder.weight", "model.model.embeddings.tok_embeddings.weight"): np_(v)
            for n, v in ret["params"].items()}
    for i, b in enumerate(ret["batches"]):
        for k, v in b.items():
            if torch.is_tensor(v):
                arrs[f"b{i}::{k}"] = np_(v)
    arrs["losses"] = np.array(ret["losses"], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, f"g2_train_epoch_{tag}.npz"), **arrs)
    json.dump({"conf": ret["conf"], "avg_loss": ret["avg_loss"], "global_step": ret["global_step"],
               "dicts": ret["dicts"], "n_batches": len(ret["batches"])},
              open(os.path.join(OUT, f"g2_train_epoch_{tag}.json"), "w"), indent=1)
    print("g2", tag, "losses", ret["losses"], "global_step", ret["global_step"])


def g2():
    """The reference's own unmodified train_epoch on the tiny config: 8 micro-steps, accum 4 ->
    2 optimizer steps (1-proc gloo), and the same with 2 gloo ranks (DDP gradient averaging)."""
    _g2_run(1, 29611, "w1")
    _g2_run(2, 29612, "w2")


# ---------------------------------------------------------------------------------------
def _full_fixture(name, B, Sq, Sd, k, lkw, step, teacher, batch_seed, extra_probe_rows=False):
    """Full 149M config through the reference: output summaries + gradient probes (weights are regenerated
    from the seed by the test; only checksums are stored)."""
    cfg = O.EncoderConfig()
    params = O.perturb_params(O.init_params(cfg, seed=42), seed=7, bias_mean=-0.2)
    gen = torch.Generator().manual_seed(batch_seed)
    b = O.synth_batch(B, Sq, Sd, cfg, gen, k=k, ragged=True, teacher=teacher)
    model = build_reference_model(cfg, params)
    loss_fn = ref_loss_module(**lkw)
    loss, d, reps, tws = run_triplet(model, loss_fn, b, global_step=step)
    loss.backward()
    arrs = {"in::" + k_: np_(v) for k_, v in b.items() if torch.is_tensor(v)}
    for tag, r, tw in zip("qpn", reps, tws):
        v, i = torch.topk(r.detach(), 256, dim=-1)
        arrs[f"out::{tag}_topv"], arrs[f"out::{tag}_topi"] = np_(v), np_(i)
        arrs[f"out::{tag}_sum"] = np_(r.detach().double().sum(-1))
        arrs[f"out::{tag}_sq"] = np_((r.detach().double() ** 2).sum(-1))
        arrs[f"out::{tag}_tw"] = np_(tw)
        arrs[f"out::{tag}_full"] = np_(r.detach()).astype(np.float16)   # coarse full vector (fp16)
        arrs[f"out::d{tag}_sum"] = np_(r.grad.double().sum(-1))
        if extra_probe_rows:
            arrs[f"out::d{tag}_row0"] = np_(r.grad[0]).astype(np.float32)
    names, gnorm = [], []
    for n, p in model.named_parameters():
        names.append(n)
        gnorm.append(float(p.grad.double().norm()))
    probes = ["model.model.layers.0.attn.Wqkv.weight", "model.model.layers.10.mlp.Wi.weight",
              "model.model.layers.21.mlp.Wo.weight", "model.head.dense.weight",
              "model.model.final_norm.weight", "model.decoder.bias"]
    pg = dict(model.named_parameters())
    for n in probes:
        g = pg[n].grad
        arrs["gprobe::" + n] = np_(g[:8, :64] if g.dim() == 2 else g[:512])
    e = pg["model.model.embeddings.tok_embeddings.weight"].grad
    arrs["gprobe::emb_rows"] = np_(e[:16, :64])
    arrs["gprobe::emb_rownorm"] = np_(e.double().norm(dim=1)).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    json.dump({"loss_kwargs": lkw, "global_step": step, "loss": float(loss.item()), "loss_dict": d,
               "num_negatives": k, "shape": {"B": B, "Sq": Sq, "Sd": Sd},
               "grad_names": names, "grad_norms": gnorm,
               "weight_sha256": {k_: sha(v) for k_, v in list(params.items())[:8]},
               "init": "oracle.init_params(seed=42) + perturb_params(seed=7, bias_mean=-0.2)",
               "batch": f"oracle.synth_batch(seed={batch_seed}, ragged, teacher={teacher})",
               "transformers": __import__("transformers").__version__, "torch": torch.__version__},
              open(os.path.join(OUT, name + ".json"), "w"), indent=1)
    print(name, "loss", loss.item(), d)


def g3():
    """full 149M config, B=4, q64/d256 ragged (BASELINE config 1/2 shape)."""
    _full_fixture("g3_full_fwd_bwd", 4, 64, 256, 1,
                  dict(lambda_q=0.01, lambda_d=0.003, temperature=1.0, flops_warmup_steps=20000,
                       lambda_initial_ratio=0.1), 1000, False, 4242)


def g7():
    """BASELINE config 5 through the reference: full 149M model, B=2, q64 / d512, k=4 `negatives`
    (flattened [B*k, S] then viewed [B, k, V], ref:train_v33_ddp.py:346-350), MarginMSE 0.5 with teacher
    scores and the other loss values of ref:configs/train_v34_multi_neg.yaml:20-28."""
    _full_fixture("g7_cfg5_d512_k4", 2, 64, 512, 4,
                  dict(lambda_q=0.01, lambda_d=0.003, temperature=1.0, flops_warmup_steps=5000,
                       lambda_kd=0.0, kd_temperature=1.0, lambda_margin_mse=0.5, lambda_initial_ratio=0.5),
                  1000, True, 5151, extra_probe_rows=True)


def g8():
    """The g3 batch (same seed -> same ids) at a temperature where InfoNCE is NOT saturated: random-init
    sparse vectors have dot products ~2e4 with a spread of a few hundred, so tau = 500 puts the logits'
    spread at O(1) and the softmax coefficients carry information (g3's tau = 1 is a one-hot)."""
    _full_fixture("g8_full_unsaturated", 4, 64, 256, 1,
                  dict(lambda_q=0.01, lambda_d=0.003, temperature=500.0, flops_warmup_steps=20000,
                       lambda_initial_ratio=0.1), 1000, False, 4242, extra_probe_rows=True)


# ---------------------------------------------------------------------------------------
def g4():
    """loss-only vectors: non-negative sparse inputs, B in {4,64}, k in {1,4,7}, steps
    {0,T/2,T,2T}; loss, terms, input grads; MarginMSE on for half the cases."""
    gen = torch.Generator().manual_seed(77)
    cases, arrs = [], {}
    ci = 0
    for B, V in ((4, 512), (64, 128)):
        for k in (1, 4, 7):
            for step in (0, 50, 100, 200):
                mm = 0.5 if (ci % 2) else 0.0
                lneg = 0.004 if (ci % 3 == 0) else 0.0
                lkw = dict(lambda_q=0.01, lambda_d=0.003, temperature=1.0 if ci % 4 else 0.7,
                           flops_warmup_steps=100, lambda_initial_ratio=0.1,
                           lambda_margin_mse=mm, lambda_neg=lneg)

                def sp(*shape):
                    x = torch.rand(*shape, generator=gen)
                    return (torch.relu(x - 0.6) * 3.0).requires_grad_(True)
                a, p = sp(B, V), sp(B, V)
                n = sp(B, k, V) if k > 1 else sp(B, V)
                tp = 0.5 + 0.5 * torch.rand(B, generator=gen)
                tn = 0.6 * torch.rand(B, k, generator=gen) if k > 1 else 0.6 * torch.rand(B, generator=gen)
                lf = ref_loss_module(**lkw)
                loss, d = lf(anchor_repr=a, positive_repr=p, negative_repr=n, global_step=step,
                             teacher_pos_scores=tp, teacher_neg_scores=tn)
                loss.backward()
                pre = f"c{ci}::"
                arrs.update({pre + "a": np_(a), pre + "p": np_(p), pre + "n": np_(n), pre + "tp": np_(tp),
                             pre + "tn": np_(tn), pre + "da": np_(a.grad), pre + "dp": np_(p.grad),
                             pre + "dn": np_(n.grad)})
                cases.append({"id": ci, "B": B, "V": V, "k": k, "step": step, "loss_kwargs": lkw,
                              "loss": float(loss.item()), "loss_dict": d,
                              "avg_nonzero": list(lf.get_avg_nonzero())})
                ci += 1
    np.savez_compressed(os.path.join(OUT, "g4_loss_vectors.npz"), **arrs)
    json.dump(cases, open(os.path.join(OUT, "g4_loss_vectors.json"), "w"), indent=1)
    print("g4 cases", len(cases))


# ---------------------------------------------------------------------------------------
class StubTokenizer:
    """Deterministic whitespace tokenizer used on BOTH sides of the collator fixture (the real
    BertTokenizer files are reference data we do not copy): id = 6 + crc32(word) % 40000."""
    pad_token_id = 49999

    def __call__(self, texts, padding=True, truncation=True, max_length=64, return_tensors="pt"):
        import zlib
        rows = []
        for t in texts:
            ids = [0] + [6 + zlib.crc32(w.encode()) % 40000 for w in t.split()] + [1]
            if truncation and len(ids) > max_length:
                ids = ids[:max_length - 1] + [1]
            rows.append(ids)
        L = max(len(r) for r in rows)
        ids = torch.full((len(rows), L), self.pad_token_id, dtype=torch.long)
        mask = torch.zeros((len(rows), L), dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = torch.tensor(r)
            mask[i, :len(r)] = 1
        return {"input_ids": ids, "attention_mask": mask}


def g5():
    """TripletCollator layout (ref:src/train/data/dataloader.py:46-164) with the stub tokenizer:
    single-negative (one item missing its negative), multi-negative (short lists padded by
    repeating the last, teacher scores)."""
    dl = _load(os.path.join(REF, "src/train/data/dataloader.py"), "ref_dataloader")
    tok = StubTokenizer()
    col = dl.TripletCollator(tokenizer=tok, max_length=12, query_max_length=6, doc_max_length=12)
    words = "alpha beta gamma delta epsilon zeta eta theta iota kappa lambda mu nu xi omicron pi rho".split()
    rng = np.random.RandomState(5)

    def text(n):
        return " ".join(rng.choice(words, n))
    single = [{"query": text(3), "positive": text(8), "negative": text(15), "teacher_pos_score": 0.9,
               "teacher_neg_score": 0.2, "pair_type": "a", "difficulty": "hard"},
              {"query": text(9), "positive": text(4), "negative": None, "teacher_pos_score": 0.7,
               "teacher_neg_score": 0.1, "pair_type": "b"},
              {"query": text(2), "positive": text(20), "negative": text(3), "teacher_pos_score": 0.8,
               "teacher_neg_score": 0.3}]
    multi = [{"query": text(3), "positive": text(8), "negatives": [text(5), text(6), text(7)],
              "teacher_pos_score": 0.9, "teacher_neg_scores": [0.1, 0.2, 0.3]},
             {"query": text(4), "positive": text(5), "negatives": [text(2)],
              "teacher_pos_score": 0.6, "teacher_neg_scores": [0.1, 0.1, 0.1]},
             {"query": text(5), "positive": text(6), "negatives": [],
              "teacher_pos_score": 0.5, "teacher_neg_scores": [0.0, 0.0, 0.0]}]
    import copy
    fixture = {}
    for tag, items in (("single", single), ("multi", multi)):
        inp = copy.deepcopy(items)
        out = col(copy.deepcopy(items))
        ser = {}
        for k, v in out.items():
            ser[k] = v.tolist() if torch.is_tensor(v) else v
        fixture[tag] = {"items": inp, "out": ser}
    fixture["collator_kwargs"] = {"max_length": 12, "query_max_length": 6, "doc_max_length": 12}
    json.dump(fixture, open(os.path.join(OUT, "g5_collator.json"), "w"), indent=1)
    print("g5 ok")


def g6():
    """Inference post-processing (ref:benchmark/encoders.py:309-345 `_encode_batch`): non-zero entries ->
    special-token / "[..." / "<..." / empty-token filter -> optional top-k by weight.  The reference method
    is run unmodified on an instance assembled without its constructor (which needs the hub): `model` is
    a stand-in returning the given sparse_repr, the token table and special ids are the fixture's."""
    _import_ref_trainer()                      # puts REF on sys.path + the sentence_transformers stub
    import importlib
    # `benchmark/__init__.py` pulls in boto3 / opensearch-py (absent here): register the package by path
    # only, so that benchmark.config / benchmark.dataset / benchmark.encoders load from their own files
    pkg = types.ModuleType("benchmark")
    pkg.__path__ = [os.path.join(REF, "benchmark")]
    sys.modules["benchmark"] = pkg
    enc_mod = importlib.import_module("benchmark.encoders")
    g = torch.Generator().manual_seed(606)
    V, B = 700, 7
    tokens = []
    for i in range(V):
        r = i % 23
        tokens.append("" if r == 5 else f"[unused{i}]" if r == 7 else f"<tok{i}>" if r == 11 else f"w{i}")
    special = [0, 1, 2, 3, 4, 699]
    rep = torch.relu(torch.randn(B, V, generator=g) - 0.8)          # ~20 % active
    rep[1] = torch.relu(torch.randn(V, generator=g) + 1.0)           # dense row
    rep[2] = 0.0                                                     # empty row
    rep[3] = (torch.randint(0, 4, (V,), generator=g).float() * 0.5)  # many exact ties
    rep[4, :] = 0.0
    rep[4, [10, 20, 30]] = torch.tensor([0.25, 0.25, 0.75])          # fewer than k
    rep[5] = rep[5].to(torch.bfloat16).float()
    rep[6, special] = 9.0                                            # large weights on filtered ids
    enc = object.__new__(enc_mod.NeuralSparseEncoderV33)
    enc.special_token_ids = set(special)
    enc._token_lookup = tokens
    enc.model = lambda input_ids=None, attention_mask=None: (rep, None)
    cases = {}
    for k in (None, 1, 3, 5, 50, 128, 699, 5000):
        out = enc._encode_batch({"input_ids": None, "attention_mask": None}, k)
        cases[str(k)] = [[[t, w] for t, w in d.items()] for d in out]
    json.dump({"V": V, "tokens": tokens, "special": special, "rep": rep.tolist(), "cases": cases},
              open(os.path.join(OUT, "g6_encode_topk.json"), "w"))
    print("g6 ok", {k: [len(d) for d in v] for k, v in cases.items()})


def g9():
    """The KL-distillation branch of SPLADELossV33 (ref:src/model/losses.py:239-253: softmax(teacher / T) vs
    log_softmax(q p^T / T), batchmean) -- inactive in the V33 trainer (it never passes `teacher_scores`,
    ref:train_v33_ddp.py:353-360) but part of the class API: loss, terms and input gradients with lambda_kd > 0."""
    gen = torch.Generator().manual_seed(909)
    cases, arrs = [], {}
    ci = 0
    for B, V, k in ((4, 512, 1), (64, 128, 1), (8, 300, 4)):
        for lkd, kT in ((0.3, 1.0), (1.0, 2.5)):
            lkw = dict(lambda_q=0.01, lambda_d=0.003, temperature=1.0, flops_warmup_steps=100,
                       lambda_initial_ratio=0.1, lambda_kd=lkd, kd_temperature=kT,
                       lambda_margin_mse=0.5 if ci % 2 else 0.0)

            def sp(*shape):
                x = torch.rand(*shape, generator=gen)
                return (torch.relu(x - 0.6) * 3.0).requires_grad_(True)
            a, p = sp(B, V), sp(B, V)
            n = sp(B, k, V) if k > 1 else sp(B, V)
            ts = 4.0 * torch.rand(B, B, generator=gen)
            tp = 0.5 + 0.5 * torch.rand(B, generator=gen)
            tn = 0.6 * torch.rand(B, k, generator=gen) if k > 1 else 0.6 * torch.rand(B, generator=gen)
            lf = ref_loss_module(**lkw)
            loss, d = lf(anchor_repr=a, positive_repr=p, negative_repr=n, global_step=40, teacher_scores=ts,
                         teacher_pos_scores=tp, teacher_neg_scores=tn)
            loss.backward()
            pre = f"c{ci}::"
            arrs.update({pre + "a": np_(a), pre + "p": np_(p), pre + "n": np_(n), pre + "ts": np_(ts),
                         pre + "tp": np_(tp), pre + "tn": np_(tn), pre + "da": np_(a.grad), pre + "dp": np_(p.grad),
                         pre + "dn": np_(n.grad)})
            cases.append({"id": ci, "B": B, "V": V, "k": k, "step": 40, "loss_kwargs": lkw,
                          "loss": float(loss.item()), "loss_dict": d})
            ci += 1
    np.savez_compressed(os.path.join(OUT, "g9_loss_kd.npz"), **arrs)
    json.dump(cases, open(os.path.join(OUT, "g9_loss_kd.json"), "w"), indent=1)
    print("g9 cases", len(cases), [c["loss_dict"]["kd"] for c in cases])


def g10():
    """BASELINE config 1 AS WRITTEN through the reference's own unmodified train_epoch (ref:train_v33_ddp.py:289-448):
    ref:configs/train_v33.yaml loss / optimizer values (lambda_q 0.01, lambda_d 0.003, tau 1.0, FLOPS warm-up 20000, lr 5e-5,
    wd 0.01, warm-up ratio 0.06, clip 1.0, accum 4, 25 epochs for the schedule's horizon), 1 process (gloo), fp32 CPU,
    B = 4, 256 synthetic triplets (q <= 64 / d <= 256 ragged, SURVEY 8(d) generator seed 42) = 64 micro-steps = 16
    optimizer steps, the 149 M model random-initialised by the SURVEY 2.2 recipe with seed 42.  Records the per-micro-step
    loss and loss_dict, the update norm of all 137 tensors and update slices of probe tensors (the weights themselves:
    597 MB, recreated from the recipe by the test).  ~8 minutes of CPU."""
    import time
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29633")
    dist.init_process_group("gloo", rank=0, world_size=1)
    T = _import_ref_trainer()
    cfg = O.EncoderConfig()
    params = O.init_params(cfg, seed=42)
    torch.manual_seed(42)
    model = build_reference_model(cfg, params)
    init = {n: p.detach().clone() for n, p in model.named_parameters()}
    ddp = T.DDP(model, broadcast_buffers=False, find_unused_parameters=False)
    conf = T.V33Config()
    import yaml
    raw = yaml.safe_load(open(os.path.join(REF, "configs/train_v33.yaml")))
    for k, v in raw["loss"].items():
        setattr(conf.loss, k, v)
    for k, v in raw["training"].items():
        setattr(conf.training, k, v)
    conf.data.batch_size = 4
    loss_fn = T.SPLADELossV33(lambda_q=conf.loss.lambda_q, lambda_d=conf.loss.lambda_d, temperature=conf.loss.temperature,
                              flops_warmup_steps=conf.loss.flops_warmup_steps, lambda_kd=conf.loss.lambda_kd,
                              kd_temperature=conf.loss.kd_temperature, lambda_initial_ratio=conf.loss.lambda_initial_ratio,
                              lambda_margin_mse=conf.loss.lambda_margin_mse, lambda_neg=conf.loss.lambda_neg)
    rec = []
    loss_fn.register_forward_hook(lambda m, i, o: rec.append((float(o[0].item()), dict(o[1]))))
    no_decay = ["bias", "LayerNorm.weight", "layer_norm.weight"]
    groups = [{"params": [p for n, p in ddp.named_parameters() if not any(nd in n for nd in no_decay)],
               "weight_decay": conf.training.weight_decay},
              {"params": [p for n, p in ddp.named_parameters() if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
    opt = T.AdamW(groups, lr=conf.training.learning_rate)
    n_micro, accum = 64, conf.training.gradient_accumulation_steps
    total_steps = (n_micro // accum) * conf.training.num_epochs          # ref:train_v33_ddp.py:584-586
    warm = int(total_steps * conf.training.warmup_ratio)
    sched = T.get_cosine_schedule_with_warmup(opt, num_warmup_steps=warm, num_training_steps=total_steps)
    gen = torch.Generator().manual_seed(42)
    batches = [O.synth_batch(4, 64, 256, cfg, gen, k=1, ragged=True) for _ in range(n_micro)]
    ds = _Batches(batches)
    sampler = T.DistributedSampler(ds, num_replicas=1, rank=0, shuffle=False)
    dl = T.DataLoader(ds, batch_size=None, sampler=sampler)
    t0 = time.time()
    avg, gs = T.train_epoch(ddp, dl, loss_fn, opt, sched, conf, epoch=1, global_step=0, device=torch.device("cpu"),
                            tb_logger=None)
    wall = time.time() - t0
    arrs = {"losses": np.array([r[0] for r in rec], dtype=np.float64)}
    names, upd_norm, init_norm = [], [], []
    for n, p in model.named_parameters():
        u = p.detach().double() - init[n].double()
        names.append(n)
        upd_norm.append(float(u.norm()))
        init_norm.append(float(init[n].double().norm()))
    probes = ["model.model.embeddings.norm.weight", "model.model.layers.0.attn.Wqkv.weight",
              "model.model.layers.10.mlp.Wi.weight", "model.model.layers.21.mlp.Wo.weight", "model.head.dense.weight",
              "model.decoder.bias", "model.model.final_norm.weight"]
    live = dict(model.named_parameters())
    for n in probes:
        u = (live[n].detach() - init[n])
        arrs["uprobe::" + n] = np_(u[:8, :64] if u.dim() == 2 else u[:512])
    np.savez_compressed(os.path.join(OUT, "g10_config1_train_epoch.npz"), **arrs)
    json.dump({"conf": {"lr": conf.training.learning_rate, "wd": conf.training.weight_decay,
                        "clip": conf.training.gradient_clip, "accum": accum, "warmup": warm, "total_steps": total_steps,
                        "lambda_q": conf.loss.lambda_q, "lambda_d": conf.loss.lambda_d, "temperature": conf.loss.temperature,
                        "flops_warmup_steps": conf.loss.flops_warmup_steps,
                        "lambda_initial_ratio": conf.loss.lambda_initial_ratio, "batch": 4, "q_len": 64, "d_len": 256,
                        "n_micro": n_micro, "batch_seed": 42, "init_seed": 42},
               "avg_loss": avg, "global_step": gs, "dicts": [r[1] for r in rec], "param_names": names,
               "update_norms": upd_norm, "init_norms": init_norm, "wall_s_cpu_fp32_8_threads": wall,
               "transformers": __import__("transformers").__version__},
              open(os.path.join(OUT, "g10_config1_train_epoch.json"), "w"), indent=1)
    print("g10 global_step", gs, "avg", avg, "first losses", [r[0] for r in rec[:4]], "wall", wall)
    dist.destroy_process_group()


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["g1", "g4", "g5", "g6", "g2", "g3"]
    for w in which:
        globals()[w]()
