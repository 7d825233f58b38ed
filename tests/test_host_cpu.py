"""CPU-only checks: the C ABI (library loads, exports every symbol of include/snx.h, argument
validation without touching a GPU), the host-side mirror of the reference interface, and the
multi-process (gloo, world_size 2) collectives used by the native data-parallel path."""
import json
import math
import os
import re
import socket

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------- C ABI
def _header_functions():
    txt = open(os.path.join(ROOT, "include", "snx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(snx_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_header_symbol():
    import snx
    from snx import _lib
    names = _header_functions()
    assert len(names) >= 30
    lib = snx.lib()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # the ctypes table binds exactly the header's functions
    assert sorted(_lib.SIGNATURES) == [n for n in names if n != "snx_model_desc"]
    assert snx.verify_exports() == []
    assert snx.fn("snx_version")() >= 1


def test_shape_validation_happens_on_the_host():
    """Mis-shaped calls are rejected before any launch (no GPU is needed to see the error code)."""
    import ctypes as C
    from snx import fn
    from snx.encoder import EncoderGeometry
    one = C.c_void_p(16)
    assert fn("snx_gemm_nt_bf16")(one, one, one, 128, 128, 100, None) == -2          # K % 64 != 0
    assert fn("snx_gemm_nt_bf16")(None, one, one, 128, 128, 128, None) == -3
    assert fn("snx_gemm_tn_accum")(one, one, one, 128, 100, 128, None, 0, None) == -2          # N % 128 != 0
    assert fn("snx_ln_fwd")(one, one, one, 4, 100, 1e-5, None) == -2                  # H % 256 != 0
    assert fn("snx_attn_fwd")(one, one, one, one, one, 64, 1, 64, 12, 32, -1, None) == -2   # head_dim != 64
    d = EncoderGeometry().desc()
    assert fn("snx_param_count")(C.byref(d)) == 137
    assert fn("snx_weight_cache_bytes")(C.byref(d)) > 2 * 2 * 110_000_000
    assert fn("snx_model_workspace_bytes")(C.byref(d), 64 * 256, 64, 1) > fn("snx_model_workspace_bytes")(C.byref(d), 64 * 256, 64, 0)
    # SPLADE-head scratch: row maxima for either decoder kernel (128-column tiles or 96-column half tiles) + the
    # 256x192 kernel's row tables (valid-row list, counts, sub-tile table)
    T, V = 36864, 50000
    rows = max((V + 127) // 128, 2 * ((V + 191) // 192))
    assert fn("snx_splade_head_scratch_bytes")(T, V) >= rows * T * 2 + 8 * T + 16 * (T // 32 + T)
    assert fn("snx_splade_head_scratch_bytes")(7, 33) >= 2 * 7 * 2 + 8 * 7 + 16 * 7
    bad = EncoderGeometry(hidden_size=512, num_attention_heads=16).desc()            # head_dim 32
    assert fn("snx_param_count")(C.byref(bad)) == -1
    # grouped weight-gradient GEMM: 1..4 problems, N and K multiples of 128, non-null operands
    from snx.ops import TnProblem
    ok = (TnProblem * 2)(TnProblem(16, 16, 16, 256, 128, 0, 0), TnProblem(16, 16, 16, 128, 128, 1, 0))
    assert fn("snx_gemm_tn_accum_group")(ok, 0, 64, None, 0, None) == -3
    assert fn("snx_gemm_tn_accum_group")(ok, 5, 64, None, 0, None) == -3
    assert fn("snx_gemm_tn_accum_group")(None, 1, 64, None, 0, None) == -3
    odd = (TnProblem * 1)(TnProblem(16, 16, 16, 100, 128, 0, 0))
    assert fn("snx_gemm_tn_accum_group")(odd, 1, 64, None, 0, None) == -2
    nul = (TnProblem * 1)(TnProblem(0, 16, 16, 128, 128, 0, 0))
    assert fn("snx_gemm_tn_accum_group")(nul, 1, 64, None, 0, None) == -3
    # ordered reduction of the weight gradients: a token-split schedule without its workspace is refused on the host
    # (no launch), and the workspace bound covers the layer group of the 149 M model at the bench's token count
    big = (TnProblem * 1)(TnProblem(16, 16, 16, 768, 768, 0, 0))
    need = fn("snx_gemm_tn_workspace_bytes")(big, 1, 36864)
    assert need >= 3 * 9 * 256 * 256 * 4                                             # >= 3 token pieces x 9 tiles
    assert fn("snx_gemm_tn_accum_group")(big, 1, 36864, None, 0, None) == -3
    assert fn("snx_gemm_tn_accum_group")(big, 1, 36864, one, need - 1, None) == -3
    assert fn("snx_gemm_tn_workspace_bytes")(odd, 1, 64) == 0                        # invalid shapes: 0
    assert fn("snx_ln_bwd_workspace_bytes")(36864, 768) == 1024 * 768 * 4            # one partial dw row per block
    assert fn("snx_embed_ln_bwd_workspace_bytes")(36864, 768, 50000) > 36864 * 768 * 4
    assert fn("snx_ln_bwd")(one, one, one, one, None, one, 64, 768, 1e-5, 0, None, 0, None) == -3
    # backward in unit ranges: the range must lie inside [0, layers + 2) and be non-empty
    nine = [one] * 14
    L = d.layers
    for ub, ue, want in ((0, L + 3, -3), (-1, 2, -3), (3, 3, -3), (5, 2, -3)):
        assert fn("snx_model_backward_units")(C.byref(d), *nine[:12], None, 64, 1, 64, ub, ue, None, None) == want, (ub, ue)
    assert fn("snx_model_backward_units")(C.byref(d), None, *nine[:11], None, 64, 1, 64, 0, L + 2, None, None) == -3
    # a pass of a micro-step arena: the row / sequence range must lie inside the plan, a true sub-range takes no groups and
    # must save for backward (host checks, no launch)
    fr = fn("snx_model_forward_range")
    ptrs = [one] * 11
    for tp, npl, r0, s0, t, ns, flags, want in ((576, 3, 512, 2, 128, 1, 1, -3),     # rows past the plan
                                                (576, 3, 0, 3, 64, 1, 1, -3),         # sequences past the plan
                                                (576, 3, -1, 0, 64, 1, 1, -3),
                                                (576, 3, 64, 1, 256, 1, 0, -3)):      # sub-range without save
        assert fr(C.byref(d), *ptrs, None, tp, npl, r0, s0, t, ns, 256, flags, None) == want, (tp, npl, r0, s0, t, ns)
    grp = (C.c_int32 * 4)(1, 0, 1, 64)
    assert fr(C.byref(d), *ptrs, grp, 576, 3, 64, 1, 256, 1, 256, 1, None) == -3     # groups on a sub-range
    br = fn("snx_model_backward_units_range")
    assert br(C.byref(d), *nine[:12], None, 64, 1, 128, 1, 64, 0, L + 2, None, None) == -3   # more rows than planned


def test_product_path_has_no_cpu_fallback():
    from snx.encoder import EncoderGeometry
    from src.model.losses import SPLADELossV33
    from src.model.splade_modern import SPLADEModernBERT
    g = dict(vocab_size=300, hidden_size=256, intermediate_size=128, num_hidden_layers=1, num_attention_heads=4,
             pad_token_id=299)
    m = SPLADEModernBERT(config=g)
    ids = torch.zeros(2, 8, dtype=torch.long)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(ids, torch.ones_like(ids))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        SPLADELossV33()(torch.rand(2, 10), torch.rand(2, 10), torch.rand(2, 10))
    with pytest.raises(ValueError):
        EncoderGeometry(hidden_size=512, num_attention_heads=16).check_supported()


# ---------------------------------------------------------------------------------- host mirror
def test_model_parameter_contract_full_geometry():
    """138 state-dict keys / 137 parameters / 149,372,240 elements, reference names."""
    from oracle import splade_oracle as O
    from src.model.splade_modern import SPLADEModernBERT
    import logging
    logging.getLogger("src.model.splade_modern").setLevel(logging.ERROR)
    m = SPLADEModernBERT()
    names = [n for n, _ in m.named_parameters()]
    assert names == O.param_names(O.EncoderConfig())
    assert sum(p.numel() for p in m.parameters()) == 149_372_240
    sd = m.state_dict()
    assert len(sd) == 138 and "model.decoder.weight" in sd
    assert m.model.decoder.weight is m.model.model.embeddings.tok_embeddings.weight
    shapes = O.param_shapes(O.EncoderConfig())
    assert all(tuple(sd[k].shape) == shapes[k] for k in names)
    # optimizer grouping quirk: only decoder.bias escapes weight decay
    from src.train.config.v33 import V33Config
    from src.train.core.ddp_trainer import build_optimizer
    opt = build_optimizer(m, V33Config())
    assert [len(g["params"]) for g in opt.param_groups] == [136, 1]
    assert opt.param_groups[1]["weight_decay"] == 0.0 and opt.param_groups[0]["weight_decay"] == 0.01
    assert m.vocab_size == 50000 and m.hidden_size == 768


def test_hf_export_roundtrip(tmp_path):
    from src.model.splade_modern import SPLADEModernBERT
    g = dict(vocab_size=300, hidden_size=256, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
             pad_token_id=299)
    m = SPLADEModernBERT(config=g)
    m.model.save_pretrained(str(tmp_path))
    cfg = json.load(open(tmp_path / "config.json"))
    assert cfg["model_type"] == "modernbert" and cfg["vocab_size"] == 300
    m2 = SPLADEModernBERT(model_name=str(tmp_path))
    for (n1, p1), (n2, p2) in zip(m.named_parameters(), m2.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2)


def test_lambda_and_lr_schedules_match_oracle():
    from oracle import splade_oracle as O
    from src.model.losses import SPLADELossV33
    from src.train.core.ddp_trainer import cosine_with_warmup_lambda
    lf = SPLADELossV33(lambda_q=0.01, lambda_d=0.003, flops_warmup_steps=200, lambda_initial_ratio=0.1)
    assert lf.lambda_neg == 0.003                                      # falls back to lambda_d
    for step in (0, 1, 57, 100, 199, 200, 5000):
        assert lf._lambda_schedule(step, 0.01) == O.lambda_schedule(step, 0.01, 200, 0.1)
    for s in range(0, 120, 7):
        assert cosine_with_warmup_lambda(s, 10, 100) == O.cosine_lr_factor(s, 10, 100)
    assert SPLADELossV33(lambda_neg=0.02).lambda_neg == 0.02


def test_config_yaml_sections_and_cli_overrides(tmp_path):
    import argparse
    import yaml
    from src.train.cli.train_v33_ddp import load_config
    from src.train.config.v33 import V33Config
    c = V33Config()
    assert (c.data.batch_size, c.data.query_max_length, c.data.doc_max_length) == (64, 64, 256)
    assert (c.training.gradient_accumulation_steps, c.training.learning_rate, c.loss.flops_warmup_steps) == (4, 5e-5, 20000)
    p = tmp_path / "cfg.yaml"
    p.write_text(yaml.safe_dump({"loss": {"lambda_q": 0.02, "lambda_margin_mse": 0.5}, "data": {"batch_size": 16},
                                 "training": {"num_epochs": 3}}))
    ns = argparse.Namespace(config=str(p), epochs=None, batch_size=8, lr=1e-4, output_dir=None, lambda_q=None,
                            lambda_d=0.001, grad_accum=2, seed=7)
    cfg = load_config(ns)
    assert cfg.loss.lambda_q == 0.02 and cfg.loss.lambda_margin_mse == 0.5 and cfg.loss.lambda_d == 0.001
    assert cfg.data.batch_size == 8 and cfg.training.num_epochs == 3 and cfg.training.learning_rate == 1e-4
    assert cfg.training.gradient_accumulation_steps == 2 and cfg.training.seed == 7
    assert V33Config(loss={"lambda_q": 1.0}).loss.lambda_q == 1.0      # dict sections are promoted


def test_jsonl_dataset_tokenizer_and_collator(tmp_path):
    from src.train.data import SyntheticTripletDataset, TripletCollator, load_training_data
    from src.train.data.collator import HashTokenizer, create_tokenizer
    f1, f2 = tmp_path / "train_0.jsonl", tmp_path / "train_1.jsonl"
    f1.write_text(json.dumps({"query": "a b", "positive": "c d e", "negative": "f"}) + "\n\n" +
                  json.dumps({"query": "g", "positive": "h i", "negative": None}) + "\n")
    f2.write_text(json.dumps({"query": "j k l", "positive": "m", "negative": "n o"}) + "\n")
    ds = load_training_data([str(tmp_path / "train_*.jsonl")])
    assert len(ds) == 3 and ds[2]["query"] == "j k l" and ds[1]["negative"] is None
    with pytest.raises(FileNotFoundError):
        load_training_data([str(tmp_path / "nope_*.jsonl")])
    tok = create_tokenizer("hash:1000")
    assert isinstance(tok, HashTokenizer) and tok.pad_token_id == 999
    with pytest.raises(FileNotFoundError):
        create_tokenizer("skt/A.X-Encoder-base")
    col = TripletCollator(tok, query_max_length=4, doc_max_length=6)
    out = col([ds[0], ds[1], ds[2]])
    assert out["query_input_ids"].shape == (3, 4) and out["num_negatives"] == 1
    assert out["negative_input_ids"][1].tolist() == out["positive_input_ids"][1].tolist()[:out["negative_input_ids"].shape[1]]
    assert int(out["query_attention_mask"][1].sum()) == 3                     # <s> g </s>
    syn = load_training_data(["synthetic:10:3"])
    assert isinstance(syn, SyntheticTripletDataset) and len(syn[4]["negatives"]) == 3 and syn[4] == syn[4]
    tok.save_pretrained(str(tmp_path / "tok"))
    assert create_tokenizer(str(tmp_path / "tok")).vocab_size == 1000


def test_checkpoint_layout_and_resume(tmp_path):
    """checkpoint_epoch{E}_step{S}/{model.pt, training_state.pt, config.json} (ref:train_v33_ddp.py:192-286)."""
    from src.model.splade_modern import SPLADEModernBERT
    from src.train.config.v33 import V33Config
    from src.train.core import ddp_trainer as T
    g = dict(vocab_size=300, hidden_size=256, intermediate_size=128, num_hidden_layers=1, num_attention_heads=4,
             pad_token_id=299)
    m = SPLADEModernBERT(config=g)
    cfg = V33Config()
    opt = T.build_optimizer(m, cfg)
    sch = T.build_scheduler(opt, 2, 10)
    for p in m.parameters():
        p.grad = torch.ones_like(p) * 1e-3
    opt.step(); sch.step()
    path = T.save_checkpoint(m, opt, sch, epoch=2, global_step=30, output_dir=str(tmp_path), config=cfg, best_metric=0.5)
    T.save_checkpoint(m, opt, sch, epoch=1, global_step=7, output_dir=str(tmp_path), config=cfg)
    assert sorted(os.listdir(path)) == ["config.json", "model.pt", "training_state.pt"]
    assert T.find_latest_checkpoint(str(tmp_path)).endswith("checkpoint_epoch2_step30")
    assert T.find_latest_checkpoint(str(tmp_path / "missing")) is None
    m2 = SPLADEModernBERT(config=g)
    opt2 = T.build_optimizer(m2, cfg)
    sch2 = T.build_scheduler(opt2, 2, 10)
    st = T.load_checkpoint(m2, opt2, sch2, path)
    assert (st["epoch"], st["global_step"], st["best_metric"]) == (2, 30, 0.5)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    assert sch2.last_epoch == 1
    os.remove(os.path.join(path, "training_state.pt"))
    assert T.load_checkpoint(m2, None, None, path) == {"epoch": -1, "global_step": 0}


def test_tensorboard_logger_does_not_need_tensorboard(tmp_path):
    from src.train.utils import TensorBoardLogger, setup_logging
    tb = TensorBoardLogger(log_dir=str(tmp_path), experiment_name="x")
    tb.log_scalar("train/loss", torch.tensor(1.5), 3)
    tb.close()
    setup_logging(output_dir=str(tmp_path), log_file="t.log").info("hello")
    assert (tmp_path / "t.log").exists()


def test_get_top_k_tokens_tie_order():
    from src.model.splade_modern import SPLADEModernBERT
    g = dict(vocab_size=300, hidden_size=256, intermediate_size=128, num_hidden_layers=1, num_attention_heads=4,
             pad_token_id=299)
    m = SPLADEModernBERT(config=g)

    class Tok:
        def decode(self, ids):
            return f" t{ids[0]} "
    v = torch.zeros(300)
    v[[5, 17, 200]] = torch.tensor([2.0, 3.0, 2.0])
    out = m.get_top_k_tokens(v, Tok(), k=10)
    assert list(out.items()) == [("t17", 3.0), ("t5", 2.0), ("t200", 2.0)]     # tie: lowest index first, zeros dropped


# ---------------------------------------------------------------------------------- multi-process (gloo)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dist_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import splade_oracle as O
    from snx import dist as sd
    torch.manual_seed(0)
    # (1) flat-gradient all-reduce == DDP's mean over ranks
    flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
    sd.allreduce_flat_grads(flat)
    ok1 = torch.allclose(flat, torch.arange(10, dtype=torch.float32) * (sum(range(1, world + 1)) / world))
    # (2) all-gather with reduce-scatter backward: cross-rank in-batch negatives identity (config 4)
    g = torch.Generator().manual_seed(123)
    B, V = 3, 40
    A = [torch.rand(B, V, generator=g) for _ in range(world)]
    P = [torch.rand(B, V, generator=g) for _ in range(world)]
    N = [torch.rand(B, V, generator=g) for _ in range(world)]
    a, p, n = (t[rank].clone().requires_grad_(True) for t in (A, P, N))
    pall = sd.all_gather_with_grad(p)
    loss = O.cross_rank_infonce(a, pall, n, rank, 1.5)
    loss.backward()
    # single-process reference: sum over ranks of each rank's loss, differentiate w.r.t. every P
    Ps = [t.clone().requires_grad_(True) for t in P]
    tot = 0
    for r in range(world):
        tot = tot + O.cross_rank_infonce(A[r], torch.cat(Ps, 0), N[r], r, 1.5)
    tot.backward()
    ok2 = torch.allclose(p.grad, Ps[rank].grad, atol=1e-6) and pall.shape == (world * B, V)
    # ... and the exchange-stream form the trainer uses (round 6): on CPU tensors it degrades to the inline collective and
    # must hand out the same tensor and the same gradient through PendingGather.wait()
    p2 = P[rank].clone().requires_grad_(True)
    pend = sd.all_gather_with_grad_async(p2)
    pall2 = pend.wait()
    O.cross_rank_infonce(A[rank], pall2, N[rank], rank, 1.5).backward()
    ok2 = ok2 and torch.equal(pall2.detach(), pall.detach()) and torch.equal(p2.grad, p.grad) and pend.wait() is pall2
    from src.train.core import ddp_trainer as T
    ok3 = T.is_main_process() == (rank == 0)
    # (3) the bucketed exchange that overlaps the backward: drive BucketedGradSync exactly as
    # EncoderRuntime.backward_impl does (unit ranges -> finished flat slices -> one all-reduce each) on a flat
    # gradient laid out like a 6-layer model's; every element must come out as the mean over ranks, reduced once
    from snx.dist import BucketedGradSync
    from snx.encoder import EncoderGeometry, EncoderRuntime
    geom = EncoderGeometry(num_hidden_layers=6, vocab_size=100, hidden_size=256, intermediate_size=128, pad_token_id=99)
    H, I, V, L = 256, 128, 100, 6
    sizes = [V * H, H]
    for l in range(L):
        sizes += ([H] if l else []) + [3 * H * H, H * H, H, 2 * I * H, H * I]
    sizes += [H, H * H, H, V]
    rt = EncoderRuntime.__new__(EncoderRuntime)
    rt.geom = geom
    rt.params = [torch.empty(n) for n in sizes]
    total = sum(sizes)
    flat = torch.arange(total, dtype=torch.float32) * (rank + 1)
    gs = BucketedGradSync(torch.device("cpu"), 3)
    gs.arm(True)
    ok4 = gs.armed
    for ub, ue in gs.unit_ranges(L + 2):
        for lo, hi in rt.unit_param_range(ub, ue):
            gs.reduce_slice(flat, lo, hi)
    gs.finished_backward()
    cov = sorted(gs.slices)
    ok4 = ok4 and cov[0][0] == 0 and cov[-1][1] == total and all(a[1] == b[0] for a, b in zip(cov, cov[1:]))
    ok4 = ok4 and gs.wait(flat) and not gs.wait(flat)
    ok4 = ok4 and torch.allclose(flat, torch.arange(total, dtype=torch.float32) * (sum(range(1, world + 1)) / world))
    # (4) a micro-step of THREE backward calls (the reference's call pattern, ref:train_v33_ddp.py:339-343 -> one
    # reduction; here SNX_FUSED_PASSES=0 or a custom loop): only the backward that returns the LAST outstanding
    # forward token may exchange, after it has written.  The stand-in for the kernels adds rank- and call-dependent
    # gradients unit range by unit range (the tied embedding matrix gets a share in unit 0 and the rest in the last
    # unit, like the real backward), through the same BucketedGradSync.run_backward the runtime uses.
    E = (0, sizes[0])
    base = torch.arange(total, dtype=torch.float32) + 1.0

    def contribution(k):
        return base * float((rank + 1) * (k + 1))

    def drive(gs, flat, k, token):
        c = contribution(k)

        def run_all():
            flat.add_(c)

        def run_units(ub, ue):
            if ub == 0:
                flat[E[0]:E[1]].add_(0.25 * c[E[0]:E[1]])
            for lo, hi in rt.unit_param_range(ub, ue):
                if lo == 0:                                  # the slice that starts with the embedding matrix
                    flat[E[0]:E[1]].add_(0.75 * c[E[0]:E[1]])
                    lo = E[1]
                flat[lo:hi].add_(c[lo:hi])
        gs.run_backward(token, flat, L + 2, run_all, run_units, rt.unit_param_range)

    ok5 = True
    mean_rank = sum(range(1, world + 1)) / world
    for mode in ("allreduce", "rs_ag"):
        gs = BucketedGradSync(torch.device("cpu"), 3, mode=mode)
        gs.keep_log = True
        flat = torch.zeros(total)
        for last in (False, True):                           # accumulation window of two micro-steps
            gs.arm(last)
            toks = [gs.on_forward() for _ in range(3)]       # query, positive, negative forwards
            for k in (2, 1, 0):                              # autograd runs the backwards in reverse order
                drive(gs, flat, k, toks[k])
        ok5 = ok5 and gs.pending and gs.wait(flat)
        want = base * (2 * (1 + 2 + 3) * mean_rank)
        ok5 = ok5 and torch.allclose(flat, want, rtol=1e-6)
        ev = [e for e in gs.log if e[0] in ("bwd", "reduce", "exchange")]
        last_bwd = max(i for i, e in enumerate(ev) if e[0] == "bwd")
        first_red = min(i for i, e in enumerate(ev) if e[0] == "reduce")
        ok5 = ok5 and ev[last_bwd][2] is True and first_red > last_bwd
        ok5 = ok5 and sum(1 for e in ev if e[0] == "bwd" and e[2]) == 1 and sum(1 for e in ev if e[0] == "exchange") == 1
        cov = sorted((e[1], e[2]) for e in ev if e[0] == "reduce")
        ok5 = ok5 and cov[0][0] == 0 and cov[-1][1] == total and all(a[1] == b[0] for a, b in zip(cov, cov[1:]))
    # (5) misuse is loud or safe: a graph built BEFORE arm() disarms the overlap (the caller's sync_gradients() then
    # reduces the whole buffer); a backward between a finished exchange and wait() raises
    gs = BucketedGradSync(torch.device("cpu"), 3)
    flat = torch.zeros(total)
    stale = gs.on_forward()                                  # not armed: no token
    gs.arm(True)
    drive(gs, flat, 0, stale)
    ok6 = stale is None and not gs.armed and not gs.pending and not gs.wait(flat)
    ok6 = ok6 and torch.equal(flat, contribution(0))         # nothing was reduced
    gs.arm(True)
    drive(gs, flat, 0, gs.on_forward())
    try:
        drive(gs, flat, 0, None)
        ok6 = False
    except RuntimeError:
        pass
    try:
        gs.arm(True)
        ok6 = False
    except RuntimeError:
        pass
    ok6 = ok6 and gs.wait(flat)
    ret[rank] = (bool(ok1), bool(ok2), bool(ok3), bool(ok4), bool(ok5), bool(ok6))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_collectives_gloo():
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dist_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: (True,) * 6, 1: (True,) * 6}


def test_bench_never_prints_a_one_gpu_line_for_a_multi_gpu_request():
    """bench.py --gpus N (N > 1): without RANK it must launch N workers itself or fail; with a launcher whose WORLD_SIZE
    differs from N it must fail.  Both refusals happen before any GPU call, so they are checked here without one."""
    import subprocess
    import sys
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SNX_BENCH_BACKEND")}
    env["HIP_VISIBLE_DEVICES"] = ""                      # no GPU of its own, whatever the host has
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "2"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0 and '"n_gpus"' not in r.stdout and "GPU(s) visible" in r.stderr
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "2"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0 and '"n_gpus"' not in r.stdout and "WORLD_SIZE=1" in r.stderr
