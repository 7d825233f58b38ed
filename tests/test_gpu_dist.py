"""Multi-GPU readiness proven on ONE GPU (SURVEY 8(e), rows a15 / e): the collectives of the data-parallel
path go through RCCL with a world-size-1 `nccl` process group.

  * torch `DistributedDataParallel(device_ids=[0], broadcast_buffers=False, find_unused_parameters=False)`
    around the model -- the literal drop-in of ref:src/train/cli/train_v33_ddp.py:539-544 -- through
    `train_epoch` (3 forwards / 1 backward per micro-step, accumulate 4) vs the oracle loop;
  * `NativeDataParallel` with SNX_DIST_FORCE=1, which removes the world()==1 early-outs so that the rank-0
    broadcast, the bucketed all_reduce(AVG) overlapped with the backward, all_gather_into_tensor and
    reduce_scatter_tensor really reach RCCL.
Scaling itself needs more than one GPU and is measured by the driver, not here."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def nccl_world1(dev):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    yield
    dist.destroy_process_group()


def test_torch_ddp_wrapped_train_epoch(dev, nccl_world1):
    from tests.test_gpu_model import train_epoch_case
    train_epoch_case(dev, "ddp")


def test_native_data_parallel_collectives_reach_rccl(dev, nccl_world1, monkeypatch):
    """Same micro-steps twice: without collectives (plain world-1 early-outs) and with SNX_DIST_FORCE=1 (rank-0
    broadcast at wrap, bucketed all-reduce(AVG) during the last backward of the window, all-gather +
    reduce-scatter of the positives for cross-GPU negatives).  At world size 1 every collective is the
    identity and every weight gradient is reduced in a fixed order (include/snx.h "det_reduce"), so losses and
    parameters must be BIT-equal -- through both optimizer steps --; the bucket slices must tile the flat gradient
    exactly once."""
    from oracle import splade_oracle as O
    from snx import dist as sdist
    from src.model.losses import SPLADELossV33
    from src.train.config.v33 import V33Config
    from src.train.core import ddp_trainer as T
    from tests.test_gpu_model import _build_model, _small_cfg
    cfg = _small_cfg()
    params = O.perturb_params(O.init_params(cfg, seed=3), seed=4, scale=2.0, bias_mean=-0.1)
    gen = torch.Generator().manual_seed(77)
    batches = [O.synth_batch(4, 24, 70, cfg, gen, k=2, ragged=True) for _ in range(4)]
    conf = V33Config()
    conf.training.gradient_accumulation_steps = 2
    conf.training.learning_rate = 1e-3

    def run(force):
        if force:
            monkeypatch.setenv("SNX_DIST_FORCE", "1")
        else:
            monkeypatch.delenv("SNX_DIST_FORCE", raising=False)
        assert sdist.active() == force
        model = T.NativeDataParallel(_build_model(cfg, params, dev), n_buckets=3)
        loss_fn = SPLADELossV33(temperature=20.0, flops_warmup_steps=4).to(dev)
        opt = T.build_optimizer(model, conf)
        sch = T.build_scheduler(opt, 0, 4)
        losses, slices = [], []
        gs = 0
        for i, b in enumerate(batches):
            last = (i + 1) % 2 == 0
            loss, _ = T.micro_step(model, loss_fn, b, gs, dev, 2, cross_gpu_negatives=True, last_of_window=last)
            losses.append(float(loss))
            if last:
                slices.append(list(model.module.runtime.grad_sync.slices))
                T.optimizer_step(model, opt, sch, conf)
                gs += 1
        torch.cuda.synchronize()
        return losses, slices, {n: p.detach().clone() for n, p in model.module.named_parameters()}, model

    l0, s0, p0, _ = run(False)
    l1, s1, p1, m1 = run(True)
    assert l0 == l1                                          # bit-equal, before and after the optimizer steps
    assert s0 == [[], []]                                    # nothing exchanged without a forced group
    total = m1.module.runtime.flat_grad.numel()
    for sl in s1:                                            # forced: 3 buckets tile the flat gradient exactly once
        cov = sorted(sl)
        assert cov[0][0] == 0 and cov[-1][1] == total and all(a[1] == b[0] for a, b in zip(cov, cov[1:])), cov
        assert 3 <= len(sl) <= 5
    for n in p0:
        assert torch.equal(p0[n], p1[n]), (n, float((p0[n] - p1[n]).abs().max()))


@pytest.mark.parametrize("mode", ["allreduce", "rs_ag"])
def test_three_backward_micro_step_exchanges_once_after_the_last_backward(dev, nccl_world1, monkeypatch, mode):
    """SNX_FUSED_PASSES=0: query / positive / negative run as three native forwards and three native backwards per
    micro-step (the reference's call pattern, ref:train_v33_ddp.py:339-343,364).  Every backward adds into the same
    flat gradient, so the overlapped exchange must run inside the LAST of the three and every slice must be reduced
    exactly once, after the last backward call that writes it (round-2 review: it ran inside the first).  Also drives
    the reduce-scatter + all-gather form of the bucket exchange through RCCL at world size 1.

    Values.  The fused pass and the three passes are the same sum over the same token rows cut at other places (one
    weight-gradient launch over q + p + n rows against three launches that accumulate): another fp32 summation TREE, so
    the accumulated gradients agree to reassociation error only -- gamma_n * sum |terms| per element (n ~ 1e3 rows:
    gamma ~ 1e-4 of the term magnitudes, measured relative L2 ~ 1e-6) -- and are compared here as a GROSS-ERROR screen,
    relative L2 <= 1e-3 per tensor (a backward that lost or doubled a pass is off by O(1)).  Each of the two runs is
    bit-reproducible by itself (asserted: the fused run twice).  Parameters after Adam are NOT compared tightly: an Adam
    step is lr * sign-like, so elements whose gradient is near zero take a different step under any reassociation;
    only the rigorous cap holds (two steps of at most lr each, per run).  Round 4 bounded the mean of that difference
    by a number read off sample runs, which went red; the summation order itself is now pinned instead."""
    from oracle import splade_oracle as O
    from snx import dist as sdist
    from snx._lib import fn
    from src.model.losses import SPLADELossV33
    from src.train.config.v33 import V33Config
    from src.train.core import ddp_trainer as T
    from tests.test_gpu_model import _build_model, _small_cfg
    cfg = _small_cfg()
    params = O.perturb_params(O.init_params(cfg, seed=3), seed=4, scale=2.0, bias_mean=-0.1)
    gen = torch.Generator().manual_seed(78)
    batches = [O.synth_batch(4, 24, 70, cfg, gen, k=1, ragged=True) for _ in range(4)]
    conf = V33Config()
    conf.training.gradient_accumulation_steps = 2
    conf.training.learning_rate = 1e-3
    monkeypatch.setenv("SNX_DIST_FORCE", "1")
    monkeypatch.setenv("SNX_GRAD_EXCHANGE", mode)
    monkeypatch.setenv("SNX_PACK", "0")                     # same (padded) execution in both runs

    def run(fused):
        monkeypatch.setenv("SNX_FUSED_PASSES", "1" if fused else "0")
        model = T.NativeDataParallel(_build_model(cfg, params, dev), n_buckets=3)
        gs_ = model.module.runtime.grad_sync
        assert gs_.mode == mode
        gs_.keep_log = True
        loss_fn = SPLADELossV33(temperature=20.0, flops_warmup_steps=4).to(dev)
        opt = T.build_optimizer(model, conf)
        sch = T.build_scheduler(opt, 0, 4)
        step = 0
        first = None
        for i, b in enumerate(batches):
            last = (i + 1) % 2 == 0
            T.micro_step(model, loss_fn, b, step, dev, 2, last_of_window=last)
            if last:
                if first is None:
                    model.sync_gradients()
                    first = model.module.runtime.flat_grad.detach().clone()     # the first window's accumulated gradient
                T.optimizer_step(model, opt, sch, conf)
                step += 1
        torch.cuda.synchronize()
        assert fn("snx_get_reserved_cus")() == 0            # no reservation outside an armed backward
        model.first_window_grad = first
        return {n: p.detach().clone() for n, p in model.module.named_parameters()}, gs_, model

    p_fused, g_fused, m_f = run(True)
    p_again, _, m_a = run(True)
    assert torch.equal(m_f.first_window_grad, m_a.first_window_grad)              # bit-reproducible ...
    # ... and independent of what ran earlier in the process: the digest is recorded so that a run of this test alone and a
    # run inside the whole suite can be compared (round-4 review: "3.2e-5 in the suite, 6.1e-6 alone" smelt of carried state;
    # profiles/r05_three_backward_digest.txt holds both)
    import hashlib
    digest = hashlib.sha256(m_f.first_window_grad.cpu().numpy().tobytes()).hexdigest()
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "three_backward_digest.txt"), "a") as fh:
            fh.write(f"{mode} {digest} pid={os.getpid()} tests_in_process={os.environ.get('PYTEST_CURRENT_TEST', '')}\n")
    assert all(torch.equal(p_fused[n], p_again[n]) for n in p_fused)              # ... through the optimizer steps
    p_three, g_three, m = run(False)
    total = m.module.runtime.flat_grad.numel()
    for gs_, nfwd in ((g_fused, 1), (g_three, 3)):
        # split the log into armed micro-steps (one epoch each)
        epochs = sorted({e[1] for e in gs_.log if e[0] == "fwd"})
        assert len(epochs) == 2                              # two armed micro-steps (the last of each window)
        for ep in epochs:
            ev = [e for e in gs_.log if (e[0] in ("fwd", "bwd", "exchange") and e[1] == ep) or e[0] == "reduce"]
            i_ex = next(i for i, e in enumerate(gs_.log) if e[0] == "exchange" and e[1] == ep)
            before = [e for e in gs_.log[:i_ex] if e[0] in ("fwd", "bwd") and e[1] == ep]
            assert [e[0] for e in before] == ["fwd"] * nfwd + ["bwd"] * nfwd, before
            assert [e[2] for e in before if e[0] == "bwd"] == [False] * (nfwd - 1) + [True]
            reds = []
            for e in gs_.log[i_ex + 1:]:
                if e[0] != "reduce":
                    break
                reds.append((e[1], e[2]))
            cov = sorted(reds)
            assert cov[0][0] == 0 and cov[-1][1] == total and all(a[1] == b[0] for a, b in zip(cov, cov[1:])), cov
            del ev
    off = 0
    for i, p_ in enumerate(m.module.runtime.params):         # the flat buffer's own order
        n = p_.numel()
        a, b = m_f.first_window_grad[off:off + n].double(), m.first_window_grad[off:off + n].double()
        off += n
        assert float((a - b).norm()) <= 1e-3 * float(b.norm()) + 1e-30, (i, float((a - b).norm()), float(b.norm()))
    for n in p_fused:                                        # two Adam steps of at most lr = 1e-3 each, per run
        assert float((p_fused[n] - p_three[n]).abs().max()) <= 2 * 2 * 1e-3 * 1.01, n


@pytest.mark.parametrize("mode,reserved", [("allreduce", 0), ("rs_ag", 0), ("allreduce", 32)])
def test_overlapped_exchange_beside_the_persistent_kernels_through_rccl(dev, nccl_world1, monkeypatch, mode, reserved):
    """One accumulation window of a geometry whose backward runs on the PERSISTENT kernels (256x256 weight-gradient GEMM,
    the 256x256 NT kernel for the wide fused Linears, 256x192 decoder: thresholds lowered through snx_configure so that
    2,176 token rows qualify), with the
    gradient exchange overlapped with the last backward -- buckets through RCCL on the exchange stream while those
    whole-CU kernels are in flight -- against the un-overlapped path (no buckets: one all-reduce of the whole buffer
    after the backward).  At world size 1 every collective is the identity and the weight gradients are reduced in a
    fixed order (include/snx.h "det_reduce"), so with the default launch (reserved = 0: the same 256-workgroup
    schedules in both runs) the two accumulated gradients must be BIT-equal; a bucket reduced while its last producer
    still writes, or a unit range that skips work, cannot hide behind a tolerance.

    reserved = 32 (opt-in, SNX_EXCHANGE_RESERVED_CUS): the armed backward launches 224 workgroups, and
    csrc/gemm_tn256.hip derives its token pieces from that count -- another fp32 summation tree than the 256-workgroup
    run's (round 4 took this for atomic-order noise and bounded it by a sampled 2e-6: red on the driver's box).  The
    224-workgroup schedules are checked against derived bounds where the operands are known
    (tests/test_gpu_ops.py::test_gemm_tn_256_with_reserved_cus, ..._nt256_with_reserved_cus); here only the plumbing
    is asserted -- the reservation holds inside the armed backward only, the slices tile the buffer once -- plus a
    gross-error screen of the gradients (relative L2 <= 1e-3 per tensor; reassociation is ~1e-6, lost work is O(1)).
    What one GPU cannot show is the contention itself (RCCL at world size 1 moves no data between devices)."""
    import snx
    from oracle import splade_oracle as O
    from snx._lib import fn
    from src.model.losses import SPLADELossV33
    from src.train.core import ddp_trainer as T
    from tests.test_gpu_model import _build_model
    cfg = O.EncoderConfig(vocab_size=3000, hidden_size=768, intermediate_size=1152, num_hidden_layers=4,
                          num_attention_heads=12, local_attention=128, pad_token_id=2999)
    params = O.perturb_params(O.init_params(cfg, seed=5), seed=6, scale=1.5, bias_mean=-0.1)
    gen = torch.Generator().manual_seed(91)
    batches = [O.synth_batch(8, 16, 128, cfg, gen, k=1, ragged=False) for _ in range(2)]     # 8 x (16 + 128 + 128) rows
    monkeypatch.setenv("SNX_DIST_FORCE", "1")
    monkeypatch.setenv("SNX_GRAD_EXCHANGE", mode)
    monkeypatch.setenv("SNX_EXCHANGE_RESERVED_CUS", str(reserved))
    before = {k: snx.config(k) for k in ("tn256_min_m", "nt256_min_m", "dec256_min_t")}
    snx.configure(tn256_min_m=1024, nt256_min_m=1024, dec256_min_t=256)
    try:
        def run(n_buckets):
            model = T.NativeDataParallel(_build_model(cfg, params, dev), n_buckets=n_buckets)
            loss_fn = SPLADELossV33(temperature=20.0, flops_warmup_steps=4).to(dev)
            seen = []
            for i, b in enumerate(batches):
                last = i + 1 == len(batches)
                T.micro_step(model, loss_fn, b, 0, dev, len(batches), last_of_window=last)
                seen.append(fn("snx_get_reserved_cus")())
            model.sync_gradients()
            torch.cuda.synchronize()
            gs_ = model.module.runtime.grad_sync
            return model.module.runtime.flat_grad.detach().clone(), gs_, seen, model
        g_over, gs_, seen, m = run(3)
        assert gs_ is not None and gs_.mode == mode and gs_.reserved_cus == reserved
        assert seen == [0, 0] and len(gs_.slices) == 3      # reserved only inside the armed backward, reset behind it
        cov = sorted(gs_.slices)
        assert cov[0][0] == 0 and cov[-1][1] == g_over.numel() and all(a[1] == b[0] for a, b in zip(cov, cov[1:])), cov
        g_plain, gs0, _, _ = run(0)
        assert gs0 is None
    finally:
        snx.configure(**before)
    assert torch.isfinite(g_over).all() and float(g_over.abs().max()) > 0
    if reserved == 0:
        assert torch.equal(g_over, g_plain), float((g_over - g_plain).abs().max())
        return
    off = 0
    for name, p_ in enumerate(m.module.runtime.params):          # the flat buffer's own order
        n = p_.numel()
        a, b = g_over[off:off + n].double(), g_plain[off:off + n].double()
        off += n
        assert float((a - b).norm()) <= 1e-3 * float(b.norm()) + 1e-30, (name, float((a - b).norm()), float(b.norm()))


def test_bucket_plan_covers_every_unit(dev):
    from snx.dist import BucketedGradSync
    for nb in (1, 2, 4, 8, 22, 40):
        r = BucketedGradSync(dev, nb).unit_ranges(24)
        assert r[0][0] == 0 and r[-1][1] == 24 and all(a[1] == b[0] for a, b in zip(r, r[1:])), (nb, r)
        assert len(r) == min(nb, 22)


@pytest.mark.parametrize("extra", [[], ["--native-dp"], ["--native-dp", "--cross-gpu-negatives"]])
def test_cli_trains_end_to_end_under_torchrun(dev, tmp_path, extra):
    """The drop-in itself: `torchrun -m src.train.cli.train_v33_ddp --config ...` (the reference's launch line,
    ref:scripts/launch_v33_b200.sh:39-45) on synthetic triplets with a small local model directory: process group
    (nccl), DistributedSampler loader + TripletCollator, torch DDP (default) or NativeDataParallel, train_epoch
    with accumulation, checkpoint + final_model written with the reference's file names."""
    import json
    import subprocess
    import sys
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "opensearch-neural-pre-train_amd")
    mdir = tmp_path / "model"
    mdir.mkdir()
    (mdir / "config.json").write_text(json.dumps(dict(
        vocab_size=1000, hidden_size=256, intermediate_size=384, num_hidden_layers=2, num_attention_heads=4,
        local_attention=16, pad_token_id=999)))
    out = tmp_path / "out"
    cfg = {"model": {"name": str(mdir)},
           "loss": {"temperature": 20.0, "flops_warmup_steps": 4, "lambda_margin_mse": 0.1},
           "data": {"train_files": ["synthetic:48:2"], "batch_size": 4, "query_max_length": 16, "doc_max_length": 32,
                    "num_workers": 0},
           "training": {"num_epochs": 1, "gradient_accumulation_steps": 2, "output_dir": str(out),
                        "log_every_n_steps": 1, "save_every_n_epochs": 1, "learning_rate": 1e-3}}
    (tmp_path / "cfg.yaml").write_text(yaml.safe_dump(cfg))
    env = dict(os.environ, PYTHONPATH=pkg + os.pathsep + os.environ.get("PYTHONPATH", ""), SNX_DIST_FORCE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29547", "-m", "src.train.cli.train_v33_ddp",
                        "--config", str(tmp_path / "cfg.yaml"), "--tokenizer", "hash:1000"] + extra,
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    log = (out / "training.log").read_text()
    assert "Training complete" in log and "Step 6 |" in log, log[-2000:]      # 12 micro-batches / accum 2
    assert (out / "final_model" / "model.pt").exists()
    ck = sorted(p.name for p in (out / "checkpoint_epoch1_step6").iterdir())
    assert ck == ["config.json", "model.pt", "training_state.pt"]
    sd = torch.load(out / "final_model" / "model.pt", map_location="cpu", weights_only=True)
    assert "model.decoder.weight" in sd and all(torch.isfinite(v).all() for v in sd.values())


def test_two_ranks_share_one_gpu(dev, tmp_path):
    """World size 2 on the GPU path.  Two processes, both on cuda:0, gloo group (RCCL refuses two ranks on one device;
    snx.dist then moves the buckets through host copies made on the exchange stream): rank 1 starts from other
    weights (rank-0 broadcast), the ranks see different batches, and after the first accumulation window every
    gradient must be the mean of the two ranks' local gradients (computed without any exchange) and bit-equal on
    both ranks; after two optimizer steps the parameters must be bit-equal on both ranks.  Fused pass and the
    reference's three-forward pattern, all_reduce and reduce-scatter + all-gather forms, cross-GPU negatives once.
    See tests/two_rank_gpu_worker.py."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "opensearch-neural-pre-train_amd")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([root, pkg, os.environ.get("PYTHONPATH", "")]),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("SNX_DIST_FORCE", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, "-m", "tests.two_rank_gpu_worker", str(r), str(port), str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=root)
             for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=420))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, (r, so[-1500:], se[-3000:])
    reps = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    assert all(len(rep["cases"]) == 6 for rep in reps)       # 4 exchange cases + the two gather-stream comparisons
    out = os.path.join(root, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_report.jsonl"), "a") as f:
        f.write(json.dumps({"test": "two_ranks_share_one_gpu", "rank0": reps[0]["cases"]}) + "\n")


@pytest.mark.parametrize("extra", [[], ["--native-dp"]])
def test_cli_trains_with_two_ranks_on_one_gpu(dev, tmp_path, extra):
    """The reference's launch line with TWO ranks (`torchrun --nproc-per-node 2 -m src.train.cli.train_v33_ddp`): the
    DistributedSampler really shards (48 triplets -> 24 per rank -> 6 micro-batches -> 3 optimizer steps at accum 2),
    torch DDP's reducer (default) or NativeDataParallel's bucketed exchange really averages over two ranks, barriers and
    the rank-0-only checkpoint / final model run with a second rank present.  One GPU: SNX_DIST_BACKEND=gloo lets the
    ranks share it (the kernels are the same; only the transport differs from a node)."""
    import json
    import subprocess
    import sys
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "opensearch-neural-pre-train_amd")
    mdir = tmp_path / "model"
    mdir.mkdir()
    (mdir / "config.json").write_text(json.dumps(dict(
        vocab_size=1000, hidden_size=256, intermediate_size=384, num_hidden_layers=2, num_attention_heads=4,
        local_attention=16, pad_token_id=999)))
    out = tmp_path / "out"
    cfg = {"model": {"name": str(mdir)},
           "loss": {"temperature": 20.0, "flops_warmup_steps": 4, "lambda_margin_mse": 0.1},
           "data": {"train_files": ["synthetic:48:2"], "batch_size": 4, "query_max_length": 16, "doc_max_length": 32,
                    "num_workers": 0},
           "training": {"num_epochs": 1, "gradient_accumulation_steps": 2, "output_dir": str(out),
                        "log_every_n_steps": 1, "save_every_n_epochs": 1, "learning_rate": 1e-3}}
    (tmp_path / "cfg.yaml").write_text(yaml.safe_dump(cfg))
    env = dict(os.environ, PYTHONPATH=pkg + os.pathsep + os.environ.get("PYTHONPATH", ""), SNX_DIST_BACKEND="gloo",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SNX_DIST_FORCE", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29549", "-m", "src.train.cli.train_v33_ddp",
                        "--config", str(tmp_path / "cfg.yaml"), "--tokenizer", "hash:1000"] + extra,
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    log = (out / "training.log").read_text()
    assert "Training complete" in log and "Step 3 |" in log and "Step 4 |" not in log, log[-2000:]
    assert (out / "final_model" / "model.pt").exists()
    ck = sorted(p.name for p in (out / "checkpoint_epoch1_step3").iterdir())
    assert ck == ["config.json", "model.pt", "training_state.pt"]
    sd = torch.load(out / "final_model" / "model.pt", map_location="cpu", weights_only=True)
    assert all(torch.isfinite(v).all() for v in sd.values())


@pytest.mark.parametrize("world,wrapper", [(1, "ddp"), (2, "ddp"), (2, "native")])
def test_reference_train_epoch_replayed_on_one_gpu(dev, tmp_path, world, wrapper):
    """Rows a14 / a15 / e against the REFERENCE itself, at world size 1 and 2: goldens g2_train_epoch_w1 / _w2 (the
    reference's unmodified train_epoch on one / two gloo DDP ranks) replayed by that many ranks sharing this GPU -- this
    repo's train_epoch, torch DDP around the model (or NativeDataParallel: one exchange per optimizer step), fp32 kernels.  Per-micro-step losses of rank 0 within 2e-4, every parameter's update
    after the two optimizer steps cos >= 0.995 with the reference's, >= 97 % of all elements within 5e-5, all ranks
    bit-equal.  See tests/two_rank_gpu_worker.py::reference_replay."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "opensearch-neural-pre-train_amd")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([root, pkg, os.environ.get("PYTHONPATH", "")]),
               HSA_ENABLE_IPC_MODE_LEGACY="0", REPLAY_WRAPPER=wrapper)
    for k in ("SNX_DIST_FORCE", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, "-m", "tests.two_rank_gpu_worker", str(r), str(port), str(tmp_path),
                               f"reference_w{world}"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env,
                              cwd=root)
             for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=420))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, (r, so[-1500:], se[-3000:])
    rep = json.load(open(tmp_path / f"w{world}_rank0.json"))
    out = os.path.join(root, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_report.jsonl"), "a") as f:
        f.write(json.dumps({"test": "reference_train_epoch_replayed", "world": world, "wrapper": wrapper, **rep}) + "\n")
