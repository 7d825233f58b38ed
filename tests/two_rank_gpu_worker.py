"""Worker of tests/test_gpu_dist.py::test_two_ranks_share_one_gpu: one of TWO data-parallel ranks that run the native
training path (HIP kernels, flat gradient, bucketed exchange overlapped with the backward, fused clip + AdamW) on the
SAME MI355X.  RCCL refuses two ranks on one device, so the process group is gloo and snx.dist moves the buckets through
host copies made on the exchange stream (snx.dist.rccl): what is exercised is everything around the collective -- the
rank-0 broadcast, which backward exchanges, the stream ordering between the producers of a bucket and its exchange, the
wait before the optimizer -- with world size 2 and different data per rank.

usage: python -m tests.two_rank_gpu_worker RANK PORT OUTDIR"""
import json
import os
import sys

import torch
import torch.distributed as dist


def main(rank: int, port: int, outdir: str) -> None:
    from oracle import splade_oracle as O
    from src.model.losses import SPLADELossV33
    from src.train.config.v33 import V33Config
    from src.train.core import ddp_trainer as T
    from tests.test_gpu_model import _build_model, _small_cfg
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
    cfg = _small_cfg()
    params = O.perturb_params(O.init_params(cfg, seed=3), seed=4, scale=2.0, bias_mean=-0.1)
    mine = {k: (v if rank == 0 else v * 1.01 + 0.001) for k, v in params.items()}     # rank 1 starts elsewhere
    gen = torch.Generator().manual_seed(500 + rank)                                     # different data per rank
    batches = [O.synth_batch(4, 24, 70, cfg, gen, k=1, ragged=True) for _ in range(4)]
    conf = V33Config()
    conf.training.gradient_accumulation_steps = 2
    conf.training.learning_rate = 1e-3
    report = {"rank": rank, "cases": []}

    def mean_over_ranks(t: torch.Tensor) -> torch.Tensor:
        c = t.detach().float().cpu().clone()
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        return c / 2

    def same_on_both_ranks(t: torch.Tensor) -> bool:
        c = t.detach().float().cpu().contiguous()
        both = [torch.empty_like(c), torch.empty_like(c)]
        dist.all_gather(both, c)
        return torch.equal(both[0], both[1])

    for fused, mode, xneg in ((True, "allreduce", False), (False, "allreduce", False), (True, "rs_ag", False),
                              (False, "rs_ag", True)):
        os.environ["SNX_FUSED_PASSES"] = "1" if fused else "0"
        os.environ["SNX_GRAD_EXCHANGE"] = mode
        # local gradients of the first window, no exchange: an unwrapped model (autograd .grad) on rank 0's weights
        plain = _build_model(cfg, params, dev)
        loss_fn = SPLADELossV33(temperature=20.0, flops_warmup_steps=4).to(dev)
        local = None
        if not xneg:
            for b in batches[:2]:
                T.micro_step(plain, loss_fn, b, 0, dev, 2)
            local = {n: p.grad.detach().clone() for n, p in plain.named_parameters()}
        del plain
        # the data-parallel run
        model = T.NativeDataParallel(_build_model(cfg, mine, dev), n_buckets=3)
        for n, p in model.module.named_parameters():
            assert torch.equal(p.detach().cpu(), params[n].to(p.dtype)), f"broadcast: {n} differs from rank 0's"
        gsync = model.module.runtime.grad_sync
        assert gsync.mode == mode
        gsync.keep_log = True
        loss_fn = SPLADELossV33(temperature=20.0, flops_warmup_steps=4).to(dev)
        opt = T.build_optimizer(model, conf)
        sch = T.build_scheduler(opt, 0, 4)
        step, worst = 0, 0.0
        for i, b in enumerate(batches):
            last = (i + 1) % 2 == 0
            loss, _ = T.micro_step(model, loss_fn, b, step, dev, 2, cross_gpu_negatives=xneg, last_of_window=last)
            assert torch.isfinite(loss)
            if last:
                if step == 0 and local is not None:
                    model.sync_gradients()                  # wait for the overlapped exchange
                    torch.cuda.synchronize()
                    for n, p in model.module.named_parameters():
                        want = mean_over_ranks(local[n])
                        got = p.grad.detach().float().cpu()
                        err = float((got - want).abs().max()) / max(float(want.abs().max()), 1e-12)
                        worst = max(worst, err)
                        assert err <= 2e-4, f"{n}: averaged gradient off by {err:.3g} of its maximum"
                        assert same_on_both_ranks(p.grad), f"{n}: gradient differs between the ranks"
                T.optimizer_step(model, opt, sch, conf)
                step += 1
        torch.cuda.synchronize()
        for n, p in model.module.named_parameters():
            assert same_on_both_ranks(p), f"{n}: parameters differ between the ranks after {step} optimizer steps"
            assert torch.isfinite(p).all()
        nfwd = 1 if fused else 3
        epochs = sorted({e[1] for e in gsync.log if e[0] == "fwd"})
        assert len(epochs) == 2, gsync.log
        total = model.module.runtime.flat_grad.numel()
        for ep in epochs:
            i_ex = next(i for i, e in enumerate(gsync.log) if e[0] == "exchange" and e[1] == ep)
            before = [e for e in gsync.log[:i_ex] if e[0] in ("fwd", "bwd") and e[1] == ep]
            assert [e[0] for e in before] == ["fwd"] * nfwd + ["bwd"] * nfwd, before
            reds = []
            for e in gsync.log[i_ex + 1:]:
                if e[0] != "reduce":
                    break
                reds.append((e[1], e[2]))
            cov = sorted(reds)
            assert cov[0][0] == 0 and cov[-1][1] == total and all(a[1] == b[0] for a, b in zip(cov, cov[1:])), cov
        report["cases"].append({"fused": fused, "mode": mode, "cross_gpu_negatives": xneg,
                                "worst_avg_grad_err_rel_max": worst})
        del model, opt, sch
    # ---- cross-GPU negatives: the all-gather on the exchange stream (issued right after the positive pass / the fused
    # forward, only the loss waits for it) against the inline collective -- same values, same gradients, so after two
    # optimizer steps the parameters must be BIT-equal (ordered reductions: two runs of one form are bit-equal too)
    def run_xneg(fused: bool, inline: bool) -> torch.Tensor:
        os.environ["SNX_FUSED_PASSES"] = "1" if fused else "0"
        os.environ["SNX_GRAD_EXCHANGE"] = "allreduce"
        os.environ["SNX_GATHER_INLINE"] = "1" if inline else "0"
        model = T.NativeDataParallel(_build_model(cfg, mine, dev), n_buckets=3)
        loss_fn = SPLADELossV33(temperature=20.0, flops_warmup_steps=4).to(dev)
        opt = T.build_optimizer(model, conf)
        sch = T.build_scheduler(opt, 0, 4)
        step = 0
        for i, b in enumerate(batches):
            last = (i + 1) % 2 == 0
            T.micro_step(model, loss_fn, b, step, dev, 2, cross_gpu_negatives=True, last_of_window=last)
            if last:
                T.optimizer_step(model, opt, sch, conf)
                step += 1
        torch.cuda.synchronize()
        return torch.cat([p.detach().reshape(-1) for p in model.module.parameters()]).clone()

    for fused in (False, True):
        a, b_ = run_xneg(fused, inline=True), run_xneg(fused, inline=False)
        assert torch.isfinite(a).all()
        assert torch.equal(a, b_), f"exchange-stream gather differs from the inline one (fused={fused}): " \
                                   f"{float((a - b_).abs().max()):.3g}"
        assert same_on_both_ranks(b_)
        report["cases"].append({"fused": fused, "gather": "exchange stream == inline, bit for bit"})
    os.environ.pop("SNX_GATHER_INLINE", None)
    dist.barrier()
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump(report, f)
    dist.destroy_process_group()


def reference_replay(rank: int, port: int, outdir: str, world: int) -> None:
    """The reference's OWN run, replayed on the GPU.  tests/golden/g2_train_epoch_w{1,2} were captured from the
    reference's unmodified `train_epoch` on one / two gloo DDP ranks (tools/make_golden.py; CPU, where its cuda autocast
    is disabled, i.e. fp32): tiny configuration, batches dealt round-robin, accumulate 4 -> 2 optimizer steps.  Here
    `world` ranks on one MI355X run this repo's `train_epoch` under torch DDP(broadcast_buffers=False,
    find_unused_parameters=False) -- ref:src/train/cli/train_v33_ddp.py:539-544 -- on the fp32 kernels
    (SNX_PRECISION=fp32: the tiny geometry exists only there, and fp32 is what the golden's arithmetic was)."""
    import numpy as np
    from torch.nn.parallel import DistributedDataParallel as DDP
    from torch.utils.data import DataLoader, Dataset
    from oracle import splade_oracle as O
    from src.model.losses import SPLADELossV33
    from src.train.config.v33 import V33Config
    from src.train.core import ddp_trainer as T
    from tests.test_gpu_model import _build_model
    os.environ["SNX_PRECISION"] = "fp32"
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(G, f"g2_train_epoch_w{world}.npz"))
    meta = json.load(open(os.path.join(G, f"g2_train_epoch_w{world}.json")))
    c = meta["conf"]
    assert c["world"] == world
    cfg = O.EncoderConfig.tiny()
    params = O.perturb_params(O.init_params(cfg, seed=42), seed=7, scale=3.0)
    batches = []
    for i in range(meta["n_batches"]):
        b = {k.split("::")[1]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith(f"b{i}::")}
        b["num_negatives"] = 1
        batches.append(b)
    mine = [batches[j * world + rank] for j in range(len(batches) // world)]     # DistributedSampler(shuffle=False)

    class DS(Dataset):
        def __len__(self):
            return len(mine)

        def __getitem__(self, i):
            return mine[i]

    conf = V33Config()
    conf.training.gradient_accumulation_steps = c["accum"]
    conf.training.learning_rate = c["lr"]
    conf.training.weight_decay = c["wd"]
    conf.training.gradient_clip = c["clip"]
    conf.training.log_every_n_steps = 1
    if os.environ.get("REPLAY_WRAPPER", "ddp") == "native":
        # --native-dp: gradients accumulated in the flat buffer and exchanged ONCE per optimizer step (overlapped with the
        # window's last backward), fused clip + AdamW -- the same mathematics as the reference's per-micro-step reduction
        model = T.NativeDataParallel(_build_model(cfg, params, dev), n_buckets=3)
    else:
        model = DDP(_build_model(cfg, params, dev), device_ids=[0], broadcast_buffers=False, find_unused_parameters=False)
    loss_fn = SPLADELossV33(lambda_q=c["lambda_q"], lambda_d=c["lambda_d"], temperature=O.LossConfig().temperature,
                            flops_warmup_steps=c["flops_warmup_steps"], lambda_initial_ratio=c["lambda_initial_ratio"]).to(dev)
    opt = T.build_optimizer(model, conf)
    sch = T.build_scheduler(opt, c["warmup"], c["total_steps"])
    losses = []
    orig = T.micro_step

    def recording(*a, **k):
        loss, d = orig(*a, **k)
        losses.append(float(loss))
        return loss, d
    T.micro_step = recording
    try:
        avg, gs = T.train_epoch(model, DataLoader(DS(), batch_size=None, shuffle=False), loss_fn, opt, sch, conf, 0, 0, dev)
    finally:
        T.micro_step = orig
    torch.cuda.synchronize()
    assert gs == meta["global_step"] == 2
    rep = {"rank": rank}
    if rank == 0:
        want = np.asarray(z["losses"], dtype=np.float64)
        got = np.asarray(losses, dtype=np.float64)
        assert got.shape == want.shape, (got.shape, want.shape)
        rel = float(np.abs(got - want).max() / np.abs(want).max())
        assert rel <= 2e-4, f"per-micro-step losses differ from the reference's by {rel:.3g}"
        assert abs(avg - meta["avg_loss"]) <= 2e-4 * abs(meta["avg_loss"])
        worst_cos, frac_close, n_all = 1.0, 0, 0
        for n, p in model.module.named_parameters():
            ref = torch.from_numpy(np.asarray(z["p::" + n])).double()
            upd_ref = ref - params[n].double()
            upd = p.detach().cpu().double() - params[n].double()
            if float(upd_ref.abs().max()) > 0:
                cos = float((upd * upd_ref).sum() / (upd.norm() * upd_ref.norm() + 1e-30))
                worst_cos = min(worst_cos, cos)
                assert cos >= 0.995, f"{n}: update cosine {cos:.5f} against the reference's two-rank run"
            frac_close += int(((p.detach().cpu().double() - ref).abs() <= 5e-5).sum())
            n_all += ref.numel()
        rep.update({"loss_rel_err": rel, "worst_update_cos": worst_cos, "frac_params_within_5e-5": frac_close / n_all})
        # Adam turns a gradient whose sign is rounding noise into a full +-lr step: a small share of elements may sit a
        # step apart; everything else must be the reference's value
        assert frac_close / n_all >= 0.97, frac_close / n_all
    # both ranks hold the same parameters (DDP averaged every micro-step, as the reference's reducer does)
    for n, p in model.module.named_parameters():
        t = p.detach().float().cpu().contiguous()
        every = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        assert all(torch.equal(every[0], e) for e in every), n
    dist.barrier()
    with open(os.path.join(outdir, f"w{world}_rank{rank}.json"), "w") as f:
        json.dump(rep, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 4 and sys.argv[4].startswith("reference_w"):
        reference_replay(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4][len("reference_w"):]))
    else:
        main(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3])
