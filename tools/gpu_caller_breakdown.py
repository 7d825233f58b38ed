#!/usr/bin/env python3
"""Where the unchanged caller's time goes (round-4 review, item 5): ref:src/train/cli/train_v33_ddp.py:339-343,444 runs
three model(...) calls and loss.item() per micro-step; bench.py's headline runs one fused native pass and no host sync.
Four legs on one box, same batches, same optimizer cadence:
   A fused pass, no sync (headline)      B fused pass + loss.item() per micro-step
   C three passes, no sync               D three passes + loss.item()  (= extra.value_with_item_sync)
plus the host's enqueue time per micro-step for the fused and the three-pass pattern (time to return from micro_step with
the device kept busy) and the per-class device time of both patterns (snx_prof, kernels timed one at a time)."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opensearch-neural-pre-train_amd"))
sys.path.insert(0, ROOT)
import torch

import bench
from snx._lib import fn
from src.model.losses import SPLADELossV33
from src.model.splade_modern import SPLADEModernBERT
from src.train.config.v33 import V33Config
from src.train.core import ddp_trainer as T

dev = torch.device("cuda:0")
torch.manual_seed(42)
import logging
logging.getLogger("src.model.splade_modern").setLevel(logging.ERROR)
model = SPLADEModernBERT().to(dev)
config = V33Config()
ACC = 4
config.training.gradient_accumulation_steps = ACC
wrapped = T.NativeDataParallel(model)
loss_fn = SPLADELossV33(lambda_q=config.loss.lambda_q, lambda_d=config.loss.lambda_d, temperature=config.loss.temperature,
                        flops_warmup_steps=config.loss.flops_warmup_steps,
                        lambda_initial_ratio=config.loss.lambda_initial_ratio).to(dev)
opt = T.build_optimizer(wrapped, config)
sch = T.build_scheduler(opt, 100, 10000)
batches = bench.make_batches(16, 64, 64, 256, 1, model.vocab_size, model.config.pad_token_id, 42, dev)
st = {"i": 0, "gs": 0}


def step(fused, sync):
    os.environ["SNX_FUSED_PASSES"] = "1" if fused else "0"
    b = batches[st["i"] % len(batches)]
    loss, _ = T.micro_step(wrapped, loss_fn, b, st["gs"], dev, ACC, False, last_of_window=(st["i"] + 1) % ACC == 0)
    st["i"] += 1
    if sync:
        loss.item()
    if st["i"] % ACC == 0:
        T.optimizer_step(wrapped, opt, sch, config)
        st["gs"] += 1


def leg(fused, sync, n=32, warm=8):
    for _ in range(warm):
        step(fused, sync)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(n):
        h0 = time.perf_counter()
        step(fused, sync)
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"ms_per_micro_step": 1e3 * dt / n, "triplets_per_s": 64 * n / dt, "host_ms_in_step_calls": 1e3 * host / n}


def classes(fused, n=4):
    fn("snx_prof_enable")(1)
    for _ in range(n):
        step(fused, False)
    torch.cuda.synchronize()
    nc = fn("snx_prof_num_classes")()
    ms, ln, wk = (C.c_double * nc)(), (C.c_int64 * nc)(), (C.c_double * nc)()
    fn("snx_prof_read")(ms, ln, wk)
    fn("snx_prof_enable")(0)
    return {fn("snx_prof_class_name")(i).decode(): {"ms": ms[i] / n, "launches": ln[i] / n} for i in range(nc) if ln[i]}


out = {}
wrapped.zero_grad()
for name, fused, sync in (("A_fused_nosync", True, False), ("B_fused_item", True, True), ("C_three_nosync", False, False),
                          ("D_three_item", False, True)):
    out[name] = leg(fused, sync)
    print(name, json.dumps(out[name]), flush=True)
out["classes_fused"] = classes(True)
out["classes_three"] = classes(False)
tot_f = sum(v["ms"] for v in out["classes_fused"].values())
tot_t = sum(v["ms"] for v in out["classes_three"].values())
out["device_ms_sum"] = {"fused": tot_f, "three": tot_t}
print("device ms per micro-step (kernels one at a time): fused %.2f, three passes %.2f" % (tot_f, tot_t))
for c in out["classes_fused"]:
    a, b = out["classes_fused"][c], out["classes_three"].get(c, {"ms": 0, "launches": 0})
    print(f"  {c:22s} fused {a['ms']:7.3f} ms / {a['launches']:6.1f}   three {b['ms']:7.3f} ms / {b['launches']:6.1f}   diff {b['ms'] - a['ms']:+.3f}")
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r5_caller_breakdown.json"), "w"), indent=1)
