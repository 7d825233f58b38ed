#!/usr/bin/env python3
"""Host-side cost of each phase of a micro-step (no device sync inside the phases)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opensearch-neural-pre-train_amd")); sys.path.insert(0, ROOT)
import torch, logging
logging.getLogger("src.model.splade_modern").setLevel(logging.ERROR)
from src.model.losses import SPLADELossV33
from src.model.splade_modern import SPLADEModernBERT
from src.train.config.v33 import V33Config
from src.train.core import ddp_trainer as T
import bench
dev = torch.device("cuda:0")
torch.manual_seed(42)
model = SPLADEModernBERT().to(dev)
cfg = V33Config()
w = T.NativeDataParallel(model)
loss_fn = SPLADELossV33().to(dev)
opt = T.build_optimizer(w, cfg); sch = T.build_scheduler(opt, 100, 10000)
batches = bench.make_batches(4, 64, 64, 256, 1, 50000, 49999, 42, dev)
def step(i, timing=None):
    b = batches[i % 4]
    t = [time.perf_counter()]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        q, _ = w(b["query_input_ids"], b["query_attention_mask"]); t.append(time.perf_counter())
        p, _ = w(b["positive_input_ids"], b["positive_attention_mask"]); t.append(time.perf_counter())
        n, _ = w(b["negative_input_ids"], b["negative_attention_mask"]); t.append(time.perf_counter())
        loss, d = loss_fn(anchor_repr=q, positive_repr=p, negative_repr=n, global_step=0); t.append(time.perf_counter())
    (loss / 4).backward(); t.append(time.perf_counter())
    if (i + 1) % 4 == 0:
        T.optimizer_step(w, opt, sch, cfg)
    t.append(time.perf_counter())
    if timing is not None:
        timing.append([1000 * (b_ - a_) for a_, b_ in zip(t, t[1:])])
for i in range(4): step(i)
torch.cuda.synchronize()
tm = []
t0 = time.perf_counter()
for i in range(8): step(i, tm)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host ms per step (fwd q, fwd p, fwd n, loss, backward, opt):")
for r in tm: print(["%.2f" % x for x in r])
print("host loop total %.1f ms, after sync %.1f ms" % (1000 * (t1 - t0), 1000 * (t2 - t0)))
