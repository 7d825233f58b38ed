#!/usr/bin/env python3
"""Host time to ENQUEUE one micro-step (Python + ctypes + launches) against the GPU time it takes: the launch thread must stay
ahead of the device.  Measures the bench's micro-step with the device idle at the start of each sample (sync before, no
sync after): the call returns when everything is queued."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "opensearch-neural-pre-train_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from src.model.losses import SPLADELossV33  # noqa: E402
from src.model.splade_modern import SPLADEModernBERT  # noqa: E402
from src.train.config.v33 import V33Config  # noqa: E402
from src.train.core import ddp_trainer as T  # noqa: E402
import logging  # noqa: E402

logging.getLogger("src.model.splade_modern").setLevel(logging.ERROR)
dev = torch.device("cuda:0")
torch.manual_seed(42)
model = SPLADEModernBERT().to(dev)
conf = V33Config()
conf.training.gradient_accumulation_steps = 4
wrapped = T.NativeDataParallel(model)
loss_fn = SPLADELossV33(lambda_q=conf.loss.lambda_q, lambda_d=conf.loss.lambda_d, temperature=conf.loss.temperature,
                        flops_warmup_steps=conf.loss.flops_warmup_steps).to(dev)
opt = T.build_optimizer(wrapped, conf)
sch = T.build_scheduler(opt, 100, 10000)
batches = bench.make_batches(4, 64, 64, 256, 1, model.vocab_size, model.config.pad_token_id, 42, dev, teacher=False)
wrapped.zero_grad()
host, total = [], []
for i in range(24):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    T.micro_step(wrapped, loss_fn, batches[i % 4], 0, dev, 4, False, last_of_window=(i + 1) % 4 == 0)
    if (i + 1) % 4 == 0:
        T.optimizer_step(wrapped, opt, sch, conf)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if i >= 8:
        host.append((t1 - t0) * 1e3)
        total.append((t2 - t0) * 1e3)
print(f"host enqueue per micro-step: median {statistics.median(host):.2f} ms (max {max(host):.2f}); device: {statistics.median(total):.2f} ms",
      flush=True)
