#!/usr/bin/env python3
"""Single-query latency of the encoder forward (inference, no gradient): B = 1, 8, 64 at 64 tokens; fp32 path (what the
reference's inference encoder computes in) and the bf16 kernels under autocast.  Host-synchronous wall time per call."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opensearch-neural-pre-train_amd"))
import torch  # noqa: E402
import logging  # noqa: E402
from src.model.splade_modern import SPLADEModernBERT  # noqa: E402

logging.getLogger("src.model.splade_modern").setLevel(logging.ERROR)
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = SPLADEModernBERT().to(dev).eval()
for B in (1, 8, 64):
    ids = torch.randint(6, 49000, (B, 64), device=dev)
    mask = torch.ones_like(ids)
    for autocast in (False, True):
        with torch.no_grad(), torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=autocast):
            for _ in range(5):
                model(ids, mask)
            torch.cuda.synchronize()
            ts = []
            for _ in range(30):
                t0 = time.perf_counter()
                out = model(ids, mask)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
        print(f"B={B:3d} x 64 tokens, {'bf16 autocast' if autocast else 'fp32 path    '}: median {statistics.median(ts):.3f} ms  min {min(ts):.3f} ms", flush=True)
