#!/usr/bin/env python3
"""Throughput of the fp32 path (inference outside autocast, ref:benchmark/encoders.py:309-345) next to the bf16 kernels
on the same 149 M model: documents / s for a batch of 64 x 256 tokens."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opensearch-neural-pre-train_amd"))
import torch  # noqa: E402
from src.model.splade_modern import SPLADEModernBERT  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
import logging  # noqa: E402
logging.getLogger("src.model.splade_modern").setLevel(logging.ERROR)
model = SPLADEModernBERT().to(dev).eval()
B, S = 64, 256
ids = torch.randint(6, 49000, (B, S), device=dev)
mask = torch.ones_like(ids)


def run(n, autocast):
    with torch.no_grad(), torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=autocast):
        for _ in range(2):
            model(ids, mask)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            model(ids, mask)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


t32 = run(3, False)
t16 = run(10, True)
print(f"fp32 path: {t32 * 1e3:.1f} ms per batch of {B} x {S} tokens = {B / t32:.0f} documents/s; "
      f"bf16 kernels: {t16 * 1e3:.1f} ms = {B / t16:.0f} documents/s", flush=True)
