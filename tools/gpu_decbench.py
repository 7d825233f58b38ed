#!/usr/bin/env python3
"""Decoder + SPLADE tail at the bench's shapes: 64 queries of 64 tokens + 128 documents of 256 tokens, V = 50000."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opensearch-neural-pre-train_amd"))
import torch
from snx import ops
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
V, K = int(os.environ.get("V", 50000)), 768
lens = [64] * 64 + [256] * 128
T = sum(lens)
cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
hd = torch.randn(T, K, device=dev).to(BF16)
W = (torch.randn(V, K, device=dev) * 0.05).to(BF16)
bias = torch.randn(V, device=dev) * 0.3
mask = torch.ones(T, dtype=torch.int64, device=dev)
def run(): return ops.decoder_splade_fwd(hd, W, bias, cu, mask, 256, validate=False)
for _ in range(3): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = int(os.environ.get("N", 10))
a.record()
for _ in range(n): run()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / n
print(f"decoder fwd: {ms:.3f} ms  {2.0 * T * V * K / ms / 1e9:.0f} TFLOP/s", flush=True)
from snx import _lib
if hasattr(_lib.lib(), "snx_dec256_trace_set"):     # diagnostics build (-DSNX_GEMM_TRACE)
    import ctypes as C
    buf = torch.zeros(256 * 4, dtype=torch.int64, device=dev)
    _lib.lib().snx_dec256_trace_set(C.c_void_p(buf.data_ptr()))
    run(); torch.cuda.synchronize()
    _lib.lib().snx_dec256_trace_set(C.c_void_p(0))
    t = buf.view(256, 4).cpu().double()
    t = t[t[:, 3] > 0]
    ghz = t[:, 0] / (t[:, 1] * 10.0)
    print(f"in-kernel: {len(t)} workgroups, clock {ghz.median():.2f} GHz, kernel {t[:, 1].median() * 0.01:.0f} us, tiles "
          f"{int(t[:, 3].min())}..{int(t[:, 3].max())}, cycles per tile {(t[:, 0] / t[:, 3]).median():.0f} of which epilogue "
          f"{(t[:, 2] / t[:, 3]).median():.0f}", flush=True)
