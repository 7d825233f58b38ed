#!/usr/bin/env python3
"""GeGLU-backward GEMM (N = 1152, K = 768) at the bench's token count: pipelined kernel vs the 128x128 kernel.
SNX_LIB selects a library variant (timing-only -DSNX_PIPE_DIAG builds: tools/_probe/libsnx_pipe<n>.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opensearch-neural-pre-train_amd"))
import torch
import snx
from snx import ops

dev = torch.device("cuda:0")
BF16 = torch.bfloat16
M = int(os.environ.get("M", 36864))
H, I = 768, 1152


def timeit(f, n=40, warm=8):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


x = (torch.randn(M, H, device=dev)).to(BF16)
wot = (torch.randn(I, H, device=dev) * 0.05).to(BF16)
u = torch.randn(M, 2 * I, device=dev).to(BF16)
tag = os.path.basename(os.environ.get("SNX_LIB", "libsnx.so"))
us = timeit(lambda: ops.gemm_nt_geglu_bwd(x, wot, u))
print(f"{tag:22s} pipelined   {us:8.1f} us  {2.0 * M * I * H / us / 1e6:7.1f} TFLOP/s", flush=True)
snx.configure(nt_pipe=1)
us = timeit(lambda: ops.gemm_nt_geglu_bwd(x, wot, u))
print(f"{tag:22s} nt stores   {us:8.1f} us  {2.0 * M * I * H / us / 1e6:7.1f} TFLOP/s", flush=True)
snx.configure(nt_pipe=0)
us = timeit(lambda: ops.gemm_nt_geglu_bwd(x, wot, u))
print(f"{tag:22s} 128x128     {us:8.1f} us  {2.0 * M * I * H / us / 1e6:7.1f} TFLOP/s", flush=True)
us = timeit(lambda: ops.gemm_nt(x, wot))
print(f"{tag:22s} plain store {us:8.1f} us  {2.0 * M * I * H / us / 1e6:7.1f} TFLOP/s", flush=True)
