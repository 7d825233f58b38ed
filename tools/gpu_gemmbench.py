#!/usr/bin/env python3
"""GEMM-only timings at the encoder's shapes."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opensearch-neural-pre-train_amd"))
import torch
from snx import ops
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
def timeit(f, n=30, warm=5):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
res = {}
VAR = 0
for (M, N, K) in [(16384, 2304, 768), (16384, 768, 768), (16384, 768, 1152), (16384, 768, 2304), (16384, 1152, 768), (4096, 2304, 768), (4096, 768, 768)]:
    a = torch.randn(M, K, device=dev).to(BF16); b = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
    h = torch.randn(M, N, device=dev)
    ms = timeit(lambda: ops.gemm_nt(a, b)); res[f"nt_{M}x{N}x{K}"] = (round(ms * 1e3, 1), round(2.0 * M * N * K / ms / 1e9))
    if N == 768:
        ms = timeit(lambda: ops.gemm_nt_resid(a, b, h)); res[f"ntres_{M}x{N}x{K}"] = (round(ms * 1e3, 1), round(2.0 * M * N * K / ms / 1e9))
    ms = timeit(lambda: torch.matmul(a, b.t())); res[f"torch_{M}x{N}x{K}"] = (round(ms * 1e3, 1), round(2.0 * M * N * K / ms / 1e9))
for (M, N, K) in [(16384, 2304, 768), (16384, 768, 768), (16384, 768, 1152), (4096, 2304, 768)]:
    dy = (torch.randn(M, N, device=dev) * 0.1).to(BF16); x = torch.randn(M, K, device=dev).to(BF16)
    dw = torch.zeros(N, K, device=dev)
    ms = timeit(lambda: ops.gemm_tn_accum(dy, x, dw)); res[f"tn_{M}x{N}x{K}"] = (round(ms * 1e3, 1), round(2.0 * M * N * K / ms / 1e9))
    ms = timeit(lambda: torch.matmul(dy.t(), x)); res[f"torch_tn_{M}x{N}x{K}"] = (round(ms * 1e3, 1), round(2.0 * M * N * K / ms / 1e9))
for k, v in res.items(): print(f"{k:28s} {v[0]:8.1f} us {v[1]:6d} TF/s")
# correctness spot-check of the selected variant
a = torch.randn(1000, 768, device=dev).to(BF16); b = (torch.randn(640, 768, device=dev) * 0.05).to(BF16)
c = ops.gemm_nt(a, b); ref = (a.float() @ b.float().t()).to(BF16)
print("variant", VAR, "max abs diff vs torch", float((c.float() - ref.float()).abs().max()), "exact frac", float((c == ref).float().mean()))
