#!/usr/bin/env python3
"""Per-kernel timings on the GPU box (HIP events via torch on the launch stream)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opensearch-neural-pre-train_amd"))
import torch
from snx import ops

dev = torch.device("cuda:0")
BF16 = torch.bfloat16


def timeit(f, n=20, warm=3):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


res = {}
for (M, N, K) in [(16384, 2304, 768), (16384, 768, 768), (16384, 768, 1152), (4096, 2304, 768), (36864, 2304, 768)]:
    a = torch.randn(M, K, device=dev).to(BF16)
    b = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
    ms = timeit(lambda: ops.gemm_nt(a, b))
    res[f"gemm_nt_{M}x{N}x{K}"] = {"ms": ms, "TFLOPs": 2.0 * M * N * K / ms / 1e9}
    ms = timeit(lambda: torch.matmul(a, b.t()))
    res[f"torch_mm_{M}x{N}x{K}"] = {"ms": ms, "TFLOPs": 2.0 * M * N * K / ms / 1e9}
for (B, S) in [(64, 256), (64, 64)]:
    T, V, K = B * S, 50000, 768
    hd = torch.randn(T, K, device=dev).to(BF16)
    W = (torch.randn(V, K, device=dev) * 0.05).to(BF16)
    bias = torch.zeros(V, device=dev)
    cu = (torch.arange(B + 1, dtype=torch.int32) * S).to(dev)
    mask = torch.ones(T, dtype=torch.int64, device=dev)
    ms = timeit(lambda: ops.decoder_splade_fwd(hd, W, bias, cu, mask, S, validate=False), n=10)
    res[f"decoder_splade_{B}x{S}"] = {"ms": ms, "TFLOPs": 2.0 * T * V * K / ms / 1e9}
    heads = 12
    qkv = torch.randn(T, 3 * heads * 64, device=dev).to(BF16)
    for w in (-1, 64):
        ms = timeit(lambda: ops.attn_fwd(qkv, cu, mask, S, heads, w, validate=False))
        res[f"attn_fwd_{B}x{S}_w{w}"] = {"ms": ms}
print(json.dumps(res, indent=1))
