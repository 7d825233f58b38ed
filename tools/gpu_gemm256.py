#!/usr/bin/env python3
import os, sys
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
# correctness first (small, edges)
for (M, N, K) in [(256, 256, 64), (256, 256, 768), (512, 768, 768), (300, 1000, 128), (1000, 2304, 768), (4096, 768, 1152)]:
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g).to(dev).to(BF16); b = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(BF16)
    c = ops.gemm_nt256(a, b); torch.cuda.synchronize()
    ref = (a.float() @ b.float().t()).to(BF16)
    d = (c.float() - ref.float()).abs()
    print(f"check {M}x{N}x{K}: max diff {float(d.max()):.4f} exact {float((c == ref).float().mean()):.5f}", flush=True)
for (M, N, K) in [(16384, 2304, 768), (36864, 2304, 768), (36864, 768, 768), (36864, 768, 2304), (36864, 1152, 768), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev).to(BF16); b = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
    t1 = timeit(lambda: ops.gemm_nt(a, b)); t2 = timeit(lambda: ops.gemm_nt256(a, b)); t3 = timeit(lambda: torch.matmul(a, b.t()))
    fl = 2.0 * M * N * K / 1e9
    print(f"{M}x{N}x{K}: nt128 {t1*1e3:7.1f} us {fl/t1:6.0f} TF | nt256 {t2*1e3:7.1f} us {fl/t2:6.0f} TF | torch {t3*1e3:7.1f} us {fl/t3:6.0f} TF", flush=True)
for (M, N, K) in [(36864, 2304, 768), (36864, 768, 768), (36864, 768, 1152)]:
    dy = (torch.randn(M, N, device=dev) * 0.1).to(BF16); x = torch.randn(M, K, device=dev).to(BF16)
    dw = torch.zeros(N, K, device=dev)
    t1 = timeit(lambda: ops.gemm_tn_accum(dy, x, dw)); t3 = timeit(lambda: torch.matmul(dy.t(), x))
    fl = 2.0 * M * N * K / 1e9
    print(f"TN {M}x{N}x{K}: mine {t1*1e3:7.1f} us {fl/t1:6.0f} TF | torch {t3*1e3:7.1f} us {fl/t3:6.0f} TF", flush=True)
