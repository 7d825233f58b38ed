#!/usr/bin/env python3
"""NT GEMM epilogue variants at the bench's token count (M = 36864): time and TFLOP/s per variant."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opensearch-neural-pre-train_amd"))
import torch
from snx import ops

dev = torch.device("cuda:0")
BF16 = torch.bfloat16
M = int(os.environ.get("M", 36864))


def timeit(f, n=30, warm=5):
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


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(BF16)


def report(name, us, flops):
    print(f"{name:34s} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)


H, I = 768, 1152
x = rnd(M, H); w = rnd(H, H, scale=0.05); hin = torch.randn(M, H, device=dev)
report("store  N=768 K=768", timeit(lambda: ops.gemm_nt(x, w)), 2.0 * M * H * H)
x3 = rnd(M, 3 * H); w3 = rnd(H, 3 * H, scale=0.05)
report("store  N=768 K=2304", timeit(lambda: ops.gemm_nt(x3, w3)), 2.0 * M * H * 3 * H)
report("torch  N=768 K=2304", timeit(lambda: torch.matmul(x3, w3.t())), 2.0 * M * H * 3 * H)
report("torch  N=768 K=768", timeit(lambda: torch.matmul(x, w.t())), 2.0 * M * H * H)
report("resid  N=768 K=768", timeit(lambda: ops.gemm_nt_resid(x, w, hin)), 2.0 * M * H * H)
y = rnd(M, I); wo = rnd(H, I, scale=0.05)
report("resid  N=768 K=1152", timeit(lambda: ops.gemm_nt_resid(y, wo, hin)), 2.0 * M * H * I)
wqkv = rnd(3 * H, H, scale=0.05)
report("store  N=2304 K=768", timeit(lambda: ops.gemm_nt(x, wqkv)), 2.0 * M * 3 * H * H)
report("torch  N=2304 K=768", timeit(lambda: torch.matmul(x, wqkv.t())), 2.0 * M * 3 * H * H)
tab = ops.rope_table(256, 64, 160000.0, dev)
pos = torch.arange(256, dtype=torch.int32, device=dev).repeat(M // 256)
report("rope   N=2304 K=768", timeit(lambda: ops.gemm_nt_rope(x, wqkv, tab, pos, 2 * H, validate=False)), 2.0 * M * 3 * H * H)
wi = rnd(2 * I, H, scale=0.05)
report("geglu_fwd N=2304 K=768", timeit(lambda: ops.gemm_nt_geglu_fwd(x, wi)), 2.0 * M * 2 * I * H)
u = rnd(M, 2 * I); wot = rnd(I, H, scale=0.05)
report("store  N=1152 K=768", timeit(lambda: ops.gemm_nt(x, wot)), 2.0 * M * I * H)
report("geglu_bwd N=1152 K=768", timeit(lambda: ops.gemm_nt_geglu_bwd(x, wot, u)), 2.0 * M * I * H)
import snx
snx.configure(nt_pipe=0)
report("geglu_bwd (128x128 kernel)", timeit(lambda: ops.gemm_nt_geglu_bwd(x, wot, u)), 2.0 * M * I * H)
snx.configure(nt_pipe=2)
