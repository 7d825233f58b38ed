#!/usr/bin/env python3
"""Weight-gradient GEMM at the bench's token count: the layer's four Linears grouped, in pairs and alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opensearch-neural-pre-train_amd"))
import torch
from snx import ops
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
M = int(os.environ.get("M", 36864)); H, I = 768, 1152
def rnd(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).to(BF16)
probs = [(rnd(M, 3 * H, scale=0.1), rnd(M, H), torch.zeros(3 * H, H, device=dev), False),
         (rnd(M, 2 * I, scale=0.1), rnd(M, H), torch.zeros(2 * I, H, device=dev), True),
         (rnd(M, H, scale=0.1), rnd(M, I), torch.zeros(H, I, device=dev), False),
         (rnd(M, H, scale=0.1), rnd(M, H), torch.zeros(H, H, device=dev), False)]
def timeit(f, n=20, warm=3):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
def fl(ps): return sum(2.0 * M * p[0].shape[1] * p[1].shape[1] for p in ps)
def rep(name, ps):
    us = timeit(lambda: ops.gemm_tn_accum_group(ps)); print(f"{name:22s} {us:8.1f} us {fl(ps) / us / 1e6:7.1f} TFLOP/s", flush=True)
rep("group of 4", probs)
from snx import _lib
if hasattr(_lib.lib(), "snx_tn256_trace_set"):      # diagnostics build (-DSNX_GEMM_TRACE): in-kernel clock and cycles
    import ctypes as C
    buf = torch.zeros(256 * 4, dtype=torch.int64, device=dev)
    _lib.lib().snx_tn256_trace_set(C.c_void_p(buf.data_ptr()))
    ops.gemm_tn_accum_group(probs); torch.cuda.synchronize()
    _lib.lib().snx_tn256_trace_set(C.c_void_p(0))
    t = buf.view(256, 4).cpu().double()
    t = t[t[:, 2] > 0]
    ghz = t[:, 0] / t[:, 1] / 10.0 * 1e-3 * 1e3 / 100.0 if False else t[:, 0] / (t[:, 1] * 10.0)   # cycles / ns
    us = t[:, 1] * 0.01
    print(f"in-kernel: {len(t)} workgroups, clock {ghz.median():.2f} GHz, K loop {us.min():.0f}..{us.max():.0f} us "
          f"(median {us.median():.0f}), K-steps {int(t[:, 2].min())}..{int(t[:, 2].max())}, "
          f"cycles per half-step {(t[:, 0] / t[:, 2] / 2).median():.0f}", flush=True)
    main = t[t[:, 3] < 204]; tail = t[t[:, 3] >= 204]
    if len(tail):
        print(f"  main workgroups {main[:, 1].median() * 0.01:.0f} us / {int(main[:, 2].median())} K-steps; tail "
              f"{tail[:, 1].median() * 0.01:.0f} us / {int(tail[:, 2].median())} K-steps", flush=True)
def torch_tn(ps):
    for dy, x, dw, _ in ps: torch.matmul(dy.t(), x)
us = timeit(lambda: torch_tn(probs)); print(f"{'torch dY^T X (bf16 out) x4':22s} {us:8.1f} us {fl(probs) / us / 1e6:7.1f} TFLOP/s", flush=True)
if os.environ.get("QUICK"): sys.exit(0)
rep("pair 0+1", probs[:2])
rep("pair 2+3", probs[2:])
for i, p in enumerate(probs):
    rep(f"single {i} N={p[0].shape[1]} K={p[1].shape[1]}", [p])
