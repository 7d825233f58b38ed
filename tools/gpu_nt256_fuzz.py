#!/usr/bin/env python3
"""Race / edge screen for the 256x256 persistent NT kernel: random shapes (M 8,192-40,000 any value, N and K multiples of
64 up to 2,560), every epilogue, each shape several times, against the 128x128 kernel bit for bit.  A synchronisation slip
in a hand-scheduled LDS-DMA loop shows as rare wrong tiles that come and go: this is the many-runs screen, the fixed shape
classes are in tests/test_gpu_ops.py.   SHAPES=40 REPEATS=3 SEED=1 python tools/gpu_nt256_fuzz.py"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opensearch-neural-pre-train_amd"))
import torch  # noqa: E402
from snx import ops  # noqa: E402
from snx._lib import fn  # noqa: E402

dev = torch.device("cuda:0")
BF16 = torch.bfloat16
rng = random.Random(int(os.environ.get("SEED", 1)))
nshapes, repeats = int(os.environ.get("SHAPES", 40)), int(os.environ.get("REPEATS", 3))
bad = 0
for si in range(nshapes):
    M = rng.randint(8192, 40000)
    N = 64 * rng.randint(1, 40)
    K = 64 * rng.randint(1, 40)
    g = torch.Generator().manual_seed(si)
    x = torch.randn(M, K, generator=g).to(dev).to(BF16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(BF16)
    hin = torch.randn(M, N, generator=g).to(dev)
    tab = ops.rope_table(256, 64, 160000.0, dev)
    pos = (torch.arange(M, dtype=torch.int32, device=dev) % 256).contiguous()
    rc = (2 * N // 3) // 64 * 64
    rows = ops.rope_rows(tab, pos)
    u = torch.randn(M, 2 * N, generator=g).to(dev).to(BF16)
    calls = {"store": lambda: (ops.gemm_nt(x, w),), "resid": lambda: (ops.gemm_nt_resid(x, w, hin),),
             "rope_rows": lambda: (ops.gemm_nt_rope_rows(x, w, tab, pos, rows, rc),),
             "geglu_fwd": lambda: ops.gemm_nt_geglu_fwd(x, w), "geglu_bwd": lambda: (ops.gemm_nt_geglu_bwd(x, w, u),)}
    for name, f in calls.items():
        fn("snx_nt256_configure")(0, 0)
        ref = [t.clone() for t in f()]
        fn("snx_nt256_configure")(2, 1024)
        for rep in range(repeats):
            got = f()
            torch.cuda.synchronize()
            if not all(torch.equal(a, b) for a, b in zip(ref, got)):
                bad += 1
                print(f"MISMATCH {name} M={M} N={N} K={K} repeat {rep}", flush=True)
    print(f"shape {si + 1}/{nshapes}: M={M} N={N} K={K} ok so far, mismatches {bad}", flush=True)
fn("snx_nt256_configure")(1, 8192)
print("FUZZ", "FAILED" if bad else "PASSED", flush=True)
sys.exit(1 if bad else 0)
