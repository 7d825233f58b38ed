#!/usr/bin/env python3
"""In-kernel timeline of the attention dQ kernel (diagnostics build: SNX_EXTRA_HIPCC_FLAGS=-DSNX_ATTN_TRACE).
Per workgroup (wave 0): cycles from entry to images-loaded, to the end of its first and second row group; wall ns."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opensearch-neural-pre-train_amd"))
import torch  # noqa: E402
from snx import ops  # noqa: E402
from snx._lib import lib  # noqa: E402

dev = torch.device("cuda:0")
heads = 12
lens = ([64] * 64 + [256] * 128) if "fused" in sys.argv else [256] * 128
groups = [(0, 64, 64), (64, 128, 256)] if "fused" in sys.argv else None
cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
T = int(cu[-1])
mask = torch.ones(T, dtype=torch.int64, device=dev)
qkv = torch.randn(T, 3 * heads * 64, device=dev).to(torch.bfloat16)
dout = torch.randn(T, heads * 64, device=dev).to(torch.bfloat16)
L = lib()
fset = L.snx_attn1p_trace_set
fset.restype = C.c_int
fset.argtypes = [C.c_void_p]
for w in (-1, 64):
    out, lse = ops.attn_fwd(qkv, cu, mask, 256, heads, w, validate=False, groups=groups)
    for _ in range(3):
        ops.attn_bwd(qkv, out, dout, lse, cu, mask, 256, heads, w, validate=False, groups=groups)
    nb = len(lens) * heads
    buf = torch.zeros(nb * 16, dtype=torch.int64, device=dev)
    fset(buf.data_ptr())
    ops.attn_bwd(qkv, out, dout, lse, cu, mask, 256, heads, w, validate=False, groups=groups)
    torch.cuda.synchronize()
    fset(0)
    b = buf.view(nb, 16).cpu().double()
    if groups:                                             # documents first (longest group first), then the queries
        nd = 128 * heads
        for name, part in (("documents", b[:nd]), ("queries", b[nd:])):
            print(f"  {name}: {part.shape[0]} workgroups, wall median {float(((part[:, 13] - part[:, 12]) * 10).median()):.0f} ns, "
                  f"first start {float(part[:, 12].min() - b[:, 12].min()) * 10 / 1e3:.1f} us, last end "
                  f"{float(part[:, 13].max() - b[:, 12].min()) * 10 / 1e3:.1f} us after the kernel's first workgroup", flush=True)
    b = b[b[:, 12] > 0]                                   # workgroups that ran (a persistent launch has fewer)
    nb = b.shape[0]
    med = lambda x: float(x.median())   # noqa: E731
    names = ["prologue", "s0 top+barrier", "s0 phase 1", "s0 barrier", "s0 phase 2", "s1 top+barrier", "s1 phase 1",
             "s1 barrier", "s1 phase 2"]
    parts = ", ".join(f"{n} {med(b[:, i + 1] - b[:, i]):.0f}" for i, n in enumerate(names))
    print(f"window={w} (cycles, wave 0, median over workgroups): {parts}; loop {med(b[:, 10] - b[:, 1]):.0f}, "
          f"epilogue {med(b[:, 11] - b[:, 10]):.0f}, whole {med(b[:, 11] - b[:, 0]):.0f}; workgroup wall "
          f"{med((b[:, 13] - b[:, 12]) * 10):.0f} ns; kernel span {float(b[:, 13].max() - b[:, 12].min()) * 10 / 1e3:.1f} us "
          f"for {nb} workgroups", flush=True)
