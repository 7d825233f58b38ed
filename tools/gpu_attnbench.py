#!/usr/bin/env python3
"""Attention fwd/bwd timings on the fused query+document layout (64 x 64-token + 128 x 256-token sequences)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opensearch-neural-pre-train_amd"))
import torch
import snx
from snx import ops
from snx._lib import fn, check

dev = torch.device("cuda:0")
DEFAULT = snx.config("attn_bwd_onepass")
BF16 = torch.bfloat16


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


heads = 12
for name, lens in [("fused q+p+n", [64] * 64 + [256] * 128), ("docs only", [256] * 128)]:
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
    T = int(cu[-1])
    mask = torch.ones(T, dtype=torch.int64, device=dev)
    qkv = torch.randn(T, 3 * heads * 64, device=dev).to(BF16)
    dout = torch.randn(T, heads * 64, device=dev).to(BF16)
    grp = [(0, 64, 64), (64, 128, 256)] if len(lens) == 192 else None
    for w in (-1, 64):
        for gr in ([None, grp] if grp else [None]):
            out, lse = ops.attn_fwd(qkv, cu, mask, 256, heads, w, validate=False, groups=gr)
            f = timeit(lambda: ops.attn_fwd(qkv, cu, mask, 256, heads, w, validate=False, groups=gr))
            res = {}
            for mode in (1, 0):
                snx.configure(attn_bwd_onepass=mode)
                t = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, cu, mask, 256, heads, w, validate=False, groups=gr))
                res[mode] = (t, ops.attn_bwd(qkv, out, dout, lse, cu, mask, 256, heads, w, validate=False, groups=gr).float())
            snx.configure(attn_bwd_onepass=DEFAULT)
            rel1 = float((res[1][1] - res[0][1]).norm() / res[0][1].norm())
            print(f"{name} window={w} groups={'yes' if gr else 'no'}: fwd {f:.1f} us, bwd one-pass {res[1][0]:.1f} us, "
                  f"two-kernel {res[0][0]:.1f} us, rel diff {rel1:.2e}", flush=True)

# the model's call: inverse RoPE fused into the dq / dk stores
lens = [64] * 64 + [256] * 128
cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
T = int(cu[-1])
mask = torch.ones(T, dtype=torch.int64, device=dev)
qkv = torch.randn(T, 3 * heads * 64, device=dev).to(BF16)
dout = torch.randn(T, heads * 64, device=dev).to(BF16)
pos = torch.cat([torch.arange(n, dtype=torch.int32) for n in lens]).to(dev)
tab = ops.rope_table(256, 64, 10000.0, dev)
grp = [(0, 64, 64), (64, 128, 256)]
for w in (-1, 64):
    out, lse = ops.attn_fwd(qkv, cu, mask, 256, heads, w, validate=False, groups=grp)
    for mode in (1, 0):
        snx.configure(attn_bwd_onepass=mode)
        b = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, cu, mask, 256, heads, w, validate=False, groups=grp,
                                        rope_table=tab, pos=pos))
        print(f"fused + inverse RoPE window={w} {('two-kernel', 'one-pass')[mode]}: bwd {b:.1f} us", flush=True)
    snx.configure(attn_bwd_onepass=DEFAULT)

# queries alone: live blocks only (max_seqlen 64) vs 3 of 4 blocks exiting at once (max_seqlen 256)
cu = (torch.arange(65, dtype=torch.int32) * 64).to(dev)
T = 64 * 64
mask = torch.ones(T, dtype=torch.int64, device=dev)
qkv = torch.randn(T, 3 * heads * 64, device=dev).to(BF16)
dout = torch.randn(T, heads * 64, device=dev).to(BF16)
for ms in (64, 256):
    out, lse = ops.attn_fwd(qkv, cu, mask, ms, heads, 64, validate=False)
    f = timeit(lambda: ops.attn_fwd(qkv, cu, mask, ms, heads, 64, validate=False))
    b = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, cu, mask, ms, heads, 64, validate=False))
    print(f"queries only, max_seqlen={ms}: fwd {f:.1f} us, bwd {b:.1f} us", flush=True)
