#!/usr/bin/env python3
"""A few launches of the attention backward on the document shape (128 x 256 tokens, 12 heads) for rocprofv3 counter
passes: `rocprofv3 --pmc ... -- python3 tools/gpu_attn_probe.py [window] [onepass 0|1]`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opensearch-neural-pre-train_amd"))
import torch
from snx import ops
from snx._lib import fn, check

dev = torch.device("cuda:0")
window = int(sys.argv[1]) if len(sys.argv) > 1 else -1
check(fn("snx_attn_configure")(int(sys.argv[2]) if len(sys.argv) > 2 else 1), "snx_attn_configure")
heads = 12
lens = [256] * 128
cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
T = int(cu[-1])
mask = torch.ones(T, dtype=torch.int64, device=dev)
qkv = torch.randn(T, 3 * heads * 64, device=dev).to(torch.bfloat16)
dout = torch.randn(T, heads * 64, device=dev).to(torch.bfloat16)
out, lse = ops.attn_fwd(qkv, cu, mask, 256, heads, window, validate=False)
for _ in range(6):
    ops.attn_bwd(qkv, out, dout, lse, cu, mask, 256, heads, window, validate=False)
torch.cuda.synchronize()
