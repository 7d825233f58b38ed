#!/usr/bin/env python3
"""Run one GEMM shape a few times (for rocprofv3 --pmc)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opensearch-neural-pre-train_amd"))
import torch
from snx import ops
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
M, N, K = 16384, 2304, 768
a = torch.randn(M, K, device=dev).to(BF16); b = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
dy = (torch.randn(M, N, device=dev) * 0.1).to(BF16); dw = torch.zeros(N, K, device=dev)
for _ in range(5):
    ops.gemm_nt(a, b)
    ops.gemm_tn_accum(dy, a, dw)
torch.cuda.synchronize()
