#!/usr/bin/env python3
"""Which hipBLASLt kernels (macro-tile, in the kernel name) torch.matmul picks for the encoder's plain GEMM shapes.
Run under `rocprofv3 --kernel-trace --stats --output-format csv`; comparison only, never on the product path."""
import torch
dev = torch.device("cuda:0")
M = 36864
for (N, K) in [(768, 2304), (2304, 768), (768, 768), (1152, 768)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    for _ in range(10):
        torch.matmul(a, b.t())
torch.cuda.synchronize()
