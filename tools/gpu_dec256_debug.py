#!/usr/bin/env python3
"""Structured-input check of the decoder kernels: Hd rows are one-hot at k = t % K and W[v, k] = k % 128 + 1, so the
row maximum of token t must be (t % K) % 128 + 1 -- a wrong value names the row (or k chunk) that was read instead."""
import os, sys
sys.path.insert(0, "/root/repo/opensearch-neural-pre-train_amd")
import torch
from snx import ops
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
B, S, V, K = 12, 256, 256, 768
T = B * S
hd = torch.zeros(T, K)
hd[torch.arange(T), torch.arange(T) % K] = 1.0
hd = hd.to(BF16).to(dev)
# W[v, k] = 1 + k + v/1024  -> logit[t, v] identifies k = t % K (integer part) exactly in bf16? use small ints
W = (torch.arange(K).float()[None, :] % 128 + 1).repeat(V, 1)
W = W.to(BF16).to(dev)
bias = torch.zeros(V, device=dev)
mask = torch.ones(B, S, dtype=torch.int64, device=dev)
cu = (torch.arange(B + 1, dtype=torch.int32) * S).to(dev)
sp, keys, tw = ops.decoder_splade_fwd(hd, W, bias, cu, mask.reshape(-1), S)
val = torch.expm1(tw)           # = (t % K) % 128 + 1 expected
exp = ((torch.arange(T) % K) % 128 + 1).float().to(dev)
bad = (val - exp).abs() > 0.51
print("bad rows frac", bad.float().mean().item())
idx = bad.nonzero().view(-1)
print("bad rows (first 60):", idx[:60].tolist())
print("val:", [round(x) for x in val[idx[:60]].tolist()])
print("exp:", [round(x) for x in exp[idx[:60]].tolist()])
print("bad rows by tile", [round(bad[i*256:(i+1)*256].float().mean().item(), 2) for i in range(B)])
print("bad by (t%256)//32", [round(bad[(torch.arange(T, device=dev) % 256)//32 == q].float().mean().item(),3) for q in range(8)])
