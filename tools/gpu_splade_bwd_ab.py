#!/usr/bin/env python3
"""A/B of the routed decoder backward's dHd gather (csrc/splade_head.hip): wave-per-row sweep against the panel-paced
form, at the bench's worst case (random init: every (sequence, vocabulary) entry active).  Checks dHd bit for bit."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opensearch-neural-pre-train_amd"))
import torch  # noqa: E402
import snx
from snx import ops  # noqa: E402
from snx._lib import check, fn  # noqa: E402
from snx.ops import _p, _stream  # noqa: E402

dev = torch.device("cuda:0")
B, S, V, H = int(os.environ.get("B", 192)), int(os.environ.get("S", 256)), 50000, 768
T = B * S
g = torch.Generator().manual_seed(1)
hd = (torch.randn(T, H, generator=g) * 1.0).to(torch.bfloat16).to(dev)
W = (torch.randn(V, H, generator=g) * 0.02).to(torch.bfloat16).to(dev)
bias = torch.zeros(V, device=dev)
gs = torch.randn(B, V, generator=g).to(dev)
cu = (torch.arange(B + 1, dtype=torch.int32) * S).to(dev)
mask = torch.ones(T, dtype=torch.int64, device=dev)
sp, keys, tw = ops.decoder_splade_fwd(hd, W, bias, cu, mask, S)
print("active entries:", int((sp > 0).sum()), "of", B * V, flush=True)
scratch = torch.empty(fn("snx_splade_bwd_scratch_bytes")(B, S, V), dtype=torch.uint8, device=dev)
out = {}
for mode in os.environ.get("MODES", "0,32,16,64,0,32").split(","):
    snx.configure(splade_dh_panels=int(mode))
    dHd = torch.full((T, H), float("nan"), dtype=torch.bfloat16, device=dev)
    gE = torch.zeros(V, H, device=dev)
    gb = torch.zeros(V, device=dev)

    def run():
        check(fn("snx_splade_bwd")(_p(gs), _p(keys), _p(hd), _p(W), _p(cu), _p(dHd), _p(gE), _p(gb), _p(scratch), T, B, S, V,
                                   H, _stream()), "snx_splade_bwd")
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    print(f"splade_dh_panels={mode}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per call (dW + bucket + dHd)", flush=True)
    out[mode] = dHd.clone()
same = all(torch.equal(out["0"].view(torch.int16), v.view(torch.int16)) for v in out.values())
print("dHd bit-identical:", same, flush=True)
sys.exit(0 if same else 1)
