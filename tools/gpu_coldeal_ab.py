#!/usr/bin/env python3
"""nt256 leftover dealing A/B on one GPU in one process (round 6): column-run dealing (nt256_coldeal = 1, default) against
the tile-order dealing (0) -- plain store at the encoder's N = 768 shapes (432 tiles on 256 workgroups = 1.69 rounds) and
the N = 2304 shapes; interleaved timing rounds, median / minimum per arm (HIP events, 20 launches per sample).

    python tools/gpu_coldeal_ab.py"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opensearch-neural-pre-train_amd"))
import torch  # noqa: E402
from snx import ops  # noqa: E402
from snx._lib import fn  # noqa: E402

dev = torch.device("cuda:0")
M = 36864


def time_us(f, n=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f(); torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return 1000.0 * a.elapsed_time(b) / n


def main():
    fn("snx_nt256_configure")(2, 1024)
    tab = ops.rope_table(256, 64, 160000.0, dev)
    pos = (torch.arange(M, dtype=torch.int32, device=dev) % 256).contiguous()
    rows = ops.rope_rows(tab, pos)
    for N, K, kind in ((768, 768, "store"), (768, 1152, "store"), (768, 2304, "store"), (2304, 768, "store"),
                       (2304, 768, "rope"), (2304, 768, "geglu_fwd")):
        x = (torch.randn(M, K, device=dev)).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        if kind == "store":
            f = lambda: ops.gemm_nt(x, w)                                            # noqa: E731
        elif kind == "rope":
            f = lambda: ops.gemm_nt_rope_rows(x, w, tab, pos, rows, 1536)            # noqa: E731
        else:
            f = lambda: ops.gemm_nt_geglu_fwd(x, w)                                  # noqa: E731
        samples = {0: [], 1: []}
        for r in range(7):
            for arm in (0, 1):
                assert fn("snx_configure")(b"nt256_coldeal", arm) == 0
                samples[arm].append(time_us(f))
        fl = 2.0 * M * N * K
        line = f"N={N} K={K} {kind:9s}"
        for arm in (0, 1):
            med, mn = statistics.median(samples[arm]), min(samples[arm])
            line += f" | coldeal={arm}: median {med:7.1f} us min {mn:7.1f} us ({fl / med / 1e9:.3f} PFLOP/s)"
        print(line, flush=True)
    fn("snx_configure")(b"nt256_coldeal", 1)
    fn("snx_nt256_configure")(1, 8192)


if __name__ == "__main__":
    main()
