#!/usr/bin/env python3
"""Does a just-WRITTEN operand come out of the Infinity Cache?  (round 6, profiles/r06_experiments.txt section 9)

The K = 2304 dX GEMMs take 117 us back to back and ~139 us inside the step.  Their A operand (170 MB) has just been written by
the previous kernel.  Cases, each timed as [prepare A] -> [GEMM] with HIP events around the GEMM only, 15 rounds, median:
  warm      : the GEMM ran on the same A just before (A read-allocated)
  written   : A rewritten by a copy kernel (170 MB read + 170 MB write) right before the GEMM
  written_nt: ... where the copy's SOURCE is a different 170 MB buffer each time (so only the WRITE can have cached A)
  flushed   : a 1 GB stream between the write and the GEMM
    python tools/gpu_mall_probe.py"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opensearch-neural-pre-train_amd"))
import torch  # noqa: E402
from snx import ops  # noqa: E402

dev = torch.device("cuda:0")
M, N, K = 36864, 768, 2304


def main():
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    srcs = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(4)]
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    junk = torch.empty(512 * 1024 * 1024, dtype=torch.uint8, device=dev)
    junk2 = torch.empty_like(junk)
    ev = lambda: torch.cuda.Event(enable_timing=True)   # noqa: E731

    def loop(prepare, with_gemm, n=15):
        # no event and no host sync between the kernels of a case (a timing event is a system-scope fence: it writes back and
        # invalidates the L2, so a kernel timed on its own ALWAYS starts on a cold L2): the GEMM's time is the difference
        # between the loop with it and the loop without it
        for r in range(4):
            prepare(r)
            if with_gemm:
                ops.gemm_nt(a, w)
        e0, e1 = ev(), ev()
        e0.record()
        for r in range(n):
            prepare(r)
            if with_gemm:
                ops.gemm_nt(a, w)
        e1.record()
        torch.cuda.synchronize()
        return 1000.0 * e0.elapsed_time(e1) / n

    def timed(prepare):
        with_g = [loop(prepare, True) for _ in range(5)]
        without = [loop(prepare, False) for _ in range(5)]
        return statistics.median(with_g) - statistics.median(without), min(with_g) - min(without)

    ops.gemm_nt(a, w)
    cases = {
        "GEMM only, back to back": lambda r: None,
        "warm (A read by the previous GEMM)": lambda r: ops.gemm_nt(a, w),
        "LayerNorm-sized stream (450 MB copy) in front": lambda r: junk2[:225 * 1024 * 1024].copy_(junk[:225 * 1024 * 1024]),
        "A just written (copy from a rotating source)": lambda r: a.copy_(srcs[r % 4]),
        "A written, then 1 GB streamed": lambda r: (a.copy_(srcs[r % 4]), junk2.copy_(junk)),
        "1 GB streamed (A last read long ago)": lambda r: junk2.copy_(junk),
    }
    for name, prep in cases.items():
        med, mn = timed(prep)
        print(f"{name:48s} median {med:7.1f} us  min {mn:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
