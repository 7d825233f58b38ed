#!/usr/bin/env python3
"""The 256x256 persistent NT GEMM (csrc/gemm_nt256.hip) against the 128x128 kernel (csrc/gemm.hip) on one GPU, in
one process: (1) results equal bit for bit on shapes that exercise short tiles, ragged M, half-empty column tiles;
(2) interleaved timing rounds at the encoder's shapes (M = 36,864 token rows), median and minimum per arm.

    python tools/gpu_nt256.py [check] [time]
"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opensearch-neural-pre-train_amd"))
import torch  # noqa: E402
from snx import ops  # noqa: E402
from snx._lib import fn  # noqa: E402

dev = torch.device("cuda:0")
BF16 = torch.bfloat16
H, I = 768, 1152


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(BF16)


def arm(on, min_m=1024):
    fn("snx_nt256_configure")(2 if on else 0, min_m)       # 2 = every eligible shape, whatever the default policy


def variants(M, N, K):
    """name -> callable returning the tensors to compare"""
    x = rnd(M, K)
    w = rnd(N, K, scale=0.05)
    out = {"store": lambda: (ops.gemm_nt(x, w),)}
    hin = torch.randn(M, N, device=dev)
    out["resid"] = lambda: (ops.gemm_nt_resid(x, w, hin),)
    if N % 64 == 0 and os.environ.get("NT256_EPIS", "1") == "1":
        S = 256
        tab = ops.rope_table(S, 64, 160000.0, dev)
        pos = (torch.arange(M, dtype=torch.int32, device=dev) % S).contiguous()
        rc = (2 * N // 3) // 64 * 64
        out["rope"] = lambda: (ops.gemm_nt_rope(x, w, tab, pos, rc, validate=False),)
        rows = ops.rope_rows(tab, pos)
        out["rope_rows"] = lambda: (ops.gemm_nt_rope_rows(x, w, tab, pos, rows, rc),)
        out["geglu_fwd"] = lambda: ops.gemm_nt_geglu_fwd(x, w)
        if N % 32 == 0:
            u = rnd(M, 2 * N)
            out["geglu_bwd"] = lambda: (ops.gemm_nt_geglu_bwd(x, w, u),)
    return out


def check():
    shapes = [(8192, 768, 768), (36864, 2304, 768), (9000, 1152, 768), (8200, 64, 64), (12345, 768, 128),
              (20000, 2304, 192), (36864, 768, 2304), (4100, 320, 256), (16384 + 64, 512, 64)]
    bad = 0
    for M, N, K in shapes:
        torch.manual_seed(M + N + K)
        for name, f in variants(M, N, K).items():
            arm(0)
            ref = [t.clone() for t in f()]
            arm(1, 1024)
            got = f()
            torch.cuda.synchronize()
            ok = all(torch.equal(a, b) for a, b in zip(ref, got))
            if not ok:
                bad += 1
                d = [(a.float() - b.float()).abs() for a, b in zip(ref, got)]
                rows = [int((x.reshape(x.shape[0], -1).amax(1) > 0).sum()) for x in d]
                first = [int((x.reshape(x.shape[0], -1).amax(1) > 0).nonzero()[0]) if r else -1 for x, r in zip(d, rows)]
                print(f"MISMATCH {name} M={M} N={N} K={K}: max {[float(x.max()) for x in d]} rows {rows} first {first}",
                      flush=True)
            else:
                print(f"ok   {name:10s} M={M} N={N} K={K}", flush=True)
    # the 128x128 kernel itself against fp32 torch on one shape, so that "equal to it" means something
    M, N, K = 8192, 768, 768
    x, w = rnd(M, K), rnd(N, K, scale=0.05)
    arm(1, 1024)
    c = ops.gemm_nt(x, w).float()
    r = (x.float() @ w.float().t())
    err = float((c - r).abs().max() / r.abs().max())
    print(f"vs fp32 torch: rel max err {err:.3e}", flush=True)
    bad += err > 1e-2
    print("CHECK", "FAILED" if bad else "PASSED", flush=True)
    return bad


def timeit(f, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def time_all():
    M = int(os.environ.get("M", 36864))
    cases = [("store", 768, 768), ("store", 768, 2304), ("store", 2304, 768), ("resid", 768, 768), ("resid", 768, 1152),
             ("rope", 2304, 768), ("rope_rows", 2304, 768), ("geglu_fwd", 2304, 768), ("geglu_bwd", 1152, 768), ("store", 1152, 768)]
    only = os.environ.get("CASES")                       # e.g. CASES=store:2304:768,rope:2304:768
    if only:
        cases = [(c.split(":")[0], int(c.split(":")[1]), int(c.split(":")[2])) for c in only.split(",")]
    rounds, n = int(os.environ.get("ROUNDS", 7)), 20
    print(f"{'variant':24s} {'128x128 med/min us':>22s} {'256x256 med/min us':>22s}  TFLOP/s(256, med)", flush=True)
    for name, N, K in cases:
        torch.manual_seed(1)
        fs = variants(M, N, K)
        if name not in fs:
            continue
        f = fs[name]
        t = {0: [], 1: []}
        for on in (0, 1):
            arm(on, 1024)
            for _ in range(3):
                f()
        for _ in range(rounds):
            for on in (0, 1):
                arm(on, 1024)
                t[on].append(timeit(f, n))
        m0, m1 = statistics.median(t[0]), statistics.median(t[1])
        print(f"{name + f' N={N} K={K}':24s} {m0:10.1f} /{min(t[0]):8.1f}   {m1:10.1f} /{min(t[1]):8.1f}   "
              f"{2.0 * M * N * K / m1 / 1e6:8.1f}", flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "time"]
    rc = 0
    if "check" in what:
        rc = check()
    if "time" in what:
        time_all()
    sys.exit(1 if rc else 0)
