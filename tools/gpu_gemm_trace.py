#!/usr/bin/env python3
"""Per-workgroup timeline of the NT GEMM (diagnostics build only).

Build the library with the trace hooks first:
    SNX_EXTRA_HIPCC_FLAGS=-DSNX_GEMM_TRACE python opensearch-neural-pre-train_amd/snx/build.py --force
Every workgroup then records the constant-rate clock (100 MHz) at entry, after its K loop, after its last
store instruction and after the stores have left the wave, plus HW_ID / XCC_ID.  This tool runs one launch
per epilogue variant at the bench's token count and prints, per variant: mean phase lengths, the turn-around
between consecutive workgroups of one CU slot, and how the two workgroups of a CU overlap (time with 0/1/2
of them in the K loop)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opensearch-neural-pre-train_amd"))
import numpy as np
import torch
from snx import _lib, ops

dev = torch.device("cuda:0")
BF16 = torch.bfloat16
M = int(os.environ.get("M", 36864))
lib = _lib.lib()
if not hasattr(lib, "snx_gemm_trace_set"):
    sys.exit("libsnx.so was built without -DSNX_GEMM_TRACE")
lib.snx_gemm_trace_set.argtypes = [C.c_void_p]
lib.snx_gemm_trace_set.restype = C.c_int


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(BF16)


def analyse(name, buf, ntiles):
    torch.cuda.synchronize()
    a = buf[:ntiles * 8].cpu().numpy().astype(np.int64).reshape(ntiles, 8)
    t = a[:, :4].astype(np.float64)
    base = t[:, 0].min()
    t = (t - base) * 0.01                                # us
    hw = a[:, 4]
    xcc = a[:, 5] & 0xF
    cu = (xcc << 8) | ((hw >> 8) & 0xFF)                 # (xcc, se, sh, cu)
    total = t[:, 3].max()
    loop = t[:, 1] - t[:, 0]
    epi = t[:, 2] - t[:, 1]
    drain = t[:, 3] - t[:, 2]
    print(f"--- {name}: {ntiles} workgroups on {len(set(cu.tolist()))} CUs, launch span {total:.1f} us")
    print(f"    K loop {loop.mean():6.2f} us (p10 {np.percentile(loop, 10):.2f} p90 {np.percentile(loop, 90):.2f})"
          f"   epilogue issue {epi.mean():5.2f} (p90 {np.percentile(epi, 90):.2f})   store drain {drain.mean():5.2f}"
          f" (p90 {np.percentile(drain, 90):.2f})")
    gaps, both_loop, one_loop, none_loop, busy = [], 0.0, 0.0, 0.0, 0.0
    first = True
    for c in sorted(set(cu.tolist())):
        idx = np.nonzero(cu == c)[0]
        idx = idx[np.argsort(t[idx, 0])]
        # slot assignment: greedily to the slot that freed first
        ends = []
        for i in idx:
            if len(ends) < 2:
                ends.append(t[i, 3])
                continue
            k = int(np.argmin(ends))
            gaps.append(t[i, 0] - ends[k])
            ends[k] = t[i, 3]
        # overlap accounting on a 20-ns grid
        lo, hi = t[idx, 0].min(), t[idx, 3].max()
        grid = np.arange(lo, hi, 0.02)
        nloop = np.zeros_like(grid)
        nlive = np.zeros_like(grid)
        for i in idx:
            nloop += (grid >= t[i, 0]) & (grid < t[i, 1])
            nlive += (grid >= t[i, 0]) & (grid < t[i, 3])
        both_loop += float((nloop >= 2).sum()) * 0.02
        one_loop += float((nloop == 1).sum()) * 0.02
        none_loop += float((nloop == 0).sum()) * 0.02
        busy += hi - lo
        if first:
            first = False
            print(f"    CU {c:#x}: " + "  ".join(
                f"[{t[i, 0]:.1f} L {t[i, 1]:.1f} E {t[i, 2]:.1f} D {t[i, 3]:.1f}]" for i in idx[:10]))
    gaps = np.array(gaps) if gaps else np.zeros(1)
    print(f"    slot turn-around {gaps.mean():.2f} us (p90 {np.percentile(gaps, 90):.2f});  CU time with 2 / 1 / 0 "
          f"workgroups in the K loop: {100 * both_loop / busy:.0f} % / {100 * one_loop / busy:.0f} % / "
          f"{100 * none_loop / busy:.0f} %", flush=True)


def run(name, f, ntiles):
    buf = torch.zeros(ntiles * 8 + 8, dtype=torch.int64, device=dev)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    assert lib.snx_gemm_trace_set(C.c_void_p(buf.data_ptr())) == 0
    f()
    torch.cuda.synchronize()
    assert lib.snx_gemm_trace_set(C.c_void_p(0)) == 0
    analyse(name, buf, ntiles)


H, I = 768, 1152
tiles = lambda n: (M // 128) * ((n + 127) // 128)
x = rnd(M, H); w = rnd(H, H, scale=0.05); hin = torch.randn(M, H, device=dev)
x3 = rnd(M, 3 * H); w3 = rnd(H, 3 * H, scale=0.05)
y = rnd(M, I); wo = rnd(H, I, scale=0.05)
wqkv = rnd(3 * H, H, scale=0.05)
tab = ops.rope_table(256, 64, 160000.0, dev)
pos = torch.arange(256, dtype=torch.int32, device=dev).repeat(M // 256)
wi = rnd(2 * I, H, scale=0.05)
u = rnd(M, 2 * I); wot = rnd(I, H, scale=0.05)
run("store N=768 K=768", lambda: ops.gemm_nt(x, w), tiles(H))
run("store N=768 K=2304", lambda: ops.gemm_nt(x3, w3), tiles(H))
run("store N=2304 K=768", lambda: ops.gemm_nt(x, wqkv), tiles(3 * H))
run("resid N=768 K=768", lambda: ops.gemm_nt_resid(x, w, hin), tiles(H))
run("resid N=768 K=1152", lambda: ops.gemm_nt_resid(y, wo, hin), tiles(H))
run("rope N=2304 K=768", lambda: ops.gemm_nt_rope(x, wqkv, tab, pos, 2 * H, validate=False), tiles(3 * H))
run("geglu_fwd N=2304 K=768", lambda: ops.gemm_nt_geglu_fwd(x, wi), tiles(2 * I))
run("geglu_bwd N=1152 K=768", lambda: ops.gemm_nt_geglu_bwd(x, wot, u), tiles(I))
