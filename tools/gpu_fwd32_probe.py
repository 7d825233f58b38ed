#!/usr/bin/env python3
"""Stand-alone timing of tools/experiments/attention_fwd32.hip (the round-4 attention-forward experiment, not part of
libsnx.so): each tools/_probe/libf32_<tag>.so is that one source file compiled with other -D flags.
Build here: `python tools/gpu_fwd32_probe.py --build TAG[=FLAGS] ...` (e.g. `p= t_trace=-DSNX_ATTN_TRACE`); run with no
arguments on the GPU box.  A -DSNX_ATTN_TRACE build also prints the in-kernel timeline (s_memtime stamps kept in registers)."""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tools", "_probe")
CSRC = os.path.join(ROOT, "opensearch-neural-pre-train_amd", "csrc")
WRAP = r'''
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"
int attn_fwd_tile32(const bf16_t*, const int32_t*, const int64_t*, bf16_t*, float*, int, int, int, const int32_t*, hipStream_t);
extern "C" int probe_fwd(const void* qkv, const void* cu, const void* mask, void* out, void* lse, int T, int heads, int window,
                         const int32_t* groups, void* st) {
  return attn_fwd_tile32((const bf16_t*)qkv, (const int32_t*)cu, (const int64_t*)mask, (bf16_t*)out, (float*)lse, T, heads,
                         window, groups, (hipStream_t)st);
}
'''

if len(sys.argv) > 1 and sys.argv[1] == "--build":
    os.makedirs(PROBE, exist_ok=True)
    open(os.path.join(PROBE, "wrap.hip"), "w").write(WRAP)
    open(os.path.join(PROBE, "stub.hip"), "w").write('extern "C" int snx_get_reserved_cus() { return 0; }\n')
    for spec in sys.argv[2:]:
        tag, _, flags = spec.partition("=")
        src = os.path.join(ROOT, "tools", "experiments", "attention_fwd32.hip")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + CSRC,
               "-I" + os.path.join(ROOT, "include"), *flags.split(), src, os.path.join(CSRC, "config.hip"),
               os.path.join(PROBE, "wrap.hip"), os.path.join(PROBE, "stub.hip"), "-o", os.path.join(PROBE, f"libf32_{tag}.so")]
        subprocess.check_call(cmd)
        print("built", tag)
    sys.exit(0)

import torch
dev = torch.device("cuda:0")
BF16 = torch.bfloat16
heads = 12


def timeit(f, n=50, warm=5):
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


cases = {"fused": ([64] * 64 + [256] * 128, [(0, 64, 64), (64, 128, 256)]), "docs": ([256] * 128, [(0, 128, 256)]),
         "ragged": ([64] * 64 + [int(x) for x in torch.randint(100, 257, (128,), generator=torch.Generator().manual_seed(1))],
                    [(0, 64, 64), (64, 128, 256)])}
ref = {}
for path in sorted(glob.glob(os.path.join(PROBE, "libf32_*.so"))):
    lib = ctypes.CDLL(path)
    tag = os.path.basename(path)[7:-3]
    for name, (lens, grp) in cases.items():
        cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
        T = int(cu[-1])
        g = torch.Generator().manual_seed(5)
        qkv = torch.randn(T, 3 * heads * 64, generator=g).to(dev).to(BF16)
        mask = torch.ones(T, dtype=torch.int64, device=dev)
        out = torch.zeros(T, heads * 64, dtype=BF16, device=dev)
        lse = torch.zeros(heads, T, dtype=torch.float32, device=dev)
        garr = (ctypes.c_int32 * (1 + 3 * len(grp)))(len(grp), *[x for e in grp for x in e])
        for w in (-1, 64):
            def run():
                rc = lib.probe_fwd(ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(cu.data_ptr()), ctypes.c_void_p(mask.data_ptr()),
                                   ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(lse.data_ptr()), T, heads, w, garr,
                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                assert rc == 0, rc
            t = timeit(run)
            key = (name, w)
            if key not in ref:
                ref[key] = out.clone()
                d = ""
            else:
                d = f" max diff vs first {float((out.float() - ref[key].float()).abs().max()):.1e}"
            print(f"{tag:>12} {name:>6} window={w:>3}: {t:6.1f} us{d}", flush=True)
            if hasattr(lib, "snx_attn_fwd32_trace_set"):
                nb = sum(-(-n * heads // (4 // -(-ml // 64))) for _, n, ml in grp)
                buf = torch.zeros(max(nb, 256) * 48, dtype=torch.int64, device=dev)
                lib.snx_attn_fwd32_trace_set(ctypes.c_void_p(buf.data_ptr()))
                run()
                torch.cuda.synchronize()
                lib.snx_attn_fwd32_trace_set(ctypes.c_void_p(0))
                b = buf.view(-1, 48).cpu().double()
                b = b[b[:, 12] > 0]
                med = lambda x: float(x.median())   # noqa: E731
                if "tile" in tag:
                    print("      cycles (wave 0, second item): first QK " + f"{med(b[:, 2] - b[:, 0]):.0f} after entry; tiles "
                          + ", ".join(f"{med(b[:, i + 1] - b[:, i]):.0f}" for i in range(2, 10)), flush=True)
                    continue
                wv = b[:, 16:48].view(-1, 8, 4)
                print("      per wave, cycles after barrier 1 (median): " + "; ".join(
                    f"w{w}: requests {med(wv[:, w, 1] - wv[:, w, 0]):.0f} tiles {med(wv[:, w, 2] - wv[:, w, 1]):.0f} end {med(wv[:, w, 3] - wv[:, 0, 0]):.0f}"
                    for w in range(8)), flush=True)
                names = ["barrier 2", "qf copy + deposit", "barrier 1", "flush (stores)", "next requests", "tiles", "pack"]
                print("      cycles (wave 0, median): " + ", ".join(f"{n} {med(b[:, i + 2] - b[:, i + 1]):.0f}" for i, n in enumerate(names))
                      + f"; whole {med(b[:, 14] - b[:, 0]):.0f}; workgroup wall {med((b[:, 13] - b[:, 12]) * 10):.0f} ns; span "
                      f"{float(b[:, 13].max() - b[:, 12].min()) * 10 / 1e3:.1f} us for {b.shape[0]} workgroups", flush=True)
