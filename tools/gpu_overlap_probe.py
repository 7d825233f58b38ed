#!/usr/bin/env python3
"""Do global loads in flight make progress while the issuing waves compute?  Microbenchmark behind the attention-forward
experiment: every wave of 256 eight-wave workgroups requests 12 x 16 B per lane (96 KiB per workgroup and
round, contiguous or in the attention row layout with `attn`), then spins on v_fma / LDS reads, then waits.
Build here: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/experiments/overlap_probe.hip -o tools/_probe/libovl.so"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_probe", "libovl.so"))
dev = torch.device("cuda:0")
rounds, grid = 6, 256
ATT = 'attn' in sys.argv
stride = -1 if ATT else 12 * 512      # f4 elements per workgroup per round: 96 KiB (-1: the attention row layout)
src = torch.randn(grid * rounds * 12 * 512 * 4 + 1024, device=dev)
stamps = torch.zeros(grid * 3, dtype=torch.int64, device=dev)
sink = torch.zeros(4, device=dev)
for mode, name in ((0, "v_fma spin"), (2, "LDS reads")):
    for spin in (0, 30, 60, 120, 250, 500):
        for _ in range(2):
            rc = lib.ovl_run(mode, ctypes.c_void_p(src.data_ptr()), ctypes.c_long(stride), spin, ctypes.c_void_p(stamps.data_ptr()),
                             ctypes.c_void_p(sink.data_ptr()), rounds, grid, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0
        torch.cuda.synchronize()
        s = stamps.view(grid, 3).double().median(0).values / rounds
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            lib.ovl_run(mode, ctypes.c_void_p(src.data_ptr()), ctypes.c_long(stride), spin, ctypes.c_void_p(stamps.data_ptr()),
                        ctypes.c_void_p(sink.data_ptr()), rounds, grid, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        b.record()
        torch.cuda.synchronize()
        print(f"{name:>10} spin={spin:5d}: per round (wave 0, median over workgroups) issue {float(s[0]):7.0f}, spin {float(s[1]):7.0f}, "
              f"wait {float(s[2]):7.0f} cycles; kernel {a.elapsed_time(b) / 10 * 1e3:7.1f} us", flush=True)
