"""Build libsnx.so (the C-ABI HIP library) in-tree for gfx950.

hipcc cross-compiles every ``csrc/*.hip`` to an object (no GPU needed); the objects are linked
with g++ against the SAME libamdhip64 that PyTorch-ROCm loads (``torch/lib``), so that streams
and device pointers handed over from torch belong to the one HIP runtime in the process.  When
torch is absent the system ROCm runtime is used (stand-alone C/C++ hosts).

    python -m snx.build            # or: python opensearch-neural-pre-train_amd/snx/build.py
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # `snx` importable when run as a script
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
CSRC = os.path.join(PKG, "csrc")
ROOT = os.path.dirname(PKG)
INCLUDE = os.path.join(ROOT, "include")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(PKG, "snx", "libsnx.so")
ARCH = "gfx950"


def _torch_lib_dir():
    try:
        import torch
        d = os.path.join(os.path.dirname(torch.__file__), "lib")
        if os.path.exists(os.path.join(d, "libamdhip64.so")):
            return d
    except Exception:
        pass
    return None


def _newer(src_list, out):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _newer([s] + headers, o):
            jobs.append([hipcc, f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-I", INCLUDE, "-I", CSRC]
                        + os.environ.get("SNX_EXTRA_HIPCC_FLAGS", "").split() + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"build failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    # Guard (snx/asmcheck.py): the persistent kernels read LDS through volatile asm; a build in which hipcc copies or
    # spills one of those destination registers before the source's s_waitcnt returns intermittently wrong rows.
    # Every (re)compiled guarded source is re-assembled with the same flags and scanned; a failing object is removed
    # so that no library can be linked from it.  Zero scratch is required too, except in the documented diagnostics
    # build (-DSNX_GEMM_TRACE keeps an 8-byte time stamp in scratch), which is held to the scan alone.
    from snx import asmcheck
    extra = os.environ.get("SNX_EXTRA_HIPCC_FLAGS", "").split()
    rebuilt = {os.path.basename(j[-3]) for j in jobs}
    for b in sorted(rebuilt & set(asmcheck.GUARDED)):
        try:
            asmcheck.check_file(b, extra, allow_scratch="-DSNX_GEMM_TRACE" in extra)
        except asmcheck.AsmGuardError:
            o = os.path.join(OBJ, b[:-4] + ".o")
            if os.path.exists(o):
                os.remove(o)
            raise
    if force or jobs or _newer(objs, LIB):
        tl = _torch_lib_dir()
        if tl:
            link = ["g++", "-shared", "-o", LIB] + objs + ["-L", tl, "-lamdhip64", f"-Wl,-rpath,{tl}"]
        else:
            link = ["g++", "-shared", "-o", LIB] + objs + ["-L", "/opt/rocm/lib", "-lamdhip64",
                                                            "-Wl,-rpath,/opt/rocm/lib"]
        run(link)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
