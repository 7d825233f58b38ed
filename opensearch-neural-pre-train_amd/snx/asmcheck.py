"""Build-time guard for the kernels that read LDS through volatile inline asm.

The persistent kernels csrc/gemm_tn256.hip and csrc/decoder256.hip issue their fragment reads as
``asm volatile("ds_read_...")`` so that hipcc's waitcnt pass does not drain the LDS-DMA ring in front of every
read (csrc/gemm_nt256.hip reads its fragments through plain C++ loads, which hipcc tracks itself: it is guarded for
the zero-scratch rule, and the scan finds nothing to object to there).  The price: the compiler does not know that the destination registers are not valid until the
``s_waitcnt lgkmcnt`` the SOURCE places behind them.  If register pressure makes it spill or copy such a register
(``v_accvgpr_write``, ``scratch_store``, ``v_mov``) between the read and that wait, the copy is taken before the data
has arrived and the kernel returns intermittently wrong rows -- seen in round 2 on builds with different flags.

``check_file`` compiles a source to gfx950 assembly with the given flags and fails unless, for every guarded kernel,
  * no scratch is used (``.amdhsa_private_segment_fixed_size 0``), and
  * no instruction between a ``ds_read*`` and the wait that retires it mentions the read's destination registers.
LDS reads return in order, so ``lgkmcnt(N)`` retires all but the N newest; while a scalar load is outstanding (it
shares the counter and may return out of order) only ``lgkmcnt(0)`` retires anything.
"""
from __future__ import annotations

import os
import re
import subprocess
import tempfile
from typing import Dict, Iterable, List, Tuple

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(os.path.dirname(PKG), "include")

# source file -> substrings of the (mangled) kernel names that must pass
GUARDED: Dict[str, Tuple[str, ...]] = {
    "gemm_tn256.hip": ("gemm_tn256_kernel",),
    "decoder256.hip": ("decoder256_kernel",),
    "gemm_nt256.hip": ("gemm_nt256_kernel",),
}

# Kernels held to the global-load / vmcnt rule: nothing may touch a load's destination before a vmcnt wait that retires it.
# Written for the first form of gemm_nt_pipe.hip (epilogue operands loaded by volatile-asm global loads, the wait left to
# a counted s_waitcnt vmcnt two K-steps later: the rule found a missing wait in its drain code at the first build); the
# helper-wave form loads through plain C++ again and passes trivially, with zero scratch.
VM_GUARDED: Dict[str, Tuple[str, ...]] = {
    "gemm_nt_pipe.hip": ("gemm_nt_geglu_bwd_pipe_kernel",),
    # the resident attention forward requests its key mask through asm volatile("global_load_dwordx2") and waits for it
    # with an asm s_waitcnt vmcnt(0) further down (round-4 advisor finding: nothing protected other flag sets)
    "attention_unit.hip": ("attn_fwd_unit_kernel",),
}

_REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")
_VMCNT = re.compile(r"vmcnt\((\d+)\)")
_VM_OPS = ("global_load", "global_store", "global_atomic", "flat_load", "flat_store", "flat_atomic", "scratch_load",
           "scratch_store", "buffer_load", "buffer_store", "buffer_atomic")
_LGKM = re.compile(r"lgkmcnt\((\d+)\)")


class AsmGuardError(RuntimeError):
    pass


def _regs(text: str) -> set:
    out = set()
    for m in _REG.finditer(text):
        if m.group(2) is not None:
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(1), i) for i in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


def kernels_of(asm: str) -> Dict[str, List[str]]:
    """name -> instruction lines of every function in a gfx950 .s file."""
    out: Dict[str, List[str]] = {}
    cur = None
    for line in asm.splitlines():
        # global symbols AND hipcc's local labels (".LBB0_12:", often followed by "; =>This Inner Loop Header ...")
        m = re.match(r"^(\.?[A-Za-z_][\w.$]*):\s*(;.*)?$", line)
        if m and not m.group(1).startswith(".L"):
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is None:
            continue
        if m:                                # a local label: kept, so that the scan can follow back-edges
            out[cur].append(m.group(1) + ":")
            continue
        s = line.split(";", 1)[0].strip()
        if s.startswith(".end_amdhsa_kernel") or s.startswith(".section") or s.startswith(".Lfunc_end"):
            if s.startswith(".Lfunc_end") or s.startswith(".section"):
                cur = None
            continue
        if not s or s.startswith("."):
            continue
        out[cur].append(s)
    return out


def _scan(lines: List[str], start: int, stop: int, pending: List[Tuple[set, str]], smem: int, bad: List[str],
          labels: Dict[str, int], follow: bool) -> None:
    """Linear scan of lines[start:stop] from the state (pending, smem).  With `follow`, every BACKWARD branch (loop
    back-edge) met with reads still pending re-scans its loop body once from the branch's state: a read issued at the
    bottom of a loop whose retiring wait sits at the loop top is then seen by a copy or spill on the back-edge path."""
    i = start
    while i < stop:
        ins = lines[i]
        i += 1
        if ins.endswith(":"):
            continue
        op = ins.split()[0]
        if op == "s_waitcnt":
            m = _LGKM.search(ins)
            if m is None and "lgkmcnt" not in ins and not re.search(r"s_waitcnt\s+(0x[0-9a-f]+|\d+)\s*$", ins):
                continue                     # vmcnt / expcnt only
            n = int(m.group(1)) if m else 0
            if n == 0:
                pending, smem = [], 0
            elif smem == 0 and len(pending) > n:
                pending = pending[len(pending) - n:]
            continue
        if op.startswith("s_load") or op.startswith("s_buffer_load"):
            smem += 1
            continue
        if follow and pending and (op.startswith("s_cbranch") or op == "s_branch"):
            tgt = labels.get(ins.split()[-1])
            if tgt is not None and tgt < i:
                _scan(lines, tgt, i - 1, list(pending), smem, bad, labels, False)
            continue
        touched = _regs(ins.split(None, 1)[1]) if " " in ins else set()
        if pending and touched:
            for dst, rd in pending:
                if dst & touched:
                    bad.append(f"{ins}   <-   {rd}")
                    break
        if op.startswith("ds_read") or op.startswith("ds_load"):
            ops = ins.split(None, 1)[1]
            pending.append((_regs(ops.split(",")[0]), ins))


def _scan_vm(lines: List[str], start: int, stop: int, pending: List[Tuple[set, str, int]], issued: int, bad: List[str],
             labels: Dict[str, int], follow: bool) -> None:
    """The vm-counter twin of _scan: vector-memory instructions retire in issue order, so `s_waitcnt vmcnt(N)` retires
    every load that is not among the N newest vm instructions (loads, stores, LDS-DMA, scratch alike)."""
    i = start
    while i < stop:
        ins = lines[i]
        i += 1
        if ins.endswith(":"):
            continue
        op = ins.split()[0]
        if op == "s_waitcnt":
            m = _VMCNT.search(ins)
            if m is None and re.search(r"s_waitcnt\s+(0x[0-9a-f]+|\d+)\s*$", ins):
                n = 0                        # an immediate without names: every counter
            elif m is None:
                continue
            else:
                n = int(m.group(1))
            pending = [p for p in pending if issued - 1 - p[2] < n]
            continue
        if follow and pending and (op.startswith("s_cbranch") or op == "s_branch"):
            tgt = labels.get(ins.split()[-1])
            if tgt is not None and tgt < i:
                _scan_vm(lines, tgt, i - 1, list(pending), issued, bad, labels, False)
            continue
        touched = _regs(ins.split(None, 1)[1]) if " " in ins else set()
        if pending and touched:
            for dst, rd, _ in pending:
                if dst & touched:
                    bad.append(f"{ins}   <-   {rd}")
                    break
        if op.startswith(_VM_OPS):
            if op.startswith("global_load") and "_lds_" not in op:
                pending.append((_regs(ins.split(None, 1)[1].split(",")[0]), ins, issued))
            issued += 1


def scan_kernel_vm(lines: Iterable[str]) -> List[str]:
    """Violations: an instruction touches the destination of a global load before a vmcnt wait that retires the load.
    hipcc places such waits for the loads it can see; the scan exists for loads issued through volatile asm."""
    lines = list(lines)
    labels = {ln[:-1]: i for i, ln in enumerate(lines) if ln.endswith(":")}
    bad: List[str] = []
    _scan_vm(lines, 0, len(lines), [], 0, bad, labels, True)
    return list(dict.fromkeys(bad))


def scan_kernel(lines: Iterable[str]) -> List[str]:
    """Violations: 'instruction <- ds_read' pairs where a pending LDS-read destination is touched too early.  The scan is
    linear in text order (forward branches: the fall-through path and the taken path both lie ahead in the text, so a
    read pending at the branch stays pending at the target) plus one pass over every loop body entered through its
    back-edge; it does not enumerate paths through nested or irreducible control flow."""
    lines = list(lines)
    labels = {ln[:-1]: i for i, ln in enumerate(lines) if ln.endswith(":")}
    bad: List[str] = []
    _scan(lines, 0, len(lines), [], 0, bad, labels, True)
    return list(dict.fromkeys(bad))


def check_asm(asm: str, names: Iterable[str], where: str = "", allow_scratch: bool = False, vm: bool = False) -> Dict[str, dict]:
    """Raises AsmGuardError on a violation; returns {kernel: {'scratch':, 'reads':}} otherwise.  allow_scratch:
    diagnostics builds (-DSNX_GEMM_TRACE keeps a time stamp in scratch) are held to the scan alone.  vm: the
    global-load / vmcnt rule (VM_GUARDED) instead of the LDS-read / lgkmcnt one."""
    ks = kernels_of(asm)
    report = {}
    for want in names:
        hits = [k for k in ks if want in k]
        if not hits:
            raise AsmGuardError(f"{where}: no kernel matching {want!r} in the assembly")
        for k in hits:
            m = re.search(re.escape(k) + r"\n(?:.*\n)*?\s*\.amdhsa_private_segment_fixed_size (\d+)", asm)
            scratch = int(m.group(1)) if m else -1
            if scratch != 0 and not allow_scratch:
                raise AsmGuardError(f"{where}: {k} uses {scratch} bytes of scratch per lane (spills to memory): the "
                                    "asm LDS reads are only safe in a kernel whose registers all stay in the register file")
            bad = scan_kernel_vm(ks[k]) if vm else scan_kernel(ks[k])
            if bad and vm:
                raise AsmGuardError(f"{where}: {k}: {len(bad)} instruction(s) touch the destination of a global load before "
                                    "a s_waitcnt vmcnt that retires it (a copy/spill the compiler placed behind an asm "
                                    "global_load):\n  " + "\n  ".join(bad[:8]))
            if bad:
                raise AsmGuardError(f"{where}: {k}: {len(bad)} instruction(s) touch the destination of an LDS read before "
                                    "the s_waitcnt lgkmcnt that retires it (a spill/copy the compiler placed behind an asm "
                                    "ds_read):\n  " + "\n  ".join(bad[:8]))
            report[k] = {"scratch": scratch, "reads": sum(1 for i in ks[k] if i.startswith(("ds_read", "ds_load")))}
    return report


def compile_asm(src: str, extra_flags: Iterable[str] = ()) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I", INCLUDE, "-I", CSRC,
               "--cuda-device-only", "-S", "-Wno-unused-command-line-argument"] + list(extra_flags) + [src, "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"asm build failed: {' '.join(cmd)}\n{r.stderr}")
        with open(out) as f:
            return f.read()


def check_file(basename: str, extra_flags: Iterable[str] = (), allow_scratch: bool = False) -> Dict[str, dict]:
    src = os.path.join(CSRC, basename)
    flags = list(extra_flags)
    where = f"{basename} {' '.join(flags)}".strip()
    asm = compile_asm(src, flags)
    rep = {}
    if basename in GUARDED:
        rep.update(check_asm(asm, GUARDED[basename], where, allow_scratch))
    if basename in VM_GUARDED:
        rep.update(check_asm(asm, VM_GUARDED[basename], where, allow_scratch, vm=True))
    return rep


def all_guarded() -> List[str]:
    return sorted(set(GUARDED) | set(VM_GUARDED))


def check_all(extra_flags: Iterable[str] = (), allow_scratch: bool = False) -> Dict[str, dict]:
    rep = {}
    for b in all_guarded():
        if os.path.exists(os.path.join(CSRC, b)):
            rep.update(check_file(b, extra_flags, allow_scratch))
    return rep


if __name__ == "__main__":
    import json
    import sys
    flags = [a for a in sys.argv[1:] if a != "--allow-scratch"]
    print(json.dumps(check_all(flags, "--allow-scratch" in sys.argv), indent=1))
