"""Build-time guard for the asm LDS reads of the persistent kernels (snx/asmcheck.py; round-2 review, Weak #1).

The kernels read their MFMA fragments through ``asm volatile("ds_read_...")``; hipcc cannot see that the
destination registers are invalid until the source's ``s_waitcnt lgkmcnt``.  A build whose register allocator copies
or spills such a register inside that window is wrong at run time, intermittently.  ``snx/build.py`` therefore
re-assembles every guarded source and scans it; these tests pin the scanner and run it on the tree (hipcc
cross-compiles gfx950 without a GPU)."""
import os

import pytest

from snx import asmcheck as A

needs_hipcc = pytest.mark.skipif(not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")), reason="no hipcc")


def test_scanner_on_hand_written_streams():
    ok = ["ds_read_b128 v[12:15], v0", "v_mfma_f32_32x32x16_bf16 a[0:15], v[4:7], v[8:11], a[0:15]",
          "s_waitcnt lgkmcnt(0)", "v_accvgpr_write_b32 a200, v12"]
    assert A.scan_kernel(ok) == []
    spill = ["ds_read_b128 v[12:15], v0", "v_accvgpr_write_b32 a200, v13", "s_waitcnt lgkmcnt(0)"]
    assert len(A.scan_kernel(spill)) == 1
    scratch = ["ds_read_b64_tr_b16 v[20:21], v3", "scratch_store_dwordx2 off, v[20:21], off offset:8", "s_waitcnt lgkmcnt(0)"]
    assert len(A.scan_kernel(scratch)) == 1
    # a vmcnt-only wait retires nothing
    vm = ["ds_read_b128 v[12:15], v0", "s_waitcnt vmcnt(0)", "v_mov_b32 v1, v12", "s_waitcnt lgkmcnt(0)"]
    assert len(A.scan_kernel(vm)) == 1
    # counted wait: LDS reads return in order, all but the N newest are retired ...
    cnt = ["ds_read_b128 v[12:15], v0", "ds_read_b128 v[16:19], v0", "s_waitcnt lgkmcnt(1)", "v_mov_b32 v1, v12"]
    assert A.scan_kernel(cnt) == []
    assert len(A.scan_kernel(cnt[:3] + ["v_mov_b32 v1, v16"])) == 1
    # ... unless a scalar load (same counter, out of order) is outstanding
    sm = ["ds_read_b128 v[12:15], v0", "s_load_dwordx2 s[0:1], s[4:5], 0x0", "ds_read_b128 v[16:19], v0",
          "s_waitcnt lgkmcnt(1)", "v_mov_b32 v1, v12"]
    assert len(A.scan_kernel(sm)) == 1
    # overwriting a pending destination is as bad as reading it
    waw = ["ds_read_b128 v[12:15], v0", "v_mov_b32 v14, 0", "s_waitcnt lgkmcnt(0)"]
    assert len(A.scan_kernel(waw)) == 1
    # a read at the bottom of a loop whose wait sits at the loop top: the copy on the back-edge path is found
    loop = [".LBB0_1:", "s_waitcnt lgkmcnt(0)", "v_mov_b32 v1, v12", "v_accvgpr_write_b32 a9, v13", "ds_read_b128 v[12:15], v0",
            "s_cbranch_scc1 .LBB0_1"]
    assert A.scan_kernel(loop) == []
    loop_bad = [".LBB0_1:", "v_accvgpr_write_b32 a9, v13", "s_waitcnt lgkmcnt(0)", "ds_read_b128 v[12:15], v0",
                "s_cbranch_scc1 .LBB0_1"]
    assert len(A.scan_kernel(loop_bad)) == 1
    # the read's own address register may be reused at once
    addr = ["ds_read_b128 v[12:15], v0", "v_add_u32 v0, 64, v0", "s_waitcnt lgkmcnt(0)"]
    assert A.scan_kernel(addr) == []


def test_vm_rule_on_hand_written_streams():
    """The global-load / vmcnt twin of the scanner (gemm_nt_pipe.hip loads its epilogue operands through volatile asm and
    leaves the wait to a counted s_waitcnt vmcnt two K-steps later): vector-memory instructions retire in issue order, so
    vmcnt(N) retires a load unless it is among the N newest vm instructions -- loads, stores and LDS-DMA alike."""
    ok = ["global_load_dwordx4 v[24:27], v[44:45], off", "global_load_lds_dwordx4 v[4:5], off",
          "global_load_lds_dwordx4 v[4:5], off", "s_waitcnt vmcnt(2)", "v_lshlrev_b32_e32 v46, 16, v24"]
    assert A.scan_kernel_vm(ok) == []
    early = ok[:3] + ["s_waitcnt vmcnt(3)", "v_lshlrev_b32_e32 v46, 16, v24"]
    assert len(A.scan_kernel_vm(early)) == 1
    # a store behind the load counts like any vm instruction; an lgkmcnt wait retires nothing here
    st = ["global_load_dwordx4 v[24:27], v[44:45], off", "global_store_dwordx4 v[2:3], v[8:11], off",
          "s_waitcnt lgkmcnt(0)", "v_mov_b32_e32 v1, v25"]
    assert len(A.scan_kernel_vm(st)) == 1
    assert A.scan_kernel_vm(st[:2] + ["s_waitcnt vmcnt(1)", "v_mov_b32_e32 v1, v25"]) == []
    # spilling the destination before the wait is the failure the rule exists for; the address registers are free at once
    sp = ["global_load_dwordx4 v[24:27], v[44:45], off", "scratch_store_dwordx4 off, v[24:27], off", "s_waitcnt vmcnt(0)"]
    assert len(A.scan_kernel_vm(sp)) == 1
    assert A.scan_kernel_vm(["global_load_dwordx4 v[24:27], v[44:45], off", "v_add_u32_e32 v44, 64, v44",
                             "s_waitcnt vmcnt(0)"]) == []
    # a load at the bottom of a loop whose wait sits at the loop top: a use on the back-edge path is found
    loop = [".LBB0_1:", "v_mov_b32_e32 v1, v24", "s_waitcnt vmcnt(0)", "global_load_dwordx4 v[24:27], v[44:45], off",
            "s_cbranch_scc1 .LBB0_1"]
    assert len(A.scan_kernel_vm(loop)) == 1


def test_kernel_splitter_and_scratch_rule():
    asm = """
\t.text
_Z3foov:
\tds_read_b128 v[0:3], v4
\tv_mov_b32 v9, v1
\ts_waitcnt lgkmcnt(0)
\ts_endpgm
\t.section\t.rodata
\t.amdhsa_kernel _Z3foov
\t\t.amdhsa_private_segment_fixed_size 0
\t.end_amdhsa_kernel
\t.text
_Z3barv:
\tds_read_b128 v[0:3], v4
\ts_waitcnt lgkmcnt(0)
\tscratch_store_dword off, v1, off
\ts_endpgm
\t.section\t.rodata
\t.amdhsa_kernel _Z3barv
\t\t.amdhsa_private_segment_fixed_size 16
\t.end_amdhsa_kernel
"""
    ks = A.kernels_of(asm)
    assert set(ks) >= {"_Z3foov", "_Z3barv"} and len(ks["_Z3foov"]) == 4
    with pytest.raises(A.AsmGuardError, match="touch the destination"):
        A.check_asm(asm, ["foo"])
    with pytest.raises(A.AsmGuardError, match="scratch"):
        A.check_asm(asm, ["bar"])
    assert A.check_asm(asm, ["bar"], allow_scratch=True)["_Z3barv"]["scratch"] == 16


def test_back_edges_are_followed_in_assembler_text():
    """hipcc's loop labels (".LBB0_3:  ; =>This Inner Loop Header: Depth=1") must survive the splitter: a read at the
    bottom of a loop whose retiring wait sits at the loop top, and a copy of its destination on the back-edge path."""
    asm = """
\t.text
\t.globl\t_Z4loopv
_Z4loopv:                               ; @_Z4loopv
; %bb.0:
\ts_mov_b32 s0, 0
.LBB0_1:                                ; =>This Inner Loop Header: Depth=1
\tv_accvgpr_write_b32 a9, v13
\ts_waitcnt lgkmcnt(0)
\tds_read_b128 v[12:15], v0
\ts_add_i32 s0, s0, 1
\ts_cmp_lt_i32 s0, 8
\ts_cbranch_scc1 .LBB0_1
; %bb.2:
\ts_waitcnt lgkmcnt(0)
\ts_endpgm
.Lfunc_end0:
\t.section\t.rodata
\t.amdhsa_kernel _Z4loopv
\t\t.amdhsa_private_segment_fixed_size 0
\t.end_amdhsa_kernel
"""
    ks = A.kernels_of(asm)
    assert ".LBB0_1:" in ks["_Z4loopv"]
    with pytest.raises(A.AsmGuardError, match="touch the destination"):
        A.check_asm(asm, ["loop"])
    good = asm.replace("\tv_accvgpr_write_b32 a9, v13\n\ts_waitcnt lgkmcnt(0)\n",
                       "\ts_waitcnt lgkmcnt(0)\n\tv_accvgpr_write_b32 a9, v13\n")
    assert A.check_asm(good, ["loop"])["_Z4loopv"]["reads"] == 1
    with pytest.raises(A.AsmGuardError, match="no kernel matching"):
        A.check_asm(asm, ["nope"])


@needs_hipcc
def test_tree_passes_with_product_flags():
    rep = A.check_all()
    names = " ".join(rep)
    assert "gemm_tn256_kernel" in names and "decoder256_kernel" in names
    assert all(v["scratch"] == 0 and v["reads"] > 0 for v in rep.values()), rep


@needs_hipcc
def test_tree_passes_as_diagnostics_build():
    # README: SNX_EXTRA_HIPCC_FLAGS=-DSNX_GEMM_TRACE (in-kernel stamps; one 8-byte stamp lives in scratch)
    rep = A.check_all(["-DSNX_GEMM_TRACE"], allow_scratch=True)
    assert len(rep) >= 3


@needs_hipcc
def test_guard_fails_on_an_over_subscribed_variant():
    # all 256 AGPRs as accumulators (256-column decoder tile): hipcc spills -- the build must refuse it
    with pytest.raises(A.AsmGuardError):
        A.check_file("decoder256.hip", ["-DSNX_DEC256_OVERSUBSCRIBE"])


@needs_hipcc
def test_build_removes_the_object_of_a_failing_kernel(tmp_path, monkeypatch):
    """snx.build with flags that over-subscribe a guarded kernel: raises and leaves no object behind to link."""
    from snx import build as B
    obj = tmp_path / "_obj"
    obj.mkdir()
    monkeypatch.setattr(B, "OBJ", str(obj))
    monkeypatch.setattr(B, "LIB", str(tmp_path / "libsnx_test.so"))
    monkeypatch.setenv("SNX_EXTRA_HIPCC_FLAGS", "-DSNX_DEC256_OVERSUBSCRIBE")
    real_glob = B.glob.glob
    # only the guarded source needs compiling for this check
    monkeypatch.setattr(B.glob, "glob", lambda pat: [f for f in real_glob(pat) if not pat.endswith("*.hip")
                                                     or f.endswith("decoder256.hip")])
    with pytest.raises(A.AsmGuardError):
        B.build()
    assert not (obj / "decoder256.o").exists() and not (tmp_path / "libsnx_test.so").exists()


@needs_hipcc
def test_diagnostics_flavour_compiles_and_the_product_flavour_has_no_diagnostics_keys(tmp_path):
    """Two flavours of the library: the product build compiles the concluded experiments' defaults in and refuses their
    keys; -DSNX_DIAG (what the microbenchmark tools ask for) keeps them behind snx_configure.  The sources that carry
    such switches must compile both ways, and no source may read the environment."""
    import glob
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    for f in glob.glob(os.path.join(A.CSRC, "*.hip")) + glob.glob(os.path.join(A.CSRC, "*.h")):
        assert "getenv" not in open(f).read(), f

    def cc(name):
        out = os.path.join(tmp_path, name + ".o")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I", A.INCLUDE, "-I", A.CSRC,
                            "-DSNX_DIAG", "-c", os.path.join(A.CSRC, name), "-o", out], capture_output=True, text=True)
        return name, r.returncode, r.stderr[-2000:]
    with ThreadPoolExecutor(max_workers=4) as ex:
        for name, rc, err in ex.map(cc, ["config.hip", "gemm.hip", "gemm_nt256.hip", "gemm_tn256.hip"]):
            assert rc == 0, (name, err)
    import snx
    assert snx.config("nt256") == 1 and snx.config("attn_bwd_onepass") == 1
    with pytest.raises(snx.SnxError):
        snx.configure(nt256_dbg=1)                       # a diagnostics key: not in the product library
    with pytest.raises(snx.SnxError):
        snx.configure(no_such_key=1)
    with pytest.raises(snx.SnxError):
        snx.configure(nt256=7)                           # out of range
