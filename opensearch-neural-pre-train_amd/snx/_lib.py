"""ctypes loader for libsnx.so (the C-ABI HIP library, see include/snx.h).

The product path fails LOUDLY when the library is missing: there is no CPU or PyTorch fallback
for any hot-path op."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must be imported first: libsnx binds to the HIP runtime torch loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
# SNX_LIB: another build of the same sources (A/B runs of kernel variants on one box); default = the in-tree library
LIB_PATH = os.environ.get("SNX_LIB") or os.path.join(_HERE, "libsnx.so")

P = C.c_void_p
I32 = C.c_int32
I64 = C.c_int64
F32 = C.c_float
SZ = C.c_size_t

# name -> (restype, argtypes); mirrors include/snx.h one to one
SIGNATURES = {
    "snx_cast_bf16": (I32, [P, P, I64, P]),
    "snx_cast_transpose_bf16": (I32, [P, P, I32, I32, P]),
    "snx_gemm_nt_bf16": (I32, [P, P, P, I32, I32, I32, P]),
    "snx_gemm_nt_resid": (I32, [P, P, P, P, I32, I32, I32, P]),
    "snx_gemm_tn_accum": (I32, [P, P, P, I32, I32, I32, P, SZ, P]),
    "snx_gemm_tn_workspace_bytes": (SZ, [P, I32, I32]),
    "snx_ln_fwd": (I32, [P, P, P, I32, I32, F32, P]),
    "snx_ln_fwd_add": (I32, [P, P, P, P, P, I32, I32, F32, P]),
    "snx_embed_ln_fwd": (I32, [P, P, P, P, P, I32, I32, F32, P]),
    "snx_gelu_ln_fwd": (I32, [P, P, P, I32, I32, F32, P]),
    "snx_ln_bwd": (I32, [P, P, P, P, P, P, I32, I32, F32, I32, P, SZ, P]),
    "snx_ln_bwd_workspace_bytes": (SZ, [I32, I32]),
    "snx_embed_ln_bwd_workspace_bytes": (SZ, [I32, I32, I32]),
    "snx_embed_ln_bwd": (I32, [P, P, P, P, P, P, I32, I32, I32, F32, I32, P, SZ, P]),
    "snx_gelu_ln_bwd": (I32, [P, P, P, P, P, I32, I32, F32, P, SZ, P]),
    "snx_rope_inplace": (I32, [P, P, P, I32, I32, I32, P]),
    "snx_geglu_fwd": (I32, [P, P, I32, I32, P]),
    "snx_geglu_bwd": (I32, [P, P, P, I32, I32, P]),
    "snx_attn_fwd": (I32, [P, P, P, P, P, I32, I32, I32, I32, I32, I32, P]),
    "snx_attn_bwd": (I32, [P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, I32, P]),
    "snx_attn_fwd_ex": (I32, [P, P, P, P, P, P, I32, I32, I32, I32, I32, I32, P]),
    "snx_attn_bwd_ex": (I32, [P, P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, I32, P]),
    "snx_attn_configure": (I32, [I32]),
    "snx_configure": (I32, [C.c_char_p, I32]),
    "snx_config_get": (I32, [C.c_char_p, P]),
    "snx_build_flags": (C.c_char_p, []),
    "snx_gemm_nt_rope": (I32, [P, P, P, P, P, I32, I32, I32, I32, P]),
    "snx_rope_rows": (I32, [P, P, P, I32, P]),
    "snx_gemm_nt_rope_rows": (I32, [P, P, P, P, P, P, I32, I32, I32, I32, P]),
    "snx_gemm_nt_geglu_fwd": (I32, [P, P, P, P, I32, I32, I32, P]),
    "snx_gemm_nt_geglu_bwd": (I32, [P, P, P, P, I32, I32, I32, P]),
    "snx_gemm_tn_accum_interleaved": (I32, [P, P, P, I32, I32, I32, P, SZ, P]),
    "snx_gemm_tn_accum_group": (I32, [P, I32, I32, P, SZ, P]),
    "snx_nt256_configure": (I32, [I32, I32]),
    "snx_set_reserved_cus": (I32, [I32]),
    "snx_get_reserved_cus": (I32, []),
    "snx_cast_geglu_interleave": (I32, [P, P, P, I32, I32, P]),
    "snx_splade_head_scratch_bytes": (SZ, [I32, I32]),
    "snx_decoder_splade_fwd": (I32, [P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, P]),
    "snx_decoder_splade_fwd_ex": (I32, [P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, I32, P]),
    "snx_splade_bwd": (I32, [P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, P]),
    "snx_loss_workspace_bytes": (SZ, [I32, I32, I32, I32]),
    "snx_loss_fwd": (I32, [P, P, P, P, P, P, P, P, P, P, P]),
    "snx_loss_bwd": (I32, [P, P, P, P, P, P, P, I32, P, P, P, P]),
    "snx_model_workspace_bytes": (SZ, [P, I32, I32, I32]),
    "snx_model_bwd_workspace_bytes": (SZ, [P, I32, I32, I32]),
    "snx_splade_bwd_scratch_bytes": (SZ, [I32, I32, I32]),
    "snx_model_keys_offset": (SZ, [P, I32, I32]),
    "snx_weight_cache_bytes": (SZ, [P]),
    "snx_weight_cache_refresh": (I32, [P, P, P, P]),
    "snx_model_forward": (I32, [P, P, P, P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, P]),
    "snx_model_forward_range": (I32, [P, P, P, P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, P]),
    "snx_model_backward_units_range": (I32, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, I32, I32, P, P]),
    "snx_model_backward": (I32, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, I32, I32, I32, P]),
    "snx_model_backward_units": (I32, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, P, P]),
    "snx_sparse_topk": (I32, [P, P, P, P, P, P, I32, I32, I32, I32, P]),
    "snx_model_workspace_bytes_f32": (SZ, [P, I32, I32, I32]),
    "snx_model_bwd_workspace_bytes_f32": (SZ, [P, I32]),
    "snx_model_forward_f32": (I32, [P, P, P, P, P, P, P, P, P, P, P, I32, I32, I32, P]),
    "snx_model_backward_f32": (I32, [P, P, P, P, P, P, P, P, P, P, P, P, I32, I32, P]),
    "snx_gemm_f32": (I32, [P, I64, I64, P, I64, I64, P, I64, P, I64, I32, I32, I32, I32, P]),
    "snx_param_count": (I32, [P]),
    "snx_adamw_scratch_bytes": (SZ, []),
    "snx_adamw_clip_step": (I32, [P, P, P, P, I64, P, I64, I64, I64, P, P, P]),
    "snx_version": (I32, []),
    "snx_prof_enable": (I32, [I32]),
    "snx_prof_num_classes": (I32, []),
    "snx_prof_class_name": (C.c_char_p, [I32]),
    "snx_prof_read": (I32, [P, P, P]),
}

_lib = None


class SnxLibraryError(RuntimeError):
    pass


def lib():
    """dlopen libsnx.so (once)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SnxLibraryError(
                f"{LIB_PATH} not found: the HIP extension has not been built. Run "
                "`python opensearch-neural-pre-train_amd/snx/build.py` (or __graft_entry__.build()). "
                "There is no CPU fallback for the product path.")
        _lib = C.CDLL(LIB_PATH)
        _check_build_flags(_lib)
        _configure(_lib)
    return _lib


def _check_build_flags(L):
    """A library compiled with a timing-only diagnostics macro returns WRONG results by design (snx/build.py
    WRONG_RESULT_MACROS): refuse it as the product library unless the caller says it knows (SNX_ALLOW_DIAG_LIB=1, the
    microbenchmark tools); trace builds compute the same results and only warn."""
    try:
        L.snx_build_flags.restype = C.c_char_p
        flags = (L.snx_build_flags() or b"").decode()
    except AttributeError:
        return
    if not flags:
        return
    from snx.build import WRONG_RESULT_MACROS
    bad = [m for m in WRONG_RESULT_MACROS if m in flags]
    if bad and os.environ.get("SNX_ALLOW_DIAG_LIB", "0") != "1":
        raise SnxLibraryError(f"{LIB_PATH} was built with {flags!r}: {', '.join(bad)} make its results wrong (timing-only "
                              "diagnostics). Rebuild without SNX_EXTRA_HIPCC_FLAGS, or set SNX_ALLOW_DIAG_LIB=1.")
    import warnings
    warnings.warn(f"libsnx.so is a diagnostics build ({flags})")


# environment variable -> key of snx_configure (include/snx.h).  The library itself reads no environment: this table is
# the whole switchboard, applied once when the library is loaded.  Diagnostics builds (-DSNX_DIAG) accept more keys
# through snx.configure(); they have no environment form.
ENV_KEYS = {
    "SNX_NT256": "nt256", "SNX_NT256_MIN_M": "nt256_min_m", "SNX_NT256_COLDEAL": "nt256_coldeal", "SNX_NT256_REV": "nt256_rev", "SNX_TN256": "tn256", "SNX_TN256_MIN_M": "tn256_min_m",
    "SNX_DEC256": "dec256", "SNX_DEC256_MIN_T": "dec256_min_t", "SNX_BWD_OVERLAP": "bwd_overlap",
    "SNX_ATTN_BWD_ONEPASS": "attn_bwd_onepass", "SNX_ATTN_INTERLEAVE": "attn_interleave", "SNX_STREAM_NT": "stream_nt", "SNX_SPLADE_DW_LAST": "splade_dw_last", "SNX_SPLADE_DH_PANELS": "splade_dh_panels", "SNX_RESID_IN_LN": "resid_in_ln", "SNX_DET_REDUCE": "det_reduce", "SNX_NT_PIPE": "nt_pipe",
}


def _configure(L):
    L.snx_configure.restype = C.c_int
    L.snx_configure.argtypes = [C.c_char_p, C.c_int32]
    for env, key in ENV_KEYS.items():
        v = os.environ.get(env)
        if v is not None and L.snx_configure(key.encode(), int(v)) != 0:
            raise SnxLibraryError(f"{env}={v!r}: not a valid value for {key!r} (include/snx.h snx_configure)")


def configure(**kw) -> None:
    """snx.configure(key=value, ...): process-wide switches of the library (keys: include/snx.h snx_configure)."""
    L = lib()
    for k, v in kw.items():
        if L.snx_configure(k.encode(), int(v)) != 0:
            raise SnxError(f"snx_configure({k!r}, {v!r}) refused: unknown key or value out of range")


def config(key: str) -> int:
    out = C.c_int32(0)
    L = lib()
    L.snx_config_get.restype = C.c_int
    L.snx_config_get.argtypes = [C.c_char_p, C.POINTER(C.c_int32)]
    if L.snx_config_get(key.encode(), C.byref(out)) != 0:
        raise SnxError(f"snx_config_get({key!r}): unknown key")
    return out.value


_bound = {}


def fn(name: str):
    """Typed entry point ``name`` of the C ABI."""
    f = _bound.get(name)
    if f is None:
        res, args = SIGNATURES[name]
        try:
            f = getattr(lib(), name)
        except AttributeError as e:
            raise SnxLibraryError(f"libsnx.so does not export {name}; rebuild it") from e
        f.restype = res
        f.argtypes = args
        _bound[name] = f
    return f


def verify_exports():
    """Every symbol declared in include/snx.h must be exported (used by the CPU test-suite)."""
    return [n for n in SIGNATURES if not hasattr(lib(), n)]


def exported_symbols():
    return list(SIGNATURES.keys())


class SnxError(RuntimeError):
    pass


_CODES = {-2: "SNX_E_SHAPE (operand shapes violate kernel tiling assumptions)",
          -3: "SNX_E_ARG (null or inconsistent argument)"}


def check(code: int, what: str):
    if code != 0:
        raise SnxError(f"{what} failed: {_CODES.get(code, f'hipError {code}')}")
