"""Tensor-level wrappers over the C ABI (one function per exported op).

Each wrapper validates device / dtype / contiguity / shapes on the host BEFORE the launch (a
mis-shaped operand must raise here, never fault on the GPU), allocates outputs with torch and
launches on torch's current stream."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from ._lib import check, fn

BF16 = torch.bfloat16


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _chk(t: torch.Tensor, dtype, name: str, shape=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f"{name}: expected a CUDA(HIP) tensor")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


# ----------------------------------------------------------------------------- casts
def cast_bf16(x: torch.Tensor) -> torch.Tensor:
    _chk(x, torch.float32, "x")
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    check(fn("snx_cast_bf16")(_p(x), _p(out), x.numel(), _stream()), "snx_cast_bf16")
    return out


def cast_transpose_bf16(w: torch.Tensor) -> torch.Tensor:
    _chk(w, torch.float32, "w")
    R, Cc = w.shape
    out = torch.empty((Cc, R), dtype=BF16, device=w.device)
    check(fn("snx_cast_transpose_bf16")(_p(w), _p(out), R, Cc, _stream()), "snx_cast_transpose_bf16")
    return out


# ----------------------------------------------------------------------------- GEMMs
def gemm_nt(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """C[M,N] = A[M,K] @ B[N,K]^T, bf16 in / fp32 accumulate / bf16 out."""
    _chk(a, BF16, "a"); _chk(b, BF16, "b")
    M, K = a.shape
    N, K2 = b.shape
    if K != K2 or K % 64:
        raise ValueError(f"gemm_nt: K mismatch or K % 64 != 0 ({K}, {K2})")
    c = torch.empty((M, N), dtype=BF16, device=a.device)
    check(fn("snx_gemm_nt_bf16")(_p(a), _p(b), _p(c), M, N, K, _stream()), "snx_gemm_nt_bf16")
    return c


def gemm_nt_resid(a: torch.Tensor, b: torch.Tensor, h_in: torch.Tensor) -> torch.Tensor:
    """h_out = h_in + bf16(A @ B^T)  (fp32 residual stream)."""
    _chk(a, BF16, "a"); _chk(b, BF16, "b")
    M, K = a.shape
    N, K2 = b.shape
    if K != K2 or K % 64:
        raise ValueError("gemm_nt_resid: bad K")
    _chk(h_in, torch.float32, "h_in", (M, N))
    h_out = torch.empty_like(h_in)
    check(fn("snx_gemm_nt_resid")(_p(a), _p(b), _p(h_in), _p(h_out), M, N, K, _stream()), "snx_gemm_nt_resid")
    return h_out


class TnProblem(C.Structure):
    _fields_ = [("dY", C.c_void_p), ("X", C.c_void_p), ("dW", C.c_void_p), ("N", C.c_int32), ("K", C.c_int32),
                ("interleaved", C.c_int32), ("reserved", C.c_int32)]


def _scratch(nbytes: int, device) -> Tuple[Optional[torch.Tensor], int]:
    """Caller-owned workspace of an ordered reduction (include/snx.h "det_reduce"): a fresh torch buffer per call -- the
    caching allocator hands it back stream-ordered, which is the one-stream-at-a-time rule of the header."""
    if nbytes <= 0:
        return None, 0
    return torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes


def _tn_ws(arr, n: int, M: int, device):
    return _scratch(int(fn("snx_gemm_tn_workspace_bytes")(arr, n, M)), device)


def gemm_tn_accum(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor) -> None:
    """dw[N,K] += dy[M,N]^T @ x[M,K]  (fp32 accumulate into the gradient buffer)."""
    _chk(dy, BF16, "dy"); _chk(x, BF16, "x")
    M, N = dy.shape
    M2, K = x.shape
    if M != M2:
        raise ValueError("gemm_tn_accum: row mismatch")
    _chk(dw, torch.float32, "dw", (N, K))
    one = (TnProblem * 1)(TnProblem(0, 0, 0, N, K, 0, 0))
    ws, nb = _tn_ws(one, 1, M, dy.device)
    check(fn("snx_gemm_tn_accum")(_p(dy), _p(x), _p(dw), M, N, K, _p(ws), nb, _stream()), "snx_gemm_tn_accum")


def gemm_nt_rope(a: torch.Tensor, b: torch.Tensor, table: torch.Tensor, pos: torch.Tensor, rope_cols: int,
                 validate: bool = True):
    """qkv = rope(A @ B^T) on columns < rope_cols (fused Wqkv + apply_rotary_pos_emb).
    ``validate`` checks pos against the table length (a device sync: off in timing loops)."""
    _chk(a, BF16, "a"); _chk(b, BF16, "b")
    M, K = a.shape; N = b.shape[0]
    _chk(table, torch.float32, "table"); _chk(pos, torch.int32, "pos", (M,))
    if b.shape[1] != K or K % 64 or N % 64 or (validate and int(pos.max()) >= table.shape[0]):
        raise ValueError("gemm_nt_rope: bad shapes")
    c = torch.empty((M, N), dtype=BF16, device=a.device)
    check(fn("snx_gemm_nt_rope")(_p(a), _p(b), _p(c), _p(table), _p(pos), rope_cols, M, N, K, _stream()),
          "snx_gemm_nt_rope")
    return c


def rope_rows(table: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
    """[T, 32, 2] fp32 = table[pos]: every token's (cos, sin) row, resolved once per pass and theta."""
    _chk(table, torch.float32, "table"); _chk(pos, torch.int32, "pos")
    T = pos.shape[0]
    rows = torch.empty((T, 32, 2), dtype=torch.float32, device=pos.device)
    check(fn("snx_rope_rows")(_p(table), _p(pos), _p(rows), T, _stream()), "snx_rope_rows")
    return rows


def gemm_nt_rope_rows(a: torch.Tensor, b: torch.Tensor, table: torch.Tensor, pos: torch.Tensor, rows: torch.Tensor,
                      rope_cols: int):
    """gemm_nt_rope with the per-token (cos, sin) rows of ``rope_rows`` handed in (same results)."""
    _chk(a, BF16, "a"); _chk(b, BF16, "b")
    M, K = a.shape; N = b.shape[0]
    _chk(table, torch.float32, "table"); _chk(pos, torch.int32, "pos", (M,)); _chk(rows, torch.float32, "rows", (M, 32, 2))
    if b.shape[1] != K or K % 64 or N % 64:
        raise ValueError("gemm_nt_rope_rows: bad shapes")
    c = torch.empty((M, N), dtype=BF16, device=a.device)
    check(fn("snx_gemm_nt_rope_rows")(_p(a), _p(b), _p(c), _p(table), _p(pos), _p(rows), rope_cols, M, N, K, _stream()),
          "snx_gemm_nt_rope_rows")
    return c


def cast_geglu_interleave(wi: torch.Tensor):
    """fp32 Wi [2I, C] -> (bf16 interleaved [2I, C], bf16 interleaved transposed [C, 2I])."""
    _chk(wi, torch.float32, "wi")
    R, Cc = wi.shape
    out = torch.empty((R, Cc), dtype=BF16, device=wi.device)
    out_t = torch.empty((Cc, R), dtype=BF16, device=wi.device)
    check(fn("snx_cast_geglu_interleave")(_p(wi), _p(out), _p(out_t), R // 2, Cc, _stream()), "snx_cast_geglu_interleave")
    return out, out_t


def gemm_nt_geglu_fwd(a: torch.Tensor, wi_interleaved: torch.Tensor):
    """-> (u [M, 2I] interleaved, y [M, I])."""
    _chk(a, BF16, "a"); _chk(wi_interleaved, BF16, "wi")
    M, K = a.shape; N = wi_interleaved.shape[0]
    if wi_interleaved.shape[1] != K or K % 64 or N % 64:
        raise ValueError("gemm_nt_geglu_fwd: bad shapes")
    u = torch.empty((M, N), dtype=BF16, device=a.device)
    y = torch.empty((M, N // 2), dtype=BF16, device=a.device)
    check(fn("snx_gemm_nt_geglu_fwd")(_p(a), _p(wi_interleaved), _p(u), _p(y), M, N, K, _stream()), "snx_gemm_nt_geglu_fwd")
    return u, y


def gemm_nt_geglu_bwd(a: torch.Tensor, b: torch.Tensor, u: torch.Tensor):
    """dy = A @ B^T [M, I];  -> du [M, 2I] (interleaved) = GeGLU backward(u, dy)."""
    _chk(a, BF16, "a"); _chk(b, BF16, "b")
    M, K = a.shape; N = b.shape[0]
    _chk(u, BF16, "u", (M, 2 * N))
    if b.shape[1] != K or K % 64 or N % 32:
        raise ValueError("gemm_nt_geglu_bwd: bad shapes")
    du = torch.empty_like(u)
    check(fn("snx_gemm_nt_geglu_bwd")(_p(a), _p(b), _p(u), _p(du), M, N, K, _stream()), "snx_gemm_nt_geglu_bwd")
    return du


def gemm_tn_accum_interleaved(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor) -> None:
    _chk(dy, BF16, "dy"); _chk(x, BF16, "x")
    M, N = dy.shape; K = x.shape[1]
    _chk(dw, torch.float32, "dw", (N, K))
    one = (TnProblem * 1)(TnProblem(0, 0, 0, N, K, 1, 0))
    ws, nb = _tn_ws(one, 1, M, dy.device)
    check(fn("snx_gemm_tn_accum_interleaved")(_p(dy), _p(x), _p(dw), M, N, K, _p(ws), nb, _stream()),
          "snx_gemm_tn_accum_interleaved")


def gemm_tn_accum_group(problems) -> None:
    """[(dy [M,N] bf16, x [M,K] bf16, dw [N,K] fp32, interleaved: bool)] (1..4 entries, same M):
    dw += dy^T @ x for all of them in ONE launch (include/snx.h snx_gemm_tn_accum_group)."""
    if not 1 <= len(problems) <= 4:
        raise ValueError("gemm_tn_accum_group: 1..4 problems")
    M = problems[0][0].shape[0]
    arr = (TnProblem * len(problems))()
    for i, (dy, x, dw, inter) in enumerate(problems):
        _chk(dy, BF16, "dy"); _chk(x, BF16, "x")
        if dy.shape[0] != M or x.shape[0] != M:
            raise ValueError("gemm_tn_accum_group: all problems share the token dimension")
        N, K = dy.shape[1], x.shape[1]
        _chk(dw, torch.float32, "dw", (N, K))
        arr[i] = TnProblem(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), N, K, int(bool(inter)), 0)
    ws, nb = _tn_ws(arr, len(problems), M, problems[0][0].device)
    check(fn("snx_gemm_tn_accum_group")(arr, len(problems), M, _p(ws), nb, _stream()), "snx_gemm_tn_accum_group")


# ----------------------------------------------------------------------------- norms
def ln_fwd(h: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    _chk(h, torch.float32, "h"); T, H = h.shape
    _chk(w, torch.float32, "w", (H,))
    x = torch.empty((T, H), dtype=BF16, device=h.device)
    check(fn("snx_ln_fwd")(_p(h), _p(w), _p(x), T, H, eps, _stream()), "snx_ln_fwd")
    return x


def embed_ln_fwd(ids: torch.Tensor, E: torch.Tensor, w: torch.Tensor, eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    _chk(ids, torch.int64, "ids"); _chk(E, torch.float32, "E")
    V, H = E.shape
    _chk(w, torch.float32, "w", (H,))
    if int(ids.min()) < 0 or int(ids.max()) >= V:
        raise ValueError("embed_ln_fwd: token id out of range")
    T = ids.numel()
    h = torch.empty((T, H), dtype=torch.float32, device=E.device)
    x0 = torch.empty((T, H), dtype=BF16, device=E.device)
    check(fn("snx_embed_ln_fwd")(_p(ids), _p(E), _p(w), _p(h), _p(x0), T, H, eps, _stream()), "snx_embed_ln_fwd")
    return h, x0


def gelu_ln_fwd(d: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    _chk(d, BF16, "d"); T, H = d.shape
    _chk(w, torch.float32, "w", (H,))
    x = torch.empty((T, H), dtype=BF16, device=d.device)
    check(fn("snx_gelu_ln_fwd")(_p(d), _p(w), _p(x), T, H, eps, _stream()), "snx_gelu_ln_fwd")
    return x


def ln_bwd(dy: torch.Tensor, h: torch.Tensor, w: torch.Tensor, dh: torch.Tensor, dw: torch.Tensor, eps: float,
           overwrite: bool = False, dh_bf16: Optional[torch.Tensor] = None) -> None:
    """dh (+)= LN_bwd(dy; h, w);  dw += sum_t dy * xhat;  dh_bf16 (optional) = bf16(dh)."""
    _chk(h, torch.float32, "h"); T, H = h.shape
    _chk(dy, BF16, "dy", (T, H)); _chk(w, torch.float32, "w", (H,))
    _chk(dh, torch.float32, "dh", (T, H)); _chk(dw, torch.float32, "dw", (H,))
    if dh_bf16 is not None:
        _chk(dh_bf16, BF16, "dh_bf16", (T, H))
    ws, nb = _scratch(int(fn("snx_ln_bwd_workspace_bytes")(T, H)), h.device)
    check(fn("snx_ln_bwd")(_p(dy), _p(h), _p(w), _p(dh), _p(dh_bf16), _p(dw), T, H, eps, int(overwrite), _p(ws), nb,
                           _stream()), "snx_ln_bwd")


def embed_ln_bwd(dh: torch.Tensor, ids: torch.Tensor, E: torch.Tensor, w: torch.Tensor, gradE: torch.Tensor,
                 dw: torch.Tensor, eps: float, pad_id: int) -> None:
    _chk(dh, torch.float32, "dh"); T, H = dh.shape
    _chk(ids, torch.int64, "ids"); _chk(E, torch.float32, "E"); _chk(gradE, torch.float32, "gradE", E.shape)
    _chk(w, torch.float32, "w", (H,)); _chk(dw, torch.float32, "dw", (H,))
    if ids.numel() != T:
        raise ValueError("embed_ln_bwd: ids/dh row mismatch")
    V = E.shape[0]
    ws, nb = _scratch(int(fn("snx_embed_ln_bwd_workspace_bytes")(T, H, V)), dh.device)
    check(fn("snx_embed_ln_bwd")(_p(dh), _p(ids), _p(E), _p(w), _p(gradE), _p(dw), T, H, V, eps, pad_id, _p(ws), nb,
                                 _stream()), "snx_embed_ln_bwd")


def gelu_ln_bwd(dy: torch.Tensor, d: torch.Tensor, w: torch.Tensor, dw: torch.Tensor, eps: float) -> torch.Tensor:
    _chk(d, BF16, "d"); T, H = d.shape
    _chk(dy, BF16, "dy", (T, H)); _chk(w, torch.float32, "w", (H,)); _chk(dw, torch.float32, "dw", (H,))
    dd = torch.empty_like(d)
    ws, nb = _scratch(int(fn("snx_ln_bwd_workspace_bytes")(T, H)), d.device)
    check(fn("snx_gelu_ln_bwd")(_p(dy), _p(d), _p(w), _p(dd), _p(dw), T, H, eps, _p(ws), nb, _stream()),
          "snx_gelu_ln_bwd")
    return dd


# ----------------------------------------------------------------------------- rope / geglu
def rope_table(max_pos: int, head_dim: int, theta: float, device) -> torch.Tensor:
    """[max_pos, head_dim/2, 2] fp32 (cos, sin), computed exactly as hf:136-163 does (fp32)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    freqs = torch.arange(max_pos, dtype=torch.float)[:, None] * inv_freq[None, :]
    return torch.stack((freqs.cos(), freqs.sin()), dim=-1).contiguous().to(device)


def rope_inplace(qkv: torch.Tensor, table: torch.Tensor, pos: torch.Tensor, heads: int, inverse: bool = False) -> None:
    _chk(qkv, BF16, "qkv"); T = qkv.shape[0]
    if qkv.numel() != T * 3 * heads * 64:
        raise ValueError("rope_inplace: qkv must be [T, 3*heads*64]")
    _chk(table, torch.float32, "table"); _chk(pos, torch.int32, "pos", (T,))
    if table.shape[1:] != (32, 2) or int(pos.max()) >= table.shape[0] or int(pos.min()) < 0:
        raise ValueError("rope_inplace: position outside the cos/sin table")
    check(fn("snx_rope_inplace")(_p(qkv), _p(table), _p(pos), T, heads, int(inverse), _stream()), "snx_rope_inplace")


def geglu_fwd(u: torch.Tensor) -> torch.Tensor:
    _chk(u, BF16, "u"); T, I2 = u.shape
    y = torch.empty((T, I2 // 2), dtype=BF16, device=u.device)
    check(fn("snx_geglu_fwd")(_p(u), _p(y), T, I2 // 2, _stream()), "snx_geglu_fwd")
    return y


def geglu_bwd(u: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    _chk(u, BF16, "u"); T, I2 = u.shape
    _chk(dy, BF16, "dy", (T, I2 // 2))
    du = torch.empty_like(u)
    check(fn("snx_geglu_bwd")(_p(u), _p(dy), _p(du), T, I2 // 2, _stream()), "snx_geglu_bwd")
    return du


# ----------------------------------------------------------------------------- attention
def _check_seqs(cu: torch.Tensor, T: int, max_seqlen: int, groups=None):
    _chk(cu, torch.int32, "cu_seqlens")
    c = cu.tolist()
    if c[0] != 0 or c[-1] != T or any(b <= a for a, b in zip(c, c[1:])) or max(b - a for a, b in zip(c, c[1:])) > max_seqlen:
        raise ValueError("cu_seqlens must start at 0, end at T, be increasing, with lengths <= max_seqlen")
    if groups is not None:                       # a group's max_len must cover its sequences (kernels size LDS by it)
        for s0, n, ml in groups:
            if s0 < 0 or n <= 0 or s0 + n > len(c) - 1 or max(c[i + 1] - c[i] for i in range(s0, s0 + n)) > ml:
                raise ValueError("sequence group does not cover its sequences")


def attn_fwd(qkv: torch.Tensor, cu: torch.Tensor, mask: torch.Tensor, max_seqlen: int, heads: int,
             window: int, validate: bool = True, groups=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """window < 0 -> global layer; else inclusive half-window.  ``groups``: optional list of
    (seq_begin, nseq, max_len) for consecutive sequence groups (launch sizing only)."""
    _chk(qkv, BF16, "qkv"); T = qkv.shape[0]
    if qkv.numel() != T * 3 * heads * 64:
        raise ValueError("attn_fwd: qkv must be [T, 3*heads*64]")
    _chk(mask, torch.int64, "mask")
    if mask.numel() != T:
        raise ValueError("attn_fwd: mask must have T elements")
    if validate:
        _check_seqs(cu, T, max_seqlen, groups)
    nseq = cu.numel() - 1
    out = torch.empty((T, heads * 64), dtype=BF16, device=qkv.device)
    lse = torch.empty((heads, T), dtype=torch.float32, device=qkv.device)
    check(fn("snx_attn_fwd_ex")(_p(qkv), _p(cu), _p(mask), _p(out), _p(lse), _groups(groups), T, nseq, max_seqlen,
                                heads, 64, window, _stream()), "snx_attn_fwd_ex")
    return out, lse


def _groups(groups):
    if groups is None:
        return None
    flat = [len(groups)] + [int(v) for g in groups for v in g]
    return (C.c_int32 * len(flat))(*flat)


def attn_bwd(qkv, out, dout, lse, cu, mask, max_seqlen: int, heads: int, window: int, validate: bool = True,
             rope_table: Optional[torch.Tensor] = None, pos: Optional[torch.Tensor] = None, groups=None):
    """Returns dqkv [T, 3*heads*64] bf16: gradients w.r.t. the post-RoPE q, k and v, or -- with
    rope_table/pos -- w.r.t. the pre-RoPE projections (inverse rotation fused into the epilogue)."""
    _chk(qkv, BF16, "qkv"); T = qkv.shape[0]
    _chk(out, BF16, "out", (T, heads * 64)); _chk(dout, BF16, "dout", (T, heads * 64))
    _chk(lse, torch.float32, "lse", (heads, T)); _chk(mask, torch.int64, "mask")
    if validate:
        _check_seqs(cu, T, max_seqlen, groups)
    nseq = cu.numel() - 1
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((heads, T), dtype=torch.float32, device=qkv.device)
    if rope_table is not None:
        _chk(rope_table, torch.float32, "rope_table"); _chk(pos, torch.int32, "pos", (T,))
    check(fn("snx_attn_bwd_ex")(_p(qkv), _p(out), _p(dout), _p(lse), _p(cu), _p(mask), _p(delta), _p(dqkv),
                                _p(rope_table), _p(pos), _groups(groups), T, nseq, max_seqlen, heads, 64, window,
                                _stream()), "snx_attn_bwd_ex")
    return dqkv


# ----------------------------------------------------------------------------- SPLADE head
def decoder_splade_fwd(hd: torch.Tensor, w_bf16: torch.Tensor, bias: torch.Tensor, cu: torch.Tensor,
                       mask: torch.Tensor, max_seqlen: int, validate: bool = True):
    """-> sparse_repr [nseq, V] fp32, keys [nseq, V] int32 (packed value|argmax), token_weights [T] fp32."""
    _chk(hd, BF16, "hd"); T, K = hd.shape
    _chk(w_bf16, BF16, "w"); V = w_bf16.shape[0]
    if w_bf16.shape[1] != K:
        raise ValueError("decoder_splade_fwd: K mismatch")
    _chk(bias, torch.float32, "bias", (V,)); _chk(mask, torch.int64, "mask")
    if mask.numel() != T:
        raise ValueError("mask must have T elements")
    if validate:
        _check_seqs(cu, T, max_seqlen)
    nseq = cu.numel() - 1
    sparse = torch.empty((nseq, V), dtype=torch.float32, device=hd.device)
    keys = torch.empty((nseq, V), dtype=torch.int32, device=hd.device)
    tw = torch.empty((T,), dtype=torch.float32, device=hd.device)
    nbytes = fn("snx_splade_head_scratch_bytes")(T, V)
    scratch = torch.empty((nbytes,), dtype=torch.uint8, device=hd.device)
    check(fn("snx_decoder_splade_fwd")(_p(hd), _p(w_bf16), _p(bias), _p(cu), _p(mask), _p(sparse), _p(keys), _p(tw),
                                       _p(scratch), T, nseq, max_seqlen, V, K, _stream()), "snx_decoder_splade_fwd")
    return sparse, keys, tw


# ----------------------------------------------------------------------------- inference post-processing
def sparse_topk(rep: torch.Tensor, allowed: torch.Tensor, k: Optional[int] = None):
    """Per row of rep [B,V] fp32: survivors (rep > 0 and allowed[v]); more than k of them -> the k largest, weight
    descending, ties lowest id first; else all survivors in id order (ref:benchmark/encoders.py:320-343).
    -> (values [B,cap] fp32, ids [B,cap] int32, counts [B] int32, sorted_flags [B] int32), cap = min(k,V) or V."""
    _chk(rep, torch.float32, "rep")
    if rep.dim() != 2:
        raise ValueError("sparse_topk: rep must be [B, V]")
    B, V = rep.shape
    _chk(allowed, torch.uint8, "allowed", (V,))
    kk = 0 if k is None else int(k)
    if k is not None and (kk < 1 or kk > 16384):
        raise ValueError("sparse_topk: k must be in [1, 16384] (or None)")
    cap = min(kk, V) if kk else V
    vals = torch.empty((B, cap), dtype=torch.float32, device=rep.device)
    ids = torch.empty((B, cap), dtype=torch.int32, device=rep.device)
    cnt = torch.empty((B,), dtype=torch.int32, device=rep.device)
    srt = torch.empty((B,), dtype=torch.int32, device=rep.device)
    check(fn("snx_sparse_topk")(_p(rep), _p(allowed), _p(vals), _p(ids), _p(cnt), _p(srt), B, V, min(kk, V), cap,
                                _stream()), "snx_sparse_topk")
    return vals, ids, cnt, srt
