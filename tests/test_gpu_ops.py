"""Op-level parity of the HIP kernels (through the C ABI) against plain PyTorch / oracle math with
the same bf16 cast points.  Needs a real MI355X:  pytest -m gpu."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BF16 = torch.bfloat16


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _ops():
    from snx import ops
    return ops


def _ulp_close(got, ref, what, max_ulp=1.01, frac_exact=0.97):
    """bf16 tensors produced from the same exact products: equal except for rare 1-ulp flips
    caused by fp32 accumulation order."""
    g, r = got.float(), ref.float()
    ulp = torch.clamp(r.abs(), min=1e-30) * 2.0 ** -8 * 2
    bad = ((g - r).abs() > max_ulp * ulp + 1e-6)
    assert bad.sum().item() == 0, f"{what}: {bad.sum().item()} elements differ by >1 bf16 ulp, max {((g-r).abs()).max().item()}"
    assert (g == r).float().mean().item() >= frac_exact, f"{what}: too few exact matches {(g == r).float().mean().item()}"


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 768), (200, 1000, 768), (4096, 2304, 768),
                                   (130, 768, 1152), (1, 128, 64)])
def test_gemm_nt(dev, M, N, K):
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N)
    a = torch.randn(M, K, generator=g).to(dev).to(BF16)
    b = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(BF16)
    c = ops.gemm_nt(a, b)
    ref = (a.float() @ b.float().t()).to(BF16)
    _ulp_close(c, ref, f"gemm_nt {M}x{N}x{K}")


def test_gemm_nt_layout_identity(dev):
    """A = I (rectangular) with an ASYMMETRIC B catches a transposed C write."""
    ops = _ops()
    M, N, K = 128, 256, 128
    a = torch.zeros(M, K, device=dev)
    a[torch.arange(M), torch.arange(M)] = 1.0
    b = (torch.arange(N * K, device=dev, dtype=torch.float32).reshape(N, K) % 251) - 125.0
    c = ops.gemm_nt(a.to(BF16), b.to(BF16))
    assert torch.equal(c.float(), b.to(BF16).float()[:, :M].t().contiguous())


def test_gemm_nt_resid(dev):
    ops = _ops()
    M, N, K = 300, 768, 1152
    g = torch.Generator().manual_seed(3)
    a = torch.randn(M, K, generator=g).to(dev).to(BF16)
    b = (torch.randn(N, K, generator=g) * 0.03).to(dev).to(BF16)
    h = torch.randn(M, N, generator=g).to(dev)
    out = ops.gemm_nt_resid(a, b, h)
    ref = h + (a.float() @ b.float().t()).to(BF16).float()
    assert (out - ref).abs().max().item() < 2e-2
    assert ((out - ref).abs() > 1e-6).float().mean().item() < 0.03


@pytest.mark.parametrize("T,H", [(7, 256), (1000, 768), (4097, 768)])
def test_layernorm_fwd(dev, T, H):
    ops = _ops()
    g = torch.Generator().manual_seed(T)
    h = (torch.randn(T, H, generator=g) * 3 + 0.5).to(dev)
    w = (1 + 0.2 * torch.randn(H, generator=g)).to(dev)
    x = ops.ln_fwd(h, w, 1e-5)
    ref = torch.nn.functional.layer_norm(h, (H,), w, None, 1e-5)
    assert (x.float() - ref).abs().max().item() < 2 ** -7 * ref.abs().max().item()
    assert (x == ref.to(BF16)).float().mean().item() > 0.99


def test_embed_ln_fwd(dev):
    ops = _ops()
    V, H, T = 1000, 768, 300
    g = torch.Generator().manual_seed(1)
    E = torch.randn(V, H, generator=g).to(dev) * 0.02
    w = (1 + 0.2 * torch.randn(H, generator=g)).to(dev)
    ids = torch.randint(0, V, (T,), generator=g).to(dev)
    h, x0 = ops.embed_ln_fwd(ids, E, w, 1e-5)
    ref = torch.nn.functional.layer_norm(E[ids], (H,), w, None, 1e-5)
    assert (h - ref).abs().max().item() < 1e-5
    assert (x0 == h.to(BF16)).all()


def test_gelu_ln_fwd_bwd(dev):
    ops = _ops()
    T, H = 515, 768
    g = torch.Generator().manual_seed(2)
    d = torch.randn(T, H, generator=g).to(dev).to(BF16)
    w = (1 + 0.2 * torch.randn(H, generator=g)).to(dev)
    dy = (torch.randn(T, H, generator=g) * 0.1).to(dev).to(BF16)
    x = ops.gelu_ln_fwd(d, w, 1e-5)
    dl = d.float().requires_grad_(True)
    wl = w.clone().requires_grad_(True)
    ge = torch.nn.functional.gelu(dl).to(BF16)
    ref = torch.nn.functional.layer_norm(ge.float(), (H,), wl, None, 1e-5)
    assert (x == ref.to(BF16)).float().mean().item() > 0.98
    assert (x.float() - ref).abs().max().item() < 0.05
    ref.backward(dy.float())
    dw = torch.zeros(H, device=dev)
    dd = ops.gelu_ln_bwd(dy, d, w, dw, 1e-5)
    assert torch.allclose(dw, wl.grad, rtol=2e-3, atol=2e-3)
    err = (dd.float() - dl.grad).abs().max().item()
    assert err < 2e-2 * dl.grad.abs().max().item() + 1e-3, err


@pytest.mark.parametrize("T,H", [(333, 768), (64, 256)])
def test_layernorm_bwd(dev, T, H):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    h = (torch.randn(T, H, generator=g) * 2).to(dev)
    w = (1 + 0.2 * torch.randn(H, generator=g)).to(dev)
    dy = (torch.randn(T, H, generator=g) * 0.1).to(dev).to(BF16)
    dh0 = torch.randn(T, H, generator=g).to(dev)
    hl, wl = h.clone().requires_grad_(True), w.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(hl, (H,), wl, None, 1e-5).backward(dy.float())
    dh = dh0.clone()
    dw = torch.zeros(H, device=dev)
    ops.ln_bwd(dy, h, w, dh, dw, 1e-5)
    assert torch.allclose(dh, dh0 + hl.grad, rtol=1e-4, atol=1e-5)
    assert torch.allclose(dw, wl.grad, rtol=1e-3, atol=1e-4)
    dh2 = torch.full_like(dh0, float("nan"))
    dw2 = torch.zeros(H, device=dev)
    dhb = torch.empty(T, H, dtype=BF16, device=dev)
    ops.ln_bwd(dy, h, w, dh2, dw2, 1e-5, overwrite=True, dh_bf16=dhb)
    assert torch.allclose(dh2, hl.grad, rtol=1e-4, atol=1e-5)
    assert torch.equal(dhb, dh2.to(BF16))


def test_embed_ln_bwd(dev):
    ops = _ops()
    V, H, T, pad = 50, 256, 400, 49
    g = torch.Generator().manual_seed(6)
    E = torch.randn(V, H, generator=g).to(dev)
    w = (1 + 0.2 * torch.randn(H, generator=g)).to(dev)
    ids = torch.randint(0, V, (T,), generator=g).to(dev)
    dh = torch.randn(T, H, generator=g).to(dev)
    El, wl = E.clone().requires_grad_(True), w.clone().requires_grad_(True)
    emb = torch.nn.functional.embedding(ids, El, padding_idx=pad)
    torch.nn.functional.layer_norm(emb, (H,), wl, None, 1e-5).backward(dh)
    gE = torch.zeros_like(E)
    dw = torch.zeros(H, device=dev)
    ops.embed_ln_bwd(dh, ids, E, w, gE, dw, 1e-5, pad)
    assert torch.allclose(gE, El.grad, rtol=1e-3, atol=1e-4)
    assert gE[pad].abs().max().item() == 0.0
    assert torch.allclose(dw, wl.grad, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("T,H,V", [(9000, 768, 3000), (70000, 256, 500)])
def test_embed_ln_bwd_skewed_ids(dev, T, H, V):
    """real text is Zipfian -- [CLS] / [SEP] once per sequence, frequent tokens several percent of
    the rows.  Ids with more than 64 tokens ("heavy") are ranked through an LDS bitmap and summed in two fixed levels
    (32-token chunks, then the chunks in order) instead of one wave walking the whole list.  Checked against an fp64
    index_add of the exact dx rows, for bit-reproducibility over two runs, and at the 64 / 65 boundary; T = 70,000 crosses
    the 65,536-index bitmap window."""
    ops = _ops()
    pad = V - 1
    g = torch.Generator().manual_seed(T)
    E = torch.randn(V, H, generator=g).to(dev)
    w = (1 + 0.2 * torch.randn(H, generator=g)).to(dev)
    ids = torch.randint(20, V, (T,), generator=g)
    r = torch.rand(T, generator=g)
    ids[r < 0.30] = 5                                   # ~30 % of the rows
    ids[(r >= 0.30) & (r < 0.33)] = 7                   # ~3 %
    ids[(r >= 0.33) & (r < 0.36)] = pad                 # padding takes no part
    free = (r >= 0.40).nonzero().view(-1)
    ids[free[:64]] = 11                                 # exactly 64: the last light size
    ids[free[64:129]] = 12                              # exactly 65: the first heavy size
    assert int((ids == 11).sum()) == 64 and int((ids == 12).sum()) == 65
    ids = ids.to(dev)
    dh = torch.randn(T, H, generator=g).to(dev)
    El, wl = E.clone().requires_grad_(True), w.clone().requires_grad_(True)
    emb = torch.nn.functional.embedding(ids, El, padding_idx=pad)
    torch.nn.functional.layer_norm(emb, (H,), wl, None, 1e-5).backward(dh)
    runs = []
    for _ in range(2):
        gE = torch.zeros_like(E)
        dw = torch.zeros(H, device=dev)
        ops.embed_ln_bwd(dh, ids, E, w, gE, dw, 1e-5, pad)
        runs.append((gE, dw))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    gE = runs[0][0]
    assert gE[pad].abs().max().item() == 0.0
    # torch's embedding backward adds fp32 rows with atomics: compare against an fp64 reference built from ITS per-row dx
    El64 = E.double().clone().requires_grad_(True)
    emb64 = torch.nn.functional.embedding(ids, El64, padding_idx=pad)
    torch.nn.functional.layer_norm(emb64, (H,), w.double(), None, 1e-5).backward(dh.double())
    ref = El64.grad
    scale = ref.abs().amax(dim=1, keepdim=True).clamp_min(1e-6)
    n_id = torch.bincount(ids, minlength=V).clamp_min(1).double()[:, None]
    # fp32 rows (1e-6 relative each) summed over n terms: error grows like sqrt(n) .. n ulps of the largest partial sum
    assert ((gE.double() - ref).abs() / scale <= 3e-6 * n_id.sqrt() + 1e-6 * n_id * 0 + 2e-5).all(), \
        float(((gE.double() - ref).abs() / scale).max())
    assert torch.allclose(runs[0][1].double(), wl.grad.double(), rtol=2e-3, atol=2e-2)


def test_rope(dev):
    from oracle import splade_oracle as O
    ops = _ops()
    B, S, heads = 3, 70, 4
    T = B * S
    g = torch.Generator().manual_seed(8)
    qkv = torch.randn(T, 3 * heads * 64, generator=g).to(BF16)
    pos = torch.arange(S, dtype=torch.int32).repeat(B)
    for theta in (10000.0, 160000.0):
        tab = ops.rope_table(128, 64, theta, dev)
        x = qkv.to(dev).clone()
        ops.rope_inplace(x, tab, pos.to(dev), heads)
        cos, sin = O.rope_tables(S, 64, theta)
        v = qkv.view(B, S, 3, heads, 64)
        ref = v.clone()
        for which in (0, 1):
            t = v[:, :, which].transpose(1, 2)                     # [B, heads, S, 64]
            ref[:, :, which] = O._apply_rope(t, cos, sin).transpose(1, 2)
        got = x.cpu().view(B, S, 3, heads, 64)
        assert torch.equal(got[:, :, 2], v[:, :, 2])               # v untouched
        diff = (got.float() - ref.float()).abs()
        assert diff.max().item() <= 2 ** -7 * 8 and (diff > 0).float().mean().item() < 0.01
        ops.rope_inplace(x, tab, pos.to(dev), heads, inverse=True)  # rotation is orthogonal
        assert (x.cpu().float() - qkv.float()).abs().max().item() < 0.06


def test_geglu(dev):
    ops = _ops()
    T, I = 257, 1152
    g = torch.Generator().manual_seed(9)
    u = torch.randn(T, 2 * I, generator=g).to(dev).to(BF16)
    dy = torch.randn(T, I, generator=g).to(dev).to(BF16)
    y = ops.geglu_fwd(u)
    a, gate = u.float().chunk(2, dim=-1)
    act = torch.nn.functional.gelu(a).to(BF16)
    ref = (act.float() * gate).to(BF16)
    assert torch.equal(y, ref) or (y.float() - ref.float()).abs().max().item() < 2 ** -6
    du = ops.geglu_bwd(u, dy)
    ul = u.float().requires_grad_(True)
    a2, g2 = ul.chunk(2, dim=-1)
    (torch.nn.functional.gelu(a2) * g2).backward(dy.float())
    err = (du.float() - ul.grad).abs().max().item()
    assert err < 0.03 * ul.grad.abs().max().item()


def _ragged(B, S, seed, min_len=1):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(min_len, S + 1, (B,), generator=g)
    lens[0] = S
    if B > 1:
        lens[1] = 1
    mask = (torch.arange(S)[None] < lens[:, None]).long()
    return lens, mask


@pytest.mark.parametrize("S,window", [(64, -1), (64, 64), (256, -1), (256, 64), (200, 8), (130, -1), (512, 64)])
def test_attention_fwd(dev, S, window):
    from oracle import splade_oracle as O
    ops = _ops()
    B, heads = 4, 3
    T = B * S
    lens, mask = _ragged(B, S, S + window)
    g = torch.Generator().manual_seed(11)
    qkv = (torch.randn(T, 3 * heads * 64, generator=g) * 1.5).to(BF16)
    cu = (torch.arange(B + 1, dtype=torch.int32) * S)
    out, lse = ops.attn_fwd(qkv.to(dev), cu.to(dev), mask.reshape(-1).to(dev), S, heads, window)
    v = qkv.view(B, S, 3, heads, 64)
    q, k, vv = (v[:, :, i].transpose(1, 2) for i in range(3))
    vis = O.attention_bias(mask, None if window < 0 else window)
    ref = O._attention(q, k, vv, vis, 0.125, "bf16").transpose(1, 2).reshape(T, heads * 64)
    valid = mask.reshape(-1).bool()
    got = out.cpu().float()[valid]
    exp = ref.float()[valid]
    assert torch.isfinite(out.float()).all()
    err = (got - exp).abs().max().item()
    assert err < 0.03, err
    assert ((got - exp).abs() > 2 ** -7 * exp.abs().clamp(min=0.25)).float().mean().item() < 0.02
    # lse check on valid rows
    s = (q.float() @ k.float().transpose(-1, -2)) * 0.125
    s = s.masked_fill(~vis, float("-inf"))
    lse_ref = torch.logsumexp(s, dim=-1).permute(1, 0, 2).reshape(heads, T)
    assert (lse.cpu()[:, valid] - lse_ref[:, valid]).abs().max().item() < 2e-3


# T = B * S >= 2048 rows run the 256x256 persistent kernel (decoder256.hip): sub-tiles of ragged sequences, a sequence
# over two tiles (S = 300), short sequences sharing a tile (S = 40, 64), vocabulary edges (V = 777, 50000)
@pytest.mark.parametrize("B,S,V,K", [(3, 64, 1000, 256), (4, 256, 1000, 768), (2, 200, 50000, 768), (5, 40, 777, 256),
                                     (2, 300, 640, 256), (12, 256, 1000, 768), (64, 40, 777, 256), (9, 300, 640, 256),
                                     (10, 256, 50000, 768), (40, 64, 1024, 128), (33, 63, 640, 256)])
def test_decoder_splade_fwd(dev, B, S, V, K):
    ops = _ops()
    T = B * S
    lens, mask = _ragged(B, S, V)
    g = torch.Generator().manual_seed(13)
    hd = torch.randn(T, K, generator=g).to(BF16)
    W = (torch.randn(V, K, generator=g) * 0.05).to(BF16)
    bias = torch.randn(V, generator=g) * 0.3
    cu = torch.arange(B + 1, dtype=torch.int32) * S
    sp, keys, tw = ops.decoder_splade_fwd(hd.to(dev), W.to(dev), bias.to(dev), cu.to(dev), mask.reshape(-1).to(dev), S)
    logits = ((hd.to(dev).float() @ W.to(dev).float().t()) + bias.to(dev).to(BF16).float()).to(BF16)   # [T, V]
    sc = torch.log1p(torch.relu(logits).float()).view(B, S, V) * mask.to(dev)[:, :, None].float()
    ref_sp = sc.max(dim=1).values
    ref_tw = sc.max(dim=-1).values.reshape(T)
    # fp32 accumulation-order flips move a bf16 logit by one ulp at most
    d = (sp - ref_sp).abs()
    assert d.max().item() < 0.02, d.max().item()
    assert (d > 1e-6).float().mean().item() < 0.02
    dt = (tw - ref_tw).abs()
    assert dt.max().item() < 0.02 and (dt > 1e-6).float().mean().item() < 0.03
    assert (tw[mask.reshape(-1).to(dev) == 0] == 0).all()
    # packed keys: value bits reproduce sparse exactly; argmax row is a valid row attaining the max
    kk = keys.to(torch.int64) & 0xFFFFFFFF
    bits = (kk >> 16).to(torch.int32)
    val = (bits << 16).view(torch.float32)
    assert torch.equal(torch.log1p(val), sp)
    row = (0xFFFF - (kk & 0xFFFF)).clamp(max=S - 1)
    pos = val > 0
    relu_logits = torch.relu(logits).float().view(B, S, V)
    at_row = torch.gather(relu_logits, 1, row.view(B, 1, V)).view(B, V)
    assert torch.equal(at_row[pos], val[pos]) or ((at_row[pos] - val[pos]).abs() <= 2 ** -7 * val[pos]).all()
    assert (torch.gather(mask.to(dev), 1, row)[pos] == 1).all()
    # first-index tie rule: no earlier valid row has the same (bf16) value
    same = (relu_logits == val.view(B, 1, V)) & mask.to(dev)[:, :, None].bool()
    first = torch.where(same.any(1), same.float().argmax(1), torch.zeros_like(row))
    agree = (first == row) | ~pos | ((at_row - val).abs() > 0)
    assert agree.float().mean().item() > 0.999


@pytest.mark.parametrize("B,S", [(6, 64), (24, 128)])
def test_decoder_splade_fwd_irregular_masks(dev, B, S):
    """Masks that are not right padding: holes, left padding, a single valid row in the middle and a fully masked
    sequence -- the 256x192 kernel (B * S >= 2048) compacts the valid rows and reports sequence positions, the 128x128
    kernel (small case) masks rows in its epilogue; both must give the masked reference and never pick a masked row."""
    ops = _ops()
    V, K = 1000, 256
    T = B * S
    g = torch.Generator().manual_seed(B * S)
    mask = (torch.rand(B, S, generator=g) < 0.7).long()
    mask[1, : S // 3] = 0                                   # left padding
    mask[1, S // 3:] = 1
    mask[2] = 0
    mask[2, S // 2] = 1                                     # one valid row
    mask[3] = 0                                             # fully masked
    hd = torch.randn(T, K, generator=g).to(BF16).to(dev)
    W = (torch.randn(V, K, generator=g) * 0.05).to(BF16).to(dev)
    bias = (torch.randn(V, generator=g) * 0.3).to(dev)
    cu = (torch.arange(B + 1, dtype=torch.int32) * S).to(dev)
    md = mask.to(dev)
    sp, keys, tw = ops.decoder_splade_fwd(hd, W, bias, cu, md.reshape(-1), S)
    logits = ((hd.float() @ W.float().t()) + bias.to(BF16).float()).to(BF16)
    sc = torch.log1p(torch.relu(logits).float()).view(B, S, V) * md[:, :, None].float()
    ref_sp, ref_tw = sc.max(dim=1).values, sc.max(dim=-1).values.reshape(T)
    d = (sp - ref_sp).abs()
    assert d.max().item() < 0.02 and (d > 1e-6).float().mean().item() < 0.02
    dt = (tw - ref_tw).abs()
    assert dt.max().item() < 0.02 and (dt > 1e-6).float().mean().item() < 0.03
    assert (tw[md.reshape(-1) == 0] == 0).all()
    assert (sp[3] == 0).all()
    kk = keys.to(torch.int64) & 0xFFFFFFFF
    val = ((kk >> 16).to(torch.int32) << 16).view(torch.float32)
    assert torch.equal(torch.log1p(val), sp)
    row = (0xFFFF - (kk & 0xFFFF)).clamp(max=S - 1)
    pos = val > 0
    assert (torch.gather(md, 1, row)[pos] == 1).all()      # the arg-max row is a valid row ...
    relu_logits = torch.relu(logits).float().view(B, S, V)
    at_row = torch.gather(relu_logits, 1, row.view(B, 1, V)).view(B, V)
    assert ((at_row[pos] - val[pos]).abs() <= 2 ** -7 * val[pos]).all()   # ... that attains the maximum ...
    same = (relu_logits == val.view(B, 1, V)) & md[:, :, None].bool()
    first = torch.where(same.any(1), same.float().argmax(1), torch.zeros_like(row))
    agree = (first == row) | ~pos | ((at_row - val).abs() > 0)
    assert agree.float().mean().item() > 0.999             # ... and the first one that does


def _natural(dy, inter):
    """interleaved GeGLU columns [a(32) g(32)]* -> natural [a | g] (fp64)"""
    d = dy.double()
    if not inter:
        return d
    M, N = d.shape
    v = d.view(M, N // 64, 2, 32)
    return torch.cat([v[:, :, 0].reshape(M, -1), v[:, :, 1].reshape(M, -1)], 1)


def _tn_check(dw, dw0, dy, x, inter=False, what=""):
    """dW = dW0 + dY^T X against the fp64 product of the same bf16 operands, with a DERIVED bound: the products of
    two bf16 values are exact in fp32, so the only error is fp32 summation -- whatever the tree (MFMA k-order inside a
    workgroup, token pieces, ordered or atomic reduction), |error| <= gamma_n * (|dW0| + sum_t |dy||x|) elementwise
    with n = M + 8 additions and gamma_n = n u / (1 - n u), u = 2^-24 (Higham, Accuracy and Stability, 4.2).  A
    dropped, duplicated or misplaced token piece is an error of the size of a partial sum, orders above that."""
    d, xx = _natural(dy, inter), x.double()
    ref = dw0.double() + d.t() @ xx
    mag = dw0.double().abs() + d.abs().t() @ xx.abs()
    n = dy.shape[0] + 8
    gamma = n * 2.0 ** -24 / (1 - n * 2.0 ** -24)
    err = (dw.double() - ref).abs()
    worst = float((err / (gamma * mag + 1e-30)).max())
    assert worst <= 1.0, (what, worst, float(err.max()))
    return ref


@pytest.mark.parametrize("M,N,K", [(64, 128, 128), (1000, 768, 256), (4096, 2304, 768), (777, 768, 1152), (200, 256, 384)])
def test_gemm_tn_accum(dev, M, N, K):
    ops = _ops()
    g = torch.Generator().manual_seed(M + K)
    dy = (torch.randn(M, N, generator=g) * 0.1).to(dev).to(BF16)
    x = torch.randn(M, K, generator=g).to(dev).to(BF16)
    dw0 = torch.randn(N, K, generator=g).to(dev)
    dw = dw0.clone()
    ops.gemm_tn_accum(dy, x, dw)
    _tn_check(dw, dw0, dy, x, what=(M, N, K))


def test_weight_gradient_reductions_are_bit_reproducible(dev):
    """Round-4 review, item 1: the weight-gradient GEMMs, the LayerNorm weight gradients and the embedding gradient used
    to add per-workgroup partial sums with float atomics (arrival order: last-bit differences from run to run, and three
    sample-derived test tolerances that went red).  Since round 5 every one of them reduces in a FIXED order through a
    caller-owned workspace (include/snx.h "det_reduce"): repeated launches must give the same BITS -- the 256x256
    persistent kernel with its token pieces + stream-K tail + ragged rest, the 128x128 token-split kernel, grouped and
    alone -- and the float-atomic form ("det_reduce" = 0) must agree with them within the derived bound."""
    import snx
    ops = _ops()
    g = torch.Generator().manual_seed(77)
    mk = lambda r, c, s=0.1: (torch.randn(r, c, generator=g) * s).to(dev).to(BF16)   # noqa: E731
    H, I = 768, 1152
    for M in (8192 + 64 * 3 + 17, 3000):                     # 256x256 kernel + ragged rest; 128x128 kernel with splits
        shapes = [(3 * H, H, False), (2 * I, H, True), (H, I, False), (H, H, False)]
        ops_in = [(mk(M, N), mk(M, K, 1.0), torch.randn(N, K, generator=g).to(dev), inter) for N, K, inter in shapes]
        outs = []
        for rep in range(3):
            probs = [(dy, x, dw0.clone(), inter) for dy, x, dw0, inter in ops_in]
            ops.gemm_tn_accum_group(probs)
            outs.append([p[2] for p in probs])
        for a, b_, c in zip(*outs):
            assert torch.equal(a, b_) and torch.equal(a, c), M
        for (dy, x, dw0, inter), dw in zip(ops_in, outs[0]):
            _tn_check(dw, dw0, dy, x, inter, what=("grouped", M))
        snx.configure(det_reduce=0)
        try:
            probs = [(dy, x, dw0.clone(), inter) for dy, x, dw0, inter in ops_in]
            ops.gemm_tn_accum_group(probs)
        finally:
            snx.configure(det_reduce=1)
        for (dy, x, dw0, inter), p_ in zip(ops_in, probs):
            _tn_check(p_[2], dw0, dy, x, inter, what=("atomics", M))
    # LayerNorm weight gradient: the blocks' partial rows added in block order
    T = 9000
    h = torch.randn(T, H, generator=g).to(dev)
    w = (1 + 0.1 * torch.randn(H, generator=g)).to(dev)
    dy = mk(T, H, 1.0)
    res = []
    for rep in range(3):
        dh, dw = torch.zeros(T, H, device=dev), torch.zeros(H, device=dev)
        ops.ln_bwd(dy, h, w, dh, dw, 1e-5)
        res.append((dh, dw))
    assert all(torch.equal(res[0][0], r[0]) and torch.equal(res[0][1], r[1]) for r in res[1:])
    hd = h.double()
    xhat = (hd - hd.mean(-1, keepdim=True)) / torch.sqrt(hd.var(-1, unbiased=False, keepdim=True) + 1e-5)
    ref = (dy.double() * xhat).sum(0)
    mag = (dy.double() * xhat).abs().sum(0)
    assert float(((res[0][1].double() - ref).abs() / (mag * (T + 64) * 2.0 ** -23 + 1e-30)).max()) <= 1.0
    # embedding gradient: rows of one id summed in ascending token order; an id used by MANY tokens (every sequence
    # starts with <s>), a pad id that takes no part, and the degenerate batch of one repeated id
    V = 1000
    E = torch.randn(V, H, generator=g).to(dev)
    for ids in (torch.randint(3, V, (T,), generator=g), torch.full((T,), 7, dtype=torch.int64)):
        ids[::64] = 0
        ids[5::97] = V - 1                                   # pad
        ids = ids.to(dev)
        dh = torch.randn(T, H, generator=g).to(dev)
        got = []
        for rep in range(3):
            gradE, dw = torch.randn(V, H, generator=torch.Generator().manual_seed(3)).to(dev), torch.zeros(H, device=dev)
            g0 = gradE.clone()
            ops.embed_ln_bwd(dh, ids, E, w, gradE, dw, 1e-5, V - 1)
            got.append((gradE, dw))
        assert all(torch.equal(got[0][0], r[0]) and torch.equal(got[0][1], r[1]) for r in got[1:])
        assert torch.equal(got[0][0][V - 1], g0[V - 1])      # nn.Embedding(padding_idx): the pad row gets nothing
        # reference: autograd through LN(E[ids]) in fp64
        Ed = E.double().requires_grad_(True)
        wd = w.double().requires_grad_(True)
        xe = Ed[ids]
        y = (xe - xe.mean(-1, keepdim=True)) / torch.sqrt(xe.var(-1, unbiased=False, keepdim=True) + 1e-5) * wd
        y.backward(dh.double())
        refE = g0.double() + Ed.grad
        refE[V - 1] = g0[V - 1].double()
        cnt = torch.bincount(ids, minlength=V).double().clamp(min=1)[:, None]
        scale = g0.double().abs() + cnt * float(dh.abs().max()) * 4      # |dx| <~ 4 max|dh| rows, cnt of them per id
        assert float(((got[0][0].double() - refE).abs() / (scale * 1e-5)).max()) <= 1.0
        assert torch.allclose(got[0][1].double(), wd.grad, rtol=1e-4, atol=1e-3 * float(wd.grad.abs().max()))


def test_gemm_tn_group_equals_separate_launches(dev):
    """The four weight gradients of one encoder layer in ONE grouped launch (different N, K, one of them with the
    interleaved GeGLU column order) == four separate launches, up to the order of the fp32 atomic adds."""
    ops = _ops()
    M, H, I = 3000, 256, 384
    g = torch.Generator().manual_seed(8)
    mk = lambda r, c, s=0.1: (torch.randn(r, c, generator=g) * s).to(dev).to(BF16)   # noqa: E731
    shapes = [(3 * H, H, False), (2 * I, H, True), (H, I, False), (H, H, False)]
    probs, refs = [], []
    for N, K, inter in shapes:
        dy, x = mk(M, N), mk(M, K, 1.0)
        dw = torch.randn(N, K, generator=g).to(dev)
        ref = dw.clone()
        (ops.gemm_tn_accum_interleaved if inter else ops.gemm_tn_accum)(dy, x, ref)
        probs.append((dy, x, dw, inter))
        refs.append(ref)
    ops.gemm_tn_accum_group(probs)
    for (dy, x, dw, inter), ref in zip(probs, refs):
        assert torch.allclose(dw, ref, rtol=1e-4, atol=1e-3), float((dw - ref).abs().max())
    with pytest.raises(ValueError):
        ops.gemm_tn_accum_group(probs + probs[:1])


@pytest.mark.parametrize("M", [8192, 8263, 20000])
def test_gemm_tn_256_form(dev, M):
    """Token ranges of 8192 rows and more run the 256x256 persistent kernel (gemm_tn256.hip): one long item per
    (token piece, tile) workgroup plus the stream-K tail workgroups, a ragged rest of the token range (M % 64 rows,
    handed to the 128x128 kernel), tiles half outside the matrix (N or K = 128 mod 256), the interleaved GeGLU row
    order, alone and grouped.  Reference: fp32 matmul of the same bf16 operands."""
    ops = _ops()
    g = torch.Generator().manual_seed(M)
    mk = lambda r, c, s=0.1: (torch.randn(r, c, generator=g) * s).to(dev).to(BF16)   # noqa: E731
    for shapes in ([(768, 384, False), (768, 384, True), (256, 1152, False), (512, 384, False)],
                   [(768, 384, False), (384, 256, False)]):
        probs, refs = [], []
        for N, K, inter in shapes:
            dy, x = mk(M, N), mk(M, K, 1.0)
            dw = torch.randn(N, K, generator=g).to(dev)
            refs.append(dw.clone())
            probs.append((dy, x, dw, inter))
        singles = [(dy, x, dw.clone(), inter) for dy, x, dw, inter in probs]
        ops.gemm_tn_accum_group(probs)
        for (dy, x, dw, inter), dw0 in zip(probs, refs):
            _tn_check(dw, dw0, dy, x, inter, what=("group", M))
        for (dy, x, dw, inter), dw0 in zip(singles, refs):
            (ops.gemm_tn_accum_interleaved if inter else ops.gemm_tn_accum)(dy, x, dw)
            _tn_check(dw, dw0, dy, x, inter, what=("single", M))


@pytest.mark.parametrize("reserved", [8, 16, 32, 29])
def test_gemm_tn_256_with_reserved_cus(dev, reserved):
    """While a gradient bucket is exchanged the persistent dW kernel leaves CUs to RCCL's channel workgroups
    (snx_set_reserved_cus: 256 - n workgroups, n rounded up to whole rounds of the 8 XCDs).  Its schedule (one long
    item per workgroup + stream-K tail) must balance any count: same fp32 reference as test_gemm_tn_256_form at 248 /
    240 / 224 workgroups, alone and as the layer group of the training step (which no longer fits 248 pieces x tiles
    exactly)."""
    from snx._lib import fn
    ops = _ops()
    M = 12288
    g = torch.Generator().manual_seed(reserved)
    mk = lambda r, c, s=0.1: (torch.randn(r, c, generator=g) * s).to(dev).to(BF16)   # noqa: E731
    assert fn("snx_set_reserved_cus")(reserved) == 0
    try:
        assert fn("snx_get_reserved_cus")() == (reserved + 7) // 8 * 8
        for shapes in ([(2304, 768), (2304, 768), (768, 1152), (768, 768)], [(768, 384)]):
            probs, refs = [], []
            for N, K in shapes:
                dy, x = mk(M, N), mk(M, K, 1.0)
                dw = torch.randn(N, K, generator=g).to(dev)
                refs.append(dw.clone())
                probs.append((dy, x, dw, False))
            again = [(dy, x, dw.clone(), False) for dy, x, dw, _ in probs]
            ops.gemm_tn_accum_group(probs)
            ops.gemm_tn_accum_group(again)
            for (dy, x, dw, _), dw0, (_, _, dw2, _) in zip(probs, refs, again):
                _tn_check(dw, dw0, dy, x, what=("reserved", reserved))
                assert torch.equal(dw, dw2)              # every workgroup count is bit-reproducible by itself
    finally:
        fn("snx_set_reserved_cus")(0)
    assert fn("snx_set_reserved_cus")(-1) != 0 and fn("snx_set_reserved_cus")(129) != 0      # argument check
    assert fn("snx_get_reserved_cus")() == 0


@pytest.mark.parametrize("reserved", [0, 8, 32])
def test_gemm_nt256_with_reserved_cus(dev, reserved):
    """The 256x256 persistent NT kernel (gemm_nt256.hip) with 256 / 248 / 224 workgroups: bit-equal to the 128x128
    kernel (both sum k in the same order) on a shape with short tiles and a ragged last panel, plain and residual."""
    from snx._lib import fn
    ops = _ops()
    M, N, K = 9000 + reserved, 768, 384
    g = torch.Generator().manual_seed(7 + reserved)
    x = (torch.randn(M, K, generator=g)).to(dev).to(BF16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(BF16)
    hin = torch.randn(M, N, generator=g).to(dev)
    fn("snx_nt256_configure")(0, 0)
    ref, ref_h = ops.gemm_nt(x, w), ops.gemm_nt_resid(x, w, hin)
    try:
        fn("snx_nt256_configure")(2, 1024)
        assert fn("snx_set_reserved_cus")(reserved) == 0
        got, got_h = ops.gemm_nt(x, w), ops.gemm_nt_resid(x, w, hin)
        torch.cuda.synchronize()
    finally:
        fn("snx_set_reserved_cus")(0)
        fn("snx_nt256_configure")(1, 8192)
    assert torch.equal(ref, got) and torch.equal(ref_h, got_h)
    assert float((got.float() - x.float() @ w.float().t()).abs().max()) < 0.05


@pytest.mark.parametrize("M,N,K", [(8192, 768, 768), (36864, 2304, 768), (9000, 1152, 768), (8200, 64, 64), (12345, 768, 128),
                                   (20000, 2304, 192), (36864, 768, 2304), (4100, 320, 256), (16448, 512, 64),
                                   (8193, 832, 1152), (10007, 1600, 320), (33333, 256, 448)])
def test_gemm_nt256_all_epilogues_equal_the_128_kernel_bit_for_bit(dev, M, N, K):
    """The round-3 kernel on every shape class it can meet: whole rounds only, short tiles of 64 / 128 / 192 rows, a ragged
    last panel (M % 64 != 0), half-empty column tiles (N % 256 != 0), one K-tile (K = 64) and 36 of them, fewer tiles than
    workgroups -- with all its epilogues (store, fp32 residual, RoPE from the table and from pre-resolved rows, GeGLU
    forward, GeGLU backward), forced on (`snx_nt256_configure(2, ...)`), against the 128x128 kernel, which sums k in the
    same order: every output tensor must be equal bit for bit.  (The 128x128 kernel itself is held to fp32 torch by the
    tests above.)"""
    from snx._lib import fn
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dev).to(BF16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(BF16)
    hin = torch.randn(M, N, generator=g).to(dev)
    S = 256
    tab = ops.rope_table(S, 64, 160000.0, dev)
    pos = (torch.arange(M, dtype=torch.int32, device=dev) % S).contiguous()
    rc = (2 * N // 3) // 64 * 64
    rows = ops.rope_rows(tab, pos)
    u = torch.randn(M, 2 * N, generator=g).to(dev).to(BF16)
    calls = {"store": lambda: (ops.gemm_nt(x, w),), "resid": lambda: (ops.gemm_nt_resid(x, w, hin),)}
    if N % 64 == 0:
        calls["rope"] = lambda: (ops.gemm_nt_rope(x, w, tab, pos, rc, validate=False),)
        calls["rope_rows"] = lambda: (ops.gemm_nt_rope_rows(x, w, tab, pos, rows, rc),)
        calls["geglu_fwd"] = lambda: ops.gemm_nt_geglu_fwd(x, w)
        calls["geglu_bwd"] = lambda: (ops.gemm_nt_geglu_bwd(x, w, u),)
    try:
        for name, f in calls.items():
            fn("snx_nt256_configure")(0, 0)
            ref = [t.clone() for t in f()]
            fn("snx_nt256_configure")(2, 1024)
            got = f()
            torch.cuda.synchronize()
            for a, b in zip(ref, got):
                assert torch.equal(a, b), (name, M, N, K, float((a.float() - b.float()).abs().max()))
    finally:
        fn("snx_nt256_configure")(1, 8192)


@pytest.mark.parametrize("M,N,K", [(36864, 768, 2304), (36864, 768, 768), (36864, 2304, 768), (20037, 768, 2304),
                                   (9000, 256, 768), (33333, 1280, 320)])
def test_gemm_nt256_column_run_dealing_of_the_leftover_units(dev, M, N, K):
    """Round 6: the 64-row units left after the whole rounds of tiles are dealt along the leftover tiles' COLUMN runs (one
    short tile per workgroup, its rows free to cross a row-panel boundary) instead of in tile order (two short tiles for
    three workgroups in eight at N = 768).  A tile's K loop does not depend on which rows share it: plain store, RoPE and
    GeGLU-forward outputs must equal the tile-order dealing's bit for bit (and, through the test above, the 128x128
    kernel's)."""
    from snx._lib import fn
    ops = _ops()
    g = torch.Generator().manual_seed(M + 3 * N + K)
    x = torch.randn(M, K, generator=g).to(dev).to(BF16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(BF16)
    tab = ops.rope_table(256, 64, 160000.0, dev)
    pos = (torch.arange(M, dtype=torch.int32, device=dev) % 256).contiguous()
    rows = ops.rope_rows(tab, pos)
    rc = (2 * N // 3) // 64 * 64
    calls = {"store": lambda: (ops.gemm_nt(x, w),), "rope_rows": lambda: (ops.gemm_nt_rope_rows(x, w, tab, pos, rows, rc),),
             "geglu_fwd": lambda: ops.gemm_nt_geglu_fwd(x, w)}
    try:
        fn("snx_nt256_configure")(2, 1024)
        for name, f in calls.items():
            assert fn("snx_configure")(b"nt256_coldeal", 0) == 0 and fn("snx_configure")(b"nt256_rev", 0) == 0
            ref = [t.clone() for t in f()]
            # ... and "nt256_rev" (the K >= 3 N plain stores walk their row panels from the last to the first): another
            # tile -> workgroup assignment again, the same tiles
            for coldeal, rev in ((1, 0), (1, 1), (0, 1)):
                assert fn("snx_configure")(b"nt256_coldeal", coldeal) == 0 and fn("snx_configure")(b"nt256_rev", rev) == 0
                got = f()
                torch.cuda.synchronize()
                for a, b in zip(ref, got):
                    assert torch.equal(a, b), (name, M, N, K, coldeal, rev, float((a.float() - b.float()).abs().max()))
    finally:
        fn("snx_configure")(b"nt256_coldeal", 1)
        fn("snx_configure")(b"nt256_rev", 0)
        fn("snx_nt256_configure")(1, 8192)


@pytest.mark.parametrize("M", [8192 + 64, 8192 + 37])
def test_gemm_tn_256_layout(dev, M):
    """Exact check of the 256x256 form: dY = a 0/1 selection pattern, so dW[n, :] = X[row(n), :] bit for bit
    (catches n/k swaps, sub-tile column maps, token-piece seams and the ragged rest)."""
    ops = _ops()
    N, K = 512, 768
    perm = (torch.arange(N, device=dev) * 37 + 5) % M
    perm[-1] = M - 1
    dy = torch.zeros(M, N, device=dev)
    dy[perm, torch.arange(N, device=dev)] = 1.0
    x = ((torch.arange(M * K, device=dev, dtype=torch.float32).reshape(M, K) % 199) - 99.0).to(BF16)
    dw = torch.zeros(N, K, device=dev)
    ops.gemm_tn_accum(dy.to(BF16), x, dw)
    assert torch.equal(dw, x.float()[perm])


def test_gemm_tn_layout(dev):
    """dY = shifted identity pattern: dW[n, :] must equal X[row(n), :] (catches n/k swaps)."""
    ops = _ops()
    M, N, K = 128, 128, 256
    dy = torch.zeros(M, N, device=dev)
    perm = (torch.arange(N, device=dev) * 37 + 5) % M
    dy[perm, torch.arange(N, device=dev)] = 1.0
    x = ((torch.arange(M * K, device=dev, dtype=torch.float32).reshape(M, K) % 199) - 99.0).to(BF16)
    dw = torch.zeros(N, K, device=dev)
    ops.gemm_tn_accum(dy.to(BF16), x, dw)
    assert torch.equal(dw, x.float()[perm])


# (512, *): BASELINE config 5 documents -- streaming kernels, 8 key tiles on global layers, 129-key band on local ones
@pytest.mark.parametrize("S,window", [(64, -1), (256, 64), (200, 8), (130, -1), (320, 64), (300, -1), (512, -1), (512, 64)])
def test_attention_bwd(dev, S, window):
    from oracle import splade_oracle as O
    ops = _ops()
    B, heads = 3, 2
    T = B * S
    lens, mask = _ragged(B, S, 3 * S + window)
    g = torch.Generator().manual_seed(21)
    qkv = (torch.randn(T, 3 * heads * 64, generator=g) * 1.2).to(BF16)
    dout = (torch.randn(T, heads * 64, generator=g) * 0.5).to(BF16)
    dout[mask.reshape(-1) == 0] = 0          # padded rows never receive gradient
    cu = torch.arange(B + 1, dtype=torch.int32) * S
    qd, cud, md = qkv.to(dev), cu.to(dev), mask.reshape(-1).to(dev)
    out, lse = ops.attn_fwd(qd, cud, md, S, heads, window)
    dqkv = ops.attn_bwd(qd, out, dout.to(dev), lse, cud, md, S, heads, window)
    x = qkv.float().view(B, S, 3, heads, 64).clone().requires_grad_(True)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    vis = O.attention_bias(mask, None if window < 0 else window)
    s = (q @ k.transpose(-1, -2)) * 0.125
    s = s.masked_fill(~vis, float("-inf"))
    rowok = vis.any(-1, keepdim=True)
    p = torch.where(rowok, torch.softmax(s.masked_fill(~rowok, 0.0), dim=-1), torch.zeros_like(s))
    o = (p @ v).transpose(1, 2).reshape(T, heads * 64)
    (o * dout.float()).sum().backward()
    ref = x.grad.view(T, 3 * heads * 64)
    valid = mask.reshape(-1).bool()
    got = dqkv.cpu().float()
    for name, sl in (("dq", slice(0, heads * 64)), ("dk", slice(heads * 64, 2 * heads * 64)), ("dv", slice(2 * heads * 64, None))):
        gg, rr = got[valid][:, sl].double().flatten(), ref[valid][:, sl].double().flatten()
        cos = float(gg @ rr / (gg.norm() * rr.norm()))
        rel = float((gg - rr).norm() / rr.norm())
        assert cos > 0.9995 and rel < 2e-2, (name, cos, rel)
    assert torch.isfinite(got).all()
    if (~valid).any():   # keys at padded positions get no gradient
        assert got[~valid][:, heads * 64:].abs().max().item() == 0.0


# (2, 1024, ...): vocabulary-ordered buckets with > 64 KiB of LDS; (1, 1100, ...): beyond 1024 rows the per-wave slot
# tables no longer fit and the bucket pass falls back to first-come slots; (2, 512, 50000, 768): config-5 documents
@pytest.mark.parametrize("B,S,V,H", [(3, 64, 1000, 256), (4, 256, 3000, 768), (2, 200, 640, 256), (2, 1024, 700, 256),
                                     (1, 1100, 512, 256), (2, 512, 50000, 768)])
def test_splade_bwd(dev, B, S, V, H, monkeypatch):
    ops = _ops()
    from snx._lib import fn, check
    from snx.ops import _p, _stream
    T = B * S
    lens, mask = _ragged(B, S, V + 1)
    g = torch.Generator().manual_seed(23)
    hd = torch.randn(T, H, generator=g).to(BF16)
    W = (torch.randn(V, H, generator=g) * 0.05).to(BF16)
    bias = torch.randn(V, generator=g) * 0.3
    gs = torch.randn(B, V, generator=g)
    gs[:, ::7] = 0.0
    cu = torch.arange(B + 1, dtype=torch.int32) * S
    hdd, Wd, bd, cud, md = hd.to(dev), W.to(dev), bias.to(dev), cu.to(dev), mask.reshape(-1).to(dev)
    sp, keys, tw = ops.decoder_splade_fwd(hdd, Wd, bd, cud, md, S)
    dHd = torch.full((T, H), float("nan"), dtype=BF16, device=dev)
    gE0 = torch.randn(V, H, generator=g).to(dev)
    gb0 = torch.randn(V, generator=g).to(dev)
    gE, gb = gE0.clone(), gb0.clone()
    scratch = torch.empty(fn("snx_splade_bwd_scratch_bytes")(B, S, V), dtype=torch.uint8, device=dev)
    check(fn("snx_splade_bwd")(_p(gs.to(dev)), _p(keys), _p(hdd), _p(Wd), _p(cud), _p(dHd), _p(gE), _p(gb),
                               _p(scratch), T, B, S, V, H, _stream()), "snx_splade_bwd")
    # the dHd gather has two forms (one wave per row / eight rows per wave walking the vocabulary in panels, the default
    # for vocabulary-ordered buckets): same accumulation order per row, so the same bits
    dHd_rows = torch.full((T, H), float("nan"), dtype=BF16, device=dev)
    import snx
    snx.configure(splade_dh_panels=0)
    gE2, gb2 = gE0.clone(), gb0.clone()
    try:
        check(fn("snx_splade_bwd")(_p(gs.to(dev)), _p(keys), _p(hdd), _p(Wd), _p(cud), _p(dHd_rows), _p(gE2), _p(gb2),
                                   _p(scratch), T, B, S, V, H, _stream()), "snx_splade_bwd")
    finally:
        snx.configure(splade_dh_panels=16)
    assert torch.equal(dHd.view(torch.int16), dHd_rows.view(torch.int16))
    # dense reference through autograd on the bf16 logits
    hl = hdd.float().requires_grad_(True)
    Wl = Wd.float().requires_grad_(True)
    bl = bd.to(BF16).float().requires_grad_(True)
    logits = (hl @ Wl.t() + bl).to(BF16)
    sc = torch.log1p(torch.relu(logits).float()).view(B, S, V) * md.view(B, S, 1).float()
    sc.max(dim=1).values.backward(gs.to(dev))
    assert torch.isfinite(dHd.float()).all()
    for name, got, ref in (("dHd", dHd.float(), hl.grad), ("dW", gE - gE0, Wl.grad), ("db", gb - gb0, bl.grad)):
        gg, rr = got.double().flatten(), ref.double().flatten()
        cos = float(gg @ rr / (gg.norm() * rr.norm() + 1e-30))
        rel = float((gg - rr).norm() / (rr.norm() + 1e-30))
        assert cos > 0.999 and rel < 2e-2, (name, cos, rel)


def _interleave_cols(x, I):
    """natural [.., 2I] (a | g) -> interleaved column order used by the fused GeGLU GEMMs."""
    n = torch.arange(2 * I, device=x.device)
    src = torch.where((n % 64) < 32, 32 * (n // 64) + (n % 32), I + 32 * (n // 64) + (n % 32))
    return x[..., src]


def test_fused_wqkv_rope_epilogue(dev):
    ops = _ops()
    B, S, heads = 3, 70, 4
    H = heads * 64
    T = B * S
    g = torch.Generator().manual_seed(31)
    x = torch.randn(T, H, generator=g).to(dev).to(BF16)
    w = (torch.randn(3 * H, H, generator=g) * 0.05).to(dev).to(BF16)
    pos = torch.arange(S, dtype=torch.int32).repeat(B).to(dev)
    tab = ops.rope_table(128, 64, 10000.0, dev)
    fused = ops.gemm_nt_rope(x, w, tab, pos, 2 * H)
    ref = ops.gemm_nt(x, w)
    ops.rope_inplace(ref, tab, pos, heads)
    assert torch.equal(fused[:, 2 * H:], ref[:, 2 * H:])                      # v third untouched
    _ulp_close(fused, ref, "fused rope", frac_exact=0.995)                    # FMA contraction may differ by 1 ulp


@pytest.mark.parametrize("T,H,I", [(300, 256, 384), (1000, 768, 1152)])
def test_fused_geglu_epilogues(dev, T, H, I):
    ops = _ops()
    g = torch.Generator().manual_seed(33)
    x = torch.randn(T, H, generator=g).to(dev).to(BF16)
    wi = (torch.randn(2 * I, H, generator=g) * 0.05).to(dev)
    wom = (torch.randn(H, I, generator=g) * 0.05).to(dev)
    dh = (torch.randn(T, H, generator=g) * 0.1).to(dev).to(BF16)
    wi_il, wi_il_t = ops.cast_geglu_interleave(wi)
    assert torch.equal(wi_il, _interleave_cols(wi.to(BF16).t(), I).t().contiguous())
    assert torch.equal(wi_il_t, wi_il.t().contiguous())
    # forward: fused == gemm + geglu, u in interleaved order
    u_ref = ops.gemm_nt(x, wi.to(BF16))
    y_ref = ops.geglu_fwd(u_ref)
    u, y = ops.gemm_nt_geglu_fwd(x, wi_il)
    assert torch.equal(u, _interleave_cols(u_ref, I)) and torch.equal(y, y_ref)
    # backward: dy = dh @ Wo (as NT with Wo^T), du = GeGLU'(u, dy)
    wom_t = ops.cast_transpose_bf16(wom)                     # [I, H]
    dy = ops.gemm_nt(dh, wom_t)
    du_ref = ops.geglu_bwd(u_ref, dy)
    du = ops.gemm_nt_geglu_bwd(dh, wom_t, u)
    assert torch.equal(du, _interleave_cols(du_ref, I))
    # dW through the interleaved TN GEMM lands in the natural row order
    dw_ref = torch.zeros(2 * I, H, device=dev)
    dw = torch.zeros(2 * I, H, device=dev)
    ops.gemm_tn_accum(du_ref, x, dw_ref)
    ops.gemm_tn_accum_interleaved(du, x, dw)
    assert torch.allclose(dw, dw_ref, rtol=1e-4, atol=1e-4)
    # dX through the interleaved transposed cache
    dx_ref = ops.gemm_nt(du_ref, ops.cast_transpose_bf16(wi))
    dx = ops.gemm_nt(du, wi_il_t)
    _ulp_close(dx, dx_ref, "dx interleaved", frac_exact=0.95)


@pytest.mark.parametrize("M,N", [(36864, 1152), (4096, 128), (4224, 384), (12800, 1152)])
def test_geglu_bwd_pipelined_kernel_equals_the_128_kernel_bit_for_bit(dev, M, N):
    """gemm_nt_pipe.hip: the GeGLU-backward GEMM in persistent workgroups of four MFMA waves + two helper waves -- a
    finished tile is handed over through LDS (bf16 image + parked registers, ordered by the K loop's own barriers) and
    its epilogue (saved u in, GELU / GELU' arithmetic, du out) runs on the helpers while the MFMA waves multiply the
    next tile.  Same products in the same order, same epilogue arithmetic: BIT-identical to the 128x128 kernel of
    gemm.hip (`nt_pipe` = 0) -- at the bench shape (5.06 tiles per workgroup: uneven tile counts, first / middle /
    drain periods), with fewer tiles than workgroups, with one tile per workgroup exactly (first + drain only), and
    three times in a row (a hand-over that races shows up as a run-to-run difference)."""
    import snx
    ops = _ops()
    K = 768
    g = torch.Generator().manual_seed(M + N)
    a = (torch.randn(M, K, generator=g) * 0.5).to(dev).to(BF16)
    b = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(BF16)
    u = torch.randn(M, 2 * N, generator=g).to(dev).to(BF16)
    assert snx.config("nt_pipe") == 2
    outs = [ops.gemm_nt_geglu_bwd(a, b, u) for _ in range(3)]
    try:
        snx.configure(nt_pipe=1)                             # the same kernel with non-temporal du stores
        outs.append(ops.gemm_nt_geglu_bwd(a, b, u))
        snx.configure(nt_pipe=0)
        ref = ops.gemm_nt_geglu_bwd(a, b, u)
    finally:
        snx.configure(nt_pipe=2)
    for o in outs:
        assert torch.equal(o, ref), float((o.float() - ref.float()).abs().max())
    # and against plain math on a slice: dy = bf16(a b^T); da = bf16(bf16(dy g) gelu'(a)), dg = bf16(dy bf16(gelu(a)))
    rows = slice(M - 256, M)
    dy = ops.gemm_nt(a[rows].contiguous(), b).float()        # the kernels' own bf16 dy (same k order: same bits)
    uu = u[rows].float().view(256, N // 32, 2, 32)
    ua, ug = uu[:, :, 0].reshape(256, N), uu[:, :, 1].reshape(256, N)
    phi = 0.5 * (1 + torch.erf(ua.double() / math.sqrt(2))).float()
    gelu = (ua * phi).to(BF16).float()
    dgelu = phi + ua * torch.exp(-0.5 * ua * ua) * 0.3989422804014327
    da = ((dy * ug).to(BF16).float() * dgelu).to(BF16)
    dg = (dy * gelu).to(BF16)
    want = torch.stack([da.view(256, N // 32, 32), dg.view(256, N // 32, 32)], 2).reshape(256, 2 * N)
    _ulp_close(ref[rows], want, "geglu_bwd vs torch", frac_exact=0.9)


def test_attention_bwd_fused_inverse_rope(dev):
    ops = _ops()
    B, S, heads = 2, 130, 2
    T = B * S
    lens, mask = _ragged(B, S, 77)
    g = torch.Generator().manual_seed(41)
    qkv = (torch.randn(T, 3 * heads * 64, generator=g) * 1.2).to(dev).to(BF16)
    dout = (torch.randn(T, heads * 64, generator=g) * 0.5).to(dev).to(BF16)
    cu = (torch.arange(B + 1, dtype=torch.int32) * S).to(dev)
    md = mask.reshape(-1).to(dev)
    pos = torch.arange(S, dtype=torch.int32).repeat(B).to(dev)
    tab = ops.rope_table(256, 64, 160000.0, dev)
    out, lse = ops.attn_fwd(qkv, cu, md, S, heads, 64)
    ref = ops.attn_bwd(qkv, out, dout, lse, cu, md, S, heads, 64)
    ops.rope_inplace(ref, tab, pos, heads, inverse=True)
    fused = ops.attn_bwd(qkv, out, dout, lse, cu, md, S, heads, 64, rope_table=tab, pos=pos)
    assert torch.equal(fused[:, 2 * heads * 64:], ref[:, 2 * heads * 64:])
    _ulp_close(fused, ref, "fused inverse rope", frac_exact=0.995)


@pytest.mark.parametrize("window", [-1, 64])
def test_attention_sequence_groups_only_size_the_launch(dev, window):
    """Sequence groups (short queries + long documents in one token buffer) give every group the tile
    count of its own max length and reorder the blocks; outputs must be bit-identical to the one-group
    launch, and malformed group tables are refused before any launch."""
    ops = _ops()
    from snx._lib import SnxError
    heads = 3
    lens = [64] * 5 + [17, 33] + [256, 200, 129, 70] + [100] * 3
    groups = [(0, 7, 64), (7, 4, 256), (11, 3, 100)]
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
    T = int(cu[-1])
    g = torch.Generator().manual_seed(43)
    qkv = (torch.randn(T, 3 * heads * 64, generator=g) * 1.1).to(dev).to(BF16)
    dout = (torch.randn(T, heads * 64, generator=g) * 0.5).to(dev).to(BF16)
    mask = torch.ones(T, dtype=torch.int64, device=dev)
    out0, lse0 = ops.attn_fwd(qkv, cu, mask, 256, heads, window)
    out1, lse1 = ops.attn_fwd(qkv, cu, mask, 256, heads, window, groups=groups)
    assert torch.equal(out0, out1) and torch.equal(lse0, lse1)
    d0 = ops.attn_bwd(qkv, out0, dout, lse0, cu, mask, 256, heads, window)
    d1 = ops.attn_bwd(qkv, out0, dout, lse0, cu, mask, 256, heads, window, groups=groups)
    assert torch.equal(d0, d1)
    # "attn_interleave" = 1 (round 6, opt-in): the groups' blocks interleaved in proportion to their counts instead of group
    # by group -- another block -> unit bijection, the same units: bit-identical outputs (one-pass and two-pass backward)
    import snx
    try:
        snx.configure(attn_interleave=1)
        out2, lse2 = ops.attn_fwd(qkv, cu, mask, 256, heads, window, groups=groups)
        d2 = ops.attn_bwd(qkv, out0, dout, lse0, cu, mask, 256, heads, window, groups=groups)
        snx.configure(attn_bwd_onepass=0)
        d3 = ops.attn_bwd(qkv, out0, dout, lse0, cu, mask, 256, heads, window, groups=groups)
        snx.configure(attn_interleave=0)
        d4 = ops.attn_bwd(qkv, out0, dout, lse0, cu, mask, 256, heads, window, groups=groups)
    finally:
        snx.configure(attn_interleave=0, attn_bwd_onepass=1)
    assert torch.equal(out0, out2) and torch.equal(lse0, lse2) and torch.equal(d0, d2) and torch.equal(d3, d4)
    for bad in ([(0, 7, 64), (8, 4, 256), (11, 3, 100)],      # gap
                [(0, 7, 64), (7, 4, 256)],                      # does not cover all sequences
                [(0, 14, 300)],                                 # max_len above max_seqlen
                [(i, 1, 64) for i in range(9)]):                # too many groups
        with pytest.raises((SnxError, ValueError)):
            ops.attn_fwd(qkv, cu, mask, 256, heads, window, groups=bad)


def test_gelu_exhaustive_over_bf16_inputs(dev):
    """GELU inputs are bf16 (Linear outputs under autocast), so the device's erfc-form GELU can be checked
    on ALL of them: bf16(gelu(x)) must equal torch's fp32 exact-erf GELU rounded to bf16 for every
    finite x > -3.14 (below that torch's own 1+erf cancellation decides the last bits: compared
    against the float64 value instead), and the derivative must be within fp32 noise of float64."""
    ops = _ops()
    bits = torch.arange(65536, dtype=torch.int32)
    a = (bits << 16).view(torch.float32)
    keep = torch.isfinite(a) & ((a.abs() > 1e-30) | (a == 0)) & (a.abs() < 1e30)
    a = a[keep]
    n = (a.numel() // 128) * 128
    a = a[:n].view(-1, 128)
    u = torch.cat([a, torch.ones_like(a)], dim=1).to(BF16).to(dev)
    y = ops.geglu_fwd(u).float().cpu()
    ref32 = torch.nn.functional.gelu(a).to(BF16).float()
    ref64 = (a.double() * 0.5 * torch.erfc(-a.double() / 2 ** 0.5))
    main = a > -3.14
    assert torch.equal(y[main], ref32[main])
    tail = ~main
    assert torch.allclose(y[tail].double(), ref64[tail], rtol=2 ** -8, atol=1e-7)    # torch fp32 itself: ~3e-7 here
    # derivative: dy = 1, g = 1 -> da = bf16(gelu'(a))
    du = ops.geglu_bwd(u, torch.ones_like(a).to(BF16).to(dev)).float().cpu()
    x = a.double()
    grad64 = 0.5 * torch.erfc(-x / 2 ** 0.5) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5
    got = du[:, :128].double()
    assert torch.allclose(got, grad64, rtol=2 ** -8, atol=4e-7)
    exact = (got == grad64.float().to(BF16).double()).double().mean()
    assert exact > 0.995, exact
    assert torch.equal(du[:, 128:], y)                # dg = dy * bf16(gelu(a))


def _topk_rows(ops, rep, allowed, k):
    vals, ids, cnt, srt = ops.sparse_topk(rep, allowed, k)
    out = []
    for j in range(rep.shape[0]):
        n = int(cnt[j])
        out.append((ids[j, :n].tolist(), vals[j, :n].tolist(), int(srt[j])))
    return out


def test_sparse_topk_matches_reference_encode_batch_golden(dev):
    """Device filter + top-k vs the reference's own `_encode_batch` outputs (g6, ref:benchmark/encoders.py:309-345)
    and vs the oracle restatement: bit-exact ids, weights and ORDER for every row and every top_k."""
    import json
    GOLDEN_DIR = os.path.join(os.path.dirname(__file__), "golden")
    from oracle import splade_oracle as O
    ops = _ops()
    fx = json.load(open(os.path.join(GOLDEN_DIR, "g6_encode_topk.json")))
    tokens, special = fx["tokens"], fx["special"]
    rep = torch.tensor(fx["rep"], dtype=torch.float32)
    allowed = torch.tensor(O.allowed_vocab_mask(tokens, special), dtype=torch.uint8)
    for k, rows in fx["cases"].items():
        top_k = None if k == "None" else min(int(k), rep.shape[1])
        got = _topk_rows(ops, rep.to(dev), allowed.to(dev), top_k)
        for j, want in enumerate(rows):
            ids, ws, _ = got[j]
            assert [tokens[i] for i in ids] == [t for t, _ in want], (k, j)
            assert ws == [float(np.float32(w)) for _, w in want], (k, j)


@pytest.mark.parametrize("V,k", [(50000, 100), (50000, None), (50000, 16384), (1000, 1), (4097, 4096), (50000, 3)])
def test_sparse_topk_vs_oracle_random(dev, V, k):
    from oracle import splade_oracle as O
    ops = _ops()
    g = torch.Generator().manual_seed(V + (k or 0))
    B = 5
    rep = torch.relu(torch.randn(B, V, generator=g) + 0.3)
    rep[1] = (rep[1] * 4).round() / 4                      # heavy ties
    rep[2] = torch.relu(torch.randn(V, generator=g) - 3.5)  # a handful active
    rep[3] = 0
    rep[4] = rep[4].to(BF16).float()                       # what the encoder emits after log1p of bf16 logits: ties
    tokens = [("" if i % 97 == 3 else f"[x{i}]" if i % 89 == 5 else f"<y{i}>" if i % 83 == 7 else f"t{i}") for i in range(V)]
    special = [0, 1, 2, V - 1]
    allowed = torch.tensor(O.allowed_vocab_mask(tokens, special), dtype=torch.uint8)
    got = _topk_rows(ops, rep.to(dev), allowed.to(dev), k)
    for j in range(B):
        want = O.encode_postprocess(rep[j].tolist(), tokens, special, k)
        ids, ws, srt = got[j]
        assert [tokens[i] for i in ids] == [t for t, _ in want]
        assert ws == [w for _, w in want]
        assert srt == int(k is not None and len(want) == k and sum(1 for i in range(V) if rep[j, i] > 0 and allowed[i]) > k)


def test_sparse_topk_rejects_bad_arguments(dev):
    ops = _ops()
    rep = torch.zeros(2, 100, device=dev)
    ok = torch.ones(100, dtype=torch.uint8, device=dev)
    with pytest.raises(ValueError):
        ops.sparse_topk(rep, ok, 0)
    with pytest.raises(ValueError):
        ops.sparse_topk(rep, ok, 20000)
    with pytest.raises(ValueError):
        ops.sparse_topk(rep, ok[:50], 5)
    with pytest.raises(ValueError):
        ops.sparse_topk(rep.to(BF16), ok, 5)


@pytest.mark.parametrize("window", [-1, 64, 8])
def test_attention_resident_and_streaming_kernels_agree_bitwise(dev, window):
    """Sequences of <= 256 tokens take the sequence-resident kernels; declaring the group's max_len > 256 routes the
    same data through the streaming (tile-by-tile) kernels.  Forward, and the two-pass resident backward: same math,
    same accumulation order, identical bits.  The default backward of resident groups is the ONE-PASS kernel
    (csrc/attention_1p.hip: 32x32x16 MFMAs, row constants folded into the accumulators): it sums in another order, so it
    is held to the streaming result within bf16 rounding of the gradients instead (and to the fp32 reference in
    test_attention_bwd)."""
    ops = _ops()
    from snx._lib import fn, check
    heads = 2
    lens = [200, 64, 1, 255, 130, 17]
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
    T = int(cu[-1])
    g = torch.Generator().manual_seed(77 + window)
    qkv = (torch.randn(T, 3 * heads * 64, generator=g) * 1.1).to(dev).to(BF16)
    dout = (torch.randn(T, heads * 64, generator=g) * 0.5).to(dev).to(BF16)
    mask = torch.ones(T, dtype=torch.int64)
    mask[5] = 0                                           # a masked key inside the first sequence
    mask = mask.to(dev)
    res = [(0, 6, 256)]
    stream = [(0, 6, 300)]
    o1, l1 = ops.attn_fwd(qkv, cu, mask, 256, heads, window, groups=res)
    o2, l2 = ops.attn_fwd(qkv, cu, mask, 300, heads, window, groups=stream)
    assert torch.equal(o1, o2) and torch.equal(l1, l2)
    d2 = ops.attn_bwd(qkv, o1, dout, l1, cu, mask, 300, heads, window, groups=stream)
    check(fn("snx_attn_configure")(0), "snx_attn_configure")
    try:
        d1 = ops.attn_bwd(qkv, o1, dout, l1, cu, mask, 256, heads, window, groups=res)
    finally:
        check(fn("snx_attn_configure")(1), "snx_attn_configure")
    assert torch.equal(d1, d2)
    d3 = ops.attn_bwd(qkv, o1, dout, l1, cu, mask, 256, heads, window, groups=res)
    assert torch.isfinite(d3.float()).all()
    a, b = d3.float(), d2.float()
    # two correct bf16 results: a few ulps of the element where it is large, the tensor's rounding noise where it is small
    tol = 2.0 ** -6 * b.abs() + 2.0 ** -8 * b.abs().max()
    assert bool(((a - b).abs() <= tol).all()), float(((a - b).abs() - tol).max())
    rel = float((a - b).double().norm() / b.double().norm())
    assert rel < 6e-3, rel
    assert torch.equal(d3[5, heads * 64:], torch.zeros_like(d3[5, heads * 64:]))     # the masked key: exact zeros
