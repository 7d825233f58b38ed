"""Shared test helpers (no GPU needed)."""
import zlib

import torch


class StubTokenizer:
    """Whitespace tokenizer with the HF call signature: id = 6 + crc32(word) % 40000, <s>=0,
    eos=1, pad=49999.  The same definition was used on the reference side when the collator
    fixture (tests/golden/g5_collator.json) was captured."""
    pad_token_id = 49999

    def __call__(self, texts, padding=True, truncation=True, max_length=64, return_tensors="pt"):
        rows = []
        for t in texts:
            ids = [0] + [6 + zlib.crc32(w.encode()) % 40000 for w in t.split()] + [1]
            if truncation and len(ids) > max_length:
                ids = ids[:max_length - 1] + [1]
            rows.append(ids)
        L = max(len(r) for r in rows)
        ids = torch.full((len(rows), L), self.pad_token_id, dtype=torch.long)
        mask = torch.zeros((len(rows), L), dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = torch.tensor(r)
            mask[i, :len(r)] = 1
        return {"input_ids": ids, "attention_mask": mask}

    def save_pretrained(self, path):
        import os
        os.makedirs(path, exist_ok=True)
        open(os.path.join(path, "stub_tokenizer.txt"), "w").write("stub\n")


def sparse_ulp_stats(got: torch.Tensor, ref: torch.Tensor, logit_abs_tol: float = 2e-3) -> dict:
    """The SURVEY 8(d)(ii) value check as an ULP statement.  Both tensors are fp32 ``log1p(relu(x))`` of a
    bf16 logit x (the reference's cast points), so ``expm1`` recovers x and the two sides can be compared in
    units of the bf16 grid the logit lives on:

      * ``flipped``  -- fraction of entries whose logits are different bf16 values (a one-ulp flip happens when
        fp32 accumulation-order / upstream one-ulp noise straddles a rounding boundary);
      * ``bad``      -- entries that differ by MORE than one bf16 ulp AND by more than ``logit_abs_tol`` in the
        logit (the absolute floor matters only where the bf16 ulp is smaller than the upstream noise,
        i.e. |x| < 0.25, including entries clipped to 0 by the ReLU on one side only);
      * ``max_abs`` / ``mean_abs`` on the sparse values themselves.
    A kernel that is off by several ulps anywhere, or by one ulp everywhere, fails on ``bad`` / ``flipped``."""
    g, r = got.detach().double().cpu(), ref.detach().double().cpu()
    xg, xr = torch.expm1(g).float(), torch.expm1(r).float()
    ig = xg.to(torch.bfloat16).view(torch.int16).to(torch.int32)        # non-negative floats: bit patterns are ordered
    ir = xr.to(torch.bfloat16).view(torch.int16).to(torch.int32)
    d = (ig - ir).abs()
    far = (d > 1) & ((xg - xr).abs() > logit_abs_tol)
    diff = (g - r).abs()
    return {"flipped": float((d != 0).float().mean()), "bad": int(far.sum()), "max_ulps": int(d[~far].max()) if (~far).any() else 0,
            "max_abs": float(diff.max()), "mean_abs": float(diff.mean()), "n": int(d.numel())}


def topk_rank_check(got: torch.Tensor, ref: torch.Tensor, k: int, err: float) -> dict:
    """Top-k index equality wherever the reference's rank gaps allow it: rank i is compared when the
    reference value is more than 2*err away from both neighbours."""
    rv, ri = torch.topk(ref.detach().cpu(), k + 1, dim=-1)
    gv, gi = torch.topk(got.detach().cpu(), k + 1, dim=-1)
    gap = (rv[:, :-1] - rv[:, 1:]) > 2 * err                 # gap[i] = rank i vs i+1
    ok = torch.ones_like(gap[:, :k])
    ok[:, 1:] &= gap[:, :k - 1]
    ok &= gap[:, :k]
    equal = bool(torch.equal(ri[:, :k][ok], gi[:, :k][ok]))
    return {"checked_frac": float(ok.float().mean()), "equal": equal,
            "top_k_sets_equal_rows": float(sum(set(a.tolist()) == set(b.tolist()) for a, b in zip(ri[:, :k], gi[:, :k])) / ri.shape[0])}
