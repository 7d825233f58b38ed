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
