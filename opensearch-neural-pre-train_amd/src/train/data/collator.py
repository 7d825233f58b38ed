"""Tokenizer factory for the trainer (``create_tokenizer`` is imported by
ref:src/train/cli/train_v33_ddp.py:44 from a module that is absent from the reference tree).

Offline rules: a local directory is loaded with ``transformers.AutoTokenizer`` (e.g. an export
like ref:huggingface/v33/ with its BertTokenizer files); the pseudo-name ``hash:<vocab>`` gives
the deterministic whitespace-hash tokenizer used for synthetic runs; a hub NAME cannot be
resolved without a network and raises."""
from __future__ import annotations

import os
import zlib

import torch


class HashTokenizer:
    """HF-call-compatible stand-in: ids = 6 + crc32(word) % (vocab - 7); <s>=0, eos=1, pad=vocab-1
    (the id layout of ref:huggingface/v33/tokenizer_config.json)."""

    def __init__(self, vocab_size: int = 50000):
        self.vocab_size = vocab_size
        self.pad_token_id = vocab_size - 1
        self.bos_token_id, self.eos_token_id = 0, 1
        self.cls_token_id = self.sep_token_id = self.unk_token_id = None

    def convert_ids_to_tokens(self, ids):
        """Synthetic token texts (hashed words have none): specials as "<...>", the rest "w<id>"."""
        names = {0: "<s>", 1: "</s>", self.pad_token_id: "<pad>"}
        return [names.get(i, f"<unused{i}>" if i < 6 else f"w{i}") for i in ids]

    def __call__(self, texts, padding=True, truncation=True, max_length=64, return_tensors="pt"):
        rows = []
        for t in texts:
            ids = [0] + [6 + zlib.crc32(w.encode()) % (self.vocab_size - 7) for w in t.split()] + [1]
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

    def decode(self, ids):
        return " ".join(f"<{i}>" for i in ids)

    def save_pretrained(self, path: str):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "hash_tokenizer.json"), "w") as f:
            f.write('{"type": "hash", "vocab_size": %d}\n' % self.vocab_size)


def create_tokenizer(name: str):
    if isinstance(name, str) and name.startswith("hash:"):
        return HashTokenizer(int(name.split(":")[1]))
    if os.path.isdir(name):
        if os.path.exists(os.path.join(name, "hash_tokenizer.json")):
            import json
            return HashTokenizer(json.load(open(os.path.join(name, "hash_tokenizer.json")))["vocab_size"])
        from transformers import AutoTokenizer
        return AutoTokenizer.from_pretrained(name)
    env = os.environ.get("SNX_MODEL_DIR")
    if env and os.path.isdir(env):
        return create_tokenizer(env)
    raise FileNotFoundError(f"tokenizer {name!r}: not a local directory (no network here). Pass a directory with "
                            "tokenizer files, set SNX_MODEL_DIR, or use 'hash:<vocab_size>' for synthetic runs")
