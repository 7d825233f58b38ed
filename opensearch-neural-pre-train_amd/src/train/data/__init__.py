"""Training-data access for the V33 trainer.

``load_training_data`` is imported by the reference trainer (ref:src/train/cli/train_v33_ddp.py:43)
but its module is absent from the reference tree (swallowed by the repo's own .gitignore); it is
supplied here.  It reads the sharded triplet JSONL written by the reference's preprocessing
(ref:src/preprocessing/pipeline.py:305-352; schema ref:scripts/mine_multi_negatives.py:15-22 and
ref:scripts/precompute_teacher_scores.py:15-20): one JSON object per line with ``query``,
``positive`` and ``negative`` or ``negatives`` (+ optional teacher scores / metadata)."""
from __future__ import annotations

import glob
import json
from typing import Dict, List

from torch.utils.data import Dataset

from .dataloader import TripletCollator, create_dataloader  # noqa: F401


class TripletJsonlDataset(Dataset):
    """Map-style dataset over JSONL shards; lines are indexed by byte offset and parsed lazily."""

    def __init__(self, files: List[str]):
        self.files = files
        self.index: List[tuple] = []
        for fi, path in enumerate(files):
            with open(path, "rb") as f:
                off = 0
                for line in f:
                    if line.strip():
                        self.index.append((fi, off))
                    off += len(line)
        self._handles: Dict[int, object] = {}

    def __len__(self) -> int:
        return len(self.index)

    def __getitem__(self, i: int) -> dict:
        fi, off = self.index[i]
        h = self._handles.get(fi)
        if h is None:
            h = self._handles[fi] = open(self.files[fi], "rb")
        h.seek(off)
        item = json.loads(h.readline())
        if "query" not in item or "positive" not in item:
            raise ValueError(f"{self.files[fi]}@{off}: triplet needs 'query' and 'positive'")
        return item

    def __getstate__(self):           # DataLoader workers re-open their own handles
        st = dict(self.__dict__)
        st["_handles"] = {}
        return st


class SyntheticTripletDataset(Dataset):
    """Deterministic synthetic text triplets (no dataset is reachable offline): word ids drawn
    from a fixed vocabulary, lengths chosen so that q<=64 / d<=256 tokens are exercised."""

    def __init__(self, n: int, num_negatives: int = 1, seed: int = 0, teacher: bool = False,
                 q_words=(4, 62), d_words=(20, 254)):
        import random
        self.n, self.k, self.seed, self.teacher = n, num_negatives, seed, teacher
        self.qw, self.dw = q_words, d_words
        self._rnd = random.Random

    def __len__(self):
        return self.n

    def _text(self, r, lo, hi):
        return " ".join(f"w{r.randrange(30000)}" for _ in range(r.randint(lo, hi)))

    def __getitem__(self, i):
        r = self._rnd(self.seed * 1_000_003 + i)
        item = {"query": self._text(r, *self.qw), "positive": self._text(r, *self.dw)}
        if self.k > 1:
            item["negatives"] = [self._text(r, *self.dw) for _ in range(self.k)]
        else:
            item["negative"] = self._text(r, *self.dw)
        if self.teacher:
            item["teacher_pos_score"] = 0.5 + 0.5 * r.random()
            if self.k > 1:
                item["teacher_neg_scores"] = [0.6 * r.random() for _ in range(self.k)]
            else:
                item["teacher_neg_score"] = 0.6 * r.random()
        return item


def load_training_data(file_globs: List[str]) -> Dataset:
    """Globs -> dataset.  The pseudo-path ``synthetic:N[:k]`` yields N synthetic triplets."""
    if len(file_globs) == 1 and str(file_globs[0]).startswith("synthetic:"):
        parts = str(file_globs[0]).split(":")
        return SyntheticTripletDataset(int(parts[1]), int(parts[2]) if len(parts) > 2 else 1)
    files: List[str] = []
    for g in file_globs:
        files.extend(sorted(glob.glob(g)))
    if not files:
        raise FileNotFoundError(f"no training files match {file_globs}")
    return TripletJsonlDataset(files)
