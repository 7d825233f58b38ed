"""Triplet batching for the SPLADE trainer (host side; produces the hot path's inputs).

Interface and semantics of ref:src/train/data/dataloader.py:46-164 (``TripletCollator``):
asymmetric max lengths for queries/documents, dynamic padding to the longest item, a missing
single negative falls back to the positive text, multi-negative lists are padded to the length
of the FIRST item's list by repeating the last entry (or the positive when empty) and flattened
to ``[B*k, S]``; teacher scores and metadata pass through."""
from typing import Any, Dict, List, Optional

import torch
from torch.utils.data import DataLoader, DistributedSampler


class TripletCollator:
    def __init__(self, tokenizer, max_length: int = 256, query_max_length: Optional[int] = None,
                 doc_max_length: Optional[int] = None, use_in_batch_negatives: bool = True):
        self.tokenizer = tokenizer
        self.max_length = max_length
        self.query_max_length = query_max_length or max_length
        self.doc_max_length = doc_max_length or max_length
        self.use_in_batch_negatives = use_in_batch_negatives

    def _encode(self, texts: List[str], limit: int):
        return self.tokenizer(texts, padding=True, truncation=True, max_length=limit, return_tensors="pt")

    def _negative_texts(self, batch):
        first = batch[0]
        if isinstance(first.get("negatives"), list):
            k = len(first["negatives"])
            flat: List[str] = []
            for item in batch:
                negs = item.get("negatives", [])
                while len(negs) < k:                      # in place, like the reference
                    negs.append(negs[-1] if negs else item["positive"])
                flat.extend(negs[:k])
            return flat, k, True
        flat = []
        for item in batch:
            neg = item.get("negative")
            flat.append(item["positive"] if neg is None else neg)
        return flat, 1, False

    def __call__(self, batch: List[Dict[str, Any]]) -> Dict[str, Any]:
        queries = [it["query"] for it in batch]
        positives = [it["positive"] for it in batch]
        negatives, k, multi = self._negative_texts(batch)
        q = self._encode(queries, self.query_max_length)
        p = self._encode(positives, self.doc_max_length)
        n = self._encode(negatives, self.doc_max_length)
        out: Dict[str, Any] = {
            "query_input_ids": q["input_ids"], "query_attention_mask": q["attention_mask"],
            "positive_input_ids": p["input_ids"], "positive_attention_mask": p["attention_mask"],
            "negative_input_ids": n["input_ids"], "negative_attention_mask": n["attention_mask"],
            "num_negatives": k, "query_texts": queries, "positive_texts": positives,
        }
        head = batch[0]
        if "teacher_pos_score" in head:
            out["teacher_pos_scores"] = torch.tensor([it["teacher_pos_score"] for it in batch], dtype=torch.float32)
        if multi and "teacher_neg_scores" in head:
            out["teacher_neg_scores"] = torch.tensor([it["teacher_neg_scores"] for it in batch], dtype=torch.float32)
        elif "teacher_neg_score" in head:
            out["teacher_neg_scores"] = torch.tensor([it.get("teacher_neg_score", 0.0) for it in batch],
                                                     dtype=torch.float32)
        if "pair_type" in head:
            out["pair_types"] = [it.get("pair_type", "unknown") for it in batch]
        if "difficulty" in head:
            out["difficulties"] = [it.get("difficulty", "medium") for it in batch]
        return out


def create_dataloader(dataset, tokenizer, batch_size: int = 32, max_length: int = 256,
                      query_max_length: Optional[int] = None, doc_max_length: Optional[int] = None,
                      num_workers: int = 4, shuffle: bool = True, use_in_batch_negatives: bool = True,
                      pin_memory: bool = True, drop_last: bool = True, distributed: bool = False,
                      world_size: int = 1, rank: int = 0) -> DataLoader:
    """ref:src/train/data/dataloader.py:167-240."""
    collator = TripletCollator(tokenizer, max_length, query_max_length, doc_max_length, use_in_batch_negatives)
    sampler = DistributedSampler(dataset, num_replicas=world_size, rank=rank, shuffle=shuffle) if distributed else None
    return DataLoader(dataset, batch_size=batch_size, shuffle=(shuffle and sampler is None), sampler=sampler,
                      num_workers=num_workers, collate_fn=collator,
                      pin_memory=pin_memory and torch.cuda.is_available(), drop_last=drop_last)
