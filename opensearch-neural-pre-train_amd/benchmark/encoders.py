"""`NeuralSparseEncoderV33` (ref:benchmark/encoders.py:249-402) on the MI355X-native encoder.

Same constructor arguments, methods and return types as the reference class.  What differs underneath:
the forward pass is the HIP path of `SPLADEModernBERT`, and the per-row post-processing of
`_encode_batch` (ref::309-345: copy each row to the host, loop over its non-zeros in Python) is one
device kernel (`snx_sparse_topk`: filter + top-k or id-ordered compaction) followed by ONE device→host
copy of the selected (id, weight) pairs.  Offline: the tokenizer / geometry come from a local directory
(or the built-in stand-ins), never from a hub name.

Assumes the tokenizer's id → token table has no duplicate strings among the kept tokens (the reference keys
its result dict by token text)."""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Dict, List, Optional, Union

import torch

from src.model.splade_modern import SPLADEModernBERT
from src.train.data.collator import create_tokenizer
from snx import ops

logger = logging.getLogger(__name__)


class NeuralSparseEncoderV33:
    """V33 sparse encoder using SPLADEModernBERT (A.X-Encoder-base geometry, 50K vocab)."""

    def __init__(self, checkpoint_path: Union[str, Path, None] = "outputs/train_v33/final_model/model.pt",
                 device: str = "cuda", query_max_length: int = 64, doc_max_length: int = 256,
                 model_name: str = "skt/A.X-Encoder-base", model: Optional[SPLADEModernBERT] = None,
                 tokenizer=None):
        self.device = device
        self.query_max_length = query_max_length
        self.doc_max_length = doc_max_length
        self.tokenizer = tokenizer if tokenizer is not None else create_tokenizer(model_name)
        if model is None:
            model = SPLADEModernBERT(model_name=model_name)
            if checkpoint_path is not None:
                checkpoint_path = Path(checkpoint_path)
                logger.info(f"Loading V33 neural sparse model from: {checkpoint_path}")
                # the reference uses weights_only=False; a state dict needs no unpickling of code
                state_dict = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
                model.load_state_dict(state_dict)
        self.model = model.to(device)
        self.model.eval()
        self.vocab_size = self.tokenizer.vocab_size
        ids = (getattr(self.tokenizer, n, None) for n in ("cls_token_id", "sep_token_id", "pad_token_id",
                                                         "unk_token_id", "bos_token_id", "eos_token_id"))
        self.special_token_ids = {tid for tid in ids if tid is not None}
        self._token_lookup = list(self.tokenizer.convert_ids_to_tokens(list(range(self.vocab_size))))
        self._allowed = None
        logger.info(f"V33 neural sparse model loaded, vocab_size: {self.vocab_size}")

    # ---- device-side filter table: ref:encoders.py:327-331 per vocabulary id ----
    def _allowed_mask(self, V: int, device) -> torch.Tensor:
        if self._allowed is None or self._allowed.numel() != V or self._allowed.device != device:
            m = torch.zeros(V, dtype=torch.uint8)
            for i in range(min(V, len(self._token_lookup))):
                tok = self._token_lookup[i]
                if i not in self.special_token_ids and tok and not tok.startswith(("[", "<")):
                    m[i] = 1
            self._allowed = m.to(device)
        return self._allowed

    def _create_collate_fn(self, max_length: Optional[int] = None):
        effective_length = max_length or self.doc_max_length

        def collate_fn(batch_texts: List[str]):
            return self.tokenizer(batch_texts, return_tensors="pt", padding=True, truncation=True,
                                  max_length=effective_length)
        return collate_fn

    def _postprocess(self, sparse_repr: torch.Tensor, top_k: Optional[int]) -> List[Dict[str, float]]:
        B, V = sparse_repr.shape
        k = None if top_k is None else max(1, min(int(top_k), V))
        if top_k is not None and int(top_k) <= 0:          # ref: len(d) > top_k with top_k <= 0 -> empty slices
            return [dict() for _ in range(B)]
        vals, ids, cnt, _ = ops.sparse_topk(sparse_repr.float().contiguous(), self._allowed_mask(V, sparse_repr.device), k)
        cnt_h = cnt.cpu()
        width = int(cnt_h.max()) if B else 0
        vals_h, ids_h = vals[:, :width].cpu(), ids[:, :width].cpu()
        out = []
        for j in range(B):
            n = int(cnt_h[j])
            toks = ids_h[j, :n].tolist()
            ws = vals_h[j, :n].tolist()
            out.append({self._token_lookup[t]: w for t, w in zip(toks, ws)})
        return out

    @torch.no_grad()
    def _encode_batch(self, inputs: Dict[str, torch.Tensor], top_k: Optional[int] = None) -> List[Dict[str, float]]:
        sparse_repr, _ = self.model(input_ids=inputs["input_ids"], attention_mask=inputs["attention_mask"])
        return self._postprocess(sparse_repr, top_k)

    @torch.no_grad()
    def encode(self, texts: Union[str, List[str]], batch_size: int = 32, num_workers: int = 4,
               top_k: Optional[int] = None) -> List[Dict[str, float]]:
        if isinstance(texts, str):
            texts = [texts]
        collate = self._create_collate_fn()
        all_sparse_vectors: List[Dict[str, float]] = []
        for s in range(0, len(texts), batch_size):          # tokenisation is cheap next to the forward: no worker pool
            inputs = collate(texts[s:s + batch_size])
            inputs = {k: v.to(self.device) for k, v in inputs.items() if torch.is_tensor(v)}
            all_sparse_vectors.extend(self._encode_batch(inputs, top_k))
        return all_sparse_vectors

    def encode_single(self, text: str, top_k: Optional[int] = None) -> Dict[str, float]:
        return self.encode([text], batch_size=1, top_k=top_k)[0]

    @torch.no_grad()
    def encode_for_query(self, text: str, top_k: int = 100) -> Dict[str, float]:
        inputs = self.tokenizer(text, return_tensors="pt", padding=True, truncation=True,
                                max_length=self.query_max_length)
        inputs = {k: v.to(self.device) for k, v in inputs.items() if torch.is_tensor(v)}
        return self._encode_batch(inputs, top_k)[0]
