"""SPLADE-max model with a ModernBERT backbone -- MI355X-native implementation.

Same public surface as the reference's ``src/model/splade_modern.py:19-114``
(``SPLADEModernBERT(model_name, dropout)``, ``forward``, ``encode``, ``get_top_k_tokens``,
``vocab_size``, ``hidden_size``, ``.config``, ``.model``, ``.relu``) and the same parameter /
state-dict names, but the whole forward and backward -- ModernBERT encoder, MLM
head, tied decoder fused with ``log1p(relu)`` -> mask -> max-pool -- run as hand-written HIP
kernels through libsnx.so (``snx.encoder.EncoderRuntime``).  ``self.model`` only HOLDS the
parameters under their HuggingFace names; it performs no computation.

Differences from the reference, all forced by the offline environment or documented:
  * the constructor never touches the network: ``model_name`` may be a local directory with a
    ``config.json`` (weights are loaded from ``model.safetensors`` / ``model.pt`` when present),
    and the default hub name resolves to the built-in A.X-Encoder-base geometry with RANDOM
    initial weights (transformers modeling_modernbert.py:353-390 recipe) plus a warning;
  * ``token_weights`` is returned but is not differentiable (the reference trainer discards it,
    ref:src/train/cli/train_v33_ddp.py:339-343);
  * ``get_top_k_tokens`` breaks ties lowest-index-first (torch.topk leaves the order open).
"""
from __future__ import annotations

import json
import logging
import math
import os
from types import SimpleNamespace
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from snx.encoder import EncoderGeometry, EncoderRuntime

logger = logging.getLogger(__name__)

_DEFAULT_NAME = "skt/A.X-Encoder-base"


class _Weight(nn.Module):
    """Parameter holder with the attribute layout of nn.Linear / nn.LayerNorm / nn.Embedding."""

    def __init__(self, *shape: int, bias: Optional[int] = None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*shape))
        if bias is not None:
            self.bias = nn.Parameter(torch.zeros(bias))


class _Attention(nn.Module):
    def __init__(self, H: int):
        super().__init__()
        self.Wqkv = _Weight(3 * H, H)
        self.Wo = _Weight(H, H)


class _MLP(nn.Module):
    def __init__(self, H: int, I: int):
        super().__init__()
        self.Wi = _Weight(2 * I, H)
        self.Wo = _Weight(H, I)


class _EncoderLayer(nn.Module):
    def __init__(self, H: int, I: int, idx: int):
        super().__init__()
        self.attn_norm = nn.Identity() if idx == 0 else _Weight(H)   # layer 0 has no attn_norm (hf:309-312)
        self.attn = _Attention(H)
        self.mlp_norm = _Weight(H)
        self.mlp = _MLP(H, I)


class _Embeddings(nn.Module):
    def __init__(self, V: int, H: int):
        super().__init__()
        self.tok_embeddings = _Weight(V, H)
        self.norm = _Weight(H)


class _Backbone(nn.Module):
    def __init__(self, g: EncoderGeometry):
        super().__init__()
        self.embeddings = _Embeddings(g.vocab_size, g.hidden_size)
        self.layers = nn.ModuleList(_EncoderLayer(g.hidden_size, g.intermediate_size, i)
                                    for i in range(g.num_hidden_layers))
        self.final_norm = _Weight(g.hidden_size)


class _Head(nn.Module):
    def __init__(self, H: int):
        super().__init__()
        self.dense = _Weight(H, H)
        self.norm = _Weight(H)


class ModernBertMaskedLMWeights(nn.Module):
    """The parameters of ``transformers.ModernBertForMaskedLM`` under the same names
    (``model.*``, ``head.*``, ``decoder.weight`` tied to the embeddings, ``decoder.bias``)."""

    def __init__(self, geom: EncoderGeometry):
        super().__init__()
        self.geom = geom
        self.model = _Backbone(geom)
        self.head = _Head(geom.hidden_size)
        self.decoder = _Weight(geom.vocab_size, geom.hidden_size, bias=geom.vocab_size)
        self.decoder.weight = self.model.embeddings.tok_embeddings.weight       # tied (hf:499)

    def canonical_parameters(self):
        """Order of include/snx.h: embeddings, per layer [attn_norm], Wqkv, Wo, mlp_norm, Wi, Wo, tail."""
        m = self.model
        out = [m.embeddings.tok_embeddings.weight, m.embeddings.norm.weight]
        for i, layer in enumerate(m.layers):
            if i > 0:
                out.append(layer.attn_norm.weight)
            out += [layer.attn.Wqkv.weight, layer.attn.Wo.weight, layer.mlp_norm.weight,
                    layer.mlp.Wi.weight, layer.mlp.Wo.weight]
        out += [m.final_norm.weight, self.head.dense.weight, self.head.norm.weight, self.decoder.bias]
        return out

    @torch.no_grad()
    def init_weights(self, initializer_range: float = 0.02, cutoff: float = 2.0):
        """hf:353-390: trunc-normal, std=0.02 for embeddings/Wqkv/Wi, 0.02/sqrt(2L) for the output
        projections, head.dense and decoder; LayerNorm weights 1; decoder bias 0."""
        L = self.geom.num_hidden_layers
        std_in, std_out = initializer_range, initializer_range / math.sqrt(2.0 * L)

        def tn(p, std):
            nn.init.trunc_normal_(p, mean=0.0, std=std, a=-cutoff * std, b=cutoff * std)
        m = self.model
        tn(m.embeddings.tok_embeddings.weight, std_in)
        m.embeddings.norm.weight.fill_(1.0)
        for i, layer in enumerate(m.layers):
            if i > 0:
                layer.attn_norm.weight.fill_(1.0)
            layer.mlp_norm.weight.fill_(1.0)
            tn(layer.attn.Wqkv.weight, std_in)
            tn(layer.attn.Wo.weight, std_out)
            tn(layer.mlp.Wi.weight, std_in)
            tn(layer.mlp.Wo.weight, std_out)
        m.final_norm.weight.fill_(1.0)
        tn(self.head.dense.weight, std_out)
        self.head.norm.weight.fill_(1.0)
        self.decoder.bias.zero_()

    def hf_config_dict(self) -> dict:
        g = self.geom
        return {"architectures": ["ModernBertForMaskedLM"], "model_type": "modernbert",
                "vocab_size": g.vocab_size, "hidden_size": g.hidden_size,
                "intermediate_size": g.intermediate_size, "num_hidden_layers": g.num_hidden_layers,
                "num_attention_heads": g.num_attention_heads,
                "global_attn_every_n_layers": g.global_attn_every_n_layers, "local_attention": g.local_attention,
                "global_rope_theta": g.global_rope_theta, "local_rope_theta": g.local_rope_theta,
                "norm_eps": g.norm_eps, "layer_norm_eps": g.norm_eps, "pad_token_id": g.pad_token_id,
                "max_position_embeddings": g.max_position_embeddings, "attention_bias": False, "mlp_bias": False,
                "norm_bias": False, "decoder_bias": True, "classifier_bias": False,
                "hidden_activation": "gelu", "classifier_activation": "gelu", "attention_dropout": 0.0,
                "embedding_dropout": 0.0, "mlp_dropout": 0.0, "classifier_dropout": 0.0,
                "bos_token_id": 0, "eos_token_id": 1, "cls_token_id": 0, "sep_token_id": 1,
                "tie_word_embeddings": True, "dtype": "float32"}

    def save_pretrained(self, save_directory: str, safe_serialization: bool = True, **kwargs):
        """HuggingFace-style export with HF key names, the counterpart of
        ``model.model.save_pretrained(str(output_dir), safe_serialization=True)`` in
        ref:scripts/export_v33_hf.py:28-32: config.json + model.safetensors (``safe_serialization=True``) or
        pytorch_model.bin (False).  The directory loads with stock
        ``transformers.ModernBertForMaskedLM.from_pretrained`` (the tied ``decoder.weight`` is left out, as
        transformers does for tied weights).  Other ``PreTrainedModel.save_pretrained`` keywords are accepted
        and ignored."""
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            json.dump(self.hf_config_dict(), f, indent=2)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items() if k != "decoder.weight"}
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(save_directory, "model.safetensors"), metadata={"format": "pt"})
        else:
            torch.save(sd, os.path.join(save_directory, "pytorch_model.bin"))


def _geometry_from_config(cfg: dict) -> EncoderGeometry:
    g = EncoderGeometry()
    for k in ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
              "global_attn_every_n_layers", "local_attention", "global_rope_theta", "local_rope_theta",
              "pad_token_id", "max_position_embeddings"):
        if k in cfg and cfg[k] is not None:
            setattr(g, k, type(getattr(g, k))(cfg[k]))
    if cfg.get("norm_eps") is not None:
        g.norm_eps = float(cfg["norm_eps"])
    return g


class SPLADEModernBERT(nn.Module):
    """SPLADE-max: MLM logits -> log(1 + ReLU) -> masked max-pool over the sequence."""

    def __init__(self, model_name: str = _DEFAULT_NAME, dropout: float = 0.1, config=None):
        super().__init__()
        self.model_name = model_name
        weights_file = None
        if config is not None:
            geom = config if isinstance(config, EncoderGeometry) else _geometry_from_config(dict(config))
        elif os.path.isdir(model_name):
            with open(os.path.join(model_name, "config.json")) as f:
                geom = _geometry_from_config(json.load(f))
            for cand in ("model.safetensors", "model.pt", "pytorch_model.bin"):
                if os.path.exists(os.path.join(model_name, cand)):
                    weights_file = os.path.join(model_name, cand)
                    break
        else:
            env_dir = os.environ.get("SNX_MODEL_DIR")
            if env_dir and os.path.isdir(env_dir):
                return self.__init__(env_dir, dropout)
            if model_name != _DEFAULT_NAME:
                raise FileNotFoundError(f"{model_name!r} is not a local directory and no network is available; pass "
                                        "a directory holding config.json (+ weights) or set SNX_MODEL_DIR")
            geom = EncoderGeometry()
        self.model = ModernBertMaskedLMWeights(geom)
        self.config = SimpleNamespace(**self.model.hf_config_dict())
        self.relu = nn.ReLU()              # kept for interface parity; the ReLU is fused into the HIP tail
        self.model.init_weights()
        if weights_file is not None:
            self._load_inner_weights(weights_file)
        elif config is None:
            logger.warning("SPLADEModernBERT(%s): no pretrained weights available offline -- parameters are "
                           "RANDOMLY initialised (ModernBERT init recipe)", model_name)
        self._runtime = EncoderRuntime(geom, self.model.canonical_parameters())

    def _load_inner_weights(self, path: str):
        if path.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd = load_file(path)
        else:
            sd = torch.load(path, map_location="cpu", weights_only=True)
        sd = {k[len("model."):] if k.startswith("model.model.") or k.startswith("model.head.")
              or k.startswith("model.decoder.") else k: v for k, v in sd.items()}
        if "decoder.weight" not in sd and "model.embeddings.tok_embeddings.weight" in sd:
            sd["decoder.weight"] = sd["model.embeddings.tok_embeddings.weight"]
        self.model.load_state_dict(sd, strict=True)

    # ------------------------------------------------------------------ reference API
    @property
    def vocab_size(self) -> int:
        return self.config.vocab_size

    @property
    def hidden_size(self) -> int:
        return self.config.hidden_size

    @property
    def runtime(self) -> EncoderRuntime:
        return self._runtime

    def forward(self, input_ids: torch.Tensor, attention_mask: torch.Tensor,
                token_type_ids: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (sparse_repr [batch, vocab] fp32, token_weights [batch, seq_len] fp32);
        ``token_type_ids`` is accepted and ignored (ref:splade_modern.py:54,63)."""
        return self._runtime(input_ids, attention_mask)

    def forward_many(self, batches, lengths=None):
        """Extension (not in the reference): encode several (input_ids, attention_mask) batches of
        different sequence length -- e.g. the query, positive and negative batch of one training
        micro-step -- in ONE native pass.  Results are identical to separate ``forward`` calls
        (sequences never interact); larger GEMMs, a third of the launches.
        ``lengths`` (one CPU int tensor per batch, right-padded inputs) switches to UNPADDED execution:
        padded positions are never computed (rank 2 of SURVEY.md §8(f))."""
        return self._runtime.forward_many(batches, lengths)

    def encode(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        return self.forward(input_ids, attention_mask)[0]

    def get_top_k_tokens(self, sparse_repr: torch.Tensor, tokenizer, k: int = 50) -> Dict[str, float]:
        """Top-k (value > 0) tokens of one [vocab] vector; ties resolve lowest index first."""
        vals, idx = torch.sort(sparse_repr, descending=True, stable=True)
        k = min(k, sparse_repr.shape[0])
        out: Dict[str, float] = {}
        for val, i in zip(vals[:k].tolist(), idx[:k].tolist()):
            if val > 0:
                out[tokenizer.decode([i]).strip()] = val
        return out
