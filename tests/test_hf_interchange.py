"""Row f3: the HF export is interchangeable with stock `transformers` in both
directions.  CPU only; skipped where `transformers` (with ModernBERT) is not importable.

  * ref:scripts/export_v33_hf.py:28-32 calls ``model.model.save_pretrained(str(output_dir),
    safe_serialization=True)`` -- the exact call is made here;
  * the exported directory loads with ``ModernBertForMaskedLM.from_pretrained`` and its logits equal the
    oracle's ``encoder_logits`` on the same ids (so key names, tying and config fields are right);
  * a directory written by transformers' own ``save_pretrained`` loads into ``SPLADEModernBERT(model_name=dir)``.
"""
import json
import logging

import pytest
import torch

from oracle import splade_oracle as O

transformers = pytest.importorskip("transformers")
try:
    from transformers import ModernBertForMaskedLM
except Exception:                                            # pragma: no cover
    pytest.skip("transformers without ModernBERT", allow_module_level=True)

GEOM = dict(vocab_size=300, hidden_size=256, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4,
            local_attention=8, pad_token_id=299)


def _ours():
    from src.model.splade_modern import SPLADEModernBERT
    logging.getLogger("src.model.splade_modern").setLevel(logging.ERROR)
    torch.manual_seed(5)
    m = SPLADEModernBERT(config=GEOM)
    with torch.no_grad():                                    # non-trivial norms / bias
        for n, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    return m


def _oracle_cfg():
    return O.EncoderConfig(**GEOM)


def test_export_loads_in_stock_transformers_and_matches_oracle(tmp_path):
    m = _ours()
    m.model.save_pretrained(str(tmp_path), safe_serialization=True)       # the reference's exact call
    hf = ModernBertForMaskedLM.from_pretrained(str(tmp_path))
    hf.eval()
    sd = hf.state_dict()
    ours = m.model.state_dict()
    assert set(sd.keys()) == set(ours.keys()) and len(sd) == len(ours)
    assert sd["decoder.weight"].data_ptr() == sd["model.embeddings.tok_embeddings.weight"].data_ptr()
    for k in ours:
        assert torch.equal(sd[k], ours[k].detach()), k
    params = {"model." + k: v.detach().clone() for k, v in ours.items() if k != "decoder.weight"}
    cfg = _oracle_cfg()
    ids, mask = O.synth_ids(3, 24, cfg, torch.Generator().manual_seed(2), ragged=True)
    with torch.no_grad():
        ref = O.encoder_logits(params, cfg, ids, mask, "fp32")
        got = hf(input_ids=ids, attention_mask=mask).logits
    valid = mask.bool()
    assert torch.allclose(got[valid], ref[valid], atol=2e-5, rtol=1e-5), float((got - ref)[valid].abs().max())
    c = json.load(open(tmp_path / "config.json"))
    assert c["model_type"] == "modernbert" and c["tie_word_embeddings"] is True


def test_full_geometry_export_has_the_138_keys(tmp_path):
    """Checkpoint contract of SURVEY 2.2 through the export: 138 state-dict keys incl. the tied alias."""
    from src.model.splade_modern import SPLADEModernBERT
    logging.getLogger("src.model.splade_modern").setLevel(logging.ERROR)
    m = SPLADEModernBERT()
    assert len(m.model.state_dict()) == 138
    hf_keys = set(ModernBertForMaskedLM(transformers.ModernBertConfig(**{
        k: v for k, v in m.model.hf_config_dict().items() if k not in ("architectures", "model_type", "dtype")})).state_dict())
    assert hf_keys == set(m.model.state_dict().keys())


def test_unsafe_serialization_writes_pytorch_bin(tmp_path):
    m = _ours()
    m.model.save_pretrained(str(tmp_path), safe_serialization=False, max_shard_size="5GB")
    assert (tmp_path / "pytorch_model.bin").exists() and not (tmp_path / "model.safetensors").exists()
    from src.model.splade_modern import SPLADEModernBERT
    m2 = SPLADEModernBERT(model_name=str(tmp_path))
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_transformers_saved_directory_loads_into_splade_modernbert(tmp_path):
    """Reverse direction: a directory written by transformers' own save_pretrained (e.g. huggingface/v33 with
    weights) -> SPLADEModernBERT(model_name=dir)."""
    from src.model.splade_modern import SPLADEModernBERT
    m = _ours()
    cfgd = {k: v for k, v in m.model.hf_config_dict().items() if k not in ("architectures", "model_type", "dtype")}
    torch.manual_seed(11)
    hf = ModernBertForMaskedLM(transformers.ModernBertConfig(**cfgd))
    hf.save_pretrained(str(tmp_path), safe_serialization=True)
    m2 = SPLADEModernBERT(model_name=str(tmp_path))
    sd_hf = hf.state_dict()
    for k, v in m2.model.state_dict().items():
        assert torch.equal(v, sd_hf[k]), k
    assert m2.model.decoder.weight is m2.model.model.embeddings.tok_embeddings.weight
    assert m2.vocab_size == GEOM["vocab_size"] and m2.config.local_attention == GEOM["local_attention"]
