"""V33 configuration sections (API data of ref:src/train/config/v33.py:21-132).

The trainer builds every section as ``V33XConfig(**yaml_section)``, so the YAML keys ARE the field names and the
reference's defaults apply when a key is absent.  The sections are generated from one schema table: a row is
(field, type, default); list defaults are copied per instance."""
from dataclasses import field, make_dataclass
from typing import List

_SCHEMA = {
    # dropout is accepted and ignored by the model (every ModernBERT dropout is 0.0)
    "V33ModelConfig": [("name", str, "skt/A.X-Encoder-base"), ("dropout", float, 0.1)],
    # lambda_neg = 0 falls back to lambda_d inside the loss
    "V33LossConfig": [("lambda_q", float, 1e-2), ("lambda_d", float, 3e-3), ("temperature", float, 1.0),
                      ("flops_warmup_steps", int, 20000), ("lambda_kd", float, 0.0), ("kd_temperature", float, 1.0),
                      ("lambda_margin_mse", float, 0.0), ("lambda_initial_ratio", float, 0.1),
                      ("lambda_neg", float, 0.0)],
    # batch_size is per GPU
    "V33DataConfig": [("train_files", List[str], ["data/v29.0/train_*.jsonl"]),
                      ("val_files", List[str], ["data/v29.0/val.jsonl"]), ("batch_size", int, 64),
                      ("query_max_length", int, 64), ("doc_max_length", int, 256), ("num_workers", int, 4),
                      ("num_hard_negatives", int, 1)],
    "V33TrainingConfig": [("num_epochs", int, 25), ("learning_rate", float, 5e-5), ("weight_decay", float, 0.01),
                          ("warmup_ratio", float, 0.06), ("gradient_clip", float, 1.0),
                          ("gradient_accumulation_steps", int, 4), ("mixed_precision", str, "bf16"),
                          ("output_dir", str, "outputs/train_v33"), ("log_every_n_steps", int, 50),
                          ("save_every_n_epochs", int, 5), ("seed", int, 42)],
}


def _section(name: str):
    rows = []
    for fname, ftype, default in _SCHEMA[name]:
        if isinstance(default, list):
            rows.append((fname, ftype, field(default_factory=lambda d=default: list(d))))
        else:
            rows.append((fname, ftype, field(default=default)))
    cls = make_dataclass(name, rows)
    cls.__module__ = __name__
    return cls


V33ModelConfig = _section("V33ModelConfig")
V33LossConfig = _section("V33LossConfig")
V33DataConfig = _section("V33DataConfig")
V33TrainingConfig = _section("V33TrainingConfig")

_SECTIONS = (("model", V33ModelConfig), ("loss", V33LossConfig), ("data", V33DataConfig),
             ("training", V33TrainingConfig))


def _coerce_sections(self) -> None:
    """dict sections (straight from YAML) become their dataclasses"""
    for key, cls in _SECTIONS:
        value = getattr(self, key)
        if isinstance(value, dict):
            setattr(self, key, cls(**value))


V33Config = make_dataclass("V33Config", [(key, cls, field(default_factory=cls)) for key, cls in _SECTIONS],
                           namespace={"__post_init__": _coerce_sections})
V33Config.__module__ = __name__
