"""V33 configuration dataclasses (API data of ref:src/train/config/v33.py:21-132).

Field names and defaults are the reference's: the trainer builds each section as
``V33XConfig(**yaml_section)`` so YAML keys must equal field names."""
from dataclasses import dataclass, field
from typing import List


@dataclass
class V33ModelConfig:
    name: str = "skt/A.X-Encoder-base"
    dropout: float = 0.1          # accepted and ignored by the model (all dropouts are 0.0)


@dataclass
class V33LossConfig:
    lambda_q: float = 1e-2
    lambda_d: float = 3e-3
    temperature: float = 1.0
    flops_warmup_steps: int = 20000
    lambda_kd: float = 0.0
    kd_temperature: float = 1.0
    lambda_margin_mse: float = 0.0
    lambda_initial_ratio: float = 0.1
    lambda_neg: float = 0.0       # 0 -> falls back to lambda_d inside the loss


@dataclass
class V33DataConfig:
    train_files: List[str] = field(default_factory=lambda: ["data/v29.0/train_*.jsonl"])
    val_files: List[str] = field(default_factory=lambda: ["data/v29.0/val.jsonl"])
    batch_size: int = 64          # per GPU
    query_max_length: int = 64
    doc_max_length: int = 256
    num_workers: int = 4
    num_hard_negatives: int = 1


@dataclass
class V33TrainingConfig:
    num_epochs: int = 25
    learning_rate: float = 5e-5
    weight_decay: float = 0.01
    warmup_ratio: float = 0.06
    gradient_clip: float = 1.0
    gradient_accumulation_steps: int = 4
    mixed_precision: str = "bf16"
    output_dir: str = "outputs/train_v33"
    log_every_n_steps: int = 50
    save_every_n_epochs: int = 5
    seed: int = 42


@dataclass
class V33Config:
    model: V33ModelConfig = field(default_factory=V33ModelConfig)
    loss: V33LossConfig = field(default_factory=V33LossConfig)
    data: V33DataConfig = field(default_factory=V33DataConfig)
    training: V33TrainingConfig = field(default_factory=V33TrainingConfig)

    def __post_init__(self) -> None:
        for name, cls in (("model", V33ModelConfig), ("loss", V33LossConfig),
                          ("data", V33DataConfig), ("training", V33TrainingConfig)):
            v = getattr(self, name)
            if isinstance(v, dict):
                setattr(self, name, cls(**v))
