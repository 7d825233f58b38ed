from .v33 import V33Config, V33DataConfig, V33LossConfig, V33ModelConfig, V33TrainingConfig  # noqa: F401
