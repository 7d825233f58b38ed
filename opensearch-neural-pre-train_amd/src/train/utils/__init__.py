"""Logging helpers the trainer imports (``setup_logging``, ``TensorBoardLogger``;
ref:src/train/utils/logging.py:69-121,140-188).  Unlike the reference's, ``TensorBoardLogger`` does
not raise when tensorboard is missing: it falls back to a JSONL scalar log in the same directory."""
from __future__ import annotations

import json
import logging
import os
import sys
from typing import Optional


def setup_logging(output_dir: Optional[str] = None, log_file: str = "training.log", level: int = logging.INFO,
                  **_ignored) -> logging.Logger:
    root = logging.getLogger()
    root.setLevel(level)
    fmt = logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s")
    if not any(isinstance(h, logging.StreamHandler) and h.stream is sys.stdout for h in root.handlers):
        sh = logging.StreamHandler(sys.stdout)
        sh.setFormatter(fmt)
        root.addHandler(sh)
    if output_dir:
        os.makedirs(output_dir, exist_ok=True)
        fh = logging.FileHandler(os.path.join(output_dir, log_file))
        fh.setFormatter(fmt)
        root.addHandler(fh)
    return root


class TensorBoardLogger:
    def __init__(self, log_dir: str, experiment_name: str = "run", **_ignored):
        self.dir = os.path.join(log_dir, experiment_name)
        os.makedirs(self.dir, exist_ok=True)
        self._tb = None
        try:
            from torch.utils.tensorboard import SummaryWriter
            self._tb = SummaryWriter(self.dir)
        except Exception:
            self._jsonl = open(os.path.join(self.dir, "scalars.jsonl"), "a")

    def log_scalar(self, tag: str, value, step: int) -> None:
        v = float(value)
        if self._tb is not None:
            self._tb.add_scalar(tag, v, step)
        else:
            self._jsonl.write(json.dumps({"tag": tag, "value": v, "step": int(step)}) + "\n")
            self._jsonl.flush()

    def close(self) -> None:
        if self._tb is not None:
            self._tb.close()
        else:
            self._jsonl.close()
