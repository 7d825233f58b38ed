"""V33 multi-GPU training entry point (counterpart of ref:src/train/cli/train_v33_ddp.py:451-732).

    torchrun --nproc_per_node=8 -m src.train.cli.train_v33_ddp --config configs/train_v33.yaml

Same flags as the reference script; the work is delegated to ``src.train.core.ddp_trainer``.
Extra opt-in flags select the MI355X-native data-parallel path:
    --native-dp              one RCCL all-reduce of the flat gradient buffer per optimizer step
                             (default: torch DDP, exactly like the reference)
    --cross-gpu-negatives    all-gather positive vectors for cross-GPU in-batch negatives
Offline: ``model.name`` may be a local directory (config.json [+ weights, tokenizer]); with the hub
name the A.X-Encoder-base geometry is random-initialised.  ``data.train_files: ["synthetic:N"]``
together with ``--tokenizer hash:50000`` runs on synthetic text triplets.
"""
from __future__ import annotations

import argparse
import logging
import os
import time
from pathlib import Path

import torch
import torch.distributed as dist
import yaml

from src.model.losses import SPLADELossV33
from src.model.splade_modern import SPLADEModernBERT
from src.train.config.v33 import V33Config, V33DataConfig, V33LossConfig, V33ModelConfig, V33TrainingConfig
from src.train.core import ddp_trainer as T
from src.train.data import load_training_data
from src.train.data.collator import create_tokenizer
from src.train.utils import TensorBoardLogger, setup_logging

logger = logging.getLogger(__name__)
_OVERRIDES = (("epochs", "training", "num_epochs"), ("batch_size", "data", "batch_size"),
              ("lr", "training", "learning_rate"), ("output_dir", "training", "output_dir"),
              ("lambda_q", "loss", "lambda_q"), ("lambda_d", "loss", "lambda_d"),
              ("grad_accum", "training", "gradient_accumulation_steps"), ("seed", "training", "seed"))


def parse_args() -> argparse.Namespace:
    ap = argparse.ArgumentParser(description="V33 DDP training: SPLADE-max with ModernBERT (MI355X-native)",
                                 formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    ap.add_argument("--config", type=str, default="configs/train_v33.yaml")
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--batch-size", type=int, default=None)
    ap.add_argument("--lr", type=float, default=None)
    ap.add_argument("--output-dir", type=str, default=None)
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--checkpoint", type=str, default=None)
    ap.add_argument("--lambda-q", type=float, default=None)
    ap.add_argument("--lambda-d", type=float, default=None)
    ap.add_argument("--grad-accum", type=int, default=None)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--debug", action="store_true")
    ap.add_argument("--native-dp", action="store_true")
    ap.add_argument("--cross-gpu-negatives", action="store_true")
    ap.add_argument("--tokenizer", type=str, default=None, help="tokenizer dir or hash:<vocab> (default: model.name)")
    return ap.parse_args()


def load_config(args: argparse.Namespace) -> V33Config:
    """YAML sections -> dataclasses, then CLI overrides (ref:train_v33_ddp.py:123-156)."""
    path = Path(args.config)
    if path.exists():
        raw = yaml.safe_load(path.read_text()) or {}
        config = V33Config(model=V33ModelConfig(**raw.get("model", {})), loss=V33LossConfig(**raw.get("loss", {})),
                           data=V33DataConfig(**raw.get("data", {})), training=V33TrainingConfig(**raw.get("training", {})))
    else:
        config = V33Config()
    for arg, section, field in _OVERRIDES:
        v = getattr(args, arg)
        if v is not None:
            setattr(getattr(config, section), field, v)
    return config


def main() -> None:
    args = parse_args()
    local_rank = T.setup_distributed()
    device = torch.device(f"cuda:{local_rank}")
    world = dist.get_world_size()
    config = load_config(args)
    out = Path(config.training.output_dir)
    if T.is_main_process():
        out.mkdir(parents=True, exist_ok=True)
    dist.barrier()
    if T.is_main_process():
        setup_logging(output_dir=str(out), log_file="training.log")
        eff = config.data.batch_size * config.training.gradient_accumulation_steps * world
        logger.info(f"V33 training | model={config.model.name} | GPUs={world} | per-GPU batch={config.data.batch_size} "
                    f"| accum={config.training.gradient_accumulation_steps} | effective batch={eff} | "
                    f"lr={config.training.learning_rate} | epochs={config.training.num_epochs} | "
                    f"lambda_q={config.loss.lambda_q} lambda_d={config.loss.lambda_d} | out={out}")
    torch.manual_seed(config.training.seed + dist.get_rank())
    tokenizer = create_tokenizer(args.tokenizer or config.model.name)
    train_ds = load_training_data(config.data.train_files)
    train_dl = T.create_dataloader_ddp(train_ds, tokenizer, config, is_train=True)
    model = SPLADEModernBERT(model_name=config.model.name, dropout=config.model.dropout).to(device)
    if T.is_main_process():
        logger.info(f"parameters: {sum(p.numel() for p in model.parameters()):,} | vocab {model.vocab_size} | "
                    f"train samples {len(train_ds):,}")
    if args.native_dp:
        model = T.NativeDataParallel(model)
    else:
        model = T.DDP(model, device_ids=[local_rank], broadcast_buffers=False, find_unused_parameters=False)
    if args.cross_gpu_negatives:
        os.environ["SNX_CROSS_GPU_NEGATIVES"] = "1"
    loss_fn = SPLADELossV33(lambda_q=config.loss.lambda_q, lambda_d=config.loss.lambda_d,
                            temperature=config.loss.temperature, flops_warmup_steps=config.loss.flops_warmup_steps,
                            lambda_kd=config.loss.lambda_kd, kd_temperature=config.loss.kd_temperature,
                            lambda_initial_ratio=config.loss.lambda_initial_ratio,
                            lambda_margin_mse=config.loss.lambda_margin_mse,
                            lambda_neg=getattr(config.loss, "lambda_neg", 0.0)).to(device)
    optimizer = T.build_optimizer(model, config)
    steps_per_epoch = len(train_dl) // config.training.gradient_accumulation_steps
    total_steps = steps_per_epoch * config.training.num_epochs
    scheduler = T.build_scheduler(optimizer, int(total_steps * config.training.warmup_ratio), total_steps)
    start_epoch, global_step, best_metric = 1, 0, None
    if args.resume or args.checkpoint:
        ckpt = args.checkpoint or T.find_latest_checkpoint(config.training.output_dir)
        if ckpt:
            st = T.load_checkpoint(T.unwrap(model), optimizer, scheduler, ckpt)
            start_epoch, global_step, best_metric = st["epoch"] + 1, st["global_step"], st.get("best_metric")
    tb = TensorBoardLogger(log_dir=str(out / "tensorboard"), experiment_name="v33_modernbert") if T.is_main_process() else None
    t_start = time.time()
    for epoch in range(start_epoch, config.training.num_epochs + 1):
        t0 = time.time()
        avg_loss, global_step = T.train_epoch(model=model, dataloader=train_dl, loss_fn=loss_fn, optimizer=optimizer,
                                              scheduler=scheduler, config=config, epoch=epoch, global_step=global_step,
                                              device=device, tb_logger=tb, debug=args.debug)
        if T.is_main_process():
            nz_q, nz_d = loss_fn.get_avg_nonzero()
            logger.info(f"Epoch {epoch}/{config.training.num_epochs} | avg_loss={avg_loss:.4f} | nz_q={nz_q:.0f} | "
                        f"nz_d={nz_d:.0f} | time={(time.time() - t0) / 60:.1f}min")
        if epoch % config.training.save_every_n_epochs == 0 or epoch == config.training.num_epochs:
            dist.barrier()
            T.save_checkpoint(model=model, optimizer=optimizer, scheduler=scheduler, epoch=epoch,
                              global_step=global_step, output_dir=str(out), config=config, best_metric=best_metric)
        if args.debug:
            break
    if T.is_main_process():
        logger.info(f"Training complete in {(time.time() - t_start) / 3600:.2f}h")
        final = out / "final_model"
        final.mkdir(parents=True, exist_ok=True)
        torch.save(T.unwrap(model).state_dict(), final / "model.pt")
        tokenizer.save_pretrained(str(final))
    T.cleanup_distributed()


if __name__ == "__main__":
    main()
