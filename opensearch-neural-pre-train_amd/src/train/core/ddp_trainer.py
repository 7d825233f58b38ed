"""DDP trainer core for SPLADE V33 (the module BASELINE.json's north star names).

The reference keeps these functions inline in ``src/train/cli/train_v33_ddp.py``; they are rehomed
here with IDENTICAL signatures (file:line of each original in the docstrings) so the reference
script's call sites work unchanged, plus the MI355X-native extras behind opt-in switches:

  * ``NativeDataParallel``: replaces torch DDP for this model -- gradients accumulate in one flat
    fp32 buffer inside the HIP backward and are all-reduced over RCCL/xGMI ONCE per optimizer step
    (DDP semantics: mean over ranks), not once per micro-batch;
  * cross-GPU in-batch negatives (``config.training.cross_gpu_negatives`` / env
    ``SNX_CROSS_GPU_NEGATIVES=1``): RCCL all-gather of the positive vectors, reduce-scatter backward;
  * no per-micro-step host syncs: losses are accumulated on the device.
"""
from __future__ import annotations

import json
import logging
import math
import os
from pathlib import Path
from typing import Dict, Optional

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.amp import autocast
from torch.nn.parallel import DistributedDataParallel as DDP
from torch.optim import AdamW
from torch.utils.data import DataLoader, DistributedSampler

from snx import dist as sdist
from src.model.losses import SPLADELossV33
from src.train.config.v33 import V33Config
from src.train.data.dataloader import TripletCollator

logger = logging.getLogger(__name__)


# ------------------------------------------------------------------------------------ process group
def setup_distributed() -> int:
    """ref:train_v33_ddp.py:105-110.  backend "nccl" is RCCL on ROCm; falls back to gloo only when
    no GPU is visible (CPU plumbing tests of the host logic)."""
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if torch.cuda.is_available():
        # SNX_DIST_BACKEND=gloo: rehearsal with more ranks than GPUs (ranks share devices, which RCCL refuses; snx.dist
        # then stages the gradient buckets through the host) -- the returned index is the DEVICE the rank uses
        backend = os.environ.get("SNX_DIST_BACKEND", "nccl")
        if backend != "nccl":
            local_rank %= max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend=backend)
    else:
        dist.init_process_group(backend="gloo")
    return local_rank


def cleanup_distributed() -> None:
    """ref:train_v33_ddp.py:113-115."""
    if dist.is_initialized():
        dist.destroy_process_group()


def is_main_process() -> bool:
    """ref:train_v33_ddp.py:118-120."""
    return (not dist.is_initialized()) or dist.get_rank() == 0


# ------------------------------------------------------------------------------------ data
def create_dataloader_ddp(dataset, tokenizer, config: V33Config, is_train: bool = True) -> DataLoader:
    """ref:train_v33_ddp.py:159-189: DistributedSampler + asymmetric TripletCollator."""
    sampler = DistributedSampler(dataset, num_replicas=sdist.world(), rank=sdist.rank(), shuffle=is_train)
    collator = TripletCollator(tokenizer=tokenizer, max_length=config.data.doc_max_length,
                               query_max_length=config.data.query_max_length,
                               doc_max_length=config.data.doc_max_length, use_in_batch_negatives=True)
    return DataLoader(dataset, batch_size=config.data.batch_size, sampler=sampler,
                      num_workers=config.data.num_workers, collate_fn=collator, pin_memory=True,
                      drop_last=is_train)


# ------------------------------------------------------------------------------------ model wrapper
class NativeDataParallel(nn.Module):
    """Data-parallel wrapper for SPLADEModernBERT on the snx backend (same ``.module`` attribute
    and call signature as torch DDP).  Parameters are broadcast from rank 0 at construction;
    gradient synchronisation is explicit (``sync_gradients``) and happens once per optimizer step."""

    def __init__(self, module: nn.Module, bucket_mb: int = 0, n_buckets: Optional[int] = None):
        super().__init__()
        self.module = module
        self.bucket_mb = bucket_mb
        rt = module.runtime
        if sdist.active():
            for p in module.parameters():
                dist.broadcast(p.data, src=0)
            rt.mark_weights_dirty()          # broadcast writes p.data without bumping p._version
        rt.enable_direct_grads(True)
        if n_buckets is None:
            n_buckets = int(os.environ.get("SNX_GRAD_BUCKETS", "4"))
        dev = next(module.parameters()).device
        rt.grad_sync = sdist.BucketedGradSync(dev, n_buckets) if (dev.type == "cuda" and n_buckets > 0) else None

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def arm_gradient_sync(self, on: bool = True) -> None:
        """Call before the LAST micro-step of an accumulation window: its backward then exchanges each
        finished gradient bucket over RCCL while the earlier layers' backward still runs."""
        gs = self.module.runtime.grad_sync
        if gs is not None:
            gs.arm(on)

    def sync_gradients(self) -> None:
        """Gradients averaged over ranks before the optimizer: wait for the overlapped exchange of the last
        backward, or (nothing was armed) all-reduce the whole flat buffer now."""
        rt = self.module.runtime
        if rt.grad_sync is not None and rt.grad_sync.wait(rt.flat_grad):
            return
        sdist.allreduce_flat_grads(rt.flat_grad, self.bucket_mb)

    def zero_grad(self, set_to_none: bool = False) -> None:   # keep the flat views alive
        self.module.runtime.zero_grads()


def unwrap(model: nn.Module) -> nn.Module:
    return model.module if hasattr(model, "module") else model


# ------------------------------------------------------------------------------------ optimizer
def build_optimizer(model: nn.Module, config: V33Config):
    """ref:train_v33_ddp.py:560-581.  The no-decay substrings match only ``decoder.bias`` under
    ModernBERT naming, so LayerNorm weights ARE decayed (quirk kept on purpose).  Under
    ``NativeDataParallel`` the same groups drive the fused clip+AdamW HIP kernel
    (``snx.optim.FusedAdamW``, same state-dict layout); SNX_FUSED_ADAMW=0 forces torch's AdamW."""
    no_decay = ["bias", "LayerNorm.weight", "layer_norm.weight"]
    named = list(model.named_parameters())
    groups = [
        {"params": [p for n, p in named if not any(nd in n for nd in no_decay)],
         "weight_decay": config.training.weight_decay},
        {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0},
    ]
    if isinstance(model, NativeDataParallel) and os.environ.get("SNX_FUSED_ADAMW", "1") != "0":
        from snx.optim import FusedAdamW
        return FusedAdamW(model.module.runtime, groups, lr=config.training.learning_rate)
    return AdamW(groups, lr=config.training.learning_rate)


def cosine_with_warmup_lambda(step: int, warmup: int, total: int, num_cycles: float = 0.5) -> float:
    """transformers.get_cosine_schedule_with_warmup's lr lambda (optimization.py:134-140)."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    progress = float(step - warmup) / float(max(1, total - warmup))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


def build_scheduler(optimizer: AdamW, num_warmup_steps: int, num_training_steps: int):
    """ref:train_v33_ddp.py:584-592 without the transformers dependency."""
    return torch.optim.lr_scheduler.LambdaLR(
        optimizer, lambda s: cosine_with_warmup_lambda(s, num_warmup_steps, num_training_steps))


# ------------------------------------------------------------------------------------ checkpoints
def save_checkpoint(model, optimizer: AdamW, scheduler, epoch: int, global_step: int, output_dir: str,
                    config: V33Config, best_metric: Optional[float] = None) -> str:
    """ref:train_v33_ddp.py:192-239: rank 0 writes checkpoint_epoch{E}_step{S}/{model.pt,
    training_state.pt, config.json} (same file names and dict keys)."""
    if not is_main_process():
        return ""
    ckpt_dir = Path(output_dir) / f"checkpoint_epoch{epoch}_step{global_step}"
    ckpt_dir.mkdir(parents=True, exist_ok=True)
    torch.save(unwrap(model).state_dict(), ckpt_dir / "model.pt")
    torch.save({"optimizer": optimizer.state_dict(), "scheduler": scheduler.state_dict(), "epoch": epoch,
                "global_step": global_step, "best_metric": best_metric}, ckpt_dir / "training_state.pt")
    with open(ckpt_dir / "config.json", "w") as f:
        json.dump({"model": config.model.__dict__, "loss": config.loss.__dict__, "data": config.data.__dict__,
                   "training": config.training.__dict__}, f, indent=2)
    logger.info(f"Saved checkpoint: {ckpt_dir}")
    return str(ckpt_dir)


def load_checkpoint(model: nn.Module, optimizer: Optional[AdamW], scheduler, checkpoint_path: str) -> Dict:
    """ref:train_v33_ddp.py:242-273 (model-only directories start a fine-tune)."""
    ckpt_dir = Path(checkpoint_path)
    state_dict = torch.load(ckpt_dir / "model.pt", map_location="cpu", weights_only=True)
    model.load_state_dict(state_dict)
    logger.info(f"Loaded model from {ckpt_dir / 'model.pt'}")
    ts_path = ckpt_dir / "training_state.pt"
    if ts_path.exists():
        ts = torch.load(ts_path, map_location="cpu", weights_only=True)
        if optimizer is not None:
            optimizer.load_state_dict(ts["optimizer"])
        if scheduler is not None:
            scheduler.load_state_dict(ts["scheduler"])
        return ts
    logger.info("No training_state.pt found, starting fresh (fine-tune)")
    return {"epoch": -1, "global_step": 0}


def find_latest_checkpoint(output_dir: str) -> Optional[str]:
    """ref:train_v33_ddp.py:276-286."""
    out = Path(output_dir)
    if not out.exists():
        return None
    cps = sorted(out.glob("checkpoint_epoch*_step*"), key=lambda p: int(p.name.split("_step")[1]))
    return str(cps[-1]) if cps else None


# ------------------------------------------------------------------------------------ the hot loop
def _cross_gpu_negatives(config) -> bool:
    return bool(getattr(config.training, "cross_gpu_negatives", False)) or \
        os.environ.get("SNX_CROSS_GPU_NEGATIVES", "0") == "1"


def _fuse_passes(model) -> bool:
    """The three encoder passes of a micro-step run as ONE native pass unless the model is wrapped in
    torch DDP (whose reducer must see ``DDP.forward``) or SNX_FUSED_PASSES=0."""
    return (not isinstance(model, DDP)) and hasattr(unwrap(model), "forward_many") and \
        os.environ.get("SNX_FUSED_PASSES", "1") != "0"


def _packed_lengths(batch: dict):
    """Sequence lengths for unpadded execution, taken from the collator's CPU masks (no device sync).
    None when the masks are already on the device, not right-padded, or fully dense, or SNX_PACK=0."""
    if os.environ.get("SNX_PACK", "1") == "0":
        return None
    out, total, valid = [], 0, 0
    for key in ("query", "positive", "negative"):
        m = batch[key + "_attention_mask"]
        if m.device.type != "cpu":
            return None
        ln = m.sum(dim=1)
        if int(ln.min()) < 1 or not torch.equal(m != 0, torch.arange(m.shape[1])[None, :] < ln[:, None]):
            return None
        out.append(ln)
        total += m.numel()
        valid += int(ln.sum())
    return out if valid < total else None


def micro_step(model, loss_fn: SPLADELossV33, batch: dict, global_step: int, device: torch.device,
               grad_accum: int, cross_gpu_negatives: bool = False, last_of_window: bool = False):
    """One micro-batch: three encoder passes, loss, backward (ref:train_v33_ddp.py:321-364).
    ``last_of_window``: this backward completes an accumulation window -> NativeDataParallel overlaps the
    gradient exchange with it."""
    nb = device.type == "cuda"
    if isinstance(model, NativeDataParallel):
        model.arm_gradient_sync(last_of_window)
    lengths = _packed_lengths(batch) if _fuse_passes(model) else None
    q_ids = batch["query_input_ids"].to(device, non_blocking=nb)
    q_mask = batch["query_attention_mask"].to(device, non_blocking=nb)
    p_ids = batch["positive_input_ids"].to(device, non_blocking=nb)
    p_mask = batch["positive_attention_mask"].to(device, non_blocking=nb)
    n_ids = batch["negative_input_ids"].to(device, non_blocking=nb)
    n_mask = batch["negative_attention_mask"].to(device, non_blocking=nb)
    num_negatives = batch.get("num_negatives", 1)
    t_pos, t_neg = batch.get("teacher_pos_scores"), batch.get("teacher_neg_scores")
    if t_pos is not None:
        t_pos = t_pos.to(device)
    if t_neg is not None:
        t_neg = t_neg.to(device)
    # ref:train_v33_ddp.py:337 autocasts to bf16 on cuda.  SNX_PRECISION=fp32 switches the block off: encoder AND loss
    # then compute in fp32, which is what the reference's trainer does where its cuda autocast is inactive (its CPU runs,
    # the source of goldens g2) -- a parity / debugging mode, not a training configuration
    with autocast(device_type=device.type, dtype=torch.bfloat16,
                  enabled=nb and os.environ.get("SNX_PRECISION", "auto") != "fp32"):
        # cross-GPU in-batch negatives (BASELINE config 4): the all-gather of the positive vectors goes to the EXCHANGE stream
        # as soon as they exist -- in the reference's call pattern right after the positive pass, so that it runs under the
        # negative pass; in the fused pass straight after the forward -- and only the loss waits for it
        xneg = cross_gpu_negatives and sdist.active()
        gather = sdist.all_gather_with_grad_async
        if os.environ.get("SNX_GATHER_INLINE", "0") == "1":  # A/B and tests: the collective inline on the compute stream
            gather = lambda x: sdist.PendingGather(sdist.all_gather_with_grad(x), None)   # noqa: E731
        gathered = None
        if _fuse_passes(model):
            (anchor_repr, _), (positive_repr, _), (negative_repr, _) = unwrap(model).forward_many(
                [(q_ids, q_mask), (p_ids, p_mask), (n_ids, n_mask)], lengths)
            if xneg:
                gathered = gather(positive_repr)
        else:
            anchor_repr, _ = model(q_ids, q_mask)
            positive_repr, _ = model(p_ids, p_mask)
            if xneg:
                gathered = gather(positive_repr)
            negative_repr, _ = model(n_ids, n_mask)
        if num_negatives > 1:
            negative_repr = negative_repr.view(anchor_repr.shape[0], num_negatives, -1)
        extra = {}
        if gathered is not None:
            positive_repr = gathered.wait()
            extra["label_offset"] = sdist.rank() * anchor_repr.shape[0]
        loss, loss_dict = loss_fn(anchor_repr=anchor_repr, positive_repr=positive_repr,
                                  negative_repr=negative_repr, global_step=global_step,
                                  teacher_pos_scores=t_pos, teacher_neg_scores=t_neg, **extra)
    (loss / grad_accum).backward()
    return loss.detach(), loss_dict


def optimizer_step(model, optimizer, scheduler, config: V33Config) -> None:
    """clip -> AdamW -> LR schedule -> zero grads (ref:train_v33_ddp.py:367-373)."""
    if isinstance(model, NativeDataParallel):
        model.sync_gradients()
    if hasattr(optimizer, "grad_norm"):                       # snx.optim.FusedAdamW: clip + AdamW in one pass
        optimizer.step(max_norm=config.training.gradient_clip)
    else:
        torch.nn.utils.clip_grad_norm_(model.parameters(), config.training.gradient_clip)
        optimizer.step()
    scheduler.step()
    if isinstance(model, NativeDataParallel):
        model.zero_grad()
    else:
        optimizer.zero_grad()


def train_epoch(model, dataloader: DataLoader, loss_fn: SPLADELossV33, optimizer: AdamW, scheduler,
                config: V33Config, epoch: int, global_step: int, device: torch.device, tb_logger=None,
                debug: bool = False) -> tuple:
    """Train one epoch (ref:train_v33_ddp.py:289-448); returns (avg_loss, global_step)."""
    model.train()
    if hasattr(dataloader, "sampler") and hasattr(dataloader.sampler, "set_epoch"):
        dataloader.sampler.set_epoch(epoch)
    accum = config.training.gradient_accumulation_steps
    xneg = _cross_gpu_negatives(config)
    total_loss = torch.zeros((), dtype=torch.float32, device=device)
    num_batches = 0
    progress = dataloader
    if is_main_process():
        try:
            from tqdm import tqdm
            progress = tqdm(dataloader, desc=f"Epoch {epoch}")
        except Exception:
            pass
    if isinstance(model, NativeDataParallel):
        model.zero_grad()
    else:
        optimizer.zero_grad()
    for batch_idx, batch in enumerate(progress):
        if debug and batch_idx >= 100:
            break
        loss, loss_dict = micro_step(model, loss_fn, batch, global_step, device, accum, xneg,
                                     last_of_window=(batch_idx + 1) % accum == 0)
        if (batch_idx + 1) % accum == 0:
            optimizer_step(model, optimizer, scheduler, config)
            global_step += 1
            if is_main_process() and global_step % config.training.log_every_n_steps == 0:
                _log_step(loss, loss_dict, loss_fn, scheduler, global_step, tb_logger)
        total_loss += loss
        num_batches += 1
    avg_loss = float(total_loss.item()) / max(num_batches, 1)
    return avg_loss, global_step


def _log_step(loss, loss_dict, loss_fn, scheduler, global_step, tb_logger) -> None:
    """Console + TensorBoard scalars with the reference's tags (ref:train_v33_ddp.py:377-442)."""
    lr = scheduler.get_last_lr()[0]
    nz_q, nz_d = loss_fn.get_avg_nonzero()
    extra = ""
    if float(loss_dict.get("kd", 0)) > 0:
        extra += f" | kd={float(loss_dict['kd']):.4f}"
    if float(loss_dict.get("margin_mse", 0)) > 0:
        extra += f" | mmse={float(loss_dict['margin_mse']):.4f}"
    logger.info(f"Step {global_step} | loss={float(loss):.4f} | infonce={float(loss_dict['infonce']):.4f} | "
                f"flops_q={float(loss_dict['flops_q']):.2f} | flops_d={float(loss_dict['flops_d']):.2f} | "
                f"flops_neg={float(loss_dict.get('flops_neg', 0)):.2f} | lam_q={loss_dict['lambda_q']:.6f} | "
                f"lam_d={loss_dict['lambda_d']:.6f} | nz_q={nz_q:.0f} | nz_d={nz_d:.0f} | lr={lr:.2e}{extra}")
    if tb_logger:
        tb_logger.log_scalar("train/loss", float(loss), global_step)
        tb_logger.log_scalar("train/lr", lr, global_step)
        for tag in ("infonce", "flops_q", "flops_d", "lambda_q", "lambda_d"):
            tb_logger.log_scalar(f"train/{tag}", float(loss_dict[tag]), global_step)
        if float(loss_dict.get("flops_neg", 0)) > 0:
            tb_logger.log_scalar("train/flops_neg", float(loss_dict["flops_neg"]), global_step)
        tb_logger.log_scalar("train/nonzero_q", nz_q, global_step)
        tb_logger.log_scalar("train/nonzero_d", nz_d, global_step)
        for tag in ("kd", "margin_mse"):
            if float(loss_dict.get(tag, 0)) > 0:
                tb_logger.log_scalar(f"train/{tag}", float(loss_dict[tag]), global_step)
