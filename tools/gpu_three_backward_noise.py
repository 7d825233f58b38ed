"""Noise of the fused-pass vs three-forward parameter comparison (tests/test_gpu_dist.py): four seeds, run under
SNX_ATTN_BWD_ONEPASS=1 and =0; output kept in profiles/r04_three_backward_noise.txt."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opensearch-neural-pre-train_amd")); sys.path.insert(0, ROOT)
os.environ["SNX_DIST_FORCE"]="1"; os.environ["SNX_PACK"]="0"
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29577")
os.environ["RANK"]="0"; os.environ["WORLD_SIZE"]="1"; os.environ["LOCAL_RANK"]="0"
import torch.distributed as dist
dev=torch.device("cuda:0"); torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=dev)
from oracle import splade_oracle as O
from src.model.losses import SPLADELossV33
from src.train.config.v33 import V33Config
from src.train.core import ddp_trainer as T
from tests.test_gpu_model import _build_model, _small_cfg
cfg=_small_cfg()
params=O.perturb_params(O.init_params(cfg, seed=3), seed=4, scale=2.0, bias_mean=-0.1)
conf=V33Config(); conf.training.gradient_accumulation_steps=2; conf.training.learning_rate=1e-3
def run(fused, seed):
    gen=torch.Generator().manual_seed(seed)
    batches=[O.synth_batch(4,24,70,cfg,gen,k=1,ragged=True) for _ in range(4)]
    os.environ["SNX_FUSED_PASSES"]="1" if fused else "0"
    model=T.NativeDataParallel(_build_model(cfg, params, dev), n_buckets=3)
    loss_fn=SPLADELossV33(temperature=20.0, flops_warmup_steps=4).to(dev)
    opt=T.build_optimizer(model, conf); sch=T.build_scheduler(opt,0,4); step=0
    for i,b in enumerate(batches):
        last=(i+1)%2==0
        T.micro_step(model, loss_fn, b, step, dev, 2, last_of_window=last)
        if last:
            T.optimizer_step(model,opt,sch,conf); step+=1
    torch.cuda.synchronize()
    return {n:p.detach().clone() for n,p in model.module.named_parameters()}
for seed in (78, 79, 80, 81):
    a=run(True, seed); b=run(False, seed); c=run(True, seed)
    worst=max(((float((a[n]-b[n]).abs().mean()), n) for n in a))
    worst_same=max(((float((a[n]-c[n]).abs().mean()), n) for n in a))
    print("onepass", os.environ.get("SNX_ATTN_BWD_ONEPASS","1"), "seed", seed, "fused-vs-three worst mean %.3e (%s); fused-vs-fused rerun worst mean %.3e (%s)" % (worst+worst_same), flush=True)
dist.destroy_process_group()
