"""bench.py's one-line JSON contract (the driver parses it): a short real run on the GPU."""
import json
import math
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "4",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int),
                     ("warmup", int), ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str),
                     ("dtype", str), ("data", str), ("config", dict), ("roofline", dict)):
        assert isinstance(d[key], typ), key
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 4
    assert d["unit"] == "triplets/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] == pytest.approx(64 * 1000.0 / d["ms_per_step"], rel=1e-6)       # 64 triplets per micro-step
    roof = d["roofline"]
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["peak"] == 2500.0
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"]) and 0.05 < roof["frac"] < 1.0
    assert roof["traffic"] is None or roof["traffic"] > 1e6
    assert 100.0 < d["value"] < 1e5
    assert math.isfinite(d["config"]["final_loss"])
    assert "no per-micro-step loss.item()" in d["config"]["workload"]
    ex = d["extra"]                                       # the reference's literal loop: three forwards + loss.item()
    assert 100.0 < ex["value_with_item_sync"] <= d["value"] * 1.05 and ex["steps"] >= 4
    sp = ex["sparse_regime"]                              # the trained model's output sparsity: tens of active dimensions
    assert 20.0 < sp["active_dims_doc"] < 120.0 and sp["active_dims_query"] < sp["active_dims_doc"] and sp["ms_per_step"] > 0
    c5 = ex["config5"]                                    # BASELINE config 5 beside the headline
    assert "error" not in c5 and 10.0 < c5["value"] < d["value"] and 0.05 < c5["mfma_roofline_frac_step"] < 1.0
    assert "config4" not in ex                            # N > 1 only
    assert roof["traffic_provenance"]["status"].split(":")[0] in ("current", "stale", "absent")
    assert (roof["traffic"] is None) == (roof["traffic_provenance"]["status"] != "current")


@pytest.mark.gpu
def test_bench_under_torchrun_one_rank_over_rccl():
    """`--gpus N` launch path with N = 1: torch.distributed.run, nccl (= RCCL) process group, rank-0 broadcast,
    bucketed gradient all-reduce overlapped with the last backward of each window, and BASELINE config 4's
    all-gather / reduce-scatter of the positive vectors (SNX_DIST_FORCE=1 issues them with one rank)."""
    env = dict(os.environ, SNX_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "8", "--warmup", "4", "--no-cpu-baseline", "--no-profile",
                        "--cross-gpu-negatives"],     # config 4: all-gather of the positives + reduce-scatter backward
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and math.isfinite(d["config"]["final_loss"]) and 100.0 < d["value"] < 1e5
    comm = d["comm"]                                      # what RCCL itself saw
    assert comm["world_size"] == 1 and "nccl" in comm["backend"] and comm["ranks_counted"] == 1.0
    assert comm["rank_checksum"] == comm["rank_checksum_expected"] == 0.0 and comm["rccl_version"][0].isdigit()


@pytest.mark.gpu
def test_bench_launch_line_with_two_ranks_rehearsed_on_one_gpu():
    """The driver's N > 1 command (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) with N = 2
    on the one GPU a test box has: SNX_BENCH_BACKEND=gloo lets the two ranks share the device (RCCL refuses that) and
    moves the gradient buckets through host copies, so the multi-rank control flow of bench.py itself runs -- per-rank
    seeds, rank-0 broadcast at wrap, barriers around the timed region, MAX over ranks, the extra profiled steps that
    keep the collectives of rank > 0 matched with rank 0's, exactly one JSON line from rank 0.  The value it prints is
    a rehearsal number (two ranks on one device), only its shape is checked."""
    env = dict(os.environ, SNX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("SNX_DIST_FORCE",):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29537", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "4", "--warmup", "4", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2"
    assert math.isfinite(d["config"]["final_loss"]) and d["value"] > 0
    assert "cpu_baseline" not in d                       # rank 0 at N = 1 only
    comm = d["comm"]                                      # the communicator's own account: two ranks, checksum 0 + 1
    assert comm["world_size"] == 2 and comm["ranks_counted"] == 2.0 and comm["rank_checksum"] == 1.0 == comm["rank_checksum_expected"]
    assert len(comm["devices"]) == 2 and comm["backend"] == "gloo"
    c4 = d["extra"]["config4"]                           # BASELINE config 4's leg ran on both ranks (gathered positives: 128)
    assert "error" not in c4 and c4["value"] > 0 and "128 positives" in c4["workload"]
    assert "config5" not in d["extra"]                   # (nccl runs only: two rehearsal ranks share ONE device's memory)


@pytest.mark.gpu
def test_bench_gpus_n_without_a_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no RANK in the environment: the process must become the launcher (fresh workers
    through torch.distributed.run, the parent never touches the GPU) -- never print an n_gpus 1 line.  On the one GPU of a
    test box the two ranks share the device (SNX_BENCH_BACKEND=gloo); with the default backend the same invocation is
    refused before anything starts, because one process per GPU over RCCL needs two GPUs."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SNX_DIST_FORCE")}
    env.update(SNX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "4",
                        "--no-cpu-baseline", "--no-profile", "--no-item-sync-leg"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2"
    import torch
    if torch.cuda.device_count() < 2:
        env.pop("SNX_BENCH_BACKEND")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4"],
                           capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode != 0 and '"n_gpus"' not in r.stdout and "GPU(s) visible" in r.stderr
