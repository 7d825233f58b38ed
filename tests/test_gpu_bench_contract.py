"""bench.py's one-line JSON contract (the driver parses it): a short real run on the GPU."""
import json
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
