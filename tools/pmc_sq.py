#!/usr/bin/env python3
"""Shader-engine counters per kernel from one rocprofv3 counter pass (MFMA utilisation, where waves spend their cycles).

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
              SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace \
              -d gpurun_out/pmc_sq -o s -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile
    python tools/pmc_sq.py gpurun_out/pmc_sq/.../s_results.db profiles/rNN_sq_counters.json

Units (MI355X_MICROARCH.md, cycle constants): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed
over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs' matrix pipes; GRBM_GUI_ACTIVE is summed over
the 8 XCDs (/ 8 = wall cycles of the dispatch: the clock held = that / duration).  Derived per kernel:
  mfma_util      = MFMA_BUSY / (wall_cycles x 1024 SIMDs)          share of the matrix pipes' time
  wait_any       = WAIT_ANY / WAVE_CYCLES                          waves parked in s_waitcnt / s_barrier
  wait_inst      = WAIT_INST_ANY / WAVE_CYCLES                     issue stalls (MFMA dependencies, pipes busy)
  active         = ACTIVE_INST_ANY / WAVE_CYCLES
Kernels are keyed by (short name, grid size) as in pmc_traffic.py."""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import kernel_source_hash, short  # noqa: E402


def main():
    db, out = sys.argv[1:3]
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, grid_size, counter_name, count(*), avg(value) from counters_collection "
                     "group by kernel_name, grid_size, counter_name").fetchall()
    res = {}
    for n, g, cn, k, v in rows:
        e = res.setdefault(f"{short(n)} grid={int(g)}", {"launches": int(k)})
        e[cn] = float(v)
    for k, e in res.items():
        wall = e.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        wc = e.get("SQ_WAVE_CYCLES", 0.0)
        if wall > 0:
            e["wall_cycles"] = wall
            e["mfma_util"] = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (wall * 1024.0)
        if wc > 0:
            e["wait_any"] = e.get("SQ_WAIT_ANY", 0.0) / wc
            e["wait_inst"] = e.get("SQ_WAIT_INST_ANY", 0.0) / wc
            e["active"] = e.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
            e["active_valu"] = e.get("SQ_ACTIVE_INST_VALU", 0.0) / wc
    order = sorted(res, key=lambda k: -res[k].get("wall_cycles", 0) * res[k]["launches"])
    with open(out, "w") as fh:
        json.dump({"kernel_source_sha256": kernel_source_hash(),
                   "workload": "python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile (BASELINE config 2), "
                               "kernels serialised by the counter collection",
                   "units": "see tools/pmc_sq.py", "kernels": {k: res[k] for k in order}}, fh, indent=1)
    print(f"{'kernel':58s} {'n':>5s} {'wall kcyc':>9s} {'mfma':>6s} {'wait':>6s} {'stall':>6s} {'active':>6s} {'valu':>6s}")
    for k in order[:30]:
        e = res[k]
        print(f"{k[:58]:58s} {e['launches']:5d} {e.get('wall_cycles', 0) / 1e3:9.1f} {e.get('mfma_util', 0):6.3f} "
              f"{e.get('wait_any', 0):6.3f} {e.get('wait_inst', 0):6.3f} {e.get('active', 0):6.3f} {e.get('active_valu', 0):6.3f}")


if __name__ == "__main__":
    main()
