#!/usr/bin/env python3
"""HBM-side traffic per kernel from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 4 --warmup 2
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 4 --warmup 2
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_results.db gpurun_out/pmc_write/w_results.db profiles/rNN_pmc_traffic.json

Corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is exact for 16-B stores and float
atomics.  Infinity-Cache hits are counted (these are fabric bytes, an upper bound on DRAM bytes).
Kernels are keyed by (short name, grid size) so that the GEMM shapes stay apart."""
import glob
import hashlib
import json
import os
import re
import sqlite3
import sys


def kernel_source_hash() -> str:
    """Same definition as bench.py: sha256 over csrc/*.hip and *.h."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "opensearch-neural-pre-train_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)                      # drop the argument list
    m = re.match(r"_Z\d+([A-Za-z_0-9]+?)(ILi[\dELi]+E)?(v|P|i)", name)
    return (m.group(1) + (m.group(2) or "")) if m else name.replace("void ", "").strip()


def load(db: str, counter: str):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, grid_size, count(*), avg(value) from counters_collection "
                     "where counter_name = ? group by kernel_name, grid_size", (counter,)).fetchall()
    return {(short(n), int(g)): (int(k), float(v)) for n, g, k, v in rows}


def main():
    fetch_db, write_db, out = sys.argv[1:4]
    f = load(fetch_db, "FETCH_SIZE")
    w = load(write_db, "WRITE_SIZE")
    res = {}
    for key in sorted(set(f) | set(w), key=lambda k: -(f.get(k, (0, 0))[1] * f.get(k, (0, 0))[0])):
        nf, kf = f.get(key, (0, 0.0))
        nw, kw = w.get(key, (0, 0.0))
        res[f"{key[0]} grid={key[1]}"] = {
            "launches": max(nf, nw), "fetch_kib_raw": kf, "write_kib": kw,
            "fabric_bytes_per_launch": kf * 1024 * 2 + kw * 1024,
        }
    with open(out, "w") as fh:
        json.dump({"kernel_source_sha256": kernel_source_hash(),
                   "workload": "python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline (BASELINE config 2)",
                   "note": "fabric (L2 memory-side) bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes; "
                           "Infinity-Cache hits included", "kernels": res}, fh, indent=1)
    for k, v in list(res.items())[:24]:
        print(f"{v['fabric_bytes_per_launch'] / 1e6:10.1f} MB/launch  n={v['launches']:5d}  {k}")


if __name__ == "__main__":
    main()
