#!/bin/bash
# A/B: weight-gradient launches on the side stream (SNX_BWD_OVERLAP=1, default) or in line (0)
cd "$(dirname "$0")/.."
for o in 1 0 1 0; do
  SNX_BWD_OVERLAP=$o python bench.py --no-cpu-baseline --no-profile 2>/dev/null > gpurun_out/ab_overlap_$o.json
  python - <<PY
import json
d = json.loads(open("gpurun_out/ab_overlap_$o.json").read().strip().splitlines()[-1])
print("SNX_BWD_OVERLAP=$o", round(d["value"], 1), "triplets/s", round(d["ms_per_step"], 3), "ms")
PY
done
