#!/bin/bash
# A/B of one environment switch on one box: bash tools/gpu_ab_env.sh SNX_BWD_OVERLAP "1 0 1 0"  -> triplets/s and ms per micro-step
VAR=$1; shift
for v in $1; do
  env $VAR=$v python bench.py --no-cpu-baseline --no-item-sync-leg --no-profile 2>/dev/null > /tmp/ab_$$.json
  python - "$VAR" "$v" /tmp/ab_$$.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], round(d["value"], 1), "triplets/s", round(d["ms_per_step"], 3), "ms", flush=True)
PY
done
