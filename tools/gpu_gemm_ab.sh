#!/bin/bash
# A/B of NT GEMM knobs at the bench's shapes (one process per setting: the knobs are read once)
cd "$(dirname "$0")/.."
for cfg in "SNX_GEMM_MID=18" "SNX_GEMM_MID=0" "SNX_GEMM_MID=2" "SNX_GEMM_MID=16"; do
  echo "=== $cfg"
  env $cfg python tools/gpu_epibench.py 2>/dev/null | grep -E "resid|geglu_bwd"
done
