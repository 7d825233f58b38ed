#!/bin/bash
# Round profile set (run on the GPU box through gpurun; summaries are copied into profiles/ afterwards):
#   bench JSON (config 2 and config 5), rocprofv3 kernel stats of the same command, PMC fabric traffic (two passes).
R=${1:-r06}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/prof_$R
rm -rf $O; mkdir -p $O
echo "[prof] kernel stats"; SNX_BWD_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-profile --no-item-sync-leg --no-sparse-regime-leg --no-config-legs > $O/stats.log 2>&1 || exit 1
echo "[prof] kernel stats overlap"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_overlap -o b -- python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-profile --no-item-sync-leg --no-sparse-regime-leg --no-config-legs > $O/stats_overlap.log 2>&1 || exit 1
echo "[prof] kernel stats cfg5"; SNX_BWD_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg5 -o b -- python3 bench.py --d-len 512 --negatives 4 --margin-mse 0.5 --steps 8 --warmup 4 --no-cpu-baseline --no-profile --no-item-sync-leg --no-sparse-regime-leg > $O/stats_cfg5.log 2>&1 || exit 1
echo "[prof] pmc fetch"; rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o f -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-item-sync-leg --no-sparse-regime-leg --no-config-legs > $O/pmc_fetch.log 2>&1 || exit 1
echo "[prof] pmc write"; rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o w -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-item-sync-leg --no-sparse-regime-leg --no-config-legs > $O/pmc_write.log 2>&1 || exit 1
F=$(find $O/pmc_fetch -name "f_results.db" | head -1); W=$(find $O/pmc_write -name "w_results.db" | head -1)
python3 tools/pmc_traffic.py "$F" "$W" $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
# the bench lines come AFTER the counter passes: bench.py reports `traffic` from profiles/${R}_pmc_traffic.json only when
# that file was measured on the kernel sources it is running (sha256 inside)
cp $O/pmc_traffic.json profiles/${R}_pmc_traffic.json
echo "[prof] bench n1"; python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
echo "[prof] bench cfg5"; python3 bench.py --d-len 512 --negatives 4 --margin-mse 0.5 --steps 16 --warmup 4 --no-cpu-baseline --no-sparse-regime-leg > $O/bench_cfg5_n1.json 2> $O/bench_cfg5.err || exit 1
# shader-engine counters (MFMA utilisation, where the waves' cycles go): one pass, the program directly behind `--`
echo "[prof] pmc sq"; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_sq -o s -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-item-sync-leg --no-sparse-regime-leg --no-config-legs > $O/pmc_sq.log 2>&1 || exit 1
S=$(find $O/pmc_sq -name "s_results.db" | head -1)
python3 tools/pmc_sq.py "$S" $O/sq_counters.json > $O/sq_counters.txt 2>&1
find $O -name "*_kernel_stats.csv" | head; ls $O
# the sqlite traces are large: keep only the summaries
find $O -name "*.db" -delete
