#!/bin/bash
# copy the summaries of the last `bash tools/gpu_profile_round.sh rNN` (merged back under gpurun_out/) into profiles/
R=${1:-r04}
O=gpurun_out/prof_$R
cp $O/bench_n1.json profiles/${R}_bench_n1.json
cp $O/bench_cfg5_n1.json profiles/${R}_bench_cfg5_n1.json
cp $O/stats/b_kernel_stats.csv profiles/${R}_bench_n1_kernel_stats.csv
cp $O/stats_overlap/b_kernel_stats.csv profiles/${R}_bench_n1_kernel_stats_overlap.csv
cp $O/pmc_traffic.json profiles/${R}_pmc_traffic.json
cp $O/sq_counters.json profiles/${R}_sq_counters.json
cp $O/sq_counters.txt profiles/${R}_sq_counters.txt
cp gpurun_out/parity_report.jsonl profiles/${R}_parity_report.jsonl
python3 - <<PY
import json
d = json.loads(open("profiles/${R}_bench_n1.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_provenance"]["status"],
      d["mfma_roofline_frac_step"], d["mfma_roofline_frac_executed"])
d5 = json.loads(open("profiles/${R}_bench_cfg5_n1.json").read().strip().splitlines()[-1])
print(d5["value"], d5["ms_per_step"], d5["mfma_roofline_frac_step"], d5["mfma_roofline_frac_executed"])
PY
