#!/bin/bash
# Everything the judged figures of a round come from, collected on the GPU box into gpurun_out/<tag>/ (copy what is to
# be kept into profiles/roundN/):  scripts/round_profiles.sh TAG
TAG=${1:-round}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
python3 bench.py --breakdown > $O/bench.json 2> $O/breakdown.txt
python3 bench.py --breakdown --no-bins --quick > $O/bench_no_bins.json 2> $O/breakdown_no_bins.txt
bash scripts/kernel_trace.sh $TAG/trace > $O/trace_gaps.txt 2>&1
cp $O/trace/t_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
bash scripts/pmc_sq.sh $TAG/sq > /dev/null 2>&1; cp $O/sq/summary.txt $O/sq_counters.txt 2>/dev/null
bash scripts/pmc_traffic.sh config4 > $O/pmc_traffic_config4.txt 2>&1; cp $R/gpurun_out/pmc_traffic/summary_config4.json $O/ 2>/dev/null
bash scripts/pmc_traffic.sh config2 > $O/pmc_traffic_config2.txt 2>&1; cp $R/gpurun_out/pmc_traffic/summary_config2.json $O/ 2>/dev/null
python3 - <<PY
import json, os
out = {}
for c in ("config4", "config2"):
    p = "$O/summary_%s.json" % c
    if os.path.exists(p): out[c] = json.load(open(p))
json.dump(out, open("$O/pmc_traffic_summary.json", "w"), indent=1)
PY
rm -rf $O/trace/*.csv.bak $O/sq/*.csv $O/trace/t_kernel_trace.csv; du -sh $O
