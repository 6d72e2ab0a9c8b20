#!/bin/bash
# tools/collect_final_r4.sh: copies what tools/final_profiles_r4.sh {A,B,C} left in gpurun_out/final4/ into profiles/ under
# the names profiles/README.md lists (run here, after the gpurun calls have merged their output back)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/final4; P=$R/profiles
for n in default cnvW1A2 cnvW2A2 lfcW1A1 lfcW1A2; do cp $O/bench_$n.json $P/r04_bench_$n.json; done
cp $O/bench_kernel_stats.csv $P/r04_bench_kernel_stats.csv
for n in cnvW2A2 cnvW1A2; do
  cp $O/bench_kernel_stats_$n.csv $P/r04_bench_kernel_stats_$n.csv
  cp $O/pmc_traffic_$n.txt $P/r04_pmc_traffic_$n.txt
  cp $O/pmc_fetch_counter_collection_$n.csv $P/r04_pmc_fetch_counter_collection_$n.csv
  cp $O/pmc_write_counter_collection_$n.csv $P/r04_pmc_write_counter_collection_$n.csv
done
cp $O/sq_cnvW2A2/sq_summary.json $P/r04_sq_summary_cnvW2A2.json
python3 - "$O/traffic.json" "$P/traffic.json" <<'PY'
import json, sys
new, cur = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
cur.update(new)
json.dump(cur, open(sys.argv[2], "w"), indent=1)
print("traffic.json:", sorted(cur))
PY
if [ -f $O/path_rates.txt ]; then
  cp $O/path_rates.txt $P/r04_path_rates.txt
  cp $O/plan_sweep.txt $P/r04_small_call_plan_sweep.txt
  cp $O/latency.txt $P/r04_latency.txt
  cp $O/kernel_stats_single_image.csv $P/r04_kernel_stats_single_image.csv
  for k in file buffer; do cp $O/timeline_cnvW1A1_10000_$k.txt $P/r04_timeline_cnvW1A1_10000_$k.txt; done
  for a in cnvW1A1_10000 lfcW1A1_10000 lfcW1A1_131072; do grep -v "^Setting\|amdgpu.ids" $O/call_trace_$a.txt > $P/r04_call_trace_$a.txt; done
fi
echo collected
