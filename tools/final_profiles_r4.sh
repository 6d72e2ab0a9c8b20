#!/bin/bash
# Round 4: what profiles/r04_* holds.  tools/final_profiles_r4.sh PART, on an MI355X box from the repository root:
#   A  GPU tests, smoke, the bench lines, kernel trace of the bench command (cnvW1A1)
#   B  BASELINE config 4 (cnvW2A2) and cnvW1A2: PMC traffic passes, kernel trace, SQ passes
#   C  the host paths: rates at the reference's call size and at the headline batch, plan sweep, latency, call traces,
#      device timelines
# rocprofv3: the program itself follows `--` (python3 ...), counters in their own passes with --kernel-trace only.
set -e
PART=${1:-A}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/final4
mkdir -p $O
cd $R
if [ $PART = A ]; then
  timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
  tail -2 $O/tests.log
  python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^Setting" | tail -3
  python3 $R/bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json
  echo "bench default done"; tail -c 300 $O/bench_default.json; echo
  for n in cnvW1A2 cnvW2A2 lfcW1A1 lfcW1A2; do python3 $R/bench.py --network $n --no-extras 2>/dev/null | tail -1 > $O/bench_$n.json; done
  cd /tmp && export TMPDIR=/tmp
  B="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras"
  BNN_MI355X_LANES=1 BNN_MI355X_NO_WARMUP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $B > $O/prof_bench.json 2>$O/prof.err
  cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv; rm -rf $O/prof
  echo "part A done"
fi
if [ $PART = B ]; then
  cd /tmp && export TMPDIR=/tmp
  export BNN_MI355X_LANES=1 BNN_MI355X_NO_WARMUP=1
  for NET in cnvW2A2 cnvW1A2; do
    B="python3 $R/bench.py --network $NET --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$NET -- $B > $O/pmc_fetch_$NET.json 2>$O/pmc_fetch_$NET.err
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$NET -- $B > $O/pmc_write_$NET.json 2>$O/pmc_write_$NET.err
    python3 $R/tools/pmc_summary.py $O/pmc_fetch_$NET $O/pmc_write_$NET 131072 > $O/pmc_traffic_$NET.txt
    cat $O/pmc_traffic_$NET.txt
    python3 $R/tools/make_traffic_json.py $NET $O/pmc_traffic_$NET.txt "profiles/r04_pmc_traffic_$NET.txt: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (round 4, tools/final_profiles_r4.sh B, one compute lane, no load-time warm-up), FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM section), per stage in launch order" $O/traffic.json
    cp $(ls $O/pmc_fetch_$NET/*/*counter_collection.csv | head -1) $O/pmc_fetch_counter_collection_$NET.csv
    cp $(ls $O/pmc_write_$NET/*/*counter_collection.csv | head -1) $O/pmc_write_counter_collection_$NET.csv
    rm -rf $O/pmc_fetch_$NET $O/pmc_write_$NET
    B="python3 $R/bench.py --network $NET --steps 10 --warmup 2 --no-cpu-baseline --no-extras"
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$NET -- $B > $O/prof_bench_$NET.json 2>$O/prof_$NET.err
    cp $(ls $O/prof_$NET/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats_$NET.csv; rm -rf $O/prof_$NET
    echo "$NET traffic + trace done"
  done
  B="python3 $R/bench.py --network cnvW2A2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
  bash $R/tools/sq_passes.sh final4/sq_cnvW2A2 -- $B
  echo "part B done"
fi
if [ $PART = C ]; then
  S=$O/path_rates.txt; : > $S
  for a in "cnvW1A1 10000" "cnvW1A1 131072" "cnvW1A1 1048576" "lfcW1A1 10000" "lfcW1A1 131072" "lfcW1A2 131072" "cnvW2A2 10000"; do
    REPS=9 python3 tools/small_call_sweep.py $a default 2>/dev/null | grep -v "^Setting" >> $S
  done
  cat $S
  P=$O/plan_sweep.txt; : > $P
  REPS=9 python3 tools/small_call_sweep.py cnvW1A1 10000 default 4096:0:16384:150 2048:0:16384:150 512:0:16384:200 1024:1024:16384:200 512:1024:16384:200 2>/dev/null | grep -v "^Setting" >> $P
  BNN_MI355X_NO_COPIER=1 REPS=9 python3 tools/small_call_sweep.py cnvW1A1 10000 default 2>/dev/null | grep -v "^Setting" >> $P
  BNN_MI355X_LANES=1 REPS=9 python3 tools/small_call_sweep.py cnvW1A1 10000 default 2>/dev/null | grep -v "^Setting" >> $P
  BNN_MI355X_NO_MAPPED_RESULTS=1 REPS=9 python3 tools/small_call_sweep.py cnvW1A1 10000 default 2>/dev/null | grep -v "^Setting" >> $P
  REPS=9 python3 tools/small_call_sweep.py cnvW1A1 131072 default 512:512:16384:200 2048:0:16384:150 2>/dev/null | grep -v "^Setting" >> $P
  REPS=9 python3 tools/small_call_sweep.py lfcW1A1 131072 default 2048:0:32768:200 8192:0:32768:150 2>/dev/null | grep -v "^Setting" >> $P
  BNN_MI355X_NO_HOST_PACK=1 REPS=9 python3 tools/small_call_sweep.py lfcW1A1 131072 default 2>/dev/null | grep -v "^Setting" >> $P
  REPS=9 python3 tools/small_call_sweep.py lfcW1A1 10000 default 2048:0:32768:200 2>/dev/null | grep -v "^Setting" >> $P
  BNN_MI355X_NO_HOST_PACK=1 REPS=9 python3 tools/small_call_sweep.py lfcW1A1 10000 default 2>/dev/null | grep -v "^Setting" >> $P
  cat $P
  python3 tools/latency.py 2>&1 | grep -v "^Setting\|amdgpu.ids" > $O/latency.txt
  BNN_MI355X_DIRECT_TIMING=host python3 tools/latency.py 2>&1 | grep -v "^Setting\|amdgpu.ids" | sed 's/^/BNN_MI355X_DIRECT_TIMING=host (opt-in: no event packets, completion word in pinned memory, usecPerImage by the host clock)  /' >> $O/latency.txt
  BNN_MI355X_NO_DIRECT=1 python3 tools/latency.py 2>&1 | grep -v "^Setting\|amdgpu.ids" | sed 's/^/BNN_MI355X_NO_DIRECT=1 (the round-3 way: copies around the launch)  /' >> $O/latency.txt
  cat $O/latency.txt
  for a in "cnvW1A1 10000" "lfcW1A1 10000" "lfcW1A1 131072"; do python3 tools/call_trace.py $a 2> $O/call_trace_$(echo $a | tr " " _).txt; done
  cd /tmp && export TMPDIR=/tmp
  for k in file buffer; do
    rocprofv3 --kernel-trace --memory-copy-trace -d $O/tl_$k -o tl --output-format csv -- python3 $R/tools/host_timeline.py cnvW1A1 10000 $([ $k = file ] && echo file) > $O/tl_$k.log 2>&1
    python3 $R/tools/timeline_summary.py $O/tl_$k > $O/timeline_cnvW1A1_10000_$k.txt 2>&1; rm -rf $O/tl_$k
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_latency -- python3 $R/tools/latency.py > $O/latency_under_trace.txt 2>$O/kt_latency.err
  cp $(ls $O/kt_latency/*/*kernel_stats.csv | head -1) $O/kernel_stats_single_image.csv; rm -rf $O/kt_latency
  echo "part C done"
fi
