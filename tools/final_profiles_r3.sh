#!/bin/bash
# Round 3: everything profiles/r03_* holds for the bench.  Run on an MI355X box from the repository root.
# rocprofv3: the program itself follows `--` (python3 ...), counters in their own passes with --kernel-trace only.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/final3
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^Setting" | tail -3
cd /tmp && export TMPDIR=/tmp
# (every rocprofv3 pass runs the bench with BNN_MI355X_LANES=1: one compute lane, whole-batch launches whose names and
#  durations are those of the bench line's stage table; `value` itself is timed on the shipped policy, which forks)
# 1. HBM traffic of the final kernels (two PMC passes, no load-time warm-up: its small launches of the same kernels would
#    be averaged in), then traffic.json, which the bench line quotes per stage
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
BNN_MI355X_LANES=1 BNN_MI355X_NO_WARMUP=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.json 2>$O/pmc_fetch.err
BNN_MI355X_LANES=1 BNN_MI355X_NO_WARMUP=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.json 2>$O/pmc_write.err
python3 $R/tools/pmc_summary.py $O/pmc_fetch $O/pmc_write 131072 > $O/pmc_traffic_cnvW1A1.txt
cat $O/pmc_traffic_cnvW1A1.txt
python3 $R/tools/make_traffic_json.py cnvW1A1 $O/pmc_traffic_cnvW1A1.txt "profiles/r03_pmc_traffic_cnvW1A1.txt: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (round 3, tools/final_profiles_r3.sh, no load-time warm-up), FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM section), per stage in launch order" $O/traffic.json $R/profiles/traffic.json
cp $(ls $O/pmc_fetch/*/*counter_collection.csv | head -1) $O/pmc_fetch_counter_collection.csv
cp $(ls $O/pmc_write/*/*counter_collection.csv | head -1) $O/pmc_write_counter_collection.csv
rm -rf $O/pmc_fetch $O/pmc_write
echo "pmc traffic done"
# 2. bench lines
python3 $R/bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json
echo "bench default done"; tail -c 400 $O/bench_default.json; echo
for n in cnvW1A2 cnvW2A2 lfcW1A1 lfcW1A2; do python3 $R/bench.py --network $n --no-extras 2>/dev/null | tail -1 > $O/bench_$n.json; done
echo "bench lines done"
# 3. kernel trace of the bench command
B="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras"
BNN_MI355X_LANES=1 BNN_MI355X_NO_WARMUP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $B > $O/prof_bench.json 2>$O/prof.err
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv; rm -rf $O/prof
echo "kernel trace done"
# 4. SQ passes: the bench (all nine CNV stages incl. k_conv0_tile), k_lfc_block_s at 10 000 images
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
export BNN_MI355X_NO_WARMUP=1 BNN_MI355X_LANES=1
bash $R/tools/sq_passes.sh final3/sq_bench -- $B
BATCHES=10000 bash $R/tools/sq_passes.sh final3/sq_lfc10k -- python3 $R/tools/batch_sweep.py lfcW1A1
cd /tmp
BATCHES=10000 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_lfc10k -- python3 $R/tools/batch_sweep.py lfcW1A1 > $O/kt_lfc10k.out 2>$O/kt_lfc10k.err
cp $(ls $O/kt_lfc10k/*/*kernel_stats.csv | head -1) $O/kernel_stats_lfc_block_s_10000.csv; rm -rf $O/kt_lfc10k
unset BNN_MI355X_NO_WARMUP BNN_MI355X_LANES
echo "sq passes done"
# 5. single-image latency: what the ABI reports next to the kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_latency -- python3 $R/tools/latency.py > $O/latency_under_trace.txt 2>$O/kt_latency.err
cp $(ls $O/kt_latency/*/*kernel_stats.csv | head -1) $O/kernel_stats_single_image.csv; rm -rf $O/kt_latency
cd $R
python3 tools/latency.py 2>&1 | grep -v "^Setting\|amdgpu.ids" > $O/latency.txt; cat $O/latency.txt
# 6. the plain multi-rank start on this one-GPU box: the JSON error, and the 4-rank gloo rehearsal
set +e
python3 bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_gpus2_on_one_gpu.json 2>/dev/null; echo "rc=$?" >> $O/bench_gpus2_on_one_gpu.json
set -e
python3 bench.py --gpus 4 --rehearse-gloo --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/rehearse_gloo_4_plain_start.json
python3 bench.py --dist-world1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/rccl_world1.json
# (the same two-rank rehearsal under a launcher, as the bench contract words it)
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --rehearse-gloo --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/rehearse_gloo_2_torchrun.json
tail -c 300 $O/rehearse_gloo_2_torchrun.json; echo
# 7. host paths and sweeps on the final build
python3 tools/path_rates.py cnvW1A1 131072 7 2>/dev/null | grep -v Setting > $O/path_rates.txt
python3 tools/path_rates.py cnvW1A1 1048576 3 2>/dev/null | grep -v Setting >> $O/path_rates.txt
python3 tools/path_rates.py lfcW1A1 131072 7 2>/dev/null | grep -v Setting >> $O/path_rates.txt
for plan in 32768:0:131072:200 131072:0:131072:200 16384:0:65536:200; do BNN_MI355X_CHUNKS=$plan python3 tools/path_rates.py lfcW1A1 131072 7 2>/dev/null | grep -v Setting >> $O/path_rates.txt; done
cat $O/path_rates.txt
BATCHES=1,256,1024,4096,4097,10000,32768,65536,131072 python3 tools/batch_sweep.py lfcW1A1 2>/dev/null | grep -v Setting > $O/batch_sweep_lfcW1A1.txt
BATCHES=1,256,1024,4096,8191,8192,10000,32768,131072 python3 tools/batch_sweep.py cnvW1A1 2>/dev/null | grep -v Setting > $O/batch_sweep_cnvW1A1.txt
cat $O/batch_sweep_lfcW1A1.txt $O/batch_sweep_cnvW1A1.txt
echo "final3 done"
