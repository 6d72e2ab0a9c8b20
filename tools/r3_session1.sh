#!/bin/bash
# round 3, first GPU session: full GPU suite, the chunk-plan A/B of the host-data entry points, the plain multi-rank
# start of bench.py, SQ passes of k_lfc_block_s at BASELINE config 2's own size, kernel trace of the single-image kernels
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3s1
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
# layer 0: tile form (default from 2048 images) against the lane-per-pixel form, same box
python3 tools/stage_times.py cnvW1A1 131072 2048 8192 > $O/l0_ab.txt 2>&1
BNN_MI355X_L0_TILE_MIN=1000000000 python3 tools/stage_times.py cnvW1A1 131072 2048 8192 >> $O/l0_ab.txt 2>&1
python3 tools/stage_times.py cnvW2A2 131072 >> $O/l0_ab.txt 2>&1
BNN_MI355X_L0_TILE_MIN=1000000000 python3 tools/stage_times.py cnvW2A2 131072 >> $O/l0_ab.txt 2>&1
cat $O/l0_ab.txt
for plan in 0:0:32768 2048:4096:32768 2048:0:32768 4096:8192:32768 2048:4096:16384 1024:2048:32768; do
  BNN_MI355X_CHUNKS=$plan timeout -k 10 300 python3 tools/path_rates.py cnvW1A1 131072 >> $O/path_rates.txt 2>$O/path_rates.err
done
for plan in 0:0:32768 2048:4096:32768; do
  BNN_MI355X_CHUNKS=$plan timeout -k 10 300 python3 tools/path_rates.py lfcW1A1 131072 >> $O/path_rates.txt 2>>$O/path_rates.err
  BNN_MI355X_CHUNKS=$plan timeout -k 10 300 python3 tools/path_rates.py cnvW1A1 10000 >> $O/path_rates.txt 2>>$O/path_rates.err
  BNN_MI355X_CHUNKS=$plan timeout -k 10 300 python3 tools/path_rates.py cnvW1A1 1048576 3 >> $O/path_rates.txt 2>>$O/path_rates.err
done
cat $O/path_rates.txt
# the driver's plain command with N > 1: one GPU here, so (a) the JSON error, (b) the 4-rank gloo rehearsal
set +e
python3 bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_gpus2_plain.json 2>$O/bench_gpus2_plain.err; echo "rc=$?" >> $O/bench_gpus2_plain.json
set -e
cat $O/bench_gpus2_plain.json
timeout -k 10 600 python3 bench.py --gpus 4 --rehearse-gloo --steps 5 --warmup 2 > $O/rehearse_gloo_4.json 2>$O/rehearse_gloo_4.err
tail -c 1500 $O/rehearse_gloo_4.json; echo
# BASELINE config 2 at its own size: k_lfc_block_s at 10 000 images
export BNN_MI355X_NO_WARMUP=1
BATCHES=10000 bash tools/sq_passes.sh r3s1/sq_lfc10k -- python3 $R/tools/batch_sweep.py lfcW1A1
cd /tmp && export TMPDIR=/tmp
BATCHES=10000 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_lfc10k -- python3 $R/tools/batch_sweep.py lfcW1A1 > $O/kt_lfc10k.out 2>$O/kt_lfc10k.err
unset BNN_MI355X_NO_WARMUP
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_latency -- python3 $R/tools/latency.py > $O/kt_latency.out 2>$O/kt_latency.err
cat $O/kt_latency.out
cd $R
for d in kt_lfc10k kt_latency; do f=$(ls $O/$d/*/*kernel_stats.csv | head -1); cp $f $O/$d.kernel_stats.csv; done
python3 tools/latency.py > $O/latency_plain.out 2>&1; cat $O/latency_plain.out
echo session1 done
