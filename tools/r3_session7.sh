#!/bin/bash
# round 3, seventh GPU session: full suite on the new policies, file path with one / two DMA queues, whole bench line
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3s7
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python3 tools/path_rates.py cnvW1A1 131072 >> $O/path_rates.txt 2>$O/path_rates.err
for t in 4 8; do
  BNN_MI355X_FEEDER_THREADS=$t timeout -k 10 300 python3 tools/path_rates.py cnvW1A1 131072 2>>$O/path_rates.err | sed "s/^/threads=$t /" >> $O/path_rates.txt
  BNN_MI355X_FEEDER_STREAMS=2 BNN_MI355X_FEEDER_THREADS=$t timeout -k 10 300 python3 tools/path_rates.py cnvW1A1 131072 2>>$O/path_rates.err | sed "s/^/streams=2 threads=$t /" >> $O/path_rates.txt
done
BNN_MI355X_FEEDER_STREAMS=2 BNN_MI355X_FEEDER_PIECE_MB=4 timeout -k 10 300 python3 tools/path_rates.py cnvW1A1 131072 2>>$O/path_rates.err | sed "s/^/streams=2 piece=4MB /" >> $O/path_rates.txt
BNN_MI355X_FEEDER_STREAMS=2 timeout -k 10 300 python3 tools/path_rates.py cnvW1A1 1048576 3 2>>$O/path_rates.err | sed "s/^/streams=2 /" >> $O/path_rates.txt
timeout -k 10 300 python3 tools/path_rates.py cnvW1A1 1048576 3 >> $O/path_rates.txt 2>>$O/path_rates.err
timeout -k 10 300 python3 tools/path_rates.py lfcW1A1 131072 >> $O/path_rates.txt 2>>$O/path_rates.err
timeout -k 10 300 python3 tools/path_rates.py cnvW1A1 10000 >> $O/path_rates.txt 2>>$O/path_rates.err
cat $O/path_rates.txt
python3 bench.py > $O/bench_default.json 2>$O/bench_default.err
tail -c 6000 $O/bench_default.json
python3 tools/latency.py > $O/latency.txt 2>&1
grep -v "^Setting\|amdgpu.ids" $O/latency.txt
echo session7 done
