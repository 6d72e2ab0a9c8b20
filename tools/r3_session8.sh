#!/bin/bash
# round 3, eighth GPU session: reader threads / piece size / plan of the file path
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3s8
mkdir -p $O
cd $R
for t in 2 3 4 5 6; do
  BNN_MI355X_FEEDER_THREADS=$t timeout -k 10 300 python3 tools/path_rates.py cnvW1A1 131072 7 2>>$O/path_rates.err | sed "s/^/threads=$t /" >> $O/path_rates.txt
done
for mb in 2 4 16; do
  BNN_MI355X_FEEDER_PIECE_MB=$mb BNN_MI355X_FEEDER_THREADS=4 timeout -k 10 300 python3 tools/path_rates.py cnvW1A1 131072 7 2>>$O/path_rates.err | sed "s/^/threads=4 piece=${mb}MB /" >> $O/path_rates.txt
done
for plan in 2048:2048:32768:150 2048:4096:32768:150 2048:0:32768:150 4096:4096:32768:125 2048:2048:16384:125; do
  BNN_MI355X_CHUNKS=$plan BNN_MI355X_FEEDER_THREADS=4 timeout -k 10 300 python3 tools/path_rates.py cnvW1A1 131072 7 2>>$O/path_rates.err | sed "s/^/threads=4 /" >> $O/path_rates.txt
done
BNN_MI355X_FEEDER_THREADS=4 timeout -k 10 300 python3 tools/path_rates.py cnvW1A1 1048576 3 2>>$O/path_rates.err | sed "s/^/threads=4 /" >> $O/path_rates.txt
BNN_MI355X_FEEDER_THREADS=4 timeout -k 10 300 python3 tools/path_rates.py lfcW1A1 131072 7 2>>$O/path_rates.err | sed "s/^/threads=4 /" >> $O/path_rates.txt
BNN_MI355X_FEEDER_THREADS=4 timeout -k 10 300 python3 tools/path_rates.py cnvW2A2 131072 5 2>>$O/path_rates.err | sed "s/^/threads=4 /" >> $O/path_rates.txt
grep -v Setting $O/path_rates.txt
echo session8 done
