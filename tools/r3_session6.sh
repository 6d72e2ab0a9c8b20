#!/bin/bash
# round 3, sixth GPU session: priority schemes of k_lfc_block_s, its policy range, chunk-plan growth, feeder piece size, mmap probe
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3s6
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lfc or file_abi or chunk" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
V=$R/bnn-pynq_amd/build/variants
for rep in 1 2; do
  BATCHES=4097,10000,16384,32768 python3 tools/batch_sweep.py lfcW1A1 >> $O/lfc_prio.txt 2>&1
  BNN_MI355X_LIBDIR=$V/quarter BATCHES=4097,10000,16384,32768 python3 tools/batch_sweep.py lfcW1A1 2>&1 | sed 's/^/quarter /' >> $O/lfc_prio.txt
done
BNN_MI355X_LFC_BLOCK_MAX=1000000 BATCHES=32768,49152,65536,98304,131072 python3 tools/batch_sweep.py lfcW1A1 2>&1 | sed 's/^/block  /' >> $O/lfc_prio.txt
BNN_MI355X_LFC_BLOCK_MAX=0 BATCHES=32768,49152,65536,98304,131072 python3 tools/batch_sweep.py lfcW1A1 2>&1 | sed 's/^/staged /' >> $O/lfc_prio.txt
BNN_MI355X_LIBDIR=$V/quarter BNN_MI355X_LFC_BLOCK_MAX=1000000 BATCHES=65536,131072 python3 tools/batch_sweep.py lfcW1A1 2>&1 | sed 's/^/quarter block /' >> $O/lfc_prio.txt
grep -v "Setting\|amdgpu.ids" $O/lfc_prio.txt
BNN_MI355X_LIBDIR=$V/stamps python3 tools/lfc_stamps.py 10000 > $O/lfc_stamps.txt 2>&1
BNN_MI355X_LIBDIR=$V/qstamps python3 tools/lfc_stamps.py 10000 > $O/lfc_qstamps.txt 2>&1
grep -v "^Setting\|amdgpu.ids" $O/lfc_stamps.txt | head -28
grep -v "^Setting\|amdgpu.ids" $O/lfc_qstamps.txt | head -28
for plan in 2048:4096:32768:200 2048:0:32768:140 2048:0:32768:150 4096:0:32768:140 2048:4096:32768:140 2048:2048:32768:125; do
  BNN_MI355X_CHUNKS=$plan timeout -k 10 300 python3 tools/path_rates.py cnvW1A1 131072 >> $O/path_rates.txt 2>>$O/path_rates.err
done
for mb in 4 16; do
  BNN_MI355X_FEEDER_PIECE_MB=$mb BNN_MI355X_CHUNKS=2048:4096:32768:140 timeout -k 10 300 python3 tools/path_rates.py cnvW1A1 131072 2>>$O/path_rates.err | sed "s/^/piece=${mb}MB /" >> $O/path_rates.txt
done
BNN_MI355X_CHUNKS=2048:0:32768:140 timeout -k 10 300 python3 tools/path_rates.py lfcW1A1 131072 >> $O/path_rates.txt 2>>$O/path_rates.err
BNN_MI355X_CHUNKS=2048:0:32768:140 timeout -k 10 300 python3 tools/path_rates.py cnvW1A1 1048576 3 >> $O/path_rates.txt 2>>$O/path_rates.err
cat $O/path_rates.txt
python3 tools/h2d_probe.py > $O/h2d_probe.txt 2>&1
tail -12 $O/h2d_probe.txt
echo session6 done
