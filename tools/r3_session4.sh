#!/bin/bash
# round 3, fourth GPU session: rotated image order in k_lfc_block_s (A/B, stamps), host -> HBM link probe
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3s4
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
V=$R/bnn-pynq_amd/build/variants
for rep in 1 2; do
  BATCHES=4097,6000,10000,16384,24576,32768,65536 python3 tools/batch_sweep.py lfcW1A1 >> $O/lfc_rot.txt 2>&1
  BNN_MI355X_LIBDIR=$V/norot BATCHES=4097,6000,10000,16384,24576,32768,65536 python3 tools/batch_sweep.py lfcW1A1 2>&1 | sed 's/^/norot /' >> $O/lfc_rot.txt
done
BNN_MI355X_LFC_BLOCK_MAX=1000000 BATCHES=32768,65536,131072 python3 tools/batch_sweep.py lfcW1A1 2>&1 | sed 's/^/blockmax /' >> $O/lfc_rot.txt
grep -v "Setting\|amdgpu.ids" $O/lfc_rot.txt
BNN_MI355X_LIBDIR=$V/stamps python3 tools/lfc_stamps.py 10000 > $O/lfc_stamps.txt 2>&1
grep -v "^Setting\|amdgpu.ids" $O/lfc_stamps.txt | head -40
# neuron order rotated per wave in k_vec_x (FC layers, CNV L4..L7): same-box A/B
for rep in 1 2; do
  python3 tools/stage_times.py lfcW1A1 131072 >> $O/vec_rot.txt 2>&1
  BNN_MI355X_LIBDIR=$V/vecnorot python3 tools/stage_times.py lfcW1A1 131072 2>&1 | sed 's/^/vecnorot /' >> $O/vec_rot.txt
done
python3 tools/stage_times.py cnvW1A1 131072 >> $O/vec_rot.txt 2>&1
BNN_MI355X_LIBDIR=$V/vecnorot python3 tools/stage_times.py cnvW1A1 131072 2>&1 | sed 's/^/vecnorot /' >> $O/vec_rot.txt
grep -v "Setting\|amdgpu.ids" $O/vec_rot.txt
python3 tools/stress_lfc_block.py > $O/stress.txt 2>&1 || { tail -5 $O/stress.txt; exit 1; }
tail -3 $O/stress.txt
python3 tools/h2d_probe.py > $O/h2d_probe.txt 2>&1
cat $O/h2d_probe.txt
echo session4 done
