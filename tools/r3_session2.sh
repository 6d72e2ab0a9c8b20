#!/bin/bash
# round 3, second GPU session: layer-0 tile form (aligned LDS reads) A/B, k_lfc_block_s time stamps and the two-chain
# variant, single-image latency with the dispatch's own timestamps
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3s2
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
python3 tools/stage_times.py cnvW1A1 131072 2048 8192 > $O/l0_ab.txt 2>&1
BNN_MI355X_L0_TILE_MIN=1000000000 python3 tools/stage_times.py cnvW1A1 131072 2048 8192 >> $O/l0_ab.txt 2>&1
python3 tools/stage_times.py cnvW2A2 131072 >> $O/l0_ab.txt 2>&1
BNN_MI355X_L0_TILE_MIN=1000000000 python3 tools/stage_times.py cnvW2A2 131072 >> $O/l0_ab.txt 2>&1
grep -v "^Setting\|amdgpu.ids" $O/l0_ab.txt
V=$R/bnn-pynq_amd/build/variants
BNN_MI355X_LIBDIR=$V/stamps python3 tools/lfc_stamps.py 10000 > $O/lfc_stamps.txt 2>&1
BNN_MI355X_LIBDIR=$V/stamps python3 tools/lfc_stamps.py 32768 >> $O/lfc_stamps.txt 2>&1
grep -v "^Setting\|amdgpu.ids" $O/lfc_stamps.txt
for rep in 1 2; do
  BATCHES=4097,10000,16384,32768 python3 tools/batch_sweep.py lfcW1A1 >> $O/lfc_chains.txt 2>&1
  BNN_MI355X_LIBDIR=$V/chains2 BATCHES=4097,10000,16384,32768 python3 tools/batch_sweep.py lfcW1A1 2>&1 | sed 's/^/chains2 /' >> $O/lfc_chains.txt
done
grep -v "Setting\|amdgpu.ids" $O/lfc_chains.txt
python3 tools/latency.py > $O/latency.txt 2>&1
grep -v "^Setting\|amdgpu.ids" $O/latency.txt
echo session2 done
