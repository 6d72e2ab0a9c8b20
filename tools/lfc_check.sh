#!/bin/bash
# tools/lfc_check.sh: the LFC parity tests, the stress of k_lfc_block_s's hand-offs, its sweep and its clock stamps (GPU box)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/lfc_check
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_layers.py -m gpu -x -q -k "lfc" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python3 tools/stress_lfc_block.py 120 > $O/stress.txt 2>&1 || { tail -5 $O/stress.txt; exit 1; }
tail -1 $O/stress.txt
for rep in 1 2; do BATCHES=1025,2048,4097,10000,16384,32768,65536,131072 python3 tools/batch_sweep.py lfcW1A1 2>&1 | grep -v "Setting\|amdgpu" >> $O/sweep.txt; done
cat $O/sweep.txt
BNN_MI355X_LIBDIR=$R/bnn-pynq_amd/build/variants/stamps python3 tools/lfc_stamps.py 10000 2>&1 | grep -v "^Setting\|amdgpu.ids" > $O/stamps.txt
head -26 $O/stamps.txt
