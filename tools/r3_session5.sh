#!/bin/bash
# round 3, fifth GPU session: per-layer wave priority in k_lfc_block_s (A/B, stamps); device timeline of the host paths
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3s5
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lfc" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
V=$R/bnn-pynq_amd/build/variants
for rep in 1 2; do
  BATCHES=4097,6000,10000,16384,24576,32768,65536 python3 tools/batch_sweep.py lfcW1A1 >> $O/lfc_prio.txt 2>&1
  BNN_MI355X_LIBDIR=$V/noprio BATCHES=4097,6000,10000,16384,24576,32768,65536 python3 tools/batch_sweep.py lfcW1A1 2>&1 | sed 's/^/noprio /' >> $O/lfc_prio.txt
done
BNN_MI355X_LFC_BLOCK_MAX=1000000 BATCHES=32768,65536,131072 python3 tools/batch_sweep.py lfcW1A1 2>&1 | sed 's/^/blockmax /' >> $O/lfc_prio.txt
grep -v "Setting\|amdgpu.ids" $O/lfc_prio.txt
BNN_MI355X_LIBDIR=$V/stamps python3 tools/lfc_stamps.py 10000 > $O/lfc_stamps.txt 2>&1
grep -v "^Setting\|amdgpu.ids" $O/lfc_stamps.txt | head -34
python3 tools/stress_lfc_block.py 100 > $O/stress.txt 2>&1 || { tail -5 $O/stress.txt; exit 1; }
tail -1 $O/stress.txt
cd /tmp && export TMPDIR=/tmp
for mode in host file; do
  for feed in 1 0; do
    if [ $feed = 0 ]; then export BNN_MI355X_NO_FEEDER=1; else unset BNN_MI355X_NO_FEEDER; fi
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tl_${mode}_$feed -- python3 $R/tools/host_timeline.py cnvW1A1 131072 $mode > $O/tl_${mode}_$feed.out 2>$O/tl_${mode}_$feed.err
    echo "== $mode feeder=$feed"; grep "^call" $O/tl_${mode}_$feed.out
    python3 $R/tools/timeline_summary.py $O/tl_${mode}_$feed | tee $O/tl_${mode}_$feed.summary.txt
    rm -rf $O/tl_${mode}_$feed
  done
done
echo session5 done
