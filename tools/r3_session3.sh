#!/bin/bash
# round 3, third GPU session: pinned-ring feeder A/B on the host-data entry points, per-wave stamps of k_lfc_block_s,
# single-image latency with device-scope time events
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3s3
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
nproc > $O/path_rates.txt; cat /sys/fs/cgroup/cpu.max >> $O/path_rates.txt 2>/dev/null || true
timeout -k 10 300 python3 tools/path_rates.py cnvW1A1 131072 >> $O/path_rates.txt 2>$O/path_rates.err
BNN_MI355X_NO_FEEDER=1 timeout -k 10 300 python3 tools/path_rates.py cnvW1A1 131072 2>>$O/path_rates.err | sed 's/^/nofeeder /' >> $O/path_rates.txt
for t in 4 8 12 14; do
  BNN_MI355X_FEEDER_THREADS=$t timeout -k 10 300 python3 tools/path_rates.py cnvW1A1 131072 2>>$O/path_rates.err | sed "s/^/threads=$t /" >> $O/path_rates.txt
done
for plan in 4096:8192:32768 2048:4096:16384 2048:0:32768; do
  BNN_MI355X_CHUNKS=$plan timeout -k 10 300 python3 tools/path_rates.py cnvW1A1 131072 >> $O/path_rates.txt 2>>$O/path_rates.err
done
timeout -k 10 300 python3 tools/path_rates.py lfcW1A1 131072 >> $O/path_rates.txt 2>>$O/path_rates.err
BNN_MI355X_NO_FEEDER=1 timeout -k 10 300 python3 tools/path_rates.py lfcW1A1 131072 2>>$O/path_rates.err | sed 's/^/nofeeder /' >> $O/path_rates.txt
timeout -k 10 300 python3 tools/path_rates.py cnvW1A1 1048576 3 >> $O/path_rates.txt 2>>$O/path_rates.err
timeout -k 10 300 python3 tools/path_rates.py cnvW1A1 10000 >> $O/path_rates.txt 2>>$O/path_rates.err
cat $O/path_rates.txt
V=$R/bnn-pynq_amd/build/variants
BNN_MI355X_LIBDIR=$V/stamps python3 tools/lfc_stamps.py 10000 > $O/lfc_stamps.txt 2>&1
grep -v "^Setting\|amdgpu.ids" $O/lfc_stamps.txt
python3 tools/latency.py > $O/latency.txt 2>&1
grep -v "^Setting\|amdgpu.ids" $O/latency.txt
echo session3 done
