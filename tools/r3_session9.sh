#!/bin/bash
# round 3, ninth GPU session: single-image LFC kernel with / without all rows requested at entry; where k_lfc_block_s takes over; fuzz
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3s9
mkdir -p $O
cd $R
V=$R/bnn-pynq_amd/build/variants
for rep in 1 2 3; do
  python3 tools/latency.py lfcW1A1 2>&1 | grep lfcW1A1 >> $O/latency_ab.txt
  BNN_MI355X_LIBDIR=$V/noallrows python3 tools/latency.py lfcW1A1 2>&1 | grep lfcW1A1 | sed 's/^/noallrows /' >> $O/latency_ab.txt
done
cat $O/latency_ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_a -- python3 $R/tools/latency.py lfcW1A1 > /dev/null 2>$O/kt_a.err
BNN_MI355X_LIBDIR=$V/noallrows rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_b -- python3 $R/tools/latency.py lfcW1A1 > /dev/null 2>$O/kt_b.err
grep "lfc_fused<1>" $O/kt_a/*/*kernel_stats.csv | cut -c1-60,160-260; grep "lfc_fused<1>" $O/kt_b/*/*kernel_stats.csv | cut -c1-60,160-260
cp $(ls $O/kt_a/*/*kernel_stats.csv | head -1) $O/kernel_stats_all_rows.csv; cp $(ls $O/kt_b/*/*kernel_stats.csv | head -1) $O/kernel_stats_two_rows.csv; rm -rf $O/kt_a $O/kt_b
cd $R
for fm in 4096 2048 1024 512 256; do
  BNN_MI355X_LFC_FUSED_MAX=$fm BATCHES=257,512,513,1024,1025,1536,2048,2049,3072,4096 python3 tools/batch_sweep.py lfcW1A1 2>&1 | grep -v Setting | sed "s/^/fused_max=$fm /" >> $O/fused_vs_block.txt
done
cat $O/fused_vs_block.txt
timeout -k 10 1000 python3 tools/fuzz_sizes.py 5 > $O/fuzz.txt 2>&1 || { tail -5 $O/fuzz.txt; exit 1; }
grep -v "^Setting" $O/fuzz.txt | tail -9
echo session9 done
