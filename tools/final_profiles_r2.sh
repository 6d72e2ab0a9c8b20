#!/bin/bash
# Round 2: everything profiles/r02_* holds.  Run on an MI355X box from the repository root:
#   bash tools/final_profiles_r2.sh           (bench lines, kernel trace, PMC passes, SQ passes)
# rocprofv3: the program itself follows `--` (python3 ...), counters in their own passes with --kernel-trace only.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/final2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json
echo "bench default done"; tail -c 600 $O/bench_default.json; echo
for n in cnvW1A2 cnvW2A2 lfcW1A1 lfcW1A2; do python3 $R/bench.py --network $n --no-extras 2>/dev/null | tail -1 > $O/bench_$n.json; done
echo "bench lines done"
B="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras"
BNN_MI355X_NO_WARMUP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $B > $O/prof_bench.json 2>$O/prof.err  # (without the load-time warm-up, whose small launches of the same kernels would be averaged in)
echo "kernel trace done"
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.json 2>$O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.json 2>$O/pmc_write.err
echo "pmc traffic done"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/sq1 -- $B > $O/sq1.json 2>$O/sq1.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/sq2 -- $B > $O/sq2.json 2>$O/sq2.err
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/grbm -- $B > $O/grbm.json 2>$O/grbm.err
echo "sq passes done"
python3 $R/tools/sq_summary.py $O/sq1 $O/sq2 $O/grbm > $O/sq_summary.json
python3 $R/tools/pmc_summary.py $O/pmc_fetch $O/pmc_write 131072 > $O/pmc_traffic.txt
echo "summaries done"
