#!/bin/bash
# tools/sq_passes.sh NAME -- PROGRAM ARGS...: the three SQ/GRBM --pmc passes of DESIGN.md 5 around one command, summary to
# gpurun_out/NAME/sq_summary.json.  The program itself follows `--` (python3 ..., never a shell).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=$1; shift; [ "$1" = "--" ] && shift
O=$R/gpurun_out/$NAME
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/sq1 -- "$@" > $O/sq1.out 2>$O/sq1.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/sq2 -- "$@" > $O/sq2.out 2>$O/sq2.err
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/grbm -- "$@" > $O/grbm.out 2>$O/grbm.err
python3 $R/tools/sq_summary.py $O/sq1 $O/sq2 $O/grbm > $O/sq_summary.json
rm -rf $O/sq1 $O/sq2 $O/grbm
echo "$NAME done"
