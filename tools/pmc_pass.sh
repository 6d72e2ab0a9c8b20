#!/bin/bash
# tools/pmc_pass.sh NAME "COUNTER COUNTER ..." -- PROGRAM ARGS...: one rocprofv3 --pmc pass (kernel-trace only) around a
# command; per kernel (largest grid only) the mean of each counter to gpurun_out/NAME.json (tools/sq_summary.py).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=$1; CTRS=$2; shift 2; [ "$1" = "--" ] && shift
O=$R/gpurun_out/$NAME.d
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $O -- "$@" > $O.out 2>$O.err
python3 $R/tools/sq_summary.py $O > $R/gpurun_out/$NAME.json
rm -rf $O
echo "$NAME done"
