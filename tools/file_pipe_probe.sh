#!/bin/bash
# tools/file_pipe_probe.sh: the variants of tools/file_pipe_probe on a 403 MB file in the page cache (GPU box)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/file_pipe
mkdir -p $O
F=/tmp/pipe_probe.bin
python3 -c "import numpy as np; np.random.default_rng(0).integers(0,256,131072*3073,dtype=np.uint8).tofile('$F')"
cat $F > /dev/null
P=$R/tools/file_pipe_probe
{
for mode in 4 5; do $P $F 6 4 12 $mode | tail -1; done
for t in 2 4 6 8 12; do $P $F $t 4 12 0 | tail -1; done
for mb in 1 2 8; do $P $F 4 $mb 12 0 | tail -1; done
for sl in 4 24 48; do $P $F 4 4 $sl 0 | tail -1; done
for t in 4 8 12; do $P $F $t 4 12 1 | tail -1; done
for t in 4 8 12; do $P $F $t 4 12 2 | tail -1; done
for t in 4 8; do $P $F $t 4 12 3 | tail -1; done
$P $F 4 2 24 3 | tail -1
} 2>&1 | tee $O/probe.txt
rm -f $F
