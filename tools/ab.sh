#!/bin/bash
# tools/ab.sh [-n NETWORK] [-b BATCH] VARIANT...: times bench.py for each variant library (run on the GPU box)
NET=cnvW1A1; BATCH=131072
while getopts "n:b:" o; do case $o in n) NET=$OPTARG;; b) BATCH=$OPTARG;; esac; done; shift $((OPTIND-1))
for v in "$@"; do
  BNN_MI355X_LIBDIR=$PWD/bnn-pynq_amd/build/variants/$v python bench.py --network $NET --batch $BATCH --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value']), {k: round(x,3) for k,x in d['roofline']['stages_ms'].items()})"
done
