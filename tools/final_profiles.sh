#!/bin/bash
# Collects everything profiles/ holds for the bench: the bench lines of all five networks, the rocprofv3
# kernel-trace summary of the default bench command, and the two PMC passes (separate runs, kernel-trace only).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json
for n in cnvW1A2 cnvW2A2 lfcW1A1 lfcW1A2; do python3 $R/bench.py --network $n --no-extras 2>/dev/null | tail -1 > $O/bench_$n.json; done
echo "bench lines done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/prof_bench.json 2>$O/prof.err
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_fetch.json 2>$O/pmc_fetch.err
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_write.json 2>$O/pmc_write.err
echo "pmc write done"
