set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final3; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json
for n in lfcW1A1 lfcW1A2; do python3 $R/bench.py --network $n --no-extras 2>/dev/null | tail -1 > $O/bench_$n.json; done
export BNN_MI355X_NO_WARMUP=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/prof_bench.json 2>$O/prof.err
echo done; tail -c 400 $O/prof_bench.json
