for n in cnvW1A2 cnvW2A2 lfcW1A1 lfcW1A2; do python bench.py --network $n --steps 10 --warmup 2 --no-extras 2>/dev/null | tail -1 > gpurun_out/bench_$n.json; done
