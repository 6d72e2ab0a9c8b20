set -e
mkdir -p gpurun_out/r2f
python3 bench.py > gpurun_out/r2f/bench_default.json 2> gpurun_out/r2f/bench_default.err || { tail -20 gpurun_out/r2f/bench_default.err; exit 1; }
tail -c 3000 gpurun_out/r2f/bench_default.json; echo
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 4 --steps 5 --warmup 2 --rehearse-gloo > gpurun_out/r2f/rehearse4.json 2> gpurun_out/r2f/rehearse4.err || { tail -30 gpurun_out/r2f/rehearse4.err; exit 1; }
tail -c 2500 gpurun_out/r2f/rehearse4.json; echo
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6
