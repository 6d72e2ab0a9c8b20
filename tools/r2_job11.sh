set -e
mkdir -p gpurun_out/r2l
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lfc" 2>&1 | tail -3
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 4 --steps 5 --warmup 2 --rehearse-gloo > gpurun_out/r2l/rehearse4.json 2> gpurun_out/r2l/rehearse4.err || { tail -30 gpurun_out/r2l/rehearse4.err; exit 1; }
python3 -c "
import json
for f in ('rehearse4',):
    d=json.loads(open('gpurun_out/r2l/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], json.dumps(d['multi_gpu']))"
bash tools/final_profiles_r2.sh
