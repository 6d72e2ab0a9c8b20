set -e
mkdir -p gpurun_out/r2c
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -8
for net in cnvW1A1 cnvW1A2 cnvW2A2 lfcW1A1 lfcW1A2; do python3 tools/stage_times.py $net 131072; done 2>&1 | grep -v "amdgpu.ids\|Setting network" | tee gpurun_out/r2c/stages.txt
for mx in 4096 200000; do echo "== BNN_MI355X_LFC_BLOCK_MAX=$mx"; BNN_MI355X_LFC_BLOCK_MAX=$mx BATCHES=4097,6000,8192,10000,16384,32768,65536,131072 python3 tools/batch_sweep.py lfcW1A1; done 2>&1 | grep -v "amdgpu.ids\|Setting network" | tee gpurun_out/r2c/lfc_block_sweep.txt
