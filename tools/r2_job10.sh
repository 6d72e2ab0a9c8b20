set -e
mkdir -p gpurun_out/r2k
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6
for mx in 4096 1000000; do echo "== BNN_MI355X_LFC_BLOCK_MAX=$mx"; BNN_MI355X_LFC_BLOCK_MAX=$mx BATCHES=4097,6000,8192,10000,12288,16384,24576,32768,49152,65536,131072 python3 tools/batch_sweep.py lfcW1A1; done 2>&1 | grep -v "amdgpu.ids\|Setting network" | tee gpurun_out/r2k/lfc_block_sweep.txt
