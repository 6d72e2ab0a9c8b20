set -e
mkdir -p gpurun_out/r2i
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mid_batches or lfc" 2>&1 | tail -4
for form in s lds; do echo "== BNN_MI355X_LFC_BLOCK=$form (max 200000)"; BNN_MI355X_LFC_BLOCK=$form BNN_MI355X_LFC_BLOCK_MAX=200000 BATCHES=4097,6000,8192,10000,12288,16384,24576,32768,65536,131072 python3 tools/batch_sweep.py lfcW1A1; done 2>&1 | grep -v "amdgpu.ids\|Setting network" | tee gpurun_out/r2i/lfc_block_forms.txt
