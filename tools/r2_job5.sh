set -e
mkdir -p gpurun_out/r2e
./tools/microbench10 2>&1 | tee gpurun_out/r2e/microbench10.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_layers.py -x -q -m gpu -k "matrix_pipe" 2>&1 | tail -15
echo "== L1 on the matrix pipe"; BNN_MI355X_L1=mfma python3 tools/stage_times.py cnvW1A1 131072 2>&1 | grep -v "amdgpu.ids\|Setting network" | tee gpurun_out/r2e/l1_mfma_stage.txt
