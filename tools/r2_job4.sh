set -e
mkdir -p gpurun_out/r2d
./tools/mfma_fp4_probe 2>&1 | tee gpurun_out/r2d/mfma_fp4_probe.txt
echo "== block kernel as built"; BNN_MI355X_LFC_BLOCK_MAX=200000 BATCHES=10000,131072 python3 tools/batch_sweep.py lfcW1A1 2>&1 | grep batch
echo "== block kernel without LDS reads in the image loop (wrong results, timing only)"; BNN_MI355X_LIBDIR=$PWD/bnn-pynq_amd/build/variants/nolds BNN_MI355X_LFC_BLOCK_MAX=200000 BATCHES=10000,131072 python3 tools/batch_sweep.py lfcW1A1 2>&1 | grep batch
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lfc_single_image_decode or two_streams or mid_batches or layer0_threshold" 2>&1 | tail -5
timeout -k 10 600 python3 -m pytest tests/test_synthesize.py -x -q -m gpu 2>&1 | tail -5
